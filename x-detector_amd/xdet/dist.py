"""Multi-GPU plumbing: one process per GPU, images sharded by rank, ONE all-gather of the
fixed-size padded detections per step.  The reference has no distributed code
at all (single tf.estimator session); this layer is new.

torch.distributed is used only as the rendezvous + collective transport: backend "nccl" is
RCCL over xGMI on the GPU box, "gloo" on CPU for the world_size-2 tests.  The payload per
image is 20 classes x 200 slots x (score + 4 box coords) x 4 B = 80 KB, so the exchange is
latency-bound (tens of microseconds), never xGMI-link-bound.
"""
import numpy as np


def shard_range(global_batch, rank, world):
    """contiguous block partition of a global batch: rank r owns [lo, hi)."""
    per = global_batch // world
    rem = global_batch % world
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def pack_detections(scores, boxes):
    """scores [B,C,K], boxes [B,C,K,4] -> one contiguous [B,C,K,5] record (score | box)."""
    import torch
    out = torch.empty(tuple(scores.shape) + (5,), dtype=scores.dtype, device=scores.device)
    out[..., 0] = scores
    out[..., 1:] = boxes
    return out


def unpack_detections(packed):
    return packed[..., 0], packed[..., 1:]


def gather_detections(local_packed, world, out=None):
    """all-gather equal-sized per-rank detection records -> [world*B, C, K, 5] on every rank."""
    import torch
    import torch.distributed as dist
    if out is None:
        out = torch.empty((world * local_packed.shape[0],) + tuple(local_packed.shape[1:]),
                          dtype=local_packed.dtype, device=local_packed.device)
    if world == 1 and not dist.is_initialized():
        out.copy_(local_packed)
        return out
    if dist.get_backend() == 'nccl':
        dist.all_gather_into_tensor(out, local_packed.contiguous())
    else:
        parts = list(out.chunk(world, dim=0))
        dist.all_gather(parts, local_packed.contiguous())
    return out


def max_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or 'cpu')
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
