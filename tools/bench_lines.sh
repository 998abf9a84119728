#!/bin/bash
# The bench lines kept under profiles/ (SURVEY 8d timing protocol): default, one stream, single-image latency
# (batch 1), batch 8, ResNet-50 trunk at batch 8 / 32, the VOC-shape uint8 stream, the RCCL path at one rank.
#   tools/bench_lines.sh <tag>        (run on the GPU box; writes gpurun_out/profiles/<tag>_bench_*.json)
set -u
TAG=$1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profiles
mkdir -p $OUT
cd $REPO
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3 --sustain-seconds 0"
$B --ops > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench_ops.txt
$B --no-parity --ways 1 --batch 128 > $OUT/${TAG}_bench_1way_b128.json 2>/dev/null
$B --no-parity --batch 128 > $OUT/${TAG}_bench_2x64.json 2>/dev/null
$B --no-parity --batch 1 --steps 300 --warmup 30 --sustain-seconds 0 > $OUT/${TAG}_bench_b1.json 2>/dev/null
$B --no-parity --batch 1 --steps 300 --warmup 30 --sustain-seconds 0 --ksplit all > $OUT/${TAG}_bench_b1_ksplit_all.json 2>/dev/null
# the reference's own operating point (light_head_rfcn_eval.py:109-111,212): rpn_post_nms_top_n = 1000, one image at a time
$B --no-parity --proposals 1000 --batch 1 --steps 300 --warmup 30 > $OUT/${TAG}_bench_R1000_b1.json 2>/dev/null
$B --no-parity --proposals 1000 > $OUT/${TAG}_bench_R1000.json 2>/dev/null
$B --no-parity --batch 8 --ways 1 --steps 100 --warmup 20 > $OUT/${TAG}_bench_b8.json 2>/dev/null
$B --no-parity --workload resnet50 --batch 8 --steps 100 --warmup 20 --sustain-seconds 0 --ops > $OUT/${TAG}_bench_resnet50_b8.json 2> $OUT/${TAG}_bench_resnet50_b8_ops.txt
$B --no-parity --workload resnet50 --batch 8 --resnet-ways 1 --steps 100 --warmup 20 --sustain-seconds 0 > $OUT/${TAG}_bench_resnet50_b8_1way.json 2>/dev/null
$B --no-parity --workload resnet50 --batch 32 --steps 50 --warmup 10 --sustain-seconds 0 > $OUT/${TAG}_bench_resnet50_b32.json 2>/dev/null
$B --no-parity --voc-stream > $OUT/${TAG}_bench_voc_stream.json 2>/dev/null
$B --no-parity --comm > $OUT/${TAG}_bench_comm_world1.json 2>/dev/null
$B --no-parity --precision f32 --batch 32 --ways 1 > $OUT/${TAG}_bench_f32_b32.json 2>/dev/null
$B --image-size 800 --batch 96 > $OUT/${TAG}_bench_800x800.json 2>/dev/null
for f in $OUT/${TAG}_bench*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get('roofline') or {}
print('%-44s %9.1f img/s  %8.3f ms/step (median %8.3f)  frac %s  whole-step %s' % (
    sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['median_ms_per_step'], r.get('frac'), r.get('frac_whole_step')))
PY
done
