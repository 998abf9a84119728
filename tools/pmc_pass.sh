#!/bin/bash
# One rocprofv3 counter pass over a short eager bench run: tools/pmc_pass.sh <outdir-name> <counter> [<counter>...]
# ONE counter (or a group known to fit) per pass: an over-subscribed request aborts and hangs.
# (--pmc is only ever combined with --kernel-trace; results land in gpurun_out/<outdir-name>)
set -u
NAME=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/gpurun_out/$NAME -- \
  python $REPO/bench.py --eager --steps 1 --warmup 1 --no-cpu-baseline --no-parity --sustain-seconds 0 > $REPO/gpurun_out/$NAME.log 2>&1
tail -2 $REPO/gpurun_out/$NAME.log | cut -c1-300
