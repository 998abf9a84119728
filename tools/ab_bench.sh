#!/bin/bash
# Same-box A/B of two builds of libxdet_hip.so (box-to-box spread of one binary is ~+-3 %, larger than most kernel changes):
#   tools/ab_bench.sh <other_build.so> [bench.py args...]      (run on the GPU box)
# runs bench.py alternately with XDET_LIB=<other_build.so> and with the in-tree library, twice each.  An older build that
# lacks newer C-ABI entries still loads (xdet/_lib.py skips missing symbols when XDET_LIB is set).
set -u
OTHER=$1; shift
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for L in "$OTHER" x-detector_amd/xdet/libxdet_hip.so; do
    XDET_LIB=$(realpath $L) python bench.py --no-cpu-baseline --no-parity --no-roofline --steps 15 --warmup 3 "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('%-48s rep $rep: %8.1f img/s  %8.3f ms/step (median %8.3f)' % ('$L', d['value'], d['ms_per_step'], d['median_ms_per_step']))"
  done
done
