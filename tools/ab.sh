#!/bin/bash
# Same-box A/B (box-to-box spread of one binary is ~+-3 %, larger than most kernel changes): two bench.py configurations run
# alternately, REPS times each, inside one gpurun call.        (run on the GPU box)
#   tools/ab.sh [-r REPS] [-l other_build.so] "<bench.py flags A>" "<bench.py flags B>" [common bench.py flags...]
# -l: configuration B runs with XDET_LIB=<other_build.so> (a library built from another tree; xdet/_lib.py skips C-ABI entries
#     an older build lacks); without it both run the in-tree library and differ in their flags only
#     (e.g. tools/ab.sh "--ways 1" "--ways 2";  tools/ab.sh "--pool-sub on" "--pool-sub off" --batch 8).
set -u
REPS=2; OTHER=""
while [ $# -gt 2 ]; do
  case $1 in -r) REPS=$2; shift 2;; -l) OTHER=$(realpath $2); shift 2;; *) break;; esac
done
A=$1; B=$2; shift 2
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in $(seq $REPS); do
  for which in A B; do
    if [ $which = A ]; then F=$A; L=""; else F=$B; L=$OTHER; fi
    env ${L:+XDET_LIB=$L} python bench.py --no-cpu-baseline --no-parity --no-roofline --steps 15 --warmup 3 --sustain-seconds 0 $F "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%s %-40s rep $rep: %8.1f img/s  %8.3f ms/step (median %8.3f)' % ('$which', '[$F]${L:+ lib=other}', d['value'], d['ms_per_step'], d['median_ms_per_step']))"
  done
done
