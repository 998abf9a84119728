#!/bin/bash
# Same-box A/B of ONE library under two environments (a kernel form selected by an environment variable):
#   tools/ab_env.sh "<VAR=value ...>" [bench.py args...]      (run on the GPU box)
# runs bench.py alternately with and without the assignment, twice each.
set -u
ENVA=$1; shift
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for E in "$ENVA" "XDET_AB_NONE=1"; do
    env $E python bench.py --no-cpu-baseline --no-parity --no-roofline --steps 15 --warmup 3 --sustain-seconds 0 "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('%-32s rep $rep: %8.1f img/s  %8.3f ms/step (median %8.3f)' % ('$E', d['value'], d['ms_per_step'], d['median_ms_per_step']))"
  done
done
