#!/bin/bash
# Same-box A/B of a bench.py flag: tools/ab_flag.sh "<flags A>" "<flags B>" [common bench.py args...]      (GPU box)
A=$1; B=$2; shift 2
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for F in "$A" "$B"; do
    python bench.py --no-cpu-baseline --no-parity --no-roofline --steps 15 --warmup 3 --sustain-seconds 0 $F "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('%-28s rep $rep: %8.1f img/s  %8.3f ms/step (median %8.3f)' % ('[$F]', d['value'], d['ms_per_step'], d['median_ms_per_step']))"
  done
done
