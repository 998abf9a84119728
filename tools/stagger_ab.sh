#!/bin/bash
# Same-box A/B of the phase-shifted second stream (bench.py --stagger): medians of 40 steps.   (GPU box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for st in 0 0.25 0.5 0.75; do
  python bench.py --no-cpu-baseline --no-parity --no-roofline --steps 40 --warmup 3 --sustain-seconds 0 --stagger $st 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('stagger $st rep $rep: %8.1f img/s  %8.3f ms/step (median %8.3f, min %8.3f)' % (d['value'], d['ms_per_step'], d['median_ms_per_step'], d['min_ms_per_step']))"
done; done
