#!/bin/bash
# is the 256 x 256 tile still the right choice?  (temporary knobs, not in the tree)
cd "$GRAFT_REPO_ROOT" || exit 1
dt() { python bench.py $2 --steps ${3:-20} --warmup 5 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j['ms_per_step'], j.get('median_ms_per_step'))"; }
for r in 1 2; do
  dt def_base ""
  XDET_CONV_T2_NEVER=1 dt def_t2never ""
  dt w1_base "--ways 1 --batch 128"
  XDET_CONV_T2_NEVER=1 dt w1_t2never "--ways 1 --batch 128"
  dt b32_base "--ways 1 --batch 32" 50
  XDET_CONV_T2_MIN=256 dt b32_t2min256 "--ways 1 --batch 32" 50
  XDET_CONV_T2_MIN=300 dt b48_t2min300 "--ways 1 --batch 48" 50
  dt b48_base "--ways 1 --batch 48" 50
done
