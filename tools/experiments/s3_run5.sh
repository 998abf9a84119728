#!/bin/bash
# tile choice for grids of about one round of 256 x 128 tiles: 128 x 128 (two workgroups per CU) instead?
cd "$GRAFT_REPO_ROOT" || exit 1
rn() { python bench.py --workload resnet50 --batch 8 --resnet-ways ${2:-1} --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j['ms_per_step'])"; }
for r in 1 2; do
  rn base
  XDET_CONV_T1_MIN=240 rn t240
  XDET_CONV_T1_MIN=460 rn t460
  XDET_CONV_T1_MIN=100000 rn tnever
done
