#!/bin/bash
# kernel choice for the folded closing convs (longer K loops than the plain closing convs): one-range ring kernel up to 512 tiles?
cd "$GRAFT_REPO_ROOT" || exit 1
rn() { python bench.py --workload resnet50 --batch 8 --resnet-ways ${2:-1} --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j['ms_per_step'])"; }
for r in 1 2; do
  rn base
  XDET_KSPLIT_ONE_TILES=512 rn one512
  XDET_KSPLIT_ONE_TILES=1024 rn one1024
done
