#!/bin/bash
# 256 x 128 tiles only for layers with at least T1_NK K steps (shorter loops: 128 x 128, two workgroups per CU)
cd "$GRAFT_REPO_ROOT" || exit 1
rn() { python bench.py --workload resnet50 --batch 8 --resnet-ways 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j['ms_per_step'])"; }
dt() { python bench.py $2 --steps ${3:-20} --warmup 5 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j['ms_per_step'], j.get('median_ms_per_step'))"; }
for r in 1 2; do
  for nk in 0 9 13 25 1000; do XDET_CONV_T1_NK=$nk rn rn_nk$nk; done
done
for r in 1 2; do
  for nk in 0 9 1000; do XDET_CONV_T1_NK=$nk dt det_nk$nk ""; done
  for nk in 0 9 1000; do XDET_CONV_T1_NK=$nk dt b8_nk$nk "--batch 8 --ways 1" 200; done
done
