#!/bin/bash
# 128 x 128 tiles instead of 256 x 128: other configurations, same box
cd "$GRAFT_REPO_ROOT" || exit 1
dt() { python bench.py $2 --steps ${3:-20} --warmup 5 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j['ms_per_step'], j.get('median_ms_per_step'))"; }
for r in 1 2; do
  for nk in 0 1073741824; do
    XDET_CONV_T1_NK=$nk dt w1b128_nk$nk "--ways 1 --batch 128"
    XDET_CONV_T1_NK=$nk dt 800_nk$nk "--image-size 800 --batch 96" 10
    XDET_CONV_T1_NK=$nk dt b32_nk$nk "--ways 1 --batch 32" 50
    XDET_CONV_T1_NK=$nk dt b2_nk$nk "--ways 1 --batch 2" 200
    XDET_CONV_T1_NK=$nk dt b4_nk$nk "--ways 1 --batch 4" 200
    XDET_CONV_T1_NK=$nk dt rn32_nk$nk "--workload resnet50 --batch 32" 50
  done
done
