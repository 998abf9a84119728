#!/bin/bash
# round-4 A/B, second set: fork of the RPN branch in front of the exit flow, split-K levels (off / on = head only / all)
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --sustain-seconds 0"
for i in 1 2; do
  for K in off on all; do
    $B --ksplit $K > gpurun_out/ab5_big_$K.$i.json 2>/dev/null
    python bench.py --batch 1 --steps 300 --warmup 30 --no-cpu-baseline --no-parity --sustain-seconds 0 --no-roofline --ksplit $K > gpurun_out/ab5_b1_$K.$i.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab5_*.json')):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, 'FAILED'); continue
    print('%-36s %8.1f img/s  median %7.3f  min %7.3f ms' % (f.split('/')[-1], d['value'], d['median_ms_per_step'], d['min_ms_per_step']))
PY
