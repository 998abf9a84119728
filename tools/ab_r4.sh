#!/bin/bash
# round-4 A/B lines on one box, interleaved: PW (buffer-load addressing) / dead-block skip / latency split-K
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --sustain-seconds 0"
run() { # tag env... -- args
  tag=$1; shift
  env "$@" > /dev/null 2>&1
}
for i in 1 2; do
  XDET_CONV_PW=0 XDET_CONV_SKIP_DEAD=0 $B --no-ksplit > gpurun_out/ab_base.$i.json 2>/dev/null
  XDET_CONV_PW=1 XDET_CONV_SKIP_DEAD=0 $B --no-ksplit > gpurun_out/ab_pw.$i.json 2>/dev/null
  XDET_CONV_PW=1 XDET_CONV_SKIP_DEAD=1 $B --no-ksplit > gpurun_out/ab_pw_skip.$i.json 2>/dev/null
  XDET_CONV_PW=1 XDET_CONV_SKIP_DEAD=1 $B > gpurun_out/ab_pw_skip_ks.$i.json 2>/dev/null
done
for i in 1 2; do
  for v in "--no-ksplit" ""; do
    python bench.py --batch 1 --steps 300 --warmup 30 --no-cpu-baseline --no-parity --sustain-seconds 0 --no-roofline $v > gpurun_out/ab_b1${v:+_noks}.$i.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_*.json')):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, 'FAILED'); continue
    r = d.get('roofline') or {}
    print('%-36s %8.1f img/s  median %7.3f  min %7.3f ms  frac %s  kernel_ms %s' % (f.split('/')[-1], d['value'], d['median_ms_per_step'], d['min_ms_per_step'], r.get('frac'), r.get('kernel_ms_per_step')))
PY
