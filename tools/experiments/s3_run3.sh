#!/bin/bash
# 64 x 64 tiles for the small-batch ring kernel: parity subset, then single-image A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s3
timeout 1200 python -m pytest tests/test_gpu_layers.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/s3/pytest_h64.log 2>&1
echo "pytest rc $?" >> gpurun_out/s3/pytest_h64.log
tail -5 gpurun_out/s3/pytest_h64.log
b1() { python bench.py --batch ${2:-1} --steps 300 --warmup 30 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j['ms_per_step'], j.get('median_ms_per_step'))"; }
for r in 1 2 3; do
  XDET_CONV_DEEP_H64=0 b1 h64off
  XDET_CONV_DEEP_H64=1 b1 h64on
done
XDET_CONV_DEEP_H64=0 b1 h64off_b2 2
XDET_CONV_DEEP_H64=1 b1 h64on_b2 2
XDET_CONV_DEEP_H64=0 b1 h64off_b8 8
XDET_CONV_DEEP_H64=1 b1 h64on_b8 8
