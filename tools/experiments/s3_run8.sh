#!/bin/bash
# one-range ring kernel for ~one round of 128 x 128 tiles, chosen per call: parity subset + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s3
timeout 1200 python -m pytest tests/test_gpu_layers.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/s3/pytest_ring.log 2>&1
echo "pytest rc $?" >> gpurun_out/s3/pytest_ring.log
tail -4 gpurun_out/s3/pytest_ring.log
dt() { python bench.py $2 --steps ${3:-20} --warmup 5 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j['ms_per_step'], j.get('median_ms_per_step'))"; }
for r in 1 2; do
  for v in 0 1; do
    XDET_CONV_ONE_RING=$v dt b1_ring$v "--batch 1" 300
    XDET_CONV_ONE_RING=$v dt b3_ring$v "--ways 1 --batch 3" 200
    XDET_CONV_ONE_RING=$v dt b4_ring$v "--ways 1 --batch 4" 200
    XDET_CONV_ONE_RING=$v dt b8_ring$v "--ways 1 --batch 8" 200
    XDET_CONV_ONE_RING=$v dt def_ring$v ""
  done
done
