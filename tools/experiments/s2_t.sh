cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_layers.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-parity --no-roofline --batch 1 --steps 300 --warmup 30 --sustain-seconds 0"
for i in 1 2 3; do
  XDET_CONV_DEEP_W8=0 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b1 4 waves', d['value'], d['median_ms_per_step'])"
  $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b1 8 waves', d['value'], d['median_ms_per_step'])"
done
B8="python bench.py --no-cpu-baseline --no-parity --no-roofline --batch 8 --ways 1 --steps 100 --warmup 20 --sustain-seconds 0"
XDET_CONV_DEEP_W8=0 $B8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b8 4 waves', d['value'], d['median_ms_per_step'])"
$B8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b8 8 waves', d['value'], d['median_ms_per_step'])"
