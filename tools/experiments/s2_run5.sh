cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s2
timeout 600 python -m pytest tests/test_gpu_resnet_bneck.py tests/test_gpu_resnet.py -x -q > gpurun_out/s2/pytest4.log 2>&1; echo "pytest rc $?" >> gpurun_out/s2/pytest4.log
tail -4 gpurun_out/s2/pytest4.log | cut -c1-300
RN="timeout 300 python bench.py --workload resnet50 --batch 8 --resnet-ways 1 --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --sustain-seconds 0"
for i in 1 2; do
  XDET_RESNET_PRECONV=0 $RN 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rn 3-launch', d['value'], d['ms_per_step'])"
  $RN 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rn fused   ', d['value'], d['ms_per_step'])"
done
RN2="timeout 300 python bench.py --workload resnet50 --batch 8 --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --sustain-seconds 0"
XDET_RESNET_PRECONV=0 $RN2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rn 2-way old', d['value'], d['ms_per_step'])"
$RN2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rn 2-way new', d['value'], d['ms_per_step'])"
