cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s2
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s2/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/s2/pytest.log
tail -3 gpurun_out/s2/pytest.log
RN="python bench.py --workload resnet50 --batch 8 --resnet-ways 1 --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --sustain-seconds 0"
for i in 1 2; do
  XDET_CONV_GBUF=256 $RN 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rn gbuf256', d['value'], d['ms_per_step'])"
  $RN 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rn gbufall', d['value'], d['ms_per_step'])"
done
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'])"
