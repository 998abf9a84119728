#!/bin/bash
# projection folded into the closing conv: tests, then same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s3
timeout 900 python -m pytest tests/test_gpu_resnet_bneck.py tests/test_gpu_resnet.py -x -q -m gpu > gpurun_out/s3/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/s3/pytest.log
tail -15 gpurun_out/s3/pytest.log
rn() { python bench.py --workload resnet50 --batch 8 --resnet-ways ${2:-1} --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j['ms_per_step'])"; }
for r in 1 2; do
  XDET_RESNET_PROJCAT=0 rn cat0
  XDET_RESNET_PROJCAT=1 rn cat1
done
XDET_RESNET_PROJCAT=0 rn cat0_2way 2
XDET_RESNET_PROJCAT=1 rn cat1_2way 2
O=$GRAFT_REPO_ROOT/gpurun_out
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $O/s3_rn -- python $GRAFT_REPO_ROOT/bench.py --workload resnet50 --batch 8 --resnet-ways 1 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --sustain-seconds 0 > $O/s3/rn.log 2>&1)
python tools/trace_step.py $O/s3_rn 2 > $O/s3/trace_cat1.txt 2>&1
tail -3 $O/s3/trace_cat1.txt
