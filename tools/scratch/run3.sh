python -m pytest tests/test_gpu_psroialign.py -x -q -m gpu 2>&1 | tail -2
b() { python bench.py $2 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['median_ms_per_step'])"; }
b "b1 R1000" "--batch 1 --proposals 1000 --steps 200 --warmup 20"
b "b1 R300" "--batch 1 --steps 200 --warmup 20"
b "b2 R300" "--batch 2 --steps 200 --warmup 20"
b "b8 R300" "--batch 8 --steps 100 --warmup 10"
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for R in 300 1000; do
rocprofv3 --kernel-trace --output-format csv -d /tmp/b1_$R -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --proposals $R --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-parity --sustain-seconds 0 > $O/b1_$R.log 2>&1
cp "$(find /tmp/b1_$R -name '*kernel_trace.csv' | head -1)" $O/b1_${R}_trace.csv
done
