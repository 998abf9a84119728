cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
R=${1:-300}
rocprofv3 --kernel-trace --output-format csv -d /tmp/b1_$R -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --proposals $R --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-parity --sustain-seconds 0 > $O/b1_$R.log 2>&1
cp "$(find /tmp/b1_$R -name '*kernel_trace.csv' | head -1)" $O/b1_${R}_trace.csv
