cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for R in 1000 300; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$R -- python $GRAFT_REPO_ROOT/bench.py --proposals $R --ways 1 --batch 128 --serial-rpn --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-parity --sustain-seconds 0 > $O/p$R.log 2>&1
cp "$(find /tmp/p$R -name '*kernel_stats.csv' | head -1)" $O/p${R}_kernel_stats.csv
done
