cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/s2; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/rn_trace -- python $GRAFT_REPO_ROOT/bench.py --workload resnet50 --batch 8 --resnet-ways 1 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --sustain-seconds 0 > $O/rn_trace.log 2>&1)
python tools/trace_step.py $O/rn_trace 2 > $O/rn_dispatches.txt
head -40 $O/rn_dispatches.txt | cut -c1-150; tail -2 $O/rn_dispatches.txt
rm -rf $O/rn_trace
