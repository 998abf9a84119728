#!/bin/bash
# per-dispatch comparison of the 64 x 64 form at batch 1 and 2
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O/s3
for b in 1 2; do for h in 0 1; do
  (cd /tmp && export TMPDIR=/tmp && XDET_CONV_DEEP_H64=$h rocprofv3 --kernel-trace --output-format csv -d $O/s3_b${b}_h${h} -- python $GRAFT_REPO_ROOT/bench.py --batch $b --serial-rpn --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-parity --sustain-seconds 0 > $O/s3/b${b}_h${h}.log 2>&1)
  python tools/trace_step.py $O/s3_b${b}_h${h} 2 > $O/s3/trace_b${b}_h${h}.txt 2>&1
  tail -1 $O/s3/trace_b${b}_h${h}.txt
done; done
