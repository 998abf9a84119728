#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline quotes (run on the GPU box via gpurun):
#   tools/collect_profile.sh <tag> [bench.py args...]
# 1) --kernel-trace --stats of the bench command, 2)+3) separate --pmc passes for FETCH_SIZE and
# WRITE_SIZE (never combined with other trace domains), then tools/summarize_profile.py -> profiles/.
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-parity $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python $REPO/bench.py $ARGS > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity $* > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity $* > $OUT/${TAG}_write.log 2>&1
cd $REPO
S=$(dirname $(find $OUT/${TAG}_stats -name '*kernel_stats.csv' | head -1))
F=$(dirname $(find $OUT/${TAG}_fetch -name '*counter_collection.csv' | head -1))
W=$(dirname $(find $OUT/${TAG}_write -name '*counter_collection.csv' | head -1))
python tools/summarize_profile.py --stats $S --fetch $F --write $W --tag $TAG \
  --note "rocprofv3 --kernel-trace --stats -- python bench.py $ARGS"
mkdir -p $OUT/profiles && cp profiles/${TAG}_* $OUT/profiles/
tail -3 $OUT/${TAG}_stats.log
