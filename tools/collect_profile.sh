#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline quotes (run on the GPU box via gpurun):
#   tools/collect_profile.sh <tag> [bench.py args...]
# 1) --kernel-trace --stats of the bench command; 2)-5) separate --pmc passes (never combined with other trace
# domains) for FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE over one eager step;
# then tools/summarize_profile.py -> profiles/<tag>_*.
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 --serial-rpn $*"
PARGS="--eager --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 --serial-rpn $*"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python $REPO/bench.py $ARGS > $OUT/${TAG}_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$C -- python $REPO/bench.py $PARGS > $OUT/${TAG}_pmc_$C.log 2>&1
done
cd $REPO
d() { dirname "$(find $OUT/$1 -name "*$2" | head -1)"; }
python tools/summarize_profile.py --stats $(d ${TAG}_stats kernel_stats.csv) --fetch $(d ${TAG}_pmc_FETCH_SIZE counter_collection.csv) \
  --write $(d ${TAG}_pmc_WRITE_SIZE counter_collection.csv) --mfma $(d ${TAG}_pmc_SQ_VALU_MFMA_BUSY_CYCLES counter_collection.csv) \
  --active $(d ${TAG}_pmc_GRBM_GUI_ACTIVE counter_collection.csv) --tag $TAG \
  --note "rocprofv3 --kernel-trace --stats -- python bench.py $ARGS" > $OUT/${TAG}_summary.log 2>&1
mkdir -p $OUT/profiles && cp profiles/${TAG}_* $OUT/profiles/
tail -3 $OUT/${TAG}_stats.log | cut -c1-300
