#!/bin/bash
# Everything kept under profiles/ for a round: rocprofv3 stats + PMC passes of the profile configuration, the bench lines,
# the default line with the CPU leg, and config 4 at world 1 (comm + VOC stream).   tools/round_profiles.sh <tag>   (GPU box)
cd $GRAFT_REPO_ROOT
TAG=${1:-r03a}
bash tools/collect_profile.sh ${TAG} --ways 1 --batch 128 > /dev/null 2>&1
bash tools/bench_lines.sh ${TAG}
python bench.py > gpurun_out/profiles/${TAG}_bench_default_full.json 2> /dev/null
python bench.py --no-cpu-baseline --no-parity --comm --voc-stream > gpurun_out/profiles/${TAG}_bench_comm_voc_world1.json 2>/dev/null
ls gpurun_out/profiles | grep ${TAG} | head -40
