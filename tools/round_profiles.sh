#!/bin/bash
# Everything kept under profiles/ for a round: rocprofv3 stats + PMC passes of the profile configuration (one sub-batch stream of
# 128 images), the bench lines, the default line with the CPU leg, config 4 at world 1 (comm + VOC stream), the single-image
# run (bench line + rocprofv3 kernel stats + one step's dispatch list), BASELINE config 2 (ResNet-50, batch 8: stats + PMC +
# one step's dispatch list) and config 5's shape (800 x 800, batch 96: stats + PMC).   tools/round_profiles.sh <tag>   (GPU box)
# Run it LAST: tests/test_profiles_match_sources.py fails when the newest summary does not match the conv kernel sources.
cd $GRAFT_REPO_ROOT
TAG=${1:-r04a}
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
bash tools/collect_profile.sh ${TAG} --ways 1 --batch 128 > /dev/null 2>&1
bash tools/bench_lines.sh ${TAG}
python bench.py > $O/profiles/${TAG}_bench_default_full.json 2> /dev/null
python bench.py --no-cpu-baseline --no-parity --comm --voc-stream > $O/profiles/${TAG}_bench_comm_voc_world1.json 2>/dev/null
# single image: kernel stats + the dispatch list of one step (RPN branch on the main stream: durations without sharing)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_b1 -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --serial-rpn --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-parity --sustain-seconds 0 > $O/${TAG}_b1.log 2>&1)
cp "$(find $O/${TAG}_b1 -name '*kernel_stats.csv' | head -1)" $O/profiles/${TAG}_batch1_kernel_stats.csv
python tools/trace_step.py $O/${TAG}_b1 2 > $O/profiles/${TAG}_batch1_step_dispatches.txt
# the reference's operating point (rpn_post_nms_top_n = 1000, light_head_rfcn_eval.py:109-111): one stream of 128 images --
# kernel stats + PMC summary (the head and PsRoiAlign grow 3.3x) -- and the single image: kernel stats + one step's dispatch list
bash tools/collect_profile.sh ${TAG}_R1000 --proposals 1000 --ways 1 --batch 128 > /dev/null 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_R1000_b1 -- python $GRAFT_REPO_ROOT/bench.py --proposals 1000 --batch 1 --serial-rpn --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-parity --sustain-seconds 0 > $O/${TAG}_R1000_b1.log 2>&1)
cp "$(find $O/${TAG}_R1000_b1 -name '*kernel_stats.csv' | head -1)" $O/profiles/${TAG}_R1000_batch1_kernel_stats.csv
python tools/trace_step.py $O/${TAG}_R1000_b1 2 > $O/profiles/${TAG}_R1000_batch1_step_dispatches.txt
# the single image as it runs (RPN branch on its side stream, graph replay): one step's dispatches with their queues
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_b1c -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-parity --sustain-seconds 0 > $O/${TAG}_b1c.log 2>&1)
python tools/trace_step.py $O/${TAG}_b1c 2 > $O/profiles/${TAG}_batch1_two_streams_step_dispatches.txt
# the proposal stage alone: exactness at every cluster size + time per call on overlap-heavy / spread inputs
python tools/nms_bench.py > $O/profiles/${TAG}_nms_bench.txt 2>&1
# BASELINE config 2 and config 5's shape: stats + PMC summaries, as the default configuration's
bash tools/collect_profile.sh ${TAG}_resnet_b8 --workload resnet50 --batch 8 --resnet-ways 1 > /dev/null 2>&1
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_rn -- python $GRAFT_REPO_ROOT/bench.py --workload resnet50 --batch 8 --resnet-ways 1 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --sustain-seconds 0 > $O/${TAG}_rn.log 2>&1)
python tools/trace_step.py $O/${TAG}_rn 2 > $O/profiles/${TAG}_resnet_b8_step_dispatches.txt
bash tools/collect_profile.sh ${TAG}_800x800 --image-size 800 --ways 1 --batch 48 > /dev/null 2>&1
ls $O/profiles | grep ${TAG} | head -60
