cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s2 gpurun_out/profiles
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/s2/pytest_full.log 2>&1; echo "pytest rc $?" >> gpurun_out/s2/pytest_full.log
tail -3 gpurun_out/s2/pytest_full.log
python tools/debug/bneck_timeline.py 8 120 120 > gpurun_out/s2/bneck_timeline.txt 2>&1
bash tools/round_profiles.sh r05d > gpurun_out/s2/round_profiles.log 2>&1
tail -70 gpurun_out/s2/round_profiles.log | cut -c1-200
