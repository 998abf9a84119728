#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s3
timeout 900 python -m pytest tests/test_gpu_proposals.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/s3/pytest_bbe.log 2>&1
echo "pytest rc $?" >> gpurun_out/s3/pytest_bbe.log
tail -4 gpurun_out/s3/pytest_bbe.log
python tools/ubench/bboxes_eval_bench.py
