cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s2
timeout 600 python -m pytest tests/test_gpu_resnet_bneck.py -q -k "op_is" -s > gpurun_out/s2/pytest3.log 2>&1; echo "pytest rc $?" >> gpurun_out/s2/pytest3.log
grep -E "mismatch|axis|passed|failed|rc|Error" gpurun_out/s2/pytest3.log | cut -c1-300 | head -60
