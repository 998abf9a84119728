python -m pytest tests/test_gpu_proposals.py tests/test_gpu_e2e.py tests/test_gpu_range.py -x -q -m gpu 2>&1 | tail -3
bash tools/scratch/prof_r1000.sh
b() { python bench.py $2 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
b "b1 R1000" "--batch 1 --proposals 1000"
b "b1 R300" "--batch 1"
b "R1000" "--proposals 1000"
b "default" ""
