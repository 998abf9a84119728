b() { python bench.py $2 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['median_ms_per_step'])"; }
b "b1 R1000" "--batch 1 --proposals 1000 --steps 200 --warmup 20"
b "b1 R300" "--batch 1 --steps 200 --warmup 20"
b "b2 R300" "--batch 2 --steps 200 --warmup 20"
b "b8 R300" "--batch 8 --steps 100 --warmup 10"
b "w1 b128" "--ways 1 --batch 128"
b "default" ""
b "default" ""
bash tools/scratch/trace_b1.sh 300
