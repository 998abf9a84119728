b() { python bench.py $2 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['median_ms_per_step'])"; }
for rep in 1 2; do
b "w2 b256" "--ways 2 --batch 256"
b "w3 b384" "--ways 3 --batch 384"
b "w4 b512" "--ways 4 --batch 512"
b "w3 b288" "--ways 3 --batch 288"
b "w4 b256" "--ways 4 --batch 256"
b "w2 b290" "--ways 2 --batch 290"
done
