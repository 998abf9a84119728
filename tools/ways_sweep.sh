cd /root/repo
for rep in 1 2; do for cfg in "2 256" "3 288" "3 384" "4 512" "2 384"; do set -- $cfg; python bench.py --no-cpu-baseline --no-parity --no-roofline --steps 10 --warmup 3 --sustain-seconds 0 --ways $1 --batch $2 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('ways $1 batch $2 rep $rep: %8.1f img/s  %8.3f ms/step' % (d['value'], d['ms_per_step']))"; done; done
