# A/B on one box: round-2 library vs the current one, same bench command, interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
for rep in 1 2; do
for L in tools/ubench/libxdet_r2.so x-detector_amd/xdet/libxdet_hip.so; do
  tag=$(basename $L .so)
  XDET_LIB=$GRAFT_REPO_ROOT/$L python bench.py --no-cpu-baseline --no-parity --no-roofline --steps 15 > gpurun_out/r03b/ab_${tag}_$rep.json 2>gpurun_out/r03b/ab_${tag}_$rep.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r03b/ab_${tag}_$rep.json'))
print('$tag rep $rep: %.1f img/s  %.3f ms/step median %.3f'%(d['value'],d['ms_per_step'],d['median_ms_per_step']))
PY
done; done
XDET_LIB=$GRAFT_REPO_ROOT/tools/ubench/libxdet_r2.so python bench.py --no-cpu-baseline --no-parity --ways 1 --batch 128 --ops --steps 10 > gpurun_out/r03b/ops_r2.json 2> gpurun_out/r03b/ops_r2.txt
python bench.py --no-cpu-baseline --no-parity --ways 1 --batch 128 --ops --steps 10 > gpurun_out/r03b/ops_new.json 2> gpurun_out/r03b/ops_new.txt
python - <<'PY'
def load(f):
    d={}
    for l in open(f):
        p=l.split()
        if len(p)>=3 and p[-4:-3]==['ms/step'] or 'ms/step' in l:
            try:
                i=p.index('ms/step'); d[' '.join(p[:i-1])]=float(p[i-1])
            except Exception: pass
    return d
a=load('gpurun_out/r03b/ops_r2.txt'); b=load('gpurun_out/r03b/ops_new.txt')
rows=[(k,a[k],b.get(k)) for k in a if b.get(k) is not None and abs(a[k]-b[k])>0.01]
for k,x,y in sorted(rows,key=lambda r:-(abs(r[1]-r[2])))[:25]: print('%-55s %.3f -> %.3f'%(k,x,y))
PY
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
