cd /tmp; export TMPDIR=/tmp
for B in 24 48 128; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mall_b$B -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-roofline --sustain-seconds 0 --serial-rpn --ways 1 --batch $B > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/mall_b$B -name "*kernel_stats.csv" | head -1)
  python - "$f" $B <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1]))); B=int(sys.argv[2])
want=['depthwise3x3_tile_kernel<1, true>','conv_dma_f16_kernel<256, 256, 2, 4, 3, 2, true, false>','conv_dma_f16_kernel<256, 128, 4, 2, 3, 2, true, false>','depthwise3x3_tile_kernel<2, true>','maxpool_v3s2_add_kernel','sepconv_pc_kernel<true, false, true, 2>']
tot=sum(float(r['TotalDurationNs']) for r in rows if 'rocclr' not in r['Name'])
print('B=%d total kernel us per image (8 steps): %.1f' % (B, tot/1e3/8/B))
for r in rows:
    for w in want:
        if w in r['Name']:
            print('   %-62s calls %4s avg %8.1f us  per image-call %6.2f us' % (w, r['Calls'], float(r['AverageNs'])/1e3, float(r['AverageNs'])/1e3/B))
PY
done
