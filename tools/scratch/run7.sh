python -m pytest tests/test_gpu_proposals.py -x -q -m gpu 2>&1 | tail -2
python tools/nms_bench.py --time 2>&1 | grep -v "^N=128" 
for b in 1 8 32; do
bash tools/ab.sh -r 2 -l tools/scratch/libxdet_fence.so "--ways 1 --batch $b --steps 100 --warmup 20" "--ways 1 --batch $b --steps 100 --warmup 20"
done
bash tools/ab.sh -r 2 -l tools/scratch/libxdet_fence.so "--ways 1 --batch 32 --proposals 1000 --steps 100 --warmup 20" "--ways 1 --batch 32 --proposals 1000 --steps 100 --warmup 20"
