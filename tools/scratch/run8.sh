python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_layers.py -x -q -m gpu 2>&1 | tail -3
bash tools/ab.sh -r 3 -l tools/scratch/libxdet_base.so "--ways 1 --batch 128" "--ways 1 --batch 128"
bash tools/ab.sh -r 3 -l tools/scratch/libxdet_base.so "" ""
