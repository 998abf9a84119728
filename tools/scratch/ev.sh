for s in 0 1 2 3 4 5; do echo stop $s; DBG_STOP=$s python tools/bboxes_eval_bench.py 2>&1 | grep "N=128"; done
