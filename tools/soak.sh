for i in 1 2 3 4 5 6 7 8; do timeout 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -1; done
