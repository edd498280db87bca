timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_vit_gpu.py -m gpu -q -x -k "attention or vit" 2>&1 | tail -6
