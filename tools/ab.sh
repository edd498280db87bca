timeout -k 10 400 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -m gpu -q -x -k "window or swin or layernorm" 2>&1 | tail -6
timeout -k 10 200 python bench.py --workload swin_b --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*'
