timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_vit_gpu.py -m gpu -q -x -k "attention or vit" 2>&1 | tail -4
timeout -k 10 300 python tools/model_times.py vit_small_patch16_224,vit_base_patch16_224 2>&1 | grep -v amdgpu
