python -m pytest tests/test_ops_gpu.py -q -k "patch_merge or layernorm or window" 2>&1 | tail -n 3
python -m pytest tests/test_models_gpu.py -q -k "swin" 2>&1 | tail -n 3
