python -m pytest tests/test_ops_gpu.py -q -k "patch_embedding" 2>&1 | tail -n 5
python -m pytest tests/test_swin_gpu.py tests/test_models_gpu.py -q -k "swin" 2>&1 | tail -n 3
python tools/ab_graph.py opt:patch_embed4 0,1 swin_b 128 2>&1 | grep batch
python tools/ab_graph.py opt:patch_embed4 0,1 swin_b 128 2>&1 | grep batch
