timeout -k 10 300 python -m pytest tests/test_conv_gpu.py tests/test_resnet_gpu.py tests/test_vit_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout -k 10 200 python tools/layer_times.py resnet50 256 2>&1 | tail -1
