timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_vit_gpu.py tests/test_models_gpu.py -m gpu -q -x -k "attention or vit or swin" 2>&1 | tail -3
timeout -k 10 200 python tools/conv_micro.py attn_swin,attn_vit 20 2>&1 | grep -v amdgpu
for w in vit_b16 swin_b; do timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*'; done
