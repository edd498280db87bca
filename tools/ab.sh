for c in 1 0 1 0; do echo "comb=$c"; TLXMI_ATTN_COMB=$c timeout -k 10 200 python bench.py --workload swin_b --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*'; done
timeout -k 10 200 python tools/conv_micro.py attn_swin 20 2>&1 | grep -v amdgpu
