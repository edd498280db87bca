timeout -k 10 300 python -m pytest tests/test_gemm_gpu.py tests/test_vit_gpu.py -m gpu -q -x -k "linear_ln or vit" 2>&1 | tail -3
for f in 1 0 1 0; do echo "--- TLXMI_LNFUSE=$f"; TLXMI_LNFUSE=$f timeout -k 10 200 python bench.py --workload vit_b16 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*'; done
