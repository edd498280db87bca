for f in 1 0 1 0; do echo "--- TLXMI_LNFUSE=$f"; TLXMI_LNFUSE=$f timeout -k 10 200 python bench.py --workload vit_b16 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*'; done
