timeout -k 10 600 python -m pytest tests/test_resnet_gpu.py -m gpu -q -x 2>&1 | tail -3
for f in 1 0 1 0; do echo "--- TLXMI_SIDE_STREAM=$f"; TLXMI_SIDE_STREAM=$f timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*'; done
