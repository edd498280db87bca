echo "--- halo tests"; timeout -k 10 300 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "thin_input or resnet50_conv" 2>&1 | tail -4
for h in 0 1; do echo "--- TLXMI_HALO=$h"; TLXMI_HALO=$h timeout -k 10 100 python tools/conv_micro.py c3x3_56,stem 20; done
