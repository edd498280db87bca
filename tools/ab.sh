timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x 2>&1 | tail -3
echo new; timeout -k 10 200 python tools/ab_tiles.py qkv 8 5 20 2>&1 | grep -v amdgpu
echo prev; TLXMI_LIB=$PWD/tlxcv_amd/libtlxmi_prev.so timeout -k 10 200 python tools/ab_tiles.py qkv 8 5 20 2>&1 | grep -v amdgpu
echo new; timeout -k 10 200 python tools/ab_tiles.py qkv 8 5 20 2>&1 | grep -v amdgpu
