timeout -k 10 300 python3 tools/layer_times.py vit_b16 256 > gpurun_out/layers_vit_b16.txt 2>&1
timeout -k 10 300 python3 tools/layer_times.py swin_b 128 > gpurun_out/layers_swin_b.txt 2>&1
tail -2 gpurun_out/layers_swin_b.txt
