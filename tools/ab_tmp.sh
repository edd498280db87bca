python tools/ab_graph.py TLXMI_GS_PANEL 0,2,3,4 vit_b16 256 2>&1 | grep batch
python tools/ab_graph.py TLXMI_GS_PANEL 0,2,3,4 swin_b 128 2>&1 | grep batch
python tools/ab_graph.py TLXMI_GS_PANEL 0,2,3,4 resnet50 256 2>&1 | grep batch
bash tools/pmc_traffic.sh vit_b16 > gpurun_out/traffic_vit_b16.log 2>&1; tail -n 1 gpurun_out/traffic_vit_b16.log | cut -c1-400
