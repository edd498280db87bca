python tools/ab_graph.py TLXMI_DEBUG 0,16 vit_b16 256 2>&1 | grep batch
python tools/ab_graph.py TLXMI_DEBUG 0,16 vit_b16 256 2>&1 | grep batch
python tools/ab_graph.py TLXMI_DEBUG 0,16 swin_b 128 2>&1 | grep batch
