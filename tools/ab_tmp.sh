python tools/ab_graph.py opt:tail_splitk 0,1 vit_b16 256 2>&1 | grep batch
python tools/ab_graph.py opt:tail_splitk 0,1 swin_b 128 2>&1 | grep batch
