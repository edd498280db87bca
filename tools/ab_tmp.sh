python tools/ab_graph.py opt:patch_linear 0,1 vit_b16 256 2>&1 | grep batch
python tools/ab_graph.py opt:patch_linear 0,1 vit_b16 256 2>&1 | grep batch
