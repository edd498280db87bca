python tools/ab_graph.py TLXMI_HALFTAIL 0,1,2 vit_b16 256 2>&1 | grep batch
python tools/ab_graph.py TLXMI_HALFTAIL 0,1,2 swin_b 128 2>&1 | grep batch
python tools/ab_graph.py TLXMI_PLAN_CUS 0,112,128,152,160,192,256 vit_b16 256 2>&1 | grep batch
