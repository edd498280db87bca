python tools/ab_graph.py TLXMI_ATTN_EVEN 1,0 vit_b16 256 2>&1 | grep batch
python tools/ab_graph.py TLXMI_ATTN_EVEN 1,0 vit_b16 256 2>&1 | grep batch
