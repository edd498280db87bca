export TLXMI_GELU_STREAM=1
python tools/ab_graph.py TLXMI_DEBUG 0,16,2 vit_b16 256 2>&1 | grep batch
