export TLXMI_FORCE="25216:768:2304:1:1=11,25216:768:768:1:1=11,25216:768:3072:1:1=11,25216:3072:768:1:1=11"
python tools/ab_graph.py TLXMI_W4_DBG 0,16,32,128 vit_b16 256
export TLXMI_FORCE="25216:768:2304:1:1=11"
python tools/ab_graph.py TLXMI_W4_DBG 0,32 vit_b16 256
export TLXMI_FORCE="25216:768:3072:1:1=11"
python tools/ab_graph.py TLXMI_W4_DBG 0,32 vit_b16 256
unset TLXMI_FORCE
python tools/ab_graph.py TLXMI_W4_DBG 0 vit_b16 256
