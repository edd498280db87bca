# gemm_w4 (candidate 11) on all four ViT-B/16 Linear layers of a two-stream forward (hipGraph replay, tuning flavour): whole, stores
# dropped (16), no epilogue / drain at all (32) — the bound of what hiding the epilogue can give.
export TLXMI_FORCE="25216:768:2304:1:1=11,25216:768:768:1:1=11,25216:768:3072:1:1=11,25216:3072:768:1:1=11"
python tools/ab_graph.py TLXMI_W4_DBG 0,16,32 vit_b16 256
unset TLXMI_FORCE
python tools/ab_graph.py TLXMI_W4_DBG 0 vit_b16 256
