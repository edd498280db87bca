S=qkv,fc1
for d in 0 4 0 4; do echo "--- tile8 debug $d"; TLXMI_DEBUG=$d TLXMI_TILE=8 timeout -k 10 60 python tools/conv_micro.py $S 20; done
