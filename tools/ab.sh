for d in 0 8 0 8; do echo "--- TLXMI_DEBUG=$d"; TLXMI_DEBUG=$d timeout -k 10 100 python tools/conv_micro.py c3x3_56,stem 30; done
