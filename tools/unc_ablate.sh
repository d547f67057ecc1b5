for a in 0 16; do echo "ablate $a"; RFN_UNCERT_ABLATE=$a python tools/kbench.py --only uncL1x 2>&1 | grep fused; done
