#!/bin/bash
# tools/micro/corr_ks.sh -- the in-workgroup channel split of the smaller correlation levels (corr9_pipe2_kernel<8,32,..,KSPLIT>:
# two 3-wave groups per tile, each over half the channels: RFN_CORR_VARIANT=47; default for <= 256 tiles) against single 3-wave workgroups
# (RFN_CORR_VARIANT=41): parity + repeat checks, kbench rows L2 / L3 / K2-L1 alternating, what the ablations leave, race check
cd "$(dirname "$0")/../.."
python -m pytest tests/test_ops_gpu.py tests/test_race_gpu.py tests/test_align_gpu.py -x -q -m gpu -k "corr or correlation or local or align" 2>&1 | tail -3
for v in 47 41 47 41 47 41; do echo "== RFN_CORR_VARIANT=$v"; RFN_CORR_VARIANT=$v python tools/kbench.py --only L2,L3,K2-L1 2>&1 | grep "corr9 +relu+l2norm\|corr9 raw" | grep -v "L1 "; done
for v in 47 41; do for a in 1 4 5; do echo "== RFN_CORR_VARIANT=$v RFN_CORR_ABLATE=$a (bit 0 no DMA, bit 2 no stores)"; RFN_CORR_VARIANT=$v RFN_CORR_ABLATE=$a python tools/kbench.py --only L2 2>&1 | grep "corr9 +relu+l2norm"; done; done
python tools/micro/corr_race.py 300 2 256 135 240
python tools/micro/corr_race.py 300 2 128 128 128
for v in 41 47 41 47; do RFN_CORR_VARIANT=$v python tools/micro/corr_ks_shapes.py 2>&1 | grep variant; done
refign_amd/lib/ab/wg_placement
