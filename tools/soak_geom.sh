#!/bin/bash
# Geometries and batch sizes the other soaks do not reach: 1280x960 in batches of 32 and 96 (the LDS-mask and the batch growing kernels at that
# size), odd crops in batches of 3 / 24 / 80.  bash tools/soak_geom.sh > gpurun_out/soak_geom.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { timeout -k 10 1000 python tools/soak.py "$@" 2>&1 | grep RESULT | sed "s/^/[$*] /"; }
run --frames 64 --kind lowtex --seed 0x6E0E0000 --chunk 32 --width 1280 --height 960
run --frames 96 --kind std --seed 0x6E0E1000 --chunk 96 --width 1280 --height 960
run --frames 96 --kind std --seed 0x6E0E2000 --chunk 3 --crop 333x517
run --frames 192 --kind lowtex --seed 0x6E0E3000 --chunk 24 --crop 241x323
run --frames 240 --kind std --seed 0x6E0E4000 --chunk 80 --crop 401x599
run --frames 128 --kind lowtex --seed 0x6E0E5000 --chunk 128 --crop 478x640
