#!/bin/bash
# The differential soak of the round's final build: fresh seeds, every batch size class (resident batch, async lines + queue heads for
# 8 frames and for a lone frame, 1280x960, an odd geometry).  bash tools/soak_final.sh > gpurun_out/soak_final.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { timeout -k 10 500 python tools/soak.py "$@" 2>&1 | grep RESULT | sed "s/^/[$*] /"; }
run --frames 512 --kind std --seed 0xD00D0000 --chunk 128
run --frames 128 --kind lowtex --seed 0xD00D1000 --chunk 8
run --frames 48 --kind std --seed 0xD00D2000 --chunk 1
run --frames 32 --kind std --seed 0xD00D3000 --chunk 4 --width 1280 --height 960
run --frames 64 --kind std --seed 0xD00D4000 --chunk 16 --crop 397x501
run --frames 64 --kind lowtex --seed 0xD00D5000 --chunk 32
