#!/bin/bash
# A longer differential soak (fresh seeds, not those of soak_final.sh).  bash tools/soak_long.sh > gpurun_out/soak_long.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { timeout -k 10 900 python tools/soak.py "$@" 2>&1 | grep RESULT | sed "s/^/[$*] /"; }
run --frames 2048 --kind std --seed 0xE00E0000 --chunk 256
run --frames 512 --kind lowtex --seed 0xE00E1000 --chunk 64
run --frames 96 --kind std --seed 0xE00E2000 --chunk 2
run --frames 96 --kind lowtex --seed 0xE00E3000 --chunk 12 --width 1280 --height 960
