#!/bin/bash
# A wider hunt after the sqrt finding: more content classes, the tail stages included.  bash tools/soak_hunt.sh > gpurun_out/soak_hunt.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { timeout -k 10 1000 python tools/soak.py "$@" 2>&1 | grep RESULT | sed "s/^/[$*] /"; }
run --frames 256 --kind std --seed 0xF00F0000 --chunk 64 --tail
run --frames 256 --kind lowtex --seed 0xF00F1000 --chunk 64 --tail
run --frames 2048 --kind lowtex --seed 0xF00F2000 --chunk 256
run --frames 2048 --kind std --seed 0xF00F3000 --chunk 256
run --frames 128 --kind lowtex --seed 0xF00F4000 --chunk 16 --crop 479x638
run --frames 64 --kind std --seed 0xF00F5000 --chunk 8 --width 1280 --height 960 --tail
