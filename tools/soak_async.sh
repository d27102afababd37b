#!/bin/bash
# The batch sizes that select the async line growing and the multi-head AHC (1-16 frames), fresh seeds.  bash tools/soak_async.sh > gpurun_out/soak_async.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { timeout -k 10 1000 python tools/soak.py "$@" 2>&1 | grep RESULT | sed "s/^/[$*] /"; }
run --frames 512 --kind lowtex --seed 0xA5A50000 --chunk 8
run --frames 512 --kind std --seed 0xA5A51000 --chunk 8
run --frames 512 --kind lowtex --seed 0xA5A52000 --chunk 16
run --frames 256 --kind std --seed 0xA5A53000 --chunk 16
run --frames 128 --kind lowtex --seed 0xA5A54000 --chunk 1
run --frames 256 --kind lowtex --seed 0xA5A55000 --chunk 5 --crop 479x638
run --frames 64 --kind lowtex --seed 0xA5A56000 --chunk 2 --width 1280 --height 960
run --frames 48 --kind std --seed 0xA5A57000 --chunk 8 --width 1280 --height 960
