#!/bin/bash
# The differential soaks (HIP path against the oracle on fresh seeds, tools/soak.py) as named plans; it replaces soak_final.sh, soak_long.sh,
# soak_hunt.sh, soak_async.sh, soak_geom.sh.  Run through gpurun:  bash tools/soak_suite.sh final > gpurun_out/soak_final.txt
#   final  every batch-size class once      long   a longer run of the same classes     hunt   more content classes, the tail stages included
#   async  the batch sizes (1-16 frames) that select the async line growing and the multi-head AHC
#   geom   geometries and batch sizes the others do not reach (1280x960 in batches of 32 / 96, odd crops)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { timeout -k 10 1000 python tools/soak.py "$@" 2>&1 | grep RESULT | sed "s/^/[$*] /"; }
case "$1" in
final)
  run --frames 512 --kind std --seed 0xD00D0000 --chunk 128
  run --frames 128 --kind lowtex --seed 0xD00D1000 --chunk 8
  run --frames 48 --kind std --seed 0xD00D2000 --chunk 1
  run --frames 32 --kind std --seed 0xD00D3000 --chunk 4 --width 1280 --height 960
  run --frames 64 --kind std --seed 0xD00D4000 --chunk 16 --crop 397x501
  run --frames 64 --kind lowtex --seed 0xD00D5000 --chunk 32 ;;
long)
  run --frames 2048 --kind std --seed 0xE00E0000 --chunk 256
  run --frames 512 --kind lowtex --seed 0xE00E1000 --chunk 64
  run --frames 96 --kind std --seed 0xE00E2000 --chunk 2
  run --frames 96 --kind lowtex --seed 0xE00E3000 --chunk 12 --width 1280 --height 960 ;;
hunt)
  run --frames 256 --kind std --seed 0xF00F0000 --chunk 64 --tail
  run --frames 256 --kind lowtex --seed 0xF00F1000 --chunk 64 --tail
  run --frames 2048 --kind lowtex --seed 0xF00F2000 --chunk 256
  run --frames 2048 --kind std --seed 0xF00F3000 --chunk 256
  run --frames 128 --kind lowtex --seed 0xF00F4000 --chunk 16 --crop 479x638
  run --frames 64 --kind std --seed 0xF00F5000 --chunk 8 --width 1280 --height 960 --tail ;;
async)
  run --frames 512 --kind lowtex --seed 0xA5A50000 --chunk 8
  run --frames 512 --kind std --seed 0xA5A51000 --chunk 8
  run --frames 512 --kind lowtex --seed 0xA5A52000 --chunk 16
  run --frames 256 --kind std --seed 0xA5A53000 --chunk 16
  run --frames 128 --kind lowtex --seed 0xA5A54000 --chunk 1
  run --frames 256 --kind lowtex --seed 0xA5A55000 --chunk 5 --crop 479x638
  run --frames 64 --kind lowtex --seed 0xA5A56000 --chunk 2 --width 1280 --height 960
  run --frames 48 --kind std --seed 0xA5A57000 --chunk 8 --width 1280 --height 960 ;;
geom)
  run --frames 64 --kind lowtex --seed 0x6E0E0000 --chunk 32 --width 1280 --height 960
  run --frames 96 --kind std --seed 0x6E0E1000 --chunk 96 --width 1280 --height 960
  run --frames 96 --kind std --seed 0x6E0E2000 --chunk 3 --crop 333x517
  run --frames 192 --kind lowtex --seed 0x6E0E3000 --chunk 24 --crop 241x323
  run --frames 240 --kind std --seed 0x6E0E4000 --chunk 80 --crop 401x599
  run --frames 128 --kind lowtex --seed 0x6E0E5000 --chunk 128 --crop 478x640 ;;
*) echo "usage: $0 final|long|hunt|async|geom"; exit 2 ;;
esac
