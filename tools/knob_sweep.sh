#!/bin/bash
# One-knob-at-a-time sweep of the overlap policy and the kernel-variant thresholds at the default bench size (run through gpurun).
run() { echo -n "$1: "; env $1 timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run "HVO_NOP=1"
for s in 2 1 0 3 4 6 7; do run "HVO_SCHED=$s"; done
run "HVO_PRIO=0,-1,1"; run "HVO_PRIO=-1,0,1"; run "HVO_PRIO=1,0,-1"
run "HVO_LSD_DENSE=0"; run "HVO_FLOOD_T=256"; run "HVO_FLOOD_T=128"; run "HVO_PEAC_GL=64"; run "HVO_PEAC_GL=32"; run "HVO_ORB_BLUR_LATE=1"; run "HVO_PEAC_EDGES=0"
