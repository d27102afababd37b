#!/bin/bash
run() { echo -n "$1: "; env $1 timeout -k 10 300 python bench.py --config big1280 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run "HVO_NOP=1"; run "HVO_FLOOD_T=128"; run "HVO_FLOOD_T=64"; run "HVO_SCHED=0"; run "HVO_SCHED=2"; run "HVO_PRIO=0,-1,1"
