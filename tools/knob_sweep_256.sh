#!/bin/bash
run() { echo -n "$1: "; env $1 timeout -k 10 300 python bench.py --config batch256 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('latency_ms'))"; }
run "HVO_NOP=1"; run "HVO_SCHED=0"; run "HVO_SCHED=5"; run "HVO_SCHED=7"; run "HVO_SCHED=2"; run "HVO_PRIO=0,0,0"; run "HVO_FLOOD_T=128"; run "HVO_PEAC_HEADS=2"; run "HVO_PEAC_HEADS=3"; run "HVO_LSD_LAT=1"
