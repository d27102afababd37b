#!/bin/bash
# One script for every knob sweep (it replaces the round-2..4 one-off shells: knob_sweep*.sh, sched_sweep.sh, b32_*_sweep.sh, lsd_async_sweep.sh).
# Run through gpurun from the repo root.  Every setting is one run of the command with that environment; settings are quoted strings of VAR=value pairs.
#
#   bash tools/sweep.sh bench   [bench.py args ...] -- "HVO_SCHED=2" "HVO_SCHED=5 HVO_PRIO=0,0,0" ...     -> value, ms_per_step, kernel times
#   bash tools/sweep.sh latency [latency.py args ...] -- "HVO_FLOOD_T=256 HVO_PEAC_HEADS=3" ...          -> B1 / B32 stage times
#   bash tools/sweep.sh preset NAME        NAME = sched | knobs | knobs1280 | knobs256 | b32planes | b32async   (the sweeps the old shells ran)
# A setting of "-" runs the defaults.  Tuning variables are read when a context builds its plans, so every run is a fresh process.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mode=$1; shift
if [ "$mode" = preset ]; then
  case "$1" in
    sched)     exec bash $0 bench --steps 6 --warmup 2 -- "HVO_SCHED=5 HVO_PRIO=0,0,0" "HVO_SCHED=2 HVO_PRIO=0,0,0" "HVO_SCHED=1 HVO_PRIO=0,-1,1" "HVO_SCHED=0" "HVO_SCHED=3" "HVO_SCHED=4" "HVO_SCHED=6" "HVO_SCHED=7" "HVO_SCHED=5 HVO_PRIO=0,-1,1" "HVO_SCHED=5 HVO_PRIO=-1,0,1" ;;
    knobs)     exec bash $0 bench --steps 6 --warmup 2 -- - "HVO_LSD_DENSE=0" "HVO_FLOOD_T=256" "HVO_FLOOD_T=128" "HVO_PEAC_GL=64" "HVO_PEAC_GL=32" "HVO_ORB_BLUR_LATE=1" "HVO_PEAC_EDGES=0" ;;
    knobs1280) exec bash $0 bench --config big1280 --steps 3 --warmup 1 -- - "HVO_FLOOD_T=64" "HVO_FLOOD_T=128" "HVO_LSD_DENSE=1" "HVO_PEAC_GL=32" "HVO_SCHED=0" "HVO_SCHED=1" "HVO_SCHED=2" "HVO_SCHED=7" "HVO_PRIO=0,-1,1" ;;
    knobs256)  exec bash $0 bench --config batch256 --steps 20 --warmup 3 -- - "HVO_SCHED=0" "HVO_SCHED=5" "HVO_SCHED=7" "HVO_SCHED=2" "HVO_PRIO=0,0,0" "HVO_FLOOD_T=128" "HVO_PEAC_HEADS=2" "HVO_PEAC_HEADS=3" "HVO_LSD_LAT=1" ;;
    b32planes) exec bash $0 latency std 640 480 1,32 -- "HVO_FLOOD_T=256 HVO_PEAC_HEADS=3" "HVO_FLOOD_T=512 HVO_PEAC_HEADS=3" "HVO_FLOOD_T=512 HVO_PEAC_HEADS=4" "HVO_FLOOD_T=256 HVO_PEAC_HEADS=4" "HVO_FLOOD_T=512 HVO_PEAC_HEADS=2" ;;
    b32async)  exec bash $0 latency std 640 480 32 -- "HVO_LSD_ASYNC=0" "HVO_LSD_ASYNC=8 HVO_LSD_ASYNC_LDS=57344" "HVO_LSD_ASYNC=8 HVO_LSD_ASYNC_LDS=0" "HVO_LSD_ASYNC=12 HVO_LSD_ASYNC_LDS=57344" "HVO_LSD_ASYNC=16 HVO_LSD_ASYNC_LDS=57344" "HVO_LSD_ASYNC=6 HVO_LSD_ASYNC_LDS=57344" ;;
    *) echo "unknown preset $1"; exit 2 ;;
  esac
fi
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
for setting in "$@"; do
  [ "$setting" = "-" ] && setting="HVO_NOP=1"
  echo -n "[$setting] "
  if [ "$mode" = bench ]; then
    env $setting timeout -k 10 400 python bench.py --no-cpu-baseline --no-extras "${args[@]}" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['frames_per_gpu'], d['value'], d['ms_per_step'], {k: round(v, 1) for k, v in d['kernel_ms_per_step_serialised'].items()})"
  else
    env $setting timeout -k 10 300 python tools/latency.py "${args[@]}" 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); print({b: {x: d[b][x] for x in ('orb_ms', 'lsd_ms', 'planes_ms', 'all_ms') if x in d[b]} for b in d if b.startswith('B')})"
  fi
done
