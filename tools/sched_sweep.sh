for cfg in "5 0,0,0" "2 0,0,0" "1 0,-1,1" "0 0,0,0" "3 0,0,0" "4 0,0,0" "6 0,0,0" "7 0,0,0" "5 0,-1,1" "5 -1,0,1"; do
  set -- $cfg
  echo -n "sched=$1 prio=$2: "
  HVO_SCHED=$1 HVO_PRIO=$2 timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
