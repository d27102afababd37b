for cfg in "0 0" "8 57344" "8 0" "12 57344" "16 57344" "6 57344"; do set -- $cfg
  echo "== HVO_LSD_ASYNC=$1 LDS=$2"; HVO_LSD_ASYNC=$1 HVO_LSD_ASYNC_LDS=$2 python tools/latency.py std 640 480 32 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print({x:d['B32'][x] for x in ('orb_ms','lsd_ms','planes_ms','all_ms')})"
done
