for cfg in "256 3" "512 3" "512 4" "256 4" "512 2"; do set -- $cfg
  echo "== FLOOD_T=$1 HEADS=$2"; HVO_FLOOD_T=$1 HVO_PEAC_HEADS=$2 python tools/latency.py std 640 480 1,32 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print({b:{x:d[b][x] for x in ('lsd_ms','planes_ms','all_ms')} for b in ('B1','B32')}, {k:d['B32']['kernels_ms'][k] for k in ('peac_cluster','peac_refine','lsd_grow')})"
done
