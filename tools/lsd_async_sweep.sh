#!/bin/bash
# sweep of the async line-growing knobs (diagnostic): early drop x W
for early in 0 1; do
  echo "== early=$early"
  HVO_LSD_ASYNC_EARLY=$early timeout -k 10 120 python tools/lsd_async_check.py $2 $1 1 2>&1 | grep -A1 "ctl:" | cut -c1-330
done
