#!/bin/bash
# Regenerates the raw material of profiles/rNN_* on the GPU box (run through gpurun from the repo root):
#   bash tools/refresh_profiles.sh            -> gpurun_out/prof/...
# then, back in the container:  python tools/collect_profiles.py gpurun_out/prof r01
# Counter passes are separate runs (--pmc is never combined with the trace domains).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
echo "[1/7] default bench"; timeout -k 10 400 python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "[2/7] kernel trace"; timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
echo "[3/7] pmc sq"; timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $O/pmc_sq -o pmc --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_sq.log 2>&1
echo "[4/7] pmc fetch"; timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
echo "[5/7] pmc write"; timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_write.log 2>&1
cd $R
echo "[6/7] pcie"; timeout -k 10 300 python3 tools/pcie_rate.py --batch 1024 > $O/pcie_note.txt 2>&1
echo "[7/7] footprint"; timeout -k 10 300 python3 tools/mem_per_frame.py > $O/hbm_footprint.txt 2>&1
# keep only the small csv files (the per-dispatch kernel trace of 7 steps is a few hundred KB)
find $O -name "*.csv" -size +8M -delete
echo done
