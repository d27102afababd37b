#!/bin/bash
# Regenerates the raw material of profiles/rNN_* on the GPU box (run through gpurun from the repo root, in two calls):
#   bash tools/refresh_profiles.sh a      -> gpurun_out/prof/... (bench lines, kernel trace, counter passes)
#   bash tools/refresh_profiles.sh b      -> gpurun_out/prof/... (stream / big1280 / batch256 lines, side measurements)
# then, back in the container:  python tools/collect_profiles.py gpurun_out/prof r03
# Counter passes are separate runs (--pmc is never combined with the trace domains); the program after `--` is python3 itself.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
mkdir -p $O
Q="--no-cpu-baseline --no-extras"
if [ "$1" = "a" ]; then
cd /tmp; export TMPDIR=/tmp
echo "[a1] default bench"; timeout -k 10 500 python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "[a2] kernel trace"; rm -rf $O/kt; timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 $Q > $O/bench_under_rocprof.json 2> $O/kt.err
echo "[a3] pmc sq"; rm -rf $O/pmc_sq; timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $O/pmc_sq -o pmc --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $Q > $O/pmc_sq.log 2>&1
echo "[a4] pmc fetch"; rm -rf $O/pmc_fetch; timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $Q > $O/pmc_fetch.log 2>&1
echo "[a5] pmc write"; rm -rf $O/pmc_write; timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $Q > $O/pmc_write.log 2>&1
find $O -name "*.csv" -size +8M -delete
else
cd $R
echo "[b1] stream bench"; timeout -k 10 400 python3 bench.py --mode stream > $O/bench_stream.json 2> $O/bench_stream.err
echo "[b2] big1280"; timeout -k 10 400 python3 bench.py --config big1280 --steps 5 > $O/bench_big1280.json 2> $O/bench_big1280.err
echo "[b3] batch256"; timeout -k 10 400 python3 bench.py --config batch256 --steps 20 > $O/bench_batch256.json 2> $O/bench_batch256.err
echo "[b4] lowtex / std latency"; timeout -k 10 300 python3 tools/latency.py std > $O/latency_std.json 2>/dev/null; timeout -k 10 300 python3 tools/latency.py lowtex > $O/latency_lowtex.json 2>/dev/null
echo "[b5] stream scaling"; timeout -k 10 300 python3 tools/stream_scaling.py 150 > $O/stream_scaling_q4.json 2>/dev/null; GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python3 tools/stream_scaling.py 150 > $O/stream_scaling_q16.json 2>/dev/null
echo "[b6] pcie"; (for a in "1024 1" "1024 3" "2048 3" "3072 3"; do timeout -k 10 200 python3 tools/pcie_overlap.py $a 4 2>/dev/null | tail -1; done; LOCKS=0 timeout -k 10 200 python3 tools/pcie_overlap.py 2048 3 4 2>/dev/null | tail -1) > $O/pcie_note.txt
echo "[b7] footprint"; timeout -k 10 300 python3 tools/mem_per_frame.py > $O/hbm_footprint.txt 2>&1
echo "[b8] matching"; (timeout -k 10 100 python3 tools/match_rate.py 2000 2000 2>/dev/null | tail -1; timeout -k 10 100 python3 tools/match_rate.py 1000 1000 2>/dev/null | tail -1) > $O/match_rate.txt
echo "[b9] ahc timing"; timeout -k 10 300 python3 tools/peac_timing.py --batch 8192 > $O/peac_timing_batch.txt 2>&1
echo "[b10] heads kernel phases, one frame's kernels, 1280x960 latency"; timeout -k 10 300 python3 tools/peac_heads_timing.py > $O/peac_heads_timing.txt 2>&1
timeout -k 10 300 python3 tools/latency.py std 1280 960 1,32 > $O/latency_1280.json 2>/dev/null
timeout -k 10 300 python3 tools/lsd_stats.py > $O/lsd_stats.txt 2>&1; timeout -k 10 300 python3 tools/lsd_stats.py 1280 960 >> $O/lsd_stats.txt 2>&1
(cd /tmp; export TMPDIR=/tmp; rm -rf $O/kt1; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt1 -o one --output-format csv -- python3 $R/tools/one_frame.py std 7 > /dev/null 2>&1; rm -f $O/kt1/*kernel_trace.csv)
fi
echo done
