#!/bin/bash
# instruction counts of k_orb_level per phase: SQ counters with HVO_LT_SKIP masks (timing experiment knob of orb_level.hip) -- needs a -DHVO_TIMING_KNOBS build of libhvo.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=${1:-1024}
O=$R/gpurun_out/prof/orb_pmc_ph
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for sk in 0 1 2 3 7; do
  rm -rf $O/s$sk
  HVO_LT_SKIP=$sk timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR -d $O/s$sk -o pmc --output-format csv -- python3 $R/tools/stage_batch_sweep.py orb $B > $O/s$sk.log 2>&1 || echo "skip $sk failed: $(tail -3 $O/s$sk.log)"
done
python3 - <<PY
import csv, glob, collections
for sk in (0, 1, 2, 3, 7):
    for f in glob.glob("$O/s%d/**/*counter_collection.csv" % sk, recursive=True):
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("k_orb_level"): acc[r["Counter_Name"]] += float(r["Counter_Value"])
        print("skip=%d" % sk, " ".join("%s=%.4g" % (c, x / 3 / $B) for c, x in sorted(acc.items())), "(per frame)")
PY
