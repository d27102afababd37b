#!/bin/bash
# SQ counter passes over the ORB stage alone (3 runs of B frames): per-frame instruction counts and wave cycles of its kernels,
# all dispatches of a kernel summed (the fused pass launches k_orb_level once per level).
#   bash tools/orb_pmc.sh [B]     (through gpurun, from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=${1:-2048}
O=$R/gpurun_out/prof/orb_pmc
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  rm -rf $O/p$i
  timeout -k 10 200 rocprofv3 --pmc $set -d $O/p$i -o pmc --output-format csv -- python3 $R/tools/stage_batch_sweep.py orb $B > $O/p$i.log 2>&1 || echo "pass $i failed: $(tail -3 $O/p$i.log)"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:24]][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if k.startswith("__amd"): continue
            print("%-24s" % k, " ".join("%s=%.5g" % (c, x / 3 / $B) for c, x in sorted(v.items())), "(per frame)")
PY
