#!/bin/bash
# Counter passes over one stage alone (tools/stage_batch_sweep.py STAGE 8192): memory-pipeline view of its kernels.
#   bash tools/stage_pmc.sh [planes|lsd|orb]         (through gpurun, from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
ST=${1:-planes}
O=$R/gpurun_out/prof/${ST}_pmc
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rm -rf $O/p$i
  timeout -k 10 200 rocprofv3 --pmc $set -d $O/p$i -o pmc --output-format csv -- python3 $R/tools/stage_batch_sweep.py $ST 8192 > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:28]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            if k.startswith("__amd"): continue
            print(k, " ".join("%s=%.4g" % (c, x / max(1, n[(k, c)]) / 8192) for c, x in v.items()), "(per frame)")
PY
