#!/bin/bash
# usage: tools/pmc.sh <tag> <kernel-substring> -- <python args...>; runs the PMC passes (counters only,
# separate runs as the MI355X guide prescribes) and prints per-dispatch averages for the kernel.
tag=$1; kern=$2; shift 3
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line --output-format csv -d $root/gpurun_out/pmc_${tag}/p$i -- "$@" > $root/gpurun_out/pmc_${tag}_p$i.log 2>&1
done <<'PASSES'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_CVT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM
FETCH_SIZE TCC_HIT_sum
WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
PASSES
cd $root
python3 - "$root/gpurun_out/pmc_${tag}" "$kern" <<'PY'
import csv, glob, sys, collections
root, kern = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:28s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")
PY
