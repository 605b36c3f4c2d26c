#!/bin/bash
# PMC passes over the attention launches of the 25-image pass (one counter group per pass; kernel-trace only, as gpurun requires)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-attn_pmc}; mkdir -p $OUT
python scripts/attn_pmc.py 10 > $OUT/timing.log 2>&1; cat $OUT/timing.log | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o a -- python $ROOT/scripts/attn_pmc.py 2 > $OUT/p$i.log 2>&1
  echo "group $i: $grp -> $(ls $OUT/p$i 2>/dev/null | head -3 | tr '\n' ' ')"
done
cd $ROOT
python - <<P
import csv, glob, collections, os
out = "$OUT"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "attn_fwd" not in k: continue
        key = k.split("(")[0][-28:] + " grid=" + r.get("Grid_Size", "?")
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]): print(f"   {c:36s} {acc[k][c] / n[k][c]:16.0f}  (n={n[k][c]})")
P
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
