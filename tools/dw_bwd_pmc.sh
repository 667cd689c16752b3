#!/bin/bash
# On the GPU box: SQ counters of the gather-backward kernels (dw_bwd2_kernel / dw_bwd2u_kernel at the QAT step's three
# stage shapes, tools/dw_bwd_bench.py) -- separate --pmc passes, kernel trace only.  Output: gpurun_out/dwbwd_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/dwbwd_pmc
rm -rf $OUT && mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN" "SQ_WAVES SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 tools/dw_bwd_bench.py > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/dwbwd_pmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "dw_bwd2" not in n:
            continue
        key = re.sub(r"\(.*", "", n.replace("void (anonymous namespace)::", "")) + " grid=%s wg=%s lds=%s" % (r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("LDS_Block_Size"))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/dwbwd_pmc.txt", "w") as out:
    for k in sorted(agg):
        out.write(k + "\n")
        for c in sorted(agg[k]):
            v = agg[k][c]
            out.write("   %-26s mean %.4g  (n=%d)\n" % (c, sum(v) / len(v), len(v)))
print(open("gpurun_out/dwbwd_pmc.txt").read())
PY
rm -rf $OUT/p*/
