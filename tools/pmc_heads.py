"""Per-kernel SQ counter ratios of the head kernels from a rocprofv3 --pmc run of tools/heads_bench.py (argv[1]: the
output directory)."""
import csv, glob, sys, collections
d = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:50]
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(rows.items()):
    if not any(t in k for t in ("head", "pwi8")):
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    busy = m.get("SQ_BUSY_CYCLES", 0); wc = m.get("SQ_WAVE_CYCLES", 0)
    print("%-40s n=%3d  valu/wave %.3f  lds/wave %.3f  | raw busy %.3g wave %.3g valu %.3g |  valu_active/busy %.2f  lds_active/busy %.2f  wait_any/wave %.2f  wait_inst/wave %.2f  lds_conflict/lds %.2f" % (
        k, len(next(iter(cs.values()))), m.get("SQ_ACTIVE_INST_VALU", 0) / max(wc, 1), m.get("SQ_ACTIVE_INST_LDS", 0) / max(wc, 1),
        busy, wc, m.get("SQ_ACTIVE_INST_VALU", 0), m.get("SQ_ACTIVE_INST_VALU", 0) / max(busy, 1), m.get("SQ_ACTIVE_INST_LDS", 0) / max(busy, 1),
        m.get("SQ_WAIT_ANY", 0) / max(wc, 1), m.get("SQ_WAIT_INST_ANY", 0) / max(wc, 1),
        m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
