"""Turns gpurun_out/<round>/ (tools/collect_profiles.sh) into the committed profiles/<round>/:
bench JSON lines, the rocprofv3 kernel-stats CSV, a PMC summary and pmc_traffic.json (bytes per step
per kernel family, FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = "gpurun_out/" + R, "profiles/" + R
os.makedirs(dst, exist_ok=True)
for f in glob.glob(src + "/*.json"):
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))
for f in glob.glob(src + "/*kernel_stats.csv") + glob.glob(src + "/backbone_kernel_order.txt"):
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))
st = glob.glob(src + "/stats/*/*kernel_stats.csv")
if st:
    shutil.copy(st[0], dst + "/bench_w4a8_fused_kernel_stats.csv")

FAMILY = {"dw2": "dw", "pwi8": "pointwise", "pw3": "pointwise", "scale_n": "scale", "unpack": "unpack"}


def pmc(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    files = glob.glob(path)
    if not files:
        return {}
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"]
        if any(t in k for t in FAMILY):
            name = (k.split("::")[1] if "::" in k else k).split("(")[0][:44] + " grid=" + r["Grid_Size"]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


fe, wr, sq = pmc(src + "/pmc_fetch/*/*counter_collection.csv"), pmc(src + "/pmc_write/*/*counter_collection.csv"), \
    pmc(src + "/pmc_sq/*/*counter_collection.csv")
per_family = collections.defaultdict(float)
lines = ["rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | SQ_*), bench.py --steps 3 --no-graph, mean per dispatch",
         "FETCH_SIZE/WRITE_SIZE are KB; gfx950 counts 1/2 of wide coalesced reads -> read MB = 2*FETCH_SIZE/1024", ""]
for k in sorted(fe):
    rd, wt = 2 * fe[k].get("FETCH_SIZE", 0) / 1024, wr.get(k, {}).get("WRITE_SIZE", 0) / 1024
    fam = next(v for t, v in FAMILY.items() if t in k)
    if wt > 0.5 or rd > 0.5:          # skip the early-exiting fallback launches
        per_family[fam] += (rd + wt) * 2 ** 20
    line = "%-64s read %7.1f MB  write %7.1f MB" % (k, rd, wt)
    s = sq.get(k)
    if s:
        wc = s["SQ_WAVE_CYCLES"]
        line += "  | wait_any/wave %.2f valu/wave %.2f lds_bank_conflict/lds_active %.2f" % (
            s["SQ_WAIT_ANY"] / wc, s["SQ_ACTIVE_INST_VALU"] / wc,
            s["SQ_LDS_BANK_CONFLICT"] / max(1, s["SQ_LDS_IDX_ACTIVE"]))
    lines.append(line)
open(dst + "/pmc_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
# scale_nhwc runs twice per step with different sizes but one grid -> its mean was counted once
for k in fe:
    if "scale_nhwc" in k:
        per_family["scale"] += (2 * fe[k].get("FETCH_SIZE", 0) + wr.get(k, {}).get("WRITE_SIZE", 0)) * 1024
json.dump({
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), bench.py default workload "
            "(CoDeNet1x 512x512 W4A8 batch 64, fused path, --no-graph --steps 3); FETCH_SIZE doubled per "
            "MI355X_MICROARCH.md; bytes per step = sum over the kernel family's launches in one step. "
            "Source: profiles/%s/pmc_summary.txt" % R,
    "workload": {"res": 512, "batch": 64, "w2": False, "fp32": False, "path": "fused"},
    "bytes_per_step": {k: int(v) for k, v in per_family.items()}}, open(dst + "/pmc_traffic.json", "w"), indent=1)
for f in sorted(glob.glob(dst + "/bench_*.json")) + sorted(glob.glob(dst + "/e2e*.json")) + sorted(glob.glob(dst + "/train*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), {k: d[k] for k in ("value", "ms_per_step", "e2e_fused_img_s", "images_per_s") if k in d})
    except Exception as e:   # noqa: BLE001
        print(os.path.basename(f), "unreadable", e)
