"""Turns gpurun_out/<round>/ (tools/collect_profiles.sh) into the committed profiles/<round>/:
bench JSON lines, the rocprofv3 kernel-stats CSVs, a PMC summary per schedule (running = reference-faithful,
frozen = byte codes in HBM) with HBM bytes (FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM), SQ wait /
VALU / LDS shares and the matrix-core utilisation, and pmc_traffic*.json (bytes per step per kernel family)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r06"
src, dst = "gpurun_out/" + R, "profiles/" + R
os.makedirs(dst, exist_ok=True)
for f in (glob.glob(src + "/*.json") + glob.glob(src + "/*kernel_stats.csv") + glob.glob(src + "/backbone_kernel_order.txt")
          + glob.glob(src + "/e2e_frozen_*_steady.txt")):
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))
for a, b in (("stats_kernel_stats.csv", "bench_w4a8_fused_kernel_stats.csv"),
             ("stats_frozen_kernel_stats.csv", "bench_w4a8_frozen_kernel_stats.csv"),
             ("e2e_stats_kernel_stats.csv", "e2e_native_kernel_stats.csv"),
             ("train_stats_kernel_stats.csv", "train_step_kernel_stats.csv")):
    if os.path.exists(os.path.join(dst, a)):
        os.replace(os.path.join(dst, a), os.path.join(dst, b))

FAMILY = {"dw2": "dw", "dw0p": "dw", "pwi8": "pointwise", "pw3": "pointwise", "pws_": "pointwise", "pwq8": "pointwise", "scale_n": "scale",
          "unpack": "unpack", "expand8": "unpack", "frozen_params": "other"}
HOT = tuple(FAMILY)


def short(k):
    return (k.split("::")[1] if "::" in k else k).split("(")[0][:64]


def is_frozen_variant(k):
    """kernels of the byte-code schedule: pwq8 / expand8 / frozen_params, the gather and scale instantiations with
    byte input and / or output (their last template flags), plus the stage-0 scale kernel both schedules share"""
    name = k.split(" grid=")[0]
    if any(t in name for t in ("pwq8", "expand8", "frozen_params", "scale_nchw")):
        return True
    if name.startswith("dw2_kernel") or name.startswith("dw0p_kernel"):
        return name.rstrip(">").endswith("true")                  # <..., X8, OUT8> / <SQ, OUT8>: OUT8
    if name.startswith("dw2u_kernel"):
        return name.rstrip(">").endswith("true")
    if name.startswith("scale_nhwc"):
        return name.rstrip(">").endswith("true, true")            # <XQ, X8>
    return False


def rows_of(path, keep=HOT):
    files = sorted(glob.glob(path), key=os.path.getmtime)      # a re-collected round leaves several: newest
    if not files:
        return
    for r in csv.DictReader(open(files[-1])):
        if keep is None or any(t in r["Kernel_Name"] for t in keep):
            yield short(r["Kernel_Name"]) + " grid=" + r["Grid_Size"], r


def pmc(path, keep=HOT):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, r in rows_of(path, keep) or ():
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def mfma_line(m):
    """matrix-core utilisation of a dispatch: busy cycles of the MFMA pipes / (CU-busy cycles x 4 SIMDs)."""
    busy, cu = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), m.get("SQ_BUSY_CU_CYCLES", 0.0)
    ops = {k[len("SQ_INSTS_VALU_MFMA_MOPS_"):]: v for k, v in m.items()
           if k.startswith("SQ_INSTS_VALU_MFMA_MOPS_") and v}
    util = busy / (4.0 * cu) if cu else 0.0
    return "mfma_busy/(4*cu_busy) %.3f  mops %s" % (util, {k: int(v) for k, v in ops.items()} or "-")


for mode in ("running", "frozen"):
    fe, wr, sq, mf = (pmc(src + "/pmc_%s_%s/*/*counter_collection.csv" % (n, mode))
                      for n in ("fetch", "write", "sq", "mfma"))
    if not fe:
        continue
    lines = ["schedule: %s  (bench.py %s--steps 3 --no-graph; rocprofv3 --pmc in separate passes: FETCH_SIZE | "
             "WRITE_SIZE | SQ_* | MFMA; mean per dispatch)" % (mode, "--frozen " if mode == "frozen" else ""),
             "FETCH_SIZE/WRITE_SIZE are KB; gfx950 counts 1/2 of wide coalesced reads -> read MB = 2*FETCH_SIZE/1024", ""]
    # the frozen run also executes the running schedule while it warms the ranges up: keep the schedule's own kernels
    member = (lambda k: is_frozen_variant(k)) if mode == "frozen" else \
        (lambda k: not is_frozen_variant(k) or "scale_nchw" in k)
    fe = {k: v for k, v in fe.items() if member(k)}
    for k in sorted(fe):
        rd, wt = 2 * fe[k].get("FETCH_SIZE", 0) / 1024, wr.get(k, {}).get("WRITE_SIZE", 0) / 1024
        line = "%-80s read %7.1f MB  write %7.1f MB" % (k, rd, wt)
        s = sq.get(k)
        if s and s.get("SQ_WAVE_CYCLES"):
            wc = s["SQ_WAVE_CYCLES"]
            line += "  | wait_any/wave %.2f valu/wave %.2f lds_conflict/lds_active %.2f" % (
                s["SQ_WAIT_ANY"] / wc, s["SQ_ACTIVE_INST_VALU"] / wc,
                s["SQ_LDS_BANK_CONFLICT"] / max(1, s["SQ_LDS_IDX_ACTIVE"]))
        if k in mf:
            line += "  | " + mfma_line(mf[k])
        lines.append(line)
    # bytes per step: every (kernel, grid) entry of the schedule is launched once per step (three stages with
    # different grids; expand8 / unpack belong to the optional hand-over to an fp32 consumer)
    per_family = collections.defaultdict(float)
    for k in fe:
        fam = next(v for t, v in FAMILY.items() if t in k)
        per_family[fam] += (2 * fe[k].get("FETCH_SIZE", 0) + wr.get(k, {}).get("WRITE_SIZE", 0)) * 1024
    tot = sum(v for f_, v in per_family.items() if f_ != "unpack")
    lines += ["", "HBM bytes per step by kernel family: " +
              ", ".join("%s %.1f MB" % (f_, v / 2 ** 20) for f_, v in sorted(per_family.items())),
              "step total without unpack: %.1f MB" % (tot / 2 ** 20)]
    name = "pmc_summary.txt" if mode == "running" else "pmc_summary_frozen.txt"
    open(os.path.join(dst, name), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    json.dump({
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), bench.py default workload "
                "(CoDeNet1x 512x512 W4A8 batch 64, fused path, %s schedule, --no-graph --steps 3); FETCH_SIZE "
                "doubled per MI355X_MICROARCH.md; bytes per step = sum over the kernel family's launches in one "
                "step. Source: profiles/%s/%s" % (mode, R, name),
        "workload": {"res": 512, "batch": 64, "w2": False, "fp32": False, "path": "fused", "frozen": mode == "frozen"},
        "bytes_per_step": {k: int(v) for k, v in per_family.items()}},
        open(os.path.join(dst, "pmc_traffic.json" if mode == "running" else "pmc_traffic_frozen.json"), "w"), indent=1)

# whole network: matrix-core utilisation per kernel (pwi8 / pw3 / pwd3 / head_small / stem ...)
mf = pmc(src + "/pmc_mfma_e2e/*/*counter_collection.csv", keep=None)
if mf:
    lines = ["whole network + decode (tools/e2e_native_bench.py --steps 3), rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES "
             "SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_*: mean per dispatch, kernels that issue MFMAs", ""]
    for k in sorted(mf):
        if mf[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
            lines.append("%-72s %s" % (k, mfma_line(mf[k])))
    open(os.path.join(dst, "pmc_mfma_e2e.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
for f in sorted(glob.glob(dst + "/bench_*.json")) + sorted(glob.glob(dst + "/e2e*.json")) + \
        sorted(glob.glob(dst + "/train*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), {k: d[k] for k in ("value", "ms_per_step", "images_per_s") if k in d},
              {k: d[k].get("ms_per_step", d[k].get("ms_per_batch")) for k in ("frozen_int8", "e2e")
               if isinstance(d.get(k), dict)})
    except Exception as e:   # noqa: BLE001
        print(os.path.basename(f), "unreadable", e)
