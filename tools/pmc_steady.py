"""HBM bytes per iteration of the STEADY-STATE tail of a profiled program, from two rocprofv3 PMC passes
(--kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of the same command).

    python tools/pmc_steady.py <fetch_dir> <write_dir> <marker> [--per-iter K] [--last N] [--skip s1,s2]

An iteration starts at every K-th dispatch whose kernel name contains <marker> (the first kernel of a step); the last
N complete iterations are averaged.  FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE counts half of the bytes of
wide coalesced reads (MI355X_MICROARCH.md, section HBM): read bytes = 2 * FETCH_SIZE * 1024.  Prints one JSON object:
bytes per iteration in total and per kernel name."""
import argparse
import collections
import csv
import glob
import json
import os


def short(k):
    k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return k.split("(")[0][:72]


def load(d, counter):
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        raise RuntimeError("no counter_collection.csv under %s" % d)
    rows = {}
    for r in csv.DictReader(open(files[-1])):
        if r["Counter_Name"] != counter:
            continue
        i = int(r["Dispatch_Id"])
        rows.setdefault(i, [r["Kernel_Name"], 0.0])
        rows[i][1] += float(r["Counter_Value"])           # (a counter may be reported per XCD / SE: summed)
    return [rows[i] for i in sorted(rows)]


def iterations(rows, marker, per_iter, last, skip):
    rows = [r for r in rows if not any(s in r[0] for s in skip)]
    starts = [i for i, r in enumerate(rows) if marker in r[0]][::per_iter]
    if len(starts) < 3:
        raise RuntimeError("marker %r found %d times" % (marker, len(starts)))
    spans = list(zip(starts[:-1], starts[1:]))[-last:]    # complete iterations only (the tail after the last start is cut)
    return rows, spans


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("marker")
    ap.add_argument("--per-iter", type=int, default=1)
    ap.add_argument("--last", type=int, default=2)
    ap.add_argument("--skip", default="__amd_rocclr,at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor")
    a = ap.parse_args()
    skip = [s for s in a.skip.split(",") if s]
    out = {"read": collections.Counter(), "write": collections.Counter()}
    n_iter, n_disp = 0, 0
    for key, d, counter, mul in (("read", a.fetch_dir, "FETCH_SIZE", 2048.0), ("write", a.write_dir, "WRITE_SIZE", 1024.0)):
        rows, spans = iterations(load(d, counter), a.marker, a.per_iter, a.last, skip)
        n_iter = len(spans)
        n_disp = (spans[-1][1] - spans[0][0]) // max(1, n_iter)
        for lo, hi in spans:
            for name, v in rows[lo:hi]:
                out[key][short(name)] += v * mul / n_iter
    kernels = sorted(set(out["read"]) | set(out["write"]), key=lambda k: -(out["read"][k] + out["write"][k]))
    res = {"iterations_averaged": n_iter, "dispatches_per_iteration": n_disp,
           "read_bytes": int(sum(out["read"].values())), "write_bytes": int(sum(out["write"].values())),
           "bytes_per_iteration": int(sum(out["read"].values()) + sum(out["write"].values())),
           "per_kernel_MB": {k: {"read": round(out["read"][k] / 2 ** 20, 2), "write": round(out["write"][k] / 2 ** 20, 2)}
                             for k in kernels[:24]},
           "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes); read = 2 x FETCH_SIZE "
                   "(gfx950 counts half of wide coalesced reads); steady-state iterations delimited by %r" % a.marker}
    print(json.dumps(res))


if __name__ == "__main__":
    try:
        main()
    except RuntimeError as exc:        # (library functions raise; only the command line turns that into an exit)
        raise SystemExit(str(exc))
