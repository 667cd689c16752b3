"""Phase timeline of every kernel of the fused hot path IN PIPELINE CONTEXT, from in-kernel stamps
(thread 0 of each workgroup records s_memrealtime at phase boundaries; library built with
-DCDN_STAMPS: `make -C codenet_amd/csrc stamps`).  Usage (GPU):
    python tools/with_lib.py codenet_amd/lib/libcodenet_dcn_stamps.so tools/stamps.py [--batch 64]
Prints, per stage and kernel: kernel span (first start .. last end), number of workgroups, mean
per-workgroup phase durations and the distribution of workgroup start times."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from codenet_amd import _native, pipeline

PHASES = {0: ("scale", ["loads+reduce", "finish"], 2),
          1: ("gather", ["staging", "barrier", "gather", "finish"], 4),
          2: ("pointwise", ["prologue", "k-loop", "epilogue", "finish"], 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--fp32", action="store_true")
    a = ap.parse_args()
    lib = _native.lib()
    lib.cdn_debug_read_stamps.argtypes = [ctypes.c_void_p]
    lib.cdn_debug_read_stamps.restype = ctypes.c_int
    dev = torch.device("cuda:0")
    layers = pipeline.build_hot_path(quantized=not a.fp32).to(dev).eval()
    x = pipeline.make_input(a.batch, a.res, device=dev)
    if not a.fp32:
        pipeline.set_running_stat(layers, True)
    fused = pipeline.FusedHotPath(layers.deconv_layers)
    for _ in range(5):
        fused(x)
    torch.cuda.synchronize()
    buf = np.zeros(3 * 2048 * 8 + 64, dtype=np.uint64)
    grids = {}

    def hook(sb):
        torch.cuda.synchronize()
        assert lib.cdn_debug_read_stamps(buf.ctypes.data) == 0
        st = buf[:3 * 2048 * 8].reshape(3, 2048, 8).astype(np.float64) / 100.0      # us
        wv = buf[3 * 2048 * 8:].astype(np.float64) / 100.0
        t_first = None
        print("stage C=%d Co=%d %dx%d" % (sb["C"], sb["Co"], sb["H"], sb["W"]))
        for reg, (name, phases, last) in PHASES.items():
            r = st[reg]
            live = r[:, 0] > 0
            # a region keeps stale entries of an earlier, larger launch: keep the most recent cluster
            if not live.any():
                continue
            t0 = r[live, 0]
            rr = r[live]
            n = rr.shape[0]
            start = rr[:, 0].min()
            if t_first is None:
                t_first = start
            lastcol = 3 if reg == 0 else last
            end = rr[:, lastcol].max()
            idx = [0, 2, 3] if reg == 0 else list(range(last + 1))
            durs = [float((rr[:, idx[i + 1]] - rr[:, idx[i]]).mean()) for i in range(last)]
            s0 = np.sort(rr[:, 0] - start)
            print("  %-9s launch+%6.1f us  span %6.1f us  wgs %4d  per-WG: %s | WG %5.1f us | starts p50 %5.1f p90 %5.1f max %5.1f" % (
                name, start - t_first, end - start, n,
                "  ".join("%s %5.1f" % (p, d) for p, d in zip(phases, durs)),
                float((rr[:, lastcol] - rr[:, 0]).mean()), s0[n // 2], s0[int(n * 0.9)], s0[-1]))
        r1 = st[1][st[1][:, 0] > 0]
        if r1.shape[0] and (r1[:, 5] > 0).all():       # persistent gather (dw0p_kernel): first-item stamps
            print("  dw0p: first DMA landed +%.1f us, first gather done +%.1f, second +%.1f, last item: barrier +%.1f, "
                  "image +%.1f, gathered +%.1f, finished +%.1f (means over %d workgroups, relative to the workgroup's start)"
                  % tuple([float((r1[:, k] - r1[:, 0]).mean()) for k in (5, 6, 7, 1, 2, 3, 4)] + [r1.shape[0]]))
        if (wv > 0).any():
            base = st[1][0, 2]
            print("  gather: per-wave end of workgroup 0 relative to its barrier: " +
                  " ".join("%.1f" % (t - base) for t in wv[wv > 0]))
        assert lib.cdn_debug_clear_stamps() == 0

    assert lib.cdn_debug_clear_stamps() == 0
    fused.stage_hook = hook
    fused(x)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
