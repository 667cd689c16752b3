"""Register / scratch budget of the hot kernels, from the compiler's own metadata (hipcc -S, no GPU needed).
The stage-0 gather runs at 127 of the 128 VGPRs its 16-waves-per-CU launch allows: one more live value and
the compiler spills, which costs 20-150 us per launch (seen with the stamp instrumentation, 56 -> 78 us, and
with an extra kernel argument, 56 -> 202 us).  tests/test_host_api.py runs this as a regression guard."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "codenet_amd", "csrc")
# kernel name fragment -> (max VGPRs incl. AGPRs, or None) ; spills / scratch must be zero for all of them
BUDGET = {
    "dw2_kernelILi64ELb0ELb0ELb1ELi1024": 128,    # stage 0, W4A8 (NCHW input, s quantised)
    "dw2_kernelILi64ELb0ELb0ELb0ELi1024": 128,    # stage 0, fp32
    "dw2u_kernelILi64ELb1ELb1": 256,              # stage 1
    "dw2u_kernelILi32ELb1ELb1": 256,              # stage 2
    "pwi8_kernelILi64ELi128ELi2ELb1": 128,        # stages 0-1: four workgroups per CU
    "pwi8_kernelILi128ELi64ELi4ELb1": 128,        # stage 2 (small M)
    "pwi8s_kernelILi4ELi3": 168,                  # stage 0 (round 5): three workgroups per CU (512 / 3 registers)
    "pwi8_kernelILi64ELi64ELi2ELb1": 128,         # Co <= 64 at large M: stage 2, layer 1, heads
    "scale_nchw_kernel": 128, "scale_nhwc_kernelILb1": 128, "unpack_kernelILb1": 128,
    "pwd3_kernelILi2": 256, "pwd3_kernelILi4": 256,   # streaming pointwise: two waves per SIMD
}
# SGPR spills go to VGPR lanes (v_writelane / v_readlane, no memory): tolerated up to this many where listed
SGPR_SPILLS_OK = {"pwd3_kernelILi4": 6}     # round 4: the window cursors of the unit-entry conv (228 VGPRs); round 5: + the NaN flag's mask
# codenet_layers.hip: no spills; the row-streaming kernels must leave two workgroups per CU
BUDGET_LAYERS = {
    "dwx_kernelILb1ELi1ELi2": 256, "dwx_kernelILb1ELi2ELi2": 256,
    "dws_kernelILb1ELi1ELi32ELi1": 256, "dws_kernelILb1ELi1ELi64ELi1": 256, "dws_kernelILb1ELi2ELi32ELi1": 256,
    "head_small_kernelILi0": 256, "head_small_kernelILi1ELi2": 256, "head_small_kernelILi2": 256,
}


# codenet_stage.hip is built WITHOUT the SLP vectoriser (csrc/Makefile): with it the QAT forward
# gather needs 152 VGPRs (three waves per SIMD) instead of 99-106; dcn_generic.hip: the seam's parameter-gradient kernel
# holds a 45-value record per lane under the 128-VGPR cap of its 1024-thread workgroups
BUDGET_STAGE = {"dw4_kernelILb1E": 128, "dw4_kernelILb0E": 128}
BUDGET_GENERIC = {"dwo_wgrad_kernelILi16E": 128, "dwo_wgrad_kernelILi8E": 128,
                  # round 6: the structured forward (four waves per SIMD: 170 VGPRs with a run-time test of the plane
                  # pointer instead of the template parameter) and the structured backward_input kernels (1024-thread
                  # workgroups: 128)
                  "dwo4_kernelILb1E": 128, "dwo4_kernelILb0E": 128,
                  # (the instantiations the CoDeNet stage shapes use: one pass with chunks of 16 / 8, the split form's
                  # grad_input pass with 4 and grad_offset pass with 8; the thin-chunk one-pass forms spill a few VGPRs
                  # and are never launched -- dwos_bwd_split)
                  "dwos_bwd_kernelILi16ELi0E": 128, "dwos_bwd_kernelILi8ELi0E": 128, "dwos_bwd_kernelILi4ELi1E": 128,
                  "dwos_bwd_kernelILi8ELi2E": 128}
SGPR_SPILLS_OK_X = 32      # (SGPR spills go to VGPR lanes: tolerated in the kernels of the two tables below)
# codenet_frozen.hip (round 6): the occupancy the serving network's small launches were measured with -- the 64 x 64
# byte-code pointwise at eight waves per SIMD, the fused depthwise + pointwise at two (256 columns) / three
BUDGET_FROZEN = {"pwq8_kernelILi64ELi64E": 64, "dwpwq8_kernelILi1ELi4ELi2ELi128E": 168, "dwpwq8_kernelILi1ELi4ELi1ELi64E": 168,
                 "dwpwq8_kernelILi2ELi2ELi1ELi64E": 168, "dwpwq8_kernelILi2ELi2ELi2ELi128E": 168,
                 "dwpwq8_kernelILi1ELi4ELi2ELi256ELi32E": 256, "dwpwq8_kernelILi2ELi2ELi2ELi256ELi32E": 256}


def kernel_resources(src="codenet_fused.hip", extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950",
                        "-munsafe-fp-atomics", *extra, "-S", "--cuda-device-only", "-o", out, src],
                       cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    res = {}
    for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size: \d+", txt, re.S):
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))   # noqa: E731
        res[name] = dict(vgpr=g("vgpr_count"), spill=g("vgpr_spill_count"), sgpr_spill=g("sgpr_spill_count"),
                         scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"))
    return res


def check():
    res = kernel_resources()
    problems = []
    res_l = kernel_resources("codenet_layers.hip")
    for frag, cap in BUDGET_LAYERS.items():
        hits = [(n, r) for n, r in res_l.items() if frag in n]
        if not hits:
            problems.append("kernel %s not found" % frag)
        for n, r in hits:
            # (a few bytes of private segment for a local array are tolerated here: no VGPR spills)
            if r["spill"] or r["sgpr_spill"]:
                problems.append("%s spills: %s" % (n, r))
            if r["vgpr"] > cap:
                problems.append("%s uses %d VGPRs (budget %d)" % (n, r["vgpr"], cap))
    res.update(res_l)
    for src, extra, budget in (("codenet_stage.hip", ("-fno-slp-vectorize",), BUDGET_STAGE), ("dcn_generic.hip", (), BUDGET_GENERIC),
                               ("codenet_frozen.hip", (), BUDGET_FROZEN)):
        res_x = kernel_resources(src, extra)
        for frag, cap in budget.items():
            hits = [(n, r) for n, r in res_x.items() if frag in n]
            if not hits:
                problems.append("kernel %s not found" % frag)
            sg_ok = 0 if budget is BUDGET_STAGE else SGPR_SPILLS_OK_X
            for n, r in hits:
                if r["spill"] or r["scratch"] or r["sgpr_spill"] > sg_ok:
                    problems.append("%s spills: %s" % (n, r))
                if r["vgpr"] > cap:
                    problems.append("%s uses %d VGPRs (budget %d)" % (n, r["vgpr"], cap))
        res.update(res_x)
    for frag, cap in BUDGET.items():
        hits = [(n, r) for n, r in res.items() if frag in n]
        if not hits:
            problems.append("kernel %s not found" % frag)
        for n, r in hits:
            if r["spill"] or r["scratch"] or r["sgpr_spill"] > SGPR_SPILLS_OK.get(frag, 0):
                problems.append("%s spills: %s" % (n, r))
            if cap is not None and r["vgpr"] > cap:
                problems.append("%s uses %d VGPRs (budget %d)" % (n, r["vgpr"], cap))
    return res, problems


if __name__ == "__main__":
    res, problems = check()
    for frag in list(BUDGET) + list(BUDGET_STAGE) + list(BUDGET_GENERIC) + list(BUDGET_FROZEN):
        for n, r in res.items():
            if frag in n:
                print("%-70s %s" % (n[18:88], r))
    if problems:
        print("\n".join(problems))
        sys.exit(1)
    print("ok")
