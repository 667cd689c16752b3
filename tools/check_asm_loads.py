"""ISA check for kernels whose global loads are inline asm behind explicit s_waitcnt (pwi8s_kernel, codenet_fused.hip).

The compiler does not know that the destination of an `asm volatile("global_load_dwordx4 %0, ...")` is written
asynchronously: any copy, spill or use it places between the load and the s_waitcnt vmcnt(N) that covers it reads the
register before the data has landed (the second version of pwi8s_kernel did exactly that: a phi copy in front of the
wait, sporadically wrong sums).  This walks the control-flow graph of the kernel's ISA with the in-order vmcnt model
(loads, LDS-DMAs and stores all count, completing in issue order on gfx9) and reports every instruction that names a
register of an asm load still in flight on SOME path.

    python tools/check_asm_loads.py <device .s file> [kernel-name substring, default pwi8s_kernel]
    (the .s: hipcc -S --cuda-device-only ..., or -save-temps)
Exit status 1 when a hazard is found; tests/test_build.py runs it on the shipped source."""
import re
import sys

VM_OPS = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load",
          "flat_store", "scratch_load", "scratch_store")


def regs_of(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return frozenset(out)


def parse(body):
    """-> list of blocks: dict(label, insts=[(text, is_asm)], succ=[labels / indices])"""
    blocks, cur, in_asm = [], {"labels": [], "insts": []}, False
    for line in body.splitlines():
        raw = line.strip()
        if raw.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if raw.startswith(";;#ASMEND"):
            in_asm = False
            continue
        code = line.split(';')[0].strip()
        if not code:
            continue
        if code.endswith(':'):
            if cur["insts"]:
                blocks.append(cur)
                cur = {"labels": [], "insts": []}
            cur["labels"].append(code[:-1])
            continue
        cur["insts"].append((code, in_asm))
        op = code.split()[0]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm"):
            blocks.append(cur)
            cur = {"labels": [], "insts": []}
    if cur["insts"]:
        blocks.append(cur)
    index = {}
    for k, b in enumerate(blocks):
        for lab in b["labels"]:
            index[lab] = k
    for k, b in enumerate(blocks):
        code = b["insts"][-1][0]
        op = code.split()[0]
        if op == "s_endpgm":
            b["succ"] = []
        elif op == "s_branch":
            b["succ"] = [index[code.split()[1]]]
        elif op.startswith("s_cbranch"):
            b["succ"] = [index[code.split()[1]]] + ([k + 1] if k + 1 < len(blocks) else [])
        else:
            b["succ"] = [k + 1] if k + 1 < len(blocks) else []
    return blocks


def check_kernel(name, body, verbose=True):
    blocks = parse(body)
    seen = set()
    work = [(0, ())]                 # (block, in-flight vm ops oldest first: frozenset of asm-loaded regs, or empty)
    hazards = {}
    nloads = set()
    while work:
        k, st = work.pop()
        if (k, st) in seen:
            continue
        seen.add((k, st))
        st = list(st)
        for code, is_asm in blocks[k]["insts"]:
            op = code.split()[0]
            if op == "s_waitcnt":
                m = re.search(r'vmcnt\((\d+)\)', code)
                if m:
                    n = int(m.group(1))
                    st = st[len(st) - n:] if n < len(st) else st
                    if n == 0:
                        st = []
                continue
            flying = frozenset().union(*st) if st else frozenset()
            if op.startswith(VM_OPS):
                if is_asm and op == "global_load_dwordx4":
                    dst = regs_of(code.split()[1])
                    addr = regs_of(code[code.index(',') + 1:])
                    if addr & flying:
                        hazards.setdefault(code, sorted(addr & flying))
                    if dst & flying:
                        hazards.setdefault(code, sorted(dst & flying))
                    nloads.add(code)
                    st.append(dst)
                else:
                    if regs_of(code) & flying:
                        hazards.setdefault(code, sorted(regs_of(code) & flying))
                    st.append(frozenset())
                if len(st) > 63:
                    st = st[-63:]
                continue
            hit = regs_of(code) & flying
            if hit:
                hazards.setdefault(code, sorted(hit))
        for nx in blocks[k]["succ"]:
            work.append((nx, tuple(st)))
    if verbose:
        print("%s: %d blocks, %d (block, state) pairs, %d distinct asm loads, %d hazards" % (
            name[:64], len(blocks), len(seen), len(nloads), len(hazards)))
        for code, regs in hazards.items():
            print("   in-flight v%s named by: %s" % (regs, code))
    return len(hazards), len(nloads)


def check(path, needle="pwi8s_kernel", verbose=True):
    text = open(path).read()
    bad = total = 0
    for m in re.finditer(r'^(_Z\S*%s\S*):[^\n]*\n(.*?)\n\.Lfunc_end' % re.escape(needle), text, re.S | re.M):
        h, n = check_kernel(m.group(1), m.group(2), verbose)
        bad += h
        total += n
    return bad, total


if __name__ == "__main__":
    bad, total = check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "pwi8s_kernel")
    if total == 0:
        print("no asm loads found")
    sys.exit(1 if bad or total == 0 else 0)
