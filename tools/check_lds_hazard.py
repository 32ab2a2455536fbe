"""Static check of the hand-issued LDS reads in conv_igemm_h3p: the kernel issues `ds_read_b128` through inline asm (the
compiler's waitcnt pass does not know about them) and orders them with explicit `s_waitcnt lgkmcnt(N)`.  The hardware does
not interlock on outstanding LDS returns, so NO instruction may read (or overwrite) the destination registers of a read
that a preceding wait has not retired.  This script disassembles the kernel (hipcc -S) and replays every basic block of
the persistent kernels: outstanding reads form a FIFO (LDS returns are in order); `lgkmcnt(N)` retires all but the newest
N; any use of a pending destination register is reported.

    python tools/check_lds_hazard.py            # compiles lungmask_amd/csrc/nn_kernels_h3.hip itself
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "lungmask_amd", "csrc", "nn_kernels_h3.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                    "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    asm = open(out).read()


def regs(tok):
    """'v[12:15]' / 'v7' -> set of vgpr indices"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


bad = 0
checked = 0


def successors(blk_lines, nxt):
    """labels this block can continue to (branch targets + fallthrough)."""
    out = []
    falls = True
    for ln in blk_lines:
        op, _, rest = ln.lstrip("@").partition(" ")
        if op.startswith("s_cbranch"):
            out.append(rest.strip())
        elif op == "s_branch":
            out.append(rest.strip())
            falls = False
        elif op in ("s_endpgm", "s_setpc_b64"):
            falls = False
    if falls and nxt is not None:
        out.append(nxt)
    return out


for km in re.finditer(r"^(_ZN2lm14conv_igemm_h3[pq]I[^:\n]*):[^\n]*\n(.*?)^\s*s_endpgm", asm, re.S | re.M):
    name, body = km.group(1), km.group(2)
    order, blocks, cur = ["<entry>"], {"<entry>": []}, "<entry>"
    in_asm = False
    for raw in body.splitlines():
        if "#ASMSTART" in raw:
            in_asm = True
            continue
        if "#ASMEND" in raw:
            in_asm = False
            continue
        ln = raw.split(";")[0].strip()
        if not ln:
            continue
        if in_asm:
            ln = "@" + ln  # hand-issued (inline asm) instruction
        m = re.match(r"^(\.?LBB\d+_\d+):", ln)
        if m:
            cur = m.group(1)
            order.append(cur)
            blocks[cur] = []
            continue
        if ln.endswith(":"):
            continue
        blocks[cur].append(ln)
    succ = {lab: successors(blocks[lab], order[i + 1] if i + 1 < len(order) else None) for i, lab in enumerate(order)}
    has_mfma = {lab: any(l.lstrip("@").startswith("v_mfma") for l in blocks[lab]) for lab in order}

    def run(lab, pending):
        """replay one block from the in-flight FIFO `pending`; returns the FIFO at its end"""
        global bad, checked
        pending = [set(x) for x in pending]
        for ln in blocks[lab]:
            hand = ln.startswith("@")
            ln = ln.lstrip("@")
            op, _, rest = ln.partition(" ")
            toks = [t.strip() for t in rest.split(",")] if rest else []
            if op == "ds_read_b128" and hand:  # the compiler's own LDS reads are covered by its waitcnt pass
                dst = regs(toks[0])
                if any(pset & dst for pset in pending):
                    print(f"{name}/{lab}: ds_read_b128 overwrites a pending destination: {ln}")
                    bad += 1
                pending.append(dst)
                continue
            if op in ("ds_write2_b64", "ds_write_b64", "ds_write_b128") and hand:  # counted in lgkmcnt like the reads; no destination
                pending.append(set())
                continue
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", ln)
                if m:
                    n = int(m.group(1))
                    if n == 0:
                        pending = []
                    elif hand and n < len(pending):  # LDS ops retire in order: a hand-placed counted wait leaves the newest n
                        pending = pending[len(pending) - n:]
                    # (a counted wait of the compiler is computed without our reads: conservatively retires nothing)
                continue
            if has_mfma[lab] and (op.startswith("s_load") or op.startswith("s_buffer_load")):
                print(f"{name}/{lab}: scalar load inside a tap block (shares lgkmcnt with the LDS reads): {ln}")
                bad += 1
            if not pending:
                continue
            used = set()
            for t in toks:
                used |= regs(t.split(" ")[0])
            if any(pset & used for pset in pending):
                print(f"{name}/{lab}: '{ln}' touches a register of an LDS read that is still in flight")
                bad += 1
        return pending

    # forward propagation of the in-flight state along the control-flow graph, seeded at the hand-scheduled blocks
    seen = set()
    work = [(lab, ()) for lab in order if has_mfma[lab]]
    while work:
        lab, pend = work.pop()
        key = (lab, tuple(tuple(sorted(x)) for x in pend))
        if key in seen:
            continue
        seen.add(key)
        out = run(lab, pend)
        if out:
            for nx in succ[lab]:
                if nx in blocks:
                    work.append((nx, tuple(frozenset(x) for x in out)))
    checked += sum(1 for lab in order for l in blocks[lab] if l.startswith("@ds_read_b128"))

# Second invariant: a workgroup barrier of these kernels either publishes LDS-DMA data -- then each wave must have drained its
# own DMAs (s_waitcnt vmcnt(0)) after its last DMA and before the s_barrier (hipcc once dropped that wait on one path; the
# hand-written lm_barrier_dma() carries it in the same asm statement) -- or it is the item-switch barrier that only orders LDS
# accesses (lm_barrier_lds(), tagged LM_BARRIER_LDS_ONLY in the asm text): that one needs lgkmcnt(0) in front.
barriers = lds_only = 0
for km in re.finditer(r"^(_ZN2lm14conv_igemm_h3[pq]I[^:\n]*):[^\n]*\n(.*?)^\s*s_endpgm", asm, re.S | re.M):
    name, raw_lines = km.group(1), km.group(2).splitlines()
    lines = [l.split(";")[0].strip() for l in raw_lines]
    for i, ln in enumerate(lines):
        if ln != "s_barrier":
            continue
        barriers += 1
        tagged = "LM_BARRIER_LDS_ONLY" in raw_lines[i]
        lds_only += tagged
        want = "lgkmcnt(0)" if tagged else "vmcnt(0)"
        ok = False
        for j in range(i - 1, max(i - 40, -1), -1):
            if "global_load_lds" in lines[j] or (lines[j].startswith("buffer_load") and lines[j].endswith("lds")) or lines[j].endswith(":") and not lines[j].startswith("."):
                break
            if lines[j].startswith("s_waitcnt") and want in lines[j]:
                ok = True
                break
        if not ok:
            print(f"{name}: s_barrier without a preceding s_waitcnt {want}")
            bad += 1
print(f"{barriers} barriers checked for the wait in front of them ({lds_only} of them LDS-only)")
print(f"{checked} hand-issued ds_read_b128 checked (in-flight state propagated along the control-flow graph), {bad} hazards")
sys.exit(1 if bad else 0)
