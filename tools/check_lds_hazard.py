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
for km in re.finditer(r"^(_ZN2lm14conv_igemm_h3pI[^:\n]*):[^\n]*\n(.*?)^\s*s_endpgm", asm, re.S | re.M):
    name, body = km.group(1), km.group(2)
    # basic blocks; only those with matrix instructions carry the hand-issued reads (elsewhere every LDS read is the
    # compiler's own and its waitcnt pass -- which also counts scalar loads in lgkmcnt -- takes care of it)
    blocks, cur = [], []
    for ln in body.splitlines():
        ln = ln.split(";")[0].strip()
        if not ln:
            continue
        if re.match(r"^\.?LBB\d+_\d+:", ln) or ln.endswith(":"):
            blocks.append(cur)
            cur = []
            continue
        cur.append(ln)
    blocks.append(cur)
    for blk in blocks:
        if not any(l.startswith("v_mfma") for l in blk):
            continue
        pending = []  # FIFO of destination register sets (LDS returns are in order)
        for ln in blk:
            op, _, rest = ln.partition(" ")
            toks = [t.strip() for t in rest.split(",")] if rest else []
            if op == "ds_read_b128":
                dst = regs(toks[0])
                for pset in pending:
                    if pset & dst:
                        print(f"{name}: ds_read_b128 overwrites a pending destination: {ln}")
                        bad += 1
                pending.append(dst)
                checked += 1
                continue
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", ln)
                if m:
                    n = int(m.group(1))
                    pending = pending[len(pending) - n:] if 0 < n < len(pending) else ([] if n == 0 else pending)
                continue
            if op.startswith("s_load") or op.startswith("s_buffer_load"):
                print(f"{name}: scalar load inside a tap block (shares lgkmcnt with the LDS reads): {ln}")
                bad += 1
            if not pending:
                continue
            used = set()
            for t in toks:
                used |= regs(t.split(" ")[0])
            for pset in pending:
                if pset & used:
                    print(f"{name}: '{ln}' touches a register of an LDS read that is still in flight")
                    bad += 1
                    break
        if pending:
            print(f"{name}: {len(pending)} reads still in flight at the end of a tap block")
            bad += 1
# Second invariant: every workgroup barrier of these kernels publishes LDS-DMA data, so each wave must have drained its own
# DMAs (s_waitcnt vmcnt(0)) after its last global_load_lds and before the s_barrier (hipcc once dropped that wait on one path).
barriers = 0
for km in re.finditer(r"^(_ZN2lm14conv_igemm_h3pI[^:\n]*):[^\n]*\n(.*?)^\s*s_endpgm", asm, re.S | re.M):
    name, lines = km.group(1), [l.split(";")[0].strip() for l in km.group(2).splitlines()]
    for i, ln in enumerate(lines):
        if ln != "s_barrier":
            continue
        barriers += 1
        ok = False
        for j in range(i - 1, max(i - 40, -1), -1):
            if "global_load_lds" in lines[j] or lines[j].endswith(":") and not lines[j].startswith("."):
                break
            if lines[j].startswith("s_waitcnt") and "vmcnt(0)" in lines[j]:
                ok = True
                break
        if not ok:
            print(f"{name}: s_barrier without a preceding s_waitcnt vmcnt(0)")
            bad += 1
print(f"{barriers} barriers checked for the vmcnt(0) in front of them")
print(f"{checked} hand-issued ds_read_b128 checked in the matrix blocks of the conv_igemm_h3p instantiations, {bad} hazards")
sys.exit(1 if bad else 0)
