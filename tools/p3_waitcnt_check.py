"""Mechanical check of the COUNTED WAITS the persistent 3x3 kernels' producers publish LDS buffers behind:

    python tools/p3_waitcnt_check.py [lssvc_amd/csrc/conv3_f16x3p.o ...]

The patch-ring schedule (round 5) and the register-prefetch schedule (round 6) signal fill(k+1) after `s_waitcnt vmcnt(N)`: the weight
DMA of that fill (global_load_lds_dwordx4) must have landed, the N patch loads requested after it stay in flight. That is correct only
while the compiler emits AT LEAST N vector-memory instructions between the last weight DMA and the wait -- fewer (merged or dropped
loads after a toolchain change) and the slot is published with the DMA possibly still in flight; the bit-identity tests would catch
that only by chance. This tool disassembles the device code of the given objects (llvm-objdump) and, for every kernel instantiation,
walks the instruction stream: after a weight DMA it counts the vector-memory instructions up to the first s_waitcnt with a vmcnt field
and requires  vmcnt <= that count  (a full drain, vmcnt(0), always passes). Pre-split-input instantiations (their patches are DMAs
themselves, counted differently) are listed but not judged. The split-roles schedule (round 6, PF = 3) keeps the DMA in a wave of its
own (drained with vmcnt(0)); its patch waves' counted waits -- like every hand-written counted wait -- carry an expcnt(6) mark and are checked by marked_waits below. Exit status 1 on a violation. tests/test_host_logic.py runs it on the
in-tree objects."""
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def device_disassembly(obj):
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        with open(obj, "rb") as f, open(local, "wb") as g:
            g.write(f.read())
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        code = [os.path.join(tmp, n) for n in os.listdir(tmp) if "amdgcn" in n]
        if not code:
            raise RuntimeError("no gfx950 code object in %s" % obj)
        return subprocess.run([OBJDUMP, "-d", code[0]], check=True, capture_output=True, text=True).stdout


def kernels(text):
    name, body = None, []
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name and "\t" in ln:
            body.append(ln.split("\t")[1].split("//")[0].strip())
    if name:
        yield name, body


def template_args(name):
    m = re.search(r"conv3_f16x3p_kernelI(.*?)EEvNS", name)
    if not m:
        return None
    return [int(a[2:]) for a in re.findall(r"L[ib]\d+", m.group(1))]      # MF, INACT, STAMP, STAGE, S, SPLIT, RPWT, PF, FLAT, WG2


def check(body):
    """-> list of (vmcnt, vector-memory instructions since the last weight DMA) for every first wait behind a DMA."""
    out, since = [], None
    for ins in body:
        op = ins.split()[0] if ins else ""
        if op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in ins):
            since = 0
        elif op.startswith(("global_load", "global_store", "buffer_load", "buffer_store", "global_atomic", "flat_")):
            if since is not None:
                since += 1
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m and since is not None:
                out.append((int(m.group(1)), since))
                since = None
    return out


def handoff_waits(body):
    """Second view, for schedules whose blocks the compiler laid out apart (the patch ring): every `s_waitcnt vmcnt(N > 0)` that is
    followed within 8 instructions by the slot-publishing ds_write_b32 is a hand-off wait; walking BACK from it, the vector-memory
    instructions up to the nearest weight DMA are counted (forward exec-skips in between are transparent; a backward branch or another
    wait ends the walk). -> list of (N, count): count >= 0 = loads between the DMA and the wait; count < 0 = -(loads between the
    previous wait / loop head and this wait), the DMA being laid out in another block -- then N <= |count| is the NECESSARY condition this
    tool can still hold the code to (N loads do precede the wait; that the DMA precedes THEM is the compiler barrier's job in the source)."""
    out = []
    for i, ins in enumerate(body):
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)", ins)
        if not m or int(m.group(1)) == 0:
            continue
        if not any(b.startswith("ds_write_b32") for b in body[i + 1:i + 9]):
            continue
        n, cnt, found = int(m.group(1)), 0, None
        for j in range(i - 1, max(-1, i - 400), -1):
            op = body[j].split()[0] if body[j] else ""
            if op.startswith("global_load_lds"):
                found = cnt
                break
            if op.startswith(("global_load", "global_store", "buffer_load", "buffer_store")):
                cnt += 1
            elif op == "s_waitcnt" and "vmcnt" in body[j]:
                break
            elif op.startswith("s_cbranch") or op == "s_branch":
                off = int(body[j].split()[1])
                if off > 32767:                      # backward
                    break
        out.append((n, found if found is not None else -cnt))      # negative: loads counted back to the block's head, the DMA is laid out elsewhere
    return out


def marked_waits(body):
    """Third view, exact (round 6): the hand-written counted waits carry expcnt(6) (conv3_f16x3p_kernel.h: p3_waitcnt) -- no compiler-made
    wait does. `s_waitcnt vmcnt(N) expcnt(6)` leaves the N youngest vector-memory requests in flight and everything older complete; what
    the code then publishes or reads (a weight DMA, or -- split roles, PF = 3, whose patch waves issue no DMA -- the register set requested
    before the new one) is older only if at least N vector-memory instructions lie between the previous vmcnt wait (or the loop head, or
    the DMA) and this wait. -> list of (N, count)."""
    out = []
    for i, ins in enumerate(body):
        m = re.match(r"s_waitcnt vmcnt\((\d+)\) expcnt\(6\)", ins)
        if not m:
            continue
        cnt = 0
        for j in range(i - 1, max(-1, i - 400), -1):
            op = body[j].split()[0] if body[j] else ""
            if op.startswith("global_load_lds"):
                break
            if op.startswith(("global_load", "global_store", "buffer_load", "buffer_store")):
                cnt += 1
            elif op == "s_waitcnt" and "vmcnt" in body[j]:
                break
            elif (op.startswith("s_cbranch") or op == "s_branch") and int(body[j].split()[1]) > 32767:
                break
        if int(m.group(1)):                 # (a marked vmcnt(0): the compiler merged its own full drain into the counted wait of a tail step)
            out.append((int(m.group(1)), cnt))
    return out


def main(objs):
    bad = judged = 0
    for obj in objs:
        for name, body in kernels(device_disassembly(obj)):
            args = template_args(name)
            if args is None:
                continue
            res = check(body)
            hand = handoff_waits(body)
            marked = marked_waits(body)
            split = len(args) > 5 and args[5] == 1
            counted = [r for r in res if r[0] > 0]
            viol = [r for r in res if r[0] > r[1]] + [r for r in hand if r[0] > abs(r[1])] + [("marked",) + r for r in marked if r[0] > r[1]]
            stage, pf = (args[3] if len(args) > 3 else 0), (args[7] if len(args) > 7 else 0)
            if not marked and pf in (1, 3):      # the register-prefetch and split-roles schedules are built on a counted wait: it must be there
                viol.append(("no marked counted wait found",))
            tag = "not judged (pre-split inputs: the patches are DMAs)" if split else ("VIOLATION %s" % viol if viol else "ok")
            print("%-22s %-42s waits behind a weight DMA: %2d, counted (N, loads since the DMA): %-12s hand-off waits (N, loads back to the DMA; negative: back to the block head): %-22s marked waits (N, loads in front): %s  %s" % (
                os.path.basename(obj), "<" + ", ".join(str(a) for a in args) + ">", len(res), sorted(set(counted)) or "-", sorted(set(hand), key=str) or "-",
                sorted(set(marked)) or "-", tag))
            if not split:
                judged += 1
                bad += 1 if viol else 0
    print("%d instantiations judged, %d with a counted wait that does not cover what it publishes" % (judged, bad))
    return 1 if bad or not judged else 0


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.exit(main(sys.argv[1:] or [os.path.join(root, "lssvc_amd", "csrc", n) for n in ("conv3_f16x3p.o", "conv3_f16x3p_r.o", "conv3_f16x3p_r2.o", "conv3_f16x3p_r3.o")]))
