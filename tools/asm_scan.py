"""Compressed view of the memory / wait structure of the kernels in a gfx950 assembly file (hipcc -save-temps): per kernel,
the sequence of global loads (G), stores (S), LDS reads / writes (L / l), MFMA runs (M<n>), other VALU runs (v<n>), scratch
accesses (SCR) and the s_waitcnt operands -- the view in which a store that waits for the previous store, a prefetch that is
drained by a spill reload, or an LDS read issued right in front of its consumer are visible at a glance:
    python tools/asm_scan.py file.s [kernel-name-substring]"""
import re
import sys


def scan(path, want=None):
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for a, name in starts:
        if want and want not in name:
            continue
        b = next(i for i in range(a, len(lines)) if ".Lfunc_end" in lines[i])
        out, run, kind = [], 0, None

        def flush():
            nonlocal run, kind
            if run:
                out.append("%s%d" % (kind, run) if run > 1 else kind)
            run, kind = 0, None
        for l in lines[a:b]:
            t = l.strip()
            if not t or t.startswith((";", ".")):
                if re.match(r"^\.LBB", t):
                    flush()
                    out.append("\n  " + t.split(":")[0] + ":")
                continue
            op = t.split()[0]
            if op.startswith("v_mfma"):
                k = "M"
            elif op.startswith("global_load") or op.startswith("buffer_load"):
                k = "G"
            elif op.startswith("global_store") or op.startswith("buffer_store"):
                k = "S"
            elif op.startswith("ds_read") or op.startswith("ds_load"):
                k = "L"
            elif op.startswith("ds_write") or op.startswith("ds_store"):
                k = "l"
            elif op.startswith("scratch"):
                k = "SCR"
            elif op.startswith("s_waitcnt"):
                flush()
                out.append("w[" + t.split(None, 1)[1].replace(" ", "").replace("vmcnt", "vm").replace("lgkmcnt", "lgkm") + "]")
                continue
            elif op.startswith("s_barrier"):
                k = "BAR"
            elif op.startswith("s_cbranch"):
                flush()
                out.append("br(" + t.split()[-1] + ")")
                continue
            elif op.startswith("v_"):
                k = "v"
            else:
                continue
            if k != kind:
                flush()
                kind = k
            run += 1
        flush()
        print("==", name)
        print(" ".join(out))


if __name__ == "__main__":
    scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
