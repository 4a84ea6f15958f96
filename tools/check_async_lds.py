#!/usr/bin/env python3
"""Static check of the GEMM device assembly: a register written by an (inline-asm) ds_read must not be read before the
next `s_waitcnt lgkmcnt(0)` of the same basic block.  The pipelined kernel issues its LDS reads early and waits by hand;
to hipcc the asm has produced its value at once, so nothing but this check (and the numerics tests) would notice a
register copy scheduled before the data has landed.   usage: check_async_lds.py file.s"""
import re
import sys


def check(path):
    txt = open(path).read()
    total = 0
    for m in re.finditer(r"^(_ZN3mml1[68]gemm_p(?:ipe|lanes)_kernel\w+):.*?s_endpgm", txt, re.S | re.M):
        name, pending, bad = m.group(1), set(), 0
        for ln in m.group(0).split("\n"):
            t = ln.strip()
            if not t or t.startswith(";"):
                continue
            if t.startswith(".LBB") or t.startswith("s_cbranch") or t.startswith("s_branch"):
                pending = set()  # only same-block ordering is checked (other paths run other code)
                continue
            op = t.split()[0]
            regs = set()
            for a, b in re.findall(r"v\[(\d+):(\d+)\]", t):
                regs |= set(range(int(a), int(b) + 1))
            regs |= {int(a) for a in re.findall(r"\bv(\d+)\b", t)}
            if op.startswith("ds_read"):
                d = re.match(r"\S+\s+(?:v\[(\d+):(\d+)\]|v(\d+))", t)
                pending |= set(range(int(d.group(1)), int(d.group(2)) + 1)) if d.group(1) else {int(d.group(3))}
                continue
            if op == "s_waitcnt" and "lgkmcnt(0)" in t:
                pending = set()
                continue
            if regs & pending:
                bad += 1
                if bad <= 3:
                    print(f"{name}: use before wait: {t[:100]}")
        total += bad
    return total


if __name__ == "__main__":
    n = check(sys.argv[1])
    print("async-LDS violations:", n)
    sys.exit(1 if n else 0)
