#!/usr/bin/env python3
"""Static check of a hand-scheduled kernel's asm LDS reads (runs here, no GPU): between a `ds_read_b128` and the next `s_waitcnt lgkmcnt`
nothing may read or overwrite its destination registers.  The reads are `asm volatile` with a counted wait far behind them, so the
compiler does not know the data is still on its way: a destination it believes dead (a fragment no MFMA uses) is handed out again at once
and then overwritten by the landing data -- that is how the first 16x16x32 port put a garbage address into an LDS-DMA and faulted.

    hipcc --offload-arch=gfx950 -O3 -std=c++20 <flags of build.py> --cuda-device-only -S csrc/field_bf16w.hip -o /tmp/w.s
    python tools/check_lds_inflight.py /tmp/w.s field_forward_bf16w_kernel
(conservative: it stops at the FIRST lgkmcnt wait after a read, which may not be the one that retires it)"""
import re
import sys


def check(asm_text, pattern):
    """-> (number of ds_read_b128 in the functions matching pattern, list of (function, read, offending instruction))"""
    tot, bad = 0, []

    def regs(tok):
        mm = re.match(r"v\[(\d+):(\d+)\]", tok)
        if mm:
            return set(range(int(mm.group(1)), int(mm.group(2)) + 1))
        mm = re.match(r"v(\d+)$", tok)
        return {int(mm.group(1))} if mm else set()

    for m in re.finditer(r"^(\S*%s\S*):[^\n]*\n(.*?)^\.Lfunc_end\d+:" % re.escape(pattern), asm_text, re.S | re.M):
        body = [l.strip() for l in m.group(2).split("\n") if l.startswith("\t") and not l.strip().startswith((";", "."))]
        for i, l in enumerate(body):
            if not l.startswith("ds_read_b128"):
                continue
            tot += 1
            dst = regs(l.split()[1].rstrip(","))
            for j in range(i + 1, min(i + 600, len(body))):
                lj = body[j]
                if lj.startswith("s_waitcnt") and "lgkmcnt" in lj:
                    break
                ops = lj.replace(",", " ").split()
                if len(ops) < 2:
                    continue
                srcs = set()
                for tok in ops[2:]:
                    srcs |= regs(tok)
                if (srcs & dst) or ((regs(ops[1]) & dst) and not lj.startswith("ds_read")):
                    bad.append((m.group(1), l, lj))
                    break
    return tot, bad


if __name__ == "__main__":
    tot, bad = check(open(sys.argv[1]).read(), sys.argv[2])
    for fn, rd, ins in bad:
        print(fn[:70], "|", rd, "| touched by:", ins)
    print("ds_read_b128:", tot, "touched before a wait:", len(bad))
    sys.exit(1 if bad else 0)
