#!/usr/bin/env python3
"""Static check of a hand-scheduled kernel's asm LDS reads (runs on the compiled ISA, no GPU): between a `ds_read_b128` and the
`s_waitcnt lgkmcnt(n)` that RETIRES it, no instruction may read or write its destination registers.

The reads are `asm volatile` with a counted wait several MFMA steps behind them; the compiler does not know the data is still on its
way, so a destination it believes free could be copied, split or handed to another value and then be overwritten by the landing data --
that is how the first 16x16x32 port put a garbage address into an LDS-DMA and faulted.  csrc/bf16_pipe.hpp makes
the dependence explicit in the source (the retiring wait takes the destination as a read-write operand); this check confirms it on the
object that is actually linked.

Which wait retires a read: LDS operations of a wave complete in issue order and `s_waitcnt lgkmcnt(n)` returns once at most n LGKM
operations are outstanding.  If the read were still outstanding, so would be the k LDS operations issued after it: outstanding >= k + 1.
Hence the first wait with n <= k retires it.  (Scalar-memory loads also count in lgkmcnt but may return out of order; they are not
counted in k, which only makes the test stricter.)  The scan is over the function's instructions in textual order and gives up after
MAX_SCAN instructions or at the function's end -- a read that is never retired is reported as such.

    hipcc --offload-arch=gfx950 -O3 -std=c++20 <flags of build.py> --cuda-device-only -S csrc/field_bf16w.hip -o /tmp/w.s
    python tools/check_lds_inflight.py /tmp/w.s field_forward_bf16w_kernel
build.py runs this on every build of the kernels listed in its HAND_SCHEDULED table."""
import re
import sys

MAX_SCAN = 4000


def _regs(tok):
    mm = re.match(r"v\[(\d+):(\d+)\]", tok)
    if mm:
        return set(range(int(mm.group(1)), int(mm.group(2)) + 1))
    mm = re.match(r"v(\d+)$", tok)
    return {int(mm.group(1))} if mm else set()


def _lgkm_of_wait(ins):
    """n of an `s_waitcnt ... lgkmcnt(n)`; a wait that names other counters only leaves lgkmcnt at its maximum (15)."""
    mm = re.search(r"lgkmcnt\((\d+)\)", ins)
    if mm:
        return int(mm.group(1))
    mm = re.match(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", ins)      # raw immediate: lgkmcnt is bits 11:8
    if mm:
        return (int(mm.group(1), 0) >> 8) & 15
    return 15


def check(asm_text, pattern):
    """-> (number of ds_read_b128 in the functions matching pattern, list of (function, read, offending instruction or reason),
           histogram {instructions between a read and its retiring wait: count})"""
    tot, bad, dist = 0, [], {}
    for m in re.finditer(r"^(\S*%s\S*):[^\n]*\n(.*?)^\.Lfunc_end\d+:" % re.escape(pattern), asm_text, re.S | re.M):
        body = [l.strip() for l in m.group(2).split("\n") if l.startswith("\t") and not l.strip().startswith((";", "."))]
        for i, l in enumerate(body):
            if not l.startswith("ds_read_b128"):
                continue
            tot += 1
            dst = _regs(l.split()[1].rstrip(","))
            k, retired = 0, False
            for j in range(i + 1, min(i + 1 + MAX_SCAN, len(body))):
                lj = body[j]
                if lj.startswith("s_waitcnt"):
                    if _lgkm_of_wait(lj) <= k:
                        retired = True
                        dist[j - i] = dist.get(j - i, 0) + 1
                        break
                    continue
                if lj.startswith("s_endpgm"):
                    break
                ops = lj.replace(",", " ").split()
                if len(ops) >= 2:
                    srcs = set()
                    for tok in ops[2:]:
                        srcs |= _regs(tok)
                    if (srcs & dst) or (_regs(ops[1]) & dst):
                        bad.append((m.group(1), l, "touched in flight by: " + lj))
                        retired = True
                        break
                if lj.startswith("ds_"):
                    k += 1
            if not retired:
                bad.append((m.group(1), l, "no retiring wait found"))
    return tot, bad, dist


if __name__ == "__main__":
    tot, bad, dist = check(open(sys.argv[1]).read(), sys.argv[2])
    for fn, rd, why in bad:
        print(fn[:70], "|", rd, "|", why)
    far = sorted(dist)[-1] if dist else 0
    print("ds_read_b128:", tot, "violations:", len(bad), "| longest in-flight window: %d instructions" % far)
    sys.exit(1 if bad else 0)
