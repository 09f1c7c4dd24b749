#!/usr/bin/env python3
"""Host-side simulation of the LGKM (LDS read) bookkeeping of csrc/field_bf16w.hip::dense_w for every layer shape of the bf16 program:
all reads are volatile asm in a fixed issue order and return in order, and the wait of step I is lgkmcnt(Sched::cnt(I)).  Checks that
at every step the A fragment it consumes has landed, that a tile's bias batch has landed at the tile's first step, and that the
count fits the 4-bit counter.  (Run before touching the schedule; the kernel's static_asserts only cover the counter width.)"""
AP = 4


def pick_G32(KB, NT):
    g = min(32 // KB, NT)
    while NT % g:
        g -= 1
    return g


def sim(KB, NT32):
    G = pick_G32(KB, NT32)
    STEPS, TOTAL, NCH = KB * 2, G * KB * 2, NT32 // G
    q, maxcnt = [("bias", 0, i) for i in range(4)], 0       # tile 0's bias: read before the first prologue
    for C in range(NCH):
        T0 = C * G
        bias_at = lambda s: 0 <= s < TOTAL and s % STEPS == 0 and T0 + s // STEPS + 1 < NT32
        aread_at = lambda s: s >= 0 and s + AP < TOTAL

        def cnt(I):
            n, lo = (min(AP, TOTAL) - 1 - I, 0) if I < AP else (0, I - AP + 1)
            return n + sum((1 if aread_at(s) else 0) + (4 if bias_at(s) else 0) for s in range(lo, I))

        q += [("A", C, i) for i in range(min(AP, TOTAL))]
        for I in range(TOTAL):
            c = cnt(I)
            maxcnt = max(maxcnt, c)
            assert c <= 15
            while len(q) > c:        # s_waitcnt lgkmcnt(c): all but the newest c reads have returned (in order)
                q.pop(0)
            t, k = T0 + I // STEPS, I % STEPS
            assert ("A", C, I) not in q, ("A fragment in flight", KB, NT32, C, I)
            assert k != 0 or not any(o[0] == "bias" and o[1] == t for o in q), ("bias in flight", KB, NT32, C, I)
            if bias_at(I):
                q += [("bias", t + 1, i) for i in range(4)]
            if aread_at(I):
                q.append(("A", C, I + AP))
    return maxcnt


if __name__ == "__main__":
    for KB, NT in [(2, 4), (4, 4), (6, 4), (2, 2), (4, 2), (3, 8), (8, 8), (11, 8), (10, 4), (8, 4)]:
        print("KB32 %2d NT32 %d: G %d, max lgkmcnt %d: ok" % (KB, NT, pick_G32(KB, NT), sim(KB, NT)))
