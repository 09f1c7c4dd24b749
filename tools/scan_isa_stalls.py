#!/usr/bin/env python3
"""What stands between the MFMAs of a kernel, read off its ISA: branches inside the matrix region and full drains of the vector-memory
counter in the middle of a chunk.  Two round-4 findings came from exactly this listing: wave-uniform guards inside an unrolled MFMA loop
compiled to a vcc branch between any two MFMAs (narrow f32 weight-gradient kernel: 0.65 -> 0.55 ms per launch once the K loop was
instantiated per shape), and ordinary global loads used while LDS-DMA is in flight, which the compiler guards with `s_waitcnt vmcnt(0)` --
a drain of the weight prefetch (f32 backward chains: 16 per sample tile until the sign words were parked in LDS).

    python tools/scan_isa_stalls.py csrc/field_bwd_chain_f32.hip [kernel-name-substring] [-DSAHS_MODEL=1 ...]

Compiles the source for gfx950 with the library's flags (device side only, -S) and prints per kernel: MFMAs, instructions and branches
between the first and the last MFMA, `vmcnt(0)` waits that are not part of a barrier (with the plain load they most likely wait for and how
many MFMAs into the kernel body they sit)."""
import importlib.util
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "sahs-deformable-nerf_amd")


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    src = sys.argv[1] if os.path.exists(sys.argv[1]) else os.path.join(PKG, sys.argv[1])
    pat = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    spec = importlib.util.spec_from_file_location("sahs_build", os.path.join(PKG, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    model = next((int(a.split("=")[1]) for a in extra if a.startswith("-DSAHS_MODEL=")), 0)
    out = os.path.join(tempfile.mkdtemp(prefix="isa_"), "k.s")
    cmd = b._compile_cmd(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), os.path.basename(src), model, []) + \
        [a for a in extra if not a.startswith("-DSAHS_MODEL=")] + ["--cuda-device-only", "-S", src, "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        if pat not in m.group(1):
            continue
        body = [l.strip().split(";")[0].strip() for l in m.group(2).split("\n")]
        body = [l for l in body if l and not l.startswith(".") and not l.endswith(":")]
        idx = [i for i, l in enumerate(body) if l.startswith("v_mfma")]
        if not idx:
            continue
        seg = body[idx[0]:idx[-1] + 1]
        print("%s\n    %d MFMAs; between the first and the last: %d instructions, %d branches, %d s_barrier" % (
            m.group(1), len(idx), len(seg), sum(l.startswith("s_cbranch") for l in seg), sum(l.startswith("s_barrier") for l in seg)))
        nm = 0
        for i, l in enumerate(body):
            nm += l.startswith("v_mfma")
            if l.startswith("s_waitcnt") and "vmcnt(0)" in l and not any(x.startswith("s_barrier") for x in body[i + 1:i + 4]):
                j = i
                while j > 0 and not (body[j].startswith(("global_load", "scratch_load", "buffer_load")) and "lds" not in body[j]):
                    j -= 1
                print("    vmcnt(0) outside a barrier after %5d MFMAs; nearest plain load %d instructions earlier: %s" % (nm, i - j, body[j][:60]))


if __name__ == "__main__":
    main()
