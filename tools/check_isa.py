#!/usr/bin/env python3
"""ISA audit of one kernel: which memory instructions and waits the compiler put into it (runs here, no GPU).

    python tools/check_isa.py field_bwd.hip gemm_dma_kernel
    python tools/check_isa.py field_bf16w.hip field_forward_bf16w_kernel

Compiles the source to gfx950 assembly with the product's flags (`build.py`), cuts out every function whose (mangled) name
contains the pattern and prints, per function: scratch size, VGPR/AGPR counts, and how many of each memory / wait instruction it
holds -- in total and inside its hottest loop (the back-edge span with the most MFMAs).  What it was written for: a counted
`s_waitcnt vmcnt(N)` behind LDS-DMA is only right if NO other VMEM instruction (spill reload, hoisted load) shares the wave's
stream; DESIGN.md section 7a records the audit of gemm_dma_kernel made with it.
"""
import importlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLASSES = [("lds_dma", r"global_load_lds|buffer_load.*\blds\b"), ("global_load", r"global_load_(?!lds)"), ("global_store", r"global_store|global_atomic"),
           ("scratch", r"scratch_"), ("buffer", r"buffer_(?!load.*\blds\b)"), ("flat", r"\bflat_"), ("ds_read", r"ds_read"), ("ds_write", r"ds_write"),
           ("mfma", r"v_mfma"), ("waitcnt_vm", r"s_waitcnt.*vmcnt"), ("waitcnt_lgkm", r"s_waitcnt.*lgkmcnt"), ("barrier", r"s_barrier")]


def count(lines):
    out = {}
    for name, pat in CLASSES:
        n = sum(1 for ln in lines if re.search(pat, ln))
        if n:
            out[name] = n
    return out


def main():
    src, pat = sys.argv[1], sys.argv[2]
    b = importlib.import_module("sahs-deformable-nerf_amd.build")
    path = os.path.join(b.CSRC, src)
    extra = (b.FIELD_FLAGS if src.startswith("field_") else []) + b.PER_FILE_FLAGS.get(src, []) + ["-D" + d for d in sys.argv[3:]]
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + b.FLAGS + extra + ["--cuda-device-only", "-S", path, "-o", asm], check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    # functions: "name:" ... ".Lfunc_end"
    for m in re.finditer(r"^(\S*%s\S*):[^\n]*\n(.*?)^\.Lfunc_end\d+:" % re.escape(pat), text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        ins = [ln.strip() for ln in body if ln.startswith("\t") and not ln.strip().startswith((".", ";"))]
        meta = re.search(r"\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel" % re.escape(name), text, re.S)
        info = {}
        if meta:
            for key in ("private_segment_fixed_size", "next_free_vgpr", "accum_offset", "group_segment_fixed_size"):
                k = re.search(r"\.amdhsa_%s (\S+)" % key, meta.group(1))
                if k:
                    info[key] = k.group(1)
        print("== %s\n   %s\n   whole function (%d instructions): %s" % (name, info, len(ins), count(ins)))
        # loops: a label followed later by a branch back to it; report the span with the most MFMAs
        labels = {}
        best = None
        for i, ln in enumerate(body):
            lm = re.match(r"^(\.LBB\d+_\d+):", ln)
            if lm:
                labels[lm.group(1)] = i
            bm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", ln)
            if bm:
                tgt = bm.group(1) or bm.group(2)
                if tgt in labels:
                    span = [x.strip() for x in body[labels[tgt]:i + 1] if x.startswith("\t")]
                    c = count(span)
                    if best is None or c.get("mfma", 0) > best[1].get("mfma", 0):
                        best = (tgt, c, len(span))
        if best:
            print("   hottest loop %s (%d instructions): %s" % (best[0], best[2], best[1]))


if __name__ == "__main__":
    main()
