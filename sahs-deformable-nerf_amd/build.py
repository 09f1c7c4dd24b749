"""Build the C-ABI HIP library (libsahs_nerf.so) in-tree for gfx950.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the
working tree.  ``python -m`` style use: ``python sahs-deformable-nerf_amd/build.py [--force]``.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsahs_nerf.so")
SOURCES = ["capi.hip", "pack.hip", "render_ops.hip", "spade_ops.hip", "field_f32.hip", "field_bf16w.hip", "field_bf16x3.hip", "field_bwd.hip", "field_bwd_chain.hip", "field_bwd_chain_f32.hip", "train_bwd.hip"]
# sources built again for the NeRFaceModel architectures (csrc/sahs_model.hpp: -DSAHS_MODEL=1 / 2, symbols suffixed _nf / _ns)
MODEL_SOURCES = ["pack.hip", "field_f32.hip", "field_bwd.hip"]
NERFACE_DEFORM_SOURCES = ["field_bf16x3.hip"]      # NeRFaceModel WITH deformation nets (SAHS_MODEL=1): their split-operand kernel (mixed precision)
MODEL1_SOURCES = ["field_bf16w.hip"]      # NeRFaceModel: the bf16 radiance nets (with deformation nets: those stay fp32; without: the whole net)
# field kernels: no sNaN-quieting v_max before every fmaxf (activations); NaNs still propagate through the MFMAs
FIELD_FLAGS = ["-fno-honor-nans", "-mno-amdgpu-ieee"]
# Kernels whose correctness or speed rests on an exact count of the wave's memory instructions (a counted s_waitcnt vmcnt behind LDS-DMA,
# a counted lgkmcnt behind hand-issued LDS reads) must not get scratch: a spill reload is a VMEM instruction the count does not know
# about, and a spilled in-flight destination would be stored before its data has landed.  The build reads the compiler's own resource
# remarks and FAILS if one of them has a non-zero scratch size (substring of the mangled name -> max bytes).  The headline fp32 forward
# kernel (audio model, no activation saving) is held to zero scratch as a performance guard (a reload's wait also drains its in-flight
# weight prefetch); its activation-saving and NeRFace builds do spill a little (build/<library>.resource_usage.txt): with the tile stores
# unconditional (f32_pipe.hpp: kStores) the saving audio kernels keep two plane pointers in scratch, reloaded twice per 128-sample tile --
# extra vector-memory operations can only make a counted wait stricter, never weaker.
NO_SCRATCH = {"gemm_dma_kernel": 0, "_ZN4sahs24field_forward_f32_kernelILb0E": 0, "field_forward_bf16w_kernel": 0,
              "field_radiance_bf16x3_kernel": 0, "field_deform_bf16x3_kernel": 0, "gemm_tn_split_kernel": 0, "gemm_tn_jobs_kernel": 0,
              "field_backward_chain_rad_kernel": 0, "field_backward_chain_def_kernel": 0, "_ZN4sahs24field_forward_f32_kernelILb1E": 96,
              "gemm_tn_jobs_f32_kernel": 0, "gemm_tn_jobs256_f32_kernel": 0,
              "field_backward_chain_rad_f32_kernel": 0, "field_backward_chain_def_f32_kernel": 0}
# Kernels with hand-issued `asm volatile ds_read_b128` + counted waits (csrc/bf16_pipe.hpp): (source, SAHS_MODEL, kernel name pattern).
# Every build compiles these to ISA as well and runs tools/check_lds_inflight.py on it: an object in which anything touches a read's
# destination before the wait that retires it is never linked.
HAND_SCHEDULED = [("field_bf16w.hip", 0, "field_forward_bf16w_kernel"), ("field_bf16w.hip", 1, "field_forward_bf16w_kernel"),
                  ("field_bf16w.hip", 2, "field_forward_bf16w_kernel"), ("field_bf16x3.hip", 0, "field_radiance_bf16x3_kernel"),
                  ("field_bf16x3.hip", 0, "field_deform_bf16x3_kernel"), ("field_bf16x3.hip", 1, "field_deform_bf16x3_kernel"),
                  ("field_bwd_chain.hip", 0, "field_backward_chain_rad_kernel"), ("field_bwd_chain.hip", 0, "field_backward_chain_def_kernel")]
# field_bf16w.hip (one wave per SIMD, 512 registers): MFMA accumulators must live in ARCH VGPRs.  Left to its heuristics the compiler
# puts them in AGPRs, and every accumulator value the activation code touches then costs a v_accvgpr_read -- which, unlike plain VALU
# work, does NOT hide under the wave's own MFMAs (tools/micro/mfma_valu_overlap.hip: 2 reads per MFMA = 55 cycles per MFMA instead of 36).
# field_bwd.hip: no SLP vectorisation -- left on, the compiler packs the operand splits' subtractions into v_pk_add_f32, which is an
# anti-lever beside MFMAs on this chip (MI355X_MICROARCH.md); without it the training step is 1.1 % faster (same-box A/B: 17.95 -> 17.75 ms).
# The forward kernels measure neutral (+-0.5 %) and keep the default.
PER_FILE_FLAGS = {"field_bwd.hip": ["-fno-slp-vectorize"],
                  "field_bf16w.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "field_bf16x3.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
                  "field_bwd_chain.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-ffp-contract=off", "-fno-math-errno", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "sahs_nerf.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _resource_usage(remarks):
    """Parse clang's -Rpass-analysis=kernel-resource-usage remarks -> {mangled kernel name: dict}."""
    out, cur = {}, None
    keys = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ",
            "LDS Size [bytes/block]": "lds"}
    for line in remarks.splitlines():
        if "remark:" not in line:
            continue
        body = line.split("remark:", 1)[1].replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
        if body.startswith("Function Name:"):
            cur = body.split(":", 1)[1].strip()
            out[cur] = dict(sgpr=0, vgpr=0, agpr=0, scratch=0, occ=0, lds=0)
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            if k.strip() in keys:
                out[cur][keys[k.strip()]] = int(v)
    return out


def _compile_cmd(hipcc, src, model, defines):
    base = os.path.basename(src)
    extra = (FIELD_FLAGS if base.startswith("field_") else []) + (["-DSAHS_MODEL=%d" % model] if model else []) + \
            ([] if "SAHS_NOTHING" in defines else PER_FILE_FLAGS.get(src, []))
    return [hipcc] + FLAGS + extra + ["-D" + d for d in defines]


def build(force=False, verbose=False, defines=(), out=None, check_inflight=True):
    """defines/out: diagnostic variants (tools/ablate.py, tools/stamp_*.py) -- extra -D flags, separate output .so.  Every timing-only
    ablation or stamp in the sources sits behind SAHS_DIAG, which the shipped library (out=None) refuses."""
    global LIB
    if out is None and defines:
        raise RuntimeError("libsahs_nerf.so is built without extra defines (SAHS_DIAG variants go to their own file: out=...)")
    if out is not None:
        LIB, force = out, True
    if not (force or _stale()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    tag = os.path.basename(LIB)
    procs = []
    for src, model in [(s, 0) for s in SOURCES] + [(s, m) for m in (1, 2) for s in MODEL_SOURCES] + [(s, m) for m in (1, 2) for s in MODEL1_SOURCES] + [(s, 1) for s in NERFACE_DEFORM_SOURCES]:
        obj = os.path.join(bdir, (tag + "." if out else "") + os.path.basename(src).replace(".hip", ".m%d.o" % model if model else ".o"))
        objs.append(obj)
        cmd = _compile_cmd(hipcc, src, model, defines) + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    # the same translation units once more, device side only, as ISA text for the in-flight check (same flags => same code; the
    # register counts of the two compilations are compared below)
    isa = []
    if check_inflight:
        for src, model, pat in HAND_SCHEDULED:
            path = os.path.join(bdir, "%s.%s.m%d.s" % (tag, src.replace(".hip", ""), model))
            cmd = _compile_cmd(hipcc, src, model, defines) + ["--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", path]
            isa.append((src, model, pat, path, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    usage = {}
    for src, p in procs:
        err = p.communicate()[1]
        diag = [l for l in err.splitlines() if "[-Rpass-analysis=kernel-resource-usage]" not in l and not l.lstrip().startswith(("|", "^")) and
                not (l[:1] == " " and l.strip()[:1].isdigit() and "|" in l)]
        if diag and (verbose or p.returncode != 0):
            print("\n".join(diag), file=sys.stderr)
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on " + src)
        usage.update(_resource_usage(err))
    with open(os.path.join(bdir, tag + ".resource_usage.txt"), "w") as f:      # one table per output library
        for k in sorted(usage):
            f.write("%-110s vgpr %3d agpr %3d sgpr %3d scratch %4d lds %6d occupancy %d\n" % (
                k, usage[k]["vgpr"], usage[k]["agpr"], usage[k]["sgpr"], usage[k]["scratch"], usage[k]["lds"], usage[k]["occ"]))
    bad = ["%s: %d bytes/lane of scratch" % (k, v["scratch"]) for k, v in usage.items() for name, cap in NO_SCRATCH.items()
           if name in k and v["scratch"] > cap]
    if bad and not defines:      # ablation builds (defines) are diagnostics and may spill
        raise RuntimeError("kernels that rely on counted waits must not use scratch:\n  " + "\n  ".join(bad))
    if isa:
        sys.path.insert(0, os.path.join(HERE, "..", "tools"))
        import check_lds_inflight as chk
        report = []
        for src, model, pat, path, p in isa:
            if p.wait() != 0:
                raise RuntimeError("hipcc -S failed on %s (SAHS_MODEL=%d)" % (src, model))
            text = open(path).read()
            total, viol, dist = chk.check(text, pat)
            # the ISA checked must be the ISA linked: same register allocation as the object's kernels (unified file: accumulation
            # registers start at the arch-VGPR count rounded up to 4)
            for n in re.findall(r"^(\S*%s\S*):" % re.escape(pat), text, re.M):
                want, meta = usage.get(n), re.search(r"\.amdhsa_kernel %s\b.*?\.end_amdhsa_kernel" % re.escape(n), text, re.S)
                nxt = re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta.group(0)) if meta else None
                if want is not None and nxt is not None:
                    expect = want["vgpr"] if not want["agpr"] else (want["vgpr"] + 3) // 4 * 4 + want["agpr"]
                    if int(nxt.group(1)) != expect:
                        raise RuntimeError("in-flight check: the ISA of %s is not the object's (next_free_vgpr %s, object vgpr %d agpr %d)"
                                           % (n, nxt.group(1), want["vgpr"], want["agpr"]))
            report.append("%s (SAHS_MODEL=%d) %s: %d hand-issued ds_read_b128, %d violations, longest in-flight window %d instructions"
                          % (src, model, pat, total, len(viol), max(dist) if dist else 0))
            if viol or total == 0:
                raise RuntimeError("hand-issued LDS reads touched in flight (tools/check_lds_inflight.py) in %s SAHS_MODEL=%d:\n  %s"
                                   % (src, model, "\n  ".join("%s | %s" % (r, w) for _, r, w in viol[:8]) or "no reads found"))
        with open(os.path.join(bdir, tag + ".lds_inflight.txt"), "w") as f:
            f.write("\n".join(report) + "\n")
        if verbose:
            print("\n".join(report))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
