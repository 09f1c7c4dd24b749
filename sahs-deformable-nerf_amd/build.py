"""Build the C-ABI HIP library (libsahs_nerf.so) in-tree for gfx950.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the
working tree.  ``python -m`` style use: ``python sahs-deformable-nerf_amd/build.py [--force]``.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsahs_nerf.so")
SOURCES = ["capi.hip", "pack.hip", "render_ops.hip", "field_f32.hip", "field_bf16.hip", "field_bwd.hip", "train_bwd.hip"]
# sources built again for the NeRFaceModel architectures (csrc/sahs_model.hpp: -DSAHS_MODEL=1 / 2, symbols suffixed _nf / _ns)
MODEL_SOURCES = ["pack.hip", "field_f32.hip", "field_bwd.hip"]
# field kernels: no sNaN-quieting v_max before every fmaxf (activations); NaNs still propagate through the MFMAs
FIELD_FLAGS = ["-fno-honor-nans", "-mno-amdgpu-ieee"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-math-errno", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "sahs_nerf.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, defines=(), out=None):
    """defines/out: ablation variants (tools/ablate.py) -- extra -D flags, separate output .so."""
    global LIB
    if out is not None:
        LIB, force = out, True
    if not (force or _stale()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for src, model in [(s, 0) for s in SOURCES] + [(s, m) for m in (1, 2) for s in MODEL_SOURCES]:
        obj = os.path.join(HERE, "build", (os.path.basename(out) + "." if out else "") + src.replace(".hip", ".m%d.o" % model if model else ".o"))
        objs.append(obj)
        extra = (FIELD_FLAGS if src.startswith("field_") else []) + (["-DSAHS_MODEL=%d" % model] if model else [])
        cmd = [hipcc] + FLAGS + extra + ["-D" + d for d in defines] + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
