"""ctypes front-end of the CPU oracle (``sahs_oracle.c``).

TEST INFRASTRUCTURE ONLY -- see the header of ``sahs_oracle.c``.  Imported by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``; never by the product
package.  All arrays are numpy fp32, C-contiguous.
"""
import ctypes
import os
import subprocess

import numpy as np

import contextlib

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}
_SO = {"audio": "liboracle.so", "nerface": "liboracle_nerface.so", "nerface_static": "liboracle_nerface_static.so"}   # SAHS_MODEL=0/1/2
_MODEL = "audio"
_F = ctypes.POINTER(ctypes.c_float)
_I64 = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    src = os.path.join(_HERE, "sahs_oracle.c")
    for name in _SO.values():
        so = os.path.join(_HERE, name)
        if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", name] + (["-B"] if force else []))
    return os.path.join(_HERE, _SO["audio"])


@contextlib.contextmanager
def model(name):
    """Select the architecture the calls inside the block restate: "audio" (AudioFaceModel) or "nerface" (NeRFaceModel)."""
    global _MODEL
    assert name in _SO, name
    prev, _MODEL = _MODEL, name
    try:
        yield
    finally:
        _MODEL = prev


def lib():
    L = _LIBS.get(_MODEL)
    if L is None:
        so = os.path.join(_HERE, _SO[_MODEL])
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.oracle_param_count.restype = ctypes.c_long
        assert L.oracle_model() == list(_SO).index(_MODEL)
        _LIBS[_MODEL] = L
    return L


def _f(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(_F)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def param_count():
    return int(lib().oracle_param_count())


def get_ray_bundle(H, W, intrinsics, c2w):
    """nerf_helpers.py:178-233 -> (ro, rd) each (H, W, 3)."""
    c2w = _c(c2w)
    intr = _c(np.asarray(intrinsics, dtype=np.float32).reshape(-1))
    if intr.size < 4:  # nerf_helpers.py:219-220
        intr = np.array([intr[0], intr[0], 0.5, 0.5], dtype=np.float32)
    ro = np.empty((H, W, 3), np.float32)
    rd = np.empty((H, W, 3), np.float32)
    lib().oracle_get_ray_bundle(ctypes.c_int(H), ctypes.c_int(W), _f(intr), _f(c2w), ctypes.c_int(c2w.shape[-1]), _f(ro), _f(rd))
    return ro, rd


def audionet(flat, audio):
    """AudioNet (audio model) -> driving76; for the NeRFaceModel the 76-d expression is the driving vector itself."""
    audio = _c(audio)
    assert audio.shape == ((16, 29) if _MODEL == "audio" else (76,))
    out = np.empty(76, np.float32)
    lib().oracle_audionet(_f(flat), _f(audio), _f(out))
    return out


def pose_encoding(pose):
    pose = _c(pose)
    out = np.empty(36, np.float32)
    lib().oracle_pose_encoding(_f(pose), ctypes.c_int(pose.shape[-1]), _f(out))
    return out


def positional_encoding(x, L, include_input=True):
    x = _c(x)
    d = x.shape[-1]
    n = int(np.prod(x.shape[:-1]))
    width = (d if include_input else 0) + 2 * d * L
    out = np.empty((n, width), np.float32)
    xf = x.reshape(n, d)
    for i in range(n):
        lib().oracle_positional_encoding(_f(xf[i]), ctypes.c_int(d), ctypes.c_int(L), ctypes.c_int(int(include_input)), _f(out[i]))
    return out.reshape(x.shape[:-1] + (width,))


def field_forward(flat, level, x, driving76, pose36, debug=False):
    """AudioFaceModel.forward over rows x=(P,>=6) -> raw (P,16) [+ dx (P,3), w (P,2), grid (P,32)]."""
    x = _c(x)
    P, xs = x.shape
    raw = np.empty((P, 16), np.float32)
    dx = np.empty((P, 3), np.float32) if debug else None
    w = np.empty((P, {"audio": 2, "nerface": 1, "nerface_static": 0}[_MODEL]), np.float32) if debug else None
    g = np.empty((P, 32), np.float32) if debug else None
    lib().oracle_field_forward(_f(flat), ctypes.c_int(int(level)), ctypes.c_long(P), _f(x), ctypes.c_int(xs),
                               _f(_c(driving76)), _f(_c(pose36)), _f(raw), _f(dx), _f(w), _f(g))
    return (raw, dx, w, g) if debug else raw


def ray_uniforms(seed, stream_id, ray0, N, S):
    """sahs_ray_uniforms restated: (N,S) uniforms keyed by (seed, stream, global ray index, sample)."""
    out = np.empty((N, S), np.float32)
    lib().oracle_ray_uniforms(ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), ctypes.c_int(int(stream_id)), ctypes.c_long(int(ray0)),
                              ctypes.c_long(int(N)), ctypes.c_int(int(S)), _f(out))
    return out


def stratified_depths(near, far, S, lindisp=False, t_rand=None):
    near = _c(np.asarray(near).reshape(-1))
    far = _c(np.asarray(far).reshape(-1))
    N = near.shape[0]
    z = np.empty((N, S), np.float32)
    lib().oracle_stratified_depths(ctypes.c_long(N), ctypes.c_int(S), _f(near), _f(far), ctypes.c_int(int(lindisp)), _f(_c(t_rand)), _f(z))
    return z


def composite(raw, z, rd, noise=None, bg=None, white_background=False):
    """volume_rendering_utils.py:7-78 (+ bg overwrite train_utils.py:135-136). raw is copied."""
    raw = np.array(raw, dtype=np.float32, order="C", copy=True)
    N, S, _ = raw.shape
    z, rd, noise, bg = _c(z), _c(rd), _c(noise), _c(bg)
    rgb = np.empty((N, 15), np.float32)
    disp = np.empty(N, np.float32)
    acc = np.empty(N, np.float32)
    wts = np.empty((N, S), np.float32)
    depth = np.empty(N, np.float32)
    lib().oracle_composite(ctypes.c_long(N), ctypes.c_int(S), _f(raw), _f(z), _f(rd), _f(noise), _f(bg),
                           ctypes.c_int(int(white_background)), _f(rgb), _f(disp), _f(acc), _f(wts), _f(depth))
    return rgb, disp, acc, wts, depth


def aten_sum(x):
    """torch.sum of a 1-D fp32 row in ATen's CPU summation order (sahs_oracle.c: aten_sum_f32)."""
    x = _c(x)
    f = lib().oracle_aten_sum_f32
    f.restype = ctypes.c_float
    return np.float32(f(_f(x), ctypes.c_int(x.size)))


def linspace01(n):
    """torch.linspace(0, 1, n), bit for bit (sahs_oracle.c: aten_linspace01)."""
    out = np.empty(n, np.float32)
    lib().oracle_linspace01(ctypes.c_int(n), _f(out))
    return out


def sample_pdf_2(bins, weights, num_samples, u=None):
    """nerf_helpers.py:454-497; u=None is det=True. Returns (samples, inds)."""
    bins, weights, u = _c(bins), _c(weights), _c(u)
    N, nb = bins.shape
    assert weights.shape == (N, nb - 1)
    out = np.empty((N, num_samples), np.float32)
    inds = np.empty((N, num_samples), np.int64)
    lib().oracle_sample_pdf_2(ctypes.c_long(N), ctypes.c_int(nb), ctypes.c_int(num_samples), _f(bins), _f(weights), _f(u), _f(out),
                              inds.ctypes.data_as(_I64))
    return out, inds


def resample(z, weights, num_fine, u=None):
    """train_utils.py:157-166: mids, sample_pdf_2, cat, sort -> (z_samples, z_sorted, inds)."""
    z, weights, u = _c(z), _c(weights), _c(u)
    N, S = z.shape
    zs = np.empty((N, num_fine), np.float32)
    zo = np.empty((N, S + num_fine), np.float32)
    inds = np.empty((N, num_fine), np.int64)
    lib().oracle_resample(ctypes.c_long(N), ctypes.c_int(S), ctypes.c_int(num_fine), _f(z), _f(weights), _f(u), _f(zs), _f(zo),
                          inds.ctypes.data_as(_I64))
    return zs, zo, inds


def render_rays(flat, rays, num_coarse, num_fine, driving76, pose36, bg=None, t_rand=None, noise_c=None,
                u=None, noise_f=None, lindisp=False, white_background=False, want_aux=False):
    """predict_and_render_radiance (train_utils.py:72-206) for one ray chunk; rays (N,>=8)."""
    rays = _c(rays)
    N, rs = rays.shape
    Sf = num_coarse + num_fine
    o = dict(rgb_c=np.empty((N, 15), np.float32), disp_c=np.empty(N, np.float32), acc_c=np.empty(N, np.float32),
             rgb_f=np.empty((N, 15), np.float32), disp_f=np.empty(N, np.float32), acc_f=np.empty(N, np.float32),
             w_bg=np.empty(N, np.float32), depth_f=np.empty(N, np.float32))
    zf = np.empty((N, Sf), np.float32) if want_aux else None
    wc = np.empty((N, num_coarse), np.float32) if want_aux else None
    lib().oracle_render_rays(_f(flat), ctypes.c_long(N), _f(rays), ctypes.c_int(rs), ctypes.c_int(num_coarse), ctypes.c_int(num_fine),
                             ctypes.c_int(int(lindisp)), ctypes.c_int(int(white_background)), _f(_c(driving76)), _f(_c(pose36)),
                             _f(_c(bg)), _f(_c(t_rand)), _f(_c(noise_c)), _f(_c(u)), _f(_c(noise_f)),
                             _f(o["rgb_c"]), _f(o["disp_c"]), _f(o["acc_c"]), _f(o["rgb_f"]), _f(o["disp_f"]), _f(o["acc_f"]),
                             _f(o["w_bg"]), _f(o["depth_f"]), _f(zf), _f(wc))
    if want_aux:
        o["z_fine"], o["weights_c"] = zf, wc
    return o


def run_one_iter_of_nerf(flat, ro, rd, near, far, num_coarse, num_fine, audio, pose, background_prior=None,
                         chunksize=131072, rand=None, lindisp=False, white_background=False):
    """run_one_iter_of_nerf (train_utils.py:209-321), flat outputs (train-mode shapes).

    ``rand`` is a list with one dict per ray chunk holding that chunk's explicit random
    tensors (keys t_rand, noise_c, u, noise_f; missing/None = draw skipped), i.e. the
    reference's RNG stream captured in draw order.
    """
    ro = _c(np.asarray(ro).reshape(-1, 3))
    rd = _c(np.asarray(rd).reshape(-1, 3))
    R = ro.shape[0]
    rays = np.concatenate([ro, rd, np.full((R, 1), near, np.float32), np.full((R, 1), far, np.float32)], axis=1).astype(np.float32)
    driving = audionet(flat, audio)
    pose36 = pose_encoding(pose)
    outs = []
    for ci, s in enumerate(range(0, R, chunksize)):
        e = min(R, s + chunksize)
        rnd = (rand[ci] if rand is not None else {}) or {}
        bg = None if background_prior is None else np.asarray(background_prior, np.float32).reshape(-1, 15)[s:e]
        outs.append(render_rays(flat, rays[s:e], num_coarse, num_fine, driving, pose36, bg=bg,
                                t_rand=rnd.get("t_rand"), noise_c=rnd.get("noise_c"), u=rnd.get("u"), noise_f=rnd.get("noise_f"),
                                lindisp=lindisp, white_background=white_background))
    keys = ["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"]
    return tuple(np.concatenate([o[k] for o in outs], axis=0) for k in keys)
