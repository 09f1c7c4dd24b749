"""Torch-facing ops over the C ABI: device memory and streams come from PyTorch, the arithmetic
is the HIP library's.  Every function requires CUDA(HIP) fp32 tensors and raises otherwise --
there is no eager/CPU path here.

autograd: ``RenderRaysFn`` (one ray chunk of predict_and_render_radiance) is the differentiable op: its
backward runs the HIP backward kernels (composite_backward, field_backward, conditioning_backward) and
returns gradients for the flat parameter buffer and the audio window.  Importance resampling is not
differentiated (the reference detaches it, train_utils.py:164).  The stand-alone seams are differentiable too:
``FieldFn`` behind model(level, x, driving, pose) (parameters and driving input) and ``CompositeFn`` behind
volume_render_radiance_field (the radiance field), so the reference's own python driver can be run over them.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import SAHS_F32, SAHS_BF16, SAHS_BF16X3, check

PRECISIONS = {"fp32": SAHS_F32, "f32": SAHS_F32, "bf16": SAHS_BF16,
              "bf16x3": SAHS_BF16X3}    # near-fp32 on the bf16 pipe: every net with hi + lo bf16 operands (3 MFMAs per product); SAHS_X3_DEFORM=f32 keeps the deformation nets on the fp32 kernel


def is_mixed(arch, precision):
    """Precisions that exist as the split chain only (a deformation launch with split bf16 operands -- fp32 under SAHS_X3_DEFORM=f32 -- and a
    low-precision radiance launch, exchanging x', w)."""
    return (precision == SAHS_BF16 and arch == "nerface") or (precision == SAHS_BF16X3 and arch == "audio")


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    """The launch stream: torch's current stream of the CURRENT device.  Every tensor handed to an op must live on that device
    (_req enforces it), so a kernel is never enqueued on another device's stream or launched without that device's attributes."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t, name, dtype=torch.float32):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.SahsError("%s must be a GPU tensor (the HIP path has no CPU fallback)" % name)
    if t.dtype != dtype:
        raise _lib.SahsError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if t.device.index != torch.cuda.current_device():
        raise _lib.SahsError("%s lives on %s but the current device is cuda:%d: one process drives one GPU (call "
                             "torch.cuda.set_device first, as torchrun-style launchers do)" % (name, t.device, torch.cuda.current_device()))
    return t if t.is_contiguous() else t.contiguous()


ARCHS = ("audio", "nerface", "nerface_static")   # = SAHS_MODEL_AUDIO / _NERFACE / _NERFACE_STATIC of include/sahs_nerf.h


def _fn(name, arch="audio"):
    """sahs_model_<name> bound to the architecture: AudioFaceModel (config/audio) | NeRFaceModel (config/expression person_2/3)
    | NeRFaceModel without deformation (config/expression/person_1)."""
    if arch not in ARCHS:
        raise _lib.SahsError("unknown architecture %r" % (arch,))
    full = "sahs_model_" + name
    f = getattr(_lib.lib(), full)
    m = ARCHS.index(arch)
    return (lambda *a: f(m, *a)), full + "(%s)" % arch


def param_count(arch="audio"):
    return int(_fn("param_count", arch)[0]())


def executed_macs_per_sample(arch="audio", precision=SAHS_F32, part=0):
    """MACs per sample evaluation the field kernel issues to the matrix pipe (padded tiles, constants folded away); part 1 / 2: the
    deformation nets / the radiance net alone (the split evaluation)."""
    return int(_fn("executed_macs_part", arch)[0](precision, part))


def pack_weights(flat, precision=SAHS_F32, arch="audio"):
    flat = _req(flat, "flat_params")
    if flat.numel() != param_count(arch):
        raise _lib.SahsError("flat_params has %d values, expected %d" % (flat.numel(), param_count(arch)))
    words = _fn("packed_words", arch)[0](precision)
    if words <= 0:
        raise _lib.SahsError("precision %r is not built for architecture %r" % (precision, arch))
    packed = torch.empty(words, dtype=torch.float32, device=flat.device)
    f, name = _fn("pack_weights", arch)
    check(f(_p(flat), _p(packed), precision, _stream()), name)
    return packed


def fold_conditioning(flat, audio, pose, arch="audio"):
    """audio: the (16, 29) DeepSpeech window (AudioFaceModel) or the 76-d expression vector (NeRFaceModel)."""
    flat, audio = _req(flat, "flat_params"), _req(audio, "audio")
    pose = _req(pose, "pose")
    want = (16, 29) if arch == "audio" else (76,)   # NeRFaceModels: the expression vector
    if tuple(audio.shape) != want:
        raise _lib.SahsError("driving input must be %s for %r, got %s" % (want, arch, tuple(audio.shape)))
    if pose.dim() != 2 or pose.shape[0] < 3 or pose.shape[1] != 4:
        raise _lib.SahsError("pose must be (3|4, 4), got %s" % (tuple(pose.shape),))
    frame = torch.empty(_fn("frame_words", arch)[0](), dtype=torch.float32, device=flat.device)
    f, name = _fn("fold_conditioning", arch)
    check(f(_p(flat), _p(audio), _p(pose), 4, _p(frame), _stream()), name)
    return frame


def get_ray_bundle(height, width, intrinsics, c2w):
    c2w = _req(c2w, "tform_cam2world")
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    ro = torch.empty(height, width, 3, dtype=torch.float32, device=c2w.device)
    rd = torch.empty_like(ro)
    check(_lib.lib().sahs_get_ray_bundle(int(height), int(width), fx, fy, cx, cy, _p(c2w), int(c2w.shape[-1]), _p(ro), _p(rd), _stream()),
          "sahs_get_ray_bundle")
    return ro, rd


def ray_uniforms(seed, stream_id, ray0, num_rays, num_samples, device):
    """(num_rays, num_samples) uniforms in [0,1) keyed by (seed, stream_id, global ray index ray0 + r, sample): the same ray
    gets the same draws however the frame is chunked or sharded."""
    out = torch.empty(int(num_rays), int(num_samples), dtype=torch.float32, device=device)
    check(_lib.lib().sahs_ray_uniforms(int(seed) & (2 ** 64 - 1), int(stream_id), int(ray0), int(num_rays), int(num_samples), _p(out), _stream()),
          "sahs_ray_uniforms")
    return out


def stratified_depths(rays, num_samples, lindisp=False, t_rand=None):
    rays, t_rand = _req(rays, "rays"), _req(t_rand, "t_rand")
    N = rays.shape[0]
    z = torch.empty(N, num_samples, dtype=torch.float32, device=rays.device)
    check(_lib.lib().sahs_stratified_depths(N, int(num_samples), _p(rays), int(rays.shape[1]), int(bool(lindisp)), _p(t_rand), _p(z),
                                             _stream()), "sahs_stratified_depths")
    return z


def field_forward(packed, frame, level, rays, z, precision=SAHS_F32, debug=False, out=None, arch="audio"):
    """raw (N,S,16) for level 0/1 at points ro + rd*z.  debug=True also returns (dx, w, grid)."""
    packed, frame, rays, z = _req(packed, "packed"), _req(frame, "frame"), _req(rays, "rays"), _req(z, "z")
    N, S = z.shape
    if rays.shape[0] != N or rays.shape[1] < 8:
        raise _lib.SahsError("rays must be (N, >=8) with N == z.shape[0]")
    raw = out if out is not None else torch.empty(N, S, 16, dtype=torch.float32, device=z.device)
    dbg = torch.zeros(N * S * 88, dtype=torch.float32, device=z.device) if debug else None
    f, name = _fn("field_forward", arch)
    check(f(_p(packed), _p(frame), int(level), N, S, _p(rays), int(rays.shape[1]), _p(z), _p(raw), _p(dbg), precision, _stream()), name)
    if debug == "full":
        return raw, dbg[: N * S * 56].view(N * S, 56), dbg[N * S * 56:].view(N * S, 32)
    if debug:
        a = dbg[: N * S * 56].view(N * S, 56)
        return raw, a[:, 0:3].reshape(N, S, 3), a[:, 3:5].reshape(N, S, 2), dbg[N * S * 56:].view(N, S, 32)
    return raw


def composite_forward(raw, z, rays, noise=None, bg=None, white_background=False):
    raw, z, rays, noise, bg = _req(raw, "radiance_field"), _req(z, "depth_values"), _req(rays, "rays"), _req(noise, "noise"), _req(bg, "background_prior")
    N, S = z.shape
    dev = z.device
    rgb = torch.empty(N, 15, dtype=torch.float32, device=dev)
    disp, acc, depth = (torch.empty(N, dtype=torch.float32, device=dev) for _ in range(3))
    weights = torch.empty(N, S, dtype=torch.float32, device=dev)
    check(_lib.lib().sahs_composite_forward(N, S, _p(raw), _p(z), _p(rays), int(rays.shape[1]), _p(noise), _p(bg), int(bool(white_background)),
                                             _p(rgb), _p(disp), _p(acc), _p(weights), _p(depth), _stream()), "sahs_composite_forward")
    return rgb, disp, acc, weights, depth


def resample(z, weights, num_fine, u=None, want_aux=False):
    z, weights, u = _req(z, "z_vals"), _req(weights, "weights"), _req(u, "u")
    N, S = z.shape
    z_out = torch.empty(N, S + num_fine, dtype=torch.float32, device=z.device)
    zs = torch.empty(N, num_fine, dtype=torch.float32, device=z.device) if want_aux else None
    inds = torch.empty(N, num_fine, dtype=torch.int64, device=z.device) if want_aux else None
    check(_lib.lib().sahs_resample(N, S, int(num_fine), _p(z), _p(weights), _p(u), _p(zs), _p(z_out), _p(inds), _stream()), "sahs_resample")
    return (z_out, zs, inds) if want_aux else z_out


def resample_merge(z, weights, num_fine, u=None):
    """resample that also returns the new samples and the merge permutation: (z_sorted (N,S+nf), z_new (N,nf), src (N,S+nf) int32)."""
    z, weights, u = _req(z, "z_vals"), _req(weights, "weights"), _req(u, "u")
    N, S = z.shape
    z_out = torch.empty(N, S + num_fine, dtype=torch.float32, device=z.device)
    z_new = torch.empty(N, num_fine, dtype=torch.float32, device=z.device)
    src = torch.empty(N, S + num_fine, dtype=torch.int32, device=z.device)
    check(_lib.lib().sahs_resample_merge(N, S, int(num_fine), _p(z), _p(weights), _p(u), _p(z_new), _p(z_out), _p(src), _stream()), "sahs_resample_merge")
    return z_out, z_new, src


FIELD_ALL, FIELD_DEFORM, FIELD_RADIANCE = 0, 1, 2


def field_forward_split(packed, frame, level, mode, rays, xw, z=None, src=None, xw_col0=0, out=None, arch="audio", precision=SAHS_F32,
                        validate_src=False):
    """The field in parts (include/sahs_nerf.h: sahs_model_field_forward_split).  xw: (N, row, 8) fp32 buffer of deformed points.
    FIELD_ALL: raw (N,S,16) for depths z, x'/w of its samples written to xw[:, xw_col0:xw_col0+S]; FIELD_DEFORM: only x'/w for depths z;
    FIELD_RADIANCE: raw for the samples xw[ray, src[ray, s]] (src (N,S) int32).  The kernel does not range-check src (precondition of the
    C ABI: 0 <= src < xw.shape[1]); validate_src=True checks a caller-made permutation here (one device reduction + sync)."""
    packed, frame, rays, z = _req(packed, "packed"), _req(frame, "frame"), _req(rays, "rays"), _req(z, "z")
    src = _req(src, "src", torch.int32)
    xw = _req(xw, "xw")
    N = rays.shape[0]
    S = src.shape[1] if mode == FIELD_RADIANCE else z.shape[1]
    if xw.dim() != 3 or xw.shape[0] != N or xw.shape[2] != 8:
        raise _lib.SahsError("xw must be (N, row, 8)")
    if src is not None and (src.dim() != 2 or src.shape[0] != N):
        raise _lib.SahsError("src must be (N, S)")
    if validate_src and src is not None and src.numel() and not (0 <= int(src.min()) and int(src.max()) < xw.shape[1]):
        raise _lib.SahsError("src indexes outside xw's %d slots per ray" % xw.shape[1])
    raw = None
    if mode != FIELD_DEFORM:
        raw = out if out is not None else torch.empty(N, S, 16, dtype=torch.float32, device=rays.device)
    f, name = _fn("field_forward_split", arch)
    check(f(_p(packed), _p(frame), int(precision), int(level), int(mode), N, int(S), _p(rays), int(rays.shape[1]), _p(z), _p(raw), _p(xw), int(xw.shape[1]), int(xw_col0),
            _p(src), _stream()), name)
    return raw


def sample_pdf(bins, weights, num_samples, u=None, want_inds=False):
    bins, weights, u = _req(bins, "bins"), _req(weights, "weights"), _req(u, "u")
    N, nb = bins.shape
    if tuple(weights.shape) != (N, nb - 1):
        raise _lib.SahsError("weights must be (N, nb-1)")
    out = torch.empty(N, num_samples, dtype=torch.float32, device=bins.device)
    inds = torch.empty(N, num_samples, dtype=torch.int64, device=bins.device) if want_inds else None
    check(_lib.lib().sahs_sample_pdf(N, nb, int(num_samples), _p(bins), _p(weights), _p(u), _p(out), _p(inds), _stream()), "sahs_sample_pdf")
    return (out, inds) if want_inds else out


def render_rays(packed, frame, rays, num_coarse, num_fine, precision=SAHS_F32, lindisp=False, white_background=False, bg=None,
                t_rand=None, noise_c=None, u=None, noise_f=None, workspace=None, arch="audio"):
    """predict_and_render_radiance for one ray chunk -> the reference's 8-tuple (flat shapes)."""
    if is_mixed(arch, precision):      # mixed precision exists as the row-writing split chain only
        rows = torch.empty(rays.shape[0], ROW_COLUMNS, dtype=torch.float32, device=rays.device)
        render_rays_rows(packed, frame, rays, num_coarse, num_fine, rows, precision=precision, lindisp=lindisp, white_background=white_background,
                         bg=bg, t_rand=t_rand, noise_c=noise_c, u=u, noise_f=noise_f, workspace=workspace, arch=arch)
        return tuple(c.contiguous() for c in (rows[:, 0:15], rows[:, 15], rows[:, 16], rows[:, 17:32], rows[:, 32], rows[:, 33], rows[:, 34], rows[:, 35]))
    packed, frame, rays = _req(packed, "packed"), _req(frame, "frame"), _req(rays, "rays")
    bg, t_rand, noise_c, u, noise_f = (_req(t, n) for t, n in ((bg, "background_prior"), (t_rand, "t_rand"), (noise_c, "noise_c"),
                                                               (u, "u"), (noise_f, "noise_f")))
    N = rays.shape[0]
    dev = rays.device
    Sf = num_coarse + num_fine
    ws = workspace if workspace is not None else {}

    def buf(name, *shape):
        t = ws.get(name)
        if t is None or tuple(t.shape) != shape or t.device != dev:
            t = torch.empty(*shape, dtype=torch.float32, device=dev)
            ws[name] = t
        return t

    z_c, z_f = buf("z_c", N, num_coarse), buf("z_f", N, Sf)
    raw, weights = buf("raw", N, Sf, 16), buf("weights", N, Sf)
    rgb_c, rgb_f = (torch.empty(N, 15, dtype=torch.float32, device=dev) for _ in range(2))
    disp_c, acc_c, disp_f, acc_f, w_bg, depth_f = (torch.empty(N, dtype=torch.float32, device=dev) for _ in range(6))
    f, name = _fn("render_rays", arch)
    check(f(_p(packed), _p(frame), precision, N, _p(rays), int(rays.shape[1]), int(num_coarse), int(num_fine),
            int(bool(lindisp)), int(bool(white_background)), _p(bg), _p(t_rand), _p(noise_c), _p(u), _p(noise_f),
            _p(z_c), _p(z_f), _p(raw), _p(weights), _p(rgb_c), _p(disp_c), _p(acc_c), _p(rgb_f), _p(disp_f),
            _p(acc_f), _p(w_bg), _p(depth_f), _stream()), name)
    if num_fine > 0:
        return rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f, w_bg, depth_f
    return rgb_c, disp_c, acc_c, None, None, None, w_bg, depth_f


ROW_COLUMNS = 36      # SAHS_ROW_* of include/sahs_nerf.h: rgb_c 0:15, disp_c 15, acc_c 16, rgb_f 17:32, disp_f 32, acc_f 33, w_bg 34, depth_f 35


def composite_forward_rows(raw, z, rays, rows, fine_pass, noise=None, bg=None, white_background=False, weights=None):
    """composite_forward writing its ray outputs into the (N, 36) row block ``rows`` (coarse pass: columns 0..16; fine pass: 17..35);
    returns the dense (N, S) weights."""
    raw, z, rays, noise, bg = _req(raw, "radiance_field"), _req(z, "depth_values"), _req(rays, "rays"), _req(noise, "noise"), _req(bg, "background_prior")
    N, S = z.shape
    if not (rows.is_cuda and rows.dtype == torch.float32 and rows.dim() == 2 and rows.shape[0] == N and rows.shape[1] >= ROW_COLUMNS and rows.stride(1) == 1):
        raise _lib.SahsError("rows must be a GPU fp32 (N, >=36) tensor with unit column stride")
    if weights is None or tuple(weights.shape) != (N, S):
        weights = torch.empty(N, S, dtype=torch.float32, device=z.device)
    check(_lib.lib().sahs_composite_forward_rows(N, S, _p(raw), _p(z), _p(rays), int(rays.shape[1]), _p(noise), _p(bg), int(bool(white_background)),
                                                  _p(weights), _p(rows), int(rows.stride(0)), int(bool(fine_pass)), _stream()), "sahs_composite_forward_rows")
    return weights


def render_rays_rows(packed, frame, rays, num_coarse, num_fine, rows, precision=SAHS_F32, lindisp=False, white_background=False, bg=None,
                     t_rand=None, noise_c=None, u=None, noise_f=None, workspace=None, arch="audio", share_deformation=True):
    """predict_and_render_radiance for one ray chunk, written IN PLACE into ``rows`` (N, 36): the 8-tuple of every ray side by
    side (a row block of the frame's (R, 36) buffer, which is also what the multi-GPU all-gather moves), so a chunk loop needs
    no per-chunk concatenation.  Returns ``rows``.  share_deformation (fp32; bf16 for the audio model): evaluate the deformation nets once per depth -- the fine
    pass reuses the coarse samples' deformed points instead of recomputing them as the reference does; identical results."""
    packed, frame, rays = _req(packed, "packed"), _req(frame, "frame"), _req(rays, "rays")
    bg, t_rand, noise_c, u, noise_f = (_req(t, n) for t, n in ((bg, "background_prior"), (t_rand, "t_rand"), (noise_c, "noise_c"),
                                                               (u, "u"), (noise_f, "noise_f")))
    N = rays.shape[0]
    if not (isinstance(rows, torch.Tensor) and rows.is_cuda and rows.dtype == torch.float32 and rows.dim() == 2 and rows.shape[0] == N
            and rows.shape[1] >= ROW_COLUMNS and rows.stride(1) == 1):
        raise _lib.SahsError("rows must be a GPU fp32 (N, >=36) tensor with unit column stride")
    dev = rays.device
    Sf = num_coarse + num_fine
    ws = workspace if workspace is not None else {}

    def buf(name, *shape):
        t = ws.get(name)
        if t is None or tuple(t.shape) != shape or t.device != dev:
            t = torch.empty(*shape, dtype=torch.float32, device=dev)
            ws[name] = t
        return t

    z_c, z_f = buf("z_c", N, num_coarse), buf("z_f", N, Sf)
    raw, weights = buf("raw", N, Sf, 16), buf("weights", N, Sf)
    xw = src = z_new = None
    mixed = is_mixed(arch, precision)      # split-operand deformation launch + low-precision radiance launch: only the split chain exists
    if mixed and not (share_deformation and num_fine > 0):
        raise _lib.SahsError("a mixed-precision model renders through the split chain (share_deformation=True, num_fine > 0)")
    if share_deformation and num_fine > 0 and arch != "nerface_static" and precision in (SAHS_F32, SAHS_BF16, SAHS_BF16X3):
        # extra workspace of the split evaluation: deformed points of every depth, the merge permutation, the new depths
        xw, z_new = buf("xw", N, Sf, 8), buf("z_new", N, num_fine)
        src = ws.get("src")
        if src is None or tuple(src.shape) != (N, Sf) or src.device != dev:
            src = ws["src"] = torch.empty(N, Sf, dtype=torch.int32, device=dev)
    f, name = _fn("render_rays_rows", arch)
    check(f(_p(packed), _p(frame), precision, N, _p(rays), int(rays.shape[1]), int(num_coarse), int(num_fine),
            int(bool(lindisp)), int(bool(white_background)), _p(bg), _p(t_rand), _p(noise_c), _p(u), _p(noise_f),
            _p(z_c), _p(z_f), _p(raw), _p(weights), _p(rows), int(rows.stride(0)), _p(xw), _p(src), _p(z_new), _stream()), name)
    return rows


def spade_modulate(x, gamma, beta, eps=1e-5, slope=1.0):
    """lrelu_slope(InstanceNorm2d(x) * (1 + gamma) + beta) for NCHW tensors of one shape, fused (include/sahs_nerf.h: sahs_spade_modulate)."""
    x, gamma, beta = _req(x, "x"), _req(gamma, "gamma"), _req(beta, "beta")
    if x.dim() != 4 or gamma.shape != x.shape or beta.shape != x.shape:
        raise _lib.SahsError("spade_modulate: x, gamma, beta must be NCHW tensors of one shape, got %s %s %s" % (tuple(x.shape), tuple(gamma.shape), tuple(beta.shape)))
    planes, hw = x.shape[0] * x.shape[1], x.shape[2] * x.shape[3]
    out = torch.empty_like(x)
    stats = torch.empty(int(_lib.lib().sahs_spade_modulate_workspace_words(planes)), dtype=torch.float32, device=x.device)
    check(_lib.lib().sahs_spade_modulate(planes, hw, _p(x), _p(gamma), _p(beta), float(eps), float(slope), _p(out), _p(stats), _stream()), "sahs_spade_modulate")
    return out


class LaunchProbe:
    """HIP events around every FIELD-kernel launch the library makes on this thread while the block is open (include/sahs_nerf.h:
    sahs_probe_*), recorded on the launch stream: per-kernel times of the product's own call chain (bench.py's roofline).
    ``records()`` -> list of dict(model, level, part (0 whole network | 1 deformation nets | 2 radiance nets), precision (of the kernel
    launched), samples, ms); it waits for the probed launches to finish."""

    def __init__(self, capacity=4096):
        self.capacity = int(capacity)

    def __enter__(self):
        check(_lib.lib().sahs_probe_arm(self.capacity), "sahs_probe_arm")
        return self

    def __exit__(self, *exc):
        _lib.lib().sahs_probe_disarm()
        return False

    @staticmethod
    def records():
        L = _lib.lib()
        if L.sahs_probe_dropped():
            raise _lib.SahsError("launch probe overflow: %d launches were not recorded" % L.sahs_probe_dropped())
        out = []
        kind, samples, ms = ctypes.c_int(), ctypes.c_long(), ctypes.c_float()
        for i in range(L.sahs_probe_count()):
            check(L.sahs_probe_read(i, ctypes.byref(kind), ctypes.byref(samples), ctypes.byref(ms)), "sahs_probe_read")
            k = kind.value
            out.append(dict(model=ARCHS[k >> 16], level=(k >> 12) & 15, part=(k >> 8) & 15, precision=k & 255, samples=samples.value, ms=ms.value))
        return out


# ---------------------------------------------------------------------------------------------------------
# training path
# ---------------------------------------------------------------------------------------------------------
def field_forward_save(packed, frame, level, rays, z, arch="audio"):
    """fp32 field forward that also returns the saved activations for field_backward: ONE buffer of P * act_words floats laid out
    as a dense [P x width] plane per layer (plane c starts at float c * P; sahs_layout.hpp, namespace act) -- NOT one row per
    sample, so it can only be handed to field_backward whole, with the same P."""
    packed, frame, rays, z = _req(packed, "packed"), _req(frame, "frame"), _req(rays, "rays"), _req(z, "z")
    N, S = z.shape
    raw = torch.empty(N, S, 16, dtype=torch.float32, device=z.device)
    act = torch.empty(N * S, _fn("act_words_per_sample", arch)[0](), dtype=torch.float32, device=z.device)
    f, name = _fn("field_forward_save", arch)
    check(f(_p(packed), _p(frame), int(level), N, S, _p(rays), int(rays.shape[1]), _p(z), _p(raw), _p(act), _stream()), name)
    return raw, act


def field_backward(flat, frame, level, act, d_raw, grad_flat, grad_cond, arch="audio"):
    flat, frame, act, d_raw = _req(flat, "flat_params"), _req(frame, "frame"), _req(act, "act"), _req(d_raw, "d_raw")
    P = act.shape[0]
    ws = torch.empty(_fn("field_backward_workspace_words", arch)[0](P), dtype=torch.float32, device=act.device)
    f, name = _fn("field_backward", arch)
    check(f(_p(flat), _p(frame), int(level), P, _p(act), _p(d_raw), _p(grad_flat), _p(grad_cond), _p(ws), _stream()), name)


def alloc_sign_bits(num_samples, mode, arch, device):
    """Buffer for the sign-bit planes a saving forward of `mode` writes beside the activations (include/sahs_nerf.h:
    sahs_model_field_forward_split_save_bits) -- the derivative masks of the fused backward walk; None where the architecture's backward
    does not read them."""
    words = int(_fn("bits_words_part", arch)[0](int(mode)))
    return torch.empty(int(num_samples), words, dtype=torch.int32, device=device) if words > 0 else None


def field_forward_split_save(packed, frame, level, mode, rays, xw, z=None, src=None, xw_col0=0, arch="audio", bits=None, precision=SAHS_F32, whole=None):
    """field_forward_split (fp32) that also keeps the activations of the layers it runs -> (raw or None, act).  act is the part's
    own buffer, (P, act_words_part(mode)) floats as dense per-layer planes; only field_backward_split of the same part reads it.
    bits (from alloc_sign_bits, same mode): also filled -- hand it to field_backward_split with act.
    precision SAHS_BF16X3 (AudioFaceModel, modes FIELD_DEFORM / FIELD_RADIANCE, bits required): the same buffers written by the
    split-operand kernels; `packed` is then pack_weights(flat, SAHS_BF16X3).
    whole=(act, bits) of a WHOLE-network save (FIELD_ALL shapes): the launch fills `mode`'s part of them instead of buffers of its own
    (a saved array of column c starts at c * P in every save), so a FIELD_DEFORM + FIELD_RADIANCE pair leaves what one FIELD_ALL launch
    leaves -> (raw or None, whole act)."""
    packed, frame, rays, z = _req(packed, "packed"), _req(frame, "frame"), _req(rays, "rays"), _req(z, "z")
    src, xw = _req(src, "src", torch.int32), _req(xw, "xw")
    N = rays.shape[0]
    S = src.shape[1] if mode == FIELD_RADIANCE else z.shape[1]
    if xw.dim() != 3 or xw.shape[0] != N or xw.shape[2] != 8:
        raise _lib.SahsError("xw must be (N, row, 8)")
    raw = None if mode == FIELD_DEFORM else torch.empty(N, S, 16, dtype=torch.float32, device=rays.device)
    words, bwords = _fn("act_words_part", arch)[0], _fn("bits_words_part", arch)[0]
    if whole is not None:
        act, bits = _req(whole[0], "whole act"), _req(whole[1], "whole bits", torch.int32)
        if int(mode) not in (FIELD_DEFORM, FIELD_RADIANCE) or tuple(act.shape) != (N * S, int(words(FIELD_ALL))) or tuple(bits.shape) != (N * S, int(bwords(FIELD_ALL))):
            raise _lib.SahsError("field_forward_split_save(whole): buffers of a FIELD_ALL save of the same samples, filled by a FIELD_DEFORM or FIELD_RADIANCE launch")
        radiance = int(mode) == FIELD_RADIANCE      # the radiance arrays / planes are the tail of the whole-network tables
        act_ptr = ctypes.c_void_p(act.data_ptr() + (4 * (int(words(FIELD_ALL)) - int(words(FIELD_RADIANCE))) * N * S if radiance else 0))
        bits_ptr = ctypes.c_void_p(bits.data_ptr() + (4 * int(bwords(FIELD_DEFORM)) * N * S if radiance else 0))
        f, name = _fn("field_forward_split_save_bits_x3" if int(precision) == SAHS_BF16X3 else "field_forward_split_save_bits", arch)
        check(f(_p(packed), _p(frame), int(level), int(mode), N, int(S), _p(rays), int(rays.shape[1]), _p(z), _p(raw), _p(xw), int(xw.shape[1]), int(xw_col0),
                _p(src), act_ptr, bits_ptr, _stream()), name)
        return raw, act
    act = torch.empty(N * S, words(int(mode)), dtype=torch.float32, device=rays.device)
    if bits is not None:
        bits = _req(bits, "bits", torch.int32)
        if tuple(bits.shape) != (N * S, int(bwords(int(mode)))):
            raise _lib.SahsError("field_forward_split_save: bits must come from alloc_sign_bits(N * S, mode, arch, device)")
        if int(precision) == SAHS_BF16X3 and arch != "audio":
            raise _lib.SahsError("field_forward_split_save: the split-operand saving forward is built for the AudioFaceModel only")
        f, name = _fn("field_forward_split_save_bits_x3" if int(precision) == SAHS_BF16X3 else "field_forward_split_save_bits", arch)
        check(f(_p(packed), _p(frame), int(level), int(mode), N, int(S), _p(rays), int(rays.shape[1]), _p(z), _p(raw), _p(xw), int(xw.shape[1]), int(xw_col0),
                _p(src), _p(act), _p(bits), _stream()), name)
        return raw, act
    if int(precision) != SAHS_F32:
        raise _lib.SahsError("field_forward_split_save: a saving forward at a precision other than fp32 needs the sign bits (bits=alloc_sign_bits(...))")
    f, name = _fn("field_forward_split_save", arch)
    check(f(_p(packed), _p(frame), int(level), int(mode), N, int(S), _p(rays), int(rays.shape[1]), _p(z), _p(raw), _p(xw), int(xw.shape[1]), int(xw_col0),
            _p(src), _p(act), _stream()), name)
    return raw, act


def fused_backward(enable=None):
    """The fused backward walk (one data-gradient chain launch + one or two weight-gradient launches per part: include/sahs_nerf.h,
    sahs_model_field_backward_fused) is taken whenever a backward is given sign bits and the architecture has it (AudioFaceModel), in the
    arithmetic backward_gemm_precision() names -- split-bf16 operands or exact fp32 products; fused_backward(False) keeps the per-layer walk
    (the A/B reference).  None queries."""
    global _FUSED_BACKWARD
    if enable is not None:
        _FUSED_BACKWARD = bool(enable)
    return _FUSED_BACKWARD


_FUSED_BACKWARD = os.environ.get("SAHS_BWD_FUSED", "1") != "0"


def training_forward_precision(precision=None):
    """Arithmetic of the SAVING forward launches of a training step (RenderRaysFn's kept path, AudioFaceModel): "fp32" (default: fp32 MFMAs,
    the form the parity tests pin) or "bf16x3" (the split-operand kernels of the SAHS_BF16X3 frame, which then also write the saved
    activations and sign bits; values within a few 1e-6 relative of the fp32 kernel's).  SAHS_TRAIN_FORWARD in the environment selects the
    initial value.  None queries."""
    global _TRAIN_FORWARD
    if precision is not None:
        if precision not in ("fp32", "bf16x3"):
            raise _lib.SahsError("training_forward_precision: 'fp32' or 'bf16x3'")
        _TRAIN_FORWARD = precision
    return _TRAIN_FORWARD


_TRAIN_FORWARD = os.environ.get("SAHS_TRAIN_FORWARD", "fp32")
if _TRAIN_FORWARD not in ("fp32", "bf16x3"):
    raise _lib.SahsError("SAHS_TRAIN_FORWARD must be 'fp32' or 'bf16x3', not %r" % (_TRAIN_FORWARD,))
_IDENTITY_SRC = {}


def _identity_src(N, S, device):
    """src of a radiance launch on the samples of its own pass in order (the coarse pass through the split kernels)"""
    key = (int(N), int(S), str(device))
    if key not in _IDENTITY_SRC:
        _IDENTITY_SRC.clear()
        _IDENTITY_SRC[key] = torch.arange(S, dtype=torch.int32, device=device).repeat(N, 1).contiguous()
    return _IDENTITY_SRC[key]


def field_backward_split(flat, frame, level, part, act, grad_flat, grad_cond, d_raw=None, xw_grad_in=None, arch="audio", full_act=False, bits=None):
    """Backward of `part` (FIELD_DEFORM, FIELD_RADIANCE, or 3 = everything) of the field over activations saved by the forward of that
    part.  The seam is d loss / d (x', w), (P,8): FIELD_RADIANCE returns it (the only part that writes one), FIELD_DEFORM starts from
    xw_grad_in, 3 adds xw_grad_in.  full_act: `act` was saved by a WHOLE-network forward and only `part` of it is walked (a saved array of
    column c starts at c * P in every save, so the part's arrays sit at their usual place behind the columns it does not use)."""
    flat, frame, act = _req(flat, "flat_params"), _req(frame, "frame"), _req(act, "act")
    d_raw, xw_grad_in = _req(d_raw, "d_raw"), _req(xw_grad_in, "xw_grad_in")
    P = act.shape[0]
    words = _fn("act_words_part", arch)[0]
    act_ptr = _p(act)
    if full_act and int(part) in (FIELD_DEFORM, FIELD_RADIANCE):
        if act.shape[1] != words(3):
            raise _lib.SahsError("field_backward_split(full_act): the activations were not saved by a whole-network forward")
        col0 = 0 if int(part) == FIELD_DEFORM else words(3) - words(FIELD_RADIANCE)      # the radiance arrays are the table's tail
        act_ptr = ctypes.c_void_p(act.data_ptr() + 4 * col0 * P)
    elif act.shape[1] != words(int(part)):
        raise _lib.SahsError("field_backward_split: the activations were not saved by a forward of part %d" % part)
    if xw_grad_in is not None and xw_grad_in.numel() != P * 8:
        raise _lib.SahsError("field_backward_split: xw_grad_in must hold (P,8)")
    out = torch.empty(P, 8, dtype=torch.float32, device=act.device) if part == FIELD_RADIANCE else None
    if bits is not None and _FUSED_BACKWARD and arch == "audio":
        bits = _req(bits, "bits", torch.int32)
        bw = lambda m: int(_fn("bits_words_part", arch)[0](int(m)))
        saved_mode = 0 if (full_act or int(part) == 3) else int(part)
        if tuple(bits.shape) != (P, bw(saved_mode)):
            raise _lib.SahsError("field_backward_split: bits were not written by the forward that saved these activations")
        # a whole-network save holds [deformation planes | radiance planes]; part 3 takes both, a part of it its own
        bits_ptr = ctypes.c_void_p(bits.data_ptr() + (4 * bw(FIELD_DEFORM) * P if (full_act and int(part) == FIELD_RADIANCE) else 0))
        words = int(_fn("field_backward_fused_workspace_words", arch)[0](int(part), P))
        ws = torch.empty(words, dtype=torch.float32, device=act.device)
        f, name = _fn("field_backward_fused", arch)
        check(f(_p(flat), _p(frame), int(level), int(part), P, act_ptr, bits_ptr, _p(d_raw), _p(xw_grad_in), _p(out), _p(grad_flat), _p(grad_cond), _p(ws),
                _stream()), name)
        return out
    ws = torch.empty(_fn("field_backward_workspace_words", arch)[0](P), dtype=torch.float32, device=act.device)
    f, name = _fn("field_backward_split", arch)
    check(f(_p(flat), _p(frame), int(level), int(part), P, act_ptr, _p(d_raw), _p(xw_grad_in), _p(out), _p(grad_flat), _p(grad_cond), _p(ws), _stream()), name)
    return out


_SIDE_STREAMS = {}


def _side_stream(dev):
    """One extra stream per device for the second of two independent backward walks (RenderRaysFn.backward)."""
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return _SIDE_STREAMS[key]


def route_xw_grad(src, g_fine, num_coarse):
    """(N,Sf) merge permutation, (N*Sf,8) seam gradient of the fine samples -> (N*Sc,8) for the coarse samples, (N*nf,8) for the new depths."""
    src, g_fine = _req(src, "src", torch.int32), _req(g_fine, "g_fine")
    N, Sf = src.shape
    nf = Sf - int(num_coarse)
    g_c = torch.empty(N * int(num_coarse), 8, dtype=torch.float32, device=src.device)
    g_n = torch.empty(N * nf, 8, dtype=torch.float32, device=src.device)
    check(_lib.lib().sahs_route_xw_grad(N, int(num_coarse), nf, _p(src), _p(g_fine), _p(g_c), _p(g_n), _stream()), "sahs_route_xw_grad")
    return g_c, g_n


def backward_gemm_precision(precision=None):
    """Arithmetic of the backward's dense-layer GEMMs: "bf16x3" (default: split operands on the bf16 matrix pipe, ~1e-5 of scale) or "fp32"
    (f32 MFMAs, the exact A/B reference).  None queries.  Process-wide (include/sahs_nerf.h: sahs_backward_gemm_precision)."""
    names = {SAHS_F32: "fp32", SAHS_BF16X3: "bf16x3"}
    if precision is None:
        return names[_lib.lib().sahs_backward_gemm_precision(-1)]
    code = PRECISIONS[precision] if isinstance(precision, str) else int(precision)
    if _lib.lib().sahs_backward_gemm_precision(code) != code:
        raise _lib.SahsError("backward GEMM precision must be 'fp32' or 'bf16x3'")
    return names[code]


def bf16_exact_leaky(enable=None):
    """The SAHS_BF16 kernels' LeakyReLU: False (default) = packed-integer form on the bf16 bit patterns (slope 0.0095 .. 0.0106), True = the
    reference's max(v, 0.01 v) in fp32 before rounding (include/sahs_nerf.h: sahs_bf16_exact_leaky).  None queries.  Process-wide."""
    return bool(_lib.lib().sahs_bf16_exact_leaky(-1 if enable is None else int(bool(enable))))


LOSS_STATS_WORDS = 64      # include/sahs_nerf.h: [0] loss, [1] last level's mse, [2:14] new sample_prob, [14:26] class counts, [26] rays


def stage1_loss_forward(map_coarse, map_fine, target, mask, class_weights):
    """The Stage-I objective (train_stage_rays_auto.py:455-468) of one ray batch in one launch -> stats (64,) fp32 (see LOSS_STATS_WORDS).
    map_*: (N,15) rendered [rgb3 | seg12] (either may be None), target (N, >=3), mask (N,12) one-hot, class_weights (12,)."""
    mc, mf = _req(map_coarse, "map_coarse"), _req(map_fine, "map_fine")
    target, mask, cw = _req(target, "target"), _req(mask, "mask"), _req(class_weights, "class_weights")
    ref = mc if mc is not None else mf
    N = ref.shape[0]
    if any(t is not None and tuple(t.shape) != (N, 15) for t in (mc, mf)) or tuple(mask.shape) != (N, 12) or target.shape[0] != N or cw.numel() != 12:
        raise _lib.SahsError("stage1_loss_forward: maps (N,15), target (N,>=3), mask (N,12), class_weights (12,)")
    stats = torch.zeros(LOSS_STATS_WORDS, dtype=torch.float32, device=ref.device)
    check(_lib.lib().sahs_stage1_loss_forward(N, _p(mc), _p(mf), _p(target), int(target.shape[1]), _p(mask), _p(cw), _p(stats), _stream()),
          "sahs_stage1_loss_forward")
    return stats


def composite_backward(raw, z, rays, noise, bg, white_background, d_rgb, d_disp, d_acc, d_depth, d_wlast, d_weights=None, loss=None):
    """loss = (map (N,15), target, mask, stats, gscale or None): the level's share of the Stage-I objective's gradient is formed inside
    the kernel (sahs_composite_backward_loss) and added to d_rgb."""
    raw, z, rays = _req(raw, "raw"), _req(z, "z"), _req(rays, "rays")
    N, S = z.shape
    d_raw = torch.empty(N, S, 16, dtype=torch.float32, device=z.device)
    if loss is not None:
        if d_weights is not None:
            raise _lib.SahsError("composite_backward: d_weights and loss together are not supported")
        lm, lt, lk, st, gsc = (_req(t, n) for t, n in zip(loss, ("loss_map", "loss_target", "loss_mask", "loss_stats", "loss_gscale")))
        if tuple(lm.shape) != (N, 15) or tuple(lk.shape) != (N, 12) or lt.shape[0] != N or st.numel() < LOSS_STATS_WORDS:
            raise _lib.SahsError("composite_backward: loss operands (N,15), (N,>=3), (N,12), stats (64,)")
        gs = [_req(g, n) for g, n in ((d_rgb, "d_rgb"), (d_disp, "d_disp"), (d_acc, "d_acc"), (d_depth, "d_depth"), (d_wlast, "d_wlast"))]
        check(_lib.lib().sahs_composite_backward_loss(N, S, _p(raw), _p(z), _p(rays), int(rays.shape[1]), _p(_req(noise, "noise")), _p(_req(bg, "bg")),
                                                       int(bool(white_background)), *[_p(g) for g in gs], _p(lm), _p(lt), int(lt.shape[1]), _p(lk),
                                                       _p(st), _p(gsc), _p(d_raw), _stream()), "sahs_composite_backward_loss")
        return d_raw
    gs = [_req(g, n) for g, n in ((d_rgb, "d_rgb"), (d_disp, "d_disp"), (d_acc, "d_acc"), (d_depth, "d_depth"), (d_wlast, "d_wlast"),
                                  (d_weights, "d_weights"))]
    check(_lib.lib().sahs_composite_backward(N, S, _p(raw), _p(z), _p(rays), int(rays.shape[1]), _p(_req(noise, "noise")), _p(_req(bg, "bg")),
                                              int(bool(white_background)), *[_p(g) for g in gs], _p(d_raw), _stream()), "sahs_composite_backward")
    return d_raw


class CompositeFn(torch.autograd.Function):
    """volume_render_radiance_field (volume_rendering_utils.py:7-78) as a differentiable op (seam B3): gradient w.r.t. the radiance
    field through sahs_composite_backward; depths and directions get none (the reference's callers never use those)."""

    @staticmethod
    def forward(ctx, raw, z, rays, noise, bg, white_background):
        outs = composite_forward(raw, z, rays, noise=noise, bg=bg, white_background=white_background)
        none = torch.empty(0, device=raw.device)
        ctx.save_for_backward(raw.detach(), z, rays, noise if noise is not None else none, bg if bg is not None else none, outs[3][:, -1])
        ctx.cfg = (bool(white_background), noise is not None, bg is not None)
        return outs

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_w, g_depth):
        raw, z, rays, noise, bg, w_last = ctx.saved_tensors
        white, has_noise, has_bg = ctx.cfg
        c = lambda t: None if t is None else t.contiguous().float()
        d_raw = composite_backward(raw, z, rays, noise if has_noise else None, bg if has_bg else None, white, c(g_rgb), c(g_disp), c(g_acc),
                                   c(g_depth), None, c(g_w))
        if has_bg and g_rgb is not None:     # the last sample's 15 channels are used verbatim (:28-35): d = w_last * d_rgb
            d_raw[:, -1, :15] += w_last[:, None] * g_rgb
        return d_raw, None, None, None, None, None


class FieldFn(torch.autograd.Function):
    """The field evaluation of seam B2, model(level, x, driving, pose) -> (P, 16), differentiable w.r.t. the parameters (as the
    canonical flat buffer) and the driving input; the sample points themselves get no gradient (no caller of the reference's asks
    for one).  fp32.  Points are passed as zero-length rays, as in the no-grad path of the seam."""

    BLOCK = 2_000_000     # samples per saved-activation block (sahs_field_backward takes at most 4e6 per call; 19 KB each)

    @staticmethod
    def forward(ctx, flat, driving, pose, rays, z, packed, level, arch):
        frame = fold_conditioning(flat.detach(), driving.detach(), pose, arch=arch)
        # the saved activations are plane-per-layer over the P of ONE forward call, so a large batch is cut into blocks HERE and
        # each block keeps its own buffer (a row slice of one big buffer would hand the backward the wrong planes)
        N = z.shape[0]
        rows = max(1, FieldFn.BLOCK // max(1, z.shape[1]))
        raws, acts = [], []
        for s in range(0, N, rows):
            r, a = field_forward_save(packed, frame, level, rays[s:s + rows].contiguous(), z[s:s + rows].contiguous(), arch)
            raws.append(r)
            acts.append(a)
        ctx.save_for_backward(flat.detach(), driving.detach(), frame, *acts)
        ctx.cfg = (level, arch, rows * z.shape[1])
        return raws[0] if len(raws) == 1 else torch.cat(raws, dim=0)

    @staticmethod
    def backward(ctx, g_raw):
        flat, driving, frame = ctx.saved_tensors[:3]
        acts = ctx.saved_tensors[3:]
        level, arch, block = ctx.cfg
        grad_flat = torch.zeros_like(flat)
        grad_cond = torch.zeros(128, dtype=torch.float32, device=flat.device)
        g = g_raw.contiguous().float().view(-1, 16)
        if sum(a.shape[0] for a in acts) != g.shape[0]:
            raise _lib.SahsError("FieldFn.backward: %d gradient rows for %d saved samples" % (g.shape[0], sum(a.shape[0] for a in acts)))
        for i, act in enumerate(acts):
            field_backward(flat, frame, level, act, g[i * block: i * block + act.shape[0]], grad_flat, grad_cond, arch)
        if arch == "audio":
            grad_drv = torch.zeros_like(driving)
            check(_lib.lib().sahs_conditioning_backward(_p(flat), _p(driving), _p(grad_cond), _p(grad_flat), _p(grad_drv), _stream()),
                  "sahs_conditioning_backward")
        else:
            grad_drv = grad_cond[:76].clone()
        return grad_flat, grad_drv, None, None, None, None, None, None


class RenderRaysFn(torch.autograd.Function):
    """predict_and_render_radiance (train_utils.py:72-206) for one ray chunk, differentiable w.r.t. the model
    parameters (as the canonical flat buffer) and the audio window.

    forward: the six forward launches; chunks of at most BLOCK_RAYS rays (a training batch) keep the field activations of
    both passes (19 KB per sample, 7.5 GB for 2048 rays x 192 samples), larger chunks only keep the depths and re-run the field
    block by block in backward.  backward: per level -- composite backward, field backward -- then the conditioning backward."""

    BLOCK_RAYS = 4096
    SHARE_DEFORMATION = True      # kept path: deformation nets once per depth, forward AND backward (the fine pass's gradient w.r.t. the
                                  # coarse samples' (x', w) is added at the seam of the coarse pass's backward); False = the plain chain

    @staticmethod
    def forward(ctx, flat, audio, pose, rays, bg, t_rand, noise_c, u, noise_f, packed, num_coarse, num_fine, lindisp, white_background,
                arch="audio", loss_target=None, loss_mask=None, loss_weights=None, packed_x3=None):
        """With loss_target (N,>=3), loss_mask (N,12), loss_weights (12,) the op also returns (loss, stats) of the Stage-I objective
        (stage1_loss_forward) and its backward forms that loss's gradient inside the composite backward kernels.
        packed_x3 (pack_weights(flat, SAHS_BF16X3); AudioFaceModel, kept path): the saving forward launches run on the split-operand
        kernels (training_forward_precision "bf16x3") -- the coarse pass as a deformation + a radiance launch into one whole-network save."""
        frame = fold_conditioning(flat.detach(), audio.detach(), pose, arch=arch)
        ctx.arch = arch
        ctx.has_loss = loss_target is not None

        def with_loss(outs, saved):
            if not ctx.has_loss:
                ctx.save_for_backward(*saved)
                return outs
            tgt, msk = loss_target.detach().float().contiguous(), loss_mask.detach().float().contiguous()
            rgb_f = outs[3] if num_fine > 0 else None
            stats = stage1_loss_forward(outs[0], rgb_f, tgt, msk, loss_weights.detach().float().contiguous())
            ctx.n_base = len(saved)
            ctx.save_for_backward(*saved, outs[0], rgb_f if rgb_f is not None else torch.empty(0, device=rays.device), tgt, msk, stats)
            ctx.mark_non_differentiable(stats)
            return tuple(outs) + (stats[0].clone(), stats)

        none = torch.empty(0, device=rays.device)
        ctx.cfg = (num_coarse, num_fine, bool(white_background), bg is not None, noise_c is not None, noise_f is not None)
        ctx.kept = rays.shape[0] <= RenderRaysFn.BLOCK_RAYS and num_fine > 0
        ctx.shared = ctx.kept and RenderRaysFn.SHARE_DEFORMATION and arch != "nerface_static"
        if ctx.shared:   # the launch chain of sahs_model_render_rays_rows' split evaluation, with the activations of every launch kept
            N = rays.shape[0]
            z_c = stratified_depths(rays, num_coarse, lindisp, t_rand)
            xw = torch.empty(N, num_coarse + num_fine, 8, dtype=torch.float32, device=rays.device)
            sb = lambda samples, mode: alloc_sign_bits(samples, mode, arch, rays.device)      # (None: this architecture's backward reads none)
            bits_c, bits_d, bits_r = sb(N * num_coarse, FIELD_ALL), sb(N * num_fine, FIELD_DEFORM), sb(N * (num_coarse + num_fine), FIELD_RADIANCE)
            x3 = packed_x3 is not None and arch == "audio" and bits_c is not None
            pk, prec = (packed_x3, SAHS_BF16X3) if x3 else (packed, SAHS_F32)
            if x3:
                act_c = torch.empty(N * num_coarse, _fn("act_words_part", arch)[0](FIELD_ALL), dtype=torch.float32, device=rays.device)
                field_forward_split_save(pk, frame, 0, FIELD_DEFORM, rays, xw, z=z_c, arch=arch, precision=prec, whole=(act_c, bits_c))
                raw_c, _ = field_forward_split_save(pk, frame, 0, FIELD_RADIANCE, rays, xw, src=_identity_src(N, num_coarse, rays.device), arch=arch,
                                                    precision=prec, whole=(act_c, bits_c))
            else:
                raw_c, act_c = field_forward_split_save(packed, frame, 0, FIELD_ALL, rays, xw, z=z_c, arch=arch, bits=bits_c)
            rgb_c, disp_c, acc_c, w_c, _ = composite_forward(raw_c, z_c, rays, noise_c, bg, white_background)
            z_f, z_new, src = resample_merge(z_c, w_c, num_fine, u=u)
            _, act_d = field_forward_split_save(pk, frame, 1, FIELD_DEFORM, rays, xw, z=z_new, xw_col0=num_coarse, arch=arch, bits=bits_d, precision=prec)
            raw_f, act_r = field_forward_split_save(pk, frame, 1, FIELD_RADIANCE, rays, xw, src=src, arch=arch, bits=bits_r, precision=prec)
            del xw
            rgb_f, disp_f, acc_f, w_f, depth_f = composite_forward(raw_f, z_f, rays, noise_f, bg, white_background)
            outs = (rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f, w_f[:, -1].contiguous(), depth_f)
            ctx.has_bits = bits_c is not None
            return with_loss(outs, (flat.detach(), audio.detach(), rays, z_c, z_f, frame, packed,
                                    *[t if t is not None else none for t in (bg, noise_c, noise_f)], raw_c, act_c, raw_f, act_r, act_d, src,
                                    *((bits_c, bits_r, bits_d) if bits_c is not None else ())))
        if ctx.kept:     # the same launch chain as sahs_render_rays, with the field activations kept
            z_c = stratified_depths(rays, num_coarse, lindisp, t_rand)
            raw_c, act_c = field_forward_save(packed, frame, 0, rays, z_c, arch)
            rgb_c, disp_c, acc_c, w_c, _ = composite_forward(raw_c, z_c, rays, noise_c, bg, white_background)
            z_f = resample(z_c, w_c, num_fine, u)
            raw_f, act_f = field_forward_save(packed, frame, 1, rays, z_f, arch)
            rgb_f, disp_f, acc_f, w_f, depth_f = composite_forward(raw_f, z_f, rays, noise_f, bg, white_background)
            outs = (rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f, w_f[:, -1].contiguous(), depth_f)
            return with_loss(outs, (flat.detach(), audio.detach(), rays, z_c, z_f, frame, packed,
                                    *[t if t is not None else none for t in (bg, noise_c, noise_f)], raw_c, act_c, raw_f, act_f))
        ws = {}
        outs = render_rays(packed, frame, rays, num_coarse, num_fine, precision=SAHS_F32, lindisp=lindisp, white_background=white_background,
                           bg=bg, t_rand=t_rand, noise_c=noise_c, u=u, noise_f=noise_f, workspace=ws, arch=arch)
        return with_loss(outs, (flat.detach(), audio.detach(), rays, ws["z_c"].clone(), ws["z_f"].clone() if num_fine > 0 else none, frame, packed,
                                *[t if t is not None else none for t in (bg, noise_c, noise_f)]))

    @staticmethod
    def backward(ctx, g_rgb_c, g_disp_c, g_acc_c, g_rgb_f, g_disp_f, g_acc_f, g_wbg, g_depth_f, g_loss=None, g_stats=None):
        flat, audio, rays, z_c, z_f, frame, packed, bg, noise_c, noise_f = ctx.saved_tensors[:10]
        loss_ops = None
        if ctx.has_loss and g_loss is not None:
            map_c, map_f, l_tgt, l_msk, l_stats = ctx.saved_tensors[ctx.n_base:ctx.n_base + 5]
            loss_ops = {0: map_c, 1: map_f}
            gscale = g_loss.detach().float().reshape(1).contiguous()
        kept = dict(zip((0, 1), (ctx.saved_tensors[10:12], ctx.saved_tensors[12:14]))) if ctx.kept else None
        act_d, src = (ctx.saved_tensors[14], ctx.saved_tensors[15]) if ctx.shared else (None, None)
        bits_c, bits_r, bits_d = ctx.saved_tensors[16:19] if (ctx.shared and getattr(ctx, "has_bits", False)) else (None, None, None)
        kept_bits = {0: bits_c, 1: bits_r}
        xwg_coarse = None      # shared deformation: the fine pass's seam gradient that belongs to the coarse samples
        nc, nf, white, has_bg, has_nc, has_nf = ctx.cfg
        bg = bg if has_bg else None
        noise_c = noise_c if has_nc else None
        noise_f = noise_f if has_nf else None
        dev = rays.device
        grad_flat = torch.zeros_like(flat)
        grad_cond = torch.zeros(128, dtype=torch.float32, device=dev)
        grad_audio = torch.zeros_like(audio)
        c = lambda t: None if t is None else t.contiguous().float()
        N = rays.shape[0]
        both_levels = loss_ops is not None or (any(g is not None for g in (g_rgb_f, g_disp_f, g_acc_f, g_depth_f, g_wbg)) and
                                               any(g is not None for g in (g_rgb_c, g_disp_c, g_acc_c)))
        # the fused walk is two full-chip persistent launches per part: run side by side they starve each other (measured: 14.0 ms per step on two
        # streams, 13.3 on one), so the pairwise two-stream issue below is for the per-layer walks only (NeRFaceModels, fused_backward(False))
        fused_walk = bits_c is not None and _FUSED_BACKWARD and ctx.arch == "audio"
        if (ctx.shared and nf > 0 and kept is not None and N <= RenderRaysFn.BLOCK_RAYS and both_levels and not fused_walk
                and not os.environ.get("SAHS_BWD_ONE_STREAM")):
            # The two levels' radiance walks are independent of each other, and so are the two deformation walks that follow them (coarse
            # depths / new depths): each pair runs on two streams.  A walk is ~75 dependent GEMM launches whose fixed costs (ring fill, a
            # K loop with one workgroup per CU, the atomic epilogue, ramp and tail: DESIGN.md section 7) leave most of the chip idle for
            # part of every launch; the other stream's launches fill it.  Everything the walks add into grad_flat / grad_cond is atomic.
            main, side = torch.cuda.current_stream(dev), _side_stream(dev)
            gb = lambda grads: [None if g is None else c(g) for g in grads]
            lv_loss = lambda level: None if loss_ops is None else (loss_ops[level].contiguous(), l_tgt.contiguous(), l_msk.contiguous(), l_stats, gscale)
            (raw1, act1), (raw0, act0) = kept[1], kept[0]
            cc = lambda t: None if t is None else t.contiguous()
            d_raw1 = composite_backward(raw1, cc(z_f), cc(rays), cc(noise_f), cc(bg), white, *gb((g_rgb_f, g_disp_f, g_acc_f, g_depth_f, g_wbg)), loss=lv_loss(1))
            d_raw0 = composite_backward(raw0, cc(z_c), cc(rays), cc(noise_c), cc(bg), white, *gb((g_rgb_c, g_disp_c, g_acc_c, None, None)), loss=lv_loss(0))
            grad_cond_side = torch.zeros_like(grad_cond)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                g_c0 = field_backward_split(flat, frame, 0, FIELD_RADIANCE, act0, grad_flat, grad_cond_side, d_raw=d_raw0.view(-1, 16), arch=ctx.arch, full_act=True,
                                            bits=bits_c)
            g_f = field_backward_split(flat, frame, 1, FIELD_RADIANCE, act1, grad_flat, grad_cond, d_raw=d_raw1.view(-1, 16), arch=ctx.arch, bits=bits_r)
            xwg_coarse, g_new = route_xw_grad(src, g_f, nc)
            main.wait_stream(side)
            xwg0 = g_c0 + xwg_coarse          # the seam gradient of the coarse samples: their own radiance walk's + the fine pass's share
            side.wait_stream(main)
            with torch.cuda.stream(side):
                field_backward_split(flat, frame, 0, FIELD_DEFORM, act0, grad_flat, grad_cond_side, xw_grad_in=xwg0, arch=ctx.arch, full_act=True, bits=bits_c)
            field_backward_split(flat, frame, 1, FIELD_DEFORM, act_d, grad_flat, grad_cond, xw_grad_in=g_new, arch=ctx.arch, bits=bits_d)
            main.wait_stream(side)
            grad_cond += grad_cond_side
            for t in (d_raw0, xwg0, grad_cond_side, act0, grad_flat, flat, frame) + ((bits_c,) if bits_c is not None else ()):      # made on this stream, used on the side stream: the
                t.record_stream(side)                                                     # allocator must not hand them out again before it is done
            g_c0.record_stream(main)                                                      # (and the other way round)
            N = 0                             # (the block loop below has nothing left to do)
        for s in range(0, N, RenderRaysFn.BLOCK_RAYS):
            e = min(N, s + RenderRaysFn.BLOCK_RAYS)
            sl = slice(s, e)
            rb = rays[sl].contiguous()
            bgb = None if bg is None else bg[sl].contiguous()
            if nf > 0:
                passes = ((1, z_f, noise_f, (g_rgb_f, g_disp_f, g_acc_f, g_depth_f, g_wbg)),
                          (0, z_c, noise_c, (g_rgb_c, g_disp_c, g_acc_c, None, None)))
            else:     # coarse only (train_utils.py:148-149): depth and weights[:, -1] of the 8-tuple are the COARSE pass's
                passes = ((0, z_c, noise_c, (g_rgb_c, g_disp_c, g_acc_c, g_depth_f, g_wbg)),)
            for level, z, noise, grads in passes:
                lvl_loss = None
                if loss_ops is not None:      # with nf == 0 the only pass is level 0 and its map is the coarse one
                    lvl_loss = (loss_ops[level][sl].contiguous(), l_tgt[sl].contiguous(), l_msk[sl].contiguous(), l_stats, gscale)
                if lvl_loss is None and all(g is None for g in grads) and not (level == 0 and xwg_coarse is not None):
                    continue
                zb = z[sl].contiguous()
                raw, act = kept[level] if kept is not None else field_forward_save(packed, frame, level, rb, zb, ctx.arch)
                nb = None if noise is None else noise[sl].contiguous()
                gb = [None if g is None else c(g[sl]) for g in grads]
                d_raw = composite_backward(raw, zb, rb, nb, bgb, white, *gb, loss=lvl_loss)
                if ctx.shared:      # (kept path: one block, sl covers every ray)
                    if level == 1:
                        g_f = field_backward_split(flat, frame, 1, FIELD_RADIANCE, act, grad_flat, grad_cond, d_raw=d_raw.view(-1, 16), arch=ctx.arch, bits=kept_bits[1])
                        xwg_coarse, g_new = route_xw_grad(src, g_f, nc)
                        field_backward_split(flat, frame, 1, FIELD_DEFORM, act_d, grad_flat, grad_cond, xw_grad_in=g_new, arch=ctx.arch, bits=bits_d)
                        del g_f, g_new
                    else:
                        field_backward_split(flat, frame, 0, 3, act, grad_flat, grad_cond, d_raw=d_raw.view(-1, 16), xw_grad_in=xwg_coarse, arch=ctx.arch,
                                             bits=kept_bits[0])
                else:
                    field_backward(flat, frame, level, act, d_raw.view(-1, 16), grad_flat, grad_cond, ctx.arch)
                del raw, act, d_raw
        if ctx.arch == "audio":
            check(_lib.lib().sahs_conditioning_backward(_p(flat), _p(audio), _p(grad_cond), _p(grad_flat), _p(grad_audio), _stream()),
                  "sahs_conditioning_backward")
        else:       # NeRFaceModel: the driving vector is the expression itself
            grad_audio = grad_cond[:76].clone()
        return (grad_flat, grad_audio) + (None,) * 17
