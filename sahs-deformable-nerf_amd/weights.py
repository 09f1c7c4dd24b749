"""Canonical parameter order and deterministic synthetic weights.

The weight ABI of the hot path is the reference model's ``state_dict``
(``eval_stage_rays.py:299-303`` builds ``AudioFaceModel(cfg)`` and calls
``load_state_dict(checkpoint["model_state_dict"])``).  This module lists those tensors in
``state_dict`` order (verified against the imported reference by
``tests/golden/make_golden.py``) and concatenates them into ONE flat fp32 buffer -- the
"canonical flat buffer" the C-ABI's ``sahs_pack_weights`` consumes.

No checkpoint is available offline, so tests and ``bench.py`` use *hash-filled* weights: every
element is a pure function of (tensor index, element index), scaled like PyTorch's default
initialisers (Linear/Conv1d: U(+-1/sqrt(fan_in)); grid: N(0, 0.01^2), ``models.py:201``).
The same function runs in the golden-vector generator and on the GPU box, so no weights need
to be committed.
"""
from collections import OrderedDict

import numpy as np

# Architecture of AudioFaceModel(config/audio/*.yml).  The trunk's skip layer is 3, not the
# YAML's 4: NeRFaceModel never forwards skip_connect_every (models.py:259-296), so NeRFMLP
# keeps its default (modules.py:176).
D_XYZ, D_DIR, D_AMB, D_POSE, D_DRV, D_GRID, G_RES = 63, 27, 18, 36, 76, 32, 32
WARP_H, HYP_H, DEF_LAYERS, DEF_SKIP = 128, 64, 6, 4
TR_H, TR_LAYERS, TR_SKIP, BR_H, N_SEG = 256, 8, 3, 128, 12
D_DEF_IN = D_XYZ + D_DRV + D_POSE      # 175
D_TR_IN = D_XYZ + D_AMB + D_POSE       # 117
D_DIR_IN = TR_H + D_DIR + D_GRID       # 315

# GEMM work per sample-evaluation as the reference writes it.
MAC_PER_SAMPLE = 927_872
FLOP_PER_SAMPLE = 2 * MAC_PER_SAMPLE


# NeRFaceModel built from config/expression/person_2.yml / person_3.yml (models.py:189-299): 15-octave position encodings
# (93), 1-D ambient coordinate encoded WITHOUT its input (30), 4-layer trunk fed [PE93 | PE30 | expression76]; the deformation
# nets still take the 36-d pose encoding (modules.py:344: dim_pose = include_pose + 36 even when include_pose is False).
NERFACE = dict(D_XYZ=93, D_AMB=30, AMB_DIM=1, TR_LAYERS=4, D_DEF_IN=93 + 76 + 36, D_TR_IN=93 + 30 + 76)
NERFACE_MAC_PER_SAMPLE = (128 * 205 + 4 * 128 * 128 + 128 * (128 + 205) + 3 * 128) + (64 * 205 + 4 * 64 * 64 + 64 * (64 + 205) + 64) \
    + (256 * 199 + 2 * 256 * 256 + 256 * (256 + 199) + 256 * 256 + 256) + (128 * 315 + 3 * 128 * 128 + 3 * 128) \
    + (128 * 256 + 3 * 128 * 128 + 12 * 128)


# NeRFaceModel built from config/expression/person_1.yml (warp.use_warp False, hyper.use_ambient False): no deformation
# nets, 10-octave encoding, trunk fed [PE63 | expression76].
NERFACE_STATIC = dict(D_XYZ=63, D_AMB=0, AMB_DIM=0, TR_LAYERS=4, D_DEF_IN=0, D_TR_IN=63 + 76)
NERFACE_STATIC_MAC_PER_SAMPLE = (256 * 139 + 2 * 256 * 256 + 256 * (256 + 139) + 256 * 256 + 256) + (128 * 315 + 3 * 128 * 128 + 3 * 128) \
    + (128 * 256 + 3 * 128 * 128 + 12 * 128)
MODELS = ("audio", "nerface", "nerface_static")


def canonical_spec(model="audio"):
    """[(state_dict key, shape)] in ``state_dict`` order."""
    if model == "nerface":
        return _spec(NERFACE["D_DEF_IN"], NERFACE["D_TR_IN"], NERFACE["AMB_DIM"], NERFACE["TR_LAYERS"], audionet=False)
    if model == "nerface_static":
        return _spec(0, NERFACE_STATIC["D_TR_IN"], 0, NERFACE_STATIC["TR_LAYERS"], audionet=False, deform=False)
    assert model == "audio", model
    return _spec(D_DEF_IN, D_TR_IN, 2, TR_LAYERS, audionet=True)


def _spec(D_DEF_IN, D_TR_IN, AMB_DIM, TR_LAYERS, audionet, deform=True):
    spec = [("spatial_embeddings", (1, D_GRID, G_RES, G_RES, G_RES))]

    def lin(name, out, inp):
        spec.append((name + ".weight", (out, inp)))
        spec.append((name + ".bias", (out,)))

    for i in range(DEF_LAYERS if deform else 0):
        inp = D_DEF_IN if i == 0 else (WARP_H + D_DEF_IN if i == DEF_SKIP else WARP_H)
        lin(f"warp_field_mlp.layers_xyz.{i}", WARP_H, inp)
    if deform:
        lin("warp_field_mlp.fc_final", 3, WARP_H)
    for i in range(DEF_LAYERS if deform else 0):
        inp = D_DEF_IN if i == 0 else (HYP_H + D_DEF_IN if i == DEF_SKIP else HYP_H)
        lin(f"hyper_sheep_mlp.layers_ambient.{i}", HYP_H, inp)
    if deform:
        lin("hyper_sheep_mlp.fc_ambient", AMB_DIM, HYP_H)
    for lvl in ("coarse", "fine"):
        p = f"nerf_mlps.{lvl}."
        for i in range(TR_LAYERS):
            inp = D_TR_IN if i == 0 else (TR_H + D_TR_IN if i == TR_SKIP else TR_H)
            lin(p + f"layers_xyz.{i}", TR_H, inp)
        lin(p + "fc_feat", TR_H, TR_H)
        lin(p + "fc_alpha", 1, TR_H)
        for i in range(4):
            lin(p + f"layers_dir.{i}", BR_H, D_DIR_IN if i == 0 else BR_H)
        lin(p + "fc_rgb", 3, BR_H)
        for i in range(4):
            lin(p + f"layers_seg.{i}", BR_H, TR_H if i == 0 else BR_H)
        lin(p + "fc_seg", N_SEG, BR_H)
    if not audionet:
        return spec
    for idx, (co, ci) in zip((0, 2, 4, 6), ((32, 29), (32, 32), (64, 32), (64, 64))):
        spec.append((f"audNet_head.encoder_conv.{idx}.weight", (co, ci, 3)))
        spec.append((f"audNet_head.encoder_conv.{idx}.bias", (co,)))
    lin("audNet_head.encoder_fc1.0", 64, 64)
    lin("audNet_head.encoder_fc1.2", D_DRV, 64)
    return spec


def param_count(model="audio"):
    return int(sum(int(np.prod(s)) for _, s in canonical_spec(model)))


def canonical_offsets(model="audio"):
    """OrderedDict key -> (offset, shape) into the flat buffer."""
    out, off = OrderedDict(), 0
    for k, s in canonical_spec(model):
        out[k] = (off, s)
        off += int(np.prod(s))
    return out


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


def hash_uniform(n, stream, seed=0):
    """n floats in [0,1): a pure function of (seed, stream, index); 24-bit mantissa."""
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint64)
        key = _splitmix64(np.uint64(seed) * np.uint64(0x632BE59BD9B4E019) + np.uint64(stream))
        x = _splitmix64(i ^ key)
    return ((x >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def hash_normal(n, stream, seed=0):
    u1 = hash_uniform(n, 2 * stream + 1_000_003, seed).astype(np.float64)
    u2 = hash_uniform(n, 2 * stream + 1_000_004, seed).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (r * np.cos(2.0 * np.pi * u2)).astype(np.float32)


# mean of fc_alpha's output (gain 1, before the bias shift) of the hdr variant over the W512 sample volume: subtracted so that the
# density logit is centred and rays terminate at varying depths in both passes instead of all-transparent / all-opaque
HDR_SIGMA_CENTER = {"coarse": -0.334, "fine": 1.172}


def hash_state_dict(seed=0, density_bias=0.0, density_gain=1.0, model="audio", hdr=False):
    """Deterministic synthetic ``state_dict`` (numpy fp32 arrays, state_dict order).

    ``density_bias``/``density_gain`` give the *density-boosted* variant of SURVEY.md section 8(d):
    fc_alpha.weight is multiplied by ``density_gain`` and ``density_bias`` is added to
    fc_alpha.bias, so that rays terminate before the background sample and the composite /
    importance-sampling stages are actually exercised.

    ``hdr=True`` is the *high-dynamic-range* variant: PyTorch's default initialiser shrinks the activation
    variance by 6 per ReLU layer, so default-scale networks end in colour/seg logits of |x| < 0.1 and any accuracy test
    on them is soft (a trained avatar has O(1)-O(10) activations and sharp features).  Here every hidden layer keeps its
    activation variance (He scaling, U(+-sqrt(6/fan_in))), the colour/seg heads are scaled to logits of sigma ~ 2.5, the
    feature grid to sigma 0.3 (reference: 0.01), and the warp head is damped so the deformation stays a few centimetres.
    """
    sd = OrderedDict()
    spec = canonical_spec(model)
    for t, (k, shape) in enumerate(spec):
        n = int(np.prod(shape))
        if k == "spatial_embeddings":
            v = hash_normal(n, t, seed) * np.float32(0.3 if hdr else 0.01)
        else:
            wshape = shape if k.endswith(".weight") else dict(spec)[k[:-5] + ".weight"]
            fan_in = int(np.prod(wshape[1:]))
            bound = np.float32(1.0 / np.sqrt(fan_in))
            v = (hash_uniform(n, t, seed) * np.float32(2.0) - np.float32(1.0)) * bound
            if hdr and k.endswith(".weight") and not k.startswith("audNet_head"):
                if ".layers_" in k or k.endswith("fc_feat.weight"):
                    v = v * np.float32(np.sqrt(6.0))
                elif k.endswith("fc_rgb.weight") or k.endswith("fc_seg.weight"):
                    v = v * np.float32(4.0)
                elif k.endswith("fc_final.weight"):
                    v = v * np.float32(0.05)
        v = v.astype(np.float32).reshape(shape)
        if k.endswith("fc_alpha.weight"):
            v = v * np.float32(density_gain)
        if k.endswith("fc_alpha.bias"):
            v = v + np.float32(density_bias)
            if hdr and model == "audio":   # centre the density logit (its mean over the W512 volume, measured once for seed 0)
                v = v - np.float32(density_gain * HDR_SIGMA_CENTER["fine" if ".fine." in k else "coarse"])
        sd[k] = v
    return sd


def flatten_state_dict(sd, model="audio"):
    """Concatenate a ``state_dict`` (numpy arrays or torch tensors) into the canonical flat buffer."""
    parts = []
    for k, shape in canonical_spec(model):
        v = sd[k]
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        v = np.asarray(v, dtype=np.float32)
        if tuple(v.shape) != tuple(shape):
            raise ValueError(f"{k}: expected shape {shape}, got {tuple(v.shape)}")
        parts.append(v.reshape(-1))
    return np.concatenate(parts)
