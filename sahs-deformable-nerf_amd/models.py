"""Drop-in for the field model of ``nerf/models.py`` (AudioFaceModel over NeRFaceModel).

Same constructor-from-config, same ``state_dict`` keys and shapes (the weight
ABI: ``eval_stage_rays.py:299-303`` does ``getattr(models, cfg.models.mask.type)(cfg)`` followed by
``load_state_dict``), same call protocol ``model(level, x, audio, pose, pose_c, latent_code=None)``
-> (P, 16) (``train_utils.py:34``, ``models.py:514``).  The parameters live in ordinary
``nn.Linear`` / ``nn.Conv1d`` holders so optimisers and checkpoints work unchanged, but nothing is
evaluated through them: ``forward`` packs them for the HIP field kernel.
"""
import torch
from torch import nn

from . import ops, weights as W


def _holder(out_f, in_f):
    return nn.Linear(in_f, out_f)


class _DeformMLP(nn.Module):
    def __init__(self, list_name, final_name, hidden, out_dim, num_layers, skip, d_in):
        super().__init__()
        layers = nn.ModuleList()
        for i in range(num_layers):
            inp = d_in if i == 0 else (hidden + d_in if i == skip else hidden)
            layers.append(_holder(hidden, inp))
        setattr(self, list_name, layers)
        setattr(self, final_name, _holder(out_dim, hidden))


class _RadianceMLP(nn.Module):
    def __init__(self, num_layers, d_in):
        super().__init__()
        self.layers_xyz = nn.ModuleList()
        for i in range(num_layers):
            inp = d_in if i == 0 else (W.TR_H + d_in if i == W.TR_SKIP else W.TR_H)
            self.layers_xyz.append(_holder(W.TR_H, inp))
        self.fc_feat = _holder(W.TR_H, W.TR_H)
        self.fc_alpha = _holder(1, W.TR_H)
        self.layers_dir = nn.ModuleList([_holder(W.BR_H, W.D_DIR_IN if i == 0 else W.BR_H) for i in range(4)])
        self.fc_rgb = _holder(3, W.BR_H)
        self.layers_seg = nn.ModuleList([_holder(W.BR_H, W.TR_H if i == 0 else W.BR_H) for i in range(4)])
        self.fc_seg = _holder(W.N_SEG, W.BR_H)


class _AudioNet(nn.Module):
    def __init__(self):
        super().__init__()
        act = lambda: nn.LeakyReLU(0.02, True)   # placeholders keep the Sequential indices 0,2,4,6 / 0,2
        self.encoder_conv = nn.Sequential(nn.Conv1d(29, 32, 3, 2, 1), act(), nn.Conv1d(32, 32, 3, 2, 1), act(),
                                          nn.Conv1d(32, 64, 3, 2, 1), act(), nn.Conv1d(64, 64, 3, 2, 1), act())
        self.encoder_fc1 = nn.Sequential(nn.Linear(64, 64), act(), nn.Linear(64, W.D_DRV))


def _radiance_cfg(m):
    return [(m.coarse.hidden_size, 256), (m.coarse.include_input_xyz, True), (m.coarse.num_encoding_fn_dir, 4),
            (m.coarse.include_input_dir, True), (m.coarse.use_viewdirs, True), (m.coarse.use_spatial_embeddings, True)]


def _common_cfg(m):
    return [(m.warp.use_warp, True), (m.warp.num_layers, 6), (m.warp.hidden_size, 128), (m.warp.skip_connect_every, 4),
            (m.hyper.use_ambient, True), (m.hyper.slice_method, "bendy_sheet"), (m.hyper.num_layers, 6), (m.hyper.hidden_size, 64),
            (m.hyper.skip_connect_every, 4), (m.hyper.include_driving, True)] + _radiance_cfg(m)


def _check_cfg(cfg, arch):
    """The kernels are specialised to the two architectures the reference's configs describe: config/audio/*.yml
    (AudioFaceModel) and config/expression/person_2.yml / person_3.yml (NeRFaceModel with warp and hyper sheet)."""
    m = cfg.models
    if arch == "audio":
        want = _common_cfg(m) + [(m.warp.num_encoding_fn_xyz, 10), (m.hyper.ambient_coord_dim, 2), (m.hyper.num_encoding_fn_ambient, 4),
                                 (m.hyper.include_input_ambient, True), (m.coarse.num_layers, 8), (m.coarse.num_encoding_fn_xyz, 10),
                                 (m.coarse.use_pose, True), (m.coarse.include_driving, False)]
    elif arch == "nerface":
        want = _common_cfg(m) + [(m.warp.num_encoding_fn_xyz, 15), (m.hyper.num_encoding_fn_xyz, 15), (m.hyper.ambient_coord_dim, 1),
                                 (m.hyper.num_encoding_fn_ambient, 15), (m.hyper.include_input_ambient, False), (m.coarse.num_layers, 4),
                                 (m.coarse.num_encoding_fn_xyz, 15), (m.coarse.use_pose, False), (m.coarse.include_driving, True)]
    else:   # nerface_static: config/expression/person_1.yml
        want = _radiance_cfg(m) + [(m.warp.use_warp, False), (m.hyper.use_ambient, False), (m.coarse.num_layers, 4),
                                   (m.coarse.num_encoding_fn_xyz, 10), (m.coarse.use_pose, False), (m.coarse.include_driving, True)]
    bad = [(a, b) for a, b in want if a != b]
    if bad or not hasattr(m, "fine"):
        raise NotImplementedError("this build covers AudioFaceModel (config/audio/*.yml) and NeRFaceModel as configured by "
                                  "config/expression/person_1.yml (no warp, no hyper sheet) or person_2.yml / person_3.yml (both on). "
                                  "Mismatches (got, want): %r" % (bad,))


class _FieldModel(nn.Module):
    """Shared plumbing: parameters in the reference's state_dict layout, packed lazily for the HIP field kernel."""
    arch = "audio"

    def _build(self, cfg, precision):
        _check_cfg(cfg, self.arch)
        self.num_coarse = cfg.nerf.train.num_coarse
        self.num_fine = cfg.nerf.train.num_fine
        self.precision = ops.PRECISIONS[precision]
        nf = {"nerface": W.NERFACE, "nerface_static": W.NERFACE_STATIC}.get(self.arch)
        d_def, d_tr, amb, trl = ((W.D_DEF_IN, W.D_TR_IN, 2, W.TR_LAYERS) if self.arch == "audio"
                                 else (nf["D_DEF_IN"], nf["D_TR_IN"], nf["AMB_DIM"], nf["TR_LAYERS"]))
        self.spatial_embeddings = nn.Parameter(torch.randn(1, W.D_GRID, W.G_RES, W.G_RES, W.G_RES) * 0.01)   # models.py:199-201
        if self.arch != "nerface_static":       # models.py:231-254: the modules exist only when use_warp / use_ambient
            self.warp_field_mlp = _DeformMLP("layers_xyz", "fc_final", W.WARP_H, 3, W.DEF_LAYERS, W.DEF_SKIP, d_def)
            self.hyper_sheep_mlp = _DeformMLP("layers_ambient", "fc_ambient", W.HYP_H, amb, W.DEF_LAYERS, W.DEF_SKIP, d_def)
        self.nerf_mlps = nn.ModuleDict({"coarse": _RadianceMLP(trl, d_tr), "fine": _RadianceMLP(trl, d_tr)})
        if self.arch == "audio":
            self.audNet_head = _AudioNet()
        keys = [k for k, _ in W.canonical_spec(self.arch)]
        assert list(self.state_dict().keys()) == keys, "state_dict layout drifted from the reference's"
        self._cache = {}

    # ---- weight plumbing ----
    def flat_params(self, differentiable=False):
        """Canonical flat buffer (state_dict order).  differentiable=True keeps the autograd link to the parameters, so a
        gradient w.r.t. the flat buffer (RenderRaysFn.backward) is scattered back to every nn.Parameter by torch.cat's backward."""
        sd = dict(self.named_parameters())
        flat = torch.cat([sd[k].reshape(-1) for k, _ in W.canonical_spec(self.arch)]).float().contiguous()
        return flat if differentiable else flat.detach()

    def load_flat(self, flat):
        off = 0
        with torch.no_grad():
            sd = dict(self.named_parameters())
            for k, shape in W.canonical_spec(self.arch):
                n = sd[k].numel()
                sd[k].copy_(torch.as_tensor(flat[off:off + n]).reshape(shape))
                off += n
        return self

    def packed(self, precision=None):
        """Packed weight stream for the HIP field kernel; re-packed when any parameter changed."""
        precision = self.precision if precision is None else precision
        params = list(self.parameters())
        key = (precision, params[0].device, tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
        hit = self._cache.get(("packed", precision))      # (one entry per precision: a training step on the split-operand forward holds two)
        if hit is None or hit[0] != key:
            flat = self.flat_params()
            hit = (key, ops.pack_weights(flat, precision, arch=self.arch), flat)
            self._cache["packed", precision] = hit
        return hit[1], hit[2]

    def frame(self, driving, pose):
        """Per-frame conditioning buffer (driving vector + pose encoding + folded biases): models.py:367-370 / 517-521."""
        _, flat = self.packed()
        return ops.fold_conditioning(flat, driving.to(torch.float32), pose.to(torch.float32), arch=self.arch)

    # ---- B2 seam ----
    def forward(self, level, x, driving=None, pose=None, pose_c=None, latent_code=None, **kwargs):
        """models.py:366-378 / 514-528: x (P, >=6) rows [xyz, raw ray direction, ...] -> (P, 16) [rgb3, seg12, sigma]."""
        if latent_code is not None:
            raise NotImplementedError("latent codes are not used by the shipped configs (latent_code_dim=0)")
        driving = kwargs.get("audio", driving)
        packed, _ = self.packed()
        P = x.shape[0]
        # a point is a zero-length ray: ro = xyz, z = 0  =>  ro + rd*0 == xyz exactly
        rays = torch.zeros(P, 8, dtype=torch.float32, device=x.device)
        rays[:, :6] = x[:, :6].detach()
        z = torch.zeros(P, 1, dtype=torch.float32, device=x.device)
        lvl = 0 if level == "coarse" else 1
        if torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or driving.requires_grad):
            if self.precision != ops.SAHS_F32:
                raise NotImplementedError("gradients run through the fp32 path; build the model with precision='fp32'")
            raw = ops.FieldFn.apply(self.flat_params(differentiable=True), driving.to(torch.float32), pose.to(torch.float32), rays, z, packed,
                                    lvl, self.arch)
        elif ops.is_mixed(self.arch, self.precision):     # split-chain precisions: split-operand deformation launch, low-precision radiance launch
            xw = torch.empty(P, 1, 8, dtype=torch.float32, device=x.device)
            raw = ops.field_forward_split(packed, self.frame(driving, pose), lvl, ops.FIELD_ALL, rays, xw, z=z, arch=self.arch, precision=self.precision)
        else:
            raw = ops.field_forward(packed, self.frame(driving, pose), lvl, rays, z, precision=self.precision, arch=self.arch)
        return raw.view(P, 16)


class AudioFaceModel(_FieldModel):
    """models.py:381-528: audio-driven (driving = AudioNet(16x29 DeepSpeech window))."""
    arch = "audio"

    def __init__(self, cfg, precision="fp32"):
        super().__init__()
        self._build(cfg, precision)


class NeRFaceModel(_FieldModel):
    """models.py:189-378: expression-driven (driving = the 76-d expression vector).  Two architectures, chosen by the config as
    the reference does (models.py:231,244): warp + hyper sheet on (config/expression/person_2.yml, person_3.yml) or both off
    (person_1.yml).  fp32 rendering and training; precision="bf16" (the deforming architecture only) is MIXED precision: a
    bf16-level error in the warp output would be 16 rad at the 15th octave, so the deformation nets run with every operand split into
    bf16 hi + lo (three MFMAs per product: x' within 9e-7 of the fp32 kernel's; round 2 ran them on the fp32 kernel, SAHS_X3_DEFORM=f32
    still does) while the radiance nets run on plain bf16 operands; rendering through run_one_iter_of_nerf only
    (it needs the per-chunk workspace)."""

    def __init__(self, cfg, precision="fp32"):
        super().__init__()
        deform = (bool(cfg.models.warp.use_warp), bool(cfg.models.hyper.use_ambient))
        if deform[0] != deform[1]:
            raise NotImplementedError("NeRFaceModel: warp and hyper sheet are built both on or both off (as in the shipped configs)")
        if precision not in ("fp32", "f32", "bf16"):
            raise NotImplementedError("NeRFaceModel: fp32, or bf16 (with deformation nets: mixed precision, those stay fp32)")
        self.arch = "nerface" if deform[0] else "nerface_static"
        self._build(cfg, precision)
