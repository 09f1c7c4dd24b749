"""The Stage-II SPADE refiner: drop-ins for ``SPADELayer``, ``SPADEBlock``, the generators built from them, of the
reference's ``nerf/_init_spade.py`` (:114-160, :235-282) with the reference's ``state_dict`` layout (including the duplicated
``conv1`` / ``conv1_sn`` entries: the reference registers each spectral-normalised convolution under two names), so its checkpoints load.

What runs where: the 3x3 convolutions are library work -- ``torch.nn.functional.conv2d`` is MIOpen on ROCm -- and
a hand-written convolution would buy nothing here; the normalise / modulate / activate chain between them, which the reference runs as
six elementwise passes, is ONE fused HIP kernel behind a statistics pass (``ops.spade_modulate``).  Inference only: the refiner's
training (GAN losses, discriminator, VGG features) is outside the hot-path scope.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import spectral_norm

from . import ops


def _conv3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride, 1)


class TiledCodeMap:
    """The modulation map Generator_audio builds from the audio code (_init_spade.py:366-370): the 64-vector repeated to a (1, channels, height,
    64 * tiles) tensor -- every channel and every row the same, the code tiled along the width; the reference materialises it (268 MB for
    its 256 x 64 x 4096) and lets every SPADELayer resize it by nearest neighbour.  Here it stays the 64 numbers: ``nearest(size)`` returns
    what ``F.interpolate(materialised, size, mode="nearest")`` returns, computed from ATen's own index rule."""

    def __init__(self, code, channels, height, tiles):
        self.code, self.channels, self.height, self.tiles = code.reshape(-1), int(channels), int(height), int(tiles)

    @staticmethod
    def _nearest_index(out_size, in_size):
        """ATen upsample_nearest (non-exact mode): min(floor(dst * float32(in / out)), in - 1), evaluated in float32 as ATen does."""
        import numpy as np
        scale = np.float32(in_size) / np.float32(out_size)
        idx = np.floor(np.arange(out_size, dtype=np.float32) * scale).astype(np.int64)
        return np.minimum(idx, in_size - 1)

    def nearest(self, size):
        h, w = int(size[0]), int(size[1])
        cols = torch.from_numpy(self._nearest_index(w, self.code.numel() * self.tiles) % self.code.numel()).to(self.code.device)
        return self.code[cols].reshape(1, 1, 1, w).expand(1, self.channels, h, w).contiguous()

    def materialise(self):
        """The reference's tensor (tests compare ``nearest`` against interpolating this)."""
        return self.code.reshape(1, 1, 1, -1).repeat(1, self.channels, self.height, self.tiles)


class SPADELayer(nn.Module):
    """_init_spade.py:114-139: out = InstanceNorm(x) * (1 + gamma(F_id)) + beta(F_id), F_id resized to x by nearest neighbour.
    The parameter-free norm and the activation of the reference's module tree hold no state and are not modules here: they are the
    fused kernel's ``eps`` and ``slope``."""
    HIDDEN = 128
    EPS = 1e-5          # nn.InstanceNorm2d's default, which the reference uses

    def __init__(self, norm_nc, label_nc):
        super().__init__()
        self.mlp_shared = nn.Sequential(_conv3(label_nc, self.HIDDEN), nn.ReLU())
        self.conv_gamma, self.conv_beta = _conv3(self.HIDDEN, norm_nc), _conv3(self.HIDDEN, norm_nc)

    def forward(self, x, F_id, _slope=1.0):
        """_slope (not in the reference): the LeakyReLU slope of the SPADEBlock that follows, fused into the modulate kernel."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("the fused SPADE modulation is inference-only: call under torch.no_grad()")
        F_id = F_id.nearest(x.shape[-2:]) if isinstance(F_id, TiledCodeMap) else F.interpolate(F_id, size=x.shape[-2:], mode="nearest")
        hidden = self.mlp_shared(F_id)
        return ops.spade_modulate(x, self.conv_gamma(hidden), self.conv_beta(hidden), eps=self.EPS, slope=_slope)


class SPADEBlock(nn.Module):
    """_init_spade.py:235-282: two SPADE -> LeakyReLU -> spectral-normalised conv stages plus a SPADE shortcut, optionally halving
    (average pool on the main branch, strided conv on the shortcut) or doubling (nearest upsample / transposed conv) the resolution
    between the stages.  Registration order = the reference's: it fixes both the state_dict key order and the order in which the
    initialisers draw from the global RNG (tests/test_spade.py regenerates the reference's seeded weights)."""
    SLOPE = 0.2

    def __init__(self, in_channels, out_channels, fid_channels, downsample=False, upsample=False):
        super().__init__()
        for tag, cin in (("1", in_channels), ("2", out_channels)):
            self.add_module("spade" + tag, SPADELayer(cin, fid_channels))
            conv = _conv3(cin, out_channels)
            self.add_module("conv" + tag, conv)
            self.add_module("conv%s_sn" % tag, spectral_norm(conv))      # the same module under a second name, as in the reference
        self.downsample, self.upsample = bool(downsample), bool(upsample)
        if self.downsample:
            self.residual_downsample = _conv3(in_channels, in_channels, stride=2)
        if self.upsample:
            self.residual_upsample = nn.ConvTranspose2d(in_channels, in_channels, 3, 2, 1, output_padding=1)
        self.spade_s = SPADELayer(in_channels, fid_channels)
        self.conv_s = spectral_norm(_conv3(in_channels, out_channels))

    def forward(self, x, fid):
        main = self.conv1_sn(self.spade1(x, fid, _slope=self.SLOPE))
        skip = x
        if self.downsample:
            main, skip = F.avg_pool2d(main, 2, 2), self.residual_downsample(skip)
        if self.upsample:
            main, skip = F.interpolate(main, scale_factor=2, mode="nearest"), self.residual_upsample(skip)
        main = self.conv2_sn(self.spade2(main, fid, _slope=self.SLOPE))
        return self.conv_s(self.spade_s(skip, fid, _slope=self.SLOPE)) + main


def _conv_bn_relu(cin, cout):
    return nn.Sequential(_conv3(cin, cout), nn.BatchNorm2d(cout), nn.ReLU())


def _stem():
    """conv 3 -> 64 at full resolution, then 2x average pool: the first layer of both the identity encoder and the refiner."""
    return nn.Sequential(_conv3(3, 64), nn.AvgPool2d(2, 2))


class ResBlock2d(nn.Module):
    """_init_spade.py:7-37 (the identity encoder's residual block; library convolutions and batch norm throughout).  As in the
    reference a non-downsampling block needs in_channels == out_channels, and a downsampling one does not use ``residual``."""

    def __init__(self, in_channels, out_channels, downsample=False):
        super().__init__()
        self.downsample = bool(downsample)
        self.initial = _conv_bn_relu(in_channels, out_channels)
        if self.downsample:
            self.downsample_layer = _conv3(in_channels, out_channels, stride=2)
            self.residual_downsample = _conv3(out_channels, out_channels, stride=2)
        self.residual = _conv_bn_relu(out_channels, out_channels)

    def forward(self, x):
        y = self.initial(x)
        return self.residual_downsample(y) + self.downsample_layer(x) if self.downsample else self.residual(y) + x


class IdEncoder(nn.Module):
    """_init_spade.py:185-204: the three identity feature maps (64, 128, 256 channels at 1/2, 1/4, 1/8 resolution) that modulate the refiner."""
    WIDTHS = (64, 128, 256)

    def __init__(self):
        super().__init__()
        self.layer1 = _stem()
        cin = self.WIDTHS[0]
        for i, cout in enumerate(self.WIDTHS):
            self.add_module("layer%d" % (i + 2), ResBlock2d(cin, cout, downsample=i > 0))
            cin = cout

    def forward(self, x):
        x, maps = self.layer1(x), []
        for i in range(len(self.WIDTHS)):
            x = getattr(self, "layer%d" % (i + 2))(x)
            maps.append(x)
        return tuple(maps)


class RefineNetwork(nn.Module):
    """_init_spade.py:284-312: stem, six SPADE blocks (down, down, same, up, up, up; block k is modulated by identity map PLAN[k][2]),
    conv to RGB."""
    #        (in, out, identity map, resample)
    PLAN = ((64, 64, 0, "down"), (64, 128, 1, "down"), (128, 256, 2, None), (256, 256, 2, "up"), (256, 128, 1, "up"), (128, 64, 0, "up"))

    def __init__(self, fid_channels1, fid_channels2, fid_channels3):
        super().__init__()
        fid_channels = (fid_channels1, fid_channels2, fid_channels3)
        self.layer1 = _stem()
        for k, (cin, cout, which, how) in enumerate(self.PLAN):
            self.add_module("layer%d" % (k + 2), SPADEBlock(cin, cout, fid_channels[which], downsample=how == "down", upsample=how == "up"))
        self.layer8 = _conv3(64, 3)

    def forward(self, x, fid1, fid2, fid3):
        fids = (fid1, fid2, fid3)
        x = self.layer1(x)
        for k, plan in enumerate(self.PLAN):
            x = getattr(self, "layer%d" % (k + 2))(x, fids[plan[2]])
        return self.layer8(x)


class Generator(nn.Module):
    """_init_spade.py:315-325: the Stage-II generator G(I_src, I_raw) of eval_get_texture_photo_audio.py:154 -- identity features of the
    source image modulate the refinement of the Stage-I render.  Inference only (see SPADELayer)."""

    def __init__(self):
        super().__init__()
        self.idencoder = IdEncoder()
        self.refine_network = RefineNetwork(*IdEncoder.WIDTHS)

    def forward(self, I_src, I_raw):
        return self.refine_network(I_raw, *self.idencoder(I_src))


class AudioNet(nn.Module):
    """_init_spade.py:327-357, the Stage-II file's own audio encoder: four stride-2 Conv1d over the 16-frame window of 29 DeepSpeech features,
    LeakyReLU(0.02) after each; its ``encoder_fc1`` exists (it is in the checkpoints) but the forward returns the 64 convolution features."""
    CHANNELS = (29, 32, 32, 64, 64)

    def __init__(self, dim_aud=76, win_size=16):
        super().__init__()
        self.win_size, self.dim_aud = win_size, dim_aud
        convs = []
        for cin, cout in zip(self.CHANNELS[:-1], self.CHANNELS[1:]):
            convs += [nn.Conv1d(cin, cout, 3, 2, 1), nn.LeakyReLU(0.02)]
        self.encoder_conv = nn.Sequential(*convs)
        self.encoder_fc1 = nn.Sequential(nn.Linear(64, 64), nn.LeakyReLU(0.02), nn.Linear(64, dim_aud))

    def forward(self, x):
        half = self.win_size // 2
        return self.encoder_conv(x[:, 8 - half:8 + half, :].transpose(1, 2)).squeeze(-1)


class Generator_audio(nn.Module):
    """_init_spade.py:359-372: G(I_src, I_raw, audio window) of eval_get_texture_photo_audio.py:156 -- the deepest identity map is replaced by
    the audio code tiled over a 256 x 64 x 4096 map (TiledCodeMap: never materialised here).  Inference only (see SPADELayer)."""

    def __init__(self):
        super().__init__()
        self.idencoder = IdEncoder()
        self.refine_network = RefineNetwork(*IdEncoder.WIDTHS)
        self.AudioNet = AudioNet(76, 16)

    def forward(self, I_src, I_raw, driving_data):
        fid1, fid2, _ = self.idencoder(I_src)
        code = self.AudioNet(driving_data.unsqueeze(0))                       # (1, 64)
        return self.refine_network(I_raw, fid1, fid2, TiledCodeMap(code, IdEncoder.WIDTHS[2], 64, 64))
