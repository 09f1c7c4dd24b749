"""Building blocks of the Stage-II SPADE refiner: drop-ins for ``SPADELayer`` and ``SPADEBlock`` of the
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


class SPADELayer(nn.Module):
    """_init_spade.py:114-139: out = InstanceNorm(x) * (1 + gamma(F_id)) + beta(F_id), F_id resized to x by nearest neighbour."""

    def __init__(self, norm_nc, label_nc):
        super().__init__()
        self.param_free_norm = nn.InstanceNorm2d(norm_nc, affine=False)      # no parameters; kept for the module tree (eps 1e-5)
        self.mlp_shared = nn.Sequential(nn.Conv2d(label_nc, 128, kernel_size=3, padding=1), nn.ReLU(inplace=False))
        self.conv_gamma = nn.Conv2d(128, norm_nc, kernel_size=3, padding=1)
        self.conv_beta = nn.Conv2d(128, norm_nc, kernel_size=3, padding=1)

    def forward(self, x, F_id, _slope=1.0):
        """_slope (not in the reference): the LeakyReLU slope of the SPADEBlock that follows, fused into the modulate kernel."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("the fused SPADE modulation is inference-only: call under torch.no_grad()")
        F_id = F.interpolate(F_id, size=x.size()[2:], mode="nearest")
        actv = self.mlp_shared(F_id)
        return ops.spade_modulate(x, self.conv_gamma(actv), self.conv_beta(actv), eps=self.param_free_norm.eps, slope=_slope)


class SPADEBlock(nn.Module):
    """_init_spade.py:235-282."""

    def __init__(self, in_channels, out_channels, fid_channels, downsample=False, upsample=False):
        super().__init__()
        self.spade1 = SPADELayer(in_channels, fid_channels)
        self.lrelu1 = nn.LeakyReLU(0.2)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)
        self.conv1_sn = spectral_norm(self.conv1)            # the same module under a second name, as in the reference
        self.spade2 = SPADELayer(out_channels, fid_channels)
        self.lrelu2 = nn.LeakyReLU(0.2)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1)
        self.conv2_sn = spectral_norm(self.conv2)
        self.downsample = downsample
        if downsample:
            self.downsampler = nn.AvgPool2d(2, stride=2)
            self.residual_downsample = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=1)
        self.upsample = upsample
        if upsample:
            self.upsampler = nn.Upsample(scale_factor=2, mode="nearest")
            self.residual_upsample = nn.ConvTranspose2d(in_channels, in_channels, kernel_size=3, stride=2, padding=1, output_padding=1)
        self.spade_s = SPADELayer(in_channels, fid_channels)
        self.lrelu_s = nn.LeakyReLU(0.2)
        self.conv_s = spectral_norm(nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1))

    def forward(self, x, fid):
        identity = x
        x1 = self.conv1_sn(self.spade1(x, fid, _slope=self.lrelu1.negative_slope))
        if self.downsample:
            x1 = self.downsampler(x1)
            identity = self.residual_downsample(identity)
        if self.upsample:
            x1 = self.upsampler(x1)
            identity = self.residual_upsample(identity)
        x2 = self.conv2_sn(self.spade2(x1, fid, _slope=self.lrelu2.negative_slope))
        x_ = self.conv_s(self.spade_s(identity, fid, _slope=self.lrelu_s.negative_slope))
        return x_ + x2


def _conv_bn_relu(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=False))


class ResBlock2d(nn.Module):
    """_init_spade.py:7-37 (the identity encoder's residual block; library convolutions and batch norm throughout)."""

    def __init__(self, in_channels, out_channels, downsample=False):
        super().__init__()
        self.downsample = downsample
        self.initial = _conv_bn_relu(in_channels, out_channels)
        if downsample:
            self.downsample_layer = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=2, padding=1)
            self.residual_downsample = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=2, padding=1)
        self.residual = _conv_bn_relu(out_channels, out_channels)

    def forward(self, x):
        out = self.initial(x)
        if self.downsample:
            return self.residual_downsample(out) + self.downsample_layer(x)
        return self.residual(out) + x


class IdEncoder(nn.Module):
    """_init_spade.py:185-204: the three identity feature maps (64, 128, 256 channels at 1/2, 1/4, 1/8 resolution) that modulate the refiner."""

    def __init__(self):
        super().__init__()
        self.layer1 = nn.Sequential(nn.Conv2d(3, 64, kernel_size=3, padding=1), nn.AvgPool2d(2, stride=2))
        self.layer2 = ResBlock2d(64, 64)
        self.layer3 = ResBlock2d(64, 128, downsample=True)
        self.layer4 = ResBlock2d(128, 256, downsample=True)

    def forward(self, x):
        x1 = self.layer2(self.layer1(x))
        x2 = self.layer3(x1)
        return x1, x2, self.layer4(x2)


class RefineNetwork(nn.Module):
    """_init_spade.py:284-312: conv + pool, six SPADE blocks (down, down, -, up, up, up), conv."""

    def __init__(self, fid_channels1, fid_channels2, fid_channels3):
        super().__init__()
        self.layer1 = nn.Sequential(nn.Conv2d(3, 64, kernel_size=3, padding=1), nn.AvgPool2d(2, stride=2))
        self.layer2 = SPADEBlock(64, 64, fid_channels1, downsample=True)
        self.layer3 = SPADEBlock(64, 128, fid_channels2, downsample=True)
        self.layer4 = SPADEBlock(128, 256, fid_channels3)
        self.layer5 = SPADEBlock(256, 256, fid_channels3, upsample=True)
        self.layer6 = SPADEBlock(256, 128, fid_channels2, upsample=True)
        self.layer7 = SPADEBlock(128, 64, fid_channels1, upsample=True)
        self.layer8 = nn.Conv2d(64, 3, kernel_size=3, padding=1)

    def forward(self, x, fid1, fid2, fid3):
        x = self.layer1(x)
        for block, fid in ((self.layer2, fid1), (self.layer3, fid2), (self.layer4, fid3), (self.layer5, fid3), (self.layer6, fid2), (self.layer7, fid1)):
            x = block(x, fid)
        return self.layer8(x)


class Generator(nn.Module):
    """_init_spade.py:315-325: the Stage-II generator G(I_src, I_raw) of eval_get_texture_photo_audio.py:154 -- identity features of the
    source image modulate the refinement of the Stage-I render.  Inference only (see SPADELayer)."""

    def __init__(self):
        super().__init__()
        self.idencoder = IdEncoder()
        self.refine_network = RefineNetwork(64, 128, 256)

    def forward(self, I_src, I_raw):
        fid1, fid2, fid3 = self.idencoder(I_src)
        return self.refine_network(I_raw, fid1, fid2, fid3)
