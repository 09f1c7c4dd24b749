"""Drop-in counterparts of the ``nerf/nerf_helpers.py`` functions that sit on the hot path.

Same names, argument meaning and return shapes as the reference (file:line cited per
function); the arithmetic runs in the HIP library.  Loss/metric helpers used by the training
caller (config[4]) are plain torch host code: they touch (2048, 12) tensors once per step.
"""
import math

import torch

from . import ops


def get_minibatches(inputs, chunksize=1024 * 8):
    """nerf_helpers.py:76-81."""
    return [inputs[i:i + chunksize] for i in range(0, inputs.shape[0], chunksize)]


def get_ray_bundle(height, width, intrinsics, tform_cam2world, center=(0.5, 0.5)):
    """nerf_helpers.py:178-233 -> (ray_origins, ray_directions), each (height, width, 3).

    ``intrinsics`` = [fx, fy, cx, cy] with cx, cy relative to the image size; a scalar / 1-element
    focal length means [f, f, 0.5, 0.5] as in the reference (:219-220).
    """
    vals = [float(v) for v in torch.as_tensor(intrinsics).reshape(-1).tolist()]
    if len(vals) < 4:
        vals = [vals[0], vals[0], 0.5, 0.5]
    return ops.get_ray_bundle(int(height), int(width), vals[:4], tform_cam2world)


def sample_pdf_2(bins, weights, num_samples, det=False):
    """nerf_helpers.py:454-497.  The uniform draw happens here, as in the reference (:472-476)."""
    u = None
    if not det:
        u = torch.rand(list(weights.shape[:-1]) + [num_samples], dtype=weights.dtype, device=weights.device)
    return ops.sample_pdf(bins, weights, num_samples, u=u)


def img2mse(img_src, img_tgt):
    """nerf_helpers.py:65-66."""
    return torch.nn.functional.mse_loss(img_src, img_tgt)


def meshgrid_xy(tensor1, tensor2):
    """nerf_helpers.py:84-96: np.meshgrid(..., indexing="xy")."""
    ii, jj = torch.meshgrid(tensor1, tensor2, indexing="ij")
    return ii.transpose(-1, -2), jj.transpose(-1, -2)


def cumprod_exclusive(tensor):
    """nerf_helpers.py:99-120: exclusive cumulative product along the last dimension (host utility; the renderer's
    transmittance product runs inside composite_forward_kernel)."""
    c = torch.cumprod(tensor, -1)
    return torch.cat((torch.ones_like(c[..., :1]), c[..., :-1]), dim=-1)


def positional_encoding(tensor, num_encoding_functions=6, include_input=True, log_sampling=True):
    """nerf_helpers.py:305-349 as a host utility for callers that want the encoding itself (the field kernels evaluate it in
    registers and never materialise it)."""
    enc = [tensor] if include_input else []
    if log_sampling:
        bands = 2.0 ** torch.linspace(0.0, num_encoding_functions - 1, num_encoding_functions, dtype=tensor.dtype, device=tensor.device)
    else:
        bands = torch.linspace(2.0 ** 0.0, 2.0 ** (num_encoding_functions - 1), num_encoding_functions, dtype=tensor.dtype, device=tensor.device)
    for f in bands:
        enc += [torch.sin(tensor * f), torch.cos(tensor * f)]
    return enc[0] if len(enc) == 1 else torch.cat(enc, dim=-1)


def get_embedding_function(num_encoding_functions=6, include_input=True, log_sampling=True):
    """nerf_helpers.py:352-359."""
    return lambda x: positional_encoding(x, num_encoding_functions, include_input, log_sampling)


def get_ray_bundle_by_mask(height, width, intrinsics, tform_cam2world, mask, center=(0.5, 0.5)):
    """nerf_helpers.py:122-175 (square images, as there): pixels inside `mask` get the posed rays, the rest camera-frame
    directions from the origin.  The posed rays come from the HIP ray kernel."""
    intr = [float(v) for v in (intrinsics if len(intrinsics) >= 4 else [intrinsics[0], intrinsics[0], 0.5, 0.5])]
    eye = torch.eye(3, 4, dtype=torch.float32, device=tform_cam2world.device)
    ro, rd = ops.get_ray_bundle(height, width, intr, tform_cam2world.to(torch.float32).contiguous())
    _, rd_cam = ops.get_ray_bundle(height, width, intr, eye)
    m = mask[..., None].to(rd.dtype)
    return m * ro, (1 - m) * rd_cam + m * rd


def mse2psnr(mse):
    """nerf_helpers.py:69-73."""
    if mse == 0:
        mse = 1e-5
    return -10.0 * math.log10(mse)


def _masked_mean(per_pixel, mask):
    mask = mask.reshape((-1, mask.shape[-1]))
    count = torch.count_nonzero(mask, dim=0)
    count = torch.where(count == 0, torch.ones_like(count), count)
    return torch.sum(per_pixel * mask, dim=0) / count


class MaskMSELoss(torch.nn.Module):
    """nerf_helpers.py:40-62: (global mean, per-class masked mean, weighted per-class) of sum_c (a-b)^2."""

    def __init__(self, weights=None):
        super().__init__()
        self.weights = weights

    def forward(self, mask, input, target):
        diff = torch.sum(torch.square(input.reshape((-1, 3)) - target.reshape((-1, 3))), dim=-1, keepdim=True)
        per_class = _masked_mean(diff, mask)
        if self.weights is None:
            self.weights = torch.ones((mask.shape[-1]), device=mask.device)
        return torch.mean(diff), per_class, self.weights * per_class


class MaskCrossEntropyLoss(torch.nn.Module):
    """nerf_helpers.py:14-37: cross entropy -sum_c t*log(p + 1e-10), masked per class."""

    def __init__(self, weights=None):
        super().__init__()
        self.weights = weights

    def forward(self, mask, input, target):
        n = input.shape[-1]
        ce = -torch.sum(target.reshape((-1, n)) * torch.log(input.reshape((-1, n)) + 1e-10), dim=-1, keepdim=True)
        per_class = _masked_mean(ce, mask)
        if self.weights is None:
            self.weights = torch.ones((mask.shape[-1]), device=mask.device)
        return torch.mean(ce), per_class, self.weights * per_class
