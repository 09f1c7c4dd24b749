"""Drop-in for ``nerf/volume_rendering_utils.py``."""
import torch

from . import ops


def volume_render_radiance_field(radiance_field, depth_values, ray_directions, radiance_field_noise_std=0.0,
                                 white_background=False, background_prior=None):
    """volume_rendering_utils.py:7-78 -> (rgb_map, disp_map, acc_map, weights, depth_map).

    As in the reference, the caller has already written ``background_prior`` into the last sample's
    15 colour channels (train_utils.py:135-136); ``background_prior`` here only selects the
    sigmoid+softmax / verbatim-last-sample colour rule (:28-35).  The noise draw happens here (:45-53).
    """
    N, S = depth_values.shape
    noise = None
    if radiance_field_noise_std > 0.0:
        noise = torch.randn(radiance_field[..., -1].shape, dtype=radiance_field.dtype, device=radiance_field.device) * radiance_field_noise_std
    rays = torch.zeros(N, 8, dtype=torch.float32, device=depth_values.device)
    rays[:, 3:6] = ray_directions
    bg = None
    if background_prior is not None:
        if background_prior.shape[1] != 15:
            raise NotImplementedError("background_prior must have 15 channels (rgb3 + seg12)")
        bg = radiance_field[:, -1, :15].contiguous()
    if torch.is_grad_enabled() and radiance_field.requires_grad:      # differentiable w.r.t. radiance_field, as the reference's
        if bg is not None:
            bg = bg.detach()
        return ops.CompositeFn.apply(radiance_field.contiguous().float(), depth_values, rays, noise, bg, bool(white_background))
    return ops.composite_forward(radiance_field, depth_values, rays, noise=noise, bg=bg, white_background=white_background)
