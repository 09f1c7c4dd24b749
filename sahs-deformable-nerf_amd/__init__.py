"""MI355X-native deformable-NeRF volume-rendering hot path (drop-in for the reference's
``nerf.train_utils`` / ``nerf.models`` / ``nerf.volume_rendering_utils`` / ``nerf.nerf_helpers`` seams).

The directory name carries a hyphen; import it with ``importlib.import_module("sahs-deformable-nerf_amd")``
or through the ``sahs_amd`` alias module at the repo root.
"""
from . import weights, cfgnode                      # noqa: F401  (pure python, no GPU needed)
from . import _lib, ops                             # noqa: F401  (binds libsahs_nerf.so lazily, fails loudly)
from . import models, nerf_helpers, train_utils, volume_rendering_utils, distributed, training, evaluation, spade   # noqa: F401
from .cfgnode import CfgNode, default_config        # noqa: F401
from .models import AudioFaceModel, NeRFaceModel    # noqa: F401
from .train_utils import run_one_iter_of_nerf, predict_and_render_radiance, run_network   # noqa: F401
from .volume_rendering_utils import volume_render_radiance_field                        # noqa: F401
from .nerf_helpers import (get_ray_bundle, get_ray_bundle_by_mask, sample_pdf_2, mse2psnr, img2mse, meshgrid_xy, cumprod_exclusive,   # noqa: F401
                           positional_encoding, get_embedding_function, get_minibatches, MaskMSELoss, MaskCrossEntropyLoss)
