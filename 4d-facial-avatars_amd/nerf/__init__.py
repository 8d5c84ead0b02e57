"""`nerf` -- MI355X-native drop-in for the hot path of gafniguy/4D-Facial-Avatars' `nerf` package.

Same public names as the reference's nerf/__init__.py (star-exports of cfgnode, load_*, models,
nerf_helpers, train_utils, volume_rendering_utils), so that train_transformed_rays.py and
eval_transformed_rays.py import it unchanged.  All numerics run in libnerface_hip.so (hand-written
gfx950 HIP kernels); there is no CPU or stock-PyTorch fallback.
"""
from .cfgnode import CfgNode
from .load_blender import load_blender_data
from .load_flame import load_flame_data
from .load_llff import load_llff_data
from . import models
from . import optim            # MI355X extension: nerf.optim.Adam = torch.optim.Adam's update for all tensors of a step in one launch
from .models import *  # noqa: F401,F403
from .nerf_helpers import *  # noqa: F401,F403
from .nerf_helpers import (choose_rays, cumprod_exclusive, dump_rays, get_embedding_function, get_minibatches, get_ray_batch, get_ray_bundle, img2mse,
                           meshgrid_xy, mse2psnr, ndc_rays, positional_encoding, sample_pdf, sample_pdf_2)
from .train_utils import *  # noqa: F401,F403
from .train_utils import GaussianSmoothing, predict_and_render_radiance, run_network, run_one_iter_of_nerf
from .volume_rendering_utils import *  # noqa: F401,F403
from .volume_rendering_utils import volume_render_radiance_field
from .ops import training_loss  # MI355X extension: the trainer's loss (TR:355-387) and its gradients in two launches
from .ops import get_mlp_precision, set_mlp_precision  # MI355X extension: "f32" (exact, default) | "f16x3" | "bf16x3" | "f16x2" (inference only)
