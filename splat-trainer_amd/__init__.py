"""MI355X-native differentiable Gaussian-splat rasterizer: drop-in for the taichi_splatting calls
splat-trainer makes at its render boundary.

Import as ``splat_trainer_amd`` (the repo-root shim ``splat_trainer_amd.py`` loads this directory,
whose name carries a hyphen)."""
from .data_types import (CameraParams, Gaussians3D, RasterConfig, RenderedPoints, Rendering,
                         pop_raster_config)
from .renderer import GradOut, frustum_cull, project_to_image, render_gaussians, render_projected
from .sh import ShFactorCollector, evaluate_sh_at
from .loss import clamped_l1_loss, clamped_mse_loss, fused_ssim, reference_loss
from ._lib import GsplatHipError
from .compat import TaichiQueue, check_finite, count_nonfinite, random_3d_gaussians, random_camera

__all__ = ["CameraParams", "Gaussians3D", "RasterConfig", "RenderedPoints", "Rendering", "pop_raster_config",
           "frustum_cull", "project_to_image", "render_projected", "render_gaussians", "evaluate_sh_at",
           "GsplatHipError", "GradOut", "fused_ssim", "clamped_mse_loss", "clamped_l1_loss", "reference_loss", "ShFactorCollector", "TaichiQueue", "count_nonfinite",
           "check_finite", "random_camera", "random_3d_gaussians"]
