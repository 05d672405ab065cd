"""ctypes binding of the C ABI in include/gsplat_hip.h.

There is no CPU fallback: if ``libgsplat_hip.so`` is missing or fails to load, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# GSPLAT_HIP_LIB: load another build of the same library (kernel experiments, tools/k67_bench.py); never a fallback
LIB_PATH = os.environ.get("GSPLAT_HIP_LIB") or os.path.join(PKG_DIR, "libgsplat_hip.so")

ABI_VERSION = 30
PREFETCH_MIN_ROWS = 1_000_000      # include/gsplat_hip.h: GSR_PREFETCH_MIN_ROWS


class GsrRasterParamsC(C.Structure):
  _fields_ = [("alpha_threshold", C.c_float), ("clamp_max_alpha", C.c_float), ("T_eps", C.c_float),
              ("q_max", C.c_float), ("blur", C.c_float), ("antialias", C.c_int32), ("tile_size", C.c_int32),
              ("margin_px", C.c_float)]


class GsrSegmentsC(C.Structure):
  _fields_ = [("tile_seg", C.c_void_p), ("seg_desc", C.c_void_p), ("seg_total", C.c_void_p), ("capacity", C.c_int64),
              ("heavy_capacity", C.c_int64), ("seg_P", C.c_void_p), ("seg_TC", C.c_void_p), ("seg_last", C.c_void_p),
              ("seg_median", C.c_void_p), ("tile_order", C.c_void_p)]


class GsrFrameC(C.Structure):
  _fields_ = [("position", C.c_void_p), ("log_scaling", C.c_void_p), ("rotation_xyzw", C.c_void_p),
              ("alpha_logit", C.c_void_p), ("sh_features", C.c_void_p), ("N", C.c_int64), ("K", C.c_int32),
              ("W", C.c_int32), ("H", C.c_int32), ("T_camera_world", C.c_void_p), ("projection", C.c_void_p),
              ("camera_pos", C.c_void_p), ("near_plane", C.c_float), ("far_plane", C.c_float),
              ("params", GsrRasterParamsC), ("want_jacobian", C.c_int32), ("want_median", C.c_int32),
              ("compute_visibility", C.c_int32), ("needs_grad", C.c_int32), ("seg_pairs", C.c_int32),
              ("seg_min_pairs", C.c_int32), ("pair_capacity", C.c_int64), ("gaussians2d", C.c_void_p),
              ("depth", C.c_void_p), ("features", C.c_void_p), ("C", C.c_int32), ("depth_order", C.c_void_p),
              ("side_stream", C.c_void_p), ("event_fork", C.c_void_p), ("event_join", C.c_void_p)]


FRAME_PLAN_FIELDS = ("out_bytes", "work_bytes", "zero_begin", "zero_bytes", "prune_cost", "split_score", "counts",
                     "tile_range", "vis_partial", "indexes", "rows", "screen_scale", "jacobian", "visibility", "image",
                     "final_T", "last", "median", "count", "offsets", "vals_a", "vals_b", "tvals_a", "tvals_b", "trank_a",
                     "trank_b", "pair_vis", "seg_tables", "seg_pix", "seg_last", "seg_capacity", "seg_heavy_capacity",
                     "cull_ws", "sort_ws", "scan_ws", "tsort_ws", "keys_a", "keys_b", "tile_hits", "tkeys_a", "tkeys_b",
                     "cull_ws_bytes", "sort_ws_bytes", "scan_ws_bytes", "tsort_ws_bytes")


class GsrFramePlanC(C.Structure):
  _fields_ = [(name, C.c_int64) for name in FRAME_PLAN_FIELDS]


class GsrFrameBackwardC(C.Structure):
  _fields_ = [("position", C.c_void_p), ("log_scaling", C.c_void_p), ("rotation_xyzw", C.c_void_p),
              ("alpha_logit", C.c_void_p), ("sh_features", C.c_void_p), ("N", C.c_int64), ("K", C.c_int32),
              ("W", C.c_int32), ("H", C.c_int32), ("C", C.c_int32), ("T_camera_world", C.c_void_p),
              ("projection", C.c_void_p), ("camera_pos", C.c_void_p), ("params", GsrRasterParamsC), ("M", C.c_int64),
              ("O", C.c_int64), ("indexes", C.c_void_p), ("rows", C.c_void_p), ("order", C.c_void_p),
              ("count", C.c_void_p), ("offsets", C.c_void_p), ("sorted_splat", C.c_void_p), ("sorted_inst", C.c_void_p),
              ("pair_vis", C.c_void_p), ("vis_partial", C.c_void_p), ("tile_range", C.c_void_p), ("final_T", C.c_void_p),
              ("last", C.c_void_p), ("image", C.c_void_p), ("jacobian", C.c_void_p), ("segments", C.c_void_p),
              ("d_image", C.c_void_p), ("d_gaussians2d", C.c_void_p), ("d_depth", C.c_void_p), ("partial", C.c_void_p),
              ("grad_rows", C.c_void_p), ("inverse", C.c_void_p), ("d_colors", C.c_void_p), ("d_position", C.c_void_p),
              ("d_log_scaling", C.c_void_p), ("d_rotation", C.c_void_p), ("d_alpha_logit", C.c_void_p),
              ("mode", C.c_int32), ("d_sh", C.c_void_p), ("sh_mode", C.c_int32), ("prune_cost", C.c_void_p),
              ("split_score", C.c_void_p), ("visibility", C.c_void_p)]


class GsrFrameResultC(C.Structure):
  _fields_ = [("order", C.c_int64), ("sorted_inst", C.c_int64), ("sorted_splat", C.c_int64), ("segments", GsrSegmentsC),
              ("has_segments", C.c_int32)]


class GsrColumnC(C.Structure):
  _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("tail", C.c_void_p), ("width_dwords", C.c_int32)]


MAX_COLUMNS = 32


def raster_params(config) -> GsrRasterParamsC:
  blur = float(config.blur_cov) + (float(config.aa_blur) if config.antialias else 0.0)
  return GsrRasterParamsC(float(config.alpha_threshold), float(config.clamp_max_alpha),
                          float(config.transmittance_eps), float(config.gaussian_scale) ** 2, blur,
                          1 if config.antialias else 0, int(config.tile_size),
                          float(config.margin_tiles * config.tile_size))


_p = C.c_void_p
_i64, _i32, _u32, _f, _sz = C.c_int64, C.c_int32, C.c_uint32, C.c_float, C.c_size_t
_pp = C.POINTER(GsrRasterParamsC)
_ps = C.POINTER(GsrSegmentsC)

# name -> (restype, argtypes); every symbol declared in include/gsplat_hip.h
PROTOTYPES = {
    "gsr_abi_version": (C.c_int, []),
    "gsr_error_string": (C.c_char_p, [C.c_int]),
    "gsr_scan_workspace_bytes": (_sz, [_i64]),
    "gsr_exclusive_scan_u32": (C.c_int, [_p, _p, _i64, _p, _p, _sz, _p]),
    "gsr_exclusive_scan_u32_checked": (C.c_int, [_p, _p, _i64, _p, _p, _p, _sz, _p]),
    "gsr_sort_workspace_bytes": (_sz, [_i64]),
    "gsr_sort_pairs_u32": (C.c_int, [_p, _p, _p, _p, _i64, C.c_int, C.c_int, C.c_int, _p, _sz, _p, _p]),
    "gsr_sort_pairs2_u32": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, C.c_int, C.c_int, C.c_int, _p, _sz, _p, _p]),
    "gsr_cull_workspace_bytes": (_sz, [_i64]),
    "gsr_frustum_cull": (C.c_int, [_p, _i64, _p, _p, _i32, _i32, _f, _f, _f, _p, _p, _p, _sz, _p]),
    "gsr_project_forward": (C.c_int, [_p, _p, _p, _p, _p, _i64, _p, _p, _pp, _p, _p, _p, _p, _u32, _u32, _p]),
    "gsr_project_backward": (C.c_int, [_p, _p, _p, _p, _p, _i64, _p, _p, _pp, _p, _p, _p, _p, _p, _p, _i32, _p]),
    "gsr_sh_forward": (C.c_int, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p]),
    "gsr_sh_backward": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _p, _p, _p, _i32, _p]),
    "gsr_sh_backward_multi": (C.c_int, [_p, _i64, _p, _i64, _i32, _p, _p, _i64, _i32, _p, _p, _i32, _p]),
    "gsr_inverse_map": (C.c_int, [_p, _i64, _i64, _p, _p]),
    "gsr_sh_backward_dense": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i32, _p, _p, _p, _p, _p]),
    "gsr_depth_key_range": (C.c_int, [_f, _f, _p, _p]),
    "gsr_depth_keys": (C.c_int, [_p, _i64, _u32, _u32, _p, _p]),
    "gsr_depth_keys_from_positions": (C.c_int, [_p, _p, _i64, _p, _p, _p, _u32, _u32, _p, _p]),
    "gsr_project_sh_forward": (C.c_int, [_p, _p, _p, _p, _p, _i32, _p, _i64, _p, _p, _p, _pp, _p, _p, _p, _p, _p, _u32,
                                         _u32, _p]),
    "gsr_project_backward_rows": (C.c_int, [_p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _pp, _p, _p, _p, _p, _p, _p, _p,
                                            _p, _p, _i32, _p, _p, _p, _p, _p]),
    "gsr_pack_rows": (C.c_int, [_p, _p, _p, _i64, _i32, _pp, _p, _p, _p]),
    "gsr_tile_count": (C.c_int, [_p, _p, _i64, _i32, _i32, _pp, _p, _p, _p, _p]),
    "gsr_tile_count_offsets_workspace_bytes": (_sz, [_i64]),
    "gsr_tile_count_offsets": (C.c_int, [_p, _p, _i64, _i32, _i32, _pp, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gsr_tile_emit": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _pp, _p, _p, _i64, _p, _p]),
    "gsr_tile_ranges": (C.c_int, [_p, _i64, _i32, _p, _p, _p]),
    "gsr_segment_thresholds": (C.c_int, [_i32, _i32, _i64, _i32, _i32, _p, _p]),
    "gsr_segment_capacity": (_i64, [_i64, _i32, _i32, _i32, _i32, _i32]),
    "gsr_segment_heavy_capacity": (_i64, [_i64, _i32, _i32, _i32, _i32, _i32]),
    "gsr_segment_plan": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i64, _p, _i64, _i64, _p, _p, _p, _p, _p]),
    "gsr_composite_forward": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _pp, _p, _p, _p, _p, _p, _p, _ps, _i32, _p]),
    "gsr_composite_backward": (C.c_int, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _pp, _p, _p, _p, _p, _p, _ps, _p]),
    "gsr_opt_point_weights": (C.c_int, [_p, _p, _i64, _p, _p, _f, _f, _f, _f, _i32, _p, _p]),
    "gsr_opt_step": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _f, _f, _f, _f, _f, _p]),
    "gsr_pixel_loss_workspace_bytes": (_sz, [_i64]),
    "gsr_pixel_loss_forward": (C.c_int, [_p, _p, _i64, _i32, _f, _f, _p, _p, _sz, _p]),
    "gsr_pixel_loss_backward": (C.c_int, [_p, _p, _i64, _i32, _f, _f, _p, _p, _p]),
    "gsr_ssim_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "gsr_ssim_forward": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _sz, _p]),
    "gsr_ssim_backward": (C.c_int, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p]),
    "gsr_msloss_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "gsr_msloss_forward": (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _f, _f, _f, _f, _f, _p, _p, _sz, _p]),
    "gsr_msloss_backward": (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _f, _f, _f, _f, _f, _p, _p, _sz, _p, _p]),
    "gsr_select_workspace_bytes": (_sz, [_i64]),
    "gsr_select_n": (C.c_int, [_p, _i64, _i64, _i32, _p, _p, _sz, _p]),
    "gsr_compact_workspace_bytes": (_sz, [_i64]),
    "gsr_compact_offsets": (C.c_int, [_p, _i64, _p, _p, _p, _sz, _p]),
    "gsr_compact_columns": (C.c_int, [_p, _i64, _p, _i64, _i64, C.POINTER(GsrColumnC), _i32, _p]),
    "gsr_point_basis": (C.c_int, [_p, _p, _p, _i64, _f, _p, _p]),
    "gsr_dp_pack": (C.c_int, [_p, _p, _p, _p, _p, _i32, _p, _i64, _i64, _p, _p, _p, _p, _p]),
    "gsr_dp_replay": (C.c_int, [_p, _i64, _p, _i32, _i64, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p]),
    "gsr_dp_pack_sharded": (C.c_int, [_p, _p, _p, _p, _p, _i32, _p, _i64, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p]),
    "gsr_dp_replay_slice": (C.c_int, [_p, _i32, _i32, _i64, _i32, _i32, _f, _f, _p, _p, _p, _p]),
    "gsr_dp_finish": (C.c_int, [_p, _i32, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "gsr_point_state_add": (C.c_int, [_p, _p, _i32, _p, _p, _p, _i64, _f, _f, _p, _p, _p, _p, _p, _p, _p]),
    "gsr_frame_plan": (C.c_int, [C.POINTER(GsrFrameC), C.POINTER(GsrFramePlanC)]),
    "gsr_struct_bytes": (_i64, [_i32]),
    "gsr_frame_backward": (C.c_int, [C.POINTER(GsrFrameBackwardC), _p, _p, _p]),
    "gsr_frame_backward_stages": (C.c_int, [C.POINTER(GsrFrameBackwardC), _i32, _p, _p, _p]),
    "gsr_dp_pack_factors_rows": (C.c_int, [_p, _p, _p, _i64, _i64, _p, _p]),
    "gsr_frame_forward": (C.c_int, [C.POINTER(GsrFrameC), C.POINTER(GsrFramePlanC), _p, _p, C.POINTER(GsrFrameResultC), _p,
                                    _p, _p, _p, _p]),
    "gsr_reduce_visibility": (C.c_int, [_p, _p, _p, _p, _i64, _p, _i64, _p, _p]),
    "gsr_reduce_gradients": (C.c_int, [_p, _p, _p, _p, _p, _i64, _p, _p]),
    "gsr_unpack_grad_rows": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p]),
}

_lib = None
_lock = threading.Lock()


def current_stream_ptr() -> int:
  """hipStream_t of torch's current stream on the current device.  (torch.cuda.current_stream() costs ~13 us per call
  on this stack -- device-availability probing -- and a step makes a dozen launches; the raw getter costs well under 1 us.)"""
  return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())    # (0 = the null stream; ctypes passes the int)


def current_stream() -> "torch.cuda.Stream":
  """torch.cuda.current_stream() of the current device without the availability probe (explicit device index)."""
  return torch.cuda.current_stream(torch._C._cuda_getDevice())


class GsplatHipError(RuntimeError):
  pass


def load() -> C.CDLL:
  """Loads (once) and returns the HIP library; raises GsplatHipError when it is absent."""
  global _lib
  if _lib is not None:
    return _lib
  with _lock:
    if _lib is not None:
      return _lib
    if not os.path.exists(LIB_PATH):
      raise GsplatHipError(
          f"{LIB_PATH} not found: build it with `python splat-trainer_amd/build.py` "
          "(hipcc --offload-arch=gfx950). There is no CPU fallback for the rasterizer path.")
    try:
      lib = C.CDLL(LIB_PATH)
    except OSError as e:
      raise GsplatHipError(f"failed to load {LIB_PATH}: {e}") from e
    for name, (restype, argtypes) in PROTOTYPES.items():
      fn = getattr(lib, name)
      fn.restype = restype
      fn.argtypes = argtypes
    if lib.gsr_abi_version() != ABI_VERSION:
      raise GsplatHipError(f"ABI mismatch: library {lib.gsr_abi_version()} != binding {ABI_VERSION}; rebuild")
    # the ctypes mirrors of the ABI's structs must have the layout the library was compiled with (gsr_struct_bytes)
    for which, mirror in enumerate((GsrRasterParamsC, GsrSegmentsC, GsrFrameC, GsrFramePlanC, GsrFrameResultC,
                                    GsrFrameBackwardC)):
      if lib.gsr_struct_bytes(which) != C.sizeof(mirror):
        raise GsplatHipError(f"struct layout mismatch: {mirror.__name__} is {C.sizeof(mirror)} bytes in the binding, "
                             f"{lib.gsr_struct_bytes(which)} in the library; rebuild")
    _lib = lib
  return _lib


def check(code: int, what: str) -> int:
  if code < 0:
    msg = load().gsr_error_string(code).decode()
    raise GsplatHipError(f"{what} failed: {msg} ({code})")
  return code
