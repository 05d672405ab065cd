"""Builds the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU).

    python splat-trainer_amd/build.py [--force]

Outputs ``splat-trainer_amd/libgsplat_hip.so`` (C ABI declared in include/gsplat_hip.h) and the small
host-only ``splat-trainer_amd/libgsr_hostmath.so`` used by the CPU unit tests of the per-splat maths
(gsr_math.h compiled with g++: same source as the device code, no GPU needed).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libgsplat_hip.so")
HOSTMATH_PATH = os.path.join(PKG_DIR, "libgsr_hostmath.so")
HIP_SOURCES = ["prims.hip", "geometry.hip", "binning.hip", "composite.hip", "frame.hip", "ssim.hip", "optim.hip", "densify.hip"]
HEADERS = ["gsr_math.h", "gsr_device.h", "gsr_dpp_reduce.h", "composite_k7_windows.inc", "composite_k7_xflex.inc",
           os.path.join("..", "..", "include", "gsplat_hip.h")]


def _newer(target: str, sources) -> bool:
  if not os.path.exists(target):
    return False
  t = os.path.getmtime(target)
  return all(os.path.getmtime(s) <= t for s in sources)


def _hipcc() -> str:
  exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
  if not os.path.exists(exe):
    raise RuntimeError("hipcc not found: the HIP library cannot be built (no fallback path exists)")
  return exe


def build_hip(force: bool = False, verbose: bool = False, out: str = LIB_PATH, defines=(), flags=()) -> str:
  """``out`` / ``defines`` / ``flags``: experimental variants (``-DNAME[=v]``, raw compiler flags) built next to the product library."""
  srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
  deps = srcs + [os.path.join(CSRC, h) for h in HEADERS]
  if not force and _newer(out, deps):
    return out
  cmd = [_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
         "-Wno-unused-value", "-o", out] + [f"-D{d}" for d in defines] + list(flags) + srcs
  if verbose:
    print(" ".join(cmd), flush=True)
  subprocess.run(cmd, check=True, cwd=CSRC)
  return out


def build_hostmath(force: bool = False, verbose: bool = False) -> str:
  src = os.path.join(CSRC, "hostmath_shim.cpp")
  deps = [src, os.path.join(CSRC, "gsr_math.h")]
  if not force and _newer(HOSTMATH_PATH, deps):
    return HOSTMATH_PATH
  cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", HOSTMATH_PATH, src]
  if verbose:
    print(" ".join(cmd), flush=True)
  subprocess.run(cmd, check=True, cwd=CSRC)
  return HOSTMATH_PATH


def build_all(force: bool = False, verbose: bool = False):
  return build_hip(force, verbose), build_hostmath(force, verbose)


if __name__ == "__main__":
  print(build_all(force="--force" in sys.argv, verbose=True))
