"""Generates the committed golden vectors from reference code that is importable in the build
container (run once there; /root/reference does not exist on the GPU box):

  * rsh_deg0_4.npz      <- /root/reference/splat_trainer/scene/mlp/rsh.py  (rsh_cart_0..4),
                           loaded BY FILE PATH (the package __init__ would pull taichi_splatting)
  * misc_vectors.json   <- /root/reference/splat_trainer/util/misc.py (exp_lerp, saturate, soft_lt,
                           sh_to_rgb / rgb_to_sh, inverse_sigmoid, sh0) -- the controller maths that
                           consumes the rasterizer's per-point outputs (point_state.py:34-57) and the
                           reg-loss saturation (mlp_scene.py:268-288)

Usage:  TORCHDYNAMO_DISABLE=1 python tests/golden/make_golden.py
"""
import importlib.util
import json
import os

import numpy as np
import torch

REF = "/root/reference/splat_trainer"
OUT = os.path.dirname(os.path.abspath(__file__))


def load(path, name):
  spec = importlib.util.spec_from_file_location(name, path)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def main():
  torch.manual_seed(1234)
  rsh = load(os.path.join(REF, "scene/mlp/rsh.py"), "ref_rsh")
  dirs = torch.randn(64, 3, dtype=torch.float64)
  dirs = dirs / dirs.norm(dim=1, keepdim=True)
  out = {"dirs": dirs.numpy()}
  for deg in range(5):
    out[f"deg{deg}"] = getattr(rsh, f"rsh_cart_{deg}")(dirs).numpy()
  np.savez(os.path.join(OUT, "rsh_deg0_4.npz"), **out)

  misc = load(os.path.join(REF, "util/misc.py"), "ref_misc")
  a = torch.tensor([0.0, 1.0, -3.0, 2.5, 10.0, -7.0, 0.0, 1e-3], dtype=torch.float32)
  b = torch.tensor([2.0, 1.0, 5.0, -2.5, 9.0, -7.5, 0.0, 3e-3], dtype=torch.float32)
  t = torch.tensor([0.1, 0.5, 0.25, 0.9, 0.0, 0.3], dtype=torch.float32)
  vec = {
      "sh0": misc.sh0,
      "a": a.tolist(), "b": b.tolist(), "t": t.tolist(),
      "exp_lerp_0.1": misc.exp_lerp(0.1, a, b).tolist(),
      "exp_lerp_0.01": misc.exp_lerp(0.01, a, b).tolist(),
      "saturate_gain4_k2": misc.saturate(t, gain=4.0, k=2.0).tolist(),
      "saturate_default": misc.saturate(t).tolist(),
      "soft_lt_0.5": misc.soft_lt(t, 0.5).tolist(),
      "soft_gt_0.5": misc.soft_gt(t, 0.5).tolist(),
      "sh_to_rgb": misc.sh_to_rgb(a).tolist(),
      "rgb_to_sh": misc.rgb_to_sh(t).tolist(),
      "inverse_sigmoid_0.3999": float(misc.inverse_sigmoid(0.3999)),
      "lerp_0.25": misc.lerp(0.25, a, b).tolist(),
  }
  with open(os.path.join(OUT, "misc_vectors.json"), "w") as f:
    json.dump(vec, f, indent=1)
  print("wrote", os.listdir(OUT))


if __name__ == "__main__":
  main()
