"""The oracle (and the harness-side controller maths) against the golden vectors generated from the
reference's own importable modules (tests/golden/make_golden.py: rsh.py and util/misc.py)."""
import json
import os

import numpy as np
import torch

from helpers import oracle


def test_sh_basis_matches_reference_rsh(golden_dir):
  z = np.load(os.path.join(golden_dir, "rsh_deg0_4.npz"))
  dirs = torch.from_numpy(z["dirs"])
  for deg in range(4):
    want = torch.from_numpy(z[f"deg{deg}"])
    got = oracle.sh_basis(dirs, deg)
    assert got.shape == want.shape == (64, (deg + 1) ** 2)
    assert torch.allclose(got, want, rtol=0, atol=1e-12), f"degree {deg}"


def test_controller_math_matches_reference_misc(golden_dir):
  from splat_trainer_amd.controller_math import exp_lerp
  v = json.load(open(os.path.join(golden_dir, "misc_vectors.json")))
  a, b = torch.tensor(v["a"]), torch.tensor(v["b"])
  assert torch.allclose(exp_lerp(0.1, a, b), torch.tensor(v["exp_lerp_0.1"]), rtol=1e-6, atol=1e-7)
  assert torch.allclose(exp_lerp(0.01, a, b), torch.tensor(v["exp_lerp_0.01"]), rtol=1e-6, atol=1e-7)
  # the survey's spot values
  assert torch.allclose(exp_lerp(0.1, torch.tensor([0., 1., -3.]), torch.tensor([2., 1., 5.])),
                        torch.tensor([0.4940, 1.0000, 2.7004]), atol=1e-4)


def test_sh_rgb_offset_convention(golden_dir):
  """evaluate_sh_at of a DC-only coefficient equals the reference's sh_to_rgb (misc.py:45-49)."""
  v = json.load(open(os.path.join(golden_dir, "misc_vectors.json")))
  a = torch.tensor(v["a"], dtype=torch.float64)
  sh = a.reshape(-1, 1, 1).repeat(1, 3, 1)
  pos = torch.randn(a.numel(), 3, dtype=torch.float64) + 3.0
  col = oracle.evaluate_sh_at(sh, pos, torch.arange(a.numel()), torch.zeros(3, dtype=torch.float64))
  assert torch.allclose(col[:, 0], torch.tensor(v["sh_to_rgb"], dtype=torch.float64), atol=1e-6)
  assert abs(oracle.SH_C0 - v["sh0"]) < 1e-15
