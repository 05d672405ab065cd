"""PLY interchange.  The first test is the reference's own round-trip test
(splat_trainer/scene/io.py:149-165) restated; the others pin the field names/order and the two conventions that
matter for 3DGS compatibility (wxyz quaternion on disk, channel-major f_rest)."""
import numpy as np
import torch

from splat_trainer_amd import ply_io


def test_read_write_round_trip_like_reference(tmp_path):
  gen = torch.Generator().manual_seed(0)
  for i in range(10):
    g = ply_io.random_gaussians((i + 1) * 1000, 3, generator=gen)
    ply_io.write_gaussians(tmp_path / f"gaussians_{i}.ply", g)
    g2 = ply_io.read_gaussians(tmp_path / f"gaussians_{i}.ply")
    assert torch.allclose(g.position, g2.position)
    assert torch.allclose(g.rotation, g2.rotation, atol=1e-6)
    assert torch.allclose(torch.sigmoid(g.alpha_logit), torch.sigmoid(g2.alpha_logit))
    assert torch.allclose(g.log_scaling, g2.log_scaling)
    assert torch.allclose(g.feature, g2.feature)
    assert g2.feature.shape == (g.position.shape[0], 3, 16)


def test_field_layout_matches_reference(tmp_path):
  g = ply_io.random_gaussians(7, 2, generator=torch.Generator().manual_seed(1))
  path = tmp_path / "g.ply"
  ply_io.write_gaussians(path, g, with_sh=True)
  raw = open(path, "rb").read()
  header = raw[:raw.index(b"end_header\n")].decode().split("\n")
  assert header[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 7"]
  names = [ln.split()[2] for ln in header if ln.startswith("property")]
  assert names[:11] == ["x", "y", "z", "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
  assert names[11:14] == ["f_dc_0", "f_dc_1", "f_dc_2"] and names[14:] == [f"f_rest_{i}" for i in range(24)]
  v = ply_io.read_ply(path)
  # quaternion stored wxyz: rot_0 is the in-memory w (index 3)
  q = torch.nn.functional.normalize(g.rotation, dim=1)
  assert np.allclose(v["rot_0"], q[:, 3].numpy()) and np.allclose(v["rot_1"], q[:, 0].numpy())
  # f_rest is channel-major: f_rest_0..7 are channel 0's coefficients 1..8
  assert np.allclose(v["f_rest_0"], g.feature[:, 0, 1].numpy()) and np.allclose(v["f_rest_8"], g.feature[:, 1, 1].numpy())
  assert np.allclose(v["f_dc_2"], g.feature[:, 2, 0].numpy())
  assert np.allclose(v["opacity"], g.alpha_logit[:, 0].numpy()) and np.allclose(v["scale_1"], g.log_scaling[:, 1].numpy())


def test_flat_features_and_ascii(tmp_path):
  g = ply_io.random_gaussians(5, 0, generator=torch.Generator().manual_seed(2))
  g.feature = torch.randn(5, 6)
  ply_io.write_gaussians(tmp_path / "f.ply", g, with_sh=False)
  g2 = ply_io.read_gaussians(tmp_path / "f.ply", with_sh=False)
  assert torch.allclose(g.feature, g2.feature) and g2.feature.shape == (5, 6)
  # an ascii file with the same header fields parses to the same values
  v = ply_io.read_ply(tmp_path / "f.ply")
  lines = ["ply", "format ascii 1.0", "element vertex 5"] + [f"property float {n}" for n in v.dtype.names] + ["end_header"]
  lines += [" ".join(repr(float(v[n][i])) for n in v.dtype.names) for i in range(5)]
  (tmp_path / "a.ply").write_text("\n".join(lines) + "\n")
  g3 = ply_io.read_gaussians(tmp_path / "a.ply", with_sh=False)
  assert torch.allclose(g3.position, g.position) and torch.allclose(g3.feature, g.feature)
