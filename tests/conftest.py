import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_libs():
  """Builds (if stale) and returns the paths of the HIP library and the host-math test shim."""
  import importlib
  build = importlib.import_module("splat_trainer_amd.build")
  return build.build_all()


@pytest.fixture(scope="session")
def golden_dir():
  return os.path.join(ROOT, "tests", "golden")


def pytest_terminal_summary(terminalreporter):
  """Observed parity numbers (tests/helpers.py: PARITY_LOG), printed and kept under gpurun_out/."""
  try:
    from helpers import PARITY_LOG
  except Exception:
    return
  if not PARITY_LOG:
    return
  lines = [f"{'case':58s} {'tensor':14s} {'max rel err':>12s} {'> tol':>9s} {'per-entry':>10s} {'of':>10s} {'tol':>8s}"]
  for label, key, worst, above, n, tol, per_entry in PARITY_LOG:
    lines.append(f"{label[:58]:58s} {key:14s} {worst:12.3e} {above:9d} {per_entry:10d} {n:10d} {tol:8.0e}")
  terminalreporter.write_sep("-", "observed parity (HIP vs oracle): max error and entries above tol relative to the tensor's largest magnitude; per-entry = entries with |a-b| > tol |b| + 1e-2 tol max|b|")
  for ln in lines:
    terminalreporter.write_line(ln)
  try:
    from helpers import EXPLAINED_LOG, FLIP_ULPS
  except Exception:
    EXPLAINED_LOG, FLIP_ULPS = [], 0
  if EXPLAINED_LOG:
    lines.append("")
    lines.append(f"explained flips (helpers.compare_explained): entries above tol must sit on a pixel / a splat within {FLIP_ULPS:g} fp32 "
                 f"operand-ulps of a decision boundary of the oracle's walk ('flagged' = share of pixels / splats that close to one), "
                 f"or -- gradient rows -- belong to a splat whose conic is ill-conditioned (error within COND_GAIN x condition of the row)")
    lines.append("per-point / gradient rows: 'own' = share of splats whose OWN decision at some pixel is that close; 'behind' = rows above tol "
                 "whose splat is not such a candidate itself but is composited behind one on a flagged pixel")
    lines.append(f"{'case':58s} {'tensor':14s} {'> tol':>9s} {'unexplained':>12s} {'largest margin':>15s} {'flagged':>9s} {'ill-cond rows':>14s} {'behind':>7s} {'own':>8s}")
    for label, key, n_bad, unexplained, largest, share, conditioned, behind, own_share in EXPLAINED_LOG:
      if n_bad:
        lines.append(f"{label[:58]:58s} {key:14s} {n_bad:9d} {unexplained:12d} {largest:15.2f} {share:9.4f} {conditioned:14d} {behind:7d} {own_share:8.4f}")
    lines.append(f"total: {sum(e[2] for e in EXPLAINED_LOG)} entries above tol in {len(EXPLAINED_LOG)} comparisons, "
                 f"{sum(e[3] for e in EXPLAINED_LOG)} unexplained")
    for ln in lines[-(4 + sum(1 for e in EXPLAINED_LOG if e[2])):]:
      terminalreporter.write_line(ln)
  out_dir = os.path.join(ROOT, "gpurun_out")
  try:
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "parity_observed.txt"), "w") as f:
      f.write("\n".join(lines) + "\n")
  except OSError:
    pass
