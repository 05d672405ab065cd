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
