"""CPU oracle of the rasterizer path.  TEST INFRASTRUCTURE ONLY: importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from splat-trainer_amd/."""
