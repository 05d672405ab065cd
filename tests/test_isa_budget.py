"""Static budget of the hot kernels' gfx950 code (hipcc cross-compiles without a GPU): a toolchain or source change must
not silently spill K6 / K7 to scratch, push them over the register budget their occupancy rests on, or introduce
barriers / matrix instructions the design excludes."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _audit():
  spec = importlib.util.spec_from_file_location("isa_stats", os.path.join(ROOT, "tools", "isa_stats.py"))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod.audit()


def test_composite_kernels_stay_inside_their_register_budget():
  a = _audit()
  # no kernel of composite.hip may use scratch memory (a spill in the per-pair loop costs more than the occupancy it buys)
  spilled = {n: k["scratch_bytes"] for n, k in a["all_kernels"].items() if k["scratch_bytes"]}
  assert not spilled, spilled
  k7, k6 = a["K7_bwd_C3"], a["K6_fwd_C3_vis"]
  assert k7["vgpr"] <= 80, k7["vgpr"]            # 6 waves per SIMD (512 / 80); 74 at the time of writing
  assert k6["vgpr"] <= 64, k6["vgpr"]            # 8 waves per SIMD; 42 at the time of writing
  for label in ("K7_bwd_C3", "K6_fwd_C3_vis", "K6_segC_C3_vis", "K6_combine_C3"):
    assert a[label]["s_barrier"] == 0, label     # one wave per workgroup: LDS exchange is fenced, never barriered
    assert a[label]["mfma"] == 0, label          # no dense contraction on this path
  mix = k7["static_loop_mix"]
  assert mix.get("valu_packed", 0) >= 40 and mix.get("lds", 0) >= 8      # packed fp32 pairs + the LDS reduction are there
