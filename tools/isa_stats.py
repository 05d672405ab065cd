"""Static audit of the hot kernels' gfx950 code: registers, scratch, barriers, and the instruction mix of the
per-pair loop of K6 / K7 (packed fp32, plain VALU, transcendental, SALU, LDS, memory).

    python tools/isa_stats.py            # table
    python tools/isa_stats.py --json     # machine-readable (bench.py's VALU roofline leg and tests/test_isa_budget.py)

hipcc cross-compiles without a GPU, so this runs anywhere the toolchain is.  The per-instruction issue costs that turn
the mix into an issue-bound time are measured on the MI355X by tools/valu_rate.hip (profiles/r02_valu_issue_costs.txt).
"""
from __future__ import annotations

import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "splat-trainer_amd", "csrc")

TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32")


def compile_isa(source: str = "composite.hip") -> str:
  hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
  with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value",
                    "-Wno-unused-command-line-argument", "-o", out, source], check=True, cwd=CSRC,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def classify(op: str) -> str:
  if op.startswith("v_pk_"):
    return "valu_packed"
  if op.startswith(TRANS):
    return "valu_trans"
  if op.startswith("v_"):
    return "valu"
  if op.startswith("ds_"):
    return "lds"
  if op.startswith(("s_load", "s_buffer_load")):
    return "smem"
  if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
    return "vmem"
  if op.startswith(("s_waitcnt", "s_nop")):
    return "wait"
  if op.startswith("s_"):
    return "salu"
  return "other"


def kernels(asm: str) -> dict:
  """name -> dict(vgpr, sgpr, scratch, lds, body=[instruction lines])."""
  meta = {}
  for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n  - |\Z)", asm, flags=re.S):
    name, blk = m.group(1), m.group(2)
    def num(key):
      mm = re.search(rf"\.{key}:\s+(\d+)", blk)
      return int(mm.group(1)) if mm else None
    if num("vgpr_count") is not None:
      meta[name] = dict(vgpr=num("vgpr_count"), sgpr=num("sgpr_count"), scratch=num("private_segment_fixed_size"),
                        lds=num("group_segment_fixed_size"))
  for name in meta:
    m = re.search(rf"^{re.escape(name)}:.*?\n(.*?)^\.Lfunc_end", asm, flags=re.S | re.M)   # (a kernel may hold several s_endpgm)
    meta[name]["body"] = m.group(1).splitlines() if m else []
  return meta


def loop_mix(body, depth: int = 2) -> dict:
  """Instruction classes inside the deepest loop nest (lines tagged 'Depth=<depth>' by the compiler delimit the blocks:
  every basic block whose label comment says it belongs to that loop)."""
  mix, inside = {}, False
  for ln in body:
    s = ln.strip()
    if s.startswith(".LBB") or s.startswith("; %bb."):
      inside = f"Depth={depth}" in s or (inside and s.startswith("; %bb.") and "in Loop" in s and f"Depth={depth}" in s)
      if s.startswith(".LBB"):
        inside = f"Depth={depth}" in s
      continue
    if not inside or not s or s.startswith((";", ".")):
      continue
    op = s.split()[0]
    if op.startswith(";"):
      continue
    c = classify(op)
    mix[c] = mix.get(c, 0) + 1
  return mix


def find(meta: dict, *needles: str) -> str:
  for name in meta:
    if all(n in name for n in needles):
      return name
  raise KeyError(needles)


def audit() -> dict:
  meta = kernels(compile_isa("composite.hip"))
  out = {}
  picks = {"K7_bwd_C3": ("composite_bwd_kernelILi3E",), "K6_fwd_C3_vis": ("composite_fwd_kernelILi3ELb1ELb0ELb0E",),
           "K6_fwd_C3_vis_prefetch": ("composite_fwd_kernelILi3ELb1ELb0ELb1E",),
           "K6_segC_C3_vis": ("seg_composite_kernelILi3ELb1ELb0ELb0E",), "K6_combine_C3": ("seg_combine_kernelILi3ELb0E",)}
  for label, needles in picks.items():
    k = meta[find(meta, *needles)]
    body = k["body"]
    depth = 2 if any("Depth=2" in ln for ln in body) else 1
    out[label] = dict(vgpr=k["vgpr"], sgpr=k["sgpr"], scratch_bytes=k["scratch"], lds_bytes=k["lds"],
                      s_barrier=sum(1 for ln in body if ln.strip().startswith("s_barrier")),
                      mfma=sum(1 for ln in body if "mfma" in ln), static_loop_mix=loop_mix(body, depth),
                      instructions=sum(1 for ln in body if ln.strip() and not ln.strip().startswith((";", "."))))
  out["all_kernels"] = {n: dict(vgpr=k["vgpr"], scratch_bytes=k["scratch"]) for n, k in meta.items()}
  return out


if __name__ == "__main__":
  a = audit()
  if "--json" in sys.argv:
    print(json.dumps(a))
  else:
    for label, k in a.items():
      if label == "all_kernels":
        continue
      print(f"{label:18s} vgpr {k['vgpr']:3d}  sgpr {k['sgpr']:3d}  scratch {k['scratch_bytes']} B  lds {k['lds_bytes']} B  "
            f"s_barrier {k['s_barrier']}  mfma {k['mfma']}  loop mix {k['static_loop_mix']}")
    worst = max(a["all_kernels"].values(), key=lambda k: k["scratch_bytes"])
    print(f"{len(a['all_kernels'])} kernels in composite.hip; max scratch {worst['scratch_bytes']} B")
