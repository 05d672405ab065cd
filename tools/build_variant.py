"""Builds an experimental variant of the library next to the product build:
    python tools/build_variant.py NAME [DEFINE[=VALUE] ...]   ->  gpurun_out/variants/libgsplat_hip_NAME.so
(gpurun_out/ is scratch; variants are for kernel experiments with tools/k67_bench.py only)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gsr_build", os.path.join(ROOT, "splat-trainer_amd", "build.py"))
build = importlib.util.module_from_spec(spec)
spec.loader.exec_module(build)
name = sys.argv[1]
defines = [a for a in sys.argv[2:] if not a.startswith("-")]
flags = [a for a in sys.argv[2:] if a.startswith("-")]          # raw compiler flags, e.g. -mllvm -amdgpu-sched-strategy=max-ilp
out_dir = os.path.join(ROOT, "variants")
os.makedirs(out_dir, exist_ok=True)
print(build.build_hip(force=True, out=os.path.join(out_dir, f"libgsplat_hip_{name}.so"), defines=defines, flags=flags))
