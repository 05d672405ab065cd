"""Timing of the fused sparse optimizer step (csrc/optim.hip) at c2 / c3 sizes, all rows visible."""
import sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import splat_trainer_amd as sta
from splat_trainer_amd import optim
from splat_trainer_amd.harness import point_basis
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
GROUPS = dict(position=dict(lr=0.3, type="local_vector"), log_scaling=dict(lr=0.08), rotation=dict(lr=0.01, type="vector"),
              alpha_logit=dict(lr=0.1), feature=dict(lr=5.0, type="vector"))
g = torch.Generator(device="cuda").manual_seed(0)
t = dict(position=torch.randn(n, 3, device="cuda", generator=g), log_scaling=torch.randn(n, 3, device="cuda", generator=g) - 3,
         rotation=torch.nn.functional.normalize(torch.randn(n, 4, device="cuda", generator=g), dim=1),
         alpha_logit=torch.randn(n, 1, device="cuda", generator=g), feature=torch.randn(n, 3, 16, device="cuda", generator=g),
         visible=torch.rand(n, device="cuda", generator=g))
pc = optim.ParameterClass(t, GROUPS, optimizer=optim.VisibilityAwareLaProp, betas=(0.8, 0.95), vis_beta=0.999, grad_clip=2.0)
for k in GROUPS: pc.tensors[k].grad = torch.randn_like(pc.tensors[k])
def hip_step():
  vis_idx = pc.visible.nonzero().squeeze(1)
  basis = point_basis(pc.log_scaling[vis_idx].detach(), pc.rotation[vis_idx].detach()).contiguous()
  pc.step(visibility=pc.visible[vis_idx], indexes=vis_idx, basis=basis)
def hip_step_only(vis_idx, w, basis):
  pc.step(visibility=w, indexes=vis_idx, basis=basis)
def timeit(f, *a, reps=20):
  for _ in range(3): f(*a)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(reps): f(*a)
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
vis_idx = pc.visible.nonzero().squeeze(1); w = pc.visible[vis_idx]
basis = point_basis(pc.log_scaling[vis_idx].detach(), pc.rotation[vis_idx].detach()).contiguous()
print(f"n={n}: scene.step (nonzero + basis + fused step) {timeit(hip_step):.3f} ms; fused step alone "
      f"{timeit(hip_step_only, vis_idx, w, basis):.3f} ms")
