"""Local (non-communication) cost of the data-parallel exchange, measured on ONE GPU: torch.distributed's collectives are
replaced by local stand-ins that move the same bytes inside the device (all_gather: this rank's block copied into every
slot; all_reduce: in place, values unchanged), so what is timed is packing, scattering, the SH-gradient rebuild over
all cameras and the replay of the controller statistics -- everything the exchange costs EXCEPT the xGMI transfers.
    python tools/dp_overhead.py [world] [workload]"""
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
import splat_trainer_amd as sta
from splat_trainer_amd.controller_math import PointState
from splat_trainer_amd.distributed import CameraShardedStep

WORLD = int(sys.argv[1]) if len(sys.argv) > 1 else 8
name = sys.argv[2] if len(sys.argv) > 2 else "c2"


class _Done:
  def wait(self):
    return True


def fake_all_gather(out, mine, group=None, async_op=False):
  out.view(WORLD, -1)[:] = mine.reshape(1, -1)
  return _Done()


def fake_all_reduce(t, op=None, group=None, async_op=False):
  return _Done()


def fake_all_to_all(out, inp, group=None, async_op=False):
  out.copy_(inp)
  return _Done()


dist.is_initialized = lambda: True
dist.get_world_size = lambda group=None: WORLD
dist.get_rank = lambda group=None: 0
dist.all_gather_into_tensor = fake_all_gather
dist.all_reduce = fake_all_reduce
dist.all_to_all_single = fake_all_to_all

import splat_trainer_amd.distributed as D
_exchange_counts = D.exchange_counts


def fake_counts(local, slots_per_rank, device, group=None):
  """Every rank reports what rank 0 has, under its own camera number (camera j lives on rank j mod world)."""
  rows = _exchange_counts(local, slots_per_rank, device, group=group)        # round trip through the (fake) gather
  return [[(rk + s * WORLD) if rows[s][0] >= 0 else -1] + list(rows[s][1:]) for rk in range(WORLD)
          for s in range(slots_per_rank)]


D.exchange_counts = fake_counts

g, cams, w = bench.make_workload(name, WORLD)
dev = torch.device("cuda:0")
cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True, blur_cov=0.3, antialias=False)
params = [t.to(dev).requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
scene = sta.Gaussians3D(position=params[0], rotation=params[2], log_scaling=params[1], alpha_logit=params[3], feature=params[4])
N = params[0].shape[0]
batch = [c.to(dev) for c in cams[:WORLD]]
target = torch.full((w["h"], w["w"], 3), 0.5, device=dev)


def render_backward(j, cam, grad_out, collector):
  with torch.enable_grad():
    r = sta.render_gaussians(scene, cam, cfg, use_sh=True, grad_out=grad_out, sh_collector=collector)
    sta.clamped_mse_loss(r.image, target).backward()
  return r


for world in (1, WORLD):
  dp = CameraShardedStep(params, world, 0, early_gather=__import__('os').environ.get('DP_EARLY', '1') == '1')
  state = PointState.new_zeros(N, dev)
  for _ in range(30):
    dp.run(batch[:world] if world == 1 else batch, render_backward, point_state=state)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(40):
    dp.run(batch[:world] if world == 1 else batch, render_backward, point_state=state)
  torch.cuda.synchronize()
  ms = (time.perf_counter() - t0) / 40 * 1e3
  print(f"{name}: world {world}: {ms:.3f} ms per step (rank 0 renders 1 camera; exchange for {world} cameras, transfers local)")
