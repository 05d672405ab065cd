"""K5 primitives on the GPU through the C ABI: exclusive scan and the stable radix sort (bit-exact)."""
import ctypes as C

import numpy as np
import pytest
import torch

from splat_trainer_amd import _lib

pytestmark = pytest.mark.gpu


def _ptr(t):
  return C.c_void_p(t.data_ptr()) if t.numel() else None


def _stream():
  return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 4095, 4096, 4097, 100_000, 5_000_001])
def test_exclusive_scan(n):
  lib = _lib.load()
  rng = np.random.default_rng(n)
  x = rng.integers(0, 50, size=n, dtype=np.int64).astype(np.int32)
  d = torch.from_numpy(x).cuda()
  out = torch.empty_like(d)
  total = torch.full((1,), -1, dtype=torch.int32, device="cuda")
  nbytes = lib.gsr_scan_workspace_bytes(n)
  ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
  _lib.check(lib.gsr_exclusive_scan_u32(_ptr(d), _ptr(out), n, _ptr(total), _ptr(ws), nbytes, _stream()), "scan")
  want = np.cumsum(x, dtype=np.int64) - x
  assert np.array_equal(out.cpu().numpy(), want.astype(np.int32))
  assert total.item() == int(x.sum())


def _sort(keys: np.ndarray, begin_bit: int, end_bit: int, iota=True, vals=None):
  lib = _lib.load()
  n = keys.shape[0]
  ka = torch.from_numpy(keys.view(np.int32)).cuda()
  va = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")
  if not iota:
    va[:n] = torch.from_numpy(vals.view(np.int32)).cuda()
  kb, vb = torch.zeros_like(ka), torch.zeros_like(va)
  nbytes = lib.gsr_sort_workspace_bytes(n)
  ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
  where = _lib.check(lib.gsr_sort_pairs_u32(_ptr(ka), _ptr(va), _ptr(kb), _ptr(vb), n, 1 if iota else 0, begin_bit,
                                            end_bit, _ptr(ws), nbytes, None, _stream()), "sort")
  k, v = (kb, vb) if where == 1 else (ka, va)
  return k.cpu().numpy().view(np.uint32)[:n], v.cpu().numpy().view(np.uint32)[:n]


@pytest.mark.parametrize("n", [1, 255, 256, 257, 1024, 1025, 4096, 4097, 70_001, 3_000_000, 6_000_003])
def test_radix_sort_full_keys_is_stable(n):
  rng = np.random.default_rng(n)
  keys = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
  keys[rng.integers(0, n, size=n // 3)] = keys[0]          # plenty of ties
  k, v = _sort(keys, 0, 32)
  order = np.argsort(keys, kind="stable")
  assert np.array_equal(k, keys[order])
  assert np.array_equal(v, order.astype(np.uint32))


@pytest.mark.parametrize("n,distinct", [(1_000_000, 1), (1_000_000, 3), (4_500_000, 2), (777_777, 300)])
def test_radix_sort_few_distinct_keys(n, distinct):
  """Every block holds the same few digits (long runs per digit, most histogram bins empty)."""
  rng = np.random.default_rng(distinct)
  pool = rng.integers(0, 2 ** 32, size=distinct, dtype=np.uint64).astype(np.uint32)
  keys = pool[rng.integers(0, distinct, size=n)]
  k, v = _sort(keys, 0, 32)
  order = np.argsort(keys, kind="stable")
  assert np.array_equal(k, keys[order])
  assert np.array_equal(v, order.astype(np.uint32))


def test_radix_sort_repeated_calls_reuse_one_workspace():
  """Nothing is assumed about the workspace's contents: a garbage-filled one is reused call after call."""
  lib = _lib.load()
  n = 300_000
  rng = np.random.default_rng(5)
  nbytes = lib.gsr_sort_workspace_bytes(n)
  ws = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device="cuda")
  for rep in range(3):
    keys = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    ka = torch.from_numpy(keys.view(np.int32)).cuda()
    va, kb, vb = torch.zeros_like(ka), torch.zeros_like(ka), torch.zeros_like(ka)
    where = _lib.check(lib.gsr_sort_pairs_u32(_ptr(ka), _ptr(va), _ptr(kb), _ptr(vb), n, 1, 0, 32, _ptr(ws), nbytes,
                                              None, _stream()), "sort")
    k, v = (kb, vb) if where == 1 else (ka, va)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k.cpu().numpy().view(np.uint32), keys[order])
    assert np.array_equal(v.cpu().numpy().view(np.uint32), order.astype(np.uint32))


@pytest.mark.parametrize("n,capacity", [(0, 5000), (1, 1024), (1000, 1025), (70_001, 100_000), (70_001, 70_001),
                                        (3_000_000, 4_500_000), (99, 5_000_000)])
def test_radix_sort_with_count_on_the_device(n, capacity):
  """n_dev: the arrays hold `capacity` slots, the element count sits in device memory (the renderer enqueues its tile
  sort before the pair count has reached the host); the first n outputs equal an exact-size sort, the slack is ignored."""
  lib = _lib.load()
  rng = np.random.default_rng(n + capacity)
  keys = rng.integers(0, 2 ** 13, size=capacity, dtype=np.uint64).astype(np.uint32)    # slack holds plausible keys too
  vals2 = rng.integers(0, 2 ** 32, size=capacity, dtype=np.uint64).astype(np.uint32)
  ka, v2a = torch.from_numpy(keys.view(np.int32)).cuda(), torch.from_numpy(vals2.view(np.int32)).cuda()
  va, kb, vb, v2b = (torch.zeros_like(ka) for _ in range(4))
  count = torch.tensor([n], dtype=torch.int32, device="cuda")
  nbytes = lib.gsr_sort_workspace_bytes(capacity)
  ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
  where = _lib.check(lib.gsr_sort_pairs2_u32(_ptr(ka), _ptr(va), _ptr(v2a), _ptr(kb), _ptr(vb), _ptr(v2b), capacity, 1, 0, 13,
                                             _ptr(ws), nbytes, _ptr(count), _stream()), "sort2 counted")
  k, v, v2 = (kb, vb, v2b) if where == 1 else (ka, va, v2a)
  order = np.argsort(keys[:n], kind="stable")
  assert np.array_equal(k.cpu().numpy().view(np.uint32)[:n], keys[:n][order])
  assert np.array_equal(v.cpu().numpy().view(np.uint32)[:n], order.astype(np.uint32))
  assert np.array_equal(v2.cpu().numpy().view(np.uint32)[:n], vals2[:n][order])
  # tile ranges over the counted prefix only
  num_tiles = 1 << 13
  rng_out = torch.zeros(num_tiles, 2, dtype=torch.int32, device="cuda")
  _lib.check(lib.gsr_tile_ranges(_ptr(k), capacity, num_tiles, _ptr(rng_out), _ptr(count), _stream()), "ranges counted")
  got = rng_out.cpu().numpy()
  sk = keys[:n][order]
  want = np.zeros((num_tiles, 2), dtype=np.int32)
  if n:
    starts = np.flatnonzero(np.r_[True, sk[1:] != sk[:-1]])
    ends = np.r_[starts[1:], n]
    want[sk[starts], 0], want[sk[starts], 1] = starts, ends
  assert np.array_equal(got, want)


@pytest.mark.parametrize("n,bits", [(257, 9), (1025, 17), (70_001, 18), (500_000, 27), (4_500_001, 27), (300_000, 26),
                                    (123_457, 25), (5_000_000, 9)])
def test_radix_sort_nine_bit_digits(n, bits):
  """Key widths where 9-bit digits (512 bins) save a pass over 8-bit ones: 9, 17-18 and 25-27 bits (the depth keys of a
  frame span 27 bits for near 0.1 / far 100).  Same stable result as numpy; bits above ``bits`` are ignored."""
  rng = np.random.default_rng(n + bits)
  keys = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
  keys[rng.integers(0, n, size=n // 4)] = keys[0]
  k, v = _sort(keys, 0, bits)
  low = keys & np.uint32((1 << bits) - 1)
  order = np.argsort(low, kind="stable")
  assert np.array_equal(k, keys[order])
  assert np.array_equal(v, order.astype(np.uint32))


def test_depth_keys_relative_to_the_near_plane_sort_like_the_depths():
  """gsr_depth_key_range + gsr_depth_keys: keys start at 0 for the near plane, are monotone in depth, need 27 bits for
  0.1 .. 100, clamp outside the range, and the stable sort over just those bits orders the depths (ties by index)."""
  lib = _lib.load()
  bias, top = C.c_uint32(0), C.c_uint32(0)
  _lib.check(lib.gsr_depth_key_range(0.1, 100.0, C.byref(bias), C.byref(top)), "range")
  assert int(top.value).bit_length() == 27
  n = 400_000
  rng = np.random.default_rng(0)
  depth = np.exp(rng.uniform(np.log(0.1000001), np.log(99.9999), size=n)).astype(np.float32)
  depth[rng.integers(0, n, size=n // 5)] = depth[7]                          # ties
  depth[:4] = [0.05, 1e-3, 150.0, 1e9]                                       # outside [near, far]: clamped, never wrapped
  d = torch.from_numpy(depth).cuda()
  keys = torch.empty(n, dtype=torch.int32, device="cuda")
  _lib.check(lib.gsr_depth_keys(_ptr(d), n, bias.value, top.value, _ptr(keys), _stream()), "keys")
  k = keys.cpu().numpy().view(np.uint32)
  assert k[0] == 0 and k[1] == 0 and k[2] == top.value and k[3] == top.value and k.max() <= top.value
  inside = np.arange(4, n)
  o = np.argsort(depth[inside], kind="stable")
  assert np.all(np.diff(k[inside][o].astype(np.int64)) >= 0)                # monotone
  assert np.array_equal(np.argsort(k[inside], kind="stable"), o)             # exactly the depths' order, ties by index
  sk, sv = _sort(k, 0, 27)
  assert np.array_equal(sv, np.argsort(k, kind="stable").astype(np.uint32))
  # an unbounded range falls back to plain 32-bit keys
  _lib.check(lib.gsr_depth_key_range(0.0, 100.0, C.byref(bias), C.byref(top)), "range")
  assert bias.value == 0 and top.value == 0xFFFFFFFF


@pytest.mark.parametrize("bits", [1, 7, 8, 13, 15, 20])
def test_radix_sort_partial_bits_keeps_input_order(bits):
  n = 500_003
  rng = np.random.default_rng(bits)
  keys = rng.integers(0, 2 ** bits, size=n, dtype=np.uint64).astype(np.uint32)
  vals = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
  k, v = _sort(keys, 0, bits, iota=False, vals=vals)
  order = np.argsort(keys, kind="stable")
  assert np.array_equal(k, keys[order])
  assert np.array_equal(v, vals[order])


def test_sort_rejects_bad_arguments():
  lib = _lib.load()
  t = torch.zeros(16, dtype=torch.int32, device="cuda")
  assert lib.gsr_sort_pairs_u32(_ptr(t), _ptr(t), _ptr(t), _ptr(t), 16, 1, 0, 40, _ptr(t), 64, None, _stream()) == -1
  assert lib.gsr_sort_pairs_u32(_ptr(t), _ptr(t), _ptr(t), _ptr(t), 16, 1, 0, 32, _ptr(t), 8, None, _stream()) == -2


@pytest.mark.parametrize("n,bits", [(1, 13), (1023, 13), (200_003, 13), (5_000_011, 15)])
def test_radix_sort_two_values(n, bits):
  """The tile sort carries (instance id = iota, depth rank) with every tile key."""
  lib = _lib.load()
  rng = np.random.default_rng(n)
  keys = rng.integers(0, 2 ** bits, size=n, dtype=np.uint64).astype(np.uint32)
  vals2 = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
  ka = torch.from_numpy(keys.view(np.int32)).cuda()
  v2a = torch.from_numpy(vals2.view(np.int32)).cuda()
  va, kb, vb, v2b = (torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(4))
  nbytes = lib.gsr_sort_workspace_bytes(n)
  ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
  where = _lib.check(lib.gsr_sort_pairs2_u32(_ptr(ka), _ptr(va), _ptr(v2a), _ptr(kb), _ptr(vb), _ptr(v2b), n, 1, 0, bits,
                                             _ptr(ws), nbytes, None, _stream()), "sort2")
  k, v, v2 = (kb, vb, v2b) if where == 1 else (ka, va, v2a)
  order = np.argsort(keys, kind="stable")
  assert np.array_equal(k.cpu().numpy().view(np.uint32), keys[order])
  assert np.array_equal(v.cpu().numpy().view(np.uint32), order.astype(np.uint32))
  assert np.array_equal(v2.cpu().numpy().view(np.uint32), vals2[order])


@pytest.mark.parametrize("m,visible", [(1, None), (255, None), (4097, None), (70_001, None), (70_001, 50_000),
                                       (1_000_000, 999_999)])
def test_tile_count_with_offsets_equals_count_then_scan(m, visible):
  """K4's count + offsets in two launches (the count pass leaves per-block totals, the scan pass adds them up itself)
  against gsr_tile_count + gsr_exclusive_scan_u32_checked: counts, hit records, offsets and the total, bit for bit."""
  lib = _lib.load()
  gen = torch.Generator().manual_seed(m)
  W, H = 1280, 720
  rows = torch.zeros(m, 16)                                              # packed rows: u v A B | C opacity ...
  sig = 0.5 + 20.0 * torch.rand(m, generator=gen) ** 3                     # a few wide splats among many small ones
  rows[:, 0] = torch.rand(m, generator=gen) * (W + 100) - 50
  rows[:, 1] = torch.rand(m, generator=gen) * (H + 100) - 50
  rows[:, 2] = 1.0 / sig ** 2
  rows[:, 4] = 1.0 / (sig * (0.5 + torch.rand(m, generator=gen))) ** 2
  rows[:, 5] = 0.05 + 0.9 * torch.rand(m, generator=gen)
  rows = rows.cuda()
  order = torch.randperm(m, generator=gen).to(torch.int32).cuda()
  params = _lib.GsrRasterParamsC(1.0 / 255.0, 0.99, 1e-4, 9.0, 0.3, 0, 16, 48.0)
  m_dev = torch.tensor([visible], dtype=torch.int32, device="cuda") if visible is not None else None
  mp = _ptr(m_dev) if m_dev is not None else None
  outs = []
  for fused in (True, False):
    count = torch.full((m,), -1, dtype=torch.int32, device="cuda")
    hits = torch.zeros(m, 4, dtype=torch.int32, device="cuda")
    offsets = torch.full((m,), -1, dtype=torch.int32, device="cuda")
    total = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    if fused:
      nbytes = lib.gsr_tile_count_offsets_workspace_bytes(m)
      ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
      _lib.check(lib.gsr_tile_count_offsets(_ptr(rows), _ptr(order), m, W, H, C.byref(params), _ptr(count), _ptr(hits), mp,
                                            _ptr(offsets), _ptr(total), _ptr(flag), _ptr(ws), nbytes, _stream()),
                 "tile_count_offsets")
    else:
      _lib.check(lib.gsr_tile_count(_ptr(rows), _ptr(order), m, W, H, C.byref(params), _ptr(count), _ptr(hits), mp,
                                    _stream()), "tile_count")
      nbytes = lib.gsr_scan_workspace_bytes(m)
      ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
      _lib.check(lib.gsr_exclusive_scan_u32_checked(_ptr(count), _ptr(offsets), m, _ptr(total), _ptr(flag), _ptr(ws),
                                                    nbytes, _stream()), "scan")
    assert flag.item() == 0
    live = m if visible is None else visible
    outs.append((count.cpu(), offsets.cpu(), total.item(), hits[:live].cpu()))
  (c1, o1, t1, h1), (c2, o2, t2, h2) = outs
  assert torch.equal(c1, c2) and torch.equal(o1, o2) and t1 == t2 and torch.equal(h1, h2)
  assert t1 == int(c1.to(torch.int64).sum()) and (m < 1000 or t1 > m // 2)
