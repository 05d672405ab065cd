// Device-wide primitives for the binning stage: exclusive scan (u32) and a stable LSD radix sort of
// (u32 key, u32 value) pairs over an arbitrary bit range.  Hand-written for wave64; no rocPRIM.
//
// K5 of SURVEY.md §2a ("radix sort by (tile, depth) + tile ranges").  The path sorts twice:
//   (1) the M visible splats by depth bits (32-bit keys, 4 passes)  -> depth order, ties by index (stable)
//   (2) the O tile instances, emitted in depth order, by tile id (ceil(log2 tiles) bits, 2 passes)
// so the big array moves through 2 passes instead of the 6 a fused 48-bit (tile|depth) key would need.
#include "gsr_device.h"
#include "../../include/gsplat_hip.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;   // 4096 elements per block

// block-wide exclusive scan of one value per thread (256 threads = 4 waves); returns exclusive prefix,
// total in *total_out (same for all threads).
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* total_out) {
  __shared__ uint32_t s_wave[4];
  const int lane = gsr_lane(), wave = threadIdx.x >> 6;
  uint32_t incl = gsr_wave_scan_incl_u32(v);
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    uint32_t c = s_wave[w];
    if (w < wave) base += c;
    total += c;
  }
  __syncthreads();
  *total_out = total;
  return base + incl - v;
}

// Overflow guard of the u32 scans: every kernel that forms a sum also forms it in float (range, not precision) and
// raises *overflow when prefix + total reaches 2^31 -- sums the 32-bit index arithmetic downstream cannot hold anyway.
// A u32 wrap is always preceded by such a crossing, at whichever level of the scan it happens.
__device__ __forceinline__ void flag_if_large(float mine, uint32_t prefix, uint32_t* overflow) {
  __shared__ float s_f[4];
  const float w = gsr_wave_sum(mine);
  if (gsr_lane() == 0) s_f[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0 && (float)prefix + ((s_f[0] + s_f[1]) + (s_f[2] + s_f[3])) >= 2147000000.f) *overflow = 1u;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const uint32_t* __restrict__ in, uint32_t n,
                                                                   uint32_t* __restrict__ block_sums,
                                                                   uint32_t* __restrict__ overflow) {
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t sum = 0;
  float fsum = 0.f;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    uint32_t idx = base + i;
    const uint32_t v = (idx < n) ? in[idx] : 0u;
    sum += v;
    fsum += (float)v;
  }
  uint32_t total;
  block_scan_excl(sum, &total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
  if (overflow) flag_if_large(fsum, 0u, overflow);
}

// out[i] = block_offsets[b] + exclusive prefix inside the block.  block_offsets may be null (single block).
// SELF: block_offsets holds the blocks' RAW sums (scan_reduce's output) and every block adds up its predecessors'
// itself -- for up to a few thousand blocks that is cheaper than a launch that scans the sums in between.
// The grand total (sum of everything) is written to *total_out by the last block when total_out != null.
template <bool SELF>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(const uint32_t* in, uint32_t n,   // in may alias out
                                                                  const uint32_t* __restrict__ block_offsets,
                                                                  uint32_t* out, uint32_t* __restrict__ total_out,
                                                                  uint32_t* __restrict__ overflow) {
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t sum = 0;
  float fsum = 0.f;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    uint32_t idx = base + i;
    v[i] = (idx < n) ? in[idx] : 0u;
    sum += v[i];
    fsum += (float)v[i];
  }
  uint32_t total;
  uint32_t excl = block_scan_excl(sum, &total);
  uint32_t boff = 0u;
  if (SELF) {
    uint32_t before = 0;
    float fbefore = 0.f;                                     // the same sum in float: range check only (see flag_if_large)
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += SCAN_THREADS) {
      const uint32_t b = block_offsets[j];
      before += b;
      fbefore += (float)b;
    }
    block_scan_excl(before, &boff);                          // boff = total over the block = sum of all earlier blocks
    fsum += fbefore;                                         // flag_if_large sums fsum over the block
  } else if (block_offsets) {
    boff = block_offsets[blockIdx.x];
  }
  if (overflow) flag_if_large(fsum, SELF ? 0u : boff, overflow);
  uint32_t run = boff + excl;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    uint32_t idx = base + i;
    if (idx < n) out[idx] = run;
    run += v[i];
  }
  if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = boff + total;
}

constexpr uint32_t SCAN_SELF_PREFIX_MAX_BLOCKS = 4096;       // 16 M elements: 16 KB of sums read per block at most

size_t scan_ws_bytes(uint64_t n) {
  size_t bytes = 0;
  while (n > SCAN_TILE) {
    uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    bytes += ((nb * sizeof(uint32_t) + 255) / 256) * 256;
    n = nb;
  }
  return bytes + 256;
}

int scan_impl(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* total_dev, uint8_t* ws, hipStream_t stream,
              uint32_t* overflow = nullptr) {
  if (n == 0) {
    if (total_dev) hipMemsetAsync(total_dev, 0, sizeof(uint32_t), stream);
    return GSR_OK;
  }
  uint32_t nb = (uint32_t)((n + SCAN_TILE - 1) / SCAN_TILE);
  if (nb == 1) {
    scan_apply_kernel<false><<<1, SCAN_THREADS, 0, stream>>>(in, (uint32_t)n, nullptr, out, total_dev, overflow);
    GSR_CHECK_LAUNCH();
    return GSR_OK;
  }
  uint32_t* block_sums = reinterpret_cast<uint32_t*>(ws);
  size_t used = (((size_t)nb * sizeof(uint32_t) + 255) / 256) * 256;
  scan_reduce_kernel<<<nb, SCAN_THREADS, 0, stream>>>(in, (uint32_t)n, block_sums, overflow);
  GSR_CHECK_LAUNCH();
  if (nb <= SCAN_SELF_PREFIX_MAX_BLOCKS) {
    scan_apply_kernel<true><<<nb, SCAN_THREADS, 0, stream>>>(in, (uint32_t)n, block_sums, out, total_dev, overflow);
    GSR_CHECK_LAUNCH();
    return GSR_OK;
  }
  int rc = scan_impl(block_sums, block_sums, nb, nullptr, ws + used, stream, overflow);   // in place
  if (rc != GSR_OK) return rc;
  scan_apply_kernel<false><<<nb, SCAN_THREADS, 0, stream>>>(in, (uint32_t)n, block_sums, out, total_dev, overflow);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

// ------------------------------------------------------------------------------------------ radix sort
// Stable LSD radix sort, three launches per pass:
//   rs_hist        per-block digit histogram (LDS atomics) -> hist[digit][block]
//   rs_digit_scan  one block per digit: exclusive scan of the digit's row over the blocks, and the digit's total
//   rs_scatter     digit bases from the totals (block scan), per-wave digit counts of the block's four quarters, then
//                  ballot-match ranking inside each wave with wave-private running slots (no block barrier per round);
//                  elements keep their input order inside every digit bucket (stable).  Up to two value arrays.
// Digits are 8 bits wide, or 9 (512 bins, two per thread) when that saves a pass: the depth keys of a frame span
// bits(far) - bits(near), 27 bits for near 0.1 / far 100 -- three 9-bit passes instead of four 8-bit ones.
constexpr int RS_THREADS = 256;
constexpr int RS_MAX_BINS = 512;

// 4096-element tiles from 1 M elements on (measured with the wave-private scatter, 3 M keys: scatter 38 -> 25 us per pass,
// histogram and digit scan shrink with the block count; at 500 k the small tiles win: 10 against 15 us)
#ifndef GSR_RS_BIG_MIN
#define GSR_RS_BIG_MIN (1ll << 20)
#endif
inline int rs_rounds_for(int64_t n) { return n < GSR_RS_BIG_MIN ? 4 : 16; }     // 1024 or 4096 pairs per block
inline int rs_digit_bits(int bits) { return (bits + 8) / 9 < (bits + 7) / 8 ? 9 : 8; }

template <int BITS>
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, int shift,
                                                             uint32_t mask, int rounds, uint32_t* __restrict__ hist,
                                                             uint32_t num_blocks, const uint32_t* __restrict__ n_dev) {
  constexpr int BINS = 1 << BITS, PER = BINS / RS_THREADS;
  __shared__ uint32_t s_hist[BINS];
  if (n_dev) n = min(n, *n_dev);              // n is a capacity: the element count is still on the device
#pragma unroll
  for (int k = 0; k < PER; ++k) s_hist[threadIdx.x + k * RS_THREADS] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * (uint32_t)(rounds * RS_THREADS);
  for (int r = 0; r < rounds; ++r) {
    uint32_t idx = base + r * RS_THREADS + threadIdx.x;
    if (idx < n) atomicAdd(&s_hist[(keys[idx] >> shift) & mask], 1u);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const uint32_t d = threadIdx.x + k * RS_THREADS;
    hist[(size_t)d * num_blocks + blockIdx.x] = s_hist[d];
  }
}

// one block per digit: exclusive scan of the digit's row over the blocks (offsets RELATIVE to the digit's first element)
// and the digit's total; the scatter kernel turns the totals into the digits' bases itself (a 256/512-wide block scan)
__global__ __launch_bounds__(RS_THREADS) void rs_digit_scan_kernel(const uint32_t* __restrict__ hist,
                                                                   uint32_t* __restrict__ offs, uint32_t num_blocks,
                                                                   uint32_t* __restrict__ digit_total) {
  __shared__ uint32_t s_wave[4];
  __shared__ uint32_t s_carry;
  const uint32_t d = blockIdx.x;
  const int lane = gsr_lane(), wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_carry = 0u;
  __syncthreads();
  const uint32_t* row = hist + (size_t)d * num_blocks;
  uint32_t* orow = offs + (size_t)d * num_blocks;
  for (uint32_t b0 = 0; b0 < num_blocks; b0 += RS_THREADS) {
    const uint32_t b = b0 + threadIdx.x;
    const uint32_t v = (b < num_blocks) ? row[b] : 0u;
    const uint32_t incl = gsr_wave_scan_incl_u32(v);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t c = s_wave[w];
      if (w < wave) wbase += c;
      total += c;
    }
    if (b < num_blocks) orow[b] = s_carry + wbase + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) s_carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) digit_total[d] = s_carry;
}

// Stable scatter with LDS staging.  The block's tile of ROUNDS * 256 elements is split into four contiguous quarters, one
// per wave; every thread keeps its ROUNDS elements in registers.
//   1. per-wave digit counts of the whole quarter (LDS atomics: only the counts matter);
//   2. one pass over the digits turns them into each wave's first slot per digit in the block's LDS image (digit-major,
//      wave-major inside a digit = input order) and the digits' global slots (block scans of the block's counts and of
//      the digit totals);
//   3. ranking, 64 elements per wave and round WITHOUT block barriers: lanes holding the same digit find each other by
//      ballot matching, the lowest takes the digit's running slot of its wave and advances it by the group's size;
//   4. the image is written out in image order: consecutive threads write consecutive addresses inside each digit
//      run, so the global stores are coalesced segments instead of 4-byte singles.
// Elements keep their input order inside every digit bucket (stable).  Up to two value arrays.
// (Round 2's form ranked 256 elements per round across the four waves: three block barriers and ~26 LDS operations per
// element, most of them on bins the round never touched; this one has four barriers per BLOCK and ~7 LDS operations per
// element.)
// GSR_RS_NEXT_HIST (experiment): price of forming the NEXT pass's block histograms while this pass
// scatters -- one integer atomic per element on hist'[next digit][destination block] (a block's elements go to ~BINS
// destination runs of ~TILE / BINS elements each, so (destination block, next digit) pairs hardly repeat inside a block:
// nothing to pre-aggregate).  The adds go to a dummy table; the sort's results are untouched.  See profiles/r04_sorts.txt.
#ifndef GSR_RS_NEXT_HIST
#define GSR_RS_NEXT_HIST 0
#endif
#if GSR_RS_NEXT_HIST
__device__ uint32_t g_rs_dummy_hist[512 * 4096];
#endif

template <bool TWO, int ROUNDS, int BITS>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                const uint32_t* __restrict__ vals_in,   // null -> iota
                                                                const uint32_t* __restrict__ vals2_in,
                                                                uint32_t* __restrict__ keys_out,
                                                                uint32_t* __restrict__ vals_out,
                                                                uint32_t* __restrict__ vals2_out, uint32_t n, int shift,
                                                                uint32_t mask, const uint32_t* __restrict__ counts,
                                                                const uint32_t* __restrict__ offsets,
                                                                const uint32_t* __restrict__ digit_total,
                                                                uint32_t num_blocks, const uint32_t* __restrict__ n_dev) {
  constexpr int TILE = ROUNDS * RS_THREADS, QUARTER = ROUNDS * 64;
  constexpr int BINS = 1 << BITS, PER = BINS / RS_THREADS;     // thread t owns the consecutive digits PER t .. PER t + PER - 1
  if (n_dev) n = min(n, *n_dev);
  if (blockIdx.x * (uint32_t)TILE >= n) return;   // uniform over the block; its histogram row is all zeros
  __shared__ uint32_t s_start[BINS];          // first image slot of each digit
  __shared__ uint32_t s_goff[BINS];           // global slot of the digit's first element of this block
  __shared__ uint32_t s_slot[4][BINS];        // per wave: digit counts of its quarter, then its running image slot per digit
  __shared__ uint32_t s_wave[4], s_wave_t[4];
  __shared__ uint32_t s_key[TILE];
  __shared__ uint32_t s_val[TILE];
  __shared__ uint32_t s_val2[TWO ? TILE : 1];
  const int lane = gsr_lane(), wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < 4; ++w)
#pragma unroll
    for (int k = 0; k < PER; ++k) s_slot[w][threadIdx.x + k * RS_THREADS] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * (uint32_t)TILE;
  const uint32_t block_n = min((uint32_t)TILE, n - base);
  const uint32_t mine = base + (uint32_t)(wave * QUARTER + lane);   // round r: element mine + 64 r
  uint32_t key[ROUNDS], val[ROUNDS], val2[TWO ? ROUNDS : 1];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint32_t idx = mine + 64u * r;
    key[r] = 0u; val[r] = 0u;
    if (TWO) val2[r] = 0u;
    if (idx < n) {
      key[r] = keys_in[idx];
      val[r] = vals_in ? vals_in[idx] : idx;
      if (TWO) val2[r] = vals2_in[idx];
      atomicAdd(&s_slot[wave][(key[r] >> shift) & mask], 1u);
    }
  }
  __syncthreads();
  {
    // block-wide exclusive scans over the digits: this block's counts -> first image slot of each digit (and of each
    // wave inside it); the digits' totals -> global base of each digit (+ its elements in earlier blocks = global slot)
    uint32_t c[PER][4], t[PER], here = 0, here_t = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const uint32_t d = PER * threadIdx.x + k;
#pragma unroll
      for (int w = 0; w < 4; ++w) { c[k][w] = s_slot[w][d]; here += c[k][w]; }
      t[k] = digit_total[d];
      here_t += t[k];
    }
    const uint32_t incl = gsr_wave_scan_incl_u32(here), incl_t = gsr_wave_scan_incl_u32(here_t);
    if (lane == 63) { s_wave[wave] = incl; s_wave_t[wave] = incl_t; }
    __syncthreads();
    uint32_t at = incl - here, gbase = incl_t - here_t;
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if (w < wave) { at += s_wave[w]; gbase += s_wave_t[w]; }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const uint32_t d = PER * threadIdx.x + k;
      s_start[d] = at;
      s_goff[d] = gbase + offsets[(size_t)d * num_blocks + blockIdx.x];
#pragma unroll
      for (int w = 0; w < 4; ++w) { s_slot[w][d] = at; at += c[k][w]; }
      gbase += t[k];
    }
  }
  __syncthreads();
  uint32_t* my_slots = s_slot[wave];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const bool valid = mine + 64u * r < n;
    const uint32_t digit = (key[r] >> shift) & mask;
    uint64_t peers = __ballot(valid);                          // lanes of this wave holding the same digit
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
      const uint64_t bit = __ballot((digit >> b) & 1u);
      peers &= ((digit >> b) & 1u) ? bit : ~bit;
    }
    const int rank = gsr_mbcnt(peers);                       // peers below me
    const uint32_t first = valid ? my_slots[digit] : 0u;     // the group reads its digit's running slot ...
    gsr_wave_lds_fence();
    if (valid && rank == 0) my_slots[digit] = first + (uint32_t)__popcll(peers);   // ... and its lowest lane advances it
    gsr_wave_lds_fence();
    if (valid) {
      const uint32_t slot = first + (uint32_t)rank;
      s_key[slot] = key[r];
      s_val[slot] = val[r];
      if (TWO) s_val2[slot] = val2[r];
    }
  }
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < block_n; e += RS_THREADS) {
    const uint32_t k = s_key[e];
    const uint32_t d = (k >> shift) & mask;
    const uint32_t dst = s_goff[d] + (e - s_start[d]);
    keys_out[dst] = k;
    vals_out[dst] = s_val[e];
    if (TWO) vals2_out[dst] = s_val2[e];
#if GSR_RS_NEXT_HIST
    atomicAdd(&g_rs_dummy_hist[(((k >> (shift + BITS)) & (BINS - 1)) * 4096u + min(dst / (uint32_t)TILE, 4095u))], 1u);
#endif
  }
}

size_t sort_ws_bytes(int64_t n) {
  if (n <= 0) return 2048 * 5 + 256;
  const uint64_t tile = (uint64_t)rs_rounds_for(n) * RS_THREADS;
  const uint64_t nb = ((uint64_t)n + tile - 1) / tile;
  const size_t hist = ((nb * RS_MAX_BINS * sizeof(uint32_t) + 255) / 256) * 256;
  return 2 * hist + 5 * RS_MAX_BINS * sizeof(uint32_t) + 256;    // counts + offsets tables + digit totals
}

template <bool TWO, int ROUNDS, int BITS>
void rs_launch_scatter(uint32_t nb, hipStream_t stream, const uint32_t* kin, const uint32_t* vin, const uint32_t* v2in,
                       uint32_t* kout, uint32_t* vout, uint32_t* v2out, uint32_t n, int bit, uint32_t mask,
                       const uint32_t* hist, const uint32_t* offs, const uint32_t* totals, const uint32_t* n_dev) {
  rs_scatter_kernel<TWO, ROUNDS, BITS><<<nb, RS_THREADS, 0, stream>>>(kin, vin, v2in, kout, vout, v2out, n, bit, mask, hist,
                                                                      offs, totals, nb, n_dev);
}

template <int BITS>
int sort_passes(uint32_t* keys_a, uint32_t* vals_a, uint32_t* vals2_a, uint32_t* keys_b, uint32_t* vals_b,
                uint32_t* vals2_b, int64_t n, int vals_are_iota, int begin_bit, int end_bit, void* workspace,
                const uint32_t* n_dev, hipStream_t stream) {
  constexpr int BINS = 1 << BITS;
  const bool two = vals2_a != nullptr;
  const int rounds = rs_rounds_for(n);
  const uint32_t tile = (uint32_t)(rounds * RS_THREADS);
  const uint32_t nb = (uint32_t)((n + tile - 1) / tile);
  uint32_t* hist = reinterpret_cast<uint32_t*>(workspace);
  const size_t hist_bytes = (((size_t)nb * RS_MAX_BINS * sizeof(uint32_t) + 255) / 256) * 256;
  uint32_t* offs = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(workspace) + hist_bytes);
  uint32_t* totals = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(workspace) + 2 * hist_bytes);
  int passes = (end_bit - begin_bit + BITS - 1) / BITS;
  if (passes < 1) passes = 1;
  if (passes > 5) return GSR_ERR_INVALID_ARGUMENT;

  uint32_t* kin = keys_a; uint32_t* kout = keys_b;
  const uint32_t* vin = vals_are_iota ? nullptr : vals_a; uint32_t* vout = vals_b;
  const uint32_t* v2in = vals2_a; uint32_t* v2out = vals2_b;
  int where = 0;
  for (int p = 0; p < passes; ++p) {
    const int bit = begin_bit + BITS * p;
    int bits = end_bit - bit; if (bits > BITS) bits = BITS; if (bits < 0) bits = 0;
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t* tot = totals + (size_t)p * RS_MAX_BINS;
    rs_hist_kernel<BITS><<<nb, RS_THREADS, 0, stream>>>(kin, (uint32_t)n, bit, mask, rounds, hist, nb, n_dev);
    GSR_CHECK_LAUNCH();
    rs_digit_scan_kernel<<<BINS, RS_THREADS, 0, stream>>>(hist, offs, nb, tot);
    GSR_CHECK_LAUNCH();
    if (two) {
      if (rounds == 4) rs_launch_scatter<true, 4, BITS>(nb, stream, kin, vin, v2in, kout, vout, v2out, (uint32_t)n, bit, mask, hist, offs, tot, n_dev);
      else rs_launch_scatter<true, 16, BITS>(nb, stream, kin, vin, v2in, kout, vout, v2out, (uint32_t)n, bit, mask, hist, offs, tot, n_dev);
    } else {
      if (rounds == 4) rs_launch_scatter<false, 4, BITS>(nb, stream, kin, vin, v2in, kout, vout, v2out, (uint32_t)n, bit, mask, hist, offs, tot, n_dev);
      else rs_launch_scatter<false, 16, BITS>(nb, stream, kin, vin, v2in, kout, vout, v2out, (uint32_t)n, bit, mask, hist, offs, tot, n_dev);
    }
    GSR_CHECK_LAUNCH();
    where ^= 1;
    if (where == 1) { kin = keys_b; vin = vals_b; v2in = vals2_b; kout = keys_a; vout = vals_a; v2out = vals2_a; }
    else            { kin = keys_a; vin = vals_a; v2in = vals2_a; kout = keys_b; vout = vals_b; v2out = vals2_b; }
  }
  return where;
}

int sort_impl(uint32_t* keys_a, uint32_t* vals_a, uint32_t* vals2_a, uint32_t* keys_b, uint32_t* vals_b,
              uint32_t* vals2_b, int64_t n, int vals_are_iota, int begin_bit, int end_bit, void* workspace,
              size_t workspace_bytes, const uint32_t* n_dev, hipStream_t stream) {
  const bool two = vals2_a != nullptr;
  if (n < 0 || n > 0x7FFFFFFFll || begin_bit < 0 || end_bit > 32 || begin_bit > end_bit) return GSR_ERR_INVALID_ARGUMENT;
  if (n > 0 && (!keys_a || !vals_a || !keys_b || !vals_b || (two && !vals2_b))) return GSR_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < sort_ws_bytes(n) || !workspace) return GSR_ERR_WORKSPACE_TOO_SMALL;
  if (n == 0) return 0;
  if (rs_digit_bits(end_bit - begin_bit) == 9)
    return sort_passes<9>(keys_a, vals_a, vals2_a, keys_b, vals_b, vals2_b, n, vals_are_iota, begin_bit, end_bit, workspace,
                          n_dev, stream);
  return sort_passes<8>(keys_a, vals_a, vals2_a, keys_b, vals_b, vals2_b, n, vals_are_iota, begin_bit, end_bit, workspace,
                        n_dev, stream);
}

}  // namespace

extern "C" {

size_t gsr_scan_workspace_bytes(int64_t n) { return scan_ws_bytes((uint64_t)(n < 0 ? 0 : n)); }

int gsr_exclusive_scan_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total_dev, void* workspace,
                           size_t workspace_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n < 0 || n > 0xFFFFFFFFll) return GSR_ERR_INVALID_ARGUMENT;
  if (n > 0 && (!in || !out)) return GSR_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < scan_ws_bytes((uint64_t)n) || (n > SCAN_TILE && !workspace)) return GSR_ERR_WORKSPACE_TOO_SMALL;
  return scan_impl(in, out, (uint64_t)n, total_dev, reinterpret_cast<uint8_t*>(workspace), stream);
}

int gsr_exclusive_scan_u32_checked(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total_dev,
                                   uint32_t* overflow_dev, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n < 0 || n > 0xFFFFFFFFll || !overflow_dev) return GSR_ERR_INVALID_ARGUMENT;
  if (n > 0 && (!in || !out)) return GSR_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < scan_ws_bytes((uint64_t)n) || (n > SCAN_TILE && !workspace)) return GSR_ERR_WORKSPACE_TOO_SMALL;
  return scan_impl(in, out, (uint64_t)n, total_dev, reinterpret_cast<uint8_t*>(workspace), stream, overflow_dev);
}

size_t gsr_sort_workspace_bytes(int64_t n) { return sort_ws_bytes(n); }

// Sorts by key bits [begin_bit, end_bit).  Ping-pongs between the *_a and *_b buffers; input is in the *_a buffers
// (vals_are_iota != 0: the input values are 0..n-1 and vals_a is only scratch).  Returns (>= 0) 0 when the result
// is in the *_a buffers and 1 when it is in the *_b buffers, or a negative error code.
int gsr_sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, int64_t n,
                       int vals_are_iota, int begin_bit, int end_bit, void* workspace, size_t workspace_bytes,
                       const uint32_t* n_dev, void* stream_) {
  return sort_impl(keys_a, vals_a, nullptr, keys_b, vals_b, nullptr, n, vals_are_iota, begin_bit, end_bit, workspace,
                   workspace_bytes, n_dev, reinterpret_cast<hipStream_t>(stream_));
}

// Same, carrying a second value array (vals2) along with every key.
int gsr_sort_pairs2_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* vals2_a, uint32_t* keys_b, uint32_t* vals_b,
                        uint32_t* vals2_b, int64_t n, int vals_are_iota, int begin_bit, int end_bit, void* workspace,
                        size_t workspace_bytes, const uint32_t* n_dev, void* stream_) {
  if (n > 0 && !vals2_a) return GSR_ERR_INVALID_ARGUMENT;
  return sort_impl(keys_a, vals_a, vals2_a, keys_b, vals_b, vals2_b, n, vals_are_iota, begin_bit, end_bit, workspace,
                   workspace_bytes, n_dev, reinterpret_cast<hipStream_t>(stream_));
}

}  // extern "C"
