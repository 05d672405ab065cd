// Microbenchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU/LDS instruction forms K6/K7 are made
// of, by waves per SIMD, on gfx950.  Cycles come from s_memtime inside the kernel (shader clock), the wall time from
// HIP events, so the effective clock is reported too.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(S0, S1, S2, S3, S4, S5, S6, S7) S0 "\n" S1 "\n" S2 "\n" S3 "\n" S4 "\n" S5 "\n" S6 "\n" S7 "\n"

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, float a, float b, int ITERS) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
  v2f va = {a, a}, vb = {b, b};
  v2f sa = {a, b};
  __shared__ float lds[64 * 16];
  lds[threadIdx.x] = x0;
  const unsigned la = threadIdx.x * 4;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0) {          // v_fma_f32, 3 VGPR sources
      asm volatile(REP8("v_fma_f32 %0, %0, %8, %9", "v_fma_f32 %1, %1, %8, %9", "v_fma_f32 %2, %2, %8, %9", "v_fma_f32 %3, %3, %8, %9",
                        "v_fma_f32 %4, %4, %8, %9", "v_fma_f32 %5, %5, %8, %9", "v_fma_f32 %6, %6, %8, %9", "v_fma_f32 %7, %7, %8, %9")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == 1) {   // v_pk_fma_f32, 3 VGPR-pair sources (x = x*a + b)
      asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9", "v_pk_fma_f32 %1, %1, %8, %9", "v_pk_fma_f32 %2, %2, %8, %9", "v_pk_fma_f32 %3, %3, %8, %9",
                        "v_pk_fma_f32 %4, %4, %8, %9", "v_pk_fma_f32 %5, %5, %8, %9", "v_pk_fma_f32 %6, %6, %8, %9", "v_pk_fma_f32 %7, %7, %8, %9")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(va), "v"(vb));
    } else if (MODE == 2) {   // v_mul_f32 VOP2, 2 VGPR sources
      asm volatile(REP8("v_mul_f32 %0, %0, %8", "v_mul_f32 %1, %1, %8", "v_mul_f32 %2, %2, %8", "v_mul_f32 %3, %3, %8",
                        "v_mul_f32 %4, %4, %8", "v_mul_f32 %5, %5, %8", "v_mul_f32 %6, %6, %8", "v_mul_f32 %7, %7, %8")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    } else if (MODE == 3) {   // v_pk_mul_f32, 2 VGPR-pair sources
      asm volatile(REP8("v_pk_mul_f32 %0, %0, %8", "v_pk_mul_f32 %1, %1, %8", "v_pk_mul_f32 %2, %2, %8", "v_pk_mul_f32 %3, %3, %8",
                        "v_pk_mul_f32 %4, %4, %8", "v_pk_mul_f32 %5, %5, %8", "v_pk_mul_f32 %6, %6, %8", "v_pk_mul_f32 %7, %7, %8")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(va));
    } else if (MODE == 4) {   // v_pk_mul_f32 with an SGPR-pair source
      asm volatile(REP8("v_pk_mul_f32 %0, %0, %8", "v_pk_mul_f32 %1, %1, %8", "v_pk_mul_f32 %2, %2, %8", "v_pk_mul_f32 %3, %3, %8",
                        "v_pk_mul_f32 %4, %4, %8", "v_pk_mul_f32 %5, %5, %8", "v_pk_mul_f32 %6, %6, %8", "v_pk_mul_f32 %7, %7, %8")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(sa));
    } else if (MODE == 5) {   // v_pk_fma_f32 accumulate form x = a*b + x with one SGPR-pair source
      asm volatile(REP8("v_pk_fma_f32 %0, %8, %9, %0", "v_pk_fma_f32 %1, %8, %9, %1", "v_pk_fma_f32 %2, %8, %9, %2", "v_pk_fma_f32 %3, %8, %9, %3",
                        "v_pk_fma_f32 %4, %8, %9, %4", "v_pk_fma_f32 %5, %8, %9, %5", "v_pk_fma_f32 %6, %8, %9, %6", "v_pk_fma_f32 %7, %8, %9, %7")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(va), "s"(sa));
    } else if (MODE == 6) {   // v_pk_add_f32
      asm volatile(REP8("v_pk_add_f32 %0, %0, %8", "v_pk_add_f32 %1, %1, %8", "v_pk_add_f32 %2, %2, %8", "v_pk_add_f32 %3, %3, %8",
                        "v_pk_add_f32 %4, %4, %8", "v_pk_add_f32 %5, %5, %8", "v_pk_add_f32 %6, %6, %8", "v_pk_add_f32 %7, %7, %8")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(va));
    } else if (MODE == 7) {   // v_exp_f32
      asm volatile(REP8("v_exp_f32 %0, %0", "v_exp_f32 %1, %1", "v_exp_f32 %2, %2", "v_exp_f32 %3, %3", "v_exp_f32 %4, %4", "v_exp_f32 %5, %5",
                        "v_exp_f32 %6, %6", "v_exp_f32 %7, %7")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (MODE == 8) {   // v_cmp_ge_f32 into an SGPR pair + v_cndmask on it
      asm volatile("v_cmp_ge_f32 s[20:21], %0, %8\nv_cndmask_b32 %1, 0, %1, s[20:21]\nv_cmp_ge_f32 s[22:23], %2, %8\nv_cndmask_b32 %3, 0, %3, s[22:23]\n"
                   "v_cmp_ge_f32 s[24:25], %4, %8\nv_cndmask_b32 %5, 0, %5, s[24:25]\nv_cmp_ge_f32 s[26:27], %6, %8\nv_cndmask_b32 %7, 0, %7, s[26:27]\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a)
                   : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    } else if (MODE == 9) {   // v_med3_f32
      asm volatile(REP8("v_med3_f32 %0, %0, %8, 0", "v_med3_f32 %1, %1, %8, 0", "v_med3_f32 %2, %2, %8, 0", "v_med3_f32 %3, %3, %8, 0",
                        "v_med3_f32 %4, %4, %8, 0", "v_med3_f32 %5, %5, %8, 0", "v_med3_f32 %6, %6, %8, 0", "v_med3_f32 %7, %7, %8, 0")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a));
    } else if (MODE == 10) {  // ds_write_b32 x8 (one wave-row each), then wait
      asm volatile("ds_write_b32 %8, %0\nds_write_b32 %8, %1 offset:256\nds_write_b32 %8, %2 offset:512\nds_write_b32 %8, %3 offset:768\n"
                   "ds_write_b32 %8, %4 offset:1024\nds_write_b32 %8, %5 offset:1280\nds_write_b32 %8, %6 offset:1536\nds_write_b32 %8, %7 offset:1792\n"
                   "s_waitcnt lgkmcnt(0)\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(la) : "memory");
    } else if (MODE == 11) {  // v_fma_f32 with dependent chain length 4 (two interleaved chains)
      asm volatile(REP8("v_fma_f32 %0, %0, %8, %9", "v_fma_f32 %1, %1, %8, %9", "v_fma_f32 %0, %0, %8, %9", "v_fma_f32 %1, %1, %8, %9",
                        "v_fma_f32 %0, %0, %8, %9", "v_fma_f32 %1, %1, %8, %9", "v_fma_f32 %0, %0, %8, %9", "v_fma_f32 %1, %1, %8, %9")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == 12) {  // v_rcp_f32
      asm volatile(REP8("v_rcp_f32 %0, %0", "v_rcp_f32 %1, %1", "v_rcp_f32 %2, %2", "v_rcp_f32 %3, %3", "v_rcp_f32 %4, %4", "v_rcp_f32 %5, %5",
                        "v_rcp_f32 %6, %6", "v_rcp_f32 %7, %7")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (MODE == 14) {  // v_fma_f32 with only lanes 0-31 enabled: does the SIMD skip the inactive half-wave?
      asm volatile("s_mov_b64 s[20:21], exec\ns_mov_b64 exec, 0xffffffff\n"
                   "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                   "s_mov_b64 exec, s[20:21]\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "s20", "s21");
    } else if (MODE == 15) {  // v_pk_fma_f32 with only lanes 0-31 enabled
      asm volatile("s_mov_b64 s[20:21], exec\ns_mov_b64 exec, 0xffffffff\n"
                   "v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9\n"
                   "s_mov_b64 exec, s[20:21]\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(va), "v"(vb) : "s20", "s21");
    } else if (MODE == 16) {  // v_fma_f32 with lanes 0-15 only
      asm volatile("s_mov_b64 s[20:21], exec\ns_mov_b64 exec, 0xffff\n"
                   "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                   "s_mov_b64 exec, s[20:21]\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "s20", "s21");
    } else if (MODE == 13) {  // mix of 4 v_pk_fma (sgpr src) + 4 v_fma: does alternating help?
      asm volatile(REP8("v_pk_fma_f32 %0, %8, %9, %0", "v_fma_f32 %4, %4, %10, %10", "v_pk_fma_f32 %1, %8, %9, %1", "v_fma_f32 %5, %5, %10, %10",
                        "v_pk_fma_f32 %2, %8, %9, %2", "v_fma_f32 %6, %6, %10, %10", "v_pk_fma_f32 %3, %8, %9, %3", "v_fma_f32 %7, %7, %10, %10")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(va), "s"(sa), "v"(a));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + lds[(threadIdx.x * 7) & 1023];
}

__global__ void empty_kernel(float* out) { if (out == nullptr) out[0] = 0.f; }

static float time_launch(void (*fn)(int, int, float*, unsigned long long*), int blocks, int iters, float* out,
                         unsigned long long* cyc) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  fn(blocks, iters, out, cyc); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0); fn(blocks, iters, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
  }
  return best;
}

template <int MODE> void launch(int blocks, int iters, float* out, unsigned long long* cyc) {
  k<MODE><<<blocks, 64>>>(out, cyc, 1.0001f, 0.5f, iters);
}

// issue cost = slope of the kernel time in the loop count (removes launch + workgroup-dispatch ramp, which is not small
// for thousands of one-wave workgroups); "ramp" = the intercept
template <int MODE> void run(const char* name, float* out, unsigned long long* cyc) {
  for (int wps : {1, 2, 4, 6, 8}) {
    int blocks = 256 * 4 * wps;
    const int n1 = 2048, n2 = 8192;
    float t1 = time_launch(launch<MODE>, blocks, n1, out, cyc), t2 = time_launch(launch<MODE>, blocks, n2, out, cyc);
    static unsigned long long h[256 * 4 * 8];
    hipMemcpy(h, cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
    double ns_per_instr = (double)(t2 - t1) * 1e6 / ((double)(n2 - n1) * 8 * wps);
    double ramp_us = (t1 - (double)(t2 - t1) * n1 / (n2 - n1)) * 1e3;
    double clock_ghz = avg / (t2 * 1e6 - ramp_us * 1e3);      // ticks of one wave over the loop part of the kernel time
    printf("%-26s waves/SIMD=%d  slope ns/instr/SIMD=%.3f  = %.2f cyc@2.4GHz   intercept %.1f us   ticks/wave(n2)=%.0f (%.2f ticks/ns)\n",
           name, wps, ns_per_instr, ns_per_instr * 2.4, ramp_us, avg, clock_ghz);
  }
}
int main(int argc, char** argv) {
  float* out; hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
  unsigned long long* cyc; hipMalloc(&cyc, 256 * 4 * 8 * 8);
  if (argc > 1) {   // quick check: half-wave EXEC masks
    run<0>("v_fma_f32 vvv", out, cyc); run<14>("v_fma_f32 exec lo32", out, cyc); run<16>("v_fma_f32 exec lo16", out, cyc);
    run<1>("v_pk_fma_f32 vvv", out, cyc); run<15>("v_pk_fma_f32 exec lo32", out, cyc);
    return 0;
  }
  run<0>("v_fma_f32 vvv", out, cyc); run<2>("v_mul_f32 vop2", out, cyc); run<1>("v_pk_fma_f32 vvv", out, cyc);
  run<5>("v_pk_fma_f32 v,s,acc", out, cyc); run<3>("v_pk_mul_f32 vv", out, cyc); run<4>("v_pk_mul_f32 v,s", out, cyc);
  run<6>("v_pk_add_f32 vv", out, cyc); run<7>("v_exp_f32", out, cyc); run<12>("v_rcp_f32", out, cyc);
  run<8>("v_cmp+v_cndmask sgpr", out, cyc); run<9>("v_med3_f32", out, cyc); run<11>("v_fma dep-chain/2", out, cyc);
  run<13>("pk_fma/fma alternating", out, cyc); run<10>("ds_write_b32 x8 + wait", out, cyc);
  // workgroup dispatch: the same 8192 waves as one-wave and as four-wave workgroups (empty kernel)
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int tpb : {64, 128, 256}) for (int waves : {1024, 8192, 32768}) {
    int blocks = waves * 64 / tpb;
    empty_kernel<<<blocks, tpb>>>(out); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) empty_kernel<<<blocks, tpb>>>(out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel: %6d waves as %5d workgroups of %3d threads: %.2f us per launch\n", waves, blocks, tpb, ms * 100.f);
  }
  return 0;
}
