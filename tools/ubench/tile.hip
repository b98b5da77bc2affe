// scratch microbenchmark: the register-only instruction stream of one attention tile (16 MFMAs + 32 v_exp + row-sum adds + 16 v_cvt_pk), in the gap
// layouts of flash_attn_sp2_kernel and variants, one and two waves per SIMD: the compute-only ceiling of a layout (no LDS, no DMA, no barrier).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// LAYOUT 0: sp2 (QK gaps: MFMA chain + 4 exp + 2 pk_add; PV gaps: MFMA (2 accs alternating) + 2 cvt)
// LAYOUT 1: the same with 4 v_add instead of 2 pk_add
// LAYOUT 2: balanced: every gap MFMA + 2 exp + 2 v_add + 1 cvt
// LAYOUT 3: LAYOUT 1 but the QK MFMAs alternate between the two score accumulators (no back-to-back dependent MFMAs)
// LAYOUT 4: no softmax at all (16 MFMAs in the sp2 dependency pattern)
// LAYOUT 5: LAYOUT 1 without the row-sum adds
// LAYOUT 6: LAYOUT 1 without the cvt
template <int LAYOUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(float* out, long long* cyc, int R) {
  f32x16 sb[2], o[2], sa[2];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) { sb[i][r] = 0.f; o[i][r] = 0.f; sa[i][r] = -0.01f * (threadIdx.x & 15) - r; }
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  float e[32];
  float rs[4] = {0, 0, 0, 0};
  f32x2 rq[2] = {{0, 0}, {0, 0}};
  u32x4 pw[4];
  for (int i = 0; i < 4; ++i) pw[i] = u32x4{0, 0, 0, 0};
  for (int i = 0; i < 32; ++i) e[i] = 0.f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < R; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {      // "QK^T" gaps
      __builtin_amdgcn_sched_barrier(0);
      const int k2 = (LAYOUT == 3) ? (g & 1) : (g >> 2);
      sb[k2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, sb[k2], 0, 0, 0);
      if (LAYOUT == 4) continue;
      const int ne = LAYOUT == 2 ? 2 : 4;
#pragma unroll
      for (int j = 0; j < ne; ++j) {
        const int i = g * ne + j;
        asm volatile("v_exp_f32_e32 %0, %1" : "=v"(e[i]) : "v"(sa[i >> 4][i & 15]));
      }
      if (LAYOUT == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x2 ep = {e[g * 4 + 2 * j], e[g * 4 + 2 * j + 1]};
          asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(rq[j]) : "v"(ep));
        }
      } else if (LAYOUT != 5) {
#pragma unroll
        for (int j = 0; j < ne; ++j) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(rs[j]) : "v"(e[g * ne + j]));
      }
      if (LAYOUT == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pw[g >> 2][g & 3]) : "v"(e[16 + 2 * g]), "v"(e[16 + 2 * g + 1]));
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {      // "PV" gaps
      __builtin_amdgcn_sched_barrier(0);
      o[g & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, pw[g >> 1]), o[g & 1], 0, 0, 0);
      if (LAYOUT == 4 || LAYOUT == 6) continue;
      if (LAYOUT == 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int i = 16 + g * 2 + j;
          asm volatile("v_exp_f32_e32 %0, %1" : "=v"(e[i]) : "v"(sa[i >> 4][i & 15]));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(rs[j]) : "v"(e[16 + g * 2 + j]));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pw[2 + (g >> 2)][g & 3]) : "v"(e[2 * g]), "v"(e[2 * g + 1]));
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int w = g * 2 + j;     // packed word w of the NEXT use (value dependencies do not matter for timing)
          asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pw[(w >> 2) & 3][w & 3]) : "v"(e[(2 * w) & 31]), "v"(e[(2 * w + 1) & 31]));
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = rs[0] + rs[1] + rs[2] + rs[3] + rq[0][0] + rq[0][1] + rq[1][0] + rq[1][1];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += sb[i][r] + o[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int LAYOUT>
void run(const char* name, float* out, long long* cyc, long long* h) {
  const int R = 500;
  double res[3], wall[3];
  for (int bpc = 1; bpc <= 2; ++bpc) {
    const int nb = 256 * bpc;
    hipFuncSetAttribute((const void*)k<LAYOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, 100000);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<LAYOUT>), dim3(nb), dim3(256), bpc == 1 ? 100000 : 70000, 0, out, cyc, R);   // the LDS request pins exactly bpc blocks per CU
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<LAYOUT>), dim3(nb), dim3(256), bpc == 1 ? 100000 : 70000, 0, out, cyc, R);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    wall[bpc] = ms * 1e6 / R / bpc;      // ns per tile per SIMD
    hipMemcpy(h, cyc, nb * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < nb; ++i) s += h[i];
    res[bpc] = s / nb / R;
  }
  printf("%-34s one wave/SIMD %7.1f ticks per tile | two waves/SIMD %7.1f ticks per wave-tile = %6.1f per tile per SIMD | wall ns per tile per SIMD: %6.1f one wave, %6.1f two waves (16 MFMAs at 2.4 GHz = 213 ns)\n", name, res[1], res[2], res[2] / 2, wall[1], wall[2]);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 256 * sizeof(float));
  hipMalloc(&cyc, 1024 * sizeof(long long));
  long long* h = (long long*)malloc(1024 * sizeof(long long));
  run<4>("MFMAs only (sp2 dependencies)", out, cyc, h);
  run<0>("sp2: 4 exp + 2 pk_add | 2 cvt", out, cyc, h);
  run<1>("4 exp + 4 add | 2 cvt", out, cyc, h);
  run<3>("same, QK MFMAs alternate accs", out, cyc, h);
  run<2>("balanced 2 exp + 2 add + 1 cvt", out, cyc, h);
  run<5>("4 exp | 2 cvt (no adds)", out, cyc, h);
  run<6>("4 exp + 4 add | - (no cvt)", out, cyc, h);
  return 0;
}
