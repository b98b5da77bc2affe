// scratch microbenchmark: how many VALU instructions hide behind one v_mfma_f32_32x32x16_f16 on a gfx950 SIMD, with one and two waves per SIMD?
// Each wave runs R iterations of { 1 MFMA ; NV VALU ops of a kind }, four independent accumulators; reports shader cycles per iteration (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int NV>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int R) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
  f32x2 pv[8];
  for (int i = 0; i < 8; ++i) pv[i] = f32x2{v[i], v[i + 8]};
  unsigned pk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < R; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      __builtin_amdgcn_sched_barrier(0);
      acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int x = (u * NV + j) & 15;
        if (KIND == 0) asm volatile("v_add_f32_e32 %0, 1.0, %0" : "+v"(v[x]));
        if (KIND == 1) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(v[x]));
        if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(pv[x & 7]));
        if (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[x & 7]) : "v"(v[x]), "v"(v[(x + 1) & 15]));
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += pv[i][0] + pv[i][1] + (float)pk[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int NV>
void run(const char* name, int blocks_per_cu, float* out, long long* cyc, long long* h) {
  const int R = 2000, nb = 256 * blocks_per_cu;
  hipLaunchKernelGGL((k<KIND, NV>), dim3(nb), dim3(256), blocks_per_cu == 1 ? 100000 : 0, 0, out, cyc, R);   // LDS request pins 1 block per CU
  hipLaunchKernelGGL((k<KIND, NV>), dim3(nb), dim3(256), blocks_per_cu == 1 ? 100000 : 0, 0, out, cyc, R);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, nb * sizeof(long long), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < nb; ++i) s += h[i];
  printf("%-10s NV=%2d waves/SIMD=%d: %7.1f memtime ticks per MFMA (per wave)\n", name, NV, blocks_per_cu, s / nb / (R * 4.0));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 256 * sizeof(float));
  hipMalloc(&cyc, 1024 * sizeof(long long));
  long long* h = (long long*)malloc(1024 * sizeof(long long));
  for (int bpc = 1; bpc <= 2; ++bpc) {
#define KINDS(NV) run<0, NV>("v_add", bpc, out, cyc, h); run<1, NV>("v_exp", bpc, out, cyc, h); run<2, NV>("v_pk_add", bpc, out, cyc, h); run<3, NV>("v_cvt_pk", bpc, out, cyc, h);
    run<0, 0>("none", bpc, out, cyc, h);
    KINDS(2) KINDS(4) KINDS(6) KINDS(8) KINDS(12)
  }
  // the clock the counter runs at: a fixed-length busy loop against wall time
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<0, 0>), dim3(256), dim3(256), 0, 0, out, cyc, 200000);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
  printf("calibration: %lld ticks in %.3f ms -> %.1f MHz tick rate; 800000 MFMAs x 32 cycles -> shader clock %.0f MHz if back to back\n", h[0], ms, h[0] / ms / 1e3, 800000.0 * 32 / ms / 1e3);
  return 0;
}
