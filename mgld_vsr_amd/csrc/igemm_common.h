// igemm_common.h — device helpers shared by the two translation units of the GEMM family (igemm.hip: the im2col / linear kernel,
// conv3q.hip: the patch-staged 3x3 convolutions): LDS-DMA staging primitive, epilogue (bias / row vector / activation / residual,
// coalesced stores through a per-wave LDS patch), row maps.  Everything here is per translation unit (anonymous namespace).
#pragma once
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

// host-side state and cross-unit entry points
namespace mgld_ig {
int num_cus();
extern thread_local float* g_ws;        // split-K scratch of the calling host thread (mgld_set_workspace)
extern thread_local size_t g_ws_bytes;
void launch_splitk_reduce(const MgldIGemm* p, hipStream_t s, int splits);
// conv3q.hip
bool conv3p_plan(const MgldIGemm* p, int* bn, int* splits, int* hchunk);
bool conv3q_plan(const MgldIGemm* p, int* id, int* splits, int* hchunk);
int dispatch_conv3p(const MgldIGemm* p, hipStream_t s, int bn, int splits, int hchunk);
int dispatch_conv3q(const MgldIGemm* p, hipStream_t s, int id, int splits, int hchunk);
void conv3q_kernel_name(const MgldIGemm* p, int id, char* buf, int buflen);
// ppgemm.hip (ping-pong LINEAR kernels)
bool ppgemm_plan(const MgldIGemm* p, int* id);
int dispatch_ppgemm(const MgldIGemm* p, hipStream_t s, int id);
void ppgemm_kernel_name(const MgldIGemm* p, int id, char* buf, int buflen);
int ppgemm_row_chunks(const MgldIGemm* p, int id);   // column tiles of configuration `id` (MgldIGemm.row_part)
// conv3r.hip (ping-pong patch convolutions)
bool conv3r_plan(const MgldIGemm* p, int* id, int* splits);
int dispatch_conv3r(const MgldIGemm* p, hipStream_t s, int id, int splits);
void conv3r_kernel_name(const MgldIGemm* p, int id, char* buf, int buflen);
int conv3r_gn_chunks(const MgldIGemm* p, int id);
// pptconv.hip (ping-pong temporal Conv3d)
bool pptconv_plan(const MgldIGemm* p, int* id, int* lgP);
int dispatch_pptconv(const MgldIGemm* p, hipStream_t s, int id, int lgP);
void pptconv_kernel_name(const MgldIGemm* p, int id, char* buf, int buflen);
}  // namespace mgld_ig

namespace {


#ifndef MGLD_IGEMM_ABLATE
#define MGLD_IGEMM_ABLATE 0   // timing-only ablation builds: 1 / 2 conv3p contiguous A / W pieces, 8 no A traffic (conv), 16 no W traffic, 32 no MFMA, 128 no compute, 256 no DMA, 512 no epilogue
#endif
constexpr int ABL = MGLD_IGEMM_ABLATE;

#ifndef MGLD_IGEMM_PF
#define MGLD_IGEMM_PF 1         // LDS fragment prefetch distance of igemm_kernel's k loop (k-steps ahead); 2 in A/B builds
#endif
constexpr int IG_PF = MGLD_IGEMM_PF;
constexpr int BK = 64;          // k depth per stage (fp16 elements) = 128 B per tile row
constexpr int ROWB = BK * 2;    // bytes per tile row in LDS

__device__ uint4 g_zero_page[4];  // 64 B of zeros: source of padded / out-of-range 16-B chunks

struct RowInfo {
  int64_t base;  // LINEAR: m*lda ; CONV: n*Hin*Win (pixel index) ; TCONV: m (row index)
  int iy0, ix0;  // CONV: top-left input coord ; TCONV: iy0 = t
  bool valid;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

struct EpiParams {
  const float* bias; const float* bias_m; const float* rowvec; const f16* R;
  int rows_per_frame, ld_rowvec, ldr, act; float alpha, beta;
};

__device__ __forceinline__ float apply_act(float x, int act) {
  if (act == MGLD_ACT_RELU) return fmaxf(x, 0.f);
  if (act == MGLD_ACT_LRELU02) return x > 0.f ? x : 0.2f * x;
  if (act == MGLD_ACT_SILU) return silu_f(x);
  if (act == MGLD_ACT_SIGMOID) return 1.0f / (1.0f + __expf(-x));
  if (act == MGLD_ACT_TANH) return tanhf(x);
  if (act == MGLD_ACT_GELU) return gelu_f(x);
  return x;
}

// tile-local output row -> global output row m (or -1: the row does not exist)
struct RowMapLinear {   // BM consecutive rows starting at bm0
  int bm0, M;
  __device__ __forceinline__ int operator()(int r) const { const int m = bm0 + r; return m < M ? m : -1; }
};
struct RowMapFrames {   // TCONV: 2^lg consecutive pixels of EVERY frame of a clip (rows >> lg = frame); lg = 31: consecutive rows
  int m0, lg, mask, HW, M;
  __device__ __forceinline__ int operator()(int r) const { const int m = m0 + (r >> lg) * HW + (r & mask); return m < M ? m : -1; }
};
template <int TX>
struct RowMap2D {       // a TY x TX pixel tile of one frame, raster order inside the tile
  int fbase, y0, x0, H, W;
  __device__ __forceinline__ int operator()(int r) const {
    const int y = y0 + r / TX, x = x0 + (r & (TX - 1));
    return (y < H && x < W) ? fbase + y * W + x : -1;
  }
};

// ---- epilogue row pass ------------------------------------------------------------------------------------------------
enum { EPI_GENERIC = 0, EPI_PLAIN_NONE = 1, EPI_PLAIN_SILU = 2, EPI_GEGLU = 3, EPI_SLAB = 4 };

#ifndef MGLD_EPI_PK
#define MGLD_EPI_PK 1     // 1: epilogue arithmetic on float pairs (v_pk_*_f32); 0: A/B build with the same code on scalars
#endif
#if MGLD_EPI_PK
typedef f32x2 e2;
__device__ __forceinline__ e2 pk(float a, float b) { return e2{a, b}; }
#else
struct e2 {
  float x, y;
  __device__ __forceinline__ float& operator[](int i) { return i ? y : x; }
  __device__ __forceinline__ float operator[](int i) const { return i ? y : x; }
};
__device__ __forceinline__ e2 pk(float a, float b) { return e2{a, b}; }
__device__ __forceinline__ e2 operator+(e2 a, e2 b) { return e2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ e2 operator-(e2 a, e2 b) { return e2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ e2 operator*(e2 a, e2 b) { return e2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ e2& operator+=(e2& a, e2 b) { a.x += b.x; a.y += b.y; return a; }
__device__ __forceinline__ e2& operator*=(e2& a, e2 b) { a.x *= b.x; a.y *= b.y; return a; }
#endif
__device__ __forceinline__ e2 silu2(e2 x) {
  const e2 t = x * pk(-1.4426950408889634f, -1.4426950408889634f);
  const e2 d = pk(__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])) + pk(1.f, 1.f);
  return x * pk(__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]));
}
// exact-GELU of a pair: erf(z) = sign(z) (1 - 2^(t q(t))), t = min(|z|, 4), q a degree-5 polynomial fitted to log2(erfc(t)) / t on [0, 4]
// (weighted minimax, tools/fit_erf.py; |erf error| <= 2.9e-7 in fp32 Horner: fp32 round-off level, three orders below the fp16 rounding of
// the stored product).  One transcendental (v_exp_f32) per element; the polynomial runs as v_pk_fma_f32 on the pair.
__device__ __forceinline__ e2 gelu2(e2 x) {
  const e2 z = x * pk(0.70710678118654752440f, 0.70710678118654752440f);
  const e2 t = pk(fminf(fabsf(z[0]), 4.f), fminf(fabsf(z[1]), 4.f));
  e2 q = t * pk(1.4204740e-04f, 1.4204740e-04f) + pk(-3.6643003e-03f, -3.6643003e-03f);
  q = q * t + pk(3.0896224e-02f, 3.0896224e-02f);
  q = q * t + pk(-1.4969946e-01f, -1.4969946e-01f);
  q = q * t + pk(-9.1816545e-01f, -9.1816545e-01f);
  q = q * t + pk(-1.6279250e+00f, -1.6279250e+00f);
  q = q * t;
  const e2 e = pk(1.f, 1.f) - pk(__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1]));
  const e2 hx = x * pk(0.5f, 0.5f);
  return hx + hx * pk(copysignf(e[0], z[0]), copysignf(e[1], z[1]));
}

// one 32-row slice of a wave's tile: rows r0 + prow of the fp32 patch -> bias / row vector / activation / residual -> global.
// KIND is compile-time, everything it excludes is not in the instruction stream.
template <int KIND, typename RowMap>
__device__ __forceinline__ void epi_rows(const MgldIGemm& p, const RowMap rmap, const float* patch, const int LDW, const int rbase,
                                         const int rpi, const int prow, const int pcv, const int n, const int Nout, const bool full,
                                         const float (&bcol)[8], const float (&bgate)[8], const f16* __restrict__ R, char* outp,
                                         const int64_t cbase, const int ldo, const bool of32, const int act, const float alpha, const bool geglu) {
  for (int r0 = 0; r0 < 32; r0 += rpi) {
    const int row = r0 + prow;
    const int m = row < 32 ? rmap(rbase + row) : -1;
    if (m < 0 || n >= Nout) continue;
    const f32x4 a0 = *(const f32x4*)(patch + row * LDW + pcv);
    const f32x4 a1 = *(const f32x4*)(patch + row * LDW + pcv + 4);
    if constexpr (KIND == EPI_SLAB) {           // split-K: raw fp32 partial sums into this split's slab
      float* cp = (float*)outp + cbase + (int64_t)m * ldo + n;
      if (full && ((((uintptr_t)cp) & 15) == 0)) {
        *(f32x4*)cp = a0;
        *(f32x4*)(cp + 4) = a1;
      } else {
        const float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
        for (int j = 0; j < 8; ++j) if (n + j < Nout) cp[j] = v[j];
      }
    } else if constexpr (KIND == EPI_PLAIN_NONE || KIND == EPI_PLAIN_SILU || KIND == EPI_GEGLU) {
      e2 v[4] = {pk(a0[0], a0[1]), pk(a0[2], a0[3]), pk(a1[0], a1[1]), pk(a1[2], a1[3])};
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += pk(bcol[2 * j], bcol[2 * j + 1]);
      if constexpr (KIND == EPI_GEGLU) {
        const f32x4 g0 = *(const f32x4*)(patch + row * LDW + 32 + pcv);
        const f32x4 g1 = *(const f32x4*)(patch + row * LDW + 32 + pcv + 4);
        const e2 g[4] = {pk(g0[0], g0[1]), pk(g0[2], g0[3]), pk(g1[0], g1[1]), pk(g1[2], g1[3])};
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= gelu2(g[j] + pk(bgate[2 * j], bgate[2 * j + 1]));
      } else {
        if (p.rowvec) {
          const float* rv = p.rowvec + (int64_t)(m / p.rows_per_frame) * p.ld_rowvec + n;
          if (full && ((((uintptr_t)rv) & 15) == 0)) {
            const f32x4 r0v = *(const f32x4*)rv, r1v = *(const f32x4*)(rv + 4);
            v[0] += pk(r0v[0], r0v[1]); v[1] += pk(r0v[2], r0v[3]); v[2] += pk(r1v[0], r1v[1]); v[3] += pk(r1v[2], r1v[3]);
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (n + j < Nout) v[j >> 1][j & 1] += rv[j];
          }
        }
        if constexpr (KIND == EPI_PLAIN_SILU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = silu2(v[j]);
        }
      }
      f16* cp = (f16*)outp + cbase + (int64_t)m * ldo + n;
      if (R) {
        const f16* rp = R + (int64_t)m * p.ldr + n;
        const e2 beta2 = pk(p.beta, p.beta);
        if (full && ((((uintptr_t)rp) & 15) == 0)) {
          const f16x8 rr = *(const f16x8*)rp;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += beta2 * pk((float)rr[2 * j], (float)rr[2 * j + 1]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) if (n + j < Nout) v[j >> 1][j & 1] += p.beta * (float)rp[j];
        }
        if (p.Rlo) {                    // low plane of the residual stream (same index as R; batch 1)
          const f16* rl = (const f16*)p.Rlo + (int64_t)m * p.ldr + n;
          const e2 bl2 = pk(p.beta * MGLD_LO_SCALE, p.beta * MGLD_LO_SCALE);
          if (full && ((((uintptr_t)rl) & 15) == 0)) {
            const f16x8 rr = *(const f16x8*)rl;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bl2 * pk((float)rr[2 * j], (float)rr[2 * j + 1]);
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (n + j < Nout) v[j >> 1][j & 1] += p.beta * MGLD_LO_SCALE * (float)rl[j];
          }
        }
      }
      const f16x8 o8 = f16x8{(f16)v[0][0], (f16)v[0][1], (f16)v[1][0], (f16)v[1][1], (f16)v[2][0], (f16)v[2][1], (f16)v[3][0], (f16)v[3][1]};
      if (full && ((((uintptr_t)cp) & 15) == 0)) {
        *(f16x8*)cp = o8;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (n + j < Nout) cp[j] = o8[j];
      }
      if (p.Clo) {                      // low plane of the output
        f16* cl = (f16*)p.Clo + cbase + (int64_t)m * ldo + n;
        f16x8 l8;
#pragma unroll
        for (int j = 0; j < 8; ++j) l8[j] = lo_plane(v[j >> 1][j & 1], o8[j]);
        if (full && ((((uintptr_t)cl) & 15) == 0)) {
          *(f16x8*)cl = l8;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) if (n + j < Nout) cl[j] = l8[j];
        }
      }
    } else {
      float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      if (geglu) {
        const f32x4 g0 = *(const f32x4*)(patch + row * LDW + 32 + pcv);
        const f32x4 g1 = *(const f32x4*)(patch + row * LDW + 32 + pcv + 4);
        const float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (v[j] + bcol[j]) * gelu_f(g[j] + bgate[j]) * alpha;
      } else {
        const float bm = p.bias_m ? p.bias_m[m] : 0.f;
        float rvv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.rowvec) {
          const float* rv = p.rowvec + (int64_t)(m / p.rows_per_frame) * p.ld_rowvec + n;
#pragma unroll
          for (int j = 0; j < 8; ++j) if (n + j < Nout) rvv[j] = rv[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = apply_act(v[j] + bm + bcol[j] + rvv[j], act) * alpha;
      }
      if (R) {
        const f16* rp = R + (int64_t)m * p.ldr + n;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (n + j < Nout) v[j] += p.beta * (float)rp[j];
        if (p.Rlo) {                    // low plane of the residual stream (MgldIGemm.Rlo)
          const f16* rl = (const f16*)p.Rlo + (R - (const f16*)p.R) + (int64_t)m * p.ldr + n;
          const float bl = p.beta * MGLD_LO_SCALE;
#pragma unroll
          for (int j = 0; j < 8; ++j) if (n + j < Nout) v[j] += bl * (float)rl[j];
        }
      } else if (p.r_f32 && p.R) {      // fp32 residual stream (MgldIGemm.r_f32: the high-precision VAE encoder)
        const float* rp = (const float*)p.R + (int64_t)m * p.ldr + n;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (n + j < Nout) v[j] += p.beta * rp[j];
      }
      if (of32) {
        float* cp = (float*)outp + cbase + (int64_t)m * ldo + n;
        if (full && ((((uintptr_t)cp) & 15) == 0)) {
          *(f32x4*)cp = f32x4{v[0], v[1], v[2], v[3]};
          *(f32x4*)(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) if (n + j < Nout) cp[j] = v[j];
        }
      } else {
        f16* cp = (f16*)outp + cbase + (int64_t)m * ldo + n;
        if (full && ((((uintptr_t)cp) & 15) == 0)) {
          *(f16x8*)cp = f16x8{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3], (f16)v[4], (f16)v[5], (f16)v[6], (f16)v[7]};
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) if (n + j < Nout) cp[j] = (f16)v[j];
        }
        if (p.Clo) {                    // low plane of the output (MgldIGemm.Clo)
          f16* cl = (f16*)p.Clo + cbase + (int64_t)m * ldo + n;
#pragma unroll
          for (int j = 0; j < 8; ++j) if (n + j < Nout) cl[j] = lo_plane(v[j], (f16)v[j]);
        }
      }
    }
  }
}

// ---- shared tile epilogue (used by igemm_kernel, conv3p_kernel and conv3q_kernel) ---------------------------------
template <int BM, int BN, int WM, int WN, typename RowMap>
__device__ __forceinline__ void tile_epilogue(const MgldIGemm& p, float* __restrict__ ws, const bool splitk, const int kz, const int bz,
                                              const RowMap rmap, const int bn0, const int wm, const int wn, const int wave,
                                              const int lane, f32x16 (&acc)[WN / 32][WM / 32], char* smem) {
  constexpr int MI = WM / 32, NI = WN / 32;
  const int M = p.M, N = p.N;
  const int l31 = lane & 31, lhi = lane >> 5;
  // D[i = n_local][j = m_local]: a lane holds ONE output row (m = lane&31) and 4-channel groups of it, i.e. the natural
  // store would be 8-byte pieces scattered over 32 rows.  Instead every wave transposes its 32-row slices through its own
  // LDS patch (fp32, bias/activation already applied) and writes whole row segments: 8 lanes x 16 B = 128 B contiguous
  // per output row, residual rows read the same way.  (Wave-local: no block barrier except the one releasing the stages.)
  __syncthreads();
  if constexpr (ABL & 512) return;  // ablation build (timing only): no epilogue
  // wave tiles wider than 64 columns (the full-N LINEAR tiles, WN = 160) go through the patch in column chunks of CW
  constexpr int CW = WN <= 64 ? WN : (WN % 64 == 0 ? 64 : 32), NC = WN / CW, NIC = CW / 32;
  static_assert(WN % 32 == 0 && NC * CW == WN, "wave tile width");
  const bool geglu = (WN == 64) && (!splitk) && (p.act == MGLD_ACT_GEGLU);   // value/gate pairing needs 64-column wave tiles
  constexpr int LDW = CW + 4;                       // patch row stride (floats): 16-B aligned, conflict-free b128
  float* patch = (float*)smem + wave * (32 * LDW);
  const int Nout = splitk ? N : (geglu ? N / 2 : N);
  const int wcols = geglu ? CW / 2 : CW;            // output columns this wave produces per chunk
  const int64_t cbase = splitk ? (int64_t)kz * M * N : (int64_t)bz * p.strideC;
  const int ldo = splitk ? N : p.ldc;
  const bool of32 = splitk || p.out_f32;
  const f16* __restrict__ R = (!splitk && p.R && !p.r_f32) ? (const f16*)p.R + (int64_t)bz * p.strideR : nullptr;
  const int act = splitk ? MGLD_ACT_NONE : p.act;
  const float alpha = splitk ? 1.f : p.alpha;
  char* outp = splitk ? (char*)ws : (char*)p.C;
  const int lpr = wcols >> 3;                       // lanes per output row (8 columns each)
  const int rpi = 64 / lpr;                         // rows per wave pass
  const int prow = lane / lpr, pcv = (lane - prow * lpr) * 8;
  int kind = EPI_GENERIC;
  if (splitk) kind = EPI_SLAB;
  else if (geglu) kind = (alpha == 1.f && !of32) ? EPI_GEGLU : EPI_GENERIC;
  else if (!of32 && !p.bias_m && alpha == 1.f && (act == MGLD_ACT_NONE || act == MGLD_ACT_SILU)) kind = act == MGLD_ACT_SILU ? EPI_PLAIN_SILU : EPI_PLAIN_NONE;
  // (column chunks unrolled by hand through compile-time indices: a runtime `c` would index acc[] dynamically = scratch memory)
  auto do_chunk = [&](auto CI) {
  constexpr int c = decltype(CI)::value;
  const int ncol0 = geglu ? (bn0 + wn * WN) / 2 : bn0 + wn * WN + c * CW;
  const int n = ncol0 + pcv;                        // this lane's 8 output columns [n, n+8)
  const bool full = (n + 8 <= Nout);
  // per-lane column constants (same for every row): bias of the 8 columns (value and gate halves for GEGLU)
  float bcol[8], bgate[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { bcol[j] = 0.f; bgate[j] = 0.f; }
  if (!splitk && p.bias) {
    const int nb = geglu ? bn0 + wn * WN + pcv : n;   // packed row index of the value half
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (nb + j < N) bcol[j] = p.bias[nb + j];
      if (geglu && nb + 32 + j < N) bgate[j] = p.bias[nb + 32 + j];
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    // ---- phase 1: raw accumulators -> patch[row = l31][col] ----
#pragma unroll
    for (int nic = 0; nic < NIC; ++nic)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
        *(f32x4*)(patch + l31 * LDW + nic * 32 + rg * 8 + lhi * 4) =
            f32x4{acc[c * NIC + nic][mi][rg * 4], acc[c * NIC + nic][mi][rg * 4 + 1], acc[c * NIC + nic][mi][rg * 4 + 2],
                  acc[c * NIC + nic][mi][rg * 4 + 3]};
    // ---- phase 2: patch rows -> epilogue math -> global, 8 columns (16 B of fp16 / 32 B of fp32) per lane ----
    // The variant (plain fp16 epilogue with a compile-time activation / GEGLU / raw split-K slab / everything else) is picked by
    // ONE block-uniform switch per 32-row slice; inside, the arithmetic is straight-line packed fp32 (v_pk_add/mul/fma_f32).
    // Short-K launches (K = 320..1280: 5-20 k-steps) spend more issue slots here than in the k loop, so a per-element
    // runtime `act` switch (4 scalar branches per element) was the dominant cost of the transformer blocks' projections.
    switch (kind) {
      case EPI_PLAIN_NONE: epi_rows<EPI_PLAIN_NONE>(p, rmap, patch, LDW, wm * WM + mi * 32, rpi, prow, pcv, n, Nout, full, bcol, bgate, R, outp, cbase, ldo, of32, act, alpha, geglu); break;
      case EPI_PLAIN_SILU: epi_rows<EPI_PLAIN_SILU>(p, rmap, patch, LDW, wm * WM + mi * 32, rpi, prow, pcv, n, Nout, full, bcol, bgate, R, outp, cbase, ldo, of32, act, alpha, geglu); break;
      case EPI_GEGLU: epi_rows<EPI_GEGLU>(p, rmap, patch, LDW, wm * WM + mi * 32, rpi, prow, pcv, n, Nout, full, bcol, bgate, R, outp, cbase, ldo, of32, act, alpha, geglu); break;
      case EPI_SLAB: epi_rows<EPI_SLAB>(p, rmap, patch, LDW, wm * WM + mi * 32, rpi, prow, pcv, n, Nout, full, bcol, bgate, R, outp, cbase, ldo, of32, act, alpha, geglu); break;
      default: epi_rows<EPI_GENERIC>(p, rmap, patch, LDW, wm * WM + mi * 32, rpi, prow, pcv, n, Nout, full, bcol, bgate, R, outp, cbase, ldo, of32, act, alpha, geglu); break;
    }
  }
  };
  static_assert(NC <= 5, "column chunks");
  do_chunk(std::integral_constant<int, 0>{});
  if constexpr (NC > 1) do_chunk(std::integral_constant<int, 1>{});
  if constexpr (NC > 2) do_chunk(std::integral_constant<int, 2>{});
  if constexpr (NC > 3) do_chunk(std::integral_constant<int, 3>{});
  if constexpr (NC > 4) do_chunk(std::integral_constant<int, 4>{});
}

}  // namespace
