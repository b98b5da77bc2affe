// igemm.hip — implicit-GEMM on CDNA4 matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate).
//
//   C[M,N] = epilogue( gather(A)[M,K] * W[N,K]^T )
//
// One kernel serves every dense contraction on the MGLD-VSR hot path (SURVEY.md §2.2 K1,K2,K4,K8,K13):
//   * LINEAR   — nn.Linear / 1x1 conv / batched NT GEMM (attention.py:323-330,51-71; openaimodel.py:515-519)
//   * CONV3X3  — 3x3 conv on NHWC, stride 1/2, asymmetric zero pad, optional nearest-2x upsample folded into the
//                gather (openaimodel.py:176,185,221; model.py:96,114-118)
//   * TCONV3   — Conv3d (3,1,1) over the frame axis (diffusionmodules/util.py:298)
//
// Structure (gfx950):
//   * K is consumed in 64-deep stages.  Both tiles are brought HBM -> LDS by the DMA path
//     (global_load_lds_dwordx4, 16 B per lane, 1 KiB per wave instruction): no staging VGPRs, no ds_write pass.  The
//     LDS image of a wave instruction is lane-linear (8 rows x 128 B), so the bank-conflict swizzle is applied on the
//     per-lane SOURCE address (16-B chunk c of row r is stored at chunk c ^ ((r>>1)&7)) and undone on the ds_read_b128
//     side; rows of a 16-lane ds_read_b128 group then hit 16 distinct 16-B slots.
//   * The implicit-GEMM gather (conv taps, stride, nearest-2x upsample, frame shifts) is just that per-lane source
//     address; zero padding / ragged tails read a 64-B device zero page.
//   * two LDS stages; stage k+1 is in flight while stage k feeds the MFMAs; one barrier per stage.
//   * the WEIGHT tile is the MFMA row operand, so a lane ends up holding 4 consecutive output channels per register
//     group -> 8-byte stores and a lane-local GEGLU pairing.
//   * problems with too few output tiles for 256 CUs but a deep K (the 16x16 / 8x8 UNet levels: M <= 2048, K up to
//     23040) are split along K over grid.z into fp32 partials and finished by a small reduce+epilogue kernel.
#include "common.h"
#include <stdio.h>
#include <type_traits>
#include <stdlib.h>

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
      }
      if (full && ((((uintptr_t)cp) & 15) == 0)) {
        *(f16x8*)cp = f16x8{(f16)v[0][0], (f16)v[0][1], (f16)v[1][0], (f16)v[1][1], (f16)v[2][0], (f16)v[2][1], (f16)v[3][0], (f16)v[3][1]};
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (n + j < Nout) cp[j] = (f16)v[j >> 1][j & 1];
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
  const f16* __restrict__ R = (!splitk && p.R) ? (const f16*)p.R + (int64_t)bz * p.strideR : nullptr;
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

template <int MODE, bool FAST, int BM, int BN, int WM, int WN, int NST>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void igemm_kernel(const MgldIGemm p, float* __restrict__ ws, int kchunk, int order) {
  constexpr int WAVES_N = BN / WN;
  constexpr int NW = (BM / WM) * (BN / WN);       // waves per block: 4 (256 threads) or 8 (512 threads)
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int JA = BM / (8 * NW), JB = BN / (8 * NW);  // glds instructions per wave per stage (8 rows each)
  constexpr int STAGE = (BM + BN) * ROWB;
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per block");
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows in units of 8*waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // Workgroups are dealt round-robin to the 8 XCDs (private L2s) in linear-id order.  order 1: the N/BN blocks that share an A
  // tile run back to back on ONE XCD (XCD k owns the row tiles k, k+8, ...): A leaves HBM once instead of once per column tile
  // (big-M / small-N problems: the 64x64-level projections).  order 2: the M/BM blocks that share a W tile run on one XCD (XCD k
  // owns the column tiles k, k+8, ...): each XCD streams an eighth of the weights (small-M / deep-K problems).  Speed only.
  int tile_m = blockIdx.x, tile_n = blockIdx.y;
  if (order == 1) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, j = lin >> 3;
    tile_m = xcd + 8 * (j / (int)gridDim.y);
    tile_n = j % (int)gridDim.y;
  } else if (order == 2) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, j = lin >> 3;
    tile_n = xcd + 8 * (j / (int)gridDim.x);
    tile_m = j % (int)gridDim.x;
  }
  const int bm0 = tile_m * BM;
  const int bn0 = tile_n * BN;
  const bool splitk = (ws != nullptr);
  const int bz = splitk ? 0 : blockIdx.z;
  const int kz = splitk ? blockIdx.z : 0;

  const f16* __restrict__ A = (const f16*)p.A + (int64_t)bz * p.strideA;
  const f16* __restrict__ W = (const f16*)p.W + (int64_t)bz * p.strideW;

  const int M = p.M, N = p.N, K = p.K;
  const int Cin = (MODE == MGLD_MODE_LINEAR) ? K : p.Cin;
  const int k_begin = splitk ? kz * kchunk : 0;
  const int k_end = splitk ? min(K, k_begin + kchunk) : K;

  // ---- per-lane staging assignment ------------------------------------------------------------------------
  // wave instruction q = j*4 + wave covers tile rows [q*8, q*8+8); lane -> row q*8 + (lane>>3), physical chunk lane&7.
  // logical chunk = phys ^ ((row>>1)&7) = (lane&7) ^ (((wave&1)*4 + (lane>>4)) & 7): the same for every j.
  const int cphys = lane & 7;
  const int clog = cphys ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
  const char* zero = (const char*)g_zero_page;
  const int nk = (k_end - k_begin + BK - 1) / BK;
  // W2 (MgldIGemm): the K range is walked TWICE over the same A — first against the scaled fp16 residual of the weights (same layout,
  // `wdelta` bytes away from W), then, after ONE multiplication of the accumulators by w2_scale, against W itself.
  const bool two = (p.W2 != nullptr);
  const int nkt = two ? 2 * nk : nk;
  const int64_t wdelta = two ? (const char*)p.W2 - (const char*)p.W : 0;
  const f16* __restrict__ Wc = two ? (const f16*)((const char*)W + wdelta) : W;     // GENERAL path: matrix of the stage being issued

  // ======== FAST path state: K % 64 == 0, and for the gather modes Cin % 64 == 0 and no upsample fold. ========
  // Every 64-deep stage then lies inside ONE tap, so the tap / channel offset is wave-uniform (SGPR) and a lane's
  // source address is  row_pointer + uniform_offset : one 64-bit add + a validity select per 16-B chunk.
  const char* fa_ptr[JA];     // LINEAR: running pointer ; CONV/TCONV: pointer of tap (0,0)/(dt=0) incl. this lane's chunk
  unsigned fa_step[JA];       // LINEAR: bytes to advance per stage (0 for rows past M -> stay on the zero page)
  unsigned fa_mask[JA];       // CONV/TCONV: bit t set = tap t is inside the image / clip for this row
  const char* fw_ptr[JB];
  unsigned fw_step[JB];
  int s_tap = 0, s_c0 = 0;    // wave-uniform tap / channel offset of the current stage
  // ======== GENERAL path state ========
  RowInfo ra[JA];
  int64_t b_base[JB];
  bool b_valid[JB];
  int kl = k_begin + clog * 8;
  int tap = 0, c = 0;

  if constexpr (FAST) {
    if constexpr (MODE != MGLD_MODE_LINEAR) {
      if (p.tap_inner) {
        constexpr int NTAP = (MODE == MGLD_MODE_CONV3X3) ? 9 : 3;
        const int st = k_begin / BK;           // stage index -> (channel block, tap)
        s_c0 = (st / NTAP) * BK;
        s_tap = st - (st / NTAP) * NTAP;
      } else {
        s_tap = k_begin / Cin;
        s_c0 = k_begin - s_tap * Cin;
      }
    }
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      const int row = (j * NW + wave) * 8 + (lane >> 3);
      const int m = bm0 + row;
      const bool valid = m < M;
      const int mm = valid ? m : 0;
      fa_step[j] = 0; fa_mask[j] = 0;
      if constexpr (MODE == MGLD_MODE_LINEAR) {
        fa_ptr[j] = valid ? (const char*)(A + (int64_t)mm * p.lda + k_begin + clog * 8) : zero;
        fa_step[j] = valid ? BK * 2 : 0;
      } else if constexpr (MODE == MGLD_MODE_CONV3X3) {
        const int hw = p.Hout * p.Wout;
        const int n = mm / hw;
        const int r = mm - n * hw;
        const int oy = r / p.Wout, ox = r - oy * p.Wout;
        const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
        fa_ptr[j] = (const char*)(A + (((int64_t)n * p.Hin + iy0) * p.Win + ix0) * p.lda + clog * 8);
        unsigned msk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = iy0 + t / 3, ix = ix0 + t % 3;
          if (valid && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) msk |= 1u << t;
        }
        fa_mask[j] = msk;
      } else {
        const int f = mm / p.HW;
        const int t0 = (f + p.t_off) % p.T;
        fa_ptr[j] = (const char*)(A + ((int64_t)mm - p.HW) * p.lda + clog * 8);   // dt = 0 reads frame t-1
        unsigned msk = 0;
#pragma unroll
        for (int t = 0; t < 3; ++t)
          if (valid && (unsigned)(t0 + t - 1) < (unsigned)p.T) msk |= 1u << t;
        fa_mask[j] = msk;
      }
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int n = bn0 + (j * NW + wave) * 8 + (lane >> 3);
      const bool valid = n < N;
      fw_ptr[j] = valid ? (const char*)(W + (int64_t)n * p.ldw + k_begin + clog * 8) + wdelta : zero;
      fw_step[j] = valid ? BK * 2 : 0;
    }
  } else {
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      const int row = (j * NW + wave) * 8 + (lane >> 3);
      const int m = bm0 + row;
      ra[j].valid = m < M;
      ra[j].base = 0; ra[j].iy0 = 0; ra[j].ix0 = 0;
      const int mm = ra[j].valid ? m : 0;  // computed unconditionally (branch-free); invalid rows read the zero page
      if constexpr (MODE == MGLD_MODE_LINEAR) {
        ra[j].base = (int64_t)mm * p.lda;
      } else if constexpr (MODE == MGLD_MODE_CONV3X3) {
        const int hw = p.Hout * p.Wout;
        const int n = mm / hw;
        const int r = mm - n * hw;
        const int oy = r / p.Wout, ox = r - oy * p.Wout;
        ra[j].base = (int64_t)n * p.Hin * p.Win;
        ra[j].iy0 = oy * p.stride - p.pad_t;
        ra[j].ix0 = ox * p.stride - p.pad_l;
      } else {  // TCONV3
        const int f = mm / p.HW;
        ra[j].base = mm;
        ra[j].iy0 = (f + p.t_off) % p.T;
      }
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int n = bn0 + (j * NW + wave) * 8 + (lane >> 3);
      b_valid[j] = n < N;
      b_base[j] = (int64_t)n * p.ldw;
    }
    tap = kl / Cin;
    c = kl - tap * Cin;
  }

  // (tap / channel offset are passed BY VALUE and advanced in the caller's scope: captured by reference and mutated
  // inside the lambda they ended up in scratch memory, turning the whole address computation into per-lane VALU work.)
  auto issue_stage = [&](int buf, const int u_tap, const int u_c0) {
    if constexpr (ABL & 256) return;  // ablation build (timing only): no global->LDS traffic at all
    char* sbase = smem + buf * STAGE + wave * 1024;
    if constexpr (FAST) {
      if constexpr (MODE == MGLD_MODE_LINEAR) {
#pragma unroll
        for (int j = 0; j < JA; ++j) {
          glds16(fa_ptr[j], sbase + j * (NW * 1024));
          fa_ptr[j] += fa_step[j];
        }
      } else {
        // uniform byte offset of this stage's tap + channel block
        int64_t soff;
        if constexpr (MODE == MGLD_MODE_CONV3X3) {
          const int ky = (u_tap * 11) >> 5, kx = u_tap - ky * 3;   // tap / 3 for tap in [0, 9)
          soff = ((int64_t)(ky * p.Win + kx) * p.lda + u_c0) * 2;
        } else {
          soff = ((int64_t)u_tap * p.HW * p.lda + u_c0) * 2;
        }
        const unsigned tbit = 1u << u_tap;
#pragma unroll
        for (int j = 0; j < JA; ++j) {
          if constexpr (ABL & 8) continue;   // (ablation build bit 8: skip the activation-tile traffic)
          const char* src = (fa_mask[j] & tbit) ? fa_ptr[j] + soff : zero;
          glds16(src, sbase + j * (NW * 1024));
        }
      }
      if constexpr (!(ABL & 16)) {  // (ablation build bit 16: skip the weight-tile traffic)
#pragma unroll
        for (int j = 0; j < JB; ++j) {
          glds16(fw_ptr[j], sbase + BM * ROWB + j * (NW * 1024));
          fw_ptr[j] += fw_step[j];
        }
      }
    } else {
      const bool kval = kl < k_end;
      // branch-free: the offset is always computed, the pointer is SELECTED (out-of-range -> zero page)
      int ky = 0, kx = 0;
      if constexpr (MODE == MGLD_MODE_CONV3X3) { const int kw = p.kw > 0 ? p.kw : 3; ky = tap / kw; kx = tap - ky * kw; }
      const int hlim = p.up2 ? 2 * p.Hin : p.Hin, wlim = p.up2 ? 2 * p.Win : p.Win, sh = p.up2 ? 1 : 0;
#pragma unroll
      for (int j = 0; j < JA; ++j) {
        bool ok = ra[j].valid & kval;
        int64_t off;
        if constexpr (MODE == MGLD_MODE_LINEAR) {
          off = ra[j].base + kl;
        } else if constexpr (MODE == MGLD_MODE_CONV3X3) {
          const int iy = ra[j].iy0 + ky, ix = ra[j].ix0 + kx;
          ok &= ((unsigned)iy < (unsigned)hlim) & ((unsigned)ix < (unsigned)wlim);
          off = (ra[j].base + (int64_t)((iy >> sh) * p.Win + (ix >> sh))) * p.lda + c;
        } else {
          const int tt = ra[j].iy0 + tap - 1;
          ok &= (unsigned)tt < (unsigned)p.T;
          off = (ra[j].base + (int64_t)(tap - 1) * p.HW) * p.lda + c;
        }
        const f16* src = ok ? (A + off) : (const f16*)zero;
        glds16(src, sbase + j * (NW * 1024));
      }
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        const f16* src = (b_valid[j] & kval) ? (Wc + b_base[j] + kl) : (const f16*)zero;
        glds16(src, sbase + BM * ROWB + j * (NW * 1024));
      }
      kl += BK;
      c += BK;
      while (c >= Cin) { c -= Cin; ++tap; }
    }
  };

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  const int l31 = lane & 31, lhi = lane >> 5;
  // fragment row byte offsets and swizzle keys (row index within the tile; wave offsets are multiples of 32)
  int a_off[MI], a_key[MI], w_off[NI], w_key[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int r = wm * WM + mi * 32 + l31;
    a_off[mi] = r * ROWB; a_key[mi] = (r >> 1) & 7;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int r = wn * WN + ni * 32 + l31;
    w_off[ni] = BM * ROWB + r * ROWB; w_key[ni] = (r >> 1) & 7;
  }

  // NST-deep ring of LDS stages: stages kt+1 .. kt+NST-1 are in flight (DMA) while stage kt feeds the MFMAs.
  // A wave waits only for ITS OWN stage-kt loads with a counted vmcnt (newer stages stay in flight across the
  // barrier), then the barrier makes every wave's stage kt visible and retires all reads of the buffer about to be
  // refilled.
  static_assert(NST == 0 || (NST >= 2 && NST <= 4), "2 .. 4 DMA stages, or 0 = register-staged tiles (two LDS buffers)");
  // wave-uniform K position of the NEXT stage to issue; kept in this scope as plain scalars (SGPRs)
  int u_tap = __builtin_amdgcn_readfirstlane(s_tap), u_c0 = __builtin_amdgcn_readfirstlane(s_c0);
#define MGLD_ADVANCE_TAP()                                                            \
  if constexpr (FAST && MODE != MGLD_MODE_LINEAR) {                                   \
    constexpr int NTAP_ = (MODE == MGLD_MODE_CONV3X3) ? 9 : 3;                        \
    if (p.tap_inner) { /* K order (chunk64, tap, c): taps of one 64-channel block back to back (L1/L2 reuse) */ \
      const bool wrap_ = (u_tap + 1 == NTAP_);                                        \
      u_c0 = wrap_ ? u_c0 + BK : u_c0;                                                \
      u_tap = wrap_ ? 0 : u_tap + 1;                                                  \
    } else {           /* K order (tap, Cin) */                                       \
      const bool wrap_ = (u_c0 + BK >= Cin);                                          \
      u_c0 = wrap_ ? u_c0 + BK - Cin : u_c0 + BK;                                     \
      u_tap = wrap_ ? u_tap + 1 : u_tap;                                              \
    }                                                                                 \
  }
  // MFMAs of one 64-deep stage held in the LDS buffer `sb`
  auto compute_stage = [&](const char* sb) {
    if constexpr (ABL & 128) return;  // ablation build (timing only): no LDS reads, no MFMA
    // fragments of k-step ks+IG_PF are fetched from LDS while the MFMAs of k-step ks run (IG_PF + 1 register sets, static indices)
    f16x8 fa[IG_PF + 1][MI], fw[IG_PF + 1][NI];
    auto load_frags = [&](int ks, int set) {
      const int cl = ks * 2 + lhi;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fa[set][mi] = *(const f16x8*)(sb + a_off[mi] + ((cl ^ a_key[mi]) << 4));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) fw[set][ni] = *(const f16x8*)(sb + w_off[ni] + ((cl ^ w_key[ni]) << 4));
    };
#pragma unroll
    for (int ks = 0; ks < IG_PF; ++ks) load_frags(ks, ks);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      if (ks + IG_PF < BK / 16) load_frags(ks + IG_PF, (ks + IG_PF) % (IG_PF + 1));
      if constexpr (ABL & 32) {  // ablation build: keep the LDS reads, skip the matrix pipe
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(fa[ks % (IG_PF + 1)][mi]));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(fw[ks % (IG_PF + 1)][ni]));
      } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[ks % (IG_PF + 1)][ni], fa[ks % (IG_PF + 1)][mi], acc[ni][mi], 0, 0, 0);
      }
    }
  };
  // after the stage that ends the residual pass has been issued: rewind A, switch to W (caller scope: plain scalars / pointers)
  int issued = 0;
#define MGLD_AFTER_ISSUE()                                                            \
  if (two && ++issued == nk) {                                                        \
    if constexpr (FAST) {                                                             \
      if constexpr (MODE == MGLD_MODE_LINEAR) {                                       \
        _Pragma("unroll") for (int j = 0; j < JA; ++j) fa_ptr[j] -= (int64_t)nk * fa_step[j]; \
      } else {                                                                        \
        u_tap = __builtin_amdgcn_readfirstlane(s_tap);                                \
        u_c0 = __builtin_amdgcn_readfirstlane(s_c0);                                  \
      }                                                                               \
      _Pragma("unroll") for (int j = 0; j < JB; ++j) fw_ptr[j] -= (int64_t)nk * fw_step[j] + (fw_step[j] ? wdelta : 0); \
    } else {                                                                          \
      kl = k_begin + clog * 8;                                                        \
      tap = kl / Cin;                                                                 \
      c = kl - tap * Cin;                                                             \
      Wc = W;                                                                         \
    }                                                                                 \
  }
#define MGLD_SCALE_ACC()                                                              \
  {                                                                                   \
    const float sc2_ = p.w2_scale;                                                    \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                 \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                               \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[ni][mi][r] *= sc2_;        \
  }
  if constexpr (NST == 0) {
    // ---- register-staged tiles (LINEAR fast path only): global_load_dwordx4 -> VGPRs -> ds_write_b128 into the same swizzled
    // LDS image the DMA path builds.  The LDS-DMA instruction costs a wave 60-185 issue cycles per KiB (MI355X_MICROARCH.md), which
    // caps a CU at ~20-35 B/clk of staging traffic — the measured bound of the short-K projections; plain vector loads stream at
    // the L1 rate and the ds_write pass rides under the other blocks' MFMAs.
    static_assert(NST != 0 || (FAST && MODE == MGLD_MODE_LINEAR), "register staging: LINEAR fast path");
    f16x8 ra[JA], rw[JB];
    auto gload = [&]() {
#pragma unroll
      for (int j = 0; j < JA; ++j) { ra[j] = *(const f16x8*)fa_ptr[j]; fa_ptr[j] += fa_step[j]; }
#pragma unroll
      for (int j = 0; j < JB; ++j) { rw[j] = *(const f16x8*)fw_ptr[j]; fw_ptr[j] += fw_step[j]; }
    };
    auto sstore = [&](int buf) {
      char* sbase = smem + buf * STAGE + wave * 1024 + lane * 16;
#pragma unroll
      for (int j = 0; j < JA; ++j) *(f16x8*)(sbase + j * (NW * 1024)) = ra[j];
#pragma unroll
      for (int j = 0; j < JB; ++j) *(f16x8*)(sbase + BM * ROWB + j * (NW * 1024)) = rw[j];
    };
    gload();
    sstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) gload();                 // next stage in flight under this stage's MFMAs
      compute_stage(smem + cur * STAGE);
      if (kt + 1 < nk) sstore(cur ^ 1);         // (that buffer was last read in stage kt-1: every wave has passed the barrier since)
      __syncthreads();
      cur ^= 1;
    }
  } else {
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nkt) {
      issue_stage(s, u_tap, u_c0);
      MGLD_ADVANCE_TAP();
      MGLD_AFTER_ISSUE();
    }
  int cur = 0;  // buffer of stage kt
  for (int kt = 0; kt < nkt; ++kt) {
    {
      const int ahead = min(NST - 2, nkt - 1 - kt);                     // newer stages that may stay outstanding
      if (NST >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (JA + JB)) : "memory");
      else if (NST >= 3 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(JA + JB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (NST == 2) {
      __syncthreads();
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    {
      const int nxt = kt + NST - 1;
      int nb = cur + NST - 1;
      if (nb >= NST) nb -= NST;
      if (nxt < nkt) {
        issue_stage(nb, u_tap, u_c0);
        MGLD_ADVANCE_TAP();
        MGLD_AFTER_ISSUE();
      }
    }
    const char* sb = smem + cur * STAGE;
    cur = (cur + 1 == NST) ? 0 : cur + 1;
    if (two && kt == nk) MGLD_SCALE_ACC()        // residual pass done: acc = w2_scale * (A W2^T); A W^T accumulates on top
    compute_stage(sb);
  }
  }
#undef MGLD_AFTER_ISSUE
#undef MGLD_SCALE_ACC

  tile_epilogue<BM, BN, WM, WN>(p, ws, splitk, kz, bz, RowMapLinear{bm0, M}, bn0, wm, wn, wave, lane, acc, smem);
}

// ---- conv3p: 3x3 / stride 1 / pad 1 conv with the activation PATCH staged once per 32 input channels -----------------
// The im2col form above re-fetches every input pixel nine times (once per tap) through the LDS-DMA path, and that path —
// not the matrix pipe — bounds the kernel (ablation: DMA alone ~75 % of the full time, activations the larger share).
// Here a tile of BM consecutive output pixels (inside one frame) stages the contiguous raster range of input pixels
// [m0 - W - 1, m0 + BM + W + 1) ONCE per 32-channel slice (64-B LDS rows) and all nine taps read their fragments from it:
// tap (dy, dx) of output pixel i is patch row i + (dy+1)*W + (dx+1).  Rows outside the frame are zero-filled by the DMA
// (zero page), the x = 0 / x = W-1 wrap of the dx = -1 / +1 taps is removed per lane by redirecting the fragment read to
// a 64-B zero row.  Weights stream as before, one kernel row (3 taps x 32 channels) per stage, double buffered; the next
// patch arrives piecewise during the three stages of the current one.  K order: (32-channel slice, dy, dx, c).
constexpr int PB = 64;   // bytes per LDS row (32 fp16 channels)

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void conv3p_kernel(const MgldIGemm p, float* __restrict__ ws, int hchunk) {
  constexpr int WAVES_N = BN / WN;
  constexpr int NW = (BM / WM) * (BN / WN);
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int BSUB = BN * PB;                 // one tap's weight sub-tile [BN][32 ch]
  constexpr int B_BYTES = 3 * BSUB;             // weight stage: the three taps of one kernel row
  constexpr int NPB = 3 * BN / 16;              // 1-KiB DMA pieces per weight stage
  constexpr int BSLOTS = (NPB + NW - 1) / NW;
  static_assert(NW == 8, "eight waves per block");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // Workgroups are dealt round-robin to the 8 XCDs in linear-id order.  Give each XCD runs of consecutive weight tiles
  // of ONE pixel tile, so the patch is fetched into that XCD's L2 once and the other N/BN - 1 blocks hit it there
  // (-4 % on the 640 -> 320 convs of the 64x64 level, neutral elsewhere).
  int tile_m = blockIdx.x, tile_n = blockIdx.y;
  if ((gridDim.x & 7) == 0) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, j = lin >> 3;
    tile_m = xcd + 8 * (j / (int)gridDim.y);
    tile_n = j % (int)gridDim.y;
  }
  const int bm0 = tile_m * BM, bn0 = tile_n * BN;
  const bool splitk = (ws != nullptr);
  const int kz = splitk ? blockIdx.z : 0;
  const int l31 = lane & 31, lhi = lane >> 5;

  const f16* __restrict__ A = (const f16*)p.A;
  const f16* __restrict__ W = (const f16*)p.W;
  const int N = p.N, Cin = p.Cin, Wd = p.Win, HW = p.Hin * p.Win;
  const int PR = BM + 2 * Wd + 2;               // patch rows
  const int NPA = (PR + 15) >> 4;               // 1-KiB pieces (16 rows) per patch; <= 3 * NW
  const int a_bytes = NPA * 1024;
  const int b_base = 2 * a_bytes;
  const int z_off = b_base + 2 * B_BYTES;       // 64-B zero row
  const int nh = Cin >> 5;
  const int h0 = splitk ? kz * hchunk : 0;
  const int h1 = splitk ? min(nh, h0 + hchunk) : nh;
  const char* zero = (const char*)g_zero_page;

  if (tid < 16) *(unsigned*)(smem + z_off + tid * 4) = 0u;

  // ---- DMA assignment: activation piece q = s*NW + wave (slot s is issued during stage s of the previous slice) ----
  const char* fa_ptr[3];
  unsigned fa_step[3];
  {
    const int frame_lo = (bm0 / HW) * HW, frame_hi = frame_lo + HW;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int j = (s * NW + wave) * 16 + (lane >> 2);
      const int g = bm0 - (Wd + 1) + j;
      const bool ok = (j < PR) && (g >= frame_lo) && (g < frame_hi);
      const int cl = (lane & 3) ^ ((j >> 2) & 3);
      fa_ptr[s] = ok ? (const char*)(A + (int64_t)g * p.lda + h0 * 32 + cl * 8) : zero;
      fa_step[s] = ok ? 64u : 0u;
      if constexpr (ABL & 1) {   // (ablation build bit 1: same byte count from perfectly contiguous addresses — wrong data)
        fa_ptr[s] = (const char*)A + ((int64_t)(bm0 / BM) * 24 + s * NW + wave) * 1024 + lane * 16;
        fa_step[s] = 0u;
      }
    }
  }
  // weight piece b = k*NW + wave: tap column dxi = b / (BN/16), rows (b % (BN/16))*16 .. +16
  const char* fw_ptr[BSLOTS];
  bool fw_ok[BSLOTS];
#pragma unroll
  for (int k = 0; k < BSLOTS; ++k) {
    const int b = k * NW + wave;
    const int dxi = b / (BN / 16), rb = b - dxi * (BN / 16);
    const int row = rb * 16 + (lane >> 2);
    const int n = bn0 + row;
    const int cl = (lane & 3) ^ ((row >> 2) & 3);
    if (p.tap_inner == 2) {   // tiled weights [N/64][Cin/32][3 dy][4 row groups][3 dx][16 rows x 32 ch in LDS-image order]:
      // every DMA piece is one linear 1-KiB read and the 12 pieces of a (64 rows, slice, kernel row) stage are contiguous
      const int g64 = (bn0 >> 6) + (rb >> 2);
      fw_ok[k] = (g64 * 64 < ((N + 63) & ~63)) && (b < NPB);
      fw_ptr[k] = (const char*)(W + (((int64_t)g64 * nh * 3 * 4 + (rb & 3)) * 3 + dxi) * 512 + lane * 8);
    } else {
      fw_ok[k] = (n < N) && (b < NPB);
      const int64_t koff = p.tap_inner ? (int64_t)dxi * 64 : (int64_t)dxi * Cin;
      fw_ptr[k] = (const char*)(W + (int64_t)(fw_ok[k] ? n : 0) * p.ldw + koff + cl * 8);
    }
  }
  auto issue_b = [&](const int buf, const int h, const int dyi) {
    if constexpr (ABL & 16) return;
    const int64_t soff = p.tap_inner == 2 ? (int64_t)(h * 3 + dyi) * (12 * 512)
                         : p.tap_inner    ? ((int64_t)(h >> 1) * 576 + dyi * 192 + (h & 1) * 32)
                                          : ((int64_t)dyi * 3 * Cin + h * 32);
#pragma unroll
    for (int k = 0; k < BSLOTS; ++k) {
      const int b = k * NW + wave;
      if (b < NPB) {
        const char* src = fw_ok[k] ? fw_ptr[k] + soff * 2 : zero;
        if constexpr (ABL & 2)   // (ablation build bit 2: contiguous weight pieces — wrong data)
          src = (const char*)W + ((int64_t)((h * 3 + dyi) * (N / BN) + bn0 / BN) * NPB + b) * 1024 + lane * 16;
        glds16(src, smem + b_base + buf * B_BYTES + b * 1024);
      }
    }
  };
#define MGLD_ISSUE_A(S, PAR)                                                     \
  if ((S) * NW + wave < NPA) {                                                   \
    if constexpr (!(ABL & 8)) glds16(fa_ptr[S], smem + (PAR) * a_bytes + ((S) * NW + wave) * 1024); \
    fa_ptr[S] += fa_step[S];                                                     \
  }

  // ---- fragment addresses (byte offsets from smem) ----
  int a_off[MI][3];     // stage dy = -1, patch buffer 0; -1 = this lane's tap lies across the image edge
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int i = wm * WM + mi * 32 + l31;
    const int x = (bm0 + i) % Wd;
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) {
      const int j = i + dxi;
      const bool ok = !(dxi == 0 && x == 0) && !(dxi == 2 && x == Wd - 1);
      a_off[mi][dxi] = ok ? j * PB + ((lhi ^ ((j >> 2) & 3)) << 4) : -1;
    }
  }
  int w_off[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int r = wn * WN + ni * 32 + l31;
    w_off[ni] = r * PB + ((lhi ^ ((r >> 2) & 3)) << 4);
  }

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  // prologue: the whole first patch + the first weight stage.  (A third weight buffer — two stages in flight behind a
  // counted vmcnt — measured no faster and costs a resident block at W = 32.)
  if (h0 < h1) {
    MGLD_ISSUE_A(0, 0)
    MGLD_ISSUE_A(1, 0)
    MGLD_ISSUE_A(2, 0)
    issue_b(0, h0, 0);
  }
  int cur = 0;
  for (int h = h0; h < h1; ++h) {
    const int pa = (h - h0) & 1;
    const bool more = (h + 1 < h1);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (more) {
        if (s == 0) { MGLD_ISSUE_A(0, pa ^ 1) }
        if (s == 1) { MGLD_ISSUE_A(1, pa ^ 1) }
        if (s == 2) { MGLD_ISSUE_A(2, pa ^ 1) }
      }
      if (s < 2) issue_b(cur ^ 1, h, s + 1);
      else if (more) issue_b(cur ^ 1, h + 1, 0);
      if constexpr (ABL & 128) { cur ^= 1; continue; }
      const int add = s * Wd * PB + pa * a_bytes;            // W % 16 == 0 keeps the swizzle key of a shifted row
      int aaddr[MI][3];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) aaddr[mi][dxi] = a_off[mi][dxi] >= 0 ? a_off[mi][dxi] + add : z_off;
      const int bb = b_base + cur * B_BYTES;
      f16x8 fa[2][MI], fw[2][NI];
      auto load = [&](const int u, const int set) {
        const int dxi = u >> 1, kx = (u & 1) << 5;           // second 16-channel step: logical chunk ^ 2 = byte offset ^ 32
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[set][mi] = *(const f16x8*)(smem + (aaddr[mi][dxi] ^ kx));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fw[set][ni] = *(const f16x8*)(smem + bb + dxi * BSUB + (w_off[ni] ^ kx));
      };
      load(0, 0);
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        if (u + 1 < 6) load(u + 1, (u + 1) & 1);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[u & 1][ni], fa[u & 1][mi], acc[ni][mi], 0, 0, 0);
      }
      cur ^= 1;
    }
  }
#undef MGLD_ISSUE_A
  tile_epilogue<BM, BN, WM, WN>(p, ws, splitk, kz, 0, RowMapLinear{bm0, p.M}, bn0, wm, wn, wave, lane, acc, smem);
}

// ---- conv3q: the patch-staged 3x3 / stride 1 / pad 1 conv on 2-D PIXEL TILES -------------------------------------------
// conv3p above tiles the raster (BM consecutive pixels of one frame): its patch grows with the image width (BM + 2W + 2 rows), so it
// stops at W = 64, and the wrap of the dx = +-1 taps needs per-lane redirects.  Here a block owns a TY x TX pixel tile and stages the
// (TY+2) x (TX+2) input patch (with its own halo columns; out-of-image rows zero-filled by the DMA) once per 32-channel slice: any image
// size (the VAE's 128^2 .. 512^2 levels, non-square frames, ragged edges), and tap (dy, dx) of a lane's pixel is its patch row plus the
// constant dy*PW + dx — nine per-lane byte offsets computed once.  UP2 folds the nearest-2x upsample of the reference's Upsample blocks
// (openaimodel.py:185, model.py:96) into those offsets: the block stages the LOW-resolution (TY/2+2) x (TX/2+2) patch (4x fewer bytes)
// and tap (dy, dx) of output pixel (y, x) reads low-res pixel ((y+dy-1)>>1, (x+dx-1)>>1); zero padding of the upsampled image falls on
// out-of-image low-res pixels.  256-pixel tiles (16x16, 8x32) run eight waves of 64 pixels x 32 channels: 3 fragment reads per 2 MFMAs
// instead of 2 per 1 and half the weight bytes per FLOP of the 128-pixel tiles.  Weights: the tiled layout of tap_inner = 2 only.
// NWB = 3 (round 3): a THIRD weight buffer.  With two, a block has one weight stage in flight while it computes on the other and drains
// `vmcnt(0)` + barrier at every stage: the two blocks of a CU fall into step, both waiting for their DMA, then both computing (PMC round 2:
// matrix pipe busy 37 %, waves parked 44 % on vmcnt / barrier).  With three, stage g + 2 is issued during stage g, the wait at the top of a
// stage is COUNTED (`vmcnt(n)`: only what the stage reads must have landed, the newest weight stage — and the piece of the next slice's
// patch issued with it — stay in flight across the raw `s_barrier`), so a stage's DMA has two stage times to land.  Only where the LDS
// budget keeps the resident blocks per CU (8x32 x 64: 2 x 80 KiB = all 160 KiB; 16x16 x 128: one block either way; 16x16 x 64).
template <int TY, int TX, int BN, int WM, int WN, bool UP2, int PF = 1, int NWB = 2>
__global__ __launch_bounds__(64 * ((TY * TX) / WM) * (BN / WN)) void conv3q_kernel(const MgldIGemm p, float* __restrict__ ws, int hchunk,
                                                                                  int tiles_x, int tiles_y, int order) {
  constexpr int BM = TY * TX;
  constexpr int WAVES_N = BN / WN;
  constexpr int NW = (BM / WM) * WAVES_N;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int PW = UP2 ? TX / 2 + 2 : TX + 2, PH = UP2 ? TY / 2 + 2 : TY + 2;
  constexpr int PR = PW * PH;                   // patch rows (one 64-B LDS row per input pixel and slice)
  constexpr int NPA = (PR + 15) / 16;           // 1-KiB DMA pieces per patch
  constexpr int ASLOTS = (NPA + NW - 1) / NW;   // pieces per wave; slot s is issued during stage s of the previous slice
  constexpr int A_BYTES = NPA * 1024;
  constexpr int BSUB = BN * PB, B_BYTES = 3 * BSUB, NPB = 3 * BN / 16, BSLOTS = (NPB + NW - 1) / NW;
  constexpr int B_BASE = 2 * A_BYTES;
  static_assert(NW == 4 || NW == 8, "four or eight waves per block");
  static_assert(ASLOTS <= 6, "the patch must arrive within the three stages of a slice (at most two pieces per wave and stage)");
  static_assert(NWB == 2 || NWB == 3, "two or three weight buffers");
  static_assert((TX & (TX - 1)) == 0 && TX >= 8 && (TY % 2) == 0 && BM % WM == 0 && BN % 32 == 0 && BN % WN == 0, "tile shape");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // Workgroups go round-robin to the 8 XCDs in linear-id order (gridDim.x % 8 == 0: the same XCD pattern in every K-split plane).
  // order 0: the N/BN blocks sharing a PATCH run back to back on one XCD (see conv3p) — the activations leave the Infinity Cache
  // once (64x64 level: A >> W).  order 1: every XCD takes a contiguous eighth of the (weight tile major, pixel tile minor) list,
  // i.e. all pixel tiles of ~N/BN/8 weight tiles: the blocks sharing a WEIGHT tile share it through that XCD's L2 instead of
  // every XCD streaming the whole matrix (16x16 / 8x8 levels: W = 30-60 MB against 1-5 MB of activations; PMC showed 3-3.8x the
  // algorithmic bytes there).
  int tile_m = blockIdx.x, tile_n = blockIdx.y;
  if ((gridDim.x & 7) == 0) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, j = lin >> 3;
    if (order == 1) {
      const int q = xcd * ((int)(gridDim.x * gridDim.y) >> 3) + j;
      tile_n = q / (int)gridDim.x;
      tile_m = q - tile_n * (int)gridDim.x;
    } else {
      tile_m = xcd + 8 * (j / (int)gridDim.y);
      tile_n = j % (int)gridDim.y;
    }
  }
  const int tpf = tiles_x * tiles_y;
  const int frame = tile_m / tpf;
  const int trem = tile_m - frame * tpf;
  const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
  const int y0 = tyi * TY, x0 = txi * TX;       // output coordinates of the tile's first pixel
  const int bn0 = tile_n * BN;
  const bool splitk = (ws != nullptr);
  const int kz = splitk ? blockIdx.z : 0;
  const int l31 = lane & 31, lhi = lane >> 5;

  const f16* __restrict__ A = (const f16*)p.A;
  const f16* __restrict__ W = (const f16*)p.W;
  const int N = p.N, Cin = p.Cin, Hin = p.Hin, Win = p.Win;
  const int nh = Cin >> 5;
  const int h0 = splitk ? kz * hchunk : 0;
  const int h1 = splitk ? min(nh, h0 + hchunk) : nh;
  const char* zero = (const char*)g_zero_page;

  // ---- DMA assignment: activation piece q = s*NW + wave = patch rows [16q, 16q+16) ----
  const char* fa_ptr[ASLOTS];
  unsigned fa_step[ASLOTS];
  {
    const int yb = (UP2 ? (y0 >> 1) : y0) - 1, xb = (UP2 ? (x0 >> 1) : x0) - 1;
#pragma unroll
    for (int s = 0; s < ASLOTS; ++s) {
      const int j = (s * NW + wave) * 16 + (lane >> 2);
      const int pr = j / PW, pc = j - pr * PW;
      const int y = yb + pr, x = xb + pc;
      const bool ok = (j < PR) && ((unsigned)y < (unsigned)Hin) && ((unsigned)x < (unsigned)Win);
      const int cl = (lane & 3) ^ ((j >> 2) & 3);
      fa_ptr[s] = ok ? (const char*)(A + (((int64_t)frame * Hin + y) * Win + x) * p.lda + h0 * 32 + cl * 8) : zero;
      fa_step[s] = ok ? 64u : 0u;
    }
  }
  // weight piece b = k*NW + wave: tap column dxi = b / (BN/16), rows (b % (BN/16))*16 .. +16 of the tiled layout
  const char* fw_ptr[BSLOTS];
  bool fw_ok[BSLOTS];
#pragma unroll
  for (int k = 0; k < BSLOTS; ++k) {
    const int b = k * NW + wave;
    const int dxi = b / (BN / 16), rb = b - dxi * (BN / 16);
    const int gr = (bn0 >> 4) + rb;             // 16-row group of the weight matrix (BN = 160 tiles start inside a 64-row group)
    const int g64 = gr >> 2;
    fw_ok[k] = (g64 * 64 < ((N + 63) & ~63)) && (b < NPB);
    fw_ptr[k] = (const char*)(W + (((int64_t)g64 * nh * 3 * 4 + (gr & 3)) * 3 + dxi) * 512 + lane * 8);
  }
  // W2 (MgldIGemm): the slices are walked TWICE — first against the scaled fp16 residual of the weights (same tiled layout, `wdelta` bytes
  // away), then, after ONE multiplication of the accumulators by w2_scale, against the weights themselves.  `lo` selects the matrix.
  const bool two = (p.W2 != nullptr);
  const int64_t wdelta = two ? (const char*)p.W2 - (const char*)p.W : 0;
  auto issue_b = [&](const int buf, const int h, const int dyi, const bool lo) {
    const int64_t soff = (int64_t)(h * 3 + dyi) * (12 * 512) * 2 + (lo ? wdelta : 0);
#pragma unroll
    for (int k = 0; k < BSLOTS; ++k) {
      const int b = k * NW + wave;
      if (b < NPB) {
        const char* src = fw_ok[k] ? fw_ptr[k] + soff : zero;
        glds16(src, smem + B_BASE + buf * B_BYTES + b * 1024);
      }
    }
  };
#define MGLD_Q_ISSUE_A(S, PAR)                                                    \
  if constexpr ((S) < ASLOTS) {                                                   \
    if ((S) * NW + wave < NPA) {                                                  \
      glds16(fa_ptr[S], smem + (PAR) * A_BYTES + ((S) * NW + wave) * 1024);       \
      fa_ptr[S] += fa_step[S];                                                    \
    }                                                                             \
  }

  // ---- fragment addresses: byte offset of this lane's patch row for each of the nine taps (first 16-channel step) ----
  int a_off[MI][9];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int r = wm * WM + mi * 32 + l31;
    const int ty = r / TX, tx = r & (TX - 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dyi = t / 3, dxi = t - dyi * 3;
      const int j = UP2 ? (((ty + dyi - 1) >> 1) + 1) * PW + ((tx + dxi - 1) >> 1) + 1 : (ty + dyi) * PW + tx + dxi;
      a_off[mi][t] = j * PB + ((lhi ^ ((j >> 2) & 3)) << 4);
    }
  }
  int w_off[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int r = wn * WN + ni * 32 + l31;
    w_off[ni] = r * PB + ((lhi ^ ((r >> 2) & 3)) << 4);
  }

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  const int ns = h1 - h0;                     // channel slices of this block (of this K split)
  const int nv = two ? 2 * ns : ns;           // slice visits: residual pass, then main pass
  // weight-stage issue cursor: stage (visit iv, kernel row idy) of slice ih goes into ring buffer ibuf (plain scalars in this scope)
  int ih = h0, idy = 0, iv = 0, ibuf = 0;
#define MGLD_Q_ISSUE_W()                                                          \
  if (iv < nv) {                                                                  \
    issue_b(ibuf, ih, idy, two && iv < ns);                                       \
    ibuf = (ibuf + 1 == NWB) ? 0 : ibuf + 1;                                      \
    if (++idy == 3) { idy = 0; ++iv; ih = (two && iv == ns) ? h0 : ih + 1; }      \
  }
  // DMA instructions this wave issues per weight stage / per patch slot (the counted waits of the three-buffer ring)
  int nw_me = 0, na_me[3] = {0, 0, 0};       // (patch pieces of stage s: slots s and s + 3)
#pragma unroll
  for (int k = 0; k < BSLOTS; ++k) nw_me += (k * NW + wave < NPB) ? 1 : 0;
#pragma unroll
  for (int s = 0; s < ASLOTS; ++s) na_me[s % 3] += (s * NW + wave < NPA) ? 1 : 0;
  if (h0 < h1) {
    MGLD_Q_ISSUE_A(0, 0)
    MGLD_Q_ISSUE_A(1, 0)
    MGLD_Q_ISSUE_A(2, 0)
    MGLD_Q_ISSUE_A(3, 0)
    MGLD_Q_ISSUE_A(4, 0)
    MGLD_Q_ISSUE_A(5, 0)
    MGLD_Q_ISSUE_W()
    if constexpr (NWB == 3) { MGLD_Q_ISSUE_W() }
  }
  int cur = 0;
  for (int v = 0; v < nv; ++v) {
    const int pa = v & 1;
    const bool more = (v + 1 < nv);
    const bool wrap = two && (v + 1 == ns);   // the next visit starts the main pass: its patch is slice h0 again
    if (two && v == ns) {                     // residual pass done: acc = w2_scale * (A W2^T), then A W^T accumulates on top
      const float sc2 = p.w2_scale;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ni][mi][r] *= sc2;
    }
    if (wrap) {
#pragma unroll
      for (int s = 0; s < ASLOTS; ++s) fa_ptr[s] -= (int64_t)ns * fa_step[s];
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if constexpr (NWB == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      } else {
        // in flight, oldest first: ..., W[g] (issued two stages ago), then last stage's issues: [patch slot s-1 of the NEXT slice, W[g+1]].
        // This stage reads W[g] and (s == 0) the whole patch, whose last slot is one of last stage's issues: allow W[g+1], and
        // for s != 0 also last stage's patch piece, to stay in flight.
        int allow = (3 * v + s + 1 < 3 * nv) ? nw_me : 0;
        if (s != 0 && more) allow += na_me[s - 1];
        switch (allow) {
          case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
          case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
          case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
          case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
          case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
          case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
          case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
          case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
          case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;     // (8 weight + 2 patch pieces: the 160-row variant)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // raw barrier: __syncthreads() would drain the DMA queue (vmcnt(0)) first
      }
      if (more) {     // (four-wave blocks with large patches carry two pieces per wave and stage: slots s and s + 3)
        if (s == 0) { MGLD_Q_ISSUE_A(0, pa ^ 1) MGLD_Q_ISSUE_A(3, pa ^ 1) }
        if (s == 1) { MGLD_Q_ISSUE_A(1, pa ^ 1) MGLD_Q_ISSUE_A(4, pa ^ 1) }
        if (s == 2) { MGLD_Q_ISSUE_A(2, pa ^ 1) MGLD_Q_ISSUE_A(5, pa ^ 1) }
      }
      MGLD_Q_ISSUE_W()                         // NWB = 2: the next stage, into the buffer read last stage; NWB = 3: the one after
      const int abase = pa * A_BYTES;
      const int bb = B_BASE + cur * B_BYTES;
      // fragments of step u + PF are fetched from LDS while the MFMAs of step u run (PF + 1 register sets, static indices);
      // PF = 2 gives a ds_read_b128 two MFMA groups (~128 issue cycles) instead of one to land
      f16x8 fa[PF + 1][MI], fw[PF + 1][NI];
      auto load = [&](const int u, const int set) {
        const int dxi = u >> 1, kx = (u & 1) << 5;           // second 16-channel step: logical chunk ^ 2 = byte offset ^ 32
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[set][mi] = *(const f16x8*)(smem + abase + (a_off[mi][s * 3 + dxi] ^ kx));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fw[set][ni] = *(const f16x8*)(smem + bb + dxi * BSUB + (w_off[ni] ^ kx));
      };
#pragma unroll
      for (int u = 0; u < PF; ++u) load(u, u);
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        if (u + PF < 6) load(u + PF, (u + PF) % (PF + 1));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[u % (PF + 1)][ni], fa[u % (PF + 1)][mi], acc[ni][mi], 0, 0, 0);
      }
      cur = (cur + 1 == NWB) ? 0 : cur + 1;
    }
  }
#undef MGLD_Q_ISSUE_A
#undef MGLD_Q_ISSUE_W
  tile_epilogue<BM, BN, WM, WN>(p, ws, splitk, kz, 0, RowMap2D<TX>{frame * p.Hout * p.Wout, y0, x0, p.Hout, p.Wout}, bn0, wm, wn, wave,
                                lane, acc, smem);
}

// split-K finish: out = alpha*act(sum_z ws[z] + bias + bias_m + rowvec) + beta*R.  One thread per 4 columns.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const MgldIGemm p, const float* __restrict__ ws, int splits) {
  const int M = p.M, N = p.N;
  const int nq = (N + 3) >> 2;
  const int64_t total = (int64_t)M * nq;
  const int64_t MN = (int64_t)M * N;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int m = (int)(idx / nq);
    const int n0 = (int)(idx - (int64_t)m * nq) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vec = (n0 + 3 < N) && ((N & 3) == 0);
    for (int z = 0; z < splits; ++z) {
      const float* src = ws + z * MN + (int64_t)m * N + n0;
      if (vec) {
        const f32x4 t = *(const f32x4*)src;
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
      } else {
        for (int j = 0; j < 4; ++j) if (n0 + j < N) v[j] += src[j];
      }
    }
    const float bm = p.bias_m ? p.bias_m[m] : 0.f;
    const float* rv = p.rowvec ? p.rowvec + (int64_t)(m / p.rows_per_frame) * p.ld_rowvec : nullptr;
    const f16* R = (const f16*)p.R;
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + j;
      if (n >= N) break;
      float x = v[j] + bm;
      if (p.bias) x += p.bias[n];
      if (rv) x += rv[n];
      x = apply_act(x, p.act) * p.alpha;
      if (R) x += p.beta * (float)R[(int64_t)m * p.ldr + n];
      if (p.out_f32) ((float*)p.C)[(int64_t)m * p.ldc + n] = x;
      else ((f16*)p.C)[(int64_t)m * p.ldc + n] = (f16)x;
    }
  }
}

int num_cus() {
  static int v = 0;
  if (!v) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n <= 0)
      n = 256;
    v = n;
  }
  return v;
}

// split-K workspace (mgld_set_workspace).  Per HOST THREAD: a thread drives one stream, so two threads that keep two segments in flight
// on one GPU (bench.py --inflight 2) each register their own scratch and their concurrent launches never share slabs.
thread_local float* g_ws = nullptr;
thread_local size_t g_ws_bytes = 0;

// XCD-aware tile order of igemm_kernel (see the kernel): 0 = dispatch order, 1 = A-sharing blocks on one XCD, 2 = W-sharing blocks
// on one XCD.  env MGLD_IGEMM_ORDER = 0 / 1 / 2 forces (A/B runs); default: by which operand is re-fetched more.
inline int tile_order(const MgldIGemm* p, int gx, int gy) {
  static int force = -2;
  if (force < -1) { const char* e = getenv("MGLD_IGEMM_ORDER"); force = e ? atoi(e) : -1; }
  int o = force;
  if (o < 0) {
    // bytes each order re-reads beyond one XCD's L2: order 1 streams W into every XCD, order 2 streams A into every XCD
    const double a_bytes = 2.0 * p->M * (p->mode == MGLD_MODE_LINEAR ? p->K : p->Cin), w_bytes = 2.0 * p->N * p->K;
    // measured (tools/igemm_bench.py lin, cold operands): neither order beats dispatch order on the UNet's projections (order 1
    // -3 %, order 2 +-1 %): the re-reads they remove are served by the Infinity Cache at no cost in time.  Kept for A/B runs.
    (void)a_bytes; (void)w_bytes;
    o = 0;
  }
  if (o == 1 && ((gx & 7) || gy < 2)) o = 0;
  if (o == 2 && ((gy & 7) || gx < 2)) o = 0;
  return o;
}

template <int MODE, bool FAST, int BM, int BN, int WM, int WN, int NST>
void launch_fast(const MgldIGemm* p, hipStream_t s, int splits, int kchunk) {
  constexpr int LDS = (NST == 0 ? 2 : NST) * (BM + BN) * ROWB;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)igemm_kernel<MODE, FAST, BM, BN, WM, WN, NST>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  constexpr int THREADS = 64 * (BM / WM) * (BN / WN);
  const int gz = splits > 1 ? splits : (p->batch > 0 ? p->batch : 1);
  dim3 grid(cdiv(p->M, BM), cdiv(p->N, BN), gz);
  hipLaunchKernelGGL((igemm_kernel<MODE, FAST, BM, BN, WM, WN, NST>), grid, dim3(THREADS), LDS, s, *p,
                     splits > 1 ? g_ws : nullptr, kchunk, tile_order(p, (int)grid.x, (int)grid.y));
}

// FAST: every 64-deep stage lies inside one tap and inside K (see the kernel)
inline bool fast_ok(const MgldIGemm* p) {
  if (p->K % BK) return false;
  if (p->mode == MGLD_MODE_LINEAR) return true;
  if (p->mode == MGLD_MODE_CONV3X3 && p->kh > 0 && !(p->kh == 3 && p->kw == 3)) return false;   // generic taps: per-lane path
  return (p->Cin % BK) == 0 && !(p->mode == MGLD_MODE_CONV3X3 && p->up2);
}

template <int MODE, int BM, int BN, int WM, int WN, int NST>
void launch_mode(const MgldIGemm* p, hipStream_t s, int splits, int kchunk) {
  if (fast_ok(p)) launch_fast<MODE, true, BM, BN, WM, WN, NST>(p, s, splits, kchunk);
  else launch_fast<MODE, false, BM, BN, WM, WN, NST>(p, s, splits, kchunk);
}

// ring depth of the LINEAR fast path.  Measured on the transformer blocks' projections (tools/igemm_bench.py lin --nst 1,2,3, MI355X,
// profiles/r02_linear_ring_depth.txt): 3- and 4-deep rings (counted vmcnt, 1-2 stages in flight across the stage barrier) are SLOWER
// than the 2-deep ring on every shape but M = 512 (178 -> 227 -> 260 ms per segment): what hides the DMA round trip is the number
// of resident blocks per CU, and a deeper ring trades exactly that away.  Default 2; env MGLD_IGEMM_NST = 2 / 3 / 4 or
// p->tune = depth - 1 select a deeper ring (tests, tuning runs).
inline int linear_ring_depth(const MgldIGemm* p, int BM, int BN) {
  static int force = -1;
  if (force < 0) { const char* e = getenv("MGLD_IGEMM_NST"); force = e ? atoi(e) : 0; }
  int nst = force ? force : 2;
  if (p->tune > 0 && p->tune < 9) nst = p->tune + 1;
  if (nst < 2) nst = 2;
  if (nst > 4) nst = 4;
  while (nst > 2 && nst * (BM + BN) * ROWB > 160 * 1024) --nst;
  const int nk = (p->K + BK - 1) / BK;
  if (nst > nk) nst = nk < 2 ? 2 : nk;
  return nst;
}

template <int BM, int BN, int WM, int WN, int NST = 2>
int launch_cfg(const MgldIGemm* p, hipStream_t s, int splits, int kchunk) {
  if (p->mode == MGLD_MODE_LINEAR && fast_ok(p) && splits <= 1) {
    static int rs = -1;   // env MGLD_IGEMM_RS = 1: register-staged tiles on the LINEAR fast path (p->tune = 9 selects them per launch)
    if (rs < 0) { const char* e = getenv("MGLD_IGEMM_RS"); rs = e ? atoi(e) : 0; }
    if (((rs && p->tune == 0) || p->tune == 9) && !p->W2) {
      launch_fast<MGLD_MODE_LINEAR, true, BM, BN, WM, WN, 0>(p, s, splits, kchunk);
      if (splits > 1) {}
      return mgld_check_launch("igemm");
    }
    const int nst = linear_ring_depth(p, BM, BN);
    if (nst == 4) launch_fast<MGLD_MODE_LINEAR, true, BM, BN, WM, WN, 4>(p, s, splits, kchunk);
    else if (nst == 3) launch_fast<MGLD_MODE_LINEAR, true, BM, BN, WM, WN, 3>(p, s, splits, kchunk);
    else launch_fast<MGLD_MODE_LINEAR, true, BM, BN, WM, WN, 2>(p, s, splits, kchunk);
  } else if (p->mode == MGLD_MODE_LINEAR) launch_mode<MGLD_MODE_LINEAR, BM, BN, WM, WN, NST>(p, s, splits, kchunk);
  else if (p->mode == MGLD_MODE_CONV3X3) launch_mode<MGLD_MODE_CONV3X3, BM, BN, WM, WN, NST>(p, s, splits, kchunk);
  else launch_mode<MGLD_MODE_TCONV3, BM, BN, WM, WN, NST>(p, s, splits, kchunk);
  if (splits > 1) {
    const int64_t total = (int64_t)p->M * ((p->N + 3) >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, *p, g_ws, splits);
  }
  return mgld_check_launch("igemm");
}

// (config, splits): config encoded BM*1000+BN
void choose(const MgldIGemm* p, int* cfg, int* splits, int* kchunk) {
  const int64_t M = p->M, N = p->N, K = p->K;
  const int batch = p->batch > 0 ? p->batch : 1;
  const int64_t t128 = (int64_t)cdiv(M, 128) * cdiv(N, 128) * batch;
  *splits = 1;
  *kchunk = (int)K;
  {
    static int force = -1;   // env MGLD_IGEMM_FORCE=<BM*1000+BN>: tile override for tuning runs (non-GEGLU, N > 64)
    if (force < 0) { const char* e = getenv("MGLD_IGEMM_FORCE"); force = e ? atoi(e) : 0; }
    if (force && p->act != MGLD_ACT_GEGLU && N > 64) { *cfg = force; return; }
  }
  // full-N tiles (128 x 320, eight waves of 32 x 160) for the 320-channel projections of the 64x64 level: the activation rows enter LDS
  // ONCE instead of once per 64-column tile, and M / 128 = 256 blocks is exactly one per CU.  Measured (profiles/r02_linear_ring_depth.txt):
  // isolated launches 18.5 vs 19.4 us (K = 320) and 40.6 vs 50.5 us (K = 1280), but the whole segment does not move (801 vs 797 ms:
  // in the pipeline the producer leaves the rows in L2 / Infinity Cache and one block per CU exposes its prologue and epilogue), so
  // the planner does not pick them: p->tune = 10 or env MGLD_IGEMM_FULLN=1 selects them, p->tune = 11 forbids them (tests).
  {
    static int fulln = -1;
    if (fulln < 0) { const char* e = getenv("MGLD_IGEMM_FULLN"); fulln = e ? atoi(e) : 0; }
    const bool can = p->mode == MGLD_MODE_LINEAR && N == 320 && (K % BK) == 0 && batch == 1 && p->act != MGLD_ACT_GEGLU;
    if (can && (p->tune == 10 || (fulln && p->tune != 11 && M >= 16384 && (M % 128) == 0))) { *cfg = 128320; return; }
  }
  if (p->act == MGLD_ACT_GEGLU) { *cfg = (t128 >= 256 || M <= 64) ? 128128 : 64128; return; }
  if (N <= 32) { *cfg = 128032; return; }
  if (N <= 64) { *cfg = 128064; return; }
  // N = 64 (mod 128), e.g. the 320-channel level: 128-wide tiles would idle a sixth of the MFMA work on padding, 64-wide
  // tiles divide N exactly and fit three blocks per CU
  if ((N & 127) == 64 && N <= 448 && (int64_t)cdiv(M, 128) * (N / 64) * batch >= 512) { *cfg = 128064; return; }
  if (t128 >= 384) { *cfg = 128128; return; }
  // too few 128x128 tiles for 256 CUs.  Between 1 and 1.5 tiles per CU (e.g. M = 8192, N = 640) half-size tiles balance
  // the CUs exactly as well as a 2-way K split (3 rounds of half the work) and need no reduce pass.
  if (t128 >= 256) { *cfg = (t128 % num_cus() == 0) ? 128128 : 64128; return; }   // exactly one tile per CU: keep the big tile
  // Fewer 128x128 tiles than CUs: pick (tile, K split) by a small cost model calibrated on MI355X (us):
  //   throughput term  rounds over 256 CUs x tile area x k-steps per block x 0.7 us (0.9 us when a CU holds a single block),
  //   latency term     k-steps per block x 0.55 us (one block cannot go faster however small its tile),
  //   split-K          + launch of the reduce pass + (s+1) fp32 passes over the M x N output at ~3.5 TB/s.
  // shallow K (<= 48 k-steps): half / quarter tiles already give every CU a block and finish before a split + reduce would
  if (K <= 48 * BK) {
    const int64_t t64 = (int64_t)cdiv(M, 64) * cdiv(N, 64) * batch, t64x128s = (int64_t)cdiv(M, 64) * cdiv(N, 128) * batch;
    if (t64 >= 2 * num_cus()) { *cfg = 64064; return; }
    if (t64x128s >= num_cus()) { *cfg = 64128; return; }
  }
  // deep K, fewer 128x128 tiles than resident blocks: split K over grid.z (about 448 blocks in all), fp32 slabs + reduce pass
  if (batch == 1 && K >= 1536 && g_ws != nullptr) {
    int s = (int)((448 + t128 - 1) / t128);
    const int smax = (int)(K / 512);
    if (s > smax) s = smax;
    if (s > 16) s = 16;
    if (s >= 2 && (size_t)s * M * N * sizeof(float) <= g_ws_bytes) {
      int kc = (int)((K + s - 1) / s);
      kc = (kc + BK - 1) / BK * BK;
      s = (int)((K + kc - 1) / kc);
      if (s >= 2) { *cfg = 128128; *splits = s; *kchunk = kc; return; }
    }
  }
  const int64_t t64x128 = (int64_t)cdiv(M, 64) * cdiv(N, 128) * batch;
  *cfg = (t64x128 >= 384) ? 64128 : 64064;
}

// ---- conv3p launch plan -------------------------------------------------------------------------------------------
constexpr int C3P_BM = 128;
inline int conv3p_lds(int Win, int BN) { return 2 * ((C3P_BM + 2 * Win + 2 + 15) >> 4) * 1024 + 2 * 3 * BN * PB + 64; }

// true when the problem takes the patch kernel; *bn = weight tile rows, *splits / *hchunk = K split in 32-channel slices
bool conv3p_plan(const MgldIGemm* p, int* bn, int* splits, int* hchunk) {
  static int knob = -1, fsplit = -1;   // env MGLD_CONV3P: 0 = off, 64 / 128 = force the weight tile; MGLD_CONV3P_SPLITS (tuning)
  if (knob < 0) { const char* e = getenv("MGLD_CONV3P"); knob = e ? atoi(e) : 1; }
  if (fsplit < 0) { const char* e = getenv("MGLD_CONV3P_SPLITS"); fsplit = e ? atoi(e) : 0; }
  if (!knob || p->mode != MGLD_MODE_CONV3X3) return false;
  if (p->kh > 0 && !(p->kh == 3 && p->kw == 3)) return false;
  if (p->stride != 1 || p->up2 || p->pad_t != 1 || p->pad_l != 1 || p->Hin != p->Hout || p->Win != p->Wout) return false;
  if ((p->Win & 15) || p->Win > 64 || (p->Hin * p->Win) % C3P_BM) return false;
  if ((p->Cin & 31) || (p->tap_inner == 1 && (p->Cin & 63)) || p->batch > 1 || p->N <= 32 || p->act == MGLD_ACT_GEGLU) return false;
  const int N = p->N;
  // 64 weight rows: W = 64 (a 128-row stage pair would not leave LDS for two blocks per CU), W = 32 (three blocks per CU
  // instead of two: measured faster), and N = 64 (mod 128); 128 rows at W = 16
  int BN = (N <= 64 || ((N & 127) == 64 && N <= 448) || p->Win >= 32) ? 64 : 128;
  if (knob == 64 || knob == 128) BN = knob;
  const int lds = conv3p_lds(p->Win, BN);
  if (lds > 160 * 1024) return false;
  const int64_t tiles = (int64_t)(p->M / C3P_BM) * cdiv(N, BN);
  const int slots = num_cus() * ((160 * 1024) / lds);
  const int nh = p->Cin >> 5;
  int s = fsplit > 0 ? fsplit : (int)(slots / tiles);
  if (s > nh / 4) s = nh / 4;
  if (s > 16) s = 16;
  if (s < 2 || g_ws == nullptr || (size_t)s * p->M * N * sizeof(float) > g_ws_bytes) s = 1;
  int hc = (nh + s - 1) / s;
  s = (nh + hc - 1) / hc;
  *bn = BN; *splits = s; *hchunk = hc;
  return true;
}

template <int BN, int WM, int WN>
int launch_conv3p(const MgldIGemm* p, hipStream_t s, int splits, int hchunk) {
  const int lds = conv3p_lds(p->Win, BN);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv3p_kernel<C3P_BM, BN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  dim3 grid(p->M / C3P_BM, cdiv(p->N, BN), splits > 1 ? splits : 1);
  hipLaunchKernelGGL((conv3p_kernel<C3P_BM, BN, WM, WN>), grid, dim3(512), lds, s, *p, splits > 1 ? g_ws : nullptr, hchunk);
  if (splits > 1) {
    const int64_t total = (int64_t)p->M * ((p->N + 3) >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, *p, g_ws, splits);
  }
  return mgld_check_launch("igemm(conv3p)");
}


// ---- conv3q launch plan -------------------------------------------------------------------------------------------
// variants (id): tile TY x TX pixels, BN weight rows, wave tile WM pixels x WN channels
//   0: 8x16 x 64, 32x32 (8 waves)      1: 16x16 x 64, 64x32 (8 waves)     2: 8x16 x 128, 64x32 (8 waves)
//   3: 16x16 x 128, 64x64 (8 waves)    4: 8x32 x 64, 64x32 (8 waves)      5: 8x16 x 64, 64x32 (4 waves)
//   6: 8x8 x 128, 32x32 (8 waves): the 8x8 UNet level, one tile per frame
//   7: 8x32 x 64, 64x64 (4 waves)      8: 8x32 x 128, 64x64 (8 waves)    (round 3: 2 x 2 MFMA tiles per wave = 1 KiB of LDS fragment
//      reads per MFMA instead of 1.5: the 64x32 wave tiles run at the LDS read bandwidth)
//   9: 8x32 x 160, 64x160 (4 waves, ONE block per CU, three weight buffers): the N = 320 convolutions of the 64x64 UNet level as
//      128 pixel tiles x 2 column tiles = exactly one block per CU (no partial round: the 64-row tiles leave 640 blocks on 512 slots);
//      10 MFMAs per 7 fragment reads (0.7 KiB of LDS per MFMA), 160 accumulator registers per lane
// (fragments prefetched TWO steps ahead — template parameter PF = 2 — measured identical to PF = 1 on every shape: not instantiated)
constexpr int Q3_NVAR = 10;
template <int TY, int TX, int BN, int WM, int WN, bool UP2, int NWB = 2>
constexpr int conv3q_lds() {
  constexpr int PW = UP2 ? TX / 2 + 2 : TX + 2, PH = UP2 ? TY / 2 + 2 : TY + 2;
  constexpr int NPA = (PW * PH + 15) / 16, NW = (TY * TX / WM) * (BN / WN);
  constexpr int stages = 2 * NPA * 1024 + NWB * 3 * BN * PB, epi = NW * 32 * (WN + 4) * 4;
  return stages > epi ? stages : epi;
}
// weight buffers of variant `id`: env MGLD_CONV3Q_NWB = 3 selects the three-buffer ring for the variants whose LDS budget keeps the
// resident blocks per CU (see the kernel); default two
inline int q3_nwb(int id, bool up2) {
  static int force = -1;
  if (force < 0) { const char* e = getenv("MGLD_CONV3Q_NWB"); force = e ? atoi(e) : 0; }
  // measured (profiles/r03_conv3q_nwb.txt): the third buffer is 1-3 % slower launch by launch, 0.5 % end to end -> opt-in only
  if (force != 3 || up2) return 2;
  return (id == 1 || id == 3 || id == 4) ? 3 : 2;
}
// weight buffers of the 160-row variant (one block per CU: nothing else on the CU covers a DMA round trip): env MGLD_CONV3Q_NWB160 = 2 / 3
inline int q3_nwb160() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MGLD_CONV3Q_NWB160"); v = e ? atoi(e) : 3; }
  return v == 2 ? 2 : 3;
}
// nearest-2x fold with four 64x64 waves per 16x16 tile (variant 7 with up2): env MGLD_CONV3Q_UP2W64 = 0 / 1 (A/B)
inline bool q3_up2_wave64() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MGLD_CONV3Q_UP2W64"); v = e ? atoi(e) : 0; }
  return v != 0;
}
inline void q3_geom(int id, int* ty, int* tx, int* bn, int* lds, bool up2) {
  const bool w3 = q3_nwb(id, up2) == 3;
  if (up2 && id == 7) { *ty = 16; *tx = 16; *bn = 64; *lds = conv3q_lds<16, 16, 64, 64, 64, true>(); return; }
  switch (id) {
    case 1: *ty = 16; *tx = 16; *bn = 64; *lds = up2 ? conv3q_lds<16, 16, 64, 64, 32, true>() : (w3 ? conv3q_lds<16, 16, 64, 64, 32, false, 3>() : conv3q_lds<16, 16, 64, 64, 32, false>()); break;
    case 2: *ty = 8; *tx = 16; *bn = 128; *lds = conv3q_lds<8, 16, 128, 64, 32, false>(); break;
    case 3: *ty = 16; *tx = 16; *bn = 128; *lds = w3 ? conv3q_lds<16, 16, 128, 64, 64, false, 3>() : conv3q_lds<16, 16, 128, 64, 64, false>(); break;
    case 4: *ty = 8; *tx = 32; *bn = 64; *lds = w3 ? conv3q_lds<8, 32, 64, 64, 32, false, 3>() : conv3q_lds<8, 32, 64, 64, 32, false>(); break;
    case 5: *ty = 8; *tx = 16; *bn = 64; *lds = conv3q_lds<8, 16, 64, 64, 32, false>(); break;
    case 6: *ty = 8; *tx = 8; *bn = 128; *lds = conv3q_lds<8, 8, 128, 32, 32, false>(); break;
    case 7: *ty = 8; *tx = 32; *bn = 64; *lds = conv3q_lds<8, 32, 64, 64, 64, false>(); break;
    case 8: *ty = 8; *tx = 32; *bn = 128; *lds = conv3q_lds<8, 32, 128, 64, 64, false>(); break;
    case 9: *ty = 8; *tx = 32; *bn = 160; *lds = q3_nwb160() == 3 ? conv3q_lds<8, 32, 160, 64, 160, false, 3>() : conv3q_lds<8, 32, 160, 64, 160, false>(); break;
    default: *ty = 8; *tx = 16; *bn = 64; *lds = up2 ? conv3q_lds<8, 16, 64, 32, 32, true>() : conv3q_lds<8, 16, 64, 32, 32, false>(); break;
  }
}

// true when the problem takes the 2-D-tile patch kernel (tiled weights, tap_inner = 2); *id = variant, *splits / *hchunk = K split
bool conv3q_plan(const MgldIGemm* p, int* id, int* splits, int* hchunk) {
  static int knob = -1, force = -2, fsplit = -1;   // env MGLD_CONV3Q=0: off; MGLD_CONV3Q_FORCE=<id>; MGLD_CONV3P_SPLITS (tuning)
  if (knob < 0) { const char* e = getenv("MGLD_CONV3Q"); knob = e ? atoi(e) : 1; }
  if (force < -1) { const char* e = getenv("MGLD_CONV3Q_FORCE"); force = e ? atoi(e) : -1; }
  if (fsplit < 0) { const char* e = getenv("MGLD_CONV3P_SPLITS"); fsplit = e ? atoi(e) : 0; }
  if (!knob || p->mode != MGLD_MODE_CONV3X3 || p->tap_inner != 2) return false;
  if (p->kh > 0 && !(p->kh == 3 && p->kw == 3)) return false;
  if (p->stride != 1 || p->pad_t != 1 || p->pad_l != 1 || p->batch > 1 || p->N <= 32 || p->act == MGLD_ACT_GEGLU || (p->Cin & 31)) return false;
  const int sc = p->up2 ? 2 : 1;
  if (p->Hout != sc * p->Hin || p->Wout != sc * p->Win || p->Wout < 8 || p->Hout < 8 || (p->M % (p->Hout * p->Wout))) return false;
  if (p->Wout < 16 && (p->up2 || p->Wout != 8 || p->Hout != 8)) return false;     // below 16 pixels: only the 8x8 level
  const int frames = p->M / (p->Hout * p->Wout), N = p->N, nh = p->Cin >> 5;
  // variant by measurement (tools/igemm_bench.py on MI355X, cold operands; profiles/r02_conv3q_variants.txt):
  //   nearest-2x fold: 16x16 tiles (low-res patch 10x10) while they give ~2 blocks per CU (209 vs 241 us on 640 -> 640 at 32 -> 64), else 8x16;
  //   16x16 frames with N % 128 == 0: one 16x16 tile = the whole frame, 128 weight rows, 64x64 wave tiles;
  //   W >= 32: 8x32 tiles (256 pixels, conflict-free fragment reads) while they still give ~2 blocks per CU, else 8x16 tiles run by
  //   four waves of 64 pixels x 32 channels (3 blocks per CU).
  const int64_t t832 = (int64_t)frames * cdiv(p->Hout, 8) * cdiv(p->Wout, 32) * cdiv(N, 64);
  const int64_t t256 = (int64_t)frames * cdiv(p->Hout, 16) * cdiv(p->Wout, 16) * cdiv(N, 64);
  int v;
  if (p->up2) v = (t256 >= 448) ? (q3_up2_wave64() ? 7 : 1) : 0;
  else if (p->Wout == 8) v = 6;
  else if (p->Wout == 16 && p->Hout == 16 && (N & 127) == 0) v = 3;
  // round 3 (profiles/r03_conv3q_variants.txt): 8x32 tiles run by FOUR waves of 64 pixels x 64 channels (2 x 2 MFMA tiles per wave: a third
  // less LDS fragment traffic per MFMA than the eight 64x32 waves of variant 4) are 5-10 % faster wherever there is at least ~1.25 block per CU
  else v = (p->Wout >= 32 && t832 >= 320) ? 7 : 5;
  {
    // the 160-row one-block-per-CU variant where its blocks fill whole rounds of the chip (N = 320 / 640 at 64x64, 8 frames)
    static int v160 = -1;
    if (v160 < 0) { const char* e = getenv("MGLD_CONV3Q_V160"); v160 = e ? atoi(e) : 0; }
    if (v160 && !p->up2 && p->Wout >= 32 && (N % 160) == 0) {
      const int64_t t160 = (int64_t)frames * cdiv(p->Hout, 8) * cdiv(p->Wout, 32) * (N / 160);
      const int64_t cus = num_cus(), rem = t160 % cus;
      if (t160 >= cus * 3 / 4 && (rem == 0 || rem >= cus * 3 / 4)) v = 9;
    }
  }
  if (force >= 0 && force < Q3_NVAR && !(p->up2 && force > 1 && force != 7) && p->Wout >= 16 && force != 6) v = force;
  if (p->tune > 0 && p->tune <= Q3_NVAR && !(p->up2 && p->tune > 2 && p->tune != 8) && p->Wout >= 16 && p->tune != 7) v = p->tune - 1;
  int ty, tx, bn, lds;
  q3_geom(v, &ty, &tx, &bn, &lds, p->up2 != 0);
  const int64_t tiles = (int64_t)frames * cdiv(p->Hout, ty) * cdiv(p->Wout, tx) * cdiv(N, bn);
  const int slots = num_cus() * ((160 * 1024) / lds);
  int s = fsplit > 0 ? fsplit : (int)(slots / tiles);
  if (s > nh / 4) s = nh / 4;
  if (s > 16) s = 16;
  if (s < 2 || g_ws == nullptr || (size_t)s * p->M * N * sizeof(float) > g_ws_bytes) s = 1;
  int hc = (nh + s - 1) / s;
  s = (nh + hc - 1) / hc;
  *id = v; *splits = s; *hchunk = hc;
  return true;
}

template <int TY, int TX, int BN, int WM, int WN, bool UP2, int PF = 1, int NWB = 2>
int launch_conv3q(const MgldIGemm* p, hipStream_t s, int splits, int hchunk) {
  constexpr int lds = conv3q_lds<TY, TX, BN, WM, WN, UP2, NWB>();
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv3q_kernel<TY, TX, BN, WM, WN, UP2, PF, NWB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  const int frames = p->M / (p->Hout * p->Wout);
  const int tiles_x = cdiv(p->Wout, TX), tiles_y = cdiv(p->Hout, TY);
  dim3 grid(frames * tiles_x * tiles_y, cdiv(p->N, BN), splits > 1 ? splits : 1);
  constexpr int THREADS = 64 * (TY * TX / WM) * (BN / WN);
  // tile order: share the weight tiles per XCD where the weights outweigh the activations (env MGLD_CONV3Q_ORDER = 0 / 1 forces)
  static int forder = -2;
  if (forder == -2) { const char* e = getenv("MGLD_CONV3Q_ORDER"); forder = e ? atoi(e) : -1; }
  const double wbytes = 2.0 * p->N * p->K, abytes = 2.0 * p->M * p->Cin / (UP2 ? 4 : 1);
  const int order = forder >= 0 ? forder : (wbytes > 2.0 * abytes ? 1 : 0);
  hipLaunchKernelGGL((conv3q_kernel<TY, TX, BN, WM, WN, UP2, PF, NWB>), grid, dim3(THREADS), lds, s, *p, splits > 1 ? g_ws : nullptr, hchunk,
                     tiles_x, tiles_y, order);
  if (splits > 1) {
    const int64_t total = (int64_t)p->M * ((p->N + 3) >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, *p, g_ws, splits);
  }
  return mgld_check_launch("igemm(conv3q)");
}

int dispatch_conv3q(const MgldIGemm* p, hipStream_t s, int id, int splits, int hchunk) {
  if (p->up2) {
    if (id == 7) return launch_conv3q<16, 16, 64, 64, 64, true>(p, s, splits, hchunk);
    return id == 1 ? launch_conv3q<16, 16, 64, 64, 32, true>(p, s, splits, hchunk) : launch_conv3q<8, 16, 64, 32, 32, true>(p, s, splits, hchunk);
  }
  const bool w3 = q3_nwb(id, false) == 3;
  switch (id) {
    case 1: return w3 ? launch_conv3q<16, 16, 64, 64, 32, false, 1, 3>(p, s, splits, hchunk) : launch_conv3q<16, 16, 64, 64, 32, false>(p, s, splits, hchunk);
    case 2: return launch_conv3q<8, 16, 128, 64, 32, false>(p, s, splits, hchunk);
    case 3: return w3 ? launch_conv3q<16, 16, 128, 64, 64, false, 1, 3>(p, s, splits, hchunk) : launch_conv3q<16, 16, 128, 64, 64, false>(p, s, splits, hchunk);
    case 4: return w3 ? launch_conv3q<8, 32, 64, 64, 32, false, 1, 3>(p, s, splits, hchunk) : launch_conv3q<8, 32, 64, 64, 32, false>(p, s, splits, hchunk);
    case 5: return launch_conv3q<8, 16, 64, 64, 32, false>(p, s, splits, hchunk);
    case 6: return launch_conv3q<8, 8, 128, 32, 32, false>(p, s, splits, hchunk);
    case 7: return launch_conv3q<8, 32, 64, 64, 64, false>(p, s, splits, hchunk);
    case 8: return launch_conv3q<8, 32, 128, 64, 64, false>(p, s, splits, hchunk);
    case 9: return q3_nwb160() == 3 ? launch_conv3q<8, 32, 160, 64, 160, false, 1, 3>(p, s, splits, hchunk) : launch_conv3q<8, 32, 160, 64, 160, false>(p, s, splits, hchunk);
    default: return launch_conv3q<8, 16, 64, 32, 32, false>(p, s, splits, hchunk);
  }
}

}  // namespace

extern "C" int mgld_set_workspace(void* ptr, int64_t bytes) {
  g_ws = (float*)ptr;
  g_ws_bytes = ptr ? (size_t)bytes : 0;
  return MGLD_OK;
}

// tile configuration the launcher picks for a problem: BM*1000 + BN (+ splits*1000000 when split along K)
extern "C" int mgld_igemm_config(const MgldIGemm* p) {
  if (!p) return 0;
  int cfg, splits, kchunk;
  if (conv3q_plan(p, &cfg, &splits, &kchunk)) return 400000 + cfg + (splits > 1 ? splits * 1000000 : 0);   // 2-D-tile patch conv
  if (conv3p_plan(p, &cfg, &splits, &kchunk)) return 300000 + cfg + (splits > 1 ? splits * 1000000 : 0);   // raster patch conv
  choose(p, &cfg, &splits, &kchunk);
  return cfg + (splits > 1 ? splits * 1000000 : 0);
}

// name of the kernel template instantiation the launcher runs for this problem, spelled as rocprofv3 prints it
extern "C" int mgld_igemm_kernel_name(const MgldIGemm* p, char* buf, int buflen) {
  MGLD_REQUIRE(p && buf && buflen > 0, "igemm_kernel_name: null");
  int cfg, splits, kchunk;
  if (conv3q_plan(p, &cfg, &splits, &kchunk)) {
    static const int g[Q3_NVAR][5] = {{8, 16, 64, 32, 32}, {16, 16, 64, 64, 32}, {8, 16, 128, 64, 32}, {16, 16, 128, 64, 64}, {8, 32, 64, 64, 32}, {8, 16, 64, 64, 32}, {8, 8, 128, 32, 32},
                                      {8, 32, 64, 64, 64}, {8, 32, 128, 64, 64}, {8, 32, 160, 64, 160}};
    if (p->up2 && cfg == 7) snprintf(buf, buflen, "conv3q_kernel<16, 16, 64, 64, 64, true, 1, 2>");
    else
      snprintf(buf, buflen, "conv3q_kernel<%d, %d, %d, %d, %d, %s, %d, %d>", g[cfg][0], g[cfg][1], g[cfg][2], g[cfg][3], g[cfg][4],
               p->up2 ? "true" : "false", 1, cfg == 9 ? q3_nwb160() : q3_nwb(cfg, p->up2 != 0));
    return splits;
  }
  if (conv3p_plan(p, &cfg, &splits, &kchunk)) {
    snprintf(buf, buflen, cfg == 64 ? "conv3p_kernel<128, 64, 32, 32>" : "conv3p_kernel<128, 128, 64, 32>");
    return splits;
  }
  choose(p, &cfg, &splits, &kchunk);
  int bm = cfg / 1000, bn = cfg % 1000, wm, wn;
  switch (cfg) {
    case 128128: wm = 64; wn = (p->act == MGLD_ACT_GEGLU) ? 64 : 32; break;
    case 128320: wm = 32; wn = 160; break;
    case 64128: wm = 32; wn = 64; break;
    case 128032: wm = 32; wn = 32; break;
    case 128064: wm = 64; wn = 32; break;
    default: bm = 64; bn = 64; wm = 32; wn = 32; break;
  }
  snprintf(buf, buflen, "igemm_kernel<%d, %s, %d, %d, %d, %d, 2>", p->mode, fast_ok(p) ? "true" : "false", bm, bn, wm, wn);
  return cfg == 128128 ? splits : 1;
}

extern "C" int mgld_igemm(const MgldIGemm* p, void* stream) {
  MGLD_REQUIRE(p && p->A && p->W && p->C, "igemm: null pointer");
  MGLD_REQUIRE(p->M > 0 && p->N > 0 && p->K > 0, "igemm: empty problem");
  MGLD_REQUIRE((p->K & 7) == 0 && (p->lda & 7) == 0 && (p->ldw & 7) == 0, "igemm: K, lda, ldw must be multiples of 8");
  MGLD_REQUIRE((((uintptr_t)p->A) & 15) == 0 && (((uintptr_t)p->W) & 15) == 0, "igemm: A/W must be 16-byte aligned");
  MGLD_REQUIRE((p->strideA & 7) == 0 && (p->strideW & 7) == 0, "igemm: batch strides must be multiples of 8");
  if (p->mode != MGLD_MODE_LINEAR) {
    MGLD_REQUIRE(p->Cin > 0 && (p->Cin & 7) == 0, "igemm: Cin must be a positive multiple of 8");
    if (p->mode == MGLD_MODE_CONV3X3 && p->kh > 0)
      MGLD_REQUIRE(p->kw > 0 && p->kh <= 15 && p->kw <= 15 && !p->up2, "igemm: conv kernel size (generic taps: 1..15, no upsample fold)");
    const int taps = (p->mode == MGLD_MODE_CONV3X3) ? (p->kh > 0 ? p->kh * p->kw : 9) : 3;
    MGLD_REQUIRE(p->K == taps * p->Cin, "igemm: K != taps*Cin");
    if (p->mode == MGLD_MODE_CONV3X3) {
      MGLD_REQUIRE(p->Hin > 0 && p->Win > 0 && p->Hout > 0 && p->Wout > 0, "igemm: conv geometry");
      MGLD_REQUIRE(p->stride == 1 || p->stride == 2, "igemm: stride must be 1 or 2");
      MGLD_REQUIRE(p->M % (p->Hout * p->Wout) == 0, "igemm: M must be frames*Hout*Wout");
    } else {
      MGLD_REQUIRE(p->T > 0 && p->HW > 0 && p->t_off >= 0, "igemm: tconv geometry");
      if (p->t_off == 0) MGLD_REQUIRE(p->M % (p->T * p->HW) == 0, "igemm: tconv M must be clips*T*HW");
      else MGLD_REQUIRE(p->M % p->HW == 0 && p->M / p->HW + p->t_off <= p->T, "igemm: sharded tconv frames exceed the clip");
    }
  }
  if (p->tap_inner == 2) {
    int c_, s_, h_;
    MGLD_REQUIRE(conv3q_plan(p, &c_, &s_, &h_) || conv3p_plan(p, &c_, &s_, &h_),
                 "igemm: tiled conv weights (tap_inner = 2) need a problem a patch conv takes");
  } else if (p->tap_inner)
    MGLD_REQUIRE(p->mode != MGLD_MODE_LINEAR && (p->Cin % BK) == 0 && !(p->mode == MGLD_MODE_CONV3X3 && p->up2) &&
                     !(p->mode == MGLD_MODE_CONV3X3 && p->kh > 0 && !(p->kh == 3 && p->kw == 3)),
                 "igemm: tap_inner needs a gather mode with Cin % 64 == 0 and no upsample fold");
  if (p->W2) {
    MGLD_REQUIRE((((uintptr_t)p->W2) & 15) == 0 && p->w2_scale > 0.f, "igemm: W2 must be 16-byte aligned with a positive w2_scale");
    MGLD_REQUIRE(!(p->mode == MGLD_MODE_LINEAR && p->tune == 9), "igemm: the register-staged LINEAR variant does not take W2");
  }
  if (p->rowvec) MGLD_REQUIRE(p->rows_per_frame > 0, "igemm: rows_per_frame");
  if (p->act == MGLD_ACT_GEGLU) MGLD_REQUIRE((p->N & 63) == 0, "igemm: GEGLU needs N % 64 == 0");
  hipStream_t s = (hipStream_t)stream;
  int cfg, splits, kchunk;
  if (conv3q_plan(p, &cfg, &splits, &kchunk)) return dispatch_conv3q(p, s, cfg, splits, kchunk);
  if (conv3p_plan(p, &cfg, &splits, &kchunk)) {
    MGLD_REQUIRE(!p->W2, "igemm: the raster patch kernel (MGLD_CONV3Q=0) does not take W2");
    return cfg == 64 ? launch_conv3p<64, 32, 32>(p, s, splits, kchunk) : launch_conv3p<128, 64, 32>(p, s, splits, kchunk);
  }
  choose(p, &cfg, &splits, &kchunk);
  switch (cfg) {
    // 128x128: eight waves of 64x32 (two blocks = 16 waves per CU) measured ~3 % faster end to end than four of 64x64:
    // more waves in flight to cover the barrier / DMA waits outweigh the 1.5x fragment loads per MFMA
    // (GEGLU pairs value / gate columns inside one wave's 64-column tile and keeps the 4-wave form.)
    case 128128:
      if (p->act == MGLD_ACT_GEGLU) return launch_cfg<128, 128, 64, 64>(p, s, splits, kchunk);
      return launch_cfg<128, 128, 64, 32>(p, s, splits, kchunk);
    case 128320:
      launch_fast<MGLD_MODE_LINEAR, true, 128, 320, 32, 160, 2>(p, s, 1, kchunk);
      return mgld_check_launch("igemm");
    case 64128: return launch_cfg<64, 128, 32, 64>(p, s, 1, kchunk);
    case 128032: return launch_cfg<128, 32, 32, 32>(p, s, 1, kchunk);
    case 128064: return launch_cfg<128, 64, 64, 32>(p, s, 1, kchunk);
    default: return launch_cfg<64, 64, 32, 32>(p, s, 1, kchunk);
  }
}
