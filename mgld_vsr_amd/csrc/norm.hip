// norm.hip — GroupNorm(32) / SPADE modulation / LayerNorm on token-major (NHWC) fp16 activations.
// HBM-bound kernels (SURVEY.md §2.2 K2,K3): 16-byte vector loads along channels, fp32 partial sums per channel,
// fp64 combine, statistics kept in fp32 exactly as the reference does (diffusionmodules/util.py:214-216).
#include "common.h"
#include <type_traits>

namespace {

// ---- stage 1: per-(frame, chunk) per-GROUP partial sums ----------------------------------------------------
// grid (chunks, frames, channel windows), 256 threads.  A block reduces a window of Cb channels (whole groups, a multiple
// of 8) over the chunk's rows: per-channel fp32 sums, combined per group in fp64 and written as
// gsums[frame][chunk][group][2] (sum, sumsq; doubles).  The channel split keeps the GPU busy on the low-resolution, wide
// tensors (8x8 / 16x16 latents with 1280-2560 channels).  The consumers (gn_apply / spade_apply) finish the reduction over
// chunks in their prologue, so there is no separate "finalize" launch.
__global__ __launch_bounds__(256) void gn_partial_kernel(const f16* __restrict__ x, int rows, int Cfull, int ld,
                                                         int rows_per_chunk, int groups, int Cb,
                                                         double* __restrict__ gsums) {
  extern __shared__ float sred[];  // [max(rpi,1)][C][2]   (C = channels of this block's window)
  const int c_off = blockIdx.z * Cb;
  const int C = min(Cb, Cfull - c_off);
  const int cg = Cfull / groups;
  const int NV = C >> 3;
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, frame = blockIdx.y, chunks = gridDim.x;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = min(rows, r0 + rows_per_chunk);
  const f16* xf = x + (int64_t)frame * rows * ld + c_off;
  double* gout = gsums + ((int64_t)(frame * chunks + chunk) * groups + c_off / cg) * 2;

  if (NV >= 256) {
    // wide rows: each thread owns vector columns tid, tid+256, ... ; one row at a time
    for (int v = tid; v < NV; v += 256) {
      float s[8], q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
      for (int r = r0; r < r1; ++r) {
        const f16x8 d = *(const f16x8*)(xf + (int64_t)r * ld + v * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = (float)d[j]; s[j] += f; q[j] += f * f; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { sred[(v * 8 + j) * 2] = s[j]; sred[(v * 8 + j) * 2 + 1] = q[j]; }
    }
  } else {
    const int rpi = 256 / NV;  // rows handled per iteration
    const int rr = tid / NV, v = tid - rr * NV;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    if (rr < rpi) {
      constexpr int U = 4;  // rows in flight per thread
      const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int r = r0 + rr; r < r1; r += rpi * U) {
        f16x8 d[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
          d[u] = (r + u * rpi < r1) ? *(const f16x8*)(xf + (int64_t)(r + u * rpi) * ld + v * 8) : z8;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float f = (float)d[u][j]; s[j] += f; q[j] += f * f; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sred[((rr * C) + v * 8 + j) * 2] = s[j];
        sred[((rr * C) + v * 8 + j) * 2 + 1] = q[j];
      }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {   // fold the rpi row slots into slot 0 (column c is touched by this thread only)
      float ss = 0.f, qq = 0.f;
      for (int r = 0; r < rpi; ++r) { ss += sred[(r * C + c) * 2]; qq += sred[(r * C + c) * 2 + 1]; }
      sred[c * 2] = ss; sred[c * 2 + 1] = qq;
    }
  }
  __syncthreads();
  for (int g = tid; g < C / cg; g += 256) {
    double s = 0.0, q = 0.0;
    for (int i = 0; i < cg; ++i) { s += (double)sred[(g * cg + i) * 2]; q += (double)sred[(g * cg + i) * 2 + 1]; }
    gout[g * 2] = s; gout[g * 2 + 1] = q;
  }
}

// ---- (mean, rstd) of every group of one frame from the chunk sums, fp64 combine; result in LDS [groups][2].
// Called by all 256 threads of a block; ends with a barrier.
constexpr int GN_MAX_GROUPS = 256;
__device__ __forceinline__ void gn_group_stats(const double* __restrict__ gs, int chunks, int groups, int rows, int cg,
                                               float eps, float (*st)[2]) {
  const int tid = threadIdx.x;
  const int sub = tid & 7;
  for (int g = tid >> 3; g < groups; g += 32) {
    double s = 0.0, q = 0.0;
    for (int ch = sub; ch < chunks; ch += 8) {
      s += gs[((int64_t)ch * groups + g) * 2];
      q += gs[((int64_t)ch * groups + g) * 2 + 1];
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (sub == 0) {
      const double n = (double)rows * cg;
      const double mean = s / n;
      double var = q / n - mean * mean;
      if (var < 0.0) var = 0.0;
      st[g][0] = (float)mean;
      st[g][1] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
}

// ---- the same from per-CHANNEL fp32 tile sums (kind 1: written by the ping-pong GEMM family's epilogue, MgldIGemm.gn_part):
// part[frame * chunks + chunk][2][C] floats (sum row, sumsq row).  Only the groups of the caller's channel window [c_off, c_off + Cw) are
// computed: threads run along channels (coalesced rows of part[], the chunk loads independent), the per-channel totals meet in LDS
// (chs[2][Cw] DOUBLES, caller-provided) and one thread per group adds its cg channels in fp64.  The cross-chunk totals are fp64 too
// (round 5): at the VAE's 256^2 / 512^2 planes a channel has 128-1024 tile sums, and for a channel with |mean| >> std the
// variance q/n - mean^2 cancels — 1e-6 of relative error in the fp32 running totals was a percent of the variance.
__device__ __forceinline__ void gn_group_stats_pc(const float* __restrict__ part, int chunks, int C, int c_off, int Cw, int rows, int cg,
                                                  float eps, float (*st)[2], double* __restrict__ chs) {
  const int tid = threadIdx.x;
  for (int c = tid; c < 2 * Cw; c += 256) {
    const int which = c >= Cw, cc = c - which * Cw;
    const float* src = part + (int64_t)which * C + c_off + cc;
    double a0 = 0.0, a1 = 0.0;
    int ch = 0;
    for (; ch + 1 < chunks; ch += 2) { a0 += (double)src[(int64_t)ch * 2 * C]; a1 += (double)src[(int64_t)(ch + 1) * 2 * C]; }
    if (ch < chunks) a0 += (double)src[(int64_t)ch * 2 * C];
    chs[c] = a0 + a1;
  }
  __syncthreads();
  const int g0 = c_off / cg;
  for (int g = tid; g < Cw / cg; g += 256) {
    double s = 0.0, q = 0.0;
    for (int i = 0; i < cg; ++i) { s += chs[g * cg + i]; q += chs[Cw + g * cg + i]; }
    const double nn = (double)rows * cg;
    const double mean = s / nn;
    double var = q / nn - mean * mean;
    if (var < 0.0) var = 0.0;
    st[g0 + g][0] = (float)mean;
    st[g0 + g][1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
}

// ---- apply: y = [silu]((x-mean)*rstd*gamma+beta) ----------------------------------------------------------------
// STATS: the block also reduces ITS OUTPUT rows (the values it stores, before their rounding to fp16) to per-group (sum, sumsq) and writes them in
// gn_partial_kernel's format with chunks = gridDim.x: the GroupNorm that reads y next (SPADE output -> the transformer's norm)
// needs no statistics launch of its own.
// LO: the residual stream's low plane (common.h: value = hi + 2^-11 lo).  Plain apply: `lo_in` is the low plane of x (same ldx) and the
// normalisation sees hi + 2^-11 lo.  SPADE: `lo_in` is the low plane of the skip tensor (same ldskip), `lo_out` receives the low plane of y
// (same ldy): the block's output `skip + spade(h)` is the next value of the stream.
template <bool SPADE, bool STATS, bool LO = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void gn_apply_kernel(const f16* __restrict__ x, int ldx, const void* __restrict__ sums, int kind,
                                                       int chunks, float eps, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const f16* __restrict__ gb, int ldgb,
                                                       const f16* __restrict__ skip, int ldskip, f16* __restrict__ y, int ldy,
                                                       int rows_per_frame, int C, int groups, int silu, int Cb,
                                                       const int* __restrict__ step_idx, int64_t gb_step_stride,
                                                       double* __restrict__ gout, const f16* __restrict__ lo_in = nullptr,
                                                       f16* __restrict__ lo_out = nullptr) {
  // grid (row chunks, frames).  A thread owns ONE 8-channel vector column for all its rows, so the per-channel scale /
  // shift (rstd*gamma, beta - mean*rstd*gamma) are computed once into registers and the row loop is load-fma-store.
  __shared__ float st[GN_MAX_GROUPS][2];
  extern __shared__ __attribute__((aligned(16))) float sred[];   // STATS: [rpi][Cw][2] floats; statistics of kind 1: [2][Cw] doubles in the prologue
  if (SPADE && step_idx) gb += (int64_t)step_idx[0] * gb_step_stride;   // gamma/beta table hoisted out of the step (see ddpm.py)
  const int c_off = blockIdx.z * Cb;            // this block's channel window [c_off, c_off + Cb)
  const int Cw = min(Cb, C - c_off);
  const int NV = Cw >> 3;
  const int cg = C / groups;
  const int frame = blockIdx.y;
  if (kind == 1)
    gn_group_stats_pc((const float*)sums + (int64_t)frame * chunks * 2 * C, chunks, C, c_off, Cw, rows_per_frame, cg, eps, st, (double*)sred);
  else
    gn_group_stats((const double*)sums + (int64_t)frame * chunks * groups * 2, chunks, groups, rows_per_frame, cg, eps, st);
  const int rows_per_chunk = (rows_per_frame + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per_chunk;
  const int r1 = min(rows_per_frame, r0 + rows_per_chunk);
  const int64_t fbase = (int64_t)frame * rows_per_frame;
  const int rpi = NV >= 256 ? 1 : 256 / NV;
  const int rr = NV >= 256 ? 0 : threadIdx.x / NV;
  if (!STATS && rr >= rpi) return;
  if (rr < rpi)
  for (int v = NV >= 256 ? threadIdx.x : threadIdx.x - rr * NV; v < NV; v += 256) {
    const int c0 = c_off + v * 8;
    float sa[8], sb[8];
    float os[8], oq[8];                          // STATS: sums of this thread's stored values
#pragma unroll
    for (int j = 0; j < 8; ++j) { os[j] = 0.f; oq[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      const int g = c / cg;
      sa[j] = st[g][1] * gamma[c];
      sb[j] = beta[c] - st[g][0] * sa[j];
    }
    constexpr int U = (SPADE && LO) ? 2 : 4;  // rows in flight per thread (five 16-byte loads per row with the low planes: two rows fill the register budget)
    for (int r = r0 + rr; r < r1; r += rpi * U) {
      f16x8 d[U], gm[U], bt[U], sk[U], dl[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t row = fbase + r + u * rpi;
        if (r + u * rpi < r1) {
          d[u] = *(const f16x8*)(x + row * ldx + c0);
          if (SPADE) {
            gm[u] = *(const f16x8*)(gb + row * ldgb + c0);
            bt[u] = *(const f16x8*)(gb + row * ldgb + C + c0);
            sk[u] = *(const f16x8*)(skip + row * ldskip + c0);
            if (LO) dl[u] = *(const f16x8*)(lo_in + row * ldskip + c0);
          } else if (LO) {
            dl[u] = *(const f16x8*)(lo_in + row * ldx + c0);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (r + u * rpi < r1) {
          const int64_t row = fbase + r + u * rpi;
          f16x8 o, ol;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float xin = (float)d[u][j];
            if (LO && !SPADE) xin += MGLD_LO_SCALE * (float)dl[u][j];
            float f = xin * sa[j] + sb[j];
            if (SPADE) {
              float skv = (float)sk[u][j];
              if (LO) skv += MGLD_LO_SCALE * (float)dl[u][j];
              f = f * (1.f + (float)gm[u][j]) + (float)bt[u][j] + skv;
            } else if (silu == 1) {
              f = silu_f(f);
            } else if (silu == 2) {
              f = fmaxf(f, 0.f);
            }
            o[j] = (f16)f;
            if (LO && SPADE) ol[j] = lo_plane(f, o[j]);
            if (STATS) { os[j] += f; oq[j] += f * f; }       // (before the fp16 rounding: see pp_epilogue_stats)
          }
          *(f16x8*)(y + row * ldy + c0) = o;
          if (LO && SPADE) *(f16x8*)(lo_out + row * ldy + c0) = ol;
        }
      }
    }
    if (STATS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sred[((rr * Cw) + v * 8 + j) * 2] = os[j];
        sred[((rr * Cw) + v * 8 + j) * 2 + 1] = oq[j];
      }
    }
    if (NV < 256) break;
  }
  if (STATS) {       // per channel over the row slots (fp32), per group in fp64: exactly gn_partial_kernel's reduction of the same rows
    __syncthreads();
    if (rpi > 1) {
      for (int c = threadIdx.x; c < Cw; c += 256) {
        float ss = 0.f, qq = 0.f;
        for (int r = 0; r < rpi; ++r) { ss += sred[(r * Cw + c) * 2]; qq += sred[(r * Cw + c) * 2 + 1]; }
        sred[c * 2] = ss; sred[c * 2 + 1] = qq;
      }
      __syncthreads();
    }
    double* go = gout + ((int64_t)(frame * gridDim.x + blockIdx.x) * groups + c_off / cg) * 2;
    for (int g = threadIdx.x; g < Cw / cg; g += 256) {
      double s = 0.0, q = 0.0;
      for (int i = 0; i < cg; ++i) { s += (double)sred[(g * cg + i) * 2]; q += (double)sred[(g * cg + i) * 2 + 1]; }
      go[g * 2] = s; go[g * 2 + 1] = q;
    }
  }
}

// ---- small frames (<= 256 rows): statistics + apply in ONE launch --------------------------------------------------------
// grid (channel windows, frames), 256 threads.  A block owns a window of whole groups (Cb channels, a multiple of 8) of one frame
// for ALL of its rows: the rows are loaded once into registers (<= GNF_R vector rows per thread), reduced per channel in fp32 and
// per group in fp64 exactly like gn_partial_kernel / gn_group_stats, and normalised from the registers.  The 16x16 / 8x8 levels of
// the UNet (and the struct-cond encoder's) are launch-bound: this halves their GroupNorm launches and reads the tensor once.
constexpr int GNF_R = 16;
template <bool SPADE, bool LO = false>
__global__ __launch_bounds__(256) void gn_fused_kernel(const f16* __restrict__ x, int ldx, float eps, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const f16* __restrict__ gb, int ldgb,
                                                       const f16* __restrict__ skip, int ldskip, f16* __restrict__ y, int ldy, int rows,
                                                       int C, int groups, int silu, int Cb, const int* __restrict__ step_idx,
                                                       int64_t gb_step_stride, const f16* __restrict__ lo_in = nullptr,
                                                       f16* __restrict__ lo_out = nullptr) {
  __shared__ float sred[256 * 8 * 2];          // [rpi][Cb][2] : rpi * Cb = 256/NV * NV*8 <= 2048 channels-slots
  __shared__ float st[GN_MAX_GROUPS][2];       // (mean, rstd) of this window's groups
  if (SPADE && step_idx) gb += (int64_t)step_idx[0] * gb_step_stride;
  const int c_off = blockIdx.x * Cb;
  const int Cw = min(Cb, C - c_off);
  const int NV = Cw >> 3;
  const int cg = C / groups;
  const int frame = blockIdx.y;
  const int tid = threadIdx.x;
  const int rpi = 256 / NV;
  const int rr = tid / NV, v = tid - rr * NV;
  const bool active = rr < rpi;
  const int64_t fbase = (int64_t)frame * rows;
  const int c0 = c_off + v * 8;
  f16x8 d[GNF_R];
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
  if (active) {
    const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < GNF_R; ++k) {
      const int r = rr + k * rpi;
      d[k] = (r < rows) ? *(const f16x8*)(x + (fbase + r) * ldx + c0) : z8;
    }
#pragma unroll
    for (int k = 0; k < GNF_R; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = (float)d[k][j]; s[j] += f; q[j] += f * f; }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sred[((rr * Cw) + v * 8 + j) * 2] = s[j];
      sred[((rr * Cw) + v * 8 + j) * 2 + 1] = q[j];
    }
  }
  __syncthreads();
  // per group: all 256 threads sum the rpi x cg per-channel partials (fp64), wave shuffle + one LDS hop across the four waves
  __shared__ double wred[4][2];
  for (int g = 0; g < Cw / cg; ++g) {
    double sd = 0.0, qd = 0.0;
    for (int i = tid; i < rpi * cg; i += 256) {
      const int r = i / cg, c = g * cg + (i - r * cg);
      sd += (double)sred[(r * Cw + c) * 2];
      qd += (double)sred[(r * Cw + c) * 2 + 1];
    }
    sd = wave_sum_d(sd);
    qd = wave_sum_d(qd);
    if ((tid & 63) == 0) { wred[tid >> 6][0] = sd; wred[tid >> 6][1] = qd; }
    __syncthreads();
    if (tid == 0) {
      const double ss = wred[0][0] + wred[1][0] + wred[2][0] + wred[3][0], qq = wred[0][1] + wred[1][1] + wred[2][1] + wred[3][1];
      const double n = (double)rows * cg;
      const double mean = ss / n;
      double var = qq / n - mean * mean;
      if (var < 0.0) var = 0.0;
      st[g][0] = (float)mean;
      st[g][1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
  }
  if (!active) return;
  float sa[8], sb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j;
    const int g = (c - c_off) / cg;
    sa[j] = st[g][1] * gamma[c];
    sb[j] = beta[c] - st[g][0] * sa[j];
  }
#pragma unroll
  for (int k = 0; k < GNF_R; ++k) {
    const int r = rr + k * rpi;
    if (r < rows) {
      const int64_t row = fbase + r;
      f16x8 gm, bt, sk, dl;
      if (SPADE) {
        gm = *(const f16x8*)(gb + row * ldgb + c0);
        bt = *(const f16x8*)(gb + row * ldgb + C + c0);
        sk = *(const f16x8*)(skip + row * ldskip + c0);
        if (LO) dl = *(const f16x8*)(lo_in + row * ldskip + c0);
      } else if (LO) {
        dl = *(const f16x8*)(lo_in + row * ldx + c0);     // (the statistics above are those of the hi plane: the zero-mean 2^-12 residual moves them by ~1e-7)
      }
      f16x8 o, ol;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float xin = (float)d[k][j];
        if (LO && !SPADE) xin += MGLD_LO_SCALE * (float)dl[j];
        float f = xin * sa[j] + sb[j];
        if (SPADE) {
          float skv = (float)sk[j];
          if (LO) skv += MGLD_LO_SCALE * (float)dl[j];
          f = f * (1.f + (float)gm[j]) + (float)bt[j] + skv;
        } else if (silu == 1) f = silu_f(f);
        else if (silu == 2) f = fmaxf(f, 0.f);
        o[j] = (f16)f;
        if (LO && SPADE) ol[j] = lo_plane(f, o[j]);
      }
      *(f16x8*)(y + row * ldy + c0) = o;
      if (LO && SPADE) *(f16x8*)(lo_out + row * ldy + c0) = ol;
    }
  }
}

// ---- LayerNorm: one wave per token row ---------------------------------------------------------------------
// LO: x is the residual stream, `xlo` its low plane (same ldx): the row is hi + 2^-11 lo, kept in fp32 registers
template <int MAXV, bool LO = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const f16* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, f16* __restrict__ y, int ldy,
                                                        int rows, int C, float eps, const f16* __restrict__ xlo = nullptr) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int NV = C >> 3;
  typedef typename std::conditional<LO, float, f16>::type elem_t;
  elem_t d[MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < NV) {
      const f16x8 h = *(const f16x8*)(x + (int64_t)row * ldx + v * 8);
      if constexpr (LO) {
        const f16x8 l = *(const f16x8*)(xlo + (int64_t)row * ldx + v * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) d[i][j] = (float)h[j] + MGLD_LO_SCALE * (float)l[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) d[i][j] = h[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)d[i][j];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < NV) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float t = (float)d[i][j] - mean; q += t * t; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < NV) {
      f16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = v * 8 + j;
        o[j] = (f16)(((float)d[i][j] - mean) * rstd * gamma[c] + beta[c]);
      }
      *(f16x8*)(y + (int64_t)row * ldy + v * 8) = o;
    }
  }
}

// rows per stats chunk.  Every apply block re-reads chunks x groups x 16 B of sums, so few chunks for the (many) latent-size
// tensors: <= 16 chunks up to 4096 rows; the big VAE planes keep 256 chunks (their stats kernel needs the row parallelism).
inline int rows_per_chunk_for(int rows) {
  if (rows <= 64) return rows;
  if (rows <= 4096) return cdiv(rows, 16) > 64 ? cdiv(rows, 16) : 64;
  return cdiv(rows, 256) > 64 ? cdiv(rows, 256) : 64;
}

}  // namespace

extern "C" int mgld_gn_chunks(int rows_per_frame) {
  if (rows_per_frame <= 0) return 0;
  return cdiv(rows_per_frame, rows_per_chunk_for(rows_per_frame));
}

// LDS of the kernels that keep [rows per pass][channel window][2] floats: a block of window Cw runs rpi(Cw) = 256 / (Cw / 8) row slots,
// and rpi(Cw) * Cw is NOT monotonic in Cw — the last, narrower window of a ragged split can need more than the full ones
// (C = 1280 in windows of 440: the 400-channel tail runs 5 slots x 400 = 2000 channel slots, the full windows 4 x 440 = 1760;
// ADVICE round 4) — so the launch is sized for the larger of the two window widths it contains.
static size_t row_slots_lds_one(int Cw) {
  const int NV = Cw >> 3;
  const int rpi = NV >= 256 ? 1 : 256 / NV;
  return (size_t)rpi * Cw * 2 * sizeof(float);
}
static size_t row_slots_lds(int C, int Cb) {
  const int last = C - (cdiv(C, Cb) - 1) * Cb;
  const size_t a = row_slots_lds_one(Cb), b = row_slots_lds_one(last);
  return a > b ? a : b;
}

// channel-window size: whole groups, a multiple of 8 channels, chosen so that about `want_blocks` blocks exist
static int channel_window(int C, int groups, int64_t blocks_without_split, int want_blocks) {
  const int cg = C / groups;
  int unit = cg;
  while (unit & 7) unit += cg;                     // lcm(cg, 8)
  if (unit > C || C % unit) return C;
  const int nmax = C / unit;
  int64_t want = (want_blocks + blocks_without_split - 1) / blocks_without_split;
  if (want < 1) want = 1;
  if (want > nmax) want = nmax;
  const int per = (int)((nmax + want - 1) / want);
  return per * unit;
}

extern "C" int mgld_gn_stats(const void* x, int frames, int rows, int C, int ld, int groups, double* gsums,
                             void* stream) {
  MGLD_REQUIRE(x && gsums, "gn_stats: null pointer");
  MGLD_REQUIRE(frames > 0 && rows > 0 && C > 0 && groups > 0, "gn_stats: empty");
  MGLD_REQUIRE((C & 7) == 0 && (ld & 7) == 0 && C % groups == 0 && groups <= GN_MAX_GROUPS, "gn_stats: C%8, ld%8, C%groups");
  MGLD_REQUIRE(((uintptr_t)x & 15) == 0, "gn_stats: alignment");
  const int rpc = rows_per_chunk_for(rows);
  const int chunks = cdiv(rows, rpc);
  const int Cb = channel_window(C, groups, (int64_t)chunks * frames, 512);
  const size_t shm = row_slots_lds(C, Cb);
  MGLD_REQUIRE(shm <= 64 * 1024, "gn_stats: LDS budget");
  hipLaunchKernelGGL(gn_partial_kernel, dim3(chunks, frames, cdiv(C, Cb)), dim3(256), shm, (hipStream_t)stream,
                     (const f16*)x, rows, C, ld, rpc, groups, Cb, gsums);
  return mgld_check_launch("gn_stats");
}

static dim3 apply_grid(int frames, int rows, int C, int groups, int* Cb) {
  int chunks = cdiv(rows, 32);
  const int cap = 4096 / (frames > 0 ? frames : 1);
  if (chunks > cap) chunks = cap > 0 ? cap : 1;
  *Cb = channel_window(C, groups, (int64_t)chunks * frames, 512);
  // a block passes over its rows rpi x 4 at a time (4 rows in flight per thread): make the chunk a whole number of passes —
  // 32-row chunks with 24 rows per pass (C = 320) spent a second, three-quarters-idle pass and its memory round trip
  const int NV = *Cb >> 3;
  const int pass = (NV >= 256 ? 1 : 256 / NV) * 4;
  if (pass <= 64 && chunks < cap) {
    int k = (32 + pass / 2) / pass;
    if (k < 1) k = 1;
    int c2 = cdiv(rows, k * pass);
    if (c2 > cap) c2 = cap;
    if (c2 >= 1) chunks = c2;
  }
  return dim3(chunks, frames, cdiv(C, *Cb));
}

// dynamic LDS of the STATS variants: [rows per pass][channel window][2] floats (largest over the windows of the launch)
// + the prologue's per-channel table when the input statistics are per-channel tile sums
static size_t apply_lds(const MgldGnStats* st, int C, int Cb, bool stats_out) {
  const size_t a = stats_out ? row_slots_lds(C, Cb) : 0, b = st->kind == MGLD_GN_CHANNEL_SUMS ? (size_t)2 * Cb * sizeof(double) : 0;
  return a > b ? a : b;
}

static int check_stats_in(const MgldGnStats* st, int rows, int C) {
  MGLD_REQUIRE(st && st->sums, "gn_apply: null statistics");
  MGLD_REQUIRE(st->kind == MGLD_GN_GROUP_SUMS || st->kind == MGLD_GN_CHANNEL_SUMS, "gn_apply: statistics kind");
  MGLD_REQUIRE(st->chunks > 0 && st->chunks <= rows, "gn_apply: statistics chunks");
  (void)C;
  return 0;
}

extern "C" int mgld_gn_apply_chunks(int frames, int rows_per_frame, int C, int groups) {
  if (frames <= 0 || rows_per_frame <= 0 || C <= 0 || groups <= 0 || C % groups) return 0;
  int Cb;
  return (int)apply_grid(frames, rows_per_frame, C, groups, &Cb).x;
}

static int gn_apply_impl(const void* x, const void* xlo, int ldx, const MgldGnStats* st, float eps, const float* gamma, const float* beta,
                         void* y, int ldy, int frames, int rows, int C, int groups, int silu, double* stats_out, void* stream) {
  MGLD_REQUIRE(x && gamma && beta && y, "gn_apply: null pointer");
  MGLD_REQUIRE(!xlo || (!stats_out && (((uintptr_t)xlo) & 15) == 0), "gn_apply: the low-plane form writes no statistics; 16-byte aligned");
  MGLD_REQUIRE((C & 7) == 0 && (ldx & 7) == 0 && (ldy & 7) == 0 && C % groups == 0 && groups <= GN_MAX_GROUPS,
               "gn_apply: alignment");
  if (int rc = check_stats_in(st, rows, C)) return rc;
  int Cb;
  const dim3 grid = apply_grid(frames, rows, C, groups, &Cb);
  const size_t shm = apply_lds(st, C, Cb, stats_out != nullptr);
  MGLD_REQUIRE(shm <= 48 * 1024, "gn_apply: LDS budget of the statistics tables");
  if (xlo) {
    hipLaunchKernelGGL((gn_apply_kernel<false, false, true>), grid, dim3(256), shm, (hipStream_t)stream, (const f16*)x, ldx, st->sums, st->kind,
                       st->chunks, eps, gamma, beta, nullptr, 0, nullptr, 0, (f16*)y, ldy, rows, C, groups, silu, Cb, nullptr, (int64_t)0,
                       nullptr, (const f16*)xlo, nullptr);
  } else if (stats_out) {
    hipLaunchKernelGGL((gn_apply_kernel<false, true>), grid, dim3(256), shm, (hipStream_t)stream, (const f16*)x, ldx, st->sums, st->kind,
                       st->chunks, eps, gamma, beta, nullptr, 0, nullptr, 0, (f16*)y, ldy, rows, C, groups, silu, Cb, nullptr, (int64_t)0,
                       stats_out);
  } else {
    hipLaunchKernelGGL((gn_apply_kernel<false, false>), grid, dim3(256), shm, (hipStream_t)stream, (const f16*)x, ldx, st->sums, st->kind,
                       st->chunks, eps, gamma, beta, nullptr, 0, nullptr, 0, (f16*)y, ldy, rows, C, groups, silu, Cb, nullptr, (int64_t)0,
                       nullptr);
  }
  return mgld_check_launch("gn_apply");
}

extern "C" int mgld_gn_apply2(const void* x, int ldx, const MgldGnStats* st, float eps, const float* gamma, const float* beta,
                              void* y, int ldy, int frames, int rows, int C, int groups, int silu, double* stats_out, void* stream) {
  return gn_apply_impl(x, nullptr, ldx, st, eps, gamma, beta, y, ldy, frames, rows, C, groups, silu, stats_out, stream);
}

extern "C" int mgld_gn_apply_lo(const void* x, const void* xlo, int ldx, const MgldGnStats* st, float eps, const float* gamma, const float* beta,
                                void* y, int ldy, int frames, int rows, int C, int groups, int silu, void* stream) {
  MGLD_REQUIRE(xlo, "gn_apply_lo: null low plane");
  return gn_apply_impl(x, xlo, ldx, st, eps, gamma, beta, y, ldy, frames, rows, C, groups, silu, nullptr, stream);
}

extern "C" int mgld_gn_apply(const void* x, int ldx, const double* gsums, float eps, const float* gamma, const float* beta,
                             void* y, int ldy, int frames, int rows, int C, int groups, int silu, void* stream) {
  const MgldGnStats st = {gsums, MGLD_GN_GROUP_SUMS, mgld_gn_chunks(rows)};
  return mgld_gn_apply2(x, ldx, &st, eps, gamma, beta, y, ldy, frames, rows, C, groups, silu, nullptr, stream);
}

static int spade_apply_impl(const void* h, int ldh, const MgldGnStats* st, float eps, const float* gamma, const float* beta,
                            const void* gb, int ldgb, const void* skip, const void* skiplo, int ldskip, void* y, void* ylo, int ldy, int frames,
                            int rows, int C, int groups, const int32_t* gb_step_idx, int64_t gb_step_stride, double* stats_out, void* stream) {
  MGLD_REQUIRE(h && gamma && beta && gb && skip && y, "spade_apply: null pointer");
  MGLD_REQUIRE((!skiplo) == (!ylo) && ((((uintptr_t)skiplo) | ((uintptr_t)ylo)) & 15) == 0, "spade_apply: low planes of skip and y come together, 16-byte aligned");
  MGLD_REQUIRE((C & 7) == 0 && (ldh & 7) == 0 && (ldy & 7) == 0 && (ldgb & 7) == 0 && (ldskip & 7) == 0 && C % groups == 0 &&
                   groups <= GN_MAX_GROUPS,
               "spade_apply: alignment");
  if (int rc = check_stats_in(st, rows, C)) return rc;
  int Cb;
  const dim3 grid = apply_grid(frames, rows, C, groups, &Cb);
  const size_t shm = apply_lds(st, C, Cb, stats_out != nullptr);
  MGLD_REQUIRE(shm <= 48 * 1024, "spade_apply: LDS budget of the statistics tables");
  if (ylo && stats_out) {
    hipLaunchKernelGGL((gn_apply_kernel<true, true, true>), grid, dim3(256), shm, (hipStream_t)stream, (const f16*)h, ldh, st->sums, st->kind,
                       st->chunks, eps, gamma, beta, (const f16*)gb, ldgb, (const f16*)skip, ldskip, (f16*)y, ldy, rows, C, groups, 0, Cb,
                       gb_step_idx, gb_step_stride, stats_out, (const f16*)skiplo, (f16*)ylo);
  } else if (ylo) {
    hipLaunchKernelGGL((gn_apply_kernel<true, false, true>), grid, dim3(256), shm, (hipStream_t)stream, (const f16*)h, ldh, st->sums, st->kind,
                       st->chunks, eps, gamma, beta, (const f16*)gb, ldgb, (const f16*)skip, ldskip, (f16*)y, ldy, rows, C, groups, 0, Cb,
                       gb_step_idx, gb_step_stride, nullptr, (const f16*)skiplo, (f16*)ylo);
  } else if (stats_out) {
    hipLaunchKernelGGL((gn_apply_kernel<true, true>), grid, dim3(256), shm, (hipStream_t)stream, (const f16*)h, ldh, st->sums, st->kind,
                       st->chunks, eps, gamma, beta, (const f16*)gb, ldgb, (const f16*)skip, ldskip, (f16*)y, ldy, rows, C, groups, 0, Cb,
                       gb_step_idx, gb_step_stride, stats_out);
  } else {
    hipLaunchKernelGGL((gn_apply_kernel<true, false>), grid, dim3(256), shm, (hipStream_t)stream, (const f16*)h, ldh, st->sums, st->kind,
                       st->chunks, eps, gamma, beta, (const f16*)gb, ldgb, (const f16*)skip, ldskip, (f16*)y, ldy, rows, C, groups, 0, Cb,
                       gb_step_idx, gb_step_stride, nullptr);
  }
  return mgld_check_launch("spade_apply");
}

extern "C" int mgld_spade_apply2(const void* h, int ldh, const MgldGnStats* st, float eps, const float* gamma, const float* beta,
                                 const void* gb, int ldgb, const void* skip, int ldskip, void* y, int ldy, int frames, int rows, int C,
                                 int groups, const int32_t* gb_step_idx, int64_t gb_step_stride, double* stats_out, void* stream) {
  return spade_apply_impl(h, ldh, st, eps, gamma, beta, gb, ldgb, skip, nullptr, ldskip, y, nullptr, ldy, frames, rows, C, groups, gb_step_idx,
                          gb_step_stride, stats_out, stream);
}

extern "C" int mgld_spade_apply_lo(const void* h, int ldh, const MgldGnStats* st, float eps, const float* gamma, const float* beta,
                                   const void* gb, int ldgb, const void* skip, const void* skiplo, int ldskip, void* y, void* ylo, int ldy,
                                   int frames, int rows, int C, int groups, const int32_t* gb_step_idx, int64_t gb_step_stride,
                                   double* stats_out, void* stream) {
  MGLD_REQUIRE(skiplo && ylo, "spade_apply_lo: null low plane");
  return spade_apply_impl(h, ldh, st, eps, gamma, beta, gb, ldgb, skip, skiplo, ldskip, y, ylo, ldy, frames, rows, C, groups, gb_step_idx,
                          gb_step_stride, stats_out, stream);
}

extern "C" int mgld_spade_apply(const void* h, int ldh, const double* gsums, float eps, const float* gamma,
                                const float* beta, const void* gb, int ldgb, const void* skip, int ldskip, void* y, int ldy,
                                int frames, int rows, int C, int groups, const int32_t* gb_step_idx, int64_t gb_step_stride,
                                void* stream) {
  const MgldGnStats st = {gsums, MGLD_GN_GROUP_SUMS, mgld_gn_chunks(rows)};
  return mgld_spade_apply2(h, ldh, &st, eps, gamma, beta, gb, ldgb, skip, ldskip, y, ldy, frames, rows, C, groups, gb_step_idx,
                           gb_step_stride, nullptr, stream);
}

// window of the single-launch GroupNorm: the smallest whole-group, multiple-of-8 channel window; 0 = shape not covered
static int fused_window(int rows, int C, int groups) {
  if (rows <= 0 || rows > 256 || groups <= 0 || C % groups || (C & 7)) return 0;
  const int cg = C / groups;
  int unit = cg;
  while (unit & 7) unit += cg;
  if (unit > C || C % unit || unit > 128) return 0;      // NV <= 16 -> >= 16 rows per pass -> <= GNF_R passes for 256 rows
  const int rpi = 256 / (unit >> 3);
  if ((rows + rpi - 1) / rpi > GNF_R) return 0;
  return unit;
}

extern "C" int mgld_gn_fused_applies(int rows_per_frame, int C, int groups) { return fused_window(rows_per_frame, C, groups) > 0; }

static int gn_fused_impl(const void* x, int ldx, float eps, const float* gamma, const float* beta, const void* gb, int ldgb,
                         const void* skip, int ldskip, void* y, int ldy, int frames, int rows, int C, int groups, int silu,
                         const int32_t* gb_step_idx, int64_t gb_step_stride, const void* lo_in, void* lo_out, void* stream) {
  MGLD_REQUIRE(x && gamma && beta && y, "gn_fused: null pointer");
  MGLD_REQUIRE(((((uintptr_t)lo_in) | ((uintptr_t)lo_out)) & 15) == 0 && (gb ? (!lo_in) == (!lo_out) : !lo_out),
               "gn_fused: low planes (SPADE: skip's and y's together; plain: x's only), 16-byte aligned");
  MGLD_REQUIRE((ldx & 7) == 0 && (ldy & 7) == 0 && groups <= GN_MAX_GROUPS && frames > 0, "gn_fused: alignment");
  const int Cb = fused_window(rows, C, groups);
  MGLD_REQUIRE(Cb > 0, "gn_fused: shape not covered (mgld_gn_fused_applies)");
  const dim3 grid(C / Cb, frames);
  if (gb) {
    MGLD_REQUIRE(skip && (ldgb & 7) == 0 && (ldskip & 7) == 0, "gn_fused: SPADE operands");
    if (lo_in)
      hipLaunchKernelGGL((gn_fused_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, ldx, eps, gamma, beta,
                         (const f16*)gb, ldgb, (const f16*)skip, ldskip, (f16*)y, ldy, rows, C, groups, 0, Cb, gb_step_idx, gb_step_stride,
                         (const f16*)lo_in, (f16*)lo_out);
    else
      hipLaunchKernelGGL((gn_fused_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, ldx, eps, gamma, beta,
                         (const f16*)gb, ldgb, (const f16*)skip, ldskip, (f16*)y, ldy, rows, C, groups, 0, Cb, gb_step_idx, gb_step_stride);
  } else if (lo_in) {
    hipLaunchKernelGGL((gn_fused_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, ldx, eps, gamma, beta,
                       nullptr, 0, nullptr, 0, (f16*)y, ldy, rows, C, groups, silu, Cb, nullptr, (int64_t)0, (const f16*)lo_in, nullptr);
  } else {
    hipLaunchKernelGGL((gn_fused_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, ldx, eps, gamma, beta,
                       nullptr, 0, nullptr, 0, (f16*)y, ldy, rows, C, groups, silu, Cb, nullptr, (int64_t)0);
  }
  return mgld_check_launch("gn_fused");
}

extern "C" int mgld_gn_fused(const void* x, int ldx, float eps, const float* gamma, const float* beta, const void* gb, int ldgb,
                             const void* skip, int ldskip, void* y, int ldy, int frames, int rows, int C, int groups, int silu,
                             const int32_t* gb_step_idx, int64_t gb_step_stride, void* stream) {
  return gn_fused_impl(x, ldx, eps, gamma, beta, gb, ldgb, skip, ldskip, y, ldy, frames, rows, C, groups, silu, gb_step_idx, gb_step_stride,
                       nullptr, nullptr, stream);
}

extern "C" int mgld_gn_fused_lo(const void* x, int ldx, float eps, const float* gamma, const float* beta, const void* gb, int ldgb,
                                const void* skip, int ldskip, void* y, int ldy, int frames, int rows, int C, int groups, int silu,
                                const int32_t* gb_step_idx, int64_t gb_step_stride, const void* lo_in, void* lo_out, void* stream) {
  MGLD_REQUIRE(lo_in, "gn_fused_lo: null low plane");
  return gn_fused_impl(x, ldx, eps, gamma, beta, gb, ldgb, skip, ldskip, y, ldy, frames, rows, C, groups, silu, gb_step_idx, gb_step_stride,
                       lo_in, lo_out, stream);
}

template <int MAXV>
static void launch_layernorm_lo(dim3 grid, hipStream_t s, const void* x, const void* xlo, int ldx, const float* gamma, const float* beta, void* y,
                                int ldy, int rows, int C, float eps) {
  hipLaunchKernelGGL((layernorm_kernel<MAXV, true>), grid, dim3(256), 0, s, (const f16*)x, ldx, gamma, beta, (f16*)y, ldy, rows, C, eps,
                     (const f16*)xlo);
}

extern "C" int mgld_layernorm_lo(const void* x, const void* xlo, int ldx, const float* gamma, const float* beta, void* y, int ldy, int rows,
                                 int C, float eps, void* stream) {
  MGLD_REQUIRE(x && xlo && gamma && beta && y, "layernorm_lo: null pointer");
  MGLD_REQUIRE((C & 7) == 0 && (ldx & 7) == 0 && (ldy & 7) == 0 && C <= 2048 && rows > 0 && (((uintptr_t)xlo) & 15) == 0, "layernorm_lo: shape");
  const int NV = C >> 3;
  dim3 grid(cdiv(rows, 4));
  if (NV <= 64) launch_layernorm_lo<1>(grid, (hipStream_t)stream, x, xlo, ldx, gamma, beta, y, ldy, rows, C, eps);
  else if (NV <= 128) launch_layernorm_lo<2>(grid, (hipStream_t)stream, x, xlo, ldx, gamma, beta, y, ldy, rows, C, eps);
  else launch_layernorm_lo<4>(grid, (hipStream_t)stream, x, xlo, ldx, gamma, beta, y, ldy, rows, C, eps);
  return mgld_check_launch("layernorm_lo");
}

extern "C" int mgld_layernorm(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, int rows,
                              int C, float eps, void* stream) {
  MGLD_REQUIRE(x && gamma && beta && y, "layernorm: null pointer");
  MGLD_REQUIRE((C & 7) == 0 && (ldx & 7) == 0 && (ldy & 7) == 0 && C <= 2048 && rows > 0, "layernorm: shape");
  const int NV = C >> 3;
  dim3 grid(cdiv(rows, 4));
  if (NV <= 64)
    hipLaunchKernelGGL((layernorm_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, ldx, gamma, beta,
                       (f16*)y, ldy, rows, C, eps);
  else if (NV <= 128)
    hipLaunchKernelGGL((layernorm_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, ldx, gamma, beta,
                       (f16*)y, ldy, rows, C, eps);
  else
    hipLaunchKernelGGL((layernorm_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, ldx, gamma, beta,
                       (f16*)y, ldy, rows, C, eps);
  return mgld_check_launch("layernorm");
}
