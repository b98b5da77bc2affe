// elementwise.hip — small dense ops, layout conversion, flow utilities, colour fix and tile stitching.
// All are HBM- or latency-bound (SURVEY.md §2.2 K9,K11,K14,K15): coalesced / 16-byte accesses, fp32 math.
#include "common.h"

namespace {

inline int egrid(int64_t total, int per_block = 256) {
  int64_t b = (total + per_block - 1) / per_block;
  if (b > 256 * 32) b = 256 * 32;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- weight-streaming small-M linear: one wave per output feature ---------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ a, int lda, const f16* __restrict__ w,
                                                           int ldw, const float* __restrict__ bias, float* __restrict__ y,
                                                           int ldy, int M, int N, int K, int silu_in, int silu_out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = 0.f;
  const f16* wr = w + (int64_t)n * ldw;
  for (int k = lane * 8; k < K; k += 64 * 8) {
    const f16x8 wv = *(const f16x8*)(wr + k);
    float wf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wf[j] = (float)wv[j];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m < M) {
        const f32x4 a0 = *(const f32x4*)(a + (int64_t)m * lda + k);
        const f32x4 a1 = *(const f32x4*)(a + (int64_t)m * lda + k + 4);
        float av[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x = silu_in ? silu_f(av[j]) : av[j];
          acc[m] += x * wf[j];
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const float s = wave_sum(acc[m]);
    if (lane == 0 && m < M) {
      float v = s + (bias ? bias[n] : 0.f);
      if (silu_out) v = silu_f(v);
      y[(int64_t)m * ldy + n] = v;
    }
  }
}

__global__ void timestep_embedding_kernel(const float* __restrict__ tvals, int t_stride, float* __restrict__ out, int M,
                                          int dim) {
  const int half = dim / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * dim) return;
  const int m = idx / dim, j = idx - m * dim;
  float v = 0.f;
  if (j < 2 * half) {
    const int jj = j < half ? j : j - half;
    // reference: freqs = exp(-log(10000) * arange(half, fp32) / half) ; args = t * freqs   (util.py:163-167)
    const float freq = expf(-9.210340371976184f * (float)jj / (float)half);
    const float arg = tvals[m * t_stride] * freq;
    v = j < half ? cosf(arg) : sinf(arg);
  }
  out[idx] = v;
}

// ---- layout ------------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, f16* __restrict__ y, int n, int c, int hw, int cpad,
                                    int ld) {
  const int64_t total = (int64_t)n * hw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = idx / hw;
    const int64_t pix = idx - f * hw;
    for (int c0 = 0; c0 < cpad; c0 += 8) {
      f16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = c0 + j;
        o[j] = ch < c ? (f16)x[(f * c + ch) * hw + pix] : (f16)0.f;
      }
      *(f16x8*)(y + idx * ld + c0) = o;
    }
  }
}

__global__ void nhwc_to_nchw_kernel(const void* __restrict__ x, int in_f32, int ld, float* __restrict__ y, int n, int c,
                                    int hw) {
  const int64_t total = (int64_t)n * c * hw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = idx % hw;
    const int64_t fc = idx / hw;
    const int ch = (int)(fc % c);
    const int64_t f = fc / c;
    const int64_t src = (f * hw + pix) * ld + ch;
    y[idx] = in_f32 ? ((const float*)x)[src] : (float)((const f16*)x)[src];
  }
}

__global__ void copy2d_kernel(const f16* __restrict__ src, int lds_, f16* __restrict__ dst, int ldd, int64_t rows, int nv) {
  const int64_t total = rows * nv;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / nv;
    const int v = (int)(idx - r * nv);
    *(f16x8*)(dst + r * ldd + v * 8) = *(const f16x8*)(src + r * lds_ + v * 8);
  }
}

__global__ void axpby_kernel(const f16* __restrict__ x, int ldx, f16* __restrict__ y, int ldy, int64_t rows, int nv, float a,
                             float b) {
  const int64_t total = rows * nv;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / nv;
    const int v = (int)(idx - r * nv);
    const f16x8 xv = *(const f16x8*)(x + r * ldx + v * 8);
    f16x8 yv = *(const f16x8*)(y + r * ldy + v * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) yv[j] = (f16)(a * (float)xv[j] + b * (float)yv[j]);
    *(f16x8*)(y + r * ldy + v * 8) = yv;
  }
}

// the same on residual-stream tensors kept as two fp16 planes (common.h: value = hi + 2^-11 lo): y <- a x + b y with x = (x, xlo) — xlo may be
// null — and y = (y, ylo); the fusion layers' `dec_feat + w * enc_feat` (model.py:1367)
__global__ void axpby_lo_kernel(const f16* __restrict__ x, const f16* __restrict__ xlo, int ldx, f16* __restrict__ y, f16* __restrict__ ylo, int ldy,
                                int64_t rows, int nv, float a, float b) {
  const int64_t total = rows * nv;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / nv;
    const int v = (int)(idx - r * nv);
    const f16x8 xv = *(const f16x8*)(x + r * ldx + v * 8);
    f16x8 xl = {0, 0, 0, 0, 0, 0, 0, 0};
    if (xlo) xl = *(const f16x8*)(xlo + r * ldx + v * 8);
    f16x8 yv = *(const f16x8*)(y + r * ldy + v * 8);
    f16x8 yl = *(const f16x8*)(ylo + r * ldy + v * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = a * ((float)xv[j] + MGLD_LO_SCALE * (float)xl[j]) + b * ((float)yv[j] + MGLD_LO_SCALE * (float)yl[j]);
      yv[j] = (f16)t;
      yl[j] = lo_plane(t, yv[j]);
    }
    *(f16x8*)(y + r * ldy + v * 8) = yv;
    *(f16x8*)(ylo + r * ldy + v * 8) = yl;
  }
}

// ---- temporal attention (attention.py:124-143): tokens (pixel) x T frames, tiny ------------------------------
// one wave per (pixel, head); lanes span the head dim.  q,k,v: [T*HW, ld] fp16 (frame-major), out same layout.
template <int DH>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                            const f16* __restrict__ v, int ld, f16* __restrict__ o,
                                                            int ldo, int T, int HW, int heads, float scale) {
  constexpr int E = DH / 64;  // elements per lane
  constexpr int TMAX = 16;
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= HW * heads) return;
  const int pix = item / heads, h = item - pix * heads;
  float kk[TMAX][E], vv[TMAX][E];
#pragma unroll
  for (int j = 0; j < TMAX; ++j)
    if (j < T) {
      const int64_t off = ((int64_t)j * HW + pix) * ld + h * DH;
#pragma unroll
      for (int e = 0; e < E; ++e) { kk[j][e] = (float)k[off + lane + e * 64]; vv[j][e] = (float)v[off + lane + e * 64]; }
    }
  for (int i = 0; i < T; ++i) {
    const int64_t off = ((int64_t)i * HW + pix) * ld + h * DH;
    float qq[E];
#pragma unroll
    for (int e = 0; e < E; ++e) qq[e] = (float)q[off + lane + e * 64];
    float s[TMAX];
    float mx = -1e30f;
#pragma unroll
    for (int j = 0; j < TMAX; ++j)
      if (j < T) {
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) d += qq[e] * kk[j][e];
        s[j] = wave_sum(d) * scale;
        mx = fmaxf(mx, s[j]);
      }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < TMAX; ++j)
      if (j < T) { s[j] = __expf(s[j] - mx); den += s[j]; }
    const float inv = 1.f / den;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < TMAX; ++j)
        if (j < T) acc += s[j] * vv[j][e];
      o[((int64_t)i * HW + pix) * ldo + h * DH + lane + e * 64] = (f16)(acc * inv);
    }
  }
}

// ---- DDPM reverse step (ddpm.py:340-353, 4344-4357) -----------------------------------------------------------
__global__ void ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ eps, int ld_eps,
                                 const float* __restrict__ noise_base, int64_t noise_step_stride,
                                 const float* __restrict__ coef, const int32_t* __restrict__ step_idx,
                                 float* __restrict__ z, int n, int c, int hw) {
  const float* cf = coef + (int64_t)step_idx[0] * 8;
  const float* noise = noise_base + (int64_t)step_idx[0] * noise_step_stride;
  const float c_recip = cf[0], c_recipm1 = cf[1], pm1 = cf[2], pm2 = cf[3], logvar = cf[4], nonzero = cf[5];
  const float sigma = nonzero * expf(0.5f * logvar);
  const int64_t total = (int64_t)n * c * hw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = idx % hw;
    const int64_t fc = idx / hw;
    const int ch = (int)(fc % c);
    const int64_t f = fc / c;
    const float e = ld_eps > 0 ? eps[(f * hw + pix) * ld_eps + ch] : eps[idx];  // ld_eps<=0: eps is NCHW like x
    const float xv = x[idx];
    const float x0 = c_recip * xv - c_recipm1 * e;
    const float mean = pm1 * x0 + pm2 * xv;
    z[idx] = mean + sigma * noise[idx];
  }
}

// bilinear tap helper: zeros padding, align_corners=True (pixel coordinates)
struct Taps {
  int x0, y0;
  float w00, w01, w10, w11;  // (y0,x0) (y0,x1) (y1,x0) (y1,x1), already zeroed when out of range
};
__device__ __forceinline__ Taps make_taps(float sx, float sy, int h, int w) {
  Taps t;
  const float fx = floorf(sx), fy = floorf(sy);
  // guard against non-finite / huge coordinates before the int conversion
  const float cx = fminf(fmaxf(fx, -2.f), (float)w + 1.f), cy = fminf(fmaxf(fy, -2.f), (float)h + 1.f);
  t.x0 = (int)cx; t.y0 = (int)cy;
  const float ax = sx - fx, ay = sy - fy;
  const bool inx0 = (fx == cx) && t.x0 >= 0 && t.x0 < w, inx1 = (fx == cx) && t.x0 + 1 >= 0 && t.x0 + 1 < w;
  const bool iny0 = (fy == cy) && t.y0 >= 0 && t.y0 < h, iny1 = (fy == cy) && t.y0 + 1 >= 0 && t.y0 + 1 < h;
  t.w00 = (iny0 && inx0) ? (1.f - ay) * (1.f - ax) : 0.f;
  t.w01 = (iny0 && inx1) ? (1.f - ay) * ax : 0.f;
  t.w10 = (iny1 && inx0) ? ay * (1.f - ax) : 0.f;
  t.w11 = (iny1 && inx1) ? ay * ax : 0.f;
  return t;
}
__device__ __forceinline__ float sample_taps(const float* __restrict__ img, const Taps& t, int w) {
  float v = 0.f;
  if (t.w00 != 0.f) v += t.w00 * img[t.y0 * w + t.x0];
  if (t.w01 != 0.f) v += t.w01 * img[t.y0 * w + t.x0 + 1];
  if (t.w10 != 0.f) v += t.w10 * img[(t.y0 + 1) * w + t.x0];
  if (t.w11 != 0.f) v += t.w11 * img[(t.y0 + 1) * w + t.x0 + 1];
  return v;
}

__global__ void flow_warp_kernel(const float* __restrict__ in, const float* __restrict__ flow, float* __restrict__ out, int n,
                                 int c, int h, int w) {
  const int hw = h * w;
  const int64_t total = (int64_t)n * hw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(idx / hw);
    const int pix = (int)(idx - (int64_t)f * hw);
    const int y = pix / w, x = pix - y * w;
    const float sx = (float)x + flow[((int64_t)f * 2 + 0) * hw + pix];
    const float sy = (float)y + flow[((int64_t)f * 2 + 1) * hw + pix];
    const Taps t = make_taps(sx, sy, h, w);
    for (int ch = 0; ch < c; ++ch) out[((int64_t)f * c + ch) * hw + pix] = sample_taps(in + ((int64_t)f * c + ch) * hw, t, w);
  }
}

// ---- motion guidance (ddpm.py:3538-3574 + autograd at 4367-4373), closed form ----------------------------------
// P_j = warp(z_j, flow_bwd_prop[j]) (j<=T-2; P_{T-1}=0) ; Q_j = warp(z_j, flow_fwd_prop[j-1]) (j>=1; Q_0=0)
__global__ void guid_warp_kernel(const float* __restrict__ z, const float* __restrict__ ff, const float* __restrict__ fb,
                                 float* __restrict__ P, float* __restrict__ Q, long long* __restrict__ G, int T, int c, int h,
                                 int w) {
  const int hw = h * w;
  const int64_t total = (int64_t)T * hw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(idx / hw);
    const int pix = (int)(idx - (int64_t)j * hw);
    const int y = pix / w, x = pix - y * w;
    Taps tp, tq;
    const bool hasP = j <= T - 2, hasQ = j >= 1;
    if (hasP) tp = make_taps((float)x + fb[((int64_t)j * 2) * hw + pix], (float)y + fb[((int64_t)j * 2 + 1) * hw + pix], h, w);
    if (hasQ)
      tq = make_taps((float)x + ff[((int64_t)(j - 1) * 2) * hw + pix], (float)y + ff[((int64_t)(j - 1) * 2 + 1) * hw + pix], h, w);
    for (int ch = 0; ch < c; ++ch) {
      const int64_t o = ((int64_t)j * c + ch) * hw + pix;
      const float* img = z + ((int64_t)j * c + ch) * hw;
      P[o] = hasP ? sample_taps(img, tp, w) : 0.f;
      Q[o] = hasQ ? sample_taps(img, tq, w) : 0.f;
      G[o] = 0;
    }
  }
}

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__device__ __forceinline__ void splat(long long* __restrict__ g, const Taps& t, int w, float val) {
  // deterministic fixed-point (2^32) accumulation of the bilinear adjoint
  const double sc = 4294967296.0;
  if (t.w00 != 0.f) atomicAdd((unsigned long long*)(g + t.y0 * w + t.x0), (unsigned long long)(long long)llrint((double)(t.w00 * val) * sc));
  if (t.w01 != 0.f) atomicAdd((unsigned long long*)(g + t.y0 * w + t.x0 + 1), (unsigned long long)(long long)llrint((double)(t.w01 * val) * sc));
  if (t.w10 != 0.f) atomicAdd((unsigned long long*)(g + (t.y0 + 1) * w + t.x0), (unsigned long long)(long long)llrint((double)(t.w10 * val) * sc));
  if (t.w11 != 0.f) atomicAdd((unsigned long long*)(g + (t.y0 + 1) * w + t.x0 + 1), (unsigned long long)(long long)llrint((double)(t.w11 * val) * sc));
}

// scatter: for j in 1..T-2, the upstream gradients of P_j (used by frame j-1) and Q_j (used by frame j+1)
__global__ void guid_scatter_kernel(const float* __restrict__ z, const float* __restrict__ ff, const float* __restrict__ fb,
                                    const float* __restrict__ focc, const float* __restrict__ bocc,
                                    const float* __restrict__ P, const float* __restrict__ Q, long long* __restrict__ G, int T,
                                    int c, int h, int w) {
  const int hw = h * w;
  const int64_t total = (int64_t)(T - 2) * hw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int j = 1 + (int)(idx / hw);
    const int pix = (int)(idx % hw);
    const int y = pix / w, x = pix - y * w;
    const Taps tp = make_taps((float)x + fb[((int64_t)j * 2) * hw + pix], (float)y + fb[((int64_t)j * 2 + 1) * hw + pix], h, w);
    const Taps tq =
        make_taps((float)x + ff[((int64_t)(j - 1) * 2) * hw + pix], (float)y + ff[((int64_t)(j - 1) * 2 + 1) * hw + pix], h, w);
    const float mf = 1.f - focc[(int64_t)(j - 1) * hw + pix];  // mask of term_b(j-1)
    const float mb = 1.f - bocc[(int64_t)j * hw + pix];        // mask of term_f(j+1)
    for (int ch = 0; ch < c; ++ch) {
      const int64_t o = ((int64_t)j * c + ch) * hw + pix;
      const float zp = z[((int64_t)(j - 1) * c + ch) * hw + pix];
      const float zn = z[((int64_t)(j + 1) * c + ch) * hw + pix];
      const float gP = mf * sgn(mf * P[o] - mf * zp);
      const float gQ = mb * sgn(mb * Q[o] - mb * zn);
      long long* gplane = G + ((int64_t)j * c + ch) * hw;
      if (gP != 0.f) splat(gplane, tp, w, gP);
      if (gQ != 0.f) splat(gplane, tq, w, gQ);
    }
  }
}

__global__ void guid_apply_kernel(const float* __restrict__ z, const float* __restrict__ focc, const float* __restrict__ bocc,
                                  const float* __restrict__ P, const float* __restrict__ Q, const long long* __restrict__ G,
                                  const float* __restrict__ coef, const int32_t* __restrict__ step_idx, float gscale,
                                  float* __restrict__ xout, int T, int c, int h, int w) {
  const int hw = h * w;
  const float logvar = coef[(int64_t)step_idx[0] * 8 + 4];
  const float inv_cnt = 1.f / (float)((int64_t)c * hw);
  const int64_t total = (int64_t)T * c * hw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int pix = (int)(idx % hw);
    const int64_t fc = idx / hw;
    const int ch = (int)(fc % c);
    const int j = (int)(fc / c);
    const float zv = z[idx];
    float g = 0.f;
    if (j <= T - 2) {
      const float m = 1.f - focc[(int64_t)j * hw + pix];
      g -= m * sgn(m * P[((int64_t)(j + 1) * c + ch) * hw + pix] - m * zv);
    }
    if (j >= 1) {
      const float m = 1.f - bocc[(int64_t)(j - 1) * hw + pix];
      g -= m * sgn(m * Q[((int64_t)(j - 1) * c + ch) * hw + pix] - m * zv);
    }
    g += (float)((double)G[idx] * (1.0 / 4294967296.0));
    xout[idx] = zv - gscale * logvar * (g * inv_cnt);
  }
}

__global__ __launch_bounds__(256) void guid_loss_kernel(const float* __restrict__ z, const float* __restrict__ focc,
                                                        const float* __restrict__ bocc, const float* __restrict__ P,
                                                        const float* __restrict__ Q, double* __restrict__ acc, int T, int c,
                                                        int h, int w) {
  __shared__ double red[4];
  const int hw = h * w;
  const int64_t total = (int64_t)T * c * hw;
  double s = 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int pix = (int)(idx % hw);
    const int64_t fc = idx / hw;
    const int ch = (int)(fc % c);
    const int j = (int)(fc / c);
    const float zv = z[idx];
    if (j <= T - 2) {
      const float m = 1.f - focc[(int64_t)j * hw + pix];
      s += (double)fabsf(m * P[((int64_t)(j + 1) * c + ch) * hw + pix] - m * zv);
    }
    if (j >= 1) {
      const float m = 1.f - bocc[(int64_t)(j - 1) * hw + pix];
      s += (double)fabsf(m * Q[((int64_t)(j - 1) * c + ch) * hw + pix] - m * zv);
    }
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, red[0] + red[1] + red[2] + red[3]);
}
__global__ void guid_loss_final_kernel(const double* __restrict__ acc, float* __restrict__ out, double inv_cnt) {
  out[0] = (float)(acc[0] * inv_cnt);
}
__global__ void zero_double_kernel(double* p) { p[0] = 0.0; }

__global__ void step_advance_kernel(int32_t* s, int d) { s[0] += d; }
__global__ void step_timestep_kernel(const float* coef, const int32_t* s, float* tv, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tv[i] = coef[(int64_t)s[0] * 8 + 6];
}

// ---- forward/backward consistency (util_flow.py:114-136) ---------------------------------------------------------
__global__ void fb_consistency_kernel(const float* __restrict__ fwd, const float* __restrict__ bwd, float alpha, float beta,
                                      float* __restrict__ focc, float* __restrict__ bocc, int n, int h, int w) {
  const int hw = h * w;
  const int64_t total = (int64_t)n * hw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(idx / hw);
    const int pix = (int)(idx - (int64_t)f * hw);
    const int y = pix / w, x = pix - y * w;
    const float* fx = fwd + ((int64_t)f * 2) * hw;
    const float* fy = fx + hw;
    const float* bx = bwd + ((int64_t)f * 2) * hw;
    const float* by = bx + hw;
    const float fu = fx[pix], fv = fy[pix], bu = bx[pix], bv = by[pix];
    const float mag = sqrtf(fu * fu + fv * fv) + sqrtf(bu * bu + bv * bv);
    const Taps tf = make_taps((float)x + fu, (float)y + fv, h, w);  // warp bwd by fwd
    const Taps tb = make_taps((float)x + bu, (float)y + bv, h, w);  // warp fwd by bwd
    const float wbu = sample_taps(bx, tf, w), wbv = sample_taps(by, tf, w);
    const float wfu = sample_taps(fx, tb, w), wfv = sample_taps(fy, tb, w);
    const float d_f = sqrtf((fu + wbu) * (fu + wbu) + (fv + wbv) * (fv + wbv));
    const float d_b = sqrtf((bu + wfu) * (bu + wfu) + (bv + wfv) * (bv + wfv));
    const float thr = alpha * mag + beta;
    focc[idx] = d_f > thr ? 1.f : 0.f;
    bocc[idx] = d_b > thr ? 1.f : 0.f;
  }
}

// bilinear interpolate, align_corners=False (PyTorch area_pixel_compute_source_index semantics), flow rescaled
__global__ void resize_flow_kernel(const float* __restrict__ flow, float* __restrict__ out, int n, int h, int w, int oh, int ow) {
  const float sh = (float)h / (float)oh, sw = (float)w / (float)ow;
  const float rh = (float)oh / (float)h, rw = (float)ow / (float)w;
  const int64_t total = (int64_t)n * 2 * oh * ow;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % ow);
    const int oy = (int)((idx / ow) % oh);
    const int64_t pl = idx / ((int64_t)oh * ow);
    const int ch = (int)(pl & 1);
    float sy = ((float)oy + 0.5f) * sh - 0.5f; if (sy < 0.f) sy = 0.f;
    float sx = ((float)ox + 0.5f) * sw - 0.5f; if (sx < 0.f) sx = 0.f;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float* p = flow + pl * h * w;
    const float scale = ch == 0 ? rw : rh;
    const float v = (1.f - ly) * ((1.f - lx) * p[y0 * w + x0] * scale + lx * p[y0 * w + x1] * scale) +
                    ly * ((1.f - lx) * p[y1 * w + x0] * scale + lx * p[y1 * w + x1] * scale);
    out[idx] = v;
  }
}

// ---- AdaIN (wavelet_color_fix.py:44-71): per-plane mean / unbiased var, fp64 sums ----------------------------------------
// HBM-bound (12 B per element: content and style read once, output written once), so the planes are cut into chunks that fill the
// chip: launch 1 = one block per (chunk, plane, tensor) -> fp64 partial (sum, sum of squares); launch 2 = the apply blocks finish the
// reduction over a plane's <= 64 chunks in their prologue (no finalize launch) and stream content -> out with 16-byte accesses.
// (Round 2 ran ONE block per plane: 24 blocks on 256 CUs, 0.013 of the HBM peak.)
// work: [2 tensors][planes][AD_MAXCH chunks][2] doubles
constexpr int AD_MAXCH = 64;
constexpr int64_t AD_CHUNK = 16384;     // elements per chunk (64 KiB): 512^2 planes -> 16 chunks, 24 planes x 2 tensors -> 768 blocks
inline int adain_chunks(int64_t hw) {
  int64_t c = (hw + AD_CHUNK - 1) / AD_CHUNK;
  return (int)(c < 1 ? 1 : (c > AD_MAXCH ? AD_MAXCH : c));
}
__global__ __launch_bounds__(256) void plane_partial_kernel(const float* __restrict__ content, const float* __restrict__ style, int64_t hw,
                                                            int chunks, int planes, double* __restrict__ part) {
  __shared__ double rs[4], rq[4];
  const int chunk = blockIdx.x, plane = blockIdx.y, tz = blockIdx.z;
  const float* p = (tz ? style : content) + (int64_t)plane * hw;
  const int64_t per = (((hw + chunks - 1) / chunks) + 3) & ~(int64_t)3;
  const int64_t i0 = (int64_t)chunk * per, i1 = min(hw, i0 + per);
  double s = 0.0, q = 0.0;
  if ((hw & 3) == 0 && ((((uintptr_t)p) & 15) == 0)) {        // planes start 16-byte aligned: 16-byte loads (i0 % 4 == 0, i1 % 4 == 0)
    for (int64_t i = i0 + (int64_t)threadIdx.x * 4; i < i1; i += 1024) {
      const f32x4 v = *(const f32x4*)(p + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) { const double d = v[k]; s += d; q += d * d; }
    }
  } else {
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) { const double v = p[i]; s += v; q += v * v; }
  }
  s = wave_sum_d(s); q = wave_sum_d(q);
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rq[threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = part + (((int64_t)tz * planes + plane) * AD_MAXCH + chunk) * 2;
    o[0] = rs[0] + rs[1] + rs[2] + rs[3];
    o[1] = rq[0] + rq[1] + rq[2] + rq[3];
  }
}
__global__ __launch_bounds__(256) void adain_apply_kernel(const float* __restrict__ content, const double* __restrict__ part,
                                                          float* __restrict__ out, int planes, int64_t hw, int chunks, float eps) {
  __shared__ float st[4];   // content mean, content std, style mean, style std
  const int pl = blockIdx.y;
  if (threadIdx.x < 64) {   // one wave finishes both reductions (chunks <= 64)
    const int t = threadIdx.x;
    const double* pc = part + ((int64_t)pl * AD_MAXCH + t) * 2;
    const double* ps = part + (((int64_t)planes + pl) * AD_MAXCH + t) * 2;
    double cs_ = t < chunks ? pc[0] : 0.0, cq = t < chunks ? pc[1] : 0.0, ss_ = t < chunks ? ps[0] : 0.0, sq = t < chunks ? ps[1] : 0.0;
    cs_ = wave_sum_d(cs_); cq = wave_sum_d(cq); ss_ = wave_sum_d(ss_); sq = wave_sum_d(sq);
    if (t == 0) {
      const double cm = cs_ / (double)hw, cv = (cq - cs_ * cm) / (double)(hw - 1);   // unbiased (Tensor.var default)
      const double sm = ss_ / (double)hw, sv = (sq - ss_ * sm) / (double)(hw - 1);
      st[0] = (float)cm; st[1] = sqrtf((float)cv + eps); st[2] = (float)sm; st[3] = sqrtf((float)sv + eps);
    }
  }
  __syncthreads();
  const float cm = st[0], cs = st[1], sm = st[2], ss = st[3];
  const float* c = content + (int64_t)pl * hw;
  float* o = out + (int64_t)pl * hw;
  if ((hw & 3) == 0 && (((((uintptr_t)c) | ((uintptr_t)o)) & 15) == 0)) {
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < hw; i += (int64_t)gridDim.x * 1024) {
      const f32x4 v = *(const f32x4*)(c + i);
      *(f32x4*)(o + i) = f32x4{(v[0] - cm) / cs * ss + sm, (v[1] - cm) / cs * ss + sm, (v[2] - cm) / cs * ss + sm, (v[3] - cm) / cs * ss + sm};
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) o[i] = (c[i] - cm) / cs * ss + sm;
  }
}

// ---- a-trous wavelet blur (wavelet_color_fix.py:73-119): 3x3 [1 2 1]^2/16, dilation r, replicate padding ---------
__global__ void wavelet_blur_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int h, int w, int r) {
  const int64_t total = (int64_t)planes * h * w;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % w);
    const int y = (int)((idx / w) % h);
    const float* p = in + (idx / ((int64_t)h * w)) * h * w;
    const int ym = max(y - r, 0), yp = min(y + r, h - 1), xm = max(x - r, 0), xp = min(x + r, w - 1);
    // same accumulation order as a 3x3 cross-correlation (row-major taps)
    float v = 0.0625f * p[ym * w + xm] + 0.125f * p[ym * w + x] + 0.0625f * p[ym * w + xp];
    v += 0.125f * p[y * w + xm] + 0.25f * p[y * w + x] + 0.125f * p[y * w + xp];
    v += 0.0625f * p[yp * w + xm] + 0.125f * p[yp * w + x] + 0.0625f * p[yp * w + xp];
    out[idx] = v;
  }
}
// hi += (img - low)
__global__ void wavelet_acc_kernel(const float* __restrict__ img, const float* __restrict__ low, float* __restrict__ hi,
                                   int64_t total, int first) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const float d = img[idx] - low[idx];
    hi[idx] = first ? d : hi[idx] + d;
  }
}
__global__ void add2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x)
    o[idx] = a[idx] + b[idx];
}

// ---- aggregation-sampling tiles -------------------------------------------------------------------------------
__global__ void crop_kernel(const float* __restrict__ src, float* __restrict__ dst, int nc, int H, int W, int y0, int x0, int th,
                            int tw) {
  const int64_t total = (int64_t)nc * th * tw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % tw);
    const int y = (int)((idx / tw) % th);
    const int64_t pl = idx / ((int64_t)th * tw);
    dst[idx] = src[(pl * H + y0 + y) * W + x0 + x];
  }
}
__global__ void tile_acc_kernel(const float* __restrict__ tile, const float* __restrict__ wgt, float* __restrict__ acc,
                                float* __restrict__ cnt, int nc, int H, int W, int y0, int x0, int th, int tw) {
  const int64_t total = (int64_t)nc * th * tw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % tw);
    const int y = (int)((idx / tw) % th);
    const int64_t pl = idx / ((int64_t)th * tw);
    const float wv = wgt[y * tw + x];
    const int64_t o = (pl * H + y0 + y) * W + x0 + x;
    acc[o] += tile[idx] * wv;
    cnt[o] += wv;
  }
}
__global__ void tile_norm_kernel(const float* __restrict__ acc, const float* __restrict__ cnt, float* __restrict__ out,
                                 int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x)
    out[idx] = acc[idx] / cnt[idx];
}

}  // namespace

#define S_(s) ((hipStream_t)(s))

extern "C" int mgld_linear_small(const float* a, int lda, const void* w, int ldw, const float* b, float* y, int ldy, int M,
                                 int N, int K, int silu_in, int silu_out, void* stream) {
  MGLD_REQUIRE(a && w && y, "linear_small: null pointer");
  MGLD_REQUIRE(M > 0 && M <= 16 && N > 0 && K > 0, "linear_small: M must be in 1..16");
  MGLD_REQUIRE((K & 7) == 0 && (ldw & 7) == 0 && (lda & 3) == 0, "linear_small: K%8, ldw%8, lda%4");
  MGLD_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0, "linear_small: alignment");
  dim3 grid(cdiv(N, 4));
#define LS(MT)                                                                                                         \
  hipLaunchKernelGGL((linear_small_kernel<MT>), grid, dim3(256), 0, S_(stream), a, lda, (const f16*)w, ldw, b, y, ldy, M, \
                     N, K, silu_in, silu_out)
  if (M <= 1) LS(1);
  else if (M <= 2) LS(2);
  else if (M <= 4) LS(4);
  else if (M <= 8) LS(8);
  else LS(16);
#undef LS
  return mgld_check_launch("linear_small");
}

extern "C" int mgld_timestep_embedding(const float* tvals, int t_stride, float* out, int M, int dim, void* stream) {
  MGLD_REQUIRE(tvals && out && M > 0 && dim > 0, "timestep_embedding: bad args");
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(cdiv((int64_t)M * dim, 256)), dim3(256), 0, S_(stream), tvals, t_stride,
                     out, M, dim);
  return mgld_check_launch("timestep_embedding");
}

extern "C" int mgld_nchw_to_nhwc(const float* x, void* y, int n, int c, int h, int w, int cpad, int ld, void* stream) {
  MGLD_REQUIRE(x && y && n > 0 && c > 0 && h > 0 && w > 0, "nchw_to_nhwc: bad args");
  MGLD_REQUIRE((cpad & 7) == 0 && cpad >= c && (ld & 7) == 0 && ld >= cpad, "nchw_to_nhwc: cpad/ld");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(egrid((int64_t)n * h * w)), dim3(256), 0, S_(stream), x, (f16*)y, n, c, h * w,
                     cpad, ld);
  return mgld_check_launch("nchw_to_nhwc");
}

extern "C" int mgld_nhwc_to_nchw(const void* x, int in_f32, int ld, float* y, int n, int c, int h, int w, void* stream) {
  MGLD_REQUIRE(x && y && n > 0 && c > 0 && h > 0 && w > 0 && ld >= c, "nhwc_to_nchw: bad args");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(egrid((int64_t)n * c * h * w)), dim3(256), 0, S_(stream), x, in_f32, ld, y, n, c,
                     h * w);
  return mgld_check_launch("nhwc_to_nchw");
}

// packed 3x3 conv weights [N, 9*Cin] (K order (tap, Cin), or (64-channel block, tap, channel) when tap_inner) -> the patch kernels' tiled
// image [N64/64][Cin/32][3 dy][4 row groups][3 dx][16 rows][4 chunks][8] (MgldIGemm.tap_inner = 2; the 16-byte slot c of tile row r holds
// logical chunk c ^ ((r >> 2) & 3), the kernels' bank swizzle; rows >= N are zero).  One thread per 16-byte chunk.
__global__ __launch_bounds__(256) void tile_conv3p_kernel(const f16* __restrict__ wp, int N, int Cin, int tap_inner, f16* __restrict__ out,
                                                          int64_t chunks) {
  const int nsl = Cin >> 5;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < chunks; o += (int64_t)gridDim.x * 256) {
    const int c = (int)(o & 3), row = (int)((o >> 2) & 15);
    int64_t r = o >> 6;
    const int dx = (int)(r % 3); r /= 3;
    const int rb = (int)(r & 3); r >>= 2;
    const int dy = (int)(r % 3); r /= 3;
    const int sl = (int)(r % nsl);
    const int g64 = (int)(r / nsl);
    const int n = g64 * 64 + rb * 16 + row, tap = dy * 3 + dx;
    const int ch = sl * 32 + ((c ^ ((row >> 2) & 3)) << 3);
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (n < N) {
      const int64_t k = tap_inner ? ((int64_t)(ch >> 6) * 9 + tap) * 64 + (ch & 63) : (int64_t)tap * Cin + ch;
      v = *(const f16x8*)(wp + (int64_t)n * 9 * Cin + k);
    }
    *(f16x8*)(out + o * 8) = v;
  }
}

extern "C" int mgld_tile_conv3p(const void* wp, int N, int Cin, int tap_inner, void* out, void* stream) {
  MGLD_REQUIRE(wp && out && N > 0 && Cin > 0 && (Cin & 31) == 0 && (!tap_inner || (Cin & 63) == 0), "tile_conv3p: Cin % 32 (64 with tap_inner)");
  MGLD_REQUIRE(((((uintptr_t)wp) | ((uintptr_t)out)) & 15) == 0, "tile_conv3p: alignment");
  const int64_t chunks = (int64_t)((N + 63) / 64 * 64) * 9 * Cin / 8;
  hipLaunchKernelGGL(tile_conv3p_kernel, dim3(egrid(chunks)), dim3(256), 0, S_(stream), (const f16*)wp, N, Cin, tap_inner ? 1 : 0, (f16*)out,
                     chunks);
  return mgld_check_launch("tile_conv3p");
}

extern "C" int mgld_copy2d(const void* src, int lds_, void* dst, int ldd, int64_t rows, int cols, void* stream) {
  MGLD_REQUIRE(src && dst && rows > 0 && cols > 0, "copy2d: bad args");
  MGLD_REQUIRE((cols & 7) == 0 && (lds_ & 7) == 0 && (ldd & 7) == 0, "copy2d: cols/ld % 8");
  hipLaunchKernelGGL(copy2d_kernel, dim3(egrid(rows * (cols >> 3))), dim3(256), 0, S_(stream), (const f16*)src, lds_, (f16*)dst,
                     ldd, rows, cols >> 3);
  return mgld_check_launch("copy2d");
}

// ---- segment prologue / epilogue arithmetic that used to run as vendor elementwise kernels -----------------------------
// init = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise)  (distributions.py:24-40 sample(), ddpm.py:3382-3389) and, when
// x_T != NULL, x_T = sa * init + soma * n0 (q_sample_respace, ddpm.py:403-406, with both schedule coefficients of t as scalars).
// moments: [n, 2c, hw] (mean planes, then logvar planes per frame); noise / n0 / outputs: [n, c, hw]
namespace {
__global__ void init_latent_kernel(const float* __restrict__ mom, const float* __restrict__ noise, const float* __restrict__ n0,
                                   float* __restrict__ init, float* __restrict__ xT, int n, int c, int64_t hw, float scale, float sa,
                                   float soma) {
  const int64_t total = (int64_t)n * c * hw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = idx / (c * hw), r = idx - f * (c * hw);
    const float mean = mom[f * 2 * c * hw + r];
    const float lv = fminf(fmaxf(mom[f * 2 * c * hw + c * hw + r], -30.f), 20.f);
    const float z = mean + expf(0.5f * lv) * noise[idx];
    const float v = scale * z;
    init[idx] = v;
    if (xT) xT[idx] = sa * v + soma * n0[idx];
  }
}
// out = clamp((x + 1) / 2, 0, 1): the scripts' final mapping of the decoded frames (oldcanvas_tile.py:471)
__global__ void to01_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = fminf(fmaxf((x[i] + 1.0f) / 2.0f, 0.f), 1.f);
}
}  // namespace

extern "C" int mgld_init_latent(const float* moments, const float* noise, const float* n0, float* init, float* x_T, int n, int c,
                                int64_t hw, float scale, float sqrt_ac, float sqrt_one_minus_ac, void* stream) {
  MGLD_REQUIRE(moments && noise && init && n > 0 && c > 0 && hw > 0 && (x_T == nullptr || n0 != nullptr), "init_latent: bad args");
  hipLaunchKernelGGL(init_latent_kernel, dim3(egrid((int64_t)n * c * hw)), dim3(256), 0, S_(stream), moments, noise, n0, init, x_T, n, c,
                     hw, scale, sqrt_ac, sqrt_one_minus_ac);
  return mgld_check_launch("init_latent");
}

extern "C" int mgld_to01(const float* x, float* out, int64_t n, void* stream) {
  MGLD_REQUIRE(x && out && n > 0, "to01: bad args");
  hipLaunchKernelGGL(to01_kernel, dim3(egrid(n)), dim3(256), 0, S_(stream), x, out, n);
  return mgld_check_launch("to01");
}

extern "C" int mgld_axpby(const void* x, int ldx, void* y, int ldy, int64_t rows, int cols, float a, float b, void* stream) {
  MGLD_REQUIRE(x && y && rows > 0 && cols > 0, "axpby: bad args");
  MGLD_REQUIRE((cols & 7) == 0 && (ldx & 7) == 0 && (ldy & 7) == 0, "axpby: cols/ld % 8");
  hipLaunchKernelGGL(axpby_kernel, dim3(egrid(rows * (cols >> 3))), dim3(256), 0, S_(stream), (const f16*)x, ldx, (f16*)y, ldy,
                     rows, cols >> 3, a, b);
  return mgld_check_launch("axpby");
}

extern "C" int mgld_axpby_lo(const void* x, const void* xlo, int ldx, void* y, void* ylo, int ldy, int64_t rows, int cols, float a, float b,
                             void* stream) {
  MGLD_REQUIRE(x && y && ylo && rows > 0 && cols > 0, "axpby_lo: bad args");
  MGLD_REQUIRE((cols & 7) == 0 && (ldx & 7) == 0 && (ldy & 7) == 0, "axpby_lo: cols/ld % 8");
  hipLaunchKernelGGL(axpby_lo_kernel, dim3(egrid(rows * (cols >> 3))), dim3(256), 0, S_(stream), (const f16*)x, (const f16*)xlo, ldx, (f16*)y,
                     (f16*)ylo, ldy, rows, cols >> 3, a, b);
  return mgld_check_launch("axpby_lo");
}

extern "C" int mgld_temporal_attention(const void* q, const void* k, const void* v, int ld, void* o, int ldo, int T, int HW,
                                       int heads, int head_dim, float scale, void* stream) {
  MGLD_REQUIRE(q && k && v && o, "temporal_attention: null pointer");
  MGLD_REQUIRE(T > 0 && T <= 16 && HW > 0 && heads > 0, "temporal_attention: T must be in 1..16");
  MGLD_REQUIRE(head_dim == 64 || head_dim == 128, "temporal_attention: head_dim 64 or 128");
  dim3 grid(cdiv((int64_t)HW * heads, 4));
  if (head_dim == 64)
    hipLaunchKernelGGL((temporal_attn_kernel<64>), grid, dim3(256), 0, S_(stream), (const f16*)q, (const f16*)k, (const f16*)v,
                       ld, (f16*)o, ldo, T, HW, heads, scale);
  else
    hipLaunchKernelGGL((temporal_attn_kernel<128>), grid, dim3(256), 0, S_(stream), (const f16*)q, (const f16*)k,
                       (const f16*)v, ld, (f16*)o, ldo, T, HW, heads, scale);
  return mgld_check_launch("temporal_attention");
}

extern "C" int mgld_ddpm_step(const float* x, const float* eps, int ld_eps, const float* noise, int64_t noise_step_stride,
                              const float* coef, const int32_t* step_idx, float* z, int n, int c, int h, int w,
                              void* stream) {
  MGLD_REQUIRE(x && eps && noise && coef && step_idx && z, "ddpm_step: null pointer");
  MGLD_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && (ld_eps <= 0 || ld_eps >= c), "ddpm_step: shape");
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(egrid((int64_t)n * c * h * w)), dim3(256), 0, S_(stream), x, eps, ld_eps, noise,
                     noise_step_stride, coef, step_idx, z, n, c, h * w);
  return mgld_check_launch("ddpm_step");
}

extern "C" int mgld_flow_warp(const float* in, const float* flow, float* out, int n, int c, int h, int w, void* stream) {
  MGLD_REQUIRE(in && flow && out && n > 0 && c > 0 && h > 0 && w > 0, "flow_warp: bad args");
  hipLaunchKernelGGL(flow_warp_kernel, dim3(egrid((int64_t)n * h * w)), dim3(256), 0, S_(stream), in, flow, out, n, c, h, w);
  return mgld_check_launch("flow_warp");
}

static int guidance_common(const float* z, const float* ff, const float* fb, void* work, int T, int c, int h, int w,
                           float** P, float** Q, long long** G, void* stream) {
  const int64_t n = (int64_t)T * c * h * w;
  *G = (long long*)work;  // 8-byte aligned region first
  *P = (float*)(*G + n);
  *Q = *P + n;
  hipLaunchKernelGGL(guid_warp_kernel, dim3(egrid((int64_t)T * h * w)), dim3(256), 0, S_(stream), z, ff, fb, *P, *Q, *G, T, c, h,
                     w);
  return MGLD_OK;
}

extern "C" int mgld_guidance(const float* z, const float* ff, const float* fb, const float* focc, const float* bocc,
                             const float* coef, const int32_t* step_idx, float gscale, float* x_out, void* work, int T, int c,
                             int h, int w, void* stream) {
  MGLD_REQUIRE(z && ff && fb && focc && bocc && coef && step_idx && x_out && work, "guidance: null pointer");
  MGLD_REQUIRE(T >= 2 && c > 0 && h > 0 && w > 0, "guidance: needs T >= 2");
  MGLD_REQUIRE(((uintptr_t)work & 7) == 0, "guidance: work must be 8-byte aligned");
  float *P, *Q;
  long long* G;
  guidance_common(z, ff, fb, work, T, c, h, w, &P, &Q, &G, stream);
  if (T > 2)
    hipLaunchKernelGGL(guid_scatter_kernel, dim3(egrid((int64_t)(T - 2) * h * w)), dim3(256), 0, S_(stream), z, ff, fb, focc,
                       bocc, P, Q, G, T, c, h, w);
  hipLaunchKernelGGL(guid_apply_kernel, dim3(egrid((int64_t)T * c * h * w)), dim3(256), 0, S_(stream), z, focc, bocc, P, Q, G,
                     coef, step_idx, gscale, x_out, T, c, h, w);
  return mgld_check_launch("guidance");
}

extern "C" int mgld_guidance_loss(const float* z, const float* ff, const float* fb, const float* focc, const float* bocc,
                                  float* loss_out, void* work, int T, int c, int h, int w, void* stream) {
  MGLD_REQUIRE(z && ff && fb && focc && bocc && loss_out && work, "guidance_loss: null pointer");
  MGLD_REQUIRE(T >= 2 && c > 0 && h > 0 && w > 0, "guidance_loss: needs T >= 2");
  MGLD_REQUIRE(((uintptr_t)work & 7) == 0, "guidance_loss: work must be 8-byte aligned");
  float *P, *Q;
  long long* G;
  guidance_common(z, ff, fb, work, T, c, h, w, &P, &Q, &G, stream);
  double* acc = (double*)G;  // G is zeroed by the warp kernel and unused by the loss path
  hipLaunchKernelGGL(zero_double_kernel, dim3(1), dim3(1), 0, S_(stream), acc);
  hipLaunchKernelGGL(guid_loss_kernel, dim3(egrid((int64_t)T * c * h * w)), dim3(256), 0, S_(stream), z, focc, bocc, P, Q, acc, T,
                     c, h, w);
  hipLaunchKernelGGL(guid_loss_final_kernel, dim3(1), dim3(1), 0, S_(stream), acc, loss_out, 1.0 / ((double)c * h * w));
  return mgld_check_launch("guidance_loss");
}

extern "C" int mgld_step_advance(int32_t* step_idx, int delta, void* stream) {
  MGLD_REQUIRE(step_idx, "step_advance: null");
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, S_(stream), step_idx, delta);
  return mgld_check_launch("step_advance");
}

extern "C" int mgld_step_timestep(const float* coef, const int32_t* step_idx, float* tvals, int n, void* stream) {
  MGLD_REQUIRE(coef && step_idx && tvals && n > 0, "step_timestep: bad args");
  hipLaunchKernelGGL(step_timestep_kernel, dim3(cdiv(n, 64)), dim3(64), 0, S_(stream), coef, step_idx, tvals, n);
  return mgld_check_launch("step_timestep");
}

extern "C" int mgld_fb_consistency(const float* fwd, const float* bwd, float alpha, float beta, float* focc, float* bocc, int n,
                                   int h, int w, void* stream) {
  MGLD_REQUIRE(fwd && bwd && focc && bocc && n > 0 && h > 0 && w > 0, "fb_consistency: bad args");
  hipLaunchKernelGGL(fb_consistency_kernel, dim3(egrid((int64_t)n * h * w)), dim3(256), 0, S_(stream), fwd, bwd, alpha, beta,
                     focc, bocc, n, h, w);
  return mgld_check_launch("fb_consistency");
}

extern "C" int mgld_resize_flow(const float* flow, float* out, int n, int h, int w, int oh, int ow, void* stream) {
  MGLD_REQUIRE(flow && out && n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "resize_flow: bad args");
  hipLaunchKernelGGL(resize_flow_kernel, dim3(egrid((int64_t)n * 2 * oh * ow)), dim3(256), 0, S_(stream), flow, out, n, h, w, oh,
                     ow);
  return mgld_check_launch("resize_flow");
}

extern "C" int mgld_adain(const float* content, const float* style, float* out, int planes, int64_t hw, float eps, float* work,
                          void* stream) {
  MGLD_REQUIRE(content && style && out && work && planes > 0 && hw > 1, "adain: bad args");
  MGLD_REQUIRE(((uintptr_t)work & 7) == 0, "adain: work alignment");
  MGLD_REQUIRE(planes <= 65535, "adain: too many planes");
  double* part = (double*)work;
  const int chunks = adain_chunks(hw);
  hipLaunchKernelGGL(plane_partial_kernel, dim3(chunks, planes, 2), dim3(256), 0, S_(stream), content, style, hw, chunks, planes, part);
  int bpp = (int)((hw + 4095) / 4096);           // apply blocks per plane: ~4 float4 per thread
  if (bpp > 256) bpp = 256;
  hipLaunchKernelGGL(adain_apply_kernel, dim3(bpp, planes), dim3(256), 0, S_(stream), content, part, out, planes, hw, chunks, eps);
  return mgld_check_launch("adain");
}

extern "C" int mgld_wavelet_reconstruction(const float* content, const float* style, float* out, int planes, int h, int w,
                                           float* work, void* stream) {
  MGLD_REQUIRE(content && style && out && work && planes > 0 && h > 0 && w > 0, "wavelet: bad args");
  const int64_t n = (int64_t)planes * h * w;
  float* a = work;          // ping
  float* b = work + n;      // pong
  float* hi = work + 2 * n; // content high-frequency accumulator
  const int g = egrid(n);
  // content: high frequency
  const float* cur = content;
  for (int i = 0; i < 5; ++i) {
    float* low = (i & 1) ? b : a;
    hipLaunchKernelGGL(wavelet_blur_kernel, dim3(g), dim3(256), 0, S_(stream), cur, low, planes, h, w, 1 << i);
    hipLaunchKernelGGL(wavelet_acc_kernel, dim3(g), dim3(256), 0, S_(stream), cur, low, hi, n, i == 0);
    cur = low;
  }
  // style: low frequency
  cur = style;
  for (int i = 0; i < 5; ++i) {
    float* low = (i & 1) ? b : a;
    hipLaunchKernelGGL(wavelet_blur_kernel, dim3(g), dim3(256), 0, S_(stream), cur, low, planes, h, w, 1 << i);
    cur = low;
  }
  hipLaunchKernelGGL(add2_kernel, dim3(g), dim3(256), 0, S_(stream), hi, cur, out, n);
  return mgld_check_launch("wavelet_reconstruction");
}

// dst[0 : n16) <- src[step * n16 : (step + 1) * n16)   (16-byte words); `step` is read from device memory so the launch is
// identical on every replay of a captured step (per-step slices of tensors that were precomputed for the whole schedule)
__global__ __launch_bounds__(256) void copy_step_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16,
                                                        const int* __restrict__ step_idx) {
  const uint4* s = src + (int64_t)step_idx[0] * n16;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) dst[i] = s[i];
}

extern "C" int mgld_copy_step(const void* src, void* dst, int64_t bytes_per_step, const int32_t* step_idx, void* stream) {
  MGLD_REQUIRE(src && dst && step_idx && bytes_per_step > 0 && (bytes_per_step & 15) == 0, "copy_step: bad args");
  MGLD_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "copy_step: 16-byte alignment");
  const int64_t n16 = bytes_per_step >> 4;
  hipLaunchKernelGGL(copy_step_kernel, dim3(egrid(n16)), dim3(256), 0, S_(stream), (const uint4*)src, (uint4*)dst, n16, step_idx);
  return mgld_check_launch("copy_step");
}

// ---- K11: host pre/post-processing moved to the device (oldcanvas_tile.py:349-357, 384-397, 523-543) ---------------
namespace {

// torch upsample_bicubic2d (align_corners=False, A=-0.75, no antialias): cubic-convolution weights of tap offsets -1..2
__device__ __forceinline__ void cubic_w(float t, float* w) {
  const float A = -0.75f;
  const float x0 = t + 1.f, x3 = 2.f - t, u = 1.f - t;
  w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
  w[2] = ((A + 2.f) * u - (A + 3.f)) * u * u + 1.f;
  w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

__global__ __launch_bounds__(256) void bicubic_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int h,
                                                      int w, int oh, int ow, float sy, float sx, float lo, float hi) {
  const int64_t total = (int64_t)planes * oh * ow;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % ow);
    const int oy = (int)((i / ow) % oh);
    const int pl = (int)(i / ((int64_t)ow * oh));
    const float fy = sy * (oy + 0.5f) - 0.5f, fx = sx * (ox + 0.5f) - 0.5f;
    const float yf = floorf(fy), xf = floorf(fx);
    float wy[4], wx[4];
    cubic_w(fy - yf, wy);
    cubic_w(fx - xf, wx);
    const int iy = (int)yf, ix = (int)xf;
    const float* p = x + (int64_t)pl * h * w;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int yy = min(max(iy - 1 + j, 0), h - 1);
      const float* row = p + (int64_t)yy * w;
      const float r = row[min(max(ix - 1, 0), w - 1)] * wx[0] + row[min(max(ix, 0), w - 1)] * wx[1] +
                      row[min(max(ix + 1, 0), w - 1)] * wx[2] + row[min(max(ix + 2, 0), w - 1)] * wx[3];
      acc += r * wy[j];
    }
    y[i] = fminf(fmaxf(acc, lo), hi);
  }
}

// torch upsample_bilinear2d (align_corners=False, no antialias) with an optional centre-crop window: output pixel (oy, ox) of the
// oh x ow window whose top-left corner sits at (cy, cx) of the rh x rw resized image
__global__ __launch_bounds__(256) void bilinear_crop_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int h, int w,
                                                            int oh, int ow, int cy, int cx, float sy, float sx) {
  const int64_t total = (int64_t)planes * oh * ow;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % ow);
    const int oy = (int)((i / ow) % oh);
    const int pl = (int)(i / ((int64_t)ow * oh));
    const float fy = fmaxf(sy * (oy + cy + 0.5f) - 0.5f, 0.f), fx = fmaxf(sx * (ox + cx + 0.5f) - 0.5f, 0.f);
    const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* p = x + (int64_t)pl * h * w;
    const float top = p[(int64_t)y0 * w + x0] * (1.f - lx) + p[(int64_t)y0 * w + x1] * lx;
    const float bot = p[(int64_t)y1 * w + x0] * (1.f - lx) + p[(int64_t)y1 * w + x1] * lx;
    y[i] = top * (1.f - ly) + bot * ly;
  }
}

// F.pad(mode="reflect") on the bottom / right edges (no edge repeat): index h+k reads h-2-k
__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int h,
                                                          int w, int oh, int ow) {
  const int64_t total = (int64_t)planes * oh * ow;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
    const int pl = (int)(i / ((int64_t)ow * oh));
    if (ox >= w) ox = 2 * (w - 1) - ox;
    if (oy >= h) oy = 2 * (h - 1) - oy;
    y[i] = x[((int64_t)pl * h + oy) * w + ox];
  }
}

// F.pad(mode="replicate") by (pl, ow-w-pl) columns and (pt, oh-h-pt) rows (RAFT's InputPadder, raft_arch.py:27-28)
__global__ __launch_bounds__(256) void replicate_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int h,
                                                            int w, int oh, int ow, int pt, int pl) {
  const int64_t total = (int64_t)planes * oh * ow;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
    const int pln = (int)(i / ((int64_t)ow * oh));
    const int sx = min(max(ox - pl, 0), w - 1), sy = min(max(oy - pt, 0), h - 1);
    y[i] = x[((int64_t)pln * h + sy) * w + sx];
  }
}

// [n,3,H,W] in [0,1] -> uint8 [n,h,w,3] (top-left crop); `(x * 255).astype(np.uint8)` of the reference = truncation
__global__ __launch_bounds__(256) void to_u8_kernel(const float* __restrict__ x, unsigned char* __restrict__ y, int n, int c,
                                                    int H, int W, int h, int w) {
  const int64_t total = (int64_t)n * h * w * c;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    const int xx = (int)((i / c) % w);
    const int yy = (int)((i / ((int64_t)c * w)) % h);
    const int f = (int)(i / ((int64_t)c * w * h));
    const float v = x[(((int64_t)f * c + ch) * H + yy) * W + xx] * 255.f;
    y[i] = (unsigned char)(int)fminf(fmaxf(v, 0.f), 255.f);
  }
}

}  // namespace

extern "C" int mgld_resize_bicubic(const float* x, float* y, int planes, int h, int w, int oh, int ow, float lo, float hi,
                                   void* stream) {
  MGLD_REQUIRE(x && y && planes > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "resize_bicubic: bad args");
  hipLaunchKernelGGL(bicubic_kernel, dim3(egrid((int64_t)planes * oh * ow)), dim3(256), 0, S_(stream), x, y, planes, h, w, oh,
                     ow, (float)h / (float)oh, (float)w / (float)ow, lo, hi);
  return mgld_check_launch("resize_bicubic");
}

extern "C" int mgld_resize_bilinear_crop(const float* x, float* y, int planes, int h, int w, int rh, int rw, int oh, int ow, int cy,
                                         int cx, void* stream) {
  MGLD_REQUIRE(x && y && planes > 0 && h > 0 && w > 0 && rh > 0 && rw > 0 && oh > 0 && ow > 0, "resize_bilinear_crop: bad args");
  MGLD_REQUIRE(cy >= 0 && cx >= 0 && cy + oh <= rh && cx + ow <= rw, "resize_bilinear_crop: window outside the resized image");
  hipLaunchKernelGGL(bilinear_crop_kernel, dim3(egrid((int64_t)planes * oh * ow)), dim3(256), 0, S_(stream), x, y, planes, h, w, oh, ow,
                     cy, cx, (float)h / (float)rh, (float)w / (float)rw);
  return mgld_check_launch("resize_bilinear_crop");
}

extern "C" int mgld_reflect_pad(const float* x, float* y, int planes, int h, int w, int oh, int ow, void* stream) {
  MGLD_REQUIRE(x && y && planes > 0 && h > 1 && w > 1, "reflect_pad: bad args");
  MGLD_REQUIRE(oh >= h && ow >= w && oh - h < h && ow - w < w, "reflect_pad: padding must be smaller than the image");
  hipLaunchKernelGGL(reflect_pad_kernel, dim3(egrid((int64_t)planes * oh * ow)), dim3(256), 0, S_(stream), x, y, planes, h, w,
                     oh, ow);
  return mgld_check_launch("reflect_pad");
}

extern "C" int mgld_replicate_pad(const float* x, float* y, int planes, int h, int w, int oh, int ow, int pt, int pl,
                                  void* stream) {
  MGLD_REQUIRE(x && y && planes > 0 && h > 0 && w > 0 && pt >= 0 && pl >= 0 && oh >= h + pt && ow >= w + pl, "replicate_pad: bad args");
  hipLaunchKernelGGL(replicate_pad_kernel, dim3(egrid((int64_t)planes * oh * ow)), dim3(256), 0, S_(stream), x, y, planes, h, w,
                     oh, ow, pt, pl);
  return mgld_check_launch("replicate_pad");
}

extern "C" int mgld_to_uint8_hwc(const float* x, void* y, int n, int c, int H, int W, int h, int w, void* stream) {
  MGLD_REQUIRE(x && y && n > 0 && c > 0 && h > 0 && w > 0 && h <= H && w <= W, "to_uint8_hwc: bad args");
  hipLaunchKernelGGL(to_u8_kernel, dim3(egrid((int64_t)n * c * h * w)), dim3(256), 0, S_(stream), x, (unsigned char*)y, n, c,
                     H, W, h, w);
  return mgld_check_launch("to_uint8_hwc");
}

extern "C" int mgld_crop(const float* src, float* dst, int n, int c, int H, int W, int y0, int x0, int th, int tw, void* stream) {
  MGLD_REQUIRE(src && dst && n > 0 && c > 0, "crop: bad args");
  MGLD_REQUIRE(y0 >= 0 && x0 >= 0 && y0 + th <= H && x0 + tw <= W && th > 0 && tw > 0, "crop: window out of range");
  hipLaunchKernelGGL(crop_kernel, dim3(egrid((int64_t)n * c * th * tw)), dim3(256), 0, S_(stream), src, dst, n * c, H, W, y0, x0,
                     th, tw);
  return mgld_check_launch("crop");
}

extern "C" int mgld_tile_accumulate(const float* tile, const float* wgt, float* acc, float* cnt, int n, int c, int H, int W,
                                    int y0, int x0, int th, int tw, void* stream) {
  MGLD_REQUIRE(tile && wgt && acc && cnt && n > 0 && c > 0, "tile_accumulate: bad args");
  MGLD_REQUIRE(y0 >= 0 && x0 >= 0 && y0 + th <= H && x0 + tw <= W && th > 0 && tw > 0, "tile_accumulate: window out of range");
  hipLaunchKernelGGL(tile_acc_kernel, dim3(egrid((int64_t)n * c * th * tw)), dim3(256), 0, S_(stream), tile, wgt, acc, cnt, n * c,
                     H, W, y0, x0, th, tw);
  return mgld_check_launch("tile_accumulate");
}

extern "C" int mgld_tile_normalize(const float* acc, const float* cnt, float* out, int64_t numel, void* stream) {
  MGLD_REQUIRE(acc && cnt && out && numel > 0, "tile_normalize: bad args");
  hipLaunchKernelGGL(tile_norm_kernel, dim3(egrid(numel)), dim3(256), 0, S_(stream), acc, cnt, out, numel);
  return mgld_check_launch("tile_normalize");
}
