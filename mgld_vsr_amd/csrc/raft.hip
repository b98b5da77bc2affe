// raft.hip — the RAFT_SR flow estimator (basicsr/archs/raft_arch.py), entirely in fp32.
// Round 5: the flows feed a THRESHOLDED forward/backward consistency check and sub-pixel warps of the latents; with fp16
// activations through the recurrent update block they sat 1.6-2.1e-3 from the reference's and flipped occlusion-mask pixels,
// which moved the sampled latents by 4.8e-3.  RAFT is ~0.07 % of a segment's arithmetic (0.3 of 435 TFLOP at 8 x 512^2), so it
// does not need the fp16 matrix rate: every contraction runs on the f32-input MFMA (`v_mfma_f32_32x32x2_f32`, exact fp32 fma
// chains at the fp32 vector rate, 157 TF peak) in `convf32_kernel`, every activation is an fp32 NHWC matrix, InstanceNorm
// keeps fp64 sums.  This file holds that kernel, the instance norm, the correlation pyramid + windowed lookup
// (raft_arch.py:37-86,519-533), the GRU gate arithmetic (:390-405), the flow update and the convex 8x upsampling (:720-731).
#include "common.h"

namespace {

inline int rgrid(int64_t total) {
  int64_t b = (total + 255) / 256;
  if (b > 65535) b = 65535;
  if (b < 1) b = 1;
  return (int)b;
}

// F.avg_pool2d(x, 2, stride=2) on [planes, h, w] fp32 (floor output size)
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t planes, int h,
                                                       int w, int oh, int ow) {
  const int64_t total = planes * oh * ow;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
    const int64_t pl = i / ((int64_t)ow * oh);
    const float* p = x + (pl * h + 2 * oy) * w + 2 * ox;
    y[i] = (p[0] + p[1] + p[w] + p[w + 1]) * 0.25f;
  }
}

// grid_sample(align_corners=True, zeros padding) of ONE plane at pixel coordinates (x, y)
__device__ __forceinline__ float sample_plane(const float* __restrict__ pl, int h, int w, float x, float y) {
  const float x0f = floorf(x), y0f = floorf(y);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const float ax = x - x0f, ay = y - y0f;
  float v = 0.f;
  const bool xin0 = (unsigned)x0 < (unsigned)w, xin1 = (unsigned)(x0 + 1) < (unsigned)w;
  if ((unsigned)y0 < (unsigned)h) {
    if (xin0) v += pl[y0 * w + x0] * (1.f - ax) * (1.f - ay);
    if (xin1) v += pl[y0 * w + x0 + 1] * ax * (1.f - ay);
  }
  if ((unsigned)(y0 + 1) < (unsigned)h) {
    if (xin0) v += pl[(y0 + 1) * w + x0] * (1.f - ax) * ay;
    if (xin1) v += pl[(y0 + 1) * w + x0 + 1] * ax * ay;
  }
  return v;
}

// CorrBlock.__call__ (raft_arch.py:54-75): out[(b,y,x)][lvl*(2r+1)^2 + i*(2r+1) + j] = bilinear(corr_lvl[(b,y,x)],
//   cx / 2^lvl + (i - r),  cy / 2^lvl + (j - r))     — the reference adds meshgrid(dy, dx) to (x, y): index i moves x.
// levels: lvl L holds [B*H*W] planes of size (hL, wL), contiguous.
struct CorrLevels {
  const float* p[4];
  int h[4], w[4];
};
__global__ __launch_bounds__(256) void corr_lookup_kernel(CorrLevels lv, int nlev, const float* __restrict__ coords, int B, int H,
                                                          int W, int r, float* __restrict__ out, int ldo) {
  const int D = 2 * r + 1;
  const int per = nlev * D * D;
  const int64_t total = (int64_t)B * H * W * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % per);
    const int64_t pix = i / per;                       // b*H*W + y*W + x
    const int b = (int)(pix / ((int64_t)H * W));
    const int yx = (int)(pix - (int64_t)b * H * W);
    const int lvl = k / (D * D);
    const int ij = k - lvl * D * D;
    const int ii = ij / D, jj = ij - ii * D;
    const float cx = coords[((int64_t)b * 2 + 0) * H * W + yx], cy = coords[((int64_t)b * 2 + 1) * H * W + yx];
    const float sc = 1.f / (float)(1 << lvl);
    const float sx = cx * sc + (float)(ii - r), sy = cy * sc + (float)(jj - r);
    const int hh = lv.h[lvl], ww = lv.w[lvl];
    out[pix * ldo + k] = sample_plane(lv.p[lvl] + pix * hh * ww, hh, ww, sx, sy);
  }
}

// rhx[:, :Ch] = r * hx[:, :Ch] ; rhx[:, Ch:Ch+Cx] = hx[:, Ch:Ch+Cx]      (cat([r*h, x]), raft_arch.py:395,402)
__global__ __launch_bounds__(256) void gru_rh_kernel(const float* __restrict__ r, int ldr, const float* __restrict__ hx, int ldhx,
                                                     float* __restrict__ rhx, int ldo, int64_t M, int Ch, int Cx) {
  const int C = Ch + Cx;
  const int64_t total = M * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t m = i / C;
    const float v = hx[m * ldhx + c];
    rhx[m * ldo + c] = c < Ch ? v * r[m * ldr + c] : v;
  }
}

// h = (1 - z) * h + z * q   in place on the h columns of hx (raft_arch.py:396,403)
__global__ __launch_bounds__(256) void gru_gate_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ q, int ldq,
                                                       float* __restrict__ h, int ldh, int64_t M, int Ch) {
  const int64_t total = M * Ch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % Ch);
    const int64_t m = i / Ch;
    const float zz = z[m * ldz + c], hh = h[m * ldh + c], qq = q[m * ldq + c];
    h[m * ldh + c] = (1.f - zz) * hh + zz * qq;
  }
}

// coords1 += delta (delta NHWC fp32 [B*H*W, ldd], columns 0,1 = dx,dy); flow = coords1 - coords0 written NCHW fp32 and as two
// fp32 columns of `mot` (the motion-feature tail, cat([out, flow]), raft_arch.py:444) and of `fin` (the flow conv input)
__global__ __launch_bounds__(256) void flow_update_kernel(float* __restrict__ coords1, const float* __restrict__ coords0,
                                                          const float* __restrict__ delta, int ldd, float* __restrict__ flow,
                                                          float* __restrict__ mot, int ldm, float* __restrict__ fin, int ldf, int B,
                                                          int HW) {
  const int64_t total = (int64_t)B * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i / HW);
    const int p = (int)(i - (int64_t)b * HW);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int64_t o = ((int64_t)b * 2 + c) * HW + p;
      float c1 = coords1[o];
      if (delta) c1 += delta[i * ldd + c];
      coords1[o] = c1;
      const float f = c1 - coords0[o];
      flow[o] = f;
      if (mot) mot[i * ldm + c] = f;
      if (fin) fin[i * ldf + c] = f;
    }
  }
}

// upsample_flow (raft_arch.py:720-731): out[b,c,8y+u,8x+v] = sum_k softmax_k(mask[b,(k,u,v),y,x]) * 8*flow[b,c,y+dy_k,x+dx_k]
// with k = 3*(dy+1)+(dx+1) (F.unfold order, zero padding); mask NHWC fp32 [B*H*W, ldm] with channel k*64 + u*8 + v.
__global__ __launch_bounds__(256) void convex_up_kernel(const float* __restrict__ flow, const float* __restrict__ mask, int ldm,
                                                        float* __restrict__ out, int B, int H, int W) {
  const int64_t total = (int64_t)B * H * W * 64;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int uv = (int)(i & 63);
    const int64_t pix = i >> 6;
    const int b = (int)(pix / ((int64_t)H * W));
    const int yx = (int)(pix - (int64_t)b * H * W);
    const int y = yx / W, x = yx - y * W;
    float m[9], mx = -1e30f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { m[k] = mask[pix * ldm + k * 64 + uv]; mx = fmaxf(mx, m[k]); }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { m[k] = expf(m[k] - mx); s += m[k]; }
    const float inv = 1.f / s;
    const int u = uv >> 3, v = uv & 7;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float* fp = flow + ((int64_t)b * 2 + c) * H * W;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        const float f = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? 8.f * fp[yy * W + xx] : 0.f;
        acc += m[k] * inv * f;
      }
      out[(((int64_t)b * 2 + c) * (8 * H) + 8 * y + u) * (8 * W) + 8 * x + v] = acc;
    }
  }
}

// ---- fp32 implicit-GEMM convolution on the f32-input MFMA -------------------------------------------------------------------
// out[m][n] = post( act( alpha * (sum_k A[m][k] * W[n][k] + bias[n]) ) + R[m][n] ),  m = (image, oy, ox), k = (tap, channel):
// the A row of an output pixel is gathered from the NHWC input (zero padding), W is [Cout][kh*kw][Cin4] fp32 (Cin4 = Cin rounded
// up to 4, zero filled).  Block = 4 waves, tile 64 pixels x 64 output channels, each wave one 32 x 32 accumulator block
// (v_mfma_f32_32x32x2_f32: lane l supplies A[l&31][l>>5], B[l>>5][l&31]; result reg r of lane l = row (r&3)+8*(r>>2)+4*(l>>5),
// column l&31).  K is walked in slices of 16 floats: every thread fetches ONE float4 of A and one of W per slice (a float4 never
// straddles a tap: Cin4 % 4 == 0) into registers while the MFMAs of the previous slice run, then stores them to one of two LDS
// buffers (row stride 17 floats: the 32 rows a wave reads per MFMA operand hit 32 different banks); one barrier per slice.
// Batched LINEAR form (kh = kw = 1, grid.z = batch, per-batch operand offsets): the all-pairs correlation fmap1 . fmap2^T.
constexpr int CF_BM = 64, CF_BN = 64, CF_BK = 16, CF_LD = 17;

__device__ __forceinline__ float act_f32(float v, int act) {
  switch (act) {
    case MGLD_ACT_RELU: return fmaxf(v, 0.f);
    case MGLD_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case MGLD_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

__global__ __launch_bounds__(256) void convf32_kernel(MgldConvF32 p) {
  __shared__ float sA[2][CF_BM * CF_LD];
  __shared__ float sW[2][CF_BN * CF_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave & 1, wc = wave >> 1;
  const int64_t m0 = (int64_t)blockIdx.x * CF_BM;
  const int n0 = blockIdx.y * CF_BN;
  const float* __restrict__ A = p.A + (int64_t)blockIdx.z * p.strideA;
  const float* __restrict__ W = p.W + (int64_t)blockIdx.z * p.strideW;
  float* __restrict__ C = p.C + (int64_t)blockIdx.z * p.strideC;
  const float* __restrict__ R = p.R ? p.R + (int64_t)blockIdx.z * p.strideC : nullptr;
  const int Cin4 = (p.Cin + 3) & ~3;
  const int K4 = p.kh * p.kw * Cin4;
  const int nk = (K4 + CF_BK - 1) / CF_BK;

  // this thread's staging slot: row lr of both tiles, float4 number lc of the slice
  const int lr = tid >> 2, lc = tid & 3;
  const int64_t m = m0 + lr;
  const bool mrow = m < p.M;
  int img = 0, oy = 0, ox = 0;
  if (mrow) {
    const int hw = p.Hout * p.Wout;
    img = (int)(m / hw);
    const int r = (int)(m - (int64_t)img * hw);
    oy = r / p.Wout;
    ox = r - oy * p.Wout;
  }
  const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
  const int nrow = n0 + lr;
  const float* __restrict__ wrow = W + (int64_t)nrow * K4;
  const bool nok = nrow < p.N;

  f32x4 ra, rw;
  auto fetch = [&](int kc) {
    const int k = kc * CF_BK + lc * 4;
    ra = f32x4{0.f, 0.f, 0.f, 0.f};
    rw = ra;
    if (k < K4) {
      const int tap = k / Cin4, c = k - tap * Cin4;
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      const int iy = iy0 + ky, ix = ix0 + kx;
      if (mrow && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) {
        const float* ap = A + ((int64_t)(img * p.Hin + iy) * p.Win + ix) * p.lda + c;
        if (c + 4 <= p.Cin) {
          ra = *(const f32x4*)ap;
        } else {                       // channel tail of an input whose width is not a multiple of 4 (the weights there are 0)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c + j < p.Cin) ra[j] = ap[j];
        }
      }
      if (nok) rw = *(const f32x4*)(wrow + k);
    }
  };
  auto stash = [&](int buf) {
    float* a = &sA[buf][lr * CF_LD + lc * 4];
    float* w = &sW[buf][lr * CF_LD + lc * 4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = ra[j]; w[j] = rw[j]; }
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int arow = (wr * 32 + (lane & 31)) * CF_LD + (lane >> 5);
  const int brow = (wc * 32 + (lane & 31)) * CF_LD + (lane >> 5);

  fetch(0);
  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    stash(buf);
    __syncthreads();
    if (kc + 1 < nk) fetch(kc + 1);
    const float* a = &sA[buf][arow];
    const float* b = &sW[buf][brow];
#pragma unroll
    for (int s = 0; s < CF_BK / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * s], b[2 * s], acc, 0, 0, 0);
  }

  const int n = n0 + wc * 32 + (lane & 31);
  if (n >= p.N) return;
  const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t mm = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (mm >= p.M) continue;
    float v = act_f32(p.alpha * (acc[r] + bv), p.act);
    if (R) v += R[mm * p.ldr + n];
    if (p.post_relu) v = fmaxf(v, 0.f);
    C[mm * p.ldc + n] = v;
  }
}

// ---- InstanceNorm2d (no affine, biased variance, eps inside the root: raft_arch.py:115-119,211) on fp32 NHWC -----------------
// two launches: per (image, row chunk, channel) fp64 (sum, sumsq) partials; the apply kernel adds the chunks in its prologue,
// normalises, applies ReLU, and optionally the ResidualBlock tail relu(skip + y) (:138).  Block = 8 row groups x 32 channels.
__global__ __launch_bounds__(256) void instnorm_part_kernel(const float* __restrict__ x, int ldx, double* __restrict__ part, int hw,
                                                            int C, int chunks) {
  __shared__ double red[2][8][32];
  const int c = blockIdx.y * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
  const int img = blockIdx.z, ch = blockIdx.x;
  const int r0 = (int)((int64_t)hw * ch / chunks), r1 = (int)((int64_t)hw * (ch + 1) / chunks);
  double s = 0.0, q = 0.0;
  if (c < C)
    for (int r = r0 + rg; r < r1; r += 8) {
      const double v = (double)x[((int64_t)img * hw + r) * ldx + c];
      s += v;
      q += v * v;
    }
  red[0][rg][threadIdx.x & 31] = s;
  red[1][rg][threadIdx.x & 31] = q;
  __syncthreads();
  if (rg == 0 && c < C) {
#pragma unroll
    for (int g = 1; g < 8; ++g) { s += red[0][g][threadIdx.x]; q += red[1][g][threadIdx.x]; }
    double* o = part + (((int64_t)img * chunks + ch) * C + c) * 2;
    o[0] = s;
    o[1] = q;
  }
}

__global__ __launch_bounds__(256) void instnorm_apply_kernel(const float* __restrict__ x, int ldx, const double* __restrict__ part,
                                                             const float* __restrict__ skip, int lds, float* __restrict__ y, int ldy,
                                                             int hw, int C, int chunks, float eps, int relu) {
  const int c = blockIdx.y * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
  const int img = blockIdx.z, ch = blockIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < chunks; ++k) {
    const double* o = part + (((int64_t)img * chunks + k) * C + c) * 2;
    s += o[0];
    q += o[1];
  }
  const double mean = s / hw;
  const double var = fmax(q / hw - mean * mean, 0.0);
  const float mu = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
  const int r0 = (int)((int64_t)hw * ch / chunks), r1 = (int)((int64_t)hw * (ch + 1) / chunks);
  for (int r = r0 + rg; r < r1; r += 8) {
    const int64_t row = (int64_t)img * hw + r;
    float v = (x[row * ldx + c] - mu) * rstd;
    if (relu) v = fmaxf(v, 0.f);
    if (skip) v = fmaxf(v + skip[row * lds + c], 0.f);
    y[row * ldy + c] = v;
  }
}

// [n,c,h,w] fp32 -> NHWC fp32 [n*h*w, ld] with columns c..ld-1 zeroed (the encoders' 3-channel input padded to 4)
__global__ __launch_bounds__(256) void nchw_to_nhwc_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int c, int hw,
                                                               int ld) {
  const int64_t total = (int64_t)n * hw * ld;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % ld);
    const int64_t pix = i / ld;
    const int img = (int)(pix / hw);
    const int r = (int)(pix - (int64_t)img * hw);
    y[i] = k < c ? x[((int64_t)img * c + k) * hw + r] : 0.f;
  }
}
}  // namespace

#define RS_(s) ((hipStream_t)(s))

extern "C" int mgld_avgpool2(const float* x, float* y, int64_t planes, int h, int w, void* stream) {
  MGLD_REQUIRE(x && y && planes > 0 && h >= 2 && w >= 2, "avgpool2: bad args");
  hipLaunchKernelGGL(avgpool2_kernel, dim3(rgrid(planes * (h / 2) * (w / 2))), dim3(256), 0, RS_(stream), x, y, planes, h, w, h / 2,
                     w / 2);
  return mgld_check_launch("avgpool2");
}

extern "C" int mgld_corr_lookup(const float* const* levels, const int* hs, const int* ws, int nlev, const float* coords, int B,
                                int H, int W, int radius, float* out, int ldo, void* stream) {
  MGLD_REQUIRE(levels && hs && ws && coords && out && nlev >= 1 && nlev <= 4 && radius >= 0 && radius <= 7, "corr_lookup: bad args");
  MGLD_REQUIRE(B > 0 && H > 0 && W > 0 && ldo >= nlev * (2 * radius + 1) * (2 * radius + 1), "corr_lookup: shape");
  CorrLevels lv;
  for (int i = 0; i < 4; ++i) {
    lv.p[i] = i < nlev ? levels[i] : nullptr;
    lv.h[i] = i < nlev ? hs[i] : 0;
    lv.w[i] = i < nlev ? ws[i] : 0;
    if (i < nlev) MGLD_REQUIRE(lv.p[i] && lv.h[i] > 0 && lv.w[i] > 0, "corr_lookup: level");
  }
  const int per = nlev * (2 * radius + 1) * (2 * radius + 1);
  hipLaunchKernelGGL(corr_lookup_kernel, dim3(rgrid((int64_t)B * H * W * per)), dim3(256), 0, RS_(stream), lv, nlev, coords, B, H,
                     W, radius, (float*)out, ldo);
  return mgld_check_launch("corr_lookup");
}

extern "C" int mgld_gru_rh(const float* r, int ldr, const float* hx, int ldhx, float* rhx, int ldo, int64_t M, int Ch, int Cx,
                           void* stream) {
  MGLD_REQUIRE(r && hx && rhx && M > 0 && Ch > 0 && Cx >= 0, "gru_rh: bad args");
  hipLaunchKernelGGL(gru_rh_kernel, dim3(rgrid(M * (Ch + Cx))), dim3(256), 0, RS_(stream), (const float*)r, ldr, (const float*)hx, ldhx,
                     (float*)rhx, ldo, M, Ch, Cx);
  return mgld_check_launch("gru_rh");
}

extern "C" int mgld_gru_gate(const float* z, int ldz, const float* q, int ldq, float* h, int ldh, int64_t M, int Ch, void* stream) {
  MGLD_REQUIRE(z && q && h && M > 0 && Ch > 0, "gru_gate: bad args");
  hipLaunchKernelGGL(gru_gate_kernel, dim3(rgrid(M * Ch)), dim3(256), 0, RS_(stream), (const float*)z, ldz, (const float*)q, ldq,
                     (float*)h, ldh, M, Ch);
  return mgld_check_launch("gru_gate");
}

extern "C" int mgld_flow_update(float* coords1, const float* coords0, const float* delta, int ldd, float* flow, float* mot, int ldm,
                                float* fin, int ldf, int B, int HW, void* stream) {
  MGLD_REQUIRE(coords1 && coords0 && flow && B > 0 && HW > 0, "flow_update: bad args");
  hipLaunchKernelGGL(flow_update_kernel, dim3(rgrid((int64_t)B * HW)), dim3(256), 0, RS_(stream), coords1, coords0, delta, ldd, flow,
                     (float*)mot, ldm, (float*)fin, ldf, B, HW);
  return mgld_check_launch("flow_update");
}

extern "C" int mgld_convex_upsample(const float* flow, const float* mask, int ldm, float* out, int B, int H, int W, void* stream) {
  MGLD_REQUIRE(flow && mask && out && B > 0 && H > 0 && W > 0 && ldm >= 576, "convex_upsample: bad args");
  hipLaunchKernelGGL(convex_up_kernel, dim3(rgrid((int64_t)B * H * W * 64)), dim3(256), 0, RS_(stream), flow, (const float*)mask, ldm,
                     out, B, H, W);
  return mgld_check_launch("convex_upsample");
}

extern "C" int mgld_conv_f32(const MgldConvF32* pp, void* stream) {
  MGLD_REQUIRE(pp && pp->A && pp->W && pp->C, "conv_f32: null operand");
  const MgldConvF32& p = *pp;
  MGLD_REQUIRE(p.M > 0 && p.N > 0 && p.Cin > 0 && p.kh >= 1 && p.kw >= 1 && p.stride >= 1 && p.batch >= 1, "conv_f32: shape");
  MGLD_REQUIRE(p.Hin > 0 && p.Win > 0 && p.Hout > 0 && p.Wout > 0 && p.M % ((int64_t)p.Hout * p.Wout) == 0, "conv_f32: geometry");
  MGLD_REQUIRE(p.lda >= p.Cin && p.ldc >= p.N && (!p.R || p.ldr >= p.N), "conv_f32: leading dimensions");
  // float4 fetches: 16-byte aligned rows on both operands (inputs of any width are read with a scalar channel tail)
  MGLD_REQUIRE(p.lda % 4 == 0 && ((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0 && p.strideA % 4 == 0 && p.strideW % 4 == 0,
               "conv_f32: operands must be 16-byte aligned with lda % 4 == 0");
  MGLD_REQUIRE(p.act == MGLD_ACT_NONE || p.act == MGLD_ACT_RELU || p.act == MGLD_ACT_SIGMOID || p.act == MGLD_ACT_TANH, "conv_f32: activation");
  MGLD_REQUIRE(p.batch == 1 || (p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0), "conv_f32: batched form is LINEAR only");
  const int64_t gx = (p.M + CF_BM - 1) / CF_BM;
  MGLD_REQUIRE(gx <= 0x7fffffff && p.batch <= 65535 && (p.N + CF_BN - 1) / CF_BN <= 65535, "conv_f32: grid");
  hipLaunchKernelGGL(convf32_kernel, dim3((unsigned)gx, (p.N + CF_BN - 1) / CF_BN, p.batch), dim3(256), 0, RS_(stream), p);
  return mgld_check_launch("conv_f32");
}

extern "C" int mgld_instnorm_chunks(int hw) {
  int c = hw / 256;          // >= 32 rows per row group of a block; 16 chunks x 28 images x 2 channel blocks fill the chip at 64^2
  return c < 1 ? 1 : (c > 16 ? 16 : c);
}

extern "C" int mgld_instnorm_f32(const float* x, int ldx, double* part, const float* skip, int lds, float* y, int ldy, int n, int hw,
                                 int C, float eps, int relu, void* stream) {
  MGLD_REQUIRE(x && part && y && n > 0 && hw > 0 && C > 0 && ldx >= C && ldy >= C && (!skip || lds >= C), "instnorm_f32: bad args");
  MGLD_REQUIRE(n <= 65535, "instnorm_f32: too many images");
  const int chunks = mgld_instnorm_chunks(hw);
  const dim3 grid(chunks, (C + 31) / 32, n);
  hipLaunchKernelGGL(instnorm_part_kernel, grid, dim3(256), 0, RS_(stream), x, ldx, part, hw, C, chunks);
  hipLaunchKernelGGL(instnorm_apply_kernel, grid, dim3(256), 0, RS_(stream), x, ldx, (const double*)part, skip, lds, y, ldy, hw, C, chunks,
                     eps, relu);
  return mgld_check_launch("instnorm_f32");
}

extern "C" int mgld_nchw_to_nhwc_f32(const float* x, float* y, int n, int c, int hw, int ld, void* stream) {
  MGLD_REQUIRE(x && y && n > 0 && c > 0 && hw > 0 && ld >= c, "nchw_to_nhwc_f32: bad args");
  hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel, dim3(rgrid((int64_t)n * hw * ld)), dim3(256), 0, RS_(stream), x, y, n, c, hw, ld);
  return mgld_check_launch("nchw_to_nhwc_f32");
}
