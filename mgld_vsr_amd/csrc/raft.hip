// raft.hip — the RAFT-specific pieces of the flow estimator (basicsr/archs/raft_arch.py).
// The convolutions / norms of RAFT run on the shared igemm / norm kernels; this file holds what has no counterpart there:
// the correlation pyramid + windowed lookup (raft_arch.py:37-86,519-533), the GRU gate arithmetic (:390-405), the flow
// update and the convex 8x upsampling (:720-731).  All tensors here are small (1/8-resolution grids of the LR frames):
// latency-bound elementwise kernels, one thread per output element.
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
                                                          int W, int r, f16* __restrict__ out, int ldo) {
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
    out[pix * ldo + k] = (f16)sample_plane(lv.p[lvl] + pix * hh * ww, hh, ww, sx, sy);
  }
}

// rhx[:, :Ch] = r * hx[:, :Ch] ; rhx[:, Ch:Ch+Cx] = hx[:, Ch:Ch+Cx]      (cat([r*h, x]), raft_arch.py:395,402)
__global__ __launch_bounds__(256) void gru_rh_kernel(const f16* __restrict__ r, int ldr, const f16* __restrict__ hx, int ldhx,
                                                     f16* __restrict__ rhx, int ldo, int64_t M, int Ch, int Cx) {
  const int C = Ch + Cx;
  const int64_t total = M * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t m = i / C;
    const float v = (float)hx[m * ldhx + c];
    rhx[m * ldo + c] = (f16)(c < Ch ? v * (float)r[m * ldr + c] : v);
  }
}

// h = (1 - z) * h + z * q   in place on the h columns of hx (raft_arch.py:396,403)
__global__ __launch_bounds__(256) void gru_gate_kernel(const f16* __restrict__ z, int ldz, const f16* __restrict__ q, int ldq,
                                                       f16* __restrict__ h, int ldh, int64_t M, int Ch) {
  const int64_t total = M * Ch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % Ch);
    const int64_t m = i / Ch;
    const float zz = (float)z[m * ldz + c], hh = (float)h[m * ldh + c], qq = (float)q[m * ldq + c];
    h[m * ldh + c] = (f16)((1.f - zz) * hh + zz * qq);
  }
}

// coords1 += delta (delta NHWC fp32 [B*H*W, ldd], columns 0,1 = dx,dy); flow = coords1 - coords0 written NCHW fp32 and as two
// fp16 columns of `mot` (the motion-feature tail, cat([out, flow]), raft_arch.py:444) and of `fin` (the flow conv input)
__global__ __launch_bounds__(256) void flow_update_kernel(float* __restrict__ coords1, const float* __restrict__ coords0,
                                                          const float* __restrict__ delta, int ldd, float* __restrict__ flow,
                                                          f16* __restrict__ mot, int ldm, f16* __restrict__ fin, int ldf, int B,
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
      if (mot) mot[i * ldm + c] = (f16)f;
      if (fin) fin[i * ldf + c] = (f16)f;
    }
  }
}

// upsample_flow (raft_arch.py:720-731): out[b,c,8y+u,8x+v] = sum_k softmax_k(mask[b,(k,u,v),y,x]) * 8*flow[b,c,y+dy_k,x+dx_k]
// with k = 3*(dy+1)+(dx+1) (F.unfold order, zero padding); mask NHWC fp16 [B*H*W, ldm] with channel k*64 + u*8 + v.
__global__ __launch_bounds__(256) void convex_up_kernel(const float* __restrict__ flow, const f16* __restrict__ mask, int ldm,
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
    for (int k = 0; k < 9; ++k) { m[k] = (float)mask[pix * ldm + k * 64 + uv]; mx = fmaxf(mx, m[k]); }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { m[k] = __expf(m[k] - mx); s += m[k]; }
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

// y = relu(a + b)   fp16 [M, C] views (ResidualBlock tail, raft_arch.py:138)
__global__ __launch_bounds__(256) void add_relu_kernel(const f16* __restrict__ a, int lda, const f16* __restrict__ b, int ldb,
                                                       f16* __restrict__ y, int ldy, int64_t M, int C) {
  const int64_t total = M * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t m = i / C;
    y[m * ldy + c] = (f16)fmaxf((float)a[m * lda + c] + (float)b[m * ldb + c], 0.f);
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
                                int H, int W, int radius, void* out, int ldo, void* stream) {
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
                     W, radius, (f16*)out, ldo);
  return mgld_check_launch("corr_lookup");
}

extern "C" int mgld_gru_rh(const void* r, int ldr, const void* hx, int ldhx, void* rhx, int ldo, int64_t M, int Ch, int Cx,
                           void* stream) {
  MGLD_REQUIRE(r && hx && rhx && M > 0 && Ch > 0 && Cx >= 0, "gru_rh: bad args");
  hipLaunchKernelGGL(gru_rh_kernel, dim3(rgrid(M * (Ch + Cx))), dim3(256), 0, RS_(stream), (const f16*)r, ldr, (const f16*)hx, ldhx,
                     (f16*)rhx, ldo, M, Ch, Cx);
  return mgld_check_launch("gru_rh");
}

extern "C" int mgld_gru_gate(const void* z, int ldz, const void* q, int ldq, void* h, int ldh, int64_t M, int Ch, void* stream) {
  MGLD_REQUIRE(z && q && h && M > 0 && Ch > 0, "gru_gate: bad args");
  hipLaunchKernelGGL(gru_gate_kernel, dim3(rgrid(M * Ch)), dim3(256), 0, RS_(stream), (const f16*)z, ldz, (const f16*)q, ldq,
                     (f16*)h, ldh, M, Ch);
  return mgld_check_launch("gru_gate");
}

extern "C" int mgld_flow_update(float* coords1, const float* coords0, const float* delta, int ldd, float* flow, void* mot, int ldm,
                                void* fin, int ldf, int B, int HW, void* stream) {
  MGLD_REQUIRE(coords1 && coords0 && flow && B > 0 && HW > 0, "flow_update: bad args");
  hipLaunchKernelGGL(flow_update_kernel, dim3(rgrid((int64_t)B * HW)), dim3(256), 0, RS_(stream), coords1, coords0, delta, ldd, flow,
                     (f16*)mot, ldm, (f16*)fin, ldf, B, HW);
  return mgld_check_launch("flow_update");
}

extern "C" int mgld_convex_upsample(const float* flow, const void* mask, int ldm, float* out, int B, int H, int W, void* stream) {
  MGLD_REQUIRE(flow && mask && out && B > 0 && H > 0 && W > 0 && ldm >= 576, "convex_upsample: bad args");
  hipLaunchKernelGGL(convex_up_kernel, dim3(rgrid((int64_t)B * H * W * 64)), dim3(256), 0, RS_(stream), flow, (const f16*)mask, ldm,
                     out, B, H, W);
  return mgld_check_launch("convex_upsample");
}

extern "C" int mgld_add_relu(const void* a, int lda, const void* b, int ldb, void* y, int ldy, int64_t M, int C, void* stream) {
  MGLD_REQUIRE(a && b && y && M > 0 && C > 0, "add_relu: bad args");
  hipLaunchKernelGGL(add_relu_kernel, dim3(rgrid(M * C)), dim3(256), 0, RS_(stream), (const f16*)a, lda, (const f16*)b, ldb, (f16*)y,
                     ldy, M, C);
  return mgld_check_launch("add_relu");
}
