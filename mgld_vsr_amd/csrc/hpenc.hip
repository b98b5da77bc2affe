// hpenc.hip — the high-precision first-stage encoder's glue kernels (round 5; ldm/modules/diffusionmodules/model.py:473-572 Encoder,
// called from ddpm.py:3906-3943 encode_first_stage).
//
// Why: the first-stage latent conditions EVERY one of the 50 sampling steps (struct_cond) — an error in it is a bias, it does not
// average out over the steps.  On smooth (realistic) frames the fp16 encoder's latent sat 1.07e-3 from the reference's and the
// sampled x_0 2.4e-3, against 9.0e-4 with the reference's latent handed to the same sampler (tools/x0_probe.py, g_work_c2s_S50):
// 85 % of the error variance of x_0 came from 2 % of the segment's arithmetic.  All three fp16 rounding sources of the encoder weigh
// alike (operands entering the contractions, weights, stored outputs), so it runs with fp32 activations between the kernels and
// SPLIT-fp16 operands inside the contractions:
//     a = ah + al (fp16 + fp16),  w = wh + wl:   a w = ah wh + al wh + ah wl + O(2^-22)
// as ONE pass of the existing fp16 MFMA kernels over a channel axis three times as long:
//     activations  [ ah | 16 al | ah / 256 ]        (written by mgld_hp_gn_split: GroupNorm + SiLU in fp32, then the split)
//     weights      [ wh | wh / 16 | 256 wl ]        (packed on the host, engine.pack_hp)
// (the power-of-two factors keep al and wl in fp16's normal range for operands above ~1e-2 / ~1e-3; below that the CORRECTION of an
// already negligible product degrades gracefully).  Products of fp16 pairs are exact in the fp32 accumulators, so the contraction is
// fp32-accurate at 3x the fp16 MFMA work (27 of 435 TFLOP per segment) instead of 16x on the f32-input MFMA.  Outputs and the residual
// stream stay fp32 (MgldIGemm.out_f32 / r_f32).  The mid attention block (34 GFLOP per frame) and the 3- / 8-channel end convolutions
// run on mgld_conv_f32 (raft.hip) directly.
#include "common.h"

namespace {

constexpr int HP_MAX_GROUPS = 64;

__device__ __forceinline__ int hp_rows_per_chunk(int rows, int chunks) { return (rows + chunks - 1) / chunks; }

// per (frame, row chunk, group) fp64 (sum, sumsq) of an fp32 NHWC tensor; deterministic (fixed-order LDS reduction, no atomics)
// block = RPB rows x TPR threads, a thread owns 4 consecutive channels and keeps one fp64 pair per channel (any group width)
__global__ __launch_bounds__(256) void hp_gn_stats_kernel(const float* __restrict__ x, int ldx, int rows, int C, int groups,
                                                          double* __restrict__ gsums) {
  __shared__ double red[256][4][2];
  const int TPR = C >> 2, RPB = 256 / TPR;
  const int tid = threadIdx.x, rr = tid / TPR, tc = tid - rr * TPR;
  const int chunk = blockIdx.x, chunks = gridDim.x, frame = blockIdx.y;
  const int rpc = hp_rows_per_chunk(rows, chunks);
  const int r0 = chunk * rpc, r1 = min(rows, r0 + rpc);
  double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
  if (rr < RPB) {
    const float* xp = x + ((int64_t)frame * rows) * ldx + tc * 4;
    for (int r = r0 + rr; r < r1; r += RPB) {
      const f32x4 v = *(const f32x4*)(xp + (int64_t)r * ldx);
#pragma unroll
      for (int j = 0; j < 4; ++j) { const double d = (double)v[j]; s[j] += d; q[j] += d * d; }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[tid][j][0] = s[j]; red[tid][j][1] = q[j]; }
  __syncthreads();
  if (tid < groups) {
    const int cg = C / groups;
    double ss = 0.0, qq = 0.0;
    for (int r = 0; r < RPB; ++r)
      for (int c = tid * cg; c < (tid + 1) * cg; ++c) { ss += red[r * TPR + (c >> 2)][c & 3][0]; qq += red[r * TPR + (c >> 2)][c & 3][1]; }
    double* o = gsums + (((int64_t)frame * chunks + chunk) * groups + tid) * 2;
    o[0] = ss;
    o[1] = qq;
  }
}

// y = [silu]( (x - mean_g) * rstd_g * gamma_c + beta_c )  (gsums == nullptr: y = x), written as the split-fp16 operand
// [ yh | 16 yl | yh / 256 ] (MODE 0, fp16 [rows, 3C]) or as fp32 (MODE 1)
template <int MODE>
__global__ __launch_bounds__(256) void hp_gn_split_kernel(const float* __restrict__ x, int ldx, const double* __restrict__ gsums, int schunks,
                                                          float eps, const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
                                                          void* __restrict__ out, int ldo, int rows, int C, int groups) {
  __shared__ float st[HP_MAX_GROUPS][2];
  const int TPR = C >> 2, RPB = 256 / TPR;
  const int tid = threadIdx.x, rr = tid / TPR, tc = tid - rr * TPR;
  const int chunk = blockIdx.x, chunks = gridDim.x, frame = blockIdx.y;
  if (gsums) {
    if (tid < groups) {
      double s = 0.0, q = 0.0;
      for (int k = 0; k < schunks; ++k) {
        const double* o = gsums + (((int64_t)frame * schunks + k) * groups + tid) * 2;
        s += o[0];
        q += o[1];
      }
      const double n = (double)rows * (C / groups);
      const double mean = s / n, var = fmax(q / n - mean * mean, 0.0);
      st[tid][0] = (float)mean;
      st[tid][1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
  }
  if (rr >= RPB) return;
  const int c = tc * 4;
  float a[4], b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { a[j] = 1.f; b[j] = 0.f; }
  if (gsums) {
    const int cg = C / groups;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = (c + j) / cg;
      a[j] = st[g][1] * gamma[c + j];
      b[j] = beta[c + j] - st[g][0] * a[j];
    }
  }
  const int rpc = hp_rows_per_chunk(rows, chunks);
  const int r0 = chunk * rpc, r1 = min(rows, r0 + rpc);
  for (int r = r0 + rr; r < r1; r += RPB) {
    const int64_t row = (int64_t)frame * rows + r;
    const f32x4 v = *(const f32x4*)(x + row * ldx + c);
    float y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = fmaf(a[j], v[j], b[j]);
      if (silu) t = t / (1.f + expf(-t));
      y[j] = t;
    }
    if constexpr (MODE == 1) {
      *(f32x4*)((float*)out + row * ldo + c) = f32x4{y[0], y[1], y[2], y[3]};
    } else {
      f16x4 h, l, h3;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h[j] = (f16)y[j];
        l[j] = (f16)((y[j] - (float)h[j]) * 16.f);
        h3[j] = (f16)((float)h[j] * (1.f / 256.f));
      }
      f16* o = (f16*)out + row * ldo + c;
      *(f16x4*)o = h;
      *(f16x4*)(o + C) = l;
      *(f16x4*)(o + 2 * C) = h3;
    }
  }
}

// in-place fp32 row softmax (the mid attention block, model.py:226-229: softmax over the keys); one block per row
__global__ __launch_bounds__(256) void hp_softmax_rows_kernel(float* __restrict__ S, int cols, int ld) {
  __shared__ float red[4];
  float* row = S + (int64_t)blockIdx.x * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = -3.0e38f;
  for (int c = tid; c < cols; c += 256) mx = fmaxf(mx, row[c]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int c = tid; c < cols; c += 256) {
    const float e = expf(row[c] - mx);
    row[c] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float inv = 1.f / ((red[0] + red[1]) + (red[2] + red[3]));
  for (int c = tid; c < cols; c += 256) row[c] *= inv;
}

}  // namespace

extern "C" int mgld_hp_chunks(int rows) {
  int c = rows / 512;                  // >= 512 rows per block pass-set; 64 chunks x 8 frames fill the chip at 128^2 and above
  return c < 1 ? 1 : (c > 256 ? 256 : c);
}

static int hp_check(int C, int groups, int ldx) {
  MGLD_REQUIRE(C > 0 && (C & 3) == 0 && C <= 1024 && (ldx & 3) == 0 && ldx >= C, "hp: C % 4 == 0, C <= 1024, ldx % 4 == 0");
  MGLD_REQUIRE(groups > 0 && groups <= HP_MAX_GROUPS && C % groups == 0, "hp: C % groups");
  return MGLD_OK;
}

extern "C" int mgld_hp_gn_stats(const float* x, int ldx, int frames, int rows, int C, int groups, double* gsums, void* stream) {
  MGLD_REQUIRE(x && gsums && frames > 0 && frames <= 65535 && rows > 0 && (((uintptr_t)x) & 15) == 0, "hp_gn_stats: bad args");
  if (int rc = hp_check(C, groups, ldx)) return rc;
  hipLaunchKernelGGL(hp_gn_stats_kernel, dim3(mgld_hp_chunks(rows), frames), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, C, groups, gsums);
  return mgld_check_launch("hp_gn_stats");
}

extern "C" int mgld_hp_gn_split(const float* x, int ldx, const double* gsums, float eps, const float* gamma, const float* beta, int silu,
                                void* out, int ldo, int out_f32, int frames, int rows, int C, int groups, void* stream) {
  MGLD_REQUIRE(x && out && frames > 0 && frames <= 65535 && rows > 0 && (((uintptr_t)x) & 15) == 0, "hp_gn_split: bad args");
  MGLD_REQUIRE(!gsums || (gamma && beta), "hp_gn_split: statistics without affine parameters");
  if (int rc = hp_check(C, gsums ? groups : 1, ldx)) return rc;
  const int chunks = mgld_hp_chunks(rows);
  const dim3 grid(chunks, frames);
  if (out_f32) {
    MGLD_REQUIRE(ldo >= C && (ldo & 3) == 0 && (((uintptr_t)out) & 15) == 0, "hp_gn_split: fp32 output layout");
    hipLaunchKernelGGL((hp_gn_split_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, gsums, chunks, eps, gamma, beta, silu, out, ldo,
                       rows, C, gsums ? groups : 1);
  } else {
    MGLD_REQUIRE(ldo >= 3 * C && (ldo & 3) == 0 && (((uintptr_t)out) & 7) == 0, "hp_gn_split: split output needs ld >= 3 C");
    hipLaunchKernelGGL((hp_gn_split_kernel<0>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, gsums, chunks, eps, gamma, beta, silu, out, ldo,
                       rows, C, gsums ? groups : 1);
  }
  return mgld_check_launch("hp_gn_split");
}

extern "C" int mgld_hp_softmax_rows(float* S, int64_t rows, int cols, int ld, void* stream) {
  MGLD_REQUIRE(S && rows > 0 && rows <= 0x7fffffff && cols > 0 && ld >= cols, "hp_softmax_rows: bad args");
  hipLaunchKernelGGL(hp_softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, S, cols, ld);
  return mgld_check_launch("hp_softmax_rows");
}
