// igemm.hip — implicit-GEMM on CDNA4 matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate).
//
//   C[M,N] = epilogue( gather(A)[M,K] * W[N,K]^T )
//
// One kernel serves every dense contraction on the MGLD-VSR hot path (SURVEY.md §2.2 K1,K2,K4,K8,K13):
//   * LINEAR   — nn.Linear / 1x1 conv / batched NT GEMM (attention.py:323-330,51-71; openaimodel.py:515-519)
//   * CONV3X3  — 3x3 conv on NHWC, stride 1/2, asymmetric zero pad, optional nearest-2x upsample folded into the
//                gather (openaimodel.py:176,185,221; model.py:96,114-118)
//   * TCONV3   — Conv3d (3,1,1) over the frame axis (diffusionmodules/util.py:298)
// Layout: activations are token-major (NHWC) fp16 with leading dimension lda; weights are [N][K] fp16 with
// K = taps*Cin contiguous.  Tiles are staged global->registers->LDS (padded rows, conflict-free ds_read_b128),
// double-buffered, one barrier per 32-deep k-step.  The MFMA is issued with the WEIGHT tile as the row operand,
// so each lane ends up holding 4 consecutive output channels per register group -> 8-byte vector stores and a
// lane-local GEGLU pairing.
#include "common.h"

namespace {

constexpr int BK = 32;     // k depth per stage (fp16 elements)
constexpr int LDSS = 40;   // LDS row stride in halves (80 B: 16 rows hit 16 distinct 16-B slots)

struct RowInfo {
  int64_t base;  // LINEAR: m*lda ; CONV: n*Hin*Win (pixel index) ; TCONV: m (row index)
  int iy0, ix0;  // CONV: top-left input coord ; TCONV: iy0 = t
  bool valid;
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_kernel(const MgldIGemm p) {
  constexpr int WAVES_N = BN / WN;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int NA = (BM * 4 + 255) / 256, NB = (BN * 4 + 255) / 256;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");

  __shared__ __attribute__((aligned(16))) f16 smem[2 * (BM + BN) * LDSS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int bm0 = blockIdx.x * BM;
  const int bn0 = blockIdx.y * BN;
  const int bz = blockIdx.z;

  const f16* __restrict__ A = (const f16*)p.A + (int64_t)bz * p.strideA;
  const f16* __restrict__ W = (const f16*)p.W + (int64_t)bz * p.strideW;

  const int M = p.M, N = p.N, K = p.K;
  const int Cin = (p.mode == MGLD_MODE_LINEAR) ? K : p.Cin;

  // ---- per-thread staging assignment -------------------------------------------------------------------
  RowInfo ra[NA];
  int a_ldsoff[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int v = tid + i * 256;
    const int row = v >> 2;
    const int m = bm0 + row;
    ra[i].valid = (v < BM * 4) && (m < M);
    a_ldsoff[i] = row * LDSS + (v & 3) * 8;
    ra[i].base = 0; ra[i].iy0 = 0; ra[i].ix0 = 0;
    if (ra[i].valid) {
      if (p.mode == MGLD_MODE_LINEAR) {
        ra[i].base = (int64_t)m * p.lda;
      } else if (p.mode == MGLD_MODE_CONV3X3) {
        const int hw = p.Hout * p.Wout;
        const int n = m / hw;
        const int r = m - n * hw;
        const int oy = r / p.Wout, ox = r - oy * p.Wout;
        ra[i].base = (int64_t)n * p.Hin * p.Win;
        ra[i].iy0 = oy * p.stride - p.pad_t;
        ra[i].ix0 = ox * p.stride - p.pad_l;
      } else {  // TCONV3
        const int f = m / p.HW;
        ra[i].base = m;
        ra[i].iy0 = f % p.T;
      }
    }
  }
  int b_ldsoff[NB];
  int64_t b_base[NB];
  bool b_valid[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int v = tid + i * 256;
    const int row = v >> 2;
    const int n = bn0 + row;
    b_valid[i] = (v < BN * 4) && (n < N);
    b_ldsoff[i] = (BM + row) * LDSS + (v & 3) * 8;
    b_base[i] = (int64_t)n * p.ldw;
  }
  const int kv8 = (tid & 3) * 8;  // this thread's k offset inside a stage (same for all its vectors)

  // incremental (tap, c) for k = kt*BK + kv8
  int tap = 0, c = kv8;
  while (c >= Cin) { c -= Cin; ++tap; }

  f16x8 regA[NA], regB[NB];
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  auto load_stage = [&](int kt) {
    const int kglob = kt * BK + kv8;
    const bool kval = kglob < K;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      f16x8 v = zero8;
      if (ra[i].valid && kval) {
        if (p.mode == MGLD_MODE_LINEAR) {
          v = *(const f16x8*)(A + ra[i].base + kglob);
        } else if (p.mode == MGLD_MODE_CONV3X3) {
          const int ky = tap / 3, kx = tap - ky * 3;
          int iy = ra[i].iy0 + ky, ix = ra[i].ix0 + kx;
          bool ok;
          if (p.up2) {
            ok = (iy >= 0) && (ix >= 0) && (iy < 2 * p.Hin) && (ix < 2 * p.Win);
            iy >>= 1; ix >>= 1;
          } else {
            ok = (iy >= 0) && (ix >= 0) && (iy < p.Hin) && (ix < p.Win);
          }
          if (ok) v = *(const f16x8*)(A + (ra[i].base + (int64_t)iy * p.Win + ix) * p.lda + c);
        } else {
          const int tt = ra[i].iy0 + tap - 1;
          if (tt >= 0 && tt < p.T) v = *(const f16x8*)(A + (ra[i].base + (int64_t)(tap - 1) * p.HW) * p.lda + c);
        }
      }
      regA[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      f16x8 v = zero8;
      if (b_valid[i] && kval) v = *(const f16x8*)(W + b_base[i] + kglob);
      regB[i] = v;
    }
    // advance (tap, c) to the next stage
    c += BK;
    while (c >= Cin) { c -= Cin; ++tap; }
  };
  auto store_stage = [&](int buf) {
    f16* s = smem + buf * (BM + BN) * LDSS;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (NA * 256 == BM * 4 || tid + i * 256 < BM * 4) *(f16x8*)(s + a_ldsoff[i]) = regA[i];
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (NB * 256 == BN * 4 || tid + i * 256 < BN * 4) *(f16x8*)(s + b_ldsoff[i]) = regB[i];
  };

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  const int nk = (K + BK - 1) / BK;
  load_stage(0);
  store_stage(0);
  __syncthreads();

  const int l31 = lane & 31, lhi = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_stage(kt + 1);
    const f16* sA = smem + buf * (BM + BN) * LDSS;
    const f16* sW = sA + BM * LDSS;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      f16x8 fa[MI], fw[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        fa[mi] = *(const f16x8*)(sA + (wm * WM + mi * 32 + l31) * LDSS + ks * 16 + lhi * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        fw[ni] = *(const f16x8*)(sW + (wn * WN + ni * 32 + l31) * LDSS + ks * 16 + lhi * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    }
    if (kt + 1 < nk) store_stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue ------------------------------------------------------------------------------------------
  // D[i = n_local][j = m_local]: lane holds column j = lane&31 (one output row m), rows i = (r&3)+8*(r>>2)+4*lhi
  const bool geglu = (p.act == MGLD_ACT_GEGLU);
  const int64_t cbase = (int64_t)bz * p.strideC;
  const f16* __restrict__ R = p.R ? (const f16*)p.R + (int64_t)bz * p.strideR : nullptr;
  const int Nout = geglu ? N / 2 : N;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = bm0 + wm * WM + mi * 32 + l31;
    if (m >= M) continue;
    const float bm = p.bias_m ? p.bias_m[m] : 0.f;
    const float* rv = p.rowvec ? p.rowvec + (int64_t)(m / p.rows_per_frame) * p.ld_rowvec : nullptr;
#pragma unroll
    for (int ni = 0; ni < (geglu ? 1 : NI); ++ni) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int nl = rg * 8 + lhi * 4;  // local n within the 32-wide MFMA tile
        float v[4];
        int n0;
        if (geglu) {
          if constexpr (NI == 2) {
            const int npk = bn0 + wn * WN + nl;  // packed row of the value half; gate half is +32
            n0 = (bn0 + wn * WN) / 2 + nl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float x = acc[0][mi][rg * 4 + j], g = acc[1][mi][rg * 4 + j];
              if (p.bias && npk + j + 32 < N) { x += p.bias[npk + j]; g += p.bias[npk + j + 32]; }
              v[j] = x * gelu_f(g);
            }
          } else {
            n0 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = 0.f;
          }
        } else {
          n0 = bn0 + wn * WN + ni * 32 + nl;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x = acc[ni][mi][rg * 4 + j] + bm;
            const int n = n0 + j;
            if (n < N) {
              if (p.bias) x += p.bias[n];
              if (rv) x += rv[n];
            }
            if (p.act == MGLD_ACT_RELU) x = fmaxf(x, 0.f);
            else if (p.act == MGLD_ACT_LRELU02) x = x > 0.f ? x : 0.2f * x;
            else if (p.act == MGLD_ACT_SILU) x = silu_f(x);
            v[j] = x;
          }
        }
        if (n0 >= Nout) continue;
        const bool full = (n0 + 3 < Nout);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] *= p.alpha;
          if (R && n0 + j < Nout) v[j] += p.beta * (float)R[(int64_t)m * p.ldr + n0 + j];
        }
        if (p.out_f32) {
          float* Cf = (float*)p.C + cbase + (int64_t)m * p.ldc + n0;
          if (full && ((p.ldc & 3) == 0) && ((((uintptr_t)Cf) & 15) == 0)) {
            *(f32x4*)Cf = f32x4{v[0], v[1], v[2], v[3]};
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n0 + j < Nout) Cf[j] = v[j];
          }
        } else {
          f16* Ch = (f16*)p.C + cbase + (int64_t)m * p.ldc + n0;
          if (full && ((((uintptr_t)Ch) & 7) == 0)) {
            *(f16x4*)Ch = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n0 + j < Nout) Ch[j] = (f16)v[j];
          }
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const MgldIGemm* p, hipStream_t s) {
  dim3 grid(cdiv(p->M, BM), cdiv(p->N, BN), p->batch > 0 ? p->batch : 1);
  hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN>), grid, dim3(256), 0, s, *p);
  return mgld_check_launch("igemm");
}

}  // namespace

extern "C" int mgld_igemm(const MgldIGemm* p, void* stream) {
  MGLD_REQUIRE(p && p->A && p->W && p->C, "igemm: null pointer");
  MGLD_REQUIRE(p->M > 0 && p->N > 0 && p->K > 0, "igemm: empty problem");
  MGLD_REQUIRE((p->K & 7) == 0 && (p->lda & 7) == 0 && (p->ldw & 7) == 0, "igemm: K, lda, ldw must be multiples of 8");
  MGLD_REQUIRE((((uintptr_t)p->A) & 15) == 0 && (((uintptr_t)p->W) & 15) == 0, "igemm: A/W must be 16-byte aligned");
  MGLD_REQUIRE((p->strideA & 7) == 0 && (p->strideW & 7) == 0, "igemm: batch strides must be multiples of 8");
  if (p->mode != MGLD_MODE_LINEAR) {
    MGLD_REQUIRE(p->Cin > 0 && (p->Cin & 7) == 0, "igemm: Cin must be a positive multiple of 8");
    const int taps = (p->mode == MGLD_MODE_CONV3X3) ? 9 : 3;
    MGLD_REQUIRE(p->K == taps * p->Cin, "igemm: K != taps*Cin");
    if (p->mode == MGLD_MODE_CONV3X3) {
      MGLD_REQUIRE(p->Hin > 0 && p->Win > 0 && p->Hout > 0 && p->Wout > 0, "igemm: conv geometry");
      MGLD_REQUIRE(p->stride == 1 || p->stride == 2, "igemm: stride must be 1 or 2");
      MGLD_REQUIRE(p->M % (p->Hout * p->Wout) == 0, "igemm: M must be frames*Hout*Wout");
    } else {
      MGLD_REQUIRE(p->T > 0 && p->HW > 0 && p->M % (p->T * p->HW) == 0, "igemm: tconv geometry");
    }
  }
  if (p->rowvec) MGLD_REQUIRE(p->rows_per_frame > 0, "igemm: rows_per_frame");
  if (p->act == MGLD_ACT_GEGLU) MGLD_REQUIRE((p->N & 63) == 0, "igemm: GEGLU needs N % 64 == 0");
  hipStream_t s = (hipStream_t)stream;
  switch (mgld_igemm_config(p)) {
    case 128128: return launch_cfg<128, 128, 64, 64>(p, s);
    case 64128: return launch_cfg<64, 128, 32, 64>(p, s);
    case 128032: return launch_cfg<128, 32, 32, 32>(p, s);
    case 128064: return launch_cfg<128, 64, 64, 32>(p, s);
    default: return launch_cfg<64, 64, 32, 32>(p, s);
  }
}

// tile configuration the launcher picks for a problem: BM*1000 + BN
extern "C" int mgld_igemm_config(const MgldIGemm* p) {
  if (!p) return 0;
  const int64_t M = p->M, N = p->N;
  const int batch = p->batch > 0 ? p->batch : 1;
  const int64_t blocks128 = (int64_t)cdiv(M, 128) * cdiv(N, 128) * batch;
  if (p->act == MGLD_ACT_GEGLU) return (blocks128 >= 256 || M <= 64) ? 128128 : 64128;
  if (N <= 32) return 128032;
  if (N <= 64) return 128064;
  if (blocks128 >= 512) return 128128;
  // not enough 128x128 tiles to fill 256 CUs: shrink the tile
  const int64_t blocks64x128 = (int64_t)cdiv(M, 64) * cdiv(N, 128) * batch;
  if (blocks64x128 >= 512) return 64128;
  return 64064;
}
