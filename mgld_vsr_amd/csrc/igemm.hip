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
#include "igemm_common.h"

namespace mgld_ig {
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

}  // namespace mgld_ig

namespace {
using namespace mgld_ig;

// TWO: the instantiation that runs the weight-residual pass (MgldIGemm.W2).  A template parameter, not a runtime flag: carrying the second
// pass as runtime state cost the one-pass kernels 8-44 VGPRs and 5-30 % (same-box A/B against the round-2 library, profiles/r03_w2_regression.txt).
template <int MODE, bool FAST, int BM, int BN, int WM, int WN, int NST, bool TWO = false>
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
  if ((order & 0xff) == 1) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, j = lin >> 3;
    tile_m = xcd + 8 * (j / (int)gridDim.y);
    tile_n = j % (int)gridDim.y;
  } else if ((order & 0xff) == 2) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, j = lin >> 3;
    tile_n = xcd + 8 * (j / (int)gridDim.x);
    tile_m = j % (int)gridDim.x;
  }
  const int bm0 = tile_m * BM;
  const int bn0 = tile_n * BN;
  // TCONV row order (order bits 8..: lg, set by the launcher): a tile takes 2^lg consecutive PIXELS of every frame of one clip instead of BM
  // consecutive rows of one frame.  The three temporal taps of the tile then read the same BM rows (shifted by one frame): each frame
  // row leaves HBM once per column tile and the other two taps hit this XCD's L2, instead of three passes 2*HW*C bytes apart (the VAE's
  // 256^2 / 512^2 levels: 67 MB per frame).  Same K order per output element: bit-identical to the consecutive order.
  RowMapFrames trows{bm0, 31, 0x7fffffff, 0, p.M};
  if constexpr (MODE == MGLD_MODE_TCONV3) {
    const int lg = order >> 8;
    if (lg) {
      const int tpc = p.HW >> lg;              // tiles per clip
      const int clip = tile_m / tpc;
      trows = RowMapFrames{clip * p.T * p.HW + ((tile_m - clip * tpc) << lg), lg, (1 << lg) - 1, p.HW, p.M};
    }
  }
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
  constexpr bool two = TWO;
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
      const int m = (MODE == MGLD_MODE_TCONV3) ? trows(row) : (bm0 + row < M ? bm0 + row : -1);
      const bool valid = m >= 0;
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
      const int m = (MODE == MGLD_MODE_TCONV3) ? trows(row) : (bm0 + row < M ? bm0 + row : -1);
      ra[j].valid = m >= 0;
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

  if constexpr (MODE == MGLD_MODE_TCONV3) tile_epilogue<BM, BN, WM, WN>(p, ws, splitk, kz, bz, trows, bn0, wm, wn, wave, lane, acc, smem);
  else tile_epilogue<BM, BN, WM, WN>(p, ws, splitk, kz, bz, RowMapLinear{bm0, M}, bn0, wm, wn, wave, lane, acc, smem);
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
      if (p.r_f32 && p.R) x += p.beta * ((const float*)p.R)[(int64_t)m * p.ldr + n];
      else if (R) {
        x += p.beta * (float)R[(int64_t)m * p.ldr + n];
        if (p.Rlo) x += p.beta * MGLD_LO_SCALE * (float)((const f16*)p.Rlo)[(int64_t)m * p.ldr + n];
      }
      if (p.out_f32) ((float*)p.C)[(int64_t)m * p.ldc + n] = x;
      else {
        const f16 hi = (f16)x;
        ((f16*)p.C)[(int64_t)m * p.ldc + n] = hi;
        if (p.Clo) ((f16*)p.Clo)[(int64_t)m * p.ldc + n] = lo_plane(x, hi);
      }
    }
  }
}

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

// TCONV row order (see the kernel): log2 of the pixels per frame a tile takes, 0 = BM consecutive rows.  Whole clips only (no frame
// shard offset), T a power of two dividing BM, whole tiles per frame — and frames of >= 24 MB, where the three taps of a row tile are too
// far apart for the L2s.  Measured (tools/tconv_bench.py, profiles/r03_tconv_rows.txt): 512^2 x 128 821 -> 766 us, 1024^2 x 128 (T = 4)
// 1658 -> 1524, 256^2 x 256 with the residual pass 731 -> 688; 128^2 x 512 (17 MB frames) 339 -> 372: slower, hence the threshold.
// The launch stays far from the HBM roofline either way (1.4 TB/s): what bounds it is the bytes a CU keeps in flight — two blocks x one
// 32 KB stage, half of it weights — not the tap re-reads.  env MGLD_TCONV_ROWS=0: consecutive rows; =2: interleave regardless of the
// frame size (tests); tune 15: consecutive rows for this launch.
inline int tconv_rows_lg(const MgldIGemm* p, int BM) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("MGLD_TCONV_ROWS"); on = e ? atoi(e) : 1; }
  const int T = p->T;
  if (!on || p->tune == 15 || p->t_off != 0 || T < 2 || (T & (T - 1)) || BM % T || BM / T < 8) return 0;
  if (on != 2 && p->tune != 14 && (int64_t)p->HW * p->Cin * 2 < (24 << 20)) return 0;
  const int ppt = BM / T;
  if (p->HW % ppt || p->M % (T * p->HW)) return 0;
  int lg = 0;
  while ((1 << lg) < ppt) ++lg;
  return lg;
}

template <int MODE, bool FAST, int BM, int BN, int WM, int WN, int NST, bool TWO = false>
void launch_fast(const MgldIGemm* p, hipStream_t s, int splits, int kchunk) {
  constexpr int LDS = (NST == 0 ? 2 : NST) * (BM + BN) * ROWB;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)igemm_kernel<MODE, FAST, BM, BN, WM, WN, NST, TWO>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  constexpr int THREADS = 64 * (BM / WM) * (BN / WN);
  const int gz = splits > 1 ? splits : (p->batch > 0 ? p->batch : 1);
  dim3 grid(cdiv(p->M, BM), cdiv(p->N, BN), gz);
  int order = tile_order(p, (int)grid.x, (int)grid.y);
  if constexpr (MODE == MGLD_MODE_TCONV3) order |= tconv_rows_lg(p, BM) << 8;
  hipLaunchKernelGGL((igemm_kernel<MODE, FAST, BM, BN, WM, WN, NST, TWO>), grid, dim3(THREADS), LDS, s, *p,
                     splits > 1 ? g_ws : nullptr, kchunk, order);
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
  if (p->W2) {      // weight-residual pass: the two-deep ring only
    if (fast_ok(p)) launch_fast<MODE, true, BM, BN, WM, WN, 2, true>(p, s, splits, kchunk);
    else launch_fast<MODE, false, BM, BN, WM, WN, 2, true>(p, s, splits, kchunk);
    return;
  }
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
  if (p->mode == MGLD_MODE_LINEAR && fast_ok(p) && splits <= 1 && !p->W2) {
    static int rs = -1;   // env MGLD_IGEMM_RS = 1: register-staged tiles on the LINEAR fast path (p->tune = 9 selects them per launch)
    if (rs < 0) { const char* e = getenv("MGLD_IGEMM_RS"); rs = e ? atoi(e) : 0; }
    if ((rs && p->tune == 0) || p->tune == 9) {
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
  if (splits > 1) launch_splitk_reduce(p, s, splits);
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
  if (N <= 32) {
    // skinny outputs over a deep K (the UNet's last conv: 320 -> 4 at 64x64, K = 2880): M / 128 blocks of one 45-stage chain each leave the
    // CUs waiting on a DMA round trip per stage (72 us at 10 TF/s); a K split over grid.z runs the chains side by side, and the reduce pass
    // over an M x 4 output costs nothing.  env MGLD_IGEMM_SKINNY_SPLIT=0: no split (A/B).
    static int skinny = -1;
    if (skinny < 0) { const char* e = getenv("MGLD_IGEMM_SKINNY_SPLIT"); skinny = e ? atoi(e) : 1; }
    const int64_t t = (int64_t)cdiv(M, 128) * batch;
    *cfg = 128032;
    if (skinny && batch == 1 && K >= 1536 && t < 2 * num_cus() && g_ws != nullptr) {
      int sp = (int)((4 * num_cus() + t - 1) / t);
      if (sp > (int)(K / 512)) sp = (int)(K / 512);
      if (sp > 8) sp = 8;
      if (sp >= 2 && (size_t)sp * M * N * sizeof(float) <= g_ws_bytes) {
        int kc = (int)((K + sp - 1) / sp);
        kc = (kc + BK - 1) / BK * BK;
        sp = (int)((K + kc - 1) / kc);
        if (sp >= 2) { *splits = sp; *kchunk = kc; }
      }
    }
    return;
  }
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

}  // namespace

namespace mgld_ig {
void launch_splitk_reduce(const MgldIGemm* p, hipStream_t s, int splits) {
  const int64_t total = (int64_t)p->M * ((p->N + 3) >> 2);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, *p, g_ws, splits);
}
}  // namespace mgld_ig
using namespace mgld_ig;

extern "C" int mgld_set_workspace(void* ptr, int64_t bytes) {
  g_ws = (float*)ptr;
  g_ws_bytes = ptr ? (size_t)bytes : 0;
  return MGLD_OK;
}

// tile configuration the launcher picks for a problem: BM*1000 + BN (+ splits*1000000 when split along K)
extern "C" int mgld_igemm_config(const MgldIGemm* p) {
  if (!p) return 0;
  int cfg, splits, kchunk;
  if (ppgemm_plan(p, &cfg)) return 500000 + cfg;                                                            // ping-pong LINEAR
  if (conv3r_plan(p, &cfg, &splits)) return 600000 + cfg + (splits > 1 ? splits * 1000000 : 0);          // ping-pong patch conv
  { int lg_; if (pptconv_plan(p, &cfg, &lg_)) return 700000 + cfg; }                                        // ping-pong temporal conv
  if (conv3q_plan(p, &cfg, &splits, &kchunk)) return 400000 + cfg + (splits > 1 ? splits * 1000000 : 0);   // 2-D-tile patch conv
  if (conv3p_plan(p, &cfg, &splits, &kchunk)) return 300000 + cfg + (splits > 1 ? splits * 1000000 : 0);   // raster patch conv
  choose(p, &cfg, &splits, &kchunk);
  return cfg + (splits > 1 ? splits * 1000000 : 0);
}

// statistics output (MgldIGemm.gn_part): tiles per frame when the kernel picked for this problem writes it, else 0
extern "C" int mgld_igemm_gn_chunks(const MgldIGemm* p) {
  if (!p) return 0;
  int cfg;
  if (ppgemm_plan(p, &cfg)) return 0;
  int sp;
  if (conv3r_plan(p, &cfg, &sp)) return ((p->N & 7) || sp > 1) ? 0 : conv3r_gn_chunks(p, cfg);
  return 0;
}

// row statistics (MgldIGemm.row_part) / folded LayerNorm (ln_part): column tiles when the ping-pong LINEAR kernel takes the problem, else 0
extern "C" int mgld_igemm_row_chunks(const MgldIGemm* p) {
  if (!p) return 0;
  int cfg;
  return ppgemm_plan(p, &cfg) ? ppgemm_row_chunks(p, cfg) : 0;
}

// name of the kernel template instantiation the launcher runs for this problem, spelled as rocprofv3 prints it
extern "C" int mgld_igemm_kernel_name(const MgldIGemm* p, char* buf, int buflen) {
  MGLD_REQUIRE(p && buf && buflen > 0, "igemm_kernel_name: null");
  int cfg, splits, kchunk;
  if (ppgemm_plan(p, &cfg)) {
    ppgemm_kernel_name(p, cfg, buf, buflen);
    return 1;
  }
  if (conv3r_plan(p, &cfg, &splits)) {
    conv3r_kernel_name(p, cfg, buf, buflen);
    return splits;
  }
  { int lg_; if (pptconv_plan(p, &cfg, &lg_)) { pptconv_kernel_name(p, cfg, buf, buflen); return 1; } }
  if (conv3q_plan(p, &cfg, &splits, &kchunk)) {
    conv3q_kernel_name(p, cfg, buf, buflen);
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
  snprintf(buf, buflen, p->W2 ? "igemm_kernel<%d, %s, %d, %d, %d, %d, 2, true>" : "igemm_kernel<%d, %s, %d, %d, %d, %d, 2, false>", p->mode,
           fast_ok(p) ? "true" : "false", bm, bn, wm, wn);
  return (cfg == 128128 || cfg == 128032) ? splits : 1;
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
  if (p->r_f32) MGLD_REQUIRE(p->R && p->out_f32 && p->batch <= 1 && p->act != MGLD_ACT_GEGLU && ((((uintptr_t)p->R) & 3) == 0),
                             "igemm: an fp32 residual (r_f32) goes with an fp32 output, batch 1");
  if (p->act == MGLD_ACT_GEGLU) MGLD_REQUIRE((p->N & 63) == 0, "igemm: GEGLU needs N % 64 == 0");
  if (p->Rlo) MGLD_REQUIRE(p->R && !p->r_f32 && p->batch <= 1 && ((((uintptr_t)p->Rlo) & 15) == 0), "igemm: Rlo goes with an fp16 residual R (same ldr), batch 1, 16-byte aligned");
  if (p->Clo) MGLD_REQUIRE(!p->out_f32 && p->batch <= 1 && p->act != MGLD_ACT_GEGLU && ((((uintptr_t)p->Clo) & 15) == 0), "igemm: Clo goes with an fp16 output (same ldc), batch 1, no GEGLU, 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  int cfg, splits, kchunk;
  if (p->gn_part) MGLD_REQUIRE(mgld_igemm_gn_chunks(p) > 0 && ((((uintptr_t)p->gn_part) & 3) == 0), "igemm: gn_part set, but the kernel picked for this problem does not write statistics (mgld_igemm_gn_chunks)");
  if (ppgemm_plan(p, &cfg)) return dispatch_ppgemm(p, s, cfg);
  MGLD_REQUIRE(!p->row_part && !p->ln_part, "igemm: row_part / ln_part set, but the kernel picked for this problem takes neither (mgld_igemm_row_chunks)");
  if (conv3r_plan(p, &cfg, &splits)) return dispatch_conv3r(p, s, cfg, splits);
  { int lg_; if (pptconv_plan(p, &cfg, &lg_)) return dispatch_pptconv(p, s, cfg, lg_); }
  if (conv3q_plan(p, &cfg, &splits, &kchunk)) return dispatch_conv3q(p, s, cfg, splits, kchunk);
  if (conv3p_plan(p, &cfg, &splits, &kchunk)) {
    MGLD_REQUIRE(!p->W2, "igemm: the raster patch kernel (MGLD_CONV3Q=0) does not take W2");
    return dispatch_conv3p(p, s, cfg, splits, kchunk);
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
      if (p->W2) launch_fast<MGLD_MODE_LINEAR, true, 128, 320, 32, 160, 2, true>(p, s, 1, kchunk);
      else launch_fast<MGLD_MODE_LINEAR, true, 128, 320, 32, 160, 2>(p, s, 1, kchunk);
      return mgld_check_launch("igemm");
    case 64128: return launch_cfg<64, 128, 32, 64>(p, s, 1, kchunk);
    case 128032: return launch_cfg<128, 32, 32, 32>(p, s, splits, kchunk);
    case 128064: return launch_cfg<128, 64, 64, 32>(p, s, 1, kchunk);
    default: return launch_cfg<64, 64, 32, 32>(p, s, 1, kchunk);
  }
}
