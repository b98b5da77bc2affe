// pptconv.hip — the temporal Conv3d (3,1,1) of SpatialTemporalConv (diffusionmodules/util.py:291-310; model.py:966,995) at the video
// decoder's large levels, on the ping-pong structure of pp_common.h.
//
// The op is HBM-bound (K = 3 C with C = 128 .. 512: 128-256 FLOP per byte moved once) and the implicit-GEMM form of igemm.hip sat at 0.17
// of the HBM peak: two blocks per CU with one 32 KiB stage in flight each, and three passes over rows 2 HW C bytes apart.  Here a tile is
// P = 256 / T pixels of EVERY frame of one clip (T = 8: 32 pixels, T = 4: 64), so the three temporal taps of a tile read the same 256 rows
// shifted by P: every activation row leaves HBM once and the other two taps hit this XCD's L2; the ring of 64-deep stages is filled by
// `buffer_load ... lds` with counted waits (48-64 KiB in flight per CU), frames before / after the clip are lanes pushed past the
// descriptor's range (the DMA writes zeros: the Conv3d's zero padding), and the blend out = a (conv + b) + (1 - a) x is the epilogue's
// alpha / beta / residual.  TWO: the weight-residual pass (MgldIGemm.W2) as a second walk over K.
// Covered: K order (tap, Cin) (engine.pack_tconv3), Cin % 64 == 0, T in {4, 8}, whole clips (t_off == 0), HW % P == 0, N % BN == 0.
#include "pp_common.h"

namespace {
using namespace mgld_ig;

template <int BM, int BN, int WGM, int WGN, int NST, bool TWO>
__global__ __launch_bounds__(512) void pptconv_kernel(const MgldIGemm p, const int lgP, const int tiles_n) {
  constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 16, NI = WN / 16;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int NP = (BM + BN) / 8;
  constexpr int CLO = NP / 8, CHI = (NP + 7) / 8;
  constexpr int JA = BM / 64;
  constexpr int H = (NST == 2) ? CHI : (CHI + 1) / 2;
  static_assert(WGM * WGN == 8 && BM == 256 && WM % 16 == 0 && WN % 16 == 0, "tile shape");
  static_assert(NST >= 2 && NST <= 4 && NST * STAGE <= 160 * 1024, "LDS ring");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave / WGN, wn = wave % WGN;
  const bool hi = wave < (NP & 7);

  const int P = 1 << lgP, T = BM >> lgP, HW = p.HW, C = p.Cin;
  const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
  const int tpc = HW >> lgP;                         // tiles per clip
  const int clip = tile_m / tpc, p0 = (tile_m - clip * tpc) << lgP;
  const int bn0 = tile_n * BN;
  const int64_t row0 = (int64_t)clip * T * HW + p0;  // global row of (frame 0, pixel p0) of this tile's clip
  const uint32_t lda2 = (uint32_t)p.lda * 2u, ldw2 = (uint32_t)p.ldw * 2u, frame2 = (uint32_t)HW * lda2;
  // activation descriptor: based ONE FRAME BEFORE the clip, so that tap dt adds dt frames (never a negative offset); the lanes of frame
  // -1 / frame T are never dereferenced: their offset is pushed past num_records and the DMA writes zeros
  const auto rsA = pp_make_rsrc((const f16*)p.A + (row0 - HW) * p.lda, 0x80000000u);
  const auto rsW = pp_make_rsrc((const f16*)p.W + (int64_t)bn0 * p.ldw, 0xffffffffu);
  const auto rsW2 = pp_make_rsrc((const f16*)(TWO ? p.W2 : p.W) + (int64_t)bn0 * p.ldw, 0xffffffffu);
  const uint32_t rl = wave * 8 + (lane >> 3);        // row inside a 64-row slot
  const uint32_t clog = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
  const int fpp = 64 >> lgP;                         // frames per slot (2 at T = 8, 1 at T = 4)
  const uint32_t fl = rl >> lgP;
  const uint32_t voffA = (fl * HW + (rl & (P - 1))) * lda2 + clog * 16;
  const uint32_t voffA0 = fl == 0 ? 0x80000000u : voffA;                    // slot 0 under tap 0: frame -1
  const uint32_t voffA2 = (int)fl == fpp - 1 ? 0x80000000u : voffA;         // last slot under tap 2: frame T
  const uint32_t voffW = rl * ldw2 + clog * 16;
  const int nk = p.K >> 6, nkt = TWO ? 2 * nk : nk;

  int is = 0, itap = 0, ic0 = 0;                      // issue cursor: stage, its tap and channel offset (K order (tap, Cin))
  auto issue = [&](const int j0, const int j1, const int buf) __attribute__((always_inline)) {
    char* dst = smem + buf * STAGE + wave * 1024;
    const uint32_t ka = (uint32_t)itap * frame2 + (uint32_t)ic0 * 2u;
    const int ik = (TWO && is >= nk) ? is - nk : is;
    const uint32_t kw = (uint32_t)ik * 128u;
    const bool lo = TWO && is < nk;
#pragma unroll
    for (int j = 0; j < CHI; ++j) {
      if (j < j0 || j >= j1) continue;
      if (j >= CLO && !hi) continue;
      if (j < JA) {
        const uint32_t va = (j == 0 && itap == 0) ? voffA0 : ((j == JA - 1 && itap == 2) ? voffA2 : voffA);
        pp_dma16(rsA, dst + j * 8192, va, (uint32_t)(j * fpp) * frame2 + ka);
      } else if (lo) pp_dma16(rsW2, dst + j * 8192, voffW, (uint32_t)(j - JA) * 64u * ldw2 + kw);
      else pp_dma16(rsW, dst + j * 8192, voffW, (uint32_t)(j - JA) * 64u * ldw2 + kw);
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
    ++is;
    ic0 += 64;
    if (ic0 == C) { ic0 = 0; ++itap; }
    if (TWO && is == nk) { itap = 0; ic0 = 0; }
  };
  auto wait_stages = [&](const int n) {
    if (n <= 0) { pp_wait_vm<0>(); return; }
    if (hi) { if (n == 1) pp_wait_vm<CHI>(); else pp_wait_vm<2 * CHI>(); }
    else { if (n == 1) pp_wait_vm<CLO>(); else pp_wait_vm<2 * CLO>(); }
  };

  const int ch = ((lane >> 4) ^ ((lane >> 1) & 7)) << 4;
  const int a_rd = (wm * WM + l15) * 128 + ch, w_rd = (BM + wn * WN + l15) * 128 + ch;

  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nkt) { issue(0, CHI, s); advance(); }
  wait_stages(min(NST - 1, nkt) - 1);
  pp_barrier();
  if (grp == 1) pp_barrier();

  f16x8 fa[MI], fw[NI];
  auto phase_reads = [&](const char* sb, const int hs) {
    const int x = hs << 6;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) fw[ni] = *(const f16x8*)(sb + (w_rd ^ x) + ni * 2048);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) fa[mi] = *(const f16x8*)(sb + (a_rd ^ x) + mi * 2048);
  };
  auto phase_mfma = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  int buf = 0, nbuf = NST - 1;
  for (int kt = 0; kt < nkt; ++kt) {
    if (TWO && kt == nk) {                           // residual pass done: acc = w2_scale * (A W2^T); A W^T accumulates on top
      const float sc2 = p.w2_scale;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] *= sc2;
    }
    const char* sb = smem + buf * STAGE;
    const bool more = kt + NST - 1 < nkt;
    phase_reads(sb, 0);
    if (more) issue(0, H, nbuf);
    pp_wait_lgkm0();
    pp_barrier();
    phase_mfma();
    pp_barrier();
    phase_reads(sb, 1);
    if constexpr (H < CHI) { if (more) issue(H, CHI, nbuf); }
    if (more) advance();
    wait_stages(min(kt + NST - 1, nkt - 1) - (kt + 1));
    pp_wait_lgkm0();
    pp_barrier();
    phase_mfma();
    pp_barrier();
    buf = (buf + 1 == NST) ? 0 : buf + 1;
    nbuf = (nbuf + 1 == NST) ? 0 : nbuf + 1;
  }
  if (grp == 0) pp_barrier();

  PPEpi e;
  e.bias = p.bias; e.rowvec = nullptr; e.R = (const f16*)p.R; e.C = (f16*)p.C; e.Rlo = (const f16*)p.Rlo; e.Clo = (f16*)p.Clo;
  e.rows_per_frame = 1; e.ld_rowvec = 0; e.ldr = p.ldr; e.ldc = p.ldc; e.act = MGLD_ACT_NONE; e.alpha = p.alpha; e.beta = p.beta;
  e.noswap = false;
  pp_epilogue<MI, NI, false>(e, acc, lane, bn0 + wn * WN, [&](const int mi) {
    const int r = wm * WM + mi * 16 + l15;
    return (int)(row0 + (int64_t)(r >> lgP) * HW + (r & (P - 1)));
  });
}

template <int BM, int BN, int WGM, int WGN, int NST>
int launch_pptconv(const MgldIGemm* p, hipStream_t s, int lgP) {
  constexpr int LDS = NST * (BM + BN) * 128;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)pptconv_kernel<BM, BN, WGM, WGN, NST, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)pptconv_kernel<BM, BN, WGM, WGN, NST, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  const int tn = p->N / BN, tm = p->M / BM;
  if (p->W2) hipLaunchKernelGGL((pptconv_kernel<BM, BN, WGM, WGN, NST, true>), dim3(tm * tn), dim3(512), LDS, s, *p, lgP, tn);
  else hipLaunchKernelGGL((pptconv_kernel<BM, BN, WGM, WGN, NST, false>), dim3(tm * tn), dim3(512), LDS, s, *p, lgP, tn);
  return mgld_check_launch("igemm(pptconv)");
}

}  // namespace

namespace mgld_ig {
// does the ping-pong temporal conv take this problem?  *id: 0 = 256 x 256 tiles, 1 = 256 x 128.  p->tune: 0 = planner (large levels only),
// 40 = wherever covered, 15 / 14 (the row-order switches of the implicit-GEMM kernel) and everything else: no.  env MGLD_PPTCONV = 0: off.
bool pptconv_plan(const MgldIGemm* p, int* id, int* lgP) {
  static int knob = -1;
  if (knob < 0) { const char* e = getenv("MGLD_PPTCONV"); knob = e ? atoi(e) : 1; }
  if (!knob || p->mode != MGLD_MODE_TCONV3 || (p->tune != 0 && p->tune != 40)) return false;
  if (p->batch > 1 || p->t_off != 0 || p->out_f32 || p->bias_m || p->rowvec || p->act != MGLD_ACT_NONE || p->tap_inner) return false;
  if (!(p->T == 8 || p->T == 4) || (p->Cin & 63) || p->K != 3 * p->Cin || (p->N & 127)) return false;
  const int lg = p->T == 8 ? 5 : 6;
  if ((p->HW & ((1 << lg) - 1)) || (p->M % (p->T * p->HW))) return false;
  if ((int64_t)(p->T + 2) * p->HW * p->lda * 2 >= 0x7fffffffLL) return false;
  if ((p->lda & 7) || (p->ldw & 7) || (p->ldc & 7) || (((uintptr_t)p->C) & 15) || (p->R && ((p->ldr & 7) || (((uintptr_t)p->R) & 15)))) return false;
  if (p->bias && (((uintptr_t)p->bias) & 15)) return false;
  if (p->W2 && ((((uintptr_t)p->W2) & 15) || !(p->w2_scale > 0.f))) return false;
  const int cfg = (p->N & 255) ? 1 : 0;
  const int64_t tiles = (int64_t)(p->M / 256) * (p->N / (cfg ? 128 : 256));
  if (p->tune == 0 && tiles < (3 * num_cus()) / 4) return false;      // small levels: the split-K implicit-GEMM kernel
  *id = cfg; *lgP = lg;
  return true;
}

int dispatch_pptconv(const MgldIGemm* p, hipStream_t s, int id, int lgP) {
  return id == 0 ? launch_pptconv<256, 256, 2, 4, 2>(p, s, lgP) : launch_pptconv<256, 128, 4, 2, 3>(p, s, lgP);
}

void pptconv_kernel_name(const MgldIGemm* p, int id, char* buf, int buflen) {
  snprintf(buf, buflen, id == 0 ? "pptconv_kernel<256, 256, 2, 4, 2, %s>" : "pptconv_kernel<256, 128, 4, 2, 3, %s>", p->W2 ? "true" : "false");
}
}  // namespace mgld_ig
