// ppgemm.hip — LINEAR contractions (nn.Linear / 1x1 conv: attention.py:51-71,323-330,406-435; openaimodel.py:515-519) on the ping-pong
// structure of pp_common.h: 256-class tiles, one 512-thread workgroup per CU, two wave groups alternating between the matrix pipe and the
// LDS / DMA side, a ring of 64-deep stages filled by `buffer_load ... lds` with counted waits, epilogue from registers.
//
// Why: the 128-class kernels of igemm.hip stage 15-31 bytes per KFLOP through the LDS-DMA path and drain it (`vmcnt(0)` + block barrier)
// at every stage; the transformer projections sit at 0.16-0.24 of the MFMA peak with 1.3-2.4x the algorithmic HBM-side traffic and
// stretch 2.5x when a second segment runs beside them (profiles/r03_inflight_stretch.txt): they are bound by the staging path.  A
// 256 x 256 tile stages 7.8 B/KFLOP, 256 x 160 10 B/KFLOP.
//
// Stage image: (BM + BN) rows x 128 B (64 fp16 of K); 16-B chunk c of row r sits at chunk c ^ ((r >> 1) & 7) (applied on the DMA source
// side, undone by the fragment reads: a 16-lane ds_read_b128 group of the 16x16x32 fragment pattern then touches 16 distinct slots).
// Covered: fp16 out, batch 1, M % BM == 0, N % BN == 0, K % 64 == 0, act in {none, SiLU, GEGLU}; everything else stays on igemm.hip.
#include "pp_common.h"

namespace {
using namespace mgld_ig;

// ablation builds (tools/build_variant.sh <name> ppgemm -DMGLD_PPG_ABLATE=<bits>; timing only, results are wrong): 1 = no epilogue, 2 = the
// K loop runs ONE stage whatever K is (what launch + prologue + epilogue cost), 4 = epilogue without the GEGLU / activation arithmetic
#ifndef MGLD_PPG_ABLATE
#define MGLD_PPG_ABLATE 0
#endif
constexpr int PPG = MGLD_PPG_ABLATE;

template <int BM, int BN, int WGM, int WGN, int NST, bool GEGLU>
__global__ __launch_bounds__(512) void ppgemm_kernel(const MgldIGemm p, const int tiles_m, const int tiles_n, const int order) {
  constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 16, NI = WN / 16;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int NP = (BM + BN) / 8;                 // 1-KiB DMA pieces per stage (8 rows each); piece q -> wave q & 7, slot q >> 3
  constexpr int CLO = NP / 8, CHI = (NP + 7) / 8;   // pieces per wave and stage: waves below NP % 8 carry CHI
  constexpr int JA = BM / 64;                       // slots [0, JA) are activation rows, the rest weight rows
  constexpr int H = (NST == 2) ? CHI : (CHI + 1) / 2;   // slots issued in the first phase of a stage (all of them with two buffers)
  static_assert(WGM * WGN == 8 && BM % 64 == 0 && WM % 16 == 0 && WN % 16 == 0, "tile shape");
  static_assert(NP % 8 == 0 || NP % 8 == 4, "pieces per stage: the same count for the four waves of a group");
  static_assert(NST >= 2 && NST <= 4 && NST * STAGE <= 160 * 1024, "LDS ring");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave / WGN, wn = wave % WGN;
  const bool hi = wave < (NP & 7);

  // tile order (speed only): blocks go round-robin to the 8 XCDs.  order 1: XCD k owns the row panels k, k + 8, ... and walks their column
  // tiles back to back (the activation panel leaves HBM once); order 2: XCD k owns the column tiles k, k + 8, ... (weights >> activations)
  int tile_m, tile_n;
  {
    const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;
    if ((order & 0xff) == 1) { tile_m = xcd + 8 * (j / tiles_n); tile_n = j % tiles_n; }
    else if ((order & 0xff) == 2) { tile_n = xcd + 8 * (j / tiles_m); tile_m = j % tiles_m; }
    else { tile_m = lin / tiles_n; tile_n = lin % tiles_n; }
  }
  const int bm0 = tile_m * BM, bn0 = tile_n * BN;
  const uint32_t lda2 = (uint32_t)p.lda * 2u, ldw2 = (uint32_t)p.ldw * 2u;
  const auto rsA = pp_make_rsrc((const f16*)p.A + (int64_t)bm0 * p.lda, 0xffffffffu);
  const auto rsW = pp_make_rsrc((const f16*)p.W + (int64_t)bn0 * p.ldw, 0xffffffffu);
  // lane's row inside a piece and its source chunk: piece rows 8 q + (l >> 3), physical chunk l & 7 holds logical chunk (l & 7) ^ key(row)
  const uint32_t prow = wave * 8 + (lane >> 3);
  const uint32_t clog = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
  const uint32_t voffA = prow * lda2 + clog * 16, voffW = prow * ldw2 + clog * 16;
  const int nk = (PPG & 2) ? 1 : p.K >> 6;

  // (slot range [j0, j1) is a compile-time constant at every call site; a generic lambda with integral_constant parameters made the host
  // pass of hipcc drop the kernel stubs of all but one instantiation)
  auto issue = [&](const int j0, const int j1, const int buf, const int kt) __attribute__((always_inline)) {
    char* dst = smem + buf * STAGE + wave * 1024;
    const uint32_t kb = (uint32_t)kt * 128u;
#pragma unroll
    for (int j = 0; j < CHI; ++j) {
      if (j < j0 || j >= j1) continue;
      if (j >= CLO && !hi) continue;
      if (j < JA) pp_dma16(rsA, dst + j * 8192, voffA, (uint32_t)j * 64u * lda2 + kb);
      else pp_dma16(rsW, dst + j * 8192, voffW, (uint32_t)(j - JA) * 64u * ldw2 + kb);
    }
  };
  // allow `n` stages of this wave's pieces to stay in flight
  auto wait_stages = [&](const int n) {
    if (n <= 0) { pp_wait_vm<0>(); return; }
    if (hi) { if (n == 1) pp_wait_vm<CHI>(); else pp_wait_vm<2 * CHI>(); }
    else { if (n == 1) pp_wait_vm<CLO>(); else pp_wait_vm<2 * CLO>(); }
  };

  // fragment read offsets (bytes inside a stage), k half 0 / 1
  const int ch = ((lane >> 4) ^ ((lane >> 1) & 7)) << 4;
  const int a_rd = (wm * WM + l15) * 128 + ch, w_rd = (BM + wn * WN + l15) * 128 + ch;

  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: NST - 1 stages in flight, stage 0 landed and visible
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue(0, CHI, s, s);
  wait_stages(min(NST - 1, nk) - 1);
  pp_barrier();
  if (grp == 1) pp_barrier();                      // group 1 runs one barrier behind group 0 from here on

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

  int buf = 0, nbuf = NST - 1;                     // buffer of stage kt / of stage kt + NST - 1
  for (int kt = 0; kt < nk; ++kt) {
    const char* sb = smem + buf * STAGE;
    const bool more = kt + NST - 1 < nk;
    // ---- phase 0 (k 0..31 of the stage)
    phase_reads(sb, 0);
    if (more) issue(0, H, nbuf, kt + NST - 1);
    pp_wait_lgkm0();
    pp_barrier();
    phase_mfma();
    pp_barrier();
    // ---- phase 1 (k 32..63); the next stage must have landed when this phase's first barrier is passed
    phase_reads(sb, 1);
    if constexpr (H < CHI) { if (more) issue(H, CHI, nbuf, kt + NST - 1); }
    wait_stages(min(kt + NST - 1, nk - 1) - (kt + 1));
    pp_wait_lgkm0();
    pp_barrier();
    phase_mfma();
    pp_barrier();
    buf = (buf + 1 == NST) ? 0 : buf + 1;
    nbuf = (nbuf + 1 == NST) ? 0 : nbuf + 1;
  }
  if (grp == 0) pp_barrier();                      // (same barrier count for both groups)

  if constexpr (PPG & 1) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(acc[ni][mi]));
    return;
  }
  PPEpi e;
  e.bias = p.bias; e.rowvec = p.rowvec; e.R = (const f16*)p.R; e.C = (f16*)p.C; e.Rlo = (const f16*)p.Rlo; e.Clo = (f16*)p.Clo;
  e.rows_per_frame = p.rows_per_frame; e.ld_rowvec = p.ld_rowvec; e.ldr = p.ldr; e.ldc = p.ldc; e.act = p.act; e.alpha = p.alpha; e.beta = p.beta;
  e.noswap = (order & 0x100) != 0;
  if (p.ln_part) { e.ln_part = p.ln_part; e.ln_s = p.ln_s; e.ln_chunks = p.ln_chunks; e.ln_M = p.M; e.ln_invK = 1.f / (float)p.K; e.ln_eps = p.ln_eps; }
  // row statistics of the output (MgldIGemm.row_part): every wave writes its columns' share of its rows into the block table (the stages are
  // dead: both wave groups are past their last fragment read), the WGN shares of a row are added after a block barrier
  float* const rtab = (float*)smem;                // [WGN][BM][2]
  if (!GEGLU && p.row_part) e.row_tab = rtab + ((wn * BM) + wm * WM) * 2;
  const int mrow = bm0 + wm * WM + l15;
  if constexpr ((PPG & 4) && GEGLU) {             // (the GEGLU tile written as a plain tile of half the width: store pattern kept, arithmetic dropped)
    e.act = MGLD_ACT_NONE;
    f32x4 half_acc[NI / 2][MI];
#pragma unroll
    for (int ni = 0; ni < NI / 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) half_acc[ni][mi] = acc[ni][mi] + acc[ni + NI / 2][mi];
    pp_epilogue<MI, NI / 2, false>(e, half_acc, lane, (bn0 + wn * WN) / 2, [&](const int mi) { return mrow + mi * 16; });
    return;
  }
  pp_epilogue<MI, NI, GEGLU>(e, acc, lane, bn0 + wn * WN, [&](const int mi) { return mrow + mi * 16; });
  if (!GEGLU && p.row_part) {
    __syncthreads();
    // (thread index rebuilt from the lane count: keeping `tid` alive across the K loop costs the 256 x 320 tile its last registers)
    const int tid2 = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    for (int r = tid2; r < BM; r += 512) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < WGN; ++w) { s1 += rtab[(w * BM + r) * 2]; s2 += rtab[(w * BM + r) * 2 + 1]; }
      *(f32x2*)(p.row_part + ((int64_t)tile_n * p.M + bm0 + r) * 2) = f32x2{s1, s2};
    }
  }
}

// ---- launch plan ----------------------------------------------------------------------------------------------------------
struct PPCfg { int bm, bn, nst; bool geglu_ok; };
constexpr int PP_NCFG = 7;
//   id : tile, wave grid, wave tile, ring                      (LDS)
//    0 : 256 x 256, 2 x 4, 128 x 64, 2 stages of 64 KiB        (128 KiB)   plain / GEGLU
//    1 : 256 x 160, 4 x 2,  64 x 80, 3 stages of 52 KiB        (156 KiB)
//    2 : 128 x 160, 4 x 2,  32 x 80, 4 stages of 36 KiB        (144 KiB)
//    3 : 128 x 256, 2 x 4,  64 x 64, 3 stages of 48 KiB        (144 KiB)   plain / GEGLU
//    4 : 256 x 128, 4 x 2,  64 x 64, 3 stages of 48 KiB        (144 KiB)   plain / GEGLU
//    5 : 128 x 128, 4 x 2,  32 x 64, 4 stages of 32 KiB        (128 KiB)   plain / GEGLU
//    6 : 256 x 320, 4 x 2,  64 x 160, 2 stages of 72 KiB       (144 KiB)
const PPCfg PP_CFG[PP_NCFG] = {{256, 256, 2, true}, {256, 160, 3, false}, {128, 160, 4, false}, {128, 256, 3, true}, {256, 128, 3, true},
                               {128, 128, 4, true}, {256, 320, 2, false}};

template <int BM, int BN, int WGM, int WGN, int NST, bool GEGLU>
int launch_pp(const MgldIGemm* p, hipStream_t s) {
  constexpr int LDS = NST * (BM + BN) * 128;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)ppgemm_kernel<BM, BN, WGM, WGN, NST, GEGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  const int tm = p->M / BM, tn = p->N / BN;
  static int forder = -2;   // env MGLD_PP_ORDER = 0 / 1 / 2 forces the tile order (A/B runs)
  if (forder == -2) { const char* e = getenv("MGLD_PP_ORDER"); forder = e ? atoi(e) : -1; }
  const double abytes = 2.0 * p->M * p->K, wbytes = 2.0 * p->N * p->K;
  int order = forder >= 0 ? forder : (abytes >= wbytes ? 1 : 2);
  if (order == 1 && (tm & 7)) order = (tn & 7) ? 0 : 2;
  if (order == 2 && (tn & 7)) order = (tm & 7) ? 0 : 1;
  static int noswap = -1;   // env MGLD_PP_NOSWAP = 1: 8-byte epilogue stores (A/B)
  if (noswap < 0) { const char* e = getenv("MGLD_PP_NOSWAP"); noswap = e ? atoi(e) : 0; }
  if (noswap) order |= 0x100;
  hipLaunchKernelGGL((ppgemm_kernel<BM, BN, WGM, WGN, NST, GEGLU>), dim3(tm * tn), dim3(512), LDS, s, *p, tm, tn, order);
  return mgld_check_launch("igemm(pp)");
}

inline bool pp_cfg_fits(const MgldIGemm* p, int id) {
  const PPCfg& c = PP_CFG[id];
  if (p->M % c.bm || p->N % c.bn) return false;
  if (p->act == MGLD_ACT_GEGLU && !c.geglu_ok) return false;
  return true;
}

}  // namespace

namespace mgld_ig {
// does the ping-pong LINEAR kernel take this problem?  *id = tile configuration.  p->tune: 20 = yes wherever it is covered (planner's
// configuration), 21 + id = that configuration; 0 = the planner decides (measured table below); env MGLD_PP = 0 switches the family off.
bool ppgemm_plan(const MgldIGemm* p, int* id) {
  static int knob = -1;
  if (knob < 0) { const char* e = getenv("MGLD_PP"); knob = e ? atoi(e) : 1; }
  if (!knob || p->mode != MGLD_MODE_LINEAR) return false;
  if (p->tune != 0 && (p->tune < 20 || p->tune > 20 + PP_NCFG)) return false;
  if (p->batch > 1 || p->W2 || p->out_f32 || p->bias_m || (p->K & 63) || p->K < 64) return false;
  if (!(p->act == MGLD_ACT_NONE || p->act == MGLD_ACT_SILU || p->act == MGLD_ACT_GEGLU)) return false;
  // operands go through unbounded buffer descriptors with 32-bit byte offsets relative to a tile's first row: a 256-row tile of either
  // operand must stay inside that range, and rows must hold K elements
  // (planner queries may leave the leading dimensions unset = 0: dense rows are assumed then)
  const int64_t lda_ = p->lda > 0 ? p->lda : p->K, ldw_ = p->ldw > 0 ? p->ldw : p->K;
  if (lda_ < p->K || ldw_ < p->K || 320 * lda_ * 2 + (int64_t)p->K * 2 >= 0x7fffffffLL || 320 * ldw_ * 2 + (int64_t)p->K * 2 >= 0x7fffffffLL) return false;
  if ((p->ldc & 7) || (((uintptr_t)p->C) & 15) || (p->R && ((p->ldr & 7) || (((uintptr_t)p->R) & 15)))) return false;
  if ((p->bias && (((uintptr_t)p->bias) & 15)) || (p->rowvec && ((((uintptr_t)p->rowvec) & 15) || (p->ld_rowvec & 3)))) return false;
  if (p->ln_part && (!p->ln_s || p->ln_chunks <= 0 || (((uintptr_t)p->ln_s) & 15) || (((uintptr_t)p->ln_part) & 7))) return false;
  if (p->row_part && (p->act == MGLD_ACT_GEGLU || (((uintptr_t)p->row_part) & 7))) return false;
  if (p->tune > 20) {
    if (!pp_cfg_fits(p, p->tune - 21)) return false;
    *id = p->tune - 21;
    return true;
  }
  // planner (measured on MI355X against the 128-class kernels, tools/igemm_bench.py lin, profiles/r04_pp_lin.txt).  A tile is launched
  // once per CU and its prologue / epilogue are exposed, so the family wins where ONE round of tiles covers the chip (tiles in (3/4, 1] x
  // CUs: the largest such tile), where K is deep (>= 1024) and the rounds are nearly full, and for the 256 x 256 tile from two rounds up
  // (GEGLU: 128 x 256 otherwise); short-K problems with several ragged rounds stay on the 128-class kernels (2-3 blocks per CU overlap).
  const int cus = num_cus();
  auto tiles_of = [&](int i) { return (int64_t)(p->M / PP_CFG[i].bm) * (p->N / PP_CFG[i].bn); };
  int bid = -1;
  if (p->act == MGLD_ACT_GEGLU) {
    if (pp_cfg_fits(p, 0) && tiles_of(0) >= 2 * cus) bid = 0;
    else if (pp_cfg_fits(p, 3) && tiles_of(3) >= cus) bid = 3;
  } else {
    int64_t area = 0;
    for (int i = 0; i < PP_NCFG; ++i) {                       // one round
      if (!pp_cfg_fits(p, i)) continue;
      const int64_t t = tiles_of(i), a = (int64_t)PP_CFG[i].bm * PP_CFG[i].bn;
      if (4 * t > 3 * cus && t <= cus && a > area) { area = a; bid = i; }
    }
    if (bid < 0 && pp_cfg_fits(p, 0) && tiles_of(0) >= 2 * cus) bid = 0;
    // round 5 (two segments batched as clips: M = 65536 at the 64^2 level): the 256 x 320 tile over WHOLE rounds — the fused q|k|v
    // projection N = 960, K = 320 is 768 tiles = three full rounds: 87.6 -> 61.8 us (profiles/r05_pp_lin_mscale2.txt); ragged rounds of
    // this short-K tile, and the 256 x 160 tile at K = 320 even in full rounds (46.8 vs 41.2 us at M = 32768), stay on the 128-class kernel
    if (bid < 0 && pp_cfg_fits(p, 6) && tiles_of(6) >= 2 * cus && tiles_of(6) % cus == 0) bid = 6;
    // (round 6, measured and NOT taken: the 256 x 160 tile for the fused q|k|v projection of the 32^2 level at two clips — M = 16384, N = 1920,
    //  K = 640, three full rounds, 58.3 -> 55.8 us in isolation and the LayerNorm in front folds into it — is 0.3 % SLOWER end to end,
    //  13.41 / 13.43 / 13.44 against 13.43 / 13.48 / 13.48 frames/s alternated on one box: stays on the 128-class kernel)
    if (bid < 0 && p->K >= 1024) {
      static const double eff[PP_NCFG] = {1.00, 0.90, 0.72, 0.85, 0.85, 0.68, 0.95};
      double best = 0.0;
      for (int i = 0; i < PP_NCFG; ++i) {
        if (!pp_cfg_fits(p, i)) continue;
        const int64_t t = tiles_of(i), rounds = (t + cus - 1) / cus;
        const double fill = (double)t / (double)(rounds * cus);
        if (fill >= 0.85 && eff[i] * fill > best) { best = eff[i] * fill; bid = i; }
      }
    }
  }
  if (bid < 0 && p->tune == 20) {                             // forced family: the largest tile that fits
    int64_t area = 0;
    for (int i = 0; i < PP_NCFG; ++i)
      if (pp_cfg_fits(p, i) && (int64_t)PP_CFG[i].bm * PP_CFG[i].bn > area) { area = (int64_t)PP_CFG[i].bm * PP_CFG[i].bn; bid = i; }
  }
  if (bid < 0) return false;
  *id = bid;
  return true;
}

int ppgemm_row_chunks(const MgldIGemm* p, int id) { return p->N / PP_CFG[id].bn; }

int dispatch_ppgemm(const MgldIGemm* p, hipStream_t s, int id) {
  const bool g = p->act == MGLD_ACT_GEGLU;
  switch (id) {
    case 0: return g ? launch_pp<256, 256, 2, 4, 2, true>(p, s) : launch_pp<256, 256, 2, 4, 2, false>(p, s);
    case 1: return launch_pp<256, 160, 4, 2, 3, false>(p, s);
    case 2: return launch_pp<128, 160, 4, 2, 4, false>(p, s);
    case 3: return g ? launch_pp<128, 256, 2, 4, 3, true>(p, s) : launch_pp<128, 256, 2, 4, 3, false>(p, s);
    case 4: return g ? launch_pp<256, 128, 4, 2, 3, true>(p, s) : launch_pp<256, 128, 4, 2, 3, false>(p, s);
    case 5: return g ? launch_pp<128, 128, 4, 2, 4, true>(p, s) : launch_pp<128, 128, 4, 2, 4, false>(p, s);
    default: return launch_pp<256, 320, 4, 2, 2, false>(p, s);
  }
}

void ppgemm_kernel_name(const MgldIGemm* p, int id, char* buf, int buflen) {
  static const int g[PP_NCFG][5] = {{256, 256, 2, 4, 2}, {256, 160, 4, 2, 3}, {128, 160, 4, 2, 4}, {128, 256, 2, 4, 3}, {256, 128, 4, 2, 3}, {128, 128, 4, 2, 4},
                                    {256, 320, 4, 2, 2}};
  snprintf(buf, buflen, "ppgemm_kernel<%d, %d, %d, %d, %d, %s>", g[id][0], g[id][1], g[id][2], g[id][3], g[id][4],
           p->act == MGLD_ACT_GEGLU ? "true" : "false");
}
}  // namespace mgld_ig
