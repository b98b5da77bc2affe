// conv3r.hip — the 3x3 / stride 1 / pad 1 convolutions (openaimodel.py:176,221,401-405,429-436; model.py:114-118) on the ping-pong
// structure of pp_common.h.  Same staging idea as conv3q.hip — a (TY+2) x (TX+2) input PATCH of a 32-channel slice is DMA'd once and all
// nine taps read their fragments from it; weights stream in the tiled layout of MgldIGemm.tap_inner = 2 — but:
//   * one 512-thread block per CU, two wave groups alternating between the matrix pipe and the LDS / DMA side (no per-stage block-wide
//     drain: conv3q sat at 0.39 of the MFMA peak with the matrix pipe 39.7 % busy and the waves parked 43.7 % on the stage barrier,
//     profiles/r03_pmc_conv3q.txt);
//   * a PHASE is one tap of one slice (K = 32): the weights of a phase are one ring slot of BN x 64 B, the ring is NSW slots deep and
//     filled NSW - 1 phases ahead with counted waits; the patch of the next slice arrives during taps 0..2 of the current one;
//   * v_mfma_f32_16x16x32_f16, so the N = 320 layers run 256 pixels x 160 channels per block (= 256 blocks at 8 x 64^2) with 64 x 80
//     wave tiles: 4.7 staged bytes per KFLOP;
//   * fragment reads are one ds_read_b128 per 16-row fragment; the four k groups of the 16x16x32 pattern read the 16-byte chunks of a row
//     in the order {0, 3, 1, 2} and the patch image is swizzled by (row >> 2) & 1: conflict-free for every tap shift (see the kernel);
// Out-of-image patch pixels: the DMA lane's offset is pushed past the descriptor's range and the hardware writes zeros.
// Covered: tap_inner = 2 weights, Cin % 32 == 0, N % BN == 0, no upsample fold, no K split, fp16 out,
// act in {none, SiLU}; everything else stays on conv3q.hip.
#include "pp_common.h"

#ifndef MGLD_PP_ABLATE
#define MGLD_PP_ABLATE 0   // timing-only ablation builds (1-8: wrong results): 1 no DMA in the loop, 2 no fragment reads, 4 no MFMA, 8 no barriers, 16 no setprio, 32 lgkmcnt(0) after the barrier, 64 tap offsets recomputed per phase
#endif

namespace {
using namespace mgld_ig;
constexpr int PPA = MGLD_PP_ABLATE;

// TWO: the instantiation that runs the weight-residual pass (MgldIGemm.W2): the slices are walked twice over the same patches — first
// against the scaled fp16 residual of the weights (same tiled layout, its own buffer descriptor), then, after ONE multiplication of the
// accumulators by w2_scale, against the weights themselves.  A "visit" v below is slice v % nh of pass v / nh.
template <int TY, int TX, int BN, int WGM, int WGN, int NSW, bool TWO, bool SPLIT>
__global__ __launch_bounds__(512) void conv3r_kernel(const MgldIGemm p, const int tiles_x, const int tiles_y, const int order, float* __restrict__ ws,
                                                     const int hsplit) {
  constexpr int BM = TY * TX, WM = BM / WGM, WN = BN / WGN, MI = WM / 16, NI = WN / 16;
  // patch rows are laid out with a row stride of PW pixels, PW a multiple of 8 (TX + 2 rounded up; the pad columns are DMA'd as zeros):
  // the chunk swizzle bit (row >> 2) & 1 of a fragment row is then the same for every tile row, i.e. the nine taps of every fragment are
  // THREE per-lane offsets (dx) plus immediates — no address arithmetic in the loop
  // TX == 8 ("frame-stacked" tiles, the 8^2 UNet level): the tile is TY / 8 whole 8 x 8 frames, its patch their (8 + 2) x PW patches one
  // below the other; a 16-row fragment is two image rows of 8 pixels (two runs of 8 patch rows 16 apart), for which the conflict-free
  // chunk swizzle is ((row >> 2) & 1) << 1 (same k-group order; enumeration over the ds_read_b128 lane groups as for the wide tiles)
  constexpr bool W8 = (TX == 8);
  constexpr int FT = W8 ? TY / 8 : 1;
  constexpr int PW = (TX + 2 + 7) & ~7, PH = W8 ? FT * 10 : TY + 2, PR = PW * PH;
  constexpr int NPA = (PR + 15) / 16, NPT = (NPA + 7) / 8;      // 1-KiB patch pieces; pieces per wave (taps 0 .. NPT-1 of the previous slice)
  constexpr int A_BYTES = NPT * 8 * 1024;
  constexpr int WSLOT = BN * 64, NPW = BN / 16;                 // one tap of one slice: BN rows x 64 B = NPW pieces
  constexpr int CWLO = NPW / 8, CWHI = (NPW + 7) / 8;
  constexpr int D = NSW - 1;                                    // weights are issued D phases ahead
  constexpr int W_BASE = 2 * A_BYTES;
  static_assert(WGM * WGN == 8 && (TX % 16 == 0 || (W8 && TY % 8 == 0 && WM <= 64)) && WM % 16 == 0 && WN % 16 == 0 && BN % 16 == 0, "tile shape");
  static_assert(WM % TX == 0, "a wave tile is whole tile rows");
  static_assert(NPT <= 10 - D && D >= 2 && D <= 7 && W_BASE + NSW * WSLOT <= 160 * 1024, "ring");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave / WGN, wn = wave % WGN;
  const bool hi = wave < (NPW & 7);

  int tile_m = blockIdx.x, tile_n = blockIdx.y;
  if ((gridDim.x & 7) == 0) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, j = lin >> 3;
    if ((order & 0xff) == 1) {
      const int q = xcd * ((int)(gridDim.x * gridDim.y) >> 3) + j;
      tile_n = q / (int)gridDim.x;
      tile_m = q - tile_n * (int)gridDim.x;
    } else {
      tile_m = xcd + 8 * (j / (int)gridDim.y);
      tile_n = j % (int)gridDim.y;
    }
  }
  const int tpf = W8 ? 1 : tiles_x * tiles_y;
  const int frame = W8 ? tile_m * FT : tile_m / tpf;          // W8: first frame of the tile (tiles_x carries the frame count)
  const int trem = W8 ? 0 : tile_m - frame * tpf;
  const int tyi = W8 ? 0 : trem / tiles_x, txi = W8 ? 0 : trem - tyi * tiles_x;
  const int y0 = tyi * TY, x0 = txi * TX;
  const int nfr = W8 ? min(FT, tiles_x - frame) : 1;        // frames of this tile that exist
  const int bn0 = tile_n * BN;
  // SPLIT (the 16^2 level: too few tiles for the chip): blockIdx.z owns the channel slices [h0, h0 + nh) and writes raw fp32 sums into its
  // slab of the K-split workspace; splitk_reduce_kernel adds the slabs and runs the epilogue
  const int Hin = p.Hin, Win = p.Win, nh_all = p.Cin >> 5;
  const int h0 = SPLIT ? (int)blockIdx.z * hsplit : 0;
  const int nh = SPLIT ? min(hsplit, nh_all - h0) : nh_all;

  // ---- DMA sources
  const int64_t fpix = (int64_t)frame * Hin * Win;
  const uint32_t fbytes = (uint32_t)min((int64_t)0x7fffffff, (int64_t)nfr * Hin * Win * p.lda * 2);
  const auto rsA = pp_make_rsrc((const f16*)p.A + fpix * p.lda, fbytes);
  const auto rsW = pp_make_rsrc(p.W, 0xffffffffu);
  const auto rsW2 = pp_make_rsrc(TWO ? p.W2 : p.W, 0xffffffffu);
  const int nv = TWO ? 2 * nh : nh;                  // visits
  uint32_t voffA[NPT];
#pragma unroll
  for (int s = 0; s < NPT; ++s) {
    const int j = (s * 8 + wave) * 16 + (lane >> 2);
    const int pr = j / PW, pc = j - pr * PW;
    if constexpr (W8) {
      const int fr = pr / 10, y = pr - fr * 10 - 1, x = pc - 1;
      const bool ok = (j < PR) && (fr < nfr) && ((unsigned)y < 8u) && ((unsigned)x < 8u);
      const int cl = (lane & 3) ^ (((j >> 2) & 1) << 1);
      voffA[s] = ok ? (uint32_t)((((fr * 8 + y) * 8 + x) * p.lda + cl * 8) * 2) : 0x80000000u;
    } else {
    const int y = y0 - 1 + pr, x = x0 - 1 + pc;
    const bool ok = (j < PR) && (pc < TX + 2) && ((unsigned)y < (unsigned)Hin) && ((unsigned)x < (unsigned)Win);
    const int cl = (lane & 3) ^ ((j >> 2) & 1);       // patch image: chunk c of row j sits at c ^ ((j >> 2) & 1)
    voffA[s] = ok ? (uint32_t)(((y * Win + x) * p.lda + cl * 8) * 2) : 0x80000000u;   // past num_records: the DMA writes zeros
    }
  }
  const uint32_t voffW = lane * 16;
  uint32_t wbase[CWHI];        // byte offset of this wave's weight pieces at (slice 0, tap 0)
#pragma unroll
  for (int k = 0; k < CWHI; ++k) {
    const int gr = (bn0 >> 4) + k * 8 + wave;
    wbase[k] = (uint32_t)((((gr >> 2) * nh_all * 3) * 4 + (gr & 3)) * 3) * 1024u;
  }
  auto issue_patch = [&](const int s, const int par, const int vv) __attribute__((always_inline)) {
    const int hh = h0 + ((TWO && vv >= nh) ? vv - nh : vv);
    pp_dma16(rsA, smem + par * A_BYTES + (s * 8 + wave) * 1024, voffA[s], (uint32_t)hh * 64u);
  };
  auto issue_w = [&](const int slot, const int vv, const int tap) __attribute__((always_inline)) {
    const int hh = h0 + ((TWO && vv >= nh) ? vv - nh : vv);
    const int dy = (tap * 11) >> 5, dx = tap - dy * 3;
    const uint32_t so = (uint32_t)hh * (36u * 1024u) + (uint32_t)dy * (12u * 1024u) + (uint32_t)dx * 1024u;
    char* dst = smem + W_BASE + slot * WSLOT + wave * 1024;
    if (TWO && vv < nh) {                          // residual pass
#pragma unroll
      for (int k = 0; k < CWHI; ++k) {
        if (k >= CWLO && !hi) continue;
        pp_dma16(rsW2, dst + k * 8192, voffW, wbase[k] + so);
      }
    } else {
#pragma unroll
      for (int k = 0; k < CWHI; ++k) {
        if (k >= CWLO && !hi) continue;
        pp_dma16(rsW, dst + k * 8192, voffW, wbase[k] + so);
      }
    }
  };

  // ---- fragment read offsets
  // k group g of a fragment reads the 16-byte chunk PG(g) = {0, 3, 1, 2}[g] of its 64-byte row (both operands alike, so the products pair
  // up).  Found by enumeration over the ds_read_b128 lane groups of MI355X_MICROARCH.md: with this order the weight image (chunk swizzle
  // (row >> 2) & 3, 16-aligned fragments) is conflict-free, and a patch image swizzled by (row >> 2) & 1 is conflict-free for EVERY
  // alignment of the 16 consecutive patch rows a tap shifts a fragment to.
  const int pg = (0x9c >> (2 * g)) & 3;
  const int w_rd = W_BASE + (wn * WN + l15) * 64 + (((pg ^ (l15 >> 2)) & 3) << 4);
  // patch row of this lane's pixel in fragment 0 at tap (0, 0); fragment mi / tap (dy, dx) add compile-time rows (multiples of 8, or 16)
  const int R0 = wm * WM;
  const int ir0 = (R0 >> 3) + (l15 >> 3);              // W8: image row of this lane's pixel, counted over the stacked frames
  const int jb0 = W8 ? (ir0 + 2 * (ir0 >> 3)) * PW + (l15 & 7) : (R0 / TX) * PW + (R0 & (TX - 1)) + l15;
  int a_off[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int jj = jb0 + dx;
    a_off[dx] = jj * 64 + (((W8 ? ((jj >> 2) & 1) << 1 : (jj >> 2) & 1) ^ pg) << 4);
  }
  // rows of fragment mi below fragment 0 (W8: two image rows per fragment, + the two halo rows of every frame boundary crossed; a wave tile
  // starts at a multiple of 32 rows and is at most one frame, so the crossing count is mi >> 2)
  auto frag_rows = [](const int mi) { return W8 ? (2 * mi + 2 * (mi >> 2)) * PW : ((mi * 16) / TX) * PW + ((mi * 16) & (TX - 1)); };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: patch of slice 0, weights of phases 0 .. D-1; phase 0's operands landed and visible
  const int F = 9 * nv;
#pragma unroll
  for (int s = 0; s < NPT; ++s) issue_patch(s, 0, 0);
#pragma unroll
  for (int f = 0; f < D; ++f)
    if (f < F) issue_w(f, f / 9, f % 9);
  if (hi) pp_wait_vm<(D - 1) * CWHI>(); else pp_wait_vm<(D - 1) * CWLO>();
  pp_barrier();
  if (grp == 1) pp_barrier();                      // group 1 runs one barrier behind group 0 from here on

  f16x8 fa[MI], fw[NI];
  if constexpr (PPA & 2) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) fa[mi] = *(const f16x8*)(smem + lane * 16 + mi * 1024);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) fw[ni] = *(const f16x8*)(smem + W_BASE + lane * 16 + ni * 1024);
  }
  int slot = 0;                                    // ring slot of the current phase
  // one slice: nine phases.  MORE: there is a next slice (its patch is issued during taps 0 .. NPT-1, weights run ahead into it)
  auto phase = [&](const int v, auto MORE_, auto T_) __attribute__((always_inline)) {
    constexpr bool MORE = decltype(MORE_)::value;
    constexpr int t = decltype(T_)::value;
    constexpr int dy = t / 3, dx = t - dy * 3;
    const int pa = v & 1;
    const int abase = pa * A_BYTES;
    // ---- L: fragments of this phase
    const char* ws = smem + slot * WSLOT;
    if constexpr (!(PPA & 2)) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      fw[ni] = *(const f16x8*)(ws + w_rd + ni * 1024);
    }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      if constexpr (PPA & 2) continue;
      fa[mi] = *(const f16x8*)(smem + abase + a_off[dx] + (dy * PW + frag_rows(mi)) * 64);
    }
    // ---- DMA: next slice's patch (taps 0 .. NPT-1), weights of phase f + D into the slot phase f - 1 read
    if constexpr (MORE && t < NPT && !(PPA & 1)) issue_patch(t, pa ^ 1, v + 1);
    if constexpr ((MORE || t + D < 9) && !(PPA & 1)) {
      const int ns = (slot == 0) ? NSW - 1 : slot - 1;
      if constexpr (t + D < 9) issue_w(ns, v, t + D); else issue_w(ns, v + 1, t + D - 9);
    }
    // ---- phase f + 1's operands must have landed: everything but the issues of the last D - 1 phases may stay in flight
    {
      constexpr int np = [] {      // patch pieces among them
        int n = 0;
        for (int i = 0; i < D - 1; ++i) { const int tau = t - i; if (tau >= 0 && MORE && tau < NPT) ++n; }
        return n;
      }();
      constexpr int nw = [] {      // phases among them that issued weights
        int n = 0;
        for (int i = 0; i < D - 1; ++i) { const int tau = t - i; if (tau < 0 || MORE || tau + D < 9) ++n; }
        return n;
      }();
      if (hi) pp_wait_vm<np + nw * CWHI>(); else pp_wait_vm<np + nw * CWLO>();
    }
    if constexpr (!(PPA & 32)) pp_wait_lgkm0();
    if constexpr (!(PPA & 8)) pp_barrier(); else __builtin_amdgcn_sched_barrier(0);
    if constexpr (PPA & 32) { pp_wait_lgkm0(); __builtin_amdgcn_sched_barrier(0); }
    // ---- M
    if constexpr (!(PPA & 16)) __builtin_amdgcn_s_setprio(1);
    if constexpr (PPA & 4) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(fa[mi]));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(fw[ni]));
    } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    }
    if constexpr (!(PPA & 16)) __builtin_amdgcn_s_setprio(0);
    if constexpr (!(PPA & 8)) pp_barrier(); else __builtin_amdgcn_sched_barrier(0);
    slot = (slot + 1 == NSW) ? 0 : slot + 1;
  };
  // one slice: nine phases.  MORE: there is a next slice (its patch is issued during taps 0 .. NPT-1, weights run ahead into it)
  auto slice = [&](const int v, auto MORE_) __attribute__((always_inline)) {
    phase(v, MORE_, std::integral_constant<int, 0>{}); phase(v, MORE_, std::integral_constant<int, 1>{}); phase(v, MORE_, std::integral_constant<int, 2>{});
    phase(v, MORE_, std::integral_constant<int, 3>{}); phase(v, MORE_, std::integral_constant<int, 4>{}); phase(v, MORE_, std::integral_constant<int, 5>{});
    phase(v, MORE_, std::integral_constant<int, 6>{}); phase(v, MORE_, std::integral_constant<int, 7>{}); phase(v, MORE_, std::integral_constant<int, 8>{});
  };
  for (int v = 0; v + 1 < nv; ++v) {
    if (TWO && v == nh) {                            // residual pass done: acc = w2_scale * (A W2^T); A W^T accumulates on top
      const float sc2 = p.w2_scale;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] *= sc2;
    }
    slice(v, std::true_type{});
  }
  if (TWO && nh == 1) {
    const float sc2 = p.w2_scale;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[ni][mi] *= sc2;
  }
  slice(nv - 1, std::false_type{});
  if (grp == 0) pp_barrier();

  if constexpr (PPA & 128) {                       // (ablation: no epilogue)
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
  const int Hout = p.Hout, Wout = p.Wout;
  const int fbase = frame * Hout * Wout;
  const auto row_of = [&](const int mi) {
    const int R = wm * WM + mi * 16 + l15;
    if constexpr (W8) return ((R >> 6) < nfr) ? fbase + R : -1;
    const int y = y0 + R / TX, x = x0 + (R & (TX - 1));
    return (y < Hout && x < Wout) ? fbase + y * Wout + x : -1;
  };
  if constexpr (SPLIT) {                           // raw sums: lane = one output row, 4 consecutive channels per fragment
    float* slab = ws + (int64_t)blockIdx.z * p.M * p.N;
    const int nb = bn0 + wn * WN + 4 * g;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = row_of(mi);
      if (m < 0) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) *(f32x4*)(slab + (int64_t)m * p.N + nb + ni * 16) = acc[ni][mi];
    }
    return;
  }
  if (p.out_f32) {                                 // fp32 rows + fp32 residual: lane = one output row, 4 consecutive channels per fragment
    float* Cf = (float*)p.C;
    const float* Rf = p.r_f32 ? (const float*)p.R : nullptr;
    const int nb = bn0 + wn * WN + 4 * g;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = row_of(mi);
      if (m < 0) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = nb + ni * 16;
        f32x4 v = acc[ni][mi];
        if (p.bias) v += *(const f32x4*)(p.bias + n);
        if (Rf) v += p.beta * *(const f32x4*)(Rf + (int64_t)m * p.ldr + n);
        *(f32x4*)(Cf + (int64_t)m * p.ldc + n) = v;
      }
    }
    return;
  }
  if (p.gn_part) {
    // GroupNorm statistics of the output (MgldIGemm.gn_part): every wave's per-channel sums of its stored rows meet in LDS (the stages
    // are dead: both wave groups are past their last fragment read), one row of part[] per tile.  A tile lies in one frame.
    float* table = (float*)smem;
    pp_epilogue_stats<MI, NI, true>(e, acc, lane, bn0 + wn * WN, p.rowvec ? (frame * Hout * Wout) / p.rows_per_frame : -1, row_of,
                              table + (wm * BN + wn * WN) * 2);
    __syncthreads();
    pp_stats_flush<WGM, BN>(table, p.gn_part, (int64_t)tile_m, p.N, bn0, tid);
    return;
  }
  pp_epilogue<MI, NI, false, true>(e, acc, lane, bn0 + wn * WN, row_of);
}

// ---- launch plan ----------------------------------------------------------------------------------------------------------
//   id : pixel tile, weight rows, wave grid, wave tile, weight ring
//    0 : 8 x 32, 160, 4 x 2,  64 x 80, 6 slots of 10 KiB       N = 320 k at 64^2 and larger
//    1 : 8 x 16, 320, 2 x 4,  64 x 80, 4 slots of 20 KiB       N = 320 k, 128-pixel tiles
//    2 : 8 x 32, 128, 4 x 2,  64 x 64, 6 slots of  8 KiB       N = 128 k (VAE 512^2, SPADE)
//    3 : 8 x 32, 256, 2 x 4, 128 x 64, 5 slots of 16 KiB       N = 256 k (VAE 128^2, 256^2)
//    4 : 16 x 16, 160, 4 x 2, 64 x 80, 6 slots                 square tiles (A/B)
//    5 : 8 x 16, 160, 4 x 2,  32 x 80, 6 slots of 10 KiB       N = 320 k at 32^2 (one block per CU at 8 frames x 640 channels)
//    6 : 16 x 32, 80, 8 x 1,  64 x 80, 5 slots of  5 KiB       512-pixel tiles: half the weight bytes per FLOP of configuration 0
//    7 : 8 x 32, 80, 8 x 1,   32 x 80, 6 slots of  5 KiB       256-pixel tiles x 80 channels (32^2 level: one block per CU)
//    8 : 16 x 32, 128, 8 x 1, 64 x 128, 5 slots of 8 KiB       512-pixel tiles for N = 128 k
//    9 : 4 frames of 8 x 8, 160, 4 x 2, 64 x 80, 5 slots       frame-stacked tiles (ty = 8 x frames, tx = 8): the 8^2 UNet level, K split
//   10 : 2 frames of 8 x 8, 160, 4 x 2, 32 x 80, 6 slots
struct R3Cfg { int ty, tx, bn; };
constexpr int R3_NCFG = 11;
const R3Cfg R3_CFG[R3_NCFG] = {{8, 32, 160}, {8, 16, 320}, {8, 32, 128}, {8, 32, 256}, {16, 16, 160}, {8, 16, 160}, {16, 32, 80}, {8, 32, 80}, {16, 32, 128},
                               {32, 8, 160}, {16, 8, 160}};

template <int TY, int TX, int BN, int NSW>
constexpr int conv3r_lds() {
  constexpr int PW = (TX + 2 + 7) & ~7, PH = TX == 8 ? (TY / 8) * 10 : TY + 2, NPA = (PW * PH + 15) / 16, NPT = (NPA + 7) / 8;
  return 2 * NPT * 8 * 1024 + NSW * BN * 64;
}

template <int TY, int TX, int BN, int WGM, int WGN, int NSW, bool TWO, bool SPLIT>
int launch_conv3r_(const MgldIGemm* p, hipStream_t s, int splits) {
  constexpr int lds = conv3r_lds<TY, TX, BN, NSW>();
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv3r_kernel<TY, TX, BN, WGM, WGN, NSW, TWO, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  const int frames = p->M / (p->Hout * p->Wout);
  constexpr bool W8 = (TX == 8);                   // frame-stacked tiles: TY / 8 frames per tile, tiles_x carries the frame count
  const int tiles_x = W8 ? frames : cdiv(p->Wout, TX), tiles_y = W8 ? 1 : cdiv(p->Hout, TY);
  const int nh = p->Cin >> 5, hsplit = cdiv(nh, splits);
  dim3 grid(W8 ? cdiv(frames, TY / 8) : frames * tiles_x * tiles_y, p->N / BN, SPLIT ? cdiv(nh, hsplit) : 1);
  static int forder = -2, noswap = -1;
  if (forder == -2) { const char* e = getenv("MGLD_CONV3R_ORDER"); forder = e ? atoi(e) : -1; }
  if (noswap < 0) { const char* e = getenv("MGLD_PP_NOSWAP"); noswap = e ? atoi(e) : 0; }
  const double wbytes = 2.0 * p->N * p->K, abytes = 2.0 * p->M * p->Cin;
  const int order = (forder >= 0 ? forder : (wbytes > 2.0 * abytes ? 1 : 0)) | (noswap ? 0x100 : 0);
  hipLaunchKernelGGL((conv3r_kernel<TY, TX, BN, WGM, WGN, NSW, TWO, SPLIT>), grid, dim3(512), lds, s, *p, tiles_x, tiles_y, order,
                     SPLIT ? g_ws : nullptr, hsplit);
  if (SPLIT) launch_splitk_reduce(p, s, (int)grid.z);
  return mgld_check_launch("igemm(conv3r)");
}
template <int TY, int TX, int BN, int WGM, int WGN, int NSW, bool SPLITTABLE = false>
int launch_conv3r(const MgldIGemm* p, hipStream_t s, int splits) {
  if constexpr (SPLITTABLE) {
    if (splits > 1)
      return p->W2 ? launch_conv3r_<TY, TX, BN, WGM, WGN, NSW, true, true>(p, s, splits) : launch_conv3r_<TY, TX, BN, WGM, WGN, NSW, false, true>(p, s, splits);
  }
  return p->W2 ? launch_conv3r_<TY, TX, BN, WGM, WGN, NSW, true, false>(p, s, 1) : launch_conv3r_<TY, TX, BN, WGM, WGN, NSW, false, false>(p, s, 1);
}

}  // namespace

namespace mgld_ig {
// does the ping-pong patch convolution take this problem?  p->tune: 30 = yes wherever it is covered, 31 + id = that configuration,
// 50 + id = that configuration with the channel slices split over grid.z where its tiles do not fill the chip (configurations 4, 5, 8),
// 0 = the planner decides; other values: no.  env MGLD_CONV3R = 0 switches the family off, MGLD_CONV3R_SPLIT = 0 the K split.
bool conv3r_plan(const MgldIGemm* p, int* id, int* splits) {
  *splits = 1;
  static int knob = -1;
  if (knob < 0) { const char* e = getenv("MGLD_CONV3R"); knob = e ? atoi(e) : 1; }
  if (!knob || p->mode != MGLD_MODE_CONV3X3 || p->tap_inner != 2) return false;
  if (p->tune != 0 && (p->tune < 30 || p->tune > 30 + R3_NCFG) && (p->tune < 50 || p->tune >= 50 + R3_NCFG)) return false;
  if (p->kh > 0 && !(p->kh == 3 && p->kw == 3)) return false;
  if (p->stride != 1 || p->pad_t != 1 || p->pad_l != 1 || p->batch > 1 || p->up2 || p->bias_m || (p->Cin & 31)) return false;
  if (!(p->act == MGLD_ACT_NONE || p->act == MGLD_ACT_SILU || p->act == MGLD_ACT_RELU)) return false;
  if (p->act == MGLD_ACT_RELU && p->tune == 0) {    // round 6: SPADE's shared convolution + ReLU (64^2: 41.0 -> 33.4 us in isolation); env MGLD_CONV3R_RELU = 0: conv3q as before (A/B)
    static int onr = -1;
    if (onr < 0) { const char* e = getenv("MGLD_CONV3R_RELU"); onr = e ? atoi(e) : 1; }
    if (!onr) return false;
  }
  // fp32 output (+ fp32 residual): the plain form only — bias, no activation / row vector / statistics (the split-fp16 convolutions of
  // the high-precision encoder, hpenc.hip)
  if (p->out_f32 && (p->act != MGLD_ACT_NONE || p->rowvec || p->gn_part || p->alpha != 1.f || (p->R && !p->r_f32) || (p->N & 3))) return false;
  if (p->r_f32 && !p->out_f32) return false;
  const bool w8 = (p->Hout == 8 && p->Wout == 8);     // the 8^2 level: frame-stacked tiles (configurations 9, 10), K split only
  if (p->Hout != p->Hin || p->Wout != p->Win || (!w8 && (p->Wout < 16 || p->Hout < 8)) || (p->M % (p->Hout * p->Wout))) return false;
  if (w8 && p->W2) return false;
  if ((p->lda & 7) || (p->ldc & 7) || (((uintptr_t)p->C) & 15) || (p->R && ((p->ldr & 7) || (((uintptr_t)p->R) & 15)))) return false;
  if (p->W2 && ((((uintptr_t)p->W2) & 15) || !(p->w2_scale > 0.f))) return false;
  if ((p->bias && (((uintptr_t)p->bias) & 15)) || (p->rowvec && ((((uintptr_t)p->rowvec) & 15) || (p->ld_rowvec & 3)))) return false;
  if ((int64_t)p->Hin * p->Win * p->lda * 2 >= 0x7fffffffLL) return false;      // a frame must fit the descriptor's 31-bit range
  if ((int64_t)((p->N + 63) / 64 * 64) * 9 * p->Cin * 2 >= 0xffffffffLL) return false;   // the weight image is addressed with 32-bit byte offsets
  auto fits = [&](int i) { return p->N % R3_CFG[i].bn == 0 && (R3_CFG[i].tx == 8) == w8; };
  const int frames = p->M / (p->Hout * p->Wout), cus = num_cus();
  auto tiles_of = [&](int i) {
    if (R3_CFG[i].tx == 8) return (int64_t)cdiv(frames, R3_CFG[i].ty / 8) * (p->N / R3_CFG[i].bn);
    return (int64_t)frames * cdiv(p->Hout, R3_CFG[i].ty) * cdiv(p->Wout, R3_CFG[i].tx) * (p->N / R3_CFG[i].bn);
  };
  // K split of configuration i: as many slabs as keep tiles x slabs within the chip (>= 2, at most one per two channel slices), workspace permitting
  auto splits_of = [&](int i) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("MGLD_CONV3R_SPLIT"); on = e ? atoi(e) : 1; }
    if (!on || !(i == 4 || i == 5 || i == 8 || i == 9 || i == 10) || p->gn_part) return 1;
    const int nh = p->Cin >> 5;
    int sp = (int)(cus / tiles_of(i));
    if (sp > nh / 2) sp = nh / 2;
    if (sp < 2 || g_ws == nullptr || (size_t)sp * p->M * p->N * sizeof(float) > g_ws_bytes || (p->N & 3)) return 1;
    return cdiv(nh, cdiv(nh, sp));            // slabs actually launched
  };
  if (p->tune >= 50) {
    const int i = p->tune - 50;
    if (!fits(i)) return false;
    *id = i;
    *splits = splits_of(i);
    return true;
  }
  if (p->tune > 30) {
    if (!fits(p->tune - 31)) return false;
    *id = p->tune - 31;
    return true;
  }
  // planner (measured against conv3q on MI355X, tools/igemm_bench.py conv / vae, profiles/r04_pp_conv.txt): the first configuration of the
  // list for N's divisibility whose tile count covers most of the chip and whose tile width the image fills.  Small frames with few
  // tiles (the 16^2 / 8^2 UNet levels) stay on conv3q and its K split.
  int bid = -1;
  auto take = [&](int i, int64_t min_tiles) {
    if (bid >= 0 || !fits(i) || p->Wout < R3_CFG[i].tx || p->Hout < R3_CFG[i].ty) return;
    if (tiles_of(i) >= min_tiles) bid = i;
  };
  const int64_t most = (3 * cus) / 4;
  if (p->N % 128 == 0) { take(8, most); take(3, most); take(2, cus / 2 - cus / 8); }
  if (p->N % 80 == 0) { take(6, most); take(0, most); take(7, most); take(5, most); }
  // round 6 (two segments batched as clips double every row count; the table above was measured at 8 frames): a choice whose LAST round of
  // tiles is mostly empty gives way to a large-tile configuration that fills whole rounds.  Measured (profiles/r06_conv_2clip.txt): 16 frames
  // x 32^2, 640 -> 640: 8 x 32 x 128 tiles = 320 blocks (1.25 rounds) 118.0 us against 16 x 32 x 80 = 256 blocks 85.0 us; 8 frames x 64^2,
  // 128 -> 640 (SPADE): 16 x 32 x 128 = 320 blocks 56.6 us against 16 x 32 x 80 = 512 blocks 46.8 us (profiles/r04_pp_conv.txt).  The small
  // tiles (configurations 1, 5, 7) are not candidates: a full round of them measures no better than a ragged round of a large tile.
  // env MGLD_CONV3R_FILL = 0: the round-4 table alone (A/B runs).
  if (bid >= 0 && p->tune == 0) {
    static int onf = -1;
    if (onf < 0) { const char* e = getenv("MGLD_CONV3R_FILL"); onf = e ? atoi(e) : 1; }
    auto fill = [&](int i) { const int64_t t = tiles_of(i); return (double)t / (double)(cdiv(t, (int64_t)cus) * cus); };
    if (onf && fill(bid) < 0.8) {
      const int cand[4] = {6, 0, 8, 2};
      int alt = -1;
      double fa = 0.95;
      for (int k = 0; k < 4; ++k) {
        const int i = cand[k];
        if (i == bid || !fits(i) || p->Wout < R3_CFG[i].tx || p->Hout < R3_CFG[i].ty || tiles_of(i) < most) continue;
        if (fill(i) >= fa && (alt < 0 || fill(i) > fa)) { fa = fill(i); alt = i; }
      }
      if (alt >= 0) bid = alt;
    }
  }
  // few tiles, deep K (the 16^2 UNet level: 8 frames x 256 pixels x 1280 channels, K = 11520 .. 23040): the channel slices split over
  // grid.z so that tiles x slabs fill the chip once (profiles/r04_conv3r_split.txt)
  if (bid < 0 && p->tune == 0 && w8 && (p->Cin >> 5) >= 8) {      // the 8^2 level (profiles/r04_conv3r_split.txt)
    // env MGLD_CONV3R_W8 = 1: the planner takes them.  Default off: -10 % / -24 % per launch in isolation (34.4 -> 30.6 us at K = 11520,
    // 54.4 -> 41.4 us at K = 23040), but no measurable change of the segment (708.9 vs 709.7 ms, A/B on one box): eight fp32 slabs + the
    // reduction launch eat what the 256-class tiles gain.  tune = 40 / 41 / 59 / 60 select them per launch (tests, tools/igemm_bench.py)
    static int on8 = -1;
    // round 6: at two clips (16 frames, M = 1024) they measure +0.5 .. 0.9 % on the whole pass (13.57 / 13.56 against 13.51 / 13.44 frames/s, two
    // alternated runs each, one box): default on from 16 frames up, off below; the env value forces either way
    if (on8 < 0) { const char* e = getenv("MGLD_CONV3R_W8"); on8 = e ? (atoi(e) ? 1 : 0) : 2; }
    const bool deep = (p->Cin >> 5) >= 64;          // measured: 4-frame tiles win from Cin = 2048 up (46.5 -> 41.4 us at 2560), 2-frame tiles below
    const int cand[2] = {deep ? 9 : 10, deep ? 10 : 9};
    for (int k = 0; k < 2 && bid < 0 && (on8 == 1 || (on8 == 2 && frames >= 16)); ++k) {
      const int i = cand[k];
      if (!fits(i)) continue;
      const int sp = splits_of(i);
      if (sp >= 2 && tiles_of(i) * sp >= most) { bid = i; *splits = sp; }
    }
  }
  if (bid < 0 && p->tune == 0 && !w8 && p->Wout >= 16 && p->Hout >= 8 && (p->Cin >> 5) >= 8) {
    const int cand[3] = {4, 5, 8};                 // 256-pixel tiles x 160 channels x 4 slabs measured best at 8 frames x 16^2 x 1280
    for (int k = 0; k < 3 && bid < 0; ++k) {
      const int i = cand[k];
      if (!fits(i) || p->Wout < R3_CFG[i].tx || p->Hout < R3_CFG[i].ty) continue;
      const int sp = splits_of(i);
      if (sp >= 2 && tiles_of(i) * sp >= most) { bid = i; *splits = sp; }
    }
  }
  if (bid < 0 && p->tune == 30) {
    const int pref[R3_NCFG] = {8, 3, 2, 6, 0, 7, 5, 1, 4, 9, 10};
    for (int k = 0; k < R3_NCFG && bid < 0; ++k) if (fits(pref[k])) bid = pref[k];
  }
  if (bid < 0) return false;
  if (p->tune == 0) {
    static int on = -1;   // env MGLD_CONV3R_AUTO = 0: the planner never takes the family (A/B runs)
    if (on < 0) { const char* e = getenv("MGLD_CONV3R_AUTO"); on = e ? atoi(e) : 1; }
    if (!on) return false;
  }
  *id = bid;
  return true;
}

int dispatch_conv3r(const MgldIGemm* p, hipStream_t s, int id, int splits) {
  switch (id) {
    case 0: return launch_conv3r<8, 32, 160, 4, 2, 6>(p, s, 1);
    case 1: return launch_conv3r<8, 16, 320, 2, 4, 4>(p, s, 1);
    case 2: return launch_conv3r<8, 32, 128, 4, 2, 6>(p, s, 1);
    case 3: return launch_conv3r<8, 32, 256, 2, 4, 5>(p, s, 1);
    case 4: return launch_conv3r<16, 16, 160, 4, 2, 6, true>(p, s, splits);
    case 5: return launch_conv3r<8, 16, 160, 4, 2, 6, true>(p, s, splits);
    case 6: return launch_conv3r<16, 32, 80, 8, 1, 5>(p, s, 1);
    case 7: return launch_conv3r<8, 32, 80, 8, 1, 6>(p, s, 1);
    case 8: return launch_conv3r<16, 32, 128, 8, 1, 5, true>(p, s, splits);
    case 9: return launch_conv3r<32, 8, 160, 4, 2, 5, true>(p, s, splits);
    default: return launch_conv3r<16, 8, 160, 4, 2, 6, true>(p, s, splits);
  }
}

// tiles per frame of configuration id (the row count of MgldIGemm.gn_part per frame)
int conv3r_gn_chunks(const MgldIGemm* p, int id) { return R3_CFG[id].tx == 8 ? 0 : cdiv(p->Hout, R3_CFG[id].ty) * cdiv(p->Wout, R3_CFG[id].tx); }

void conv3r_kernel_name(const MgldIGemm* p, int id, char* buf, int buflen) {
  static const int g[R3_NCFG][6] = {{8, 32, 160, 4, 2, 6}, {8, 16, 320, 2, 4, 4}, {8, 32, 128, 4, 2, 6}, {8, 32, 256, 2, 4, 5}, {16, 16, 160, 4, 2, 6},
                                    {8, 16, 160, 4, 2, 6}, {16, 32, 80, 8, 1, 5}, {8, 32, 80, 8, 1, 6}, {16, 32, 128, 8, 1, 5}, {32, 8, 160, 4, 2, 5},
                                    {16, 8, 160, 4, 2, 6}};
  int id_, splits = 1;
  (void)conv3r_plan(p, &id_, &splits);
  snprintf(buf, buflen, "conv3r_kernel<%d, %d, %d, %d, %d, %d, %s, %s>", g[id][0], g[id][1], g[id][2], g[id][3], g[id][4], g[id][5], p->W2 ? "true" : "false",
           splits > 1 ? "true" : "false");
}
}  // namespace mgld_ig
