// conv3q.hip — the patch-staged 3x3 / stride 1 / pad 1 convolutions of the GEMM family (second translation unit; the launcher in
// igemm.hip routes here through mgld_ig::conv3q_plan / conv3p_plan).  Kernels: conv3p (raster tiles, W <= 64) and conv3q (2-D pixel tiles,
// any size, nearest-2x upsample fold).  See igemm.hip for the common structure (LDS-DMA staging, MFMA 32x32x16, epilogue).
#include "igemm_common.h"

namespace {
using namespace mgld_ig;

// ---- conv3p: 3x3 / stride 1 / pad 1 conv with the activation PATCH staged once per 32 input channels -----------------
// The im2col form above re-fetches every input pixel nine times (once per tap) through the LDS-DMA path, and that path —
// not the matrix pipe — bounds the kernel (ablation: DMA alone ~75 % of the full time, activations the larger share).
// Here a tile of BM consecutive output pixels (inside one frame) stages the contiguous raster range of input pixels
// [m0 - W - 1, m0 + BM + W + 1) ONCE per 32-channel slice (64-B LDS rows) and all nine taps read their fragments from it:
// tap (dy, dx) of output pixel i is patch row i + (dy+1)*W + (dx+1).  Rows outside the frame are zero-filled by the DMA
// (zero page), the x = 0 / x = W-1 wrap of the dx = -1 / +1 taps is removed per lane by redirecting the fragment read to
// a 64-B zero row.  Weights stream as before, one kernel row (3 taps x 32 channels) per stage, double buffered; the next
// patch arrives piecewise during the three stages of the current one.  K order: (32-channel slice, dy, dx, c).
constexpr int PB = 64;   // bytes per LDS row (32 fp16 channels)

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void conv3p_kernel(const MgldIGemm p, float* __restrict__ ws, int hchunk) {
  constexpr int WAVES_N = BN / WN;
  constexpr int NW = (BM / WM) * (BN / WN);
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int BSUB = BN * PB;                 // one tap's weight sub-tile [BN][32 ch]
  constexpr int B_BYTES = 3 * BSUB;             // weight stage: the three taps of one kernel row
  constexpr int NPB = 3 * BN / 16;              // 1-KiB DMA pieces per weight stage
  constexpr int BSLOTS = (NPB + NW - 1) / NW;
  static_assert(NW == 8, "eight waves per block");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // Workgroups are dealt round-robin to the 8 XCDs in linear-id order.  Give each XCD runs of consecutive weight tiles
  // of ONE pixel tile, so the patch is fetched into that XCD's L2 once and the other N/BN - 1 blocks hit it there
  // (-4 % on the 640 -> 320 convs of the 64x64 level, neutral elsewhere).
  int tile_m = blockIdx.x, tile_n = blockIdx.y;
  if ((gridDim.x & 7) == 0) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, j = lin >> 3;
    tile_m = xcd + 8 * (j / (int)gridDim.y);
    tile_n = j % (int)gridDim.y;
  }
  const int bm0 = tile_m * BM, bn0 = tile_n * BN;
  const bool splitk = (ws != nullptr);
  const int kz = splitk ? blockIdx.z : 0;
  const int l31 = lane & 31, lhi = lane >> 5;

  const f16* __restrict__ A = (const f16*)p.A;
  const f16* __restrict__ W = (const f16*)p.W;
  const int N = p.N, Cin = p.Cin, Wd = p.Win, HW = p.Hin * p.Win;
  const int PR = BM + 2 * Wd + 2;               // patch rows
  const int NPA = (PR + 15) >> 4;               // 1-KiB pieces (16 rows) per patch; <= 3 * NW
  const int a_bytes = NPA * 1024;
  const int b_base = 2 * a_bytes;
  const int z_off = b_base + 2 * B_BYTES;       // 64-B zero row
  const int nh = Cin >> 5;
  const int h0 = splitk ? kz * hchunk : 0;
  const int h1 = splitk ? min(nh, h0 + hchunk) : nh;
  const char* zero = (const char*)g_zero_page;

  if (tid < 16) *(unsigned*)(smem + z_off + tid * 4) = 0u;

  // ---- DMA assignment: activation piece q = s*NW + wave (slot s is issued during stage s of the previous slice) ----
  const char* fa_ptr[3];
  unsigned fa_step[3];
  {
    const int frame_lo = (bm0 / HW) * HW, frame_hi = frame_lo + HW;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int j = (s * NW + wave) * 16 + (lane >> 2);
      const int g = bm0 - (Wd + 1) + j;
      const bool ok = (j < PR) && (g >= frame_lo) && (g < frame_hi);
      const int cl = (lane & 3) ^ ((j >> 2) & 3);
      fa_ptr[s] = ok ? (const char*)(A + (int64_t)g * p.lda + h0 * 32 + cl * 8) : zero;
      fa_step[s] = ok ? 64u : 0u;
      if constexpr (ABL & 1) {   // (ablation build bit 1: same byte count from perfectly contiguous addresses — wrong data)
        fa_ptr[s] = (const char*)A + ((int64_t)(bm0 / BM) * 24 + s * NW + wave) * 1024 + lane * 16;
        fa_step[s] = 0u;
      }
    }
  }
  // weight piece b = k*NW + wave: tap column dxi = b / (BN/16), rows (b % (BN/16))*16 .. +16
  const char* fw_ptr[BSLOTS];
  bool fw_ok[BSLOTS];
#pragma unroll
  for (int k = 0; k < BSLOTS; ++k) {
    const int b = k * NW + wave;
    const int dxi = b / (BN / 16), rb = b - dxi * (BN / 16);
    const int row = rb * 16 + (lane >> 2);
    const int n = bn0 + row;
    const int cl = (lane & 3) ^ ((row >> 2) & 3);
    if (p.tap_inner == 2) {   // tiled weights [N/64][Cin/32][3 dy][4 row groups][3 dx][16 rows x 32 ch in LDS-image order]:
      // every DMA piece is one linear 1-KiB read and the 12 pieces of a (64 rows, slice, kernel row) stage are contiguous
      const int g64 = (bn0 >> 6) + (rb >> 2);
      fw_ok[k] = (g64 * 64 < ((N + 63) & ~63)) && (b < NPB);
      fw_ptr[k] = (const char*)(W + (((int64_t)g64 * nh * 3 * 4 + (rb & 3)) * 3 + dxi) * 512 + lane * 8);
    } else {
      fw_ok[k] = (n < N) && (b < NPB);
      const int64_t koff = p.tap_inner ? (int64_t)dxi * 64 : (int64_t)dxi * Cin;
      fw_ptr[k] = (const char*)(W + (int64_t)(fw_ok[k] ? n : 0) * p.ldw + koff + cl * 8);
    }
  }
  auto issue_b = [&](const int buf, const int h, const int dyi) {
    if constexpr (ABL & 16) return;
    const int64_t soff = p.tap_inner == 2 ? (int64_t)(h * 3 + dyi) * (12 * 512)
                         : p.tap_inner    ? ((int64_t)(h >> 1) * 576 + dyi * 192 + (h & 1) * 32)
                                          : ((int64_t)dyi * 3 * Cin + h * 32);
#pragma unroll
    for (int k = 0; k < BSLOTS; ++k) {
      const int b = k * NW + wave;
      if (b < NPB) {
        const char* src = fw_ok[k] ? fw_ptr[k] + soff * 2 : zero;
        if constexpr (ABL & 2)   // (ablation build bit 2: contiguous weight pieces — wrong data)
          src = (const char*)W + ((int64_t)((h * 3 + dyi) * (N / BN) + bn0 / BN) * NPB + b) * 1024 + lane * 16;
        glds16(src, smem + b_base + buf * B_BYTES + b * 1024);
      }
    }
  };
#define MGLD_ISSUE_A(S, PAR)                                                     \
  if ((S) * NW + wave < NPA) {                                                   \
    if constexpr (!(ABL & 8)) glds16(fa_ptr[S], smem + (PAR) * a_bytes + ((S) * NW + wave) * 1024); \
    fa_ptr[S] += fa_step[S];                                                     \
  }

  // ---- fragment addresses (byte offsets from smem) ----
  int a_off[MI][3];     // stage dy = -1, patch buffer 0; -1 = this lane's tap lies across the image edge
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int i = wm * WM + mi * 32 + l31;
    const int x = (bm0 + i) % Wd;
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) {
      const int j = i + dxi;
      const bool ok = !(dxi == 0 && x == 0) && !(dxi == 2 && x == Wd - 1);
      a_off[mi][dxi] = ok ? j * PB + ((lhi ^ ((j >> 2) & 3)) << 4) : -1;
    }
  }
  int w_off[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int r = wn * WN + ni * 32 + l31;
    w_off[ni] = r * PB + ((lhi ^ ((r >> 2) & 3)) << 4);
  }

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  // prologue: the whole first patch + the first weight stage.  (A third weight buffer — two stages in flight behind a
  // counted vmcnt — measured no faster and costs a resident block at W = 32.)
  if (h0 < h1) {
    MGLD_ISSUE_A(0, 0)
    MGLD_ISSUE_A(1, 0)
    MGLD_ISSUE_A(2, 0)
    issue_b(0, h0, 0);
  }
  int cur = 0;
  for (int h = h0; h < h1; ++h) {
    const int pa = (h - h0) & 1;
    const bool more = (h + 1 < h1);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (more) {
        if (s == 0) { MGLD_ISSUE_A(0, pa ^ 1) }
        if (s == 1) { MGLD_ISSUE_A(1, pa ^ 1) }
        if (s == 2) { MGLD_ISSUE_A(2, pa ^ 1) }
      }
      if (s < 2) issue_b(cur ^ 1, h, s + 1);
      else if (more) issue_b(cur ^ 1, h + 1, 0);
      if constexpr (ABL & 128) { cur ^= 1; continue; }
      const int add = s * Wd * PB + pa * a_bytes;            // W % 16 == 0 keeps the swizzle key of a shifted row
      int aaddr[MI][3];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) aaddr[mi][dxi] = a_off[mi][dxi] >= 0 ? a_off[mi][dxi] + add : z_off;
      const int bb = b_base + cur * B_BYTES;
      f16x8 fa[2][MI], fw[2][NI];
      auto load = [&](const int u, const int set) {
        const int dxi = u >> 1, kx = (u & 1) << 5;           // second 16-channel step: logical chunk ^ 2 = byte offset ^ 32
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[set][mi] = *(const f16x8*)(smem + (aaddr[mi][dxi] ^ kx));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fw[set][ni] = *(const f16x8*)(smem + bb + dxi * BSUB + (w_off[ni] ^ kx));
      };
      load(0, 0);
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        if (u + 1 < 6) load(u + 1, (u + 1) & 1);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[u & 1][ni], fa[u & 1][mi], acc[ni][mi], 0, 0, 0);
      }
      cur ^= 1;
    }
  }
#undef MGLD_ISSUE_A
  tile_epilogue<BM, BN, WM, WN>(p, ws, splitk, kz, 0, RowMapLinear{bm0, p.M}, bn0, wm, wn, wave, lane, acc, smem);
}

// ---- conv3q: the patch-staged 3x3 / stride 1 / pad 1 conv on 2-D PIXEL TILES -------------------------------------------
// conv3p above tiles the raster (BM consecutive pixels of one frame): its patch grows with the image width (BM + 2W + 2 rows), so it
// stops at W = 64, and the wrap of the dx = +-1 taps needs per-lane redirects.  Here a block owns a TY x TX pixel tile and stages the
// (TY+2) x (TX+2) input patch (with its own halo columns; out-of-image rows zero-filled by the DMA) once per 32-channel slice: any image
// size (the VAE's 128^2 .. 512^2 levels, non-square frames, ragged edges), and tap (dy, dx) of a lane's pixel is its patch row plus the
// constant dy*PW + dx — nine per-lane byte offsets computed once.  UP2 folds the nearest-2x upsample of the reference's Upsample blocks
// (openaimodel.py:185, model.py:96) into those offsets: the block stages the LOW-resolution (TY/2+2) x (TX/2+2) patch (4x fewer bytes)
// and tap (dy, dx) of output pixel (y, x) reads low-res pixel ((y+dy-1)>>1, (x+dx-1)>>1); zero padding of the upsampled image falls on
// out-of-image low-res pixels.  256-pixel tiles (16x16, 8x32) run eight waves of 64 pixels x 32 channels: 3 fragment reads per 2 MFMAs
// instead of 2 per 1 and half the weight bytes per FLOP of the 128-pixel tiles.  Weights: the tiled layout of tap_inner = 2 only.
// NWB = 3 (round 3): a THIRD weight buffer.  With two, a block has one weight stage in flight while it computes on the other and drains
// `vmcnt(0)` + barrier at every stage: the two blocks of a CU fall into step, both waiting for their DMA, then both computing (PMC round 2:
// matrix pipe busy 37 %, waves parked 44 % on vmcnt / barrier).  With three, stage g + 2 is issued during stage g, the wait at the top of a
// stage is COUNTED (`vmcnt(n)`: only what the stage reads must have landed, the newest weight stage — and the piece of the next slice's
// patch issued with it — stay in flight across the raw `s_barrier`), so a stage's DMA has two stage times to land.  Only where the LDS
// budget keeps the resident blocks per CU (8x32 x 64: 2 x 80 KiB = all 160 KiB; 16x16 x 128: one block either way; 16x16 x 64).
// TWO: the instantiation that runs the weight-residual pass (MgldIGemm.W2) — a template parameter, so the one-pass kernels carry none of it.
template <int TY, int TX, int BN, int WM, int WN, bool UP2, int PF = 1, int NWB = 2, bool TWO = false>
__global__ __launch_bounds__(64 * ((TY * TX) / WM) * (BN / WN)) void conv3q_kernel(const MgldIGemm p, float* __restrict__ ws, int hchunk,
                                                                                  int tiles_x, int tiles_y, int order) {
  constexpr int BM = TY * TX;
  constexpr int WAVES_N = BN / WN;
  constexpr int NW = (BM / WM) * WAVES_N;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int PW = UP2 ? TX / 2 + 2 : TX + 2, PH = UP2 ? TY / 2 + 2 : TY + 2;
  constexpr int PR = PW * PH;                   // patch rows (one 64-B LDS row per input pixel and slice)
  constexpr int NPA = (PR + 15) / 16;           // 1-KiB DMA pieces per patch
  constexpr int ASLOTS = (NPA + NW - 1) / NW;   // pieces per wave; slot s is issued during stage s of the previous slice
  constexpr int A_BYTES = NPA * 1024;
  constexpr int BSUB = BN * PB, B_BYTES = 3 * BSUB, NPB = 3 * BN / 16, BSLOTS = (NPB + NW - 1) / NW;
  constexpr int B_BASE = 2 * A_BYTES;
  static_assert(NW == 4 || NW == 8, "four or eight waves per block");
  static_assert(ASLOTS <= 6, "the patch must arrive within the three stages of a slice (at most two pieces per wave and stage)");
  static_assert(NWB == 2 || NWB == 3, "two or three weight buffers");
  static_assert((TX & (TX - 1)) == 0 && TX >= 8 && (TY % 2) == 0 && BM % WM == 0 && BN % 32 == 0 && BN % WN == 0, "tile shape");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // Workgroups go round-robin to the 8 XCDs in linear-id order (gridDim.x % 8 == 0: the same XCD pattern in every K-split plane).
  // order 0: the N/BN blocks sharing a PATCH run back to back on one XCD (see conv3p) — the activations leave the Infinity Cache
  // once (64x64 level: A >> W).  order 1: every XCD takes a contiguous eighth of the (weight tile major, pixel tile minor) list,
  // i.e. all pixel tiles of ~N/BN/8 weight tiles: the blocks sharing a WEIGHT tile share it through that XCD's L2 instead of
  // every XCD streaming the whole matrix (16x16 / 8x8 levels: W = 30-60 MB against 1-5 MB of activations; PMC showed 3-3.8x the
  // algorithmic bytes there).
  // order bit 8 (env MGLD_CONV3Q_PRIO=1, A/B): static issue priority by hardware wave slot.  The blocks sharing a CU run the same stage
  // structure and fall into lock-step (PMC: waves parked 44 % on the stage barrier / waitcnt, matrix pipe 40 % busy); raising the priority of
  // the waves in the upper wave slots of every SIMD makes one co-resident block the pipe's owner and the other the filler of its gaps.
  if (order & 0x100) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID, bits 3:0 = wave slot on the SIMD
    if (__builtin_amdgcn_readfirstlane(hwid) & 2) __builtin_amdgcn_s_setprio(1);
  }
  order &= 0xff;
  int tile_m = blockIdx.x, tile_n = blockIdx.y;
  if ((gridDim.x & 7) == 0) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, j = lin >> 3;
    if (order == 1) {
      const int q = xcd * ((int)(gridDim.x * gridDim.y) >> 3) + j;
      tile_n = q / (int)gridDim.x;
      tile_m = q - tile_n * (int)gridDim.x;
    } else {
      tile_m = xcd + 8 * (j / (int)gridDim.y);
      tile_n = j % (int)gridDim.y;
    }
  }
  const int tpf = tiles_x * tiles_y;
  const int frame = tile_m / tpf;
  const int trem = tile_m - frame * tpf;
  const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
  const int y0 = tyi * TY, x0 = txi * TX;       // output coordinates of the tile's first pixel
  const int bn0 = tile_n * BN;
  const bool splitk = (ws != nullptr);
  const int kz = splitk ? blockIdx.z : 0;
  const int l31 = lane & 31, lhi = lane >> 5;

  const f16* __restrict__ A = (const f16*)p.A;
  const f16* __restrict__ W = (const f16*)p.W;
  const int N = p.N, Cin = p.Cin, Hin = p.Hin, Win = p.Win;
  const int nh = Cin >> 5;
  const int h0 = splitk ? kz * hchunk : 0;
  const int h1 = splitk ? min(nh, h0 + hchunk) : nh;
  const char* zero = (const char*)g_zero_page;

  // ---- DMA assignment: activation piece q = s*NW + wave = patch rows [16q, 16q+16) ----
  const char* fa_ptr[ASLOTS];
  unsigned fa_step[ASLOTS];
  {
    const int yb = (UP2 ? (y0 >> 1) : y0) - 1, xb = (UP2 ? (x0 >> 1) : x0) - 1;
#pragma unroll
    for (int s = 0; s < ASLOTS; ++s) {
      const int j = (s * NW + wave) * 16 + (lane >> 2);
      const int pr = j / PW, pc = j - pr * PW;
      const int y = yb + pr, x = xb + pc;
      const bool ok = (j < PR) && ((unsigned)y < (unsigned)Hin) && ((unsigned)x < (unsigned)Win);
      const int cl = (lane & 3) ^ ((j >> 2) & 3);
      fa_ptr[s] = ok ? (const char*)(A + (((int64_t)frame * Hin + y) * Win + x) * p.lda + h0 * 32 + cl * 8) : zero;
      fa_step[s] = ok ? 64u : 0u;
    }
  }
  // weight piece b = k*NW + wave: tap column dxi = b / (BN/16), rows (b % (BN/16))*16 .. +16 of the tiled layout
  const char* fw_ptr[BSLOTS];
  bool fw_ok[BSLOTS];
#pragma unroll
  for (int k = 0; k < BSLOTS; ++k) {
    const int b = k * NW + wave;
    const int dxi = b / (BN / 16), rb = b - dxi * (BN / 16);
    const int gr = (bn0 >> 4) + rb;             // 16-row group of the weight matrix (BN = 160 tiles start inside a 64-row group)
    const int g64 = gr >> 2;
    fw_ok[k] = (g64 * 64 < ((N + 63) & ~63)) && (b < NPB);
    fw_ptr[k] = (const char*)(W + (((int64_t)g64 * nh * 3 * 4 + (gr & 3)) * 3 + dxi) * 512 + lane * 8);
  }
  // W2 (MgldIGemm): the slices are walked TWICE — first against the scaled fp16 residual of the weights (same tiled layout, `wdelta` bytes
  // away), then, after ONE multiplication of the accumulators by w2_scale, against the weights themselves.  `lo` selects the matrix.
  constexpr bool two = TWO;
  const int64_t wdelta = two ? (const char*)p.W2 - (const char*)p.W : 0;
  auto issue_b = [&](const int buf, const int h, const int dyi, const bool lo) {
    const int64_t soff = (int64_t)(h * 3 + dyi) * (12 * 512) * 2 + (lo ? wdelta : 0);
#pragma unroll
    for (int k = 0; k < BSLOTS; ++k) {
      const int b = k * NW + wave;
      if (b < NPB) {
        const char* src = fw_ok[k] ? fw_ptr[k] + soff : zero;
        glds16(src, smem + B_BASE + buf * B_BYTES + b * 1024);
      }
    }
  };
#define MGLD_Q_ISSUE_A(S, PAR)                                                    \
  if constexpr ((S) < ASLOTS) {                                                   \
    if ((S) * NW + wave < NPA) {                                                  \
      glds16(fa_ptr[S], smem + (PAR) * A_BYTES + ((S) * NW + wave) * 1024);       \
      fa_ptr[S] += fa_step[S];                                                    \
    }                                                                             \
  }

  // ---- fragment addresses: byte offset of this lane's patch row for each of the nine taps (first 16-channel step) ----
  int a_off[MI][9];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int r = wm * WM + mi * 32 + l31;
    const int ty = r / TX, tx = r & (TX - 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dyi = t / 3, dxi = t - dyi * 3;
      const int j = UP2 ? (((ty + dyi - 1) >> 1) + 1) * PW + ((tx + dxi - 1) >> 1) + 1 : (ty + dyi) * PW + tx + dxi;
      a_off[mi][t] = j * PB + ((lhi ^ ((j >> 2) & 3)) << 4);
    }
  }
  int w_off[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int r = wn * WN + ni * 32 + l31;
    w_off[ni] = r * PB + ((lhi ^ ((r >> 2) & 3)) << 4);
  }

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  const int ns = h1 - h0;                     // channel slices of this block (of this K split)
  const int nv = two ? 2 * ns : ns;           // slice visits: residual pass, then main pass
  // weight-stage issue cursor: stage (visit iv, kernel row idy) of slice ih goes into ring buffer ibuf (plain scalars in this scope)
  int ih = h0, idy = 0, iv = 0, ibuf = 0;
#define MGLD_Q_ISSUE_W()                                                          \
  if (iv < nv) {                                                                  \
    issue_b(ibuf, ih, idy, two && iv < ns);                                       \
    ibuf = (ibuf + 1 == NWB) ? 0 : ibuf + 1;                                      \
    if (++idy == 3) { idy = 0; ++iv; ih = (two && iv == ns) ? h0 : ih + 1; }      \
  }
  // DMA instructions this wave issues per weight stage / per patch slot (the counted waits of the three-buffer ring)
  int nw_me = 0, na_me[3] = {0, 0, 0};       // (patch pieces of stage s: slots s and s + 3)
#pragma unroll
  for (int k = 0; k < BSLOTS; ++k) nw_me += (k * NW + wave < NPB) ? 1 : 0;
#pragma unroll
  for (int s = 0; s < ASLOTS; ++s) na_me[s % 3] += (s * NW + wave < NPA) ? 1 : 0;
  if (h0 < h1) {
    MGLD_Q_ISSUE_A(0, 0)
    MGLD_Q_ISSUE_A(1, 0)
    MGLD_Q_ISSUE_A(2, 0)
    MGLD_Q_ISSUE_A(3, 0)
    MGLD_Q_ISSUE_A(4, 0)
    MGLD_Q_ISSUE_A(5, 0)
    if constexpr (NWB == 3) { MGLD_Q_ISSUE_W() MGLD_Q_ISSUE_W() }
    else issue_b(0, h0, 0, two);
  }
  int cur = 0;
  int h = h0;
  for (int v = 0; v < nv; ++v) {
    const int pa = v & 1;
    const bool more = (v + 1 < nv);
    const bool wrap = two && (v + 1 == ns);   // the next visit starts the main pass: its patch is slice h0 again
    const int hn = wrap ? h0 : h + 1;
    const bool lo = two && (v < ns), lo_next = two && (v + 1 < ns);
    if (two && v == ns) {                     // residual pass done: acc = w2_scale * (A W2^T), then A W^T accumulates on top
      const float sc2 = p.w2_scale;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ni][mi][r] *= sc2;
    }
    if (wrap) {
#pragma unroll
      for (int s = 0; s < ASLOTS; ++s) fa_ptr[s] -= (int64_t)ns * fa_step[s];
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if constexpr (NWB == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      } else {
        // in flight, oldest first: ..., W[g] (issued two stages ago), then last stage's issues: [patch slot s-1 of the NEXT slice, W[g+1]].
        // This stage reads W[g] and (s == 0) the whole patch, whose last slot is one of last stage's issues: allow W[g+1], and
        // for s != 0 also last stage's patch piece, to stay in flight.
        int allow = (3 * v + s + 1 < 3 * nv) ? nw_me : 0;
        if (s != 0 && more) allow += na_me[s - 1];
        switch (allow) {
          case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
          case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
          case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
          case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
          case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
          case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
          case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
          case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
          case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;     // (8 weight + 2 patch pieces: the 160-row variant)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // raw barrier: __syncthreads() would drain the DMA queue (vmcnt(0)) first
      }
      if (more) {     // (four-wave blocks with large patches carry two pieces per wave and stage: slots s and s + 3)
        if (s == 0) { MGLD_Q_ISSUE_A(0, pa ^ 1) MGLD_Q_ISSUE_A(3, pa ^ 1) }
        if (s == 1) { MGLD_Q_ISSUE_A(1, pa ^ 1) MGLD_Q_ISSUE_A(4, pa ^ 1) }
        if (s == 2) { MGLD_Q_ISSUE_A(2, pa ^ 1) MGLD_Q_ISSUE_A(5, pa ^ 1) }
      }
      if constexpr (NWB == 3) { MGLD_Q_ISSUE_W() }      // the stage after the next, into the buffer read last stage
      else {                                            // the next stage (kernel row s + 1, or row 0 of the next slice): compile-time row
        if (s < 2) issue_b(cur ^ 1, h, s + 1, lo);
        else if (more) issue_b(cur ^ 1, hn, 0, lo_next);
      }
      const int abase = pa * A_BYTES;
      const int bb = B_BASE + cur * B_BYTES;
      // fragments of step u + PF are fetched from LDS while the MFMAs of step u run (PF + 1 register sets, static indices);
      // PF = 2 gives a ds_read_b128 two MFMA groups (~128 issue cycles) instead of one to land
      f16x8 fa[PF + 1][MI], fw[PF + 1][NI];
      auto load = [&](const int u, const int set) {
        const int dxi = u >> 1, kx = (u & 1) << 5;           // second 16-channel step: logical chunk ^ 2 = byte offset ^ 32
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[set][mi] = *(const f16x8*)(smem + abase + (a_off[mi][s * 3 + dxi] ^ kx));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fw[set][ni] = *(const f16x8*)(smem + bb + dxi * BSUB + (w_off[ni] ^ kx));
      };
#pragma unroll
      for (int u = 0; u < PF; ++u) load(u, u);
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        if (u + PF < 6) load(u + PF, (u + PF) % (PF + 1));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[u % (PF + 1)][ni], fa[u % (PF + 1)][mi], acc[ni][mi], 0, 0, 0);
      }
      if constexpr (NWB == 2) cur ^= 1;
      else cur = (cur + 1 == NWB) ? 0 : cur + 1;
    }
    h = hn;
  }
#undef MGLD_Q_ISSUE_A
#undef MGLD_Q_ISSUE_W
  tile_epilogue<BM, BN, WM, WN>(p, ws, splitk, kz, 0, RowMap2D<TX>{frame * p.Hout * p.Wout, y0, x0, p.Hout, p.Wout}, bn0, wm, wn, wave,
                                lane, acc, smem);
}

}  // namespace

namespace mgld_ig {
// ---- conv3p launch plan -------------------------------------------------------------------------------------------
constexpr int C3P_BM = 128;
inline int conv3p_lds(int Win, int BN) { return 2 * ((C3P_BM + 2 * Win + 2 + 15) >> 4) * 1024 + 2 * 3 * BN * PB + 64; }

// true when the problem takes the patch kernel; *bn = weight tile rows, *splits / *hchunk = K split in 32-channel slices
bool conv3p_plan(const MgldIGemm* p, int* bn, int* splits, int* hchunk) {
  static int knob = -1, fsplit = -1;   // env MGLD_CONV3P: 0 = off, 64 / 128 = force the weight tile; MGLD_CONV3P_SPLITS (tuning)
  if (knob < 0) { const char* e = getenv("MGLD_CONV3P"); knob = e ? atoi(e) : 1; }
  if (fsplit < 0) { const char* e = getenv("MGLD_CONV3P_SPLITS"); fsplit = e ? atoi(e) : 0; }
  if (!knob || p->mode != MGLD_MODE_CONV3X3) return false;
  if (p->kh > 0 && !(p->kh == 3 && p->kw == 3)) return false;
  if (p->stride != 1 || p->up2 || p->pad_t != 1 || p->pad_l != 1 || p->Hin != p->Hout || p->Win != p->Wout) return false;
  if ((p->Win & 15) || p->Win > 64 || (p->Hin * p->Win) % C3P_BM) return false;
  if ((p->Cin & 31) || (p->tap_inner == 1 && (p->Cin & 63)) || p->batch > 1 || p->N <= 32 || p->act == MGLD_ACT_GEGLU) return false;
  const int N = p->N;
  // 64 weight rows: W = 64 (a 128-row stage pair would not leave LDS for two blocks per CU), W = 32 (three blocks per CU
  // instead of two: measured faster), and N = 64 (mod 128); 128 rows at W = 16
  int BN = (N <= 64 || ((N & 127) == 64 && N <= 448) || p->Win >= 32) ? 64 : 128;
  if (knob == 64 || knob == 128) BN = knob;
  const int lds = conv3p_lds(p->Win, BN);
  if (lds > 160 * 1024) return false;
  const int64_t tiles = (int64_t)(p->M / C3P_BM) * cdiv(N, BN);
  const int slots = num_cus() * ((160 * 1024) / lds);
  const int nh = p->Cin >> 5;
  int s = fsplit > 0 ? fsplit : (int)(slots / tiles);
  if (s > nh / 4) s = nh / 4;
  if (s > 16) s = 16;
  if (s < 2 || g_ws == nullptr || (size_t)s * p->M * N * sizeof(float) > g_ws_bytes) s = 1;
  int hc = (nh + s - 1) / s;
  s = (nh + hc - 1) / hc;
  *bn = BN; *splits = s; *hchunk = hc;
  return true;
}

template <int BN, int WM, int WN>
int launch_conv3p(const MgldIGemm* p, hipStream_t s, int splits, int hchunk) {
  const int lds = conv3p_lds(p->Win, BN);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv3p_kernel<C3P_BM, BN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  dim3 grid(p->M / C3P_BM, cdiv(p->N, BN), splits > 1 ? splits : 1);
  hipLaunchKernelGGL((conv3p_kernel<C3P_BM, BN, WM, WN>), grid, dim3(512), lds, s, *p, splits > 1 ? g_ws : nullptr, hchunk);
  if (splits > 1) launch_splitk_reduce(p, s, splits);
  return mgld_check_launch("igemm(conv3p)");
}


// ---- conv3q launch plan -------------------------------------------------------------------------------------------
// variants (id): tile TY x TX pixels, BN weight rows, wave tile WM pixels x WN channels
//   0: 8x16 x 64, 32x32 (8 waves)      1: 16x16 x 64, 64x32 (8 waves)     2: 8x16 x 128, 64x32 (8 waves)
//   3: 16x16 x 128, 64x64 (8 waves)    4: 8x32 x 64, 64x32 (8 waves)      5: 8x16 x 64, 64x32 (4 waves)
//   6: 8x8 x 128, 32x32 (8 waves): the 8x8 UNet level, one tile per frame
//   7: 8x32 x 64, 64x64 (4 waves; with up2: 16x16 x 64, 64x64)    (round 3: 2 x 2 MFMA tiles per wave = 1 KiB of LDS fragment reads per
//      MFMA instead of 1.5: the 64x32 wave tiles run at the LDS read bandwidth)
//   (measured and removed, profiles/r03_conv3q_variants.txt: 8x32 x 128 with eight 64x64 waves — one block per CU, no better than 7 — and
//    8x32 x 160 with four 64x160 waves, one block per CU = exactly 256 blocks for the N = 320 convolutions of the 64x64 level: 9-22 % SLOWER
//    than two blocks of variant 7 per CU; the template still takes any BN % 32 == 0, WN = BN)
// (fragments prefetched TWO steps ahead — template parameter PF = 2 — measured identical to PF = 1 on every shape: not instantiated)
constexpr int Q3_NVAR = 8;
template <int TY, int TX, int BN, int WM, int WN, bool UP2, int NWB = 2>
constexpr int conv3q_lds() {
  constexpr int PW = UP2 ? TX / 2 + 2 : TX + 2, PH = UP2 ? TY / 2 + 2 : TY + 2;
  constexpr int NPA = (PW * PH + 15) / 16, NW = (TY * TX / WM) * (BN / WN);
  constexpr int stages = 2 * NPA * 1024 + NWB * 3 * BN * PB, epi = NW * 32 * (WN + 4) * 4;
  return stages > epi ? stages : epi;
}
// weight buffers of variant `id`: env MGLD_CONV3Q_NWB = 3 selects the three-buffer ring for the variants whose LDS budget keeps the
// resident blocks per CU (see the kernel); default two
inline int q3_nwb(int id, bool up2) {
  static int force = -1;
  if (force < 0) { const char* e = getenv("MGLD_CONV3Q_NWB"); force = e ? atoi(e) : 0; }
  // measured (profiles/r03_conv3q_nwb.txt): the third buffer is 1-3 % slower launch by launch, 0.5 % end to end -> opt-in only
  if (force != 3 || up2) return 2;
  return (id == 1 || id == 3 || id == 4) ? 3 : 2;
}
// nearest-2x fold with four 64x64 waves per 16x16 tile (variant 7 with up2): env MGLD_CONV3Q_UP2W64 = 0 / 1 (A/B)
inline bool q3_up2_wave64() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MGLD_CONV3Q_UP2W64"); v = e ? atoi(e) : 1; }   // measured -7..-8 % on every up-conv (profiles/r03_conv3q_variants.txt)
  return v != 0;
}
inline void q3_geom(int id, int* ty, int* tx, int* bn, int* lds, bool up2) {
  const bool w3 = q3_nwb(id, up2) == 3;
  if (up2 && id == 7) { *ty = 16; *tx = 16; *bn = 64; *lds = conv3q_lds<16, 16, 64, 64, 64, true>(); return; }
  switch (id) {
    case 1: *ty = 16; *tx = 16; *bn = 64; *lds = up2 ? conv3q_lds<16, 16, 64, 64, 32, true>() : (w3 ? conv3q_lds<16, 16, 64, 64, 32, false, 3>() : conv3q_lds<16, 16, 64, 64, 32, false>()); break;
    case 2: *ty = 8; *tx = 16; *bn = 128; *lds = conv3q_lds<8, 16, 128, 64, 32, false>(); break;
    case 3: *ty = 16; *tx = 16; *bn = 128; *lds = w3 ? conv3q_lds<16, 16, 128, 64, 64, false, 3>() : conv3q_lds<16, 16, 128, 64, 64, false>(); break;
    case 4: *ty = 8; *tx = 32; *bn = 64; *lds = w3 ? conv3q_lds<8, 32, 64, 64, 32, false, 3>() : conv3q_lds<8, 32, 64, 64, 32, false>(); break;
    case 5: *ty = 8; *tx = 16; *bn = 64; *lds = conv3q_lds<8, 16, 64, 64, 32, false>(); break;
    case 6: *ty = 8; *tx = 8; *bn = 128; *lds = conv3q_lds<8, 8, 128, 32, 32, false>(); break;
    case 7: *ty = 8; *tx = 32; *bn = 64; *lds = conv3q_lds<8, 32, 64, 64, 64, false>(); break;
    default: *ty = 8; *tx = 16; *bn = 64; *lds = up2 ? conv3q_lds<8, 16, 64, 32, 32, true>() : conv3q_lds<8, 16, 64, 32, 32, false>(); break;
  }
}

// true when the problem takes the 2-D-tile patch kernel (tiled weights, tap_inner = 2); *id = variant, *splits / *hchunk = K split
bool conv3q_plan(const MgldIGemm* p, int* id, int* splits, int* hchunk) {
  static int knob = -1, force = -2, fsplit = -1;   // env MGLD_CONV3Q=0: off; MGLD_CONV3Q_FORCE=<id>; MGLD_CONV3P_SPLITS (tuning)
  if (knob < 0) { const char* e = getenv("MGLD_CONV3Q"); knob = e ? atoi(e) : 1; }
  if (force < -1) { const char* e = getenv("MGLD_CONV3Q_FORCE"); force = e ? atoi(e) : -1; }
  if (fsplit < 0) { const char* e = getenv("MGLD_CONV3P_SPLITS"); fsplit = e ? atoi(e) : 0; }
  if (!knob || p->mode != MGLD_MODE_CONV3X3 || p->tap_inner != 2) return false;
  if (p->kh > 0 && !(p->kh == 3 && p->kw == 3)) return false;
  if (p->stride != 1 || p->pad_t != 1 || p->pad_l != 1 || p->batch > 1 || p->N <= 32 || p->act == MGLD_ACT_GEGLU || (p->Cin & 31)) return false;
  const int sc = p->up2 ? 2 : 1;
  if (p->Hout != sc * p->Hin || p->Wout != sc * p->Win || p->Wout < 8 || p->Hout < 8 || (p->M % (p->Hout * p->Wout))) return false;
  if (p->Wout < 16 && (p->up2 || p->Wout != 8 || p->Hout != 8)) return false;     // below 16 pixels: only the 8x8 level
  const int frames = p->M / (p->Hout * p->Wout), N = p->N, nh = p->Cin >> 5;
  // variant by measurement (tools/igemm_bench.py on MI355X, cold operands; profiles/r02_conv3q_variants.txt):
  //   nearest-2x fold: 16x16 tiles (low-res patch 10x10) while they give ~2 blocks per CU (209 vs 241 us on 640 -> 640 at 32 -> 64), else 8x16;
  //   16x16 frames with N % 128 == 0: one 16x16 tile = the whole frame, 128 weight rows, 64x64 wave tiles;
  //   W >= 32: 8x32 tiles (256 pixels, conflict-free fragment reads) while they still give ~2 blocks per CU, else 8x16 tiles run by
  //   four waves of 64 pixels x 32 channels (3 blocks per CU).
  const int64_t t832 = (int64_t)frames * cdiv(p->Hout, 8) * cdiv(p->Wout, 32) * cdiv(N, 64);
  const int64_t t256 = (int64_t)frames * cdiv(p->Hout, 16) * cdiv(p->Wout, 16) * cdiv(N, 64);
  int v;
  if (p->up2) v = (t256 >= 448) ? (q3_up2_wave64() ? 7 : 1) : 0;
  else if (p->Wout == 8) v = 6;
  else if (p->Wout == 16 && p->Hout == 16 && (N & 127) == 0) v = 3;
  // (round 3, profiles/r03_conv3q_variants.txt: variant 7 — 8x32 tiles run by FOUR waves of 64 pixels x 64 channels, a third less LDS
  // fragment traffic per MFMA — is 5-7 % SLOWER than variant 4 on the UNet's 64x64 shapes and 2-4 % faster on three VAE shapes: not picked;
  // p->tune = 8 / MGLD_CONV3Q_FORCE=7 select it)
  else v = (p->Wout >= 32 && t832 >= 448) ? 4 : 5;
  if (force >= 0 && force < Q3_NVAR && !(p->up2 && force > 1 && force != 7) && p->Wout >= 16 && force != 6) v = force;
  if (p->tune > 0 && p->tune <= Q3_NVAR && !(p->up2 && p->tune > 2 && p->tune != 8) && p->Wout >= 16 && p->tune != 7) v = p->tune - 1;
  int ty, tx, bn, lds;
  q3_geom(v, &ty, &tx, &bn, &lds, p->up2 != 0);
  const int64_t tiles = (int64_t)frames * cdiv(p->Hout, ty) * cdiv(p->Wout, tx) * cdiv(N, bn);
  const int slots = num_cus() * ((160 * 1024) / lds);
  int s = fsplit > 0 ? fsplit : (int)(slots / tiles);
  if (s > nh / 4) s = nh / 4;
  if (s > 16) s = 16;
  if (s < 2 || g_ws == nullptr || (size_t)s * p->M * N * sizeof(float) > g_ws_bytes) s = 1;
  int hc = (nh + s - 1) / s;
  s = (nh + hc - 1) / hc;
  *id = v; *splits = s; *hchunk = hc;
  return true;
}

template <int TY, int TX, int BN, int WM, int WN, bool UP2, int PF = 1, int NWB = 2, bool TWO = false>
int launch_conv3q_(const MgldIGemm* p, hipStream_t s, int splits, int hchunk) {
  constexpr int lds = conv3q_lds<TY, TX, BN, WM, WN, UP2, NWB>();
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv3q_kernel<TY, TX, BN, WM, WN, UP2, PF, NWB, TWO>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  const int frames = p->M / (p->Hout * p->Wout);
  const int tiles_x = cdiv(p->Wout, TX), tiles_y = cdiv(p->Hout, TY);
  dim3 grid(frames * tiles_x * tiles_y, cdiv(p->N, BN), splits > 1 ? splits : 1);
  constexpr int THREADS = 64 * (TY * TX / WM) * (BN / WN);
  // tile order: share the weight tiles per XCD where the weights outweigh the activations (env MGLD_CONV3Q_ORDER = 0 / 1 forces)
  static int forder = -2;
  if (forder == -2) { const char* e = getenv("MGLD_CONV3Q_ORDER"); forder = e ? atoi(e) : -1; }
  const double wbytes = 2.0 * p->N * p->K, abytes = 2.0 * p->M * p->Cin / (UP2 ? 4 : 1);
  static int prio = -1;
  if (prio < 0) { const char* e = getenv("MGLD_CONV3Q_PRIO"); prio = e ? atoi(e) : 0; }
  const int order = (forder >= 0 ? forder : (wbytes > 2.0 * abytes ? 1 : 0)) | (prio ? 0x100 : 0);
  hipLaunchKernelGGL((conv3q_kernel<TY, TX, BN, WM, WN, UP2, PF, NWB, TWO>), grid, dim3(THREADS), lds, s, *p, splits > 1 ? g_ws : nullptr, hchunk,
                     tiles_x, tiles_y, order);
  if (splits > 1) launch_splitk_reduce(p, s, splits);
  return mgld_check_launch("igemm(conv3q)");
}

// weight-residual pass (W2): the two-buffer instantiation with TWO = true
template <int TY, int TX, int BN, int WM, int WN, bool UP2, int PF = 1, int NWB = 2>
int launch_conv3q(const MgldIGemm* p, hipStream_t s, int splits, int hchunk) {
  if (p->W2) return launch_conv3q_<TY, TX, BN, WM, WN, UP2, PF, 2, true>(p, s, splits, hchunk);
  return launch_conv3q_<TY, TX, BN, WM, WN, UP2, PF, NWB, false>(p, s, splits, hchunk);
}

int dispatch_conv3q(const MgldIGemm* p, hipStream_t s, int id, int splits, int hchunk) {
  if (p->up2) {
    if (id == 7) return launch_conv3q<16, 16, 64, 64, 64, true>(p, s, splits, hchunk);
    return id == 1 ? launch_conv3q<16, 16, 64, 64, 32, true>(p, s, splits, hchunk) : launch_conv3q<8, 16, 64, 32, 32, true>(p, s, splits, hchunk);
  }
  const bool w3 = q3_nwb(id, false) == 3 && !p->W2;
  switch (id) {
    case 1: return w3 ? launch_conv3q<16, 16, 64, 64, 32, false, 1, 3>(p, s, splits, hchunk) : launch_conv3q<16, 16, 64, 64, 32, false>(p, s, splits, hchunk);
    case 2: return launch_conv3q<8, 16, 128, 64, 32, false>(p, s, splits, hchunk);
    case 3: return w3 ? launch_conv3q<16, 16, 128, 64, 64, false, 1, 3>(p, s, splits, hchunk) : launch_conv3q<16, 16, 128, 64, 64, false>(p, s, splits, hchunk);
    case 4: return w3 ? launch_conv3q<8, 32, 64, 64, 32, false, 1, 3>(p, s, splits, hchunk) : launch_conv3q<8, 32, 64, 64, 32, false>(p, s, splits, hchunk);
    case 5: return launch_conv3q<8, 16, 64, 64, 32, false>(p, s, splits, hchunk);
    case 6: return launch_conv3q<8, 8, 128, 32, 32, false>(p, s, splits, hchunk);
    case 7: return launch_conv3q<8, 32, 64, 64, 64, false>(p, s, splits, hchunk);
    default: return launch_conv3q<8, 16, 64, 32, 32, false>(p, s, splits, hchunk);
  }
}

int dispatch_conv3p(const MgldIGemm* p, hipStream_t s, int bn, int splits, int hchunk) {
  return bn == 64 ? launch_conv3p<64, 32, 32>(p, s, splits, hchunk) : launch_conv3p<128, 64, 32>(p, s, splits, hchunk);
}

// name of the conv3q instantiation variant `id` runs, spelled as rocprofv3 prints it
void conv3q_kernel_name(const MgldIGemm* p, int id, char* buf, int buflen) {
  static const int g[Q3_NVAR][5] = {{8, 16, 64, 32, 32}, {16, 16, 64, 64, 32}, {8, 16, 128, 64, 32}, {16, 16, 128, 64, 64}, {8, 32, 64, 64, 32}, {8, 16, 64, 64, 32}, {8, 8, 128, 32, 32},
                                    {8, 32, 64, 64, 64}};
  const char* two = p->W2 ? "true" : "false";
  if (p->up2 && id == 7) snprintf(buf, buflen, "conv3q_kernel<16, 16, 64, 64, 64, true, 1, 2, %s>", two);
  else
    snprintf(buf, buflen, "conv3q_kernel<%d, %d, %d, %d, %d, %s, %d, %d, %s>", g[id][0], g[id][1], g[id][2], g[id][3], g[id][4],
             p->up2 ? "true" : "false", 1, p->W2 ? 2 : q3_nwb(id, p->up2 != 0), two);
}
}  // namespace mgld_ig
