/*
 * mgld_hip.h — C ABI of libmgld_hip.so, the gfx950 (MI355X / CDNA4) kernel library behind the
 * MGLD-VSR per-segment inference hot path.
 *
 * The reference is pure Python: its "FFI" for this path is the set of vendor kernels it reaches through
 * PyTorch / xformers.  Each entry point below replaces one of those call sites (cited per function as
 * reference file:line, paths relative to the reference tree).  Conventions:
 *   - plain pointers + sizes only; no torch types.  All pointers are DEVICE pointers borrowed for the call.
 *   - every launcher takes the hipStream_t (as void*) to enqueue on; no host sync, no allocation inside,
 *     so every call is hipGraph-capturable.
 *   - return 0 on success, negative MGLD_E_* on error (host side raises RuntimeError).
 *   - activations are NHWC ("tokens x channels") fp16 matrices with an explicit leading dimension `ld`
 *     (elements), so channel-concatenation is done by producers writing into slices of a wider buffer.
 *   - accumulation, normalisation statistics, softmax and the DDPM/guidance arithmetic are fp32.
 */
#ifndef MGLD_HIP_H
#define MGLD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGLD_OK 0
#define MGLD_E_ARG (-1)      /* bad argument (shape / alignment / unsupported combination) */
#define MGLD_E_LAUNCH (-2)   /* hipLaunch / runtime error */
#define MGLD_E_UNSUPPORTED (-3)

/* ---- runtime ------------------------------------------------------------------------------------------- */
int mgld_version(void);
/* last HIP error string seen by the library (static storage) */
const char* mgld_last_error(void);
/* device properties: out[0]=CU count, out[1]=LDS bytes/CU, out[2]=clock kHz, out[3]=gcn arch number (950) */
int mgld_device_info(int device, int64_t* out4);
/* stream-capture helpers (hipStreamBeginCapture / EndCapture / GraphInstantiate / GraphLaunch) */
int mgld_graph_begin(void* stream);
int mgld_graph_end(void* stream, void** graph_exec_out);
int mgld_graph_launch(void* graph_exec, void* stream);
int mgld_graph_destroy(void* graph_exec);
/* hipEvent timing on an arbitrary stream (bench.py: roofline.achieved is measured with these) */
int mgld_event_create(void** ev_out);
int mgld_event_record(void* ev, void* stream);
int mgld_event_sync(void* ev);
int mgld_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out);
int mgld_event_destroy(void* ev);

/* ---- K1/K2/K4/K8/K13: implicit-GEMM on MFMA ---------------------------------------------------------------
 * C[M,N] = epilogue( A_gather[M,K] * W[N,K]^T ), fp16 operands, fp32 accumulate (v_mfma_f32_32x32x16_f16).
 * Replaces: nn.Conv2d 3x3 / 1x1 (openaimodel.py:176,221,271-304,401-436; spade.py:83-88; model.py:84-183;
 * rrdbnet_arch.py:9-38), nn.Linear (attention.py:51,64,71,323-330,510,524; openaimodel.py:2021-2025),
 * nn.Conv1d 1x1 (openaimodel.py:515-519), nn.Conv3d (3,1,1) (diffusionmodules/util.py:298).
 */
enum {
  MGLD_MODE_LINEAR = 0,  /* A row m = A + m*lda                                             */
  MGLD_MODE_CONV3X3 = 1, /* m=(n,oy,ox); K = 9*Cin; tap (ky,kx) reads pixel (oy*s+ky-pad_t, ox*s+kx-pad_l);
                            zero outside; optional nearest-2x upsample folded into the gather */
  MGLD_MODE_TCONV3 = 2   /* m=(clip,t,p); K = 3*Cin; tap dt reads frame t+dt-1 (zero outside [0,T)) */
};
enum {
  MGLD_ACT_NONE = 0,
  MGLD_ACT_RELU = 1,
  MGLD_ACT_LRELU02 = 2,
  MGLD_ACT_SILU = 3,
  MGLD_ACT_GEGLU = 4, /* W rows packed [32 value | 32 gate] per 64; output has N/2 columns: v*gelu(g) */
  MGLD_ACT_SIGMOID = 5,
  MGLD_ACT_TANH = 6,
  MGLD_ACT_GELU = 7   /* exact (erf) GELU: the text transformer's MLP (open_clip nn.GELU) */
};

typedef struct MgldIGemm {
  const void* A;      /* fp16 activations                                                   */
  const void* W;      /* fp16 weights [N, ldw], K contiguous ([Cout][tap][Cin] for conv)    */
  void* C;            /* fp16 (or fp32 if out_f32) output [M, ldc]                          */
  const float* bias;  /* [N] or NULL                                                        */
  const float* bias_m;/* [M] per-row bias or NULL (transposed projections)                  */
  const float* rowvec;/* [M / rows_per_frame, ld_rowvec] per-frame per-channel add, or NULL (time-embedding add) */
  const void* R;      /* fp16 residual [M, ldr] or NULL                                     */
  int32_t M, N, K;
  int32_t lda, ldw, ldc, ldr, ld_rowvec;
  int32_t mode;
  int32_t Cin;            /* channels per tap (K = taps*Cin); Cin % 8 == 0                  */
  int32_t Hin, Win;       /* stored input image size                                        */
  int32_t Hout, Wout;     /* output image size                                              */
  int32_t stride;         /* 1 or 2                                                         */
  int32_t pad_t, pad_l;   /* top/left zero padding (bottom/right implied by Hout/Wout)      */
  int32_t up2;            /* 1: gather from a virtual nearest-2x-upsampled input            */
  int32_t T, HW;          /* TCONV3: frames per clip, pixels per frame                      */
  int32_t rows_per_frame; /* for rowvec                                                     */
  int32_t act;
  int32_t out_f32;
  float alpha, beta;      /* out = alpha*act(acc+bias+rowvec) + beta*R                      */
  int32_t batch;          /* grid.z batches (>=1)                                           */
  int32_t tap_inner;      /* CONV3X3/TCONV3 with Cin % 64 == 0 and no upsample fold: 1 = the K axis of W is ordered
                             (64-channel block, tap, channel) instead of (tap, Cin): all taps of one channel block are
                             consumed back to back, so the shifted re-reads of the input hit L1/L2.
                             2 = CONV3X3 problems a patch-staged kernel takes (3x3, stride 1, pad 1, output = input size or
                             twice it with up2, Wout >= 16, Hout >= 8, Cin % 32 == 0, N > 32, no batch; query with
                             mgld_igemm_config on a struct with tap_inner = 2: code % 1000000 >= 300000): W is tiled [ceil(N/64)][Cin/32][3 kernel rows]
                             [4 groups of 16 rows][3 kernel columns][16 rows][32 channels] (rows past N zero; within a
                             16x32 tile the 16-byte slot c of row r holds channels 8*(c ^ ((r>>2)&3)) .. +8, the LDS
                             image), so each 1-KiB DMA piece is one linear read; ldw unused */
  int64_t strideA, strideW, strideC, strideR; /* element strides between batches            */
  int32_t t_off;          /* TCONV3 on a frame-sharded clip: output frame f sits at position f + t_off of a clip of T
                             frames whose rows start t_off frames BEFORE `A` (A points at the first output frame inside a
                             halo-extended buffer); M = (#output frames)*HW.  0 = whole clips (M % (T*HW) == 0).       */
  int32_t kh, kw;         /* CONV3X3 mode with a general kernel: kh x kw taps (1..15 each), K = kh*kw*Cin, W laid out
                             [Cout][ky][kx][Cin]; 0,0 (or 3,3) = the 3x3 kernel.  Other sizes use the per-lane gather path
                             (RAFT's 7x7, 1x5, 5x1 and strided 1x1 convolutions, raft_arch.py:216,383-389,430).         */
  int32_t tune;           /* 0 = the launcher picks the kernel variant.  > 0 (tuning runs / variant tests): force variant
                             tune - 1 of the 2-D-tile patch conv where it applies (ids: csrc/igemm.hip "conv3q launch plan");
                             TCONV3: 15 = tiles of consecutive rows, 14 = the frame-interleaved row order whatever the frame size (bit-identical). */
  float w2_scale;         /* see W2 */
  const void* W2;         /* NULL, or the fp16 ROUNDING RESIDUAL of the weights in the layout / strides of W:
                             W2 = fp16((W_fp32 - fp32(W)) / w2_scale), w2_scale a power of two (2^-11 keeps the residual in fp16's normal
                             range).  The kernel then computes acc = w2_scale * (A W2^T) + A W^T in ONE launch (the K loop runs twice over
                             the same A: residual pass first, accumulators scaled once, main pass): the product sees weights exact to
                             ~2^-21 instead of 2^-11, at twice the MFMA work and no extra activation traffic beyond L2.  For the layers
                             whose fp16 weight rounding dominates the output error.  Not with split-fp32 outputs of the
                             raster patch kernel (conv3p).                                                                                  */
  float* gn_part;         /* NULL, or where the kernel ALSO writes the GroupNorm statistics of its output: per output tile and channel
                             the (sum, sumsq) of the values it stores (taken before their rounding to fp16: the rounding noise is zero-mean and moves the
                             moments of a frame by ~1e-7 relative), float [frames * chunks][2][N] with chunks =
                             mgld_igemm_gn_chunks(p) tiles per frame (> 0: the kernel the launcher picks produces them — the ping-pong
                             patch convolution today; 0: it does not, and a launch with gn_part set is refused).  The consumer passes
                             {gn_part, MGLD_GN_CHANNEL_SUMS, chunks} to mgld_gn_apply2 / mgld_spade_apply2 instead of launching mgld_gn_stats
                             (GroupNorm32 on a convolution's output: openaimodel.py:401-405,429-436).                                */
  int32_t r_f32;          /* 1: R is fp32 [M, ldr] (needs out_f32, batch <= 1): the fp32 residual stream of the high-precision first-stage
                             encoder (mgld_hp_*, model.py:124-183 ResnetBlock `x + h`); kernels with an fp16-only epilogue are not picked  */
  const void* Rlo;        /* NULL, or the LOW PLANE of the residual: fp16 [M, ldr] (same leading dimension / batch stride as R), the residual
                             the kernel adds is R + 2^-11 * Rlo.  The residual STREAM of the networks (`x + f(x)` handed from block to block:
                             openaimodel.py:359,482, attention.py:431-435,546, model.py:183) is kept as a pair of fp16 planes — hi = fp16(x),
                             lo = fp16((x - hi) * 2^11) — so its rounding error is 2^-22 instead of 2^-11 while every kernel that takes the tensor
                             as a contraction operand keeps reading the hi plane as it is (tests/analysis/resid_sim.py)    */
  void* Clo;              /* NULL, or where the kernel also stores the low plane of its fp16 output: fp16 [M, ldc], Clo = fp16((x - fp16(x)) * 2^11)
                             of the fp32 value x it rounds into C.  Not with out_f32 / GEGLU / batch > 1.                                       */
  /* ---- LayerNorm folded into the projection that consumes it (LINEAR problems the ping-pong kernel takes: mgld_igemm_row_chunks > 0) ----
   * attention.py:427-435: `attn1(norm1(x))`, `attn2(norm2(x))`, `ff(norm3(x))` — LayerNorm over the K = C channels of a token row followed by
   * a linear map is   y[m, n] = rstd[m] * (sum_k x[m, k] g[k] W[n, k]  -  mean[m] * s[n]) + b'[n],   s[n] = sum_k g[k] W[n, k],
   * b'[n] = sum_k beta[k] W[n, k] + bias[n]:  the contraction runs on the RAW token rows against W' = W diag(g) (packed on the host), and
   * the epilogue applies the per-row scale and the rank-one mean correction.  No normalised copy of the tokens is written or read, no
   * LayerNorm launch, and the operand skips one rounding to fp16.  The per-row sums come from the kernel that PRODUCED the token rows:     */
  float* row_part;        /* producer side.  NULL, or float [row_chunks][M][2]: the kernel also writes, per output row and column tile, the (sum, sumsq)
                             of the values it stores in that tile (taken before their rounding to fp16); row_chunks = mgld_igemm_row_chunks(p)    */
  const float* ln_part;   /* consumer side.  NULL, or the row sums of A's rows as written through row_part: float [ln_chunks][M][2]; mean / rstd of
                             row m over the K channels = the chunk sums added up, / K, eps = ln_eps.  The epilogue becomes
                             act(rstd * (acc - mean * ln_s[n]) + bias[n] + rowvec) ...; W must be W' and bias b' as above (GEGLU: in the packed
                             row order of W).                                                                                                      */
  const float* ln_s;      /* [N] the column sums s[n] of W' (fp32, of the fp16-rounded W' so that the correction cancels what the MFMA summed)    */
  int32_t ln_chunks;
  float ln_eps;
} MgldIGemm;

/* tiles per frame of the statistics output (see gn_part), or 0 when the kernel picked for this problem does not produce it */
int mgld_igemm_gn_chunks(const MgldIGemm* p);
/* column tiles of the row-statistics output (see row_part), or 0 when the kernel picked for this problem does not produce it / does not take
 * ln_part (today: the ping-pong LINEAR kernels produce and consume; everything else 0, and a launch with either field set is refused) */
int mgld_igemm_row_chunks(const MgldIGemm* p);

int mgld_igemm(const MgldIGemm* p, void* stream);
/* block tile the launcher selects for this problem, encoded BM*1000+BN, plus splits*1000000 when the problem is
 * split along K (profiling / roofline bookkeeping) */
int mgld_igemm_config(const MgldIGemm* p);
/* name of the kernel instantiation the launcher runs for this problem, spelled as rocprofv3 --kernel-trace prints it (e.g.
 * "conv3q_kernel<16, 16, 64, 64, 32, false>"); returns the number of K splits (>= 1) or a negative error.  bench.py groups
 * its in-sequence hipEvent timings by this name so that `roofline` and profiles/ speak about the same kernel. */
int mgld_igemm_kernel_name(const MgldIGemm* p, char* buf, int buflen);
/* fp32 scratch for split-K partial sums (few output tiles, deep K: the 16x16 / 8x8 UNet levels).  Caller-owned device
 * memory, must stay valid while launches that may use it are in flight / captured; single-stream use.  Without it the
 * launcher falls back to smaller tiles.  The registration is per calling host thread (one thread drives one stream): threads that
 * launch concurrently on different streams register different buffers. */
int mgld_set_workspace(void* ptr, int64_t bytes);

/* ---- K3: GroupNorm (32 groups) on NHWC fp16, fp32 statistics -----------------------------------------------
 * Replaces nn.GroupNorm / GroupNorm32 (diffusionmodules/util.py:214-216, model.py:80-81, attention.py:87-88).
 * mgld_gn_stats: one launch; per (frame, row chunk) the per-group (sum, sumsq) in fp64 ->
 *   gsums[frames][chunks = mgld_gn_chunks(rows)][groups][2] doubles.  The consumers below finish the reduction over
 *   chunks (fp64) in their prologue and derive mean / rstd themselves — no separate finalize launch.  groups <= 256.
 */
int mgld_gn_chunks(int rows_per_frame);
int mgld_gn_stats(const void* x, int frames, int rows_per_frame, int C, int ld, int groups, double* gsums, void* stream);
/* y = act((x-mean)*rstd*gamma+beta); x,y fp16 NHWC; act: 0 none, 1 SiLU, 2 ReLU.  groups <= 256 (groups == C gives
 * InstanceNorm2d, raft_arch.py:115-119,211) */
int mgld_gn_apply(const void* x, int ldx, const double* gsums, float eps, const float* gamma, const float* beta,
                  void* y, int ldy, int frames, int rows_per_frame, int C, int groups, int silu, void* stream);
/* SPADE modulation + residual (spade.py:93-111, openaimodel.py:481-482):
 * y = skip + ((h-mean)*rstd*gamma+beta) * (1+gb[:, 0:C]) + gb[:, C:2C]          */
int mgld_spade_apply(const void* h, int ldh, const double* gsums, float eps, const float* gamma, const float* beta,
                     const void* gb, int ldgb, const void* skip, int ldskip, void* y, int ldy,
                     int frames, int rows_per_frame, int C, int groups, const int32_t* gb_step_idx, int64_t gb_step_stride,
                     void* stream);
/* Where a GroupNorm consumer finds the statistics of its input, when they were produced by the kernel that WROTE that input
 * (no mgld_gn_stats launch): kind MGLD_GN_GROUP_SUMS = double [frames][chunks][groups][2] (mgld_gn_stats' format with any chunk
 * count: the stats_out of mgld_gn_apply2 / mgld_spade_apply2, chunks = mgld_gn_apply_chunks(...)); kind MGLD_GN_CHANNEL_SUMS =
 * float [frames * chunks][2][C] per-channel (sum, sumsq) of every output tile (MgldIGemm.gn_part, chunks = mgld_igemm_gn_chunks). */
#define MGLD_GN_GROUP_SUMS 0
#define MGLD_GN_CHANNEL_SUMS 1
typedef struct MgldGnStats {
  const void* sums;
  int kind;
  int chunks; /* per frame */
} MgldGnStats;
/* mgld_gn_apply / mgld_spade_apply with the statistics source spelled out and, when stats_out != NULL, the per-group sums of the
 * OUTPUT y (the values stored, before their fp16 rounding) written as double [frames][mgld_gn_apply_chunks(frames, rows, C, groups)][groups][2]. */
int mgld_gn_apply_chunks(int frames, int rows_per_frame, int C, int groups);
int mgld_gn_apply2(const void* x, int ldx, const MgldGnStats* st, float eps, const float* gamma, const float* beta,
                   void* y, int ldy, int frames, int rows_per_frame, int C, int groups, int silu, double* stats_out, void* stream);
int mgld_spade_apply2(const void* h, int ldh, const MgldGnStats* st, float eps, const float* gamma, const float* beta,
                      const void* gb, int ldgb, const void* skip, int ldskip, void* y, int ldy,
                      int frames, int rows_per_frame, int C, int groups, const int32_t* gb_step_idx, int64_t gb_step_stride,
                      double* stats_out, void* stream);
/* Statistics + apply in ONE launch for small frames (rows_per_frame <= 256: the UNet's 16x16 / 8x8 levels), plain
 * (gb == NULL: y = act(GN(x))) or SPADE (gb, skip given: the mgld_spade_apply formula).  Same arithmetic as
 * mgld_gn_stats + mgld_gn_apply / mgld_spade_apply (fp32 per-channel sums, fp64 per-group combine); mgld_gn_fused_applies
 * tells whether a shape is covered (whole-group windows of <= 128 channels). */
int mgld_gn_fused_applies(int rows_per_frame, int C, int groups);
int mgld_gn_fused(const void* x, int ldx, float eps, const float* gamma, const float* beta, const void* gb, int ldgb,
                  const void* skip, int ldskip, void* y, int ldy, int frames, int rows_per_frame, int C, int groups, int silu,
                  const int32_t* gb_step_idx, int64_t gb_step_stride, void* stream);
/* gb_step_idx != NULL: `gb` is a table holding the modulation of EVERY schedule step (it depends on the struct-cond features
 * only, not on the sample); the kernel reads slice gb + gb_step_idx[0]*gb_step_stride (elements), the index living in device
 * memory so a captured step replays unchanged. */
/* LayerNorm over C per token (attention.py:125,427-429), eps 1e-5 */
int mgld_layernorm(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy,
                   int rows, int C, float eps, void* stream);

/* ---- the residual stream as two fp16 planes (see MgldIGemm.Rlo / Clo) ------------------------------------------------------------
 * The normalisations that READ the stream (`normalization(channels)` / `Normalize` in front of a block: openaimodel.py:271,401, attention.py:
 * 427-429,536, model.py:134) and the one kernel besides the contractions that WRITES it (ResBlockDual's `skip_connection(x) + spade(h)`,
 * openaimodel.py:478-482) in the two-plane form: value = hi + 2^-11 * lo, low planes share the leading dimension of their hi plane.
 * mgld_gn_apply_lo / mgld_layernorm_lo: x is (x, xlo); statistics `st` are those of the hi plane (the zero-mean 2^-12 residual moves the
 * moments of a group by ~1e-7).  mgld_spade_apply_lo: skip is (skip, skiplo), the result goes to (y, ylo); h is an ordinary fp16 tensor.
 * mgld_gn_fused_lo: plain form (gb == NULL): lo_in = low plane of x, lo_out = NULL; SPADE form: lo_in = low plane of skip, lo_out = of y. */
int mgld_gn_apply_lo(const void* x, const void* xlo, int ldx, const MgldGnStats* st, float eps, const float* gamma, const float* beta,
                     void* y, int ldy, int frames, int rows_per_frame, int C, int groups, int silu, void* stream);
int mgld_spade_apply_lo(const void* h, int ldh, const MgldGnStats* st, float eps, const float* gamma, const float* beta,
                        const void* gb, int ldgb, const void* skip, const void* skiplo, int ldskip, void* y, void* ylo, int ldy,
                        int frames, int rows_per_frame, int C, int groups, const int32_t* gb_step_idx, int64_t gb_step_stride,
                        double* stats_out, void* stream);
int mgld_gn_fused_lo(const void* x, int ldx, float eps, const float* gamma, const float* beta, const void* gb, int ldgb,
                     const void* skip, int ldskip, void* y, int ldy, int frames, int rows_per_frame, int C, int groups, int silu,
                     const int32_t* gb_step_idx, int64_t gb_step_stride, const void* lo_in, void* lo_out, void* stream);
int mgld_layernorm_lo(const void* x, const void* xlo, int ldx, const float* gamma, const float* beta, void* y, int ldy,
                      int rows, int C, float eps, void* stream);

/* ---- K5/K6/K7/K9: attention ----------------------------------------------------------------------------------
 * Flash attention, fp16 q/k/v, fp32 online softmax, exact softmax(q k^T * scale) v.
 * Replaces xformers.ops.memory_efficient_attention (attention.py:298,371; openaimodel.py:582).
 * Element (b, row i, head h, dim d):  q: Q + b*q_sb + i*q_si + h*q_sh + d   (same for k, o)
 * V is consumed TRANSPOSED: vt element (b, h, d, key j) at Vt + b*vt_sb + h*vt_sh + d*vt_sd + j
 * (the V projection GEMM writes it that way; vt_sd % 4 == 0, rows zero-padded past Nkv).
 * v_rowmajor = 1 (round 3): `Vt` points at V itself, laid out like K — element (b, key j, h, d) at Vt + b*vt_sb + j*vt_sd + h*vt_sh + d —
 * so that q, k and v can be the three column blocks of ONE fused projection (attention.py:323-330: to_q / to_k / to_v as one
 * GEMM with N = 3C); the kernel transposes on the LDS read (ds_read_b64_tr_b16).
 * head_dim in {64, 128}.
 */
typedef struct MgldAttn {
  const void* Q; const void* K; const void* Vt; void* O;
  int32_t batch, heads, Nq, Nkv, head_dim;
  int64_t q_sb, q_si, q_sh;
  int64_t k_sb, k_si, k_sh;
  int64_t vt_sb, vt_sh, vt_sd;
  int64_t o_sb, o_si, o_sh;
  float scale;
  int32_t v_rowmajor;
} MgldAttn;
int mgld_attention(const MgldAttn* p, void* stream);
/* name of the kernel instantiation mgld_attention launches for this problem, spelled as rocprofv3 --kernel-trace prints it (bench.py groups
 * its in-sequence timings by it, like mgld_igemm_kernel_name) */
int mgld_attention_kernel_name(const MgldAttn* p, char* buf, int buflen);
/* TemporalAttention core (attention.py:124-143 -> 262-308): per pixel and head, softmax over the T frames.
 * q,k,v,o: frame-major token matrices [T*HW, ld] fp16 (row = t*HW + pixel), head h at columns h*head_dim.. ;
 * T <= 16, head_dim in {64,128}. One wave per (pixel, head). */
int mgld_temporal_attention(const void* q, const void* k, const void* v, int ld, void* o, int ldo, int T, int HW,
                            int heads, int head_dim, float scale, void* stream);
/* row softmax fp32 [rows, ld_s] -> fp16 [rows, ld_p] (VAE mid attention d=512, model.py:220-244) */
int mgld_softmax_rows(const float* S, int64_t ld_s, void* P, int64_t ld_p, int64_t rows, int cols, void* stream);
/* same with a causal mask inside blocks of `causal_period` rows (column c of row r kept iff c <= r % period; 0 = none) and
 * zero-filled pad columns [cols, cols_pad): the OpenCLIP text transformer's attention (modules.py:181-191) */
int mgld_softmax_rows_masked(const float* S, int64_t ld_s, void* P, int64_t ld_p, int64_t rows, int cols, int cols_pad,
                             int causal_period, void* stream);

/* ---- small dense ops -----------------------------------------------------------------------------------------
 * y[m,n] = act_out( sum_k act_in(a[m,k]) * w[n,k] + b[n] ), M <= 16, fp32 a/y, fp16 w: weight-streaming GEMV
 * for the time-embedding MLPs (openaimodel.py:2020-2025, 289-295; diffusionmodules/util.py:151-171). */
int mgld_linear_small(const float* a, int lda, const void* w, int ldw, const float* b, float* y, int ldy,
                      int M, int N, int K, int silu_in, int silu_out, void* stream);
/* sinusoidal embedding: out[m, :dim] = [cos(t*f), sin(t*f)], t = tvals[m*t_stride] (float) */
int mgld_timestep_embedding(const float* tvals, int t_stride, float* out, int M, int dim, void* stream);

/* ---- layout / copies -------------------------------------------------------------------------------------- */
/* fp32 NCHW [n,c,h,w] -> fp16 NHWC [n*h*w, ld] (channels >= c zero-filled up to cpad) */
int mgld_nchw_to_nhwc(const float* x, void* y, int n, int c, int h, int w, int cpad, int ld, void* stream);
/* fp16/fp32 NHWC [n*h*w, ld] (first c channels) -> fp32 NCHW */
int mgld_nhwc_to_nchw(const void* x, int in_f32, int ld, float* y, int n, int c, int h, int w, void* stream);
/* packed 3x3 conv weights fp16 [N, 9*Cin] (K order (tap, Cin); tap_inner = 1: (64-channel block, tap, channel)) -> the tiled image the patch
 * convolutions take (MgldIGemm.tap_inner = 2): fp16 [(N rounded up to 64) * 9 * Cin / 32, 32], zero rows past N.  The one-off re-layout
 * of a convolution's weights on first use (round 5: a kernel instead of a torch gather). */
int mgld_tile_conv3p(const void* wp, int N, int Cin, int tap_inner, void* out, void* stream);
/* strided fp16 2-D copy: dst[r, 0:cols] = src[r, 0:cols]; cols % 8 == 0 */
int mgld_copy2d(const void* src, int lds, void* dst, int ldd, int64_t rows, int cols, void* stream);
/* y = a*x + b*y (fp16, strided) */
int mgld_axpby(const void* x, int ldx, void* y, int ldy, int64_t rows, int cols, float a, float b, void* stream);
/* the same on residual-stream tensors kept as two fp16 planes (MgldIGemm.Rlo): (y, ylo) <- a (x, xlo) + b (y, ylo), value = hi + 2^-11 lo; xlo may be
 * NULL.  The fusion layers' `dec_feat + w * enc_feat` (model.py:1367). */
int mgld_axpby_lo(const void* x, const void* xlo, int ldx, void* y, void* ylo, int ldy, int64_t rows, int cols, float a, float b, void* stream);

/* ---- B2/B3/F1/K10/K15: one reverse-diffusion step + motion guidance (ddpm.py:340-353, 3538-3574, 4325-4380) ---
 * coef table row (8 floats per schedule index i): {sqrt_recip_ac, sqrt_recipm1_ac, post_mean_coef1,
 * post_mean_coef2, post_log_var_clipped, nonzero(i!=0), t_replace, unused}. `step_idx` is a device int. */
/* z = mean(x, eps) + nonzero*exp(0.5*logvar)*noise ; x,noise,z fp32 NCHW [n,4,h,w]; eps fp32 NHWC [n*h*w, ld_eps]
 * (ld_eps <= 0: eps is NCHW like x — the stitched aggregation-sampling canvas);
 * the noise of schedule index i is read at noise + i*noise_step_stride (stride 0: one tensor) so that a captured
 * hipGraph of the step can be replayed for every i without host involvement. */
int mgld_ddpm_step(const float* x, const float* eps, int ld_eps, const float* noise, int64_t noise_step_stride,
                   const float* coef, const int32_t* step_idx, float* z, int n, int c, int h, int w, void* stream);
/* bilinear backward warp, zeros padding, align_corners=True (arch_util.py:156-194): out[n,c,y,x] =
 * bilinear(in[n,c], x+flow[n,0,y,x], y+flow[n,1,y,x]); flow fp32 [n,2,h,w] */
int mgld_flow_warp(const float* in, const float* flow, float* out, int n, int c, int h, int w, void* stream);
/* guidance: given z [T,c,h,w], flows fwd_prop/bwd_prop [T-1,2,h,w], occlusion masks fwd_occ/bwd_occ [T-1,h,w]:
 * x_out = z - guidance_scale*logvar(step)*grad(L)(z), L = compute_temporal_condition_v4.
 * work: >= (2*T*c*h*w) floats + (T*c*h*w) int64 (scatter accumulators), caller-provided. */
int mgld_guidance(const float* z, const float* flow_fwd_prop, const float* flow_bwd_prop,
                  const float* fwd_occ, const float* bwd_occ, const float* coef, const int32_t* step_idx,
                  float guidance_scale, float* x_out, void* work, int T, int c, int h, int w, void* stream);
/* scalar loss value of compute_temporal_condition_v4 (tests) -> loss_out[0] (device) */
int mgld_guidance_loss(const float* z, const float* flow_fwd_prop, const float* flow_bwd_prop,
                       const float* fwd_occ, const float* bwd_occ, float* loss_out, void* work,
                       int T, int c, int h, int w, void* stream);
/* step_idx[0] += delta */
int mgld_step_advance(int32_t* step_idx, int delta, void* stream);
/* tvals[m] = coef[step][6] (t_replace) for m < n  (feeds mgld_timestep_embedding) */
int mgld_step_timestep(const float* coef, const int32_t* step_idx, float* tvals, int n, void* stream);

/* ---- F3/F4: flow utilities (util_flow.py:114-136, arch_util.py:235-270) --------------------------------- */
int mgld_fb_consistency(const float* fwd_flow, const float* bwd_flow, float alpha, float beta,
                        float* fwd_occ, float* bwd_occ, int n, int h, int w, void* stream);
/* bilinear resize (align_corners=False) of [n,2,h,w] flow to [n,2,oh,ow] with u,v rescaled */
int mgld_resize_flow(const float* flow, float* out, int n, int h, int w, int oh, int ow, void* stream);

/* ---- G1 tail / A4 / H4: per-segment scalar arithmetic around the networks ------------------------------- */
/* init = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise): DiagonalGaussianDistribution.sample + get_first_stage_encoding
 * (distributions.py:24-40, ddpm.py:3382-3389) on the encoder's moments [n, 2c, hw] (mean planes then logvar planes per frame), and,
 * when x_T != NULL, x_T = sqrt_ac * init + sqrt_one_minus_ac * n0: q_sample_respace at ONE timestep (ddpm.py:403-406; the scripts
 * noise every frame to t = 999).  noise, n0, init, x_T: [n, c, hw] fp32 */
int mgld_init_latent(const float* moments, const float* noise, const float* n0, float* init, float* x_T, int n, int c, int64_t hw,
                     float scale, float sqrt_ac, float sqrt_one_minus_ac, void* stream);
/* out = clamp((x + 1) / 2, 0, 1): the final mapping of decoded frames to [0,1] (oldcanvas_tile.py:471); in place allowed */
int mgld_to01(const float* x, float* out, int64_t n, void* stream);

/* ---- H1/H2: colour fix (wavelet_color_fix.py:44-119) --------------------------------------------------- */
/* out = (x-mean_x)/sqrt(var_x+eps)*sqrt(var_s+eps)+mean_s per (n,c) plane; unbiased variance; fp32 NCHW.
 * work: >= 512 * planes floats, 8-byte aligned (fp64 partial sums of up to 64 chunks per plane and tensor) */
int mgld_adain(const float* content, const float* style, float* out, int planes, int64_t hw, float eps,
               float* work, void* stream);
/* 5-level a-trous wavelet: out = high(content) + low(style); work >= 4*planes*h*w floats */
int mgld_wavelet_reconstruction(const float* content, const float* style, float* out, int planes, int h, int w,
                                float* work, void* stream);

/* ---- C1: aggregation-sampling tile ops (ddpm.py:4191-4322, 4601-4616) ---------------------------------- */
/* crop fp32 NCHW src[n,c,H,W] window (y0,x0,th,tw) -> dst[n,c,th,tw] */
int mgld_crop(const float* src, float* dst, int n, int c, int H, int W, int y0, int x0, int th, int tw, void* stream);
/* acc[n,c,y0+y,x0+x] += tile[n,c,y,x]*wgt[y,x]; cnt[.. same ..] += wgt[y,x]   (tile fp32 NCHW) */
int mgld_tile_accumulate(const float* tile, const float* wgt, float* acc, float* cnt, int n, int c, int H, int W,
                         int y0, int x0, int th, int tw, void* stream);
/* out = acc / cnt */
int mgld_tile_normalize(const float* acc, const float* cnt, float* out, int64_t numel, void* stream);


/* dst[0:bytes_per_step) = src[step*bytes_per_step : ...) with `step` read from device memory at run time: selects the
 * current DDPM step's slice of a tensor precomputed for the whole schedule (the struct-cond features, which depend on the
 * timestep but not on the sample) inside a replayed hipGraph.  bytes_per_step % 16 == 0, 16-byte aligned pointers. */
int mgld_copy_step(const void* src, void* dst, int64_t bytes_per_step, const int32_t* step_idx, void* stream);

/* ---- K11: pre/post-processing on the device (oldcanvas_tile.py:349-357,384-397,523-543) -----------
 * bicubic resize with torch.nn.functional.interpolate(mode="bicubic", align_corners=False) semantics (A = -0.75, border
 * taps clamped, no antialias), fp32 planes [planes, h, w] -> [planes, oh, ow], result clamped to [lo, hi]
 * (pass -inf/+inf for none): the x4 pre-upsampling of the LR frames and the /4 downscale fed to the flow network. */
int mgld_resize_bicubic(const float* x, float* y, int planes, int h, int w, int oh, int ow, float lo, float hi, void* stream);
/* torchvision Resize + CenterCrop on a tensor, as scripts/vsr_val_ddpm_text_T_vqganfin_old.py:253-256,315 apply them
 * (F.interpolate(mode="bilinear", align_corners=False), no antialias, to rh x rw; then the oh x ow window at (cy, cx)) in
 * one pass: fp32 planes [planes, h, w] -> [planes, oh, ow] */
int mgld_resize_bilinear_crop(const float* x, float* y, int planes, int h, int w, int rh, int rw, int oh, int ow, int cy, int cx,
                              void* stream);
/* F.pad(x, (0, ow-w, 0, oh-h), mode="reflect"): bottom / right reflect padding to the next multiple of 32 */
int mgld_reflect_pad(const float* x, float* y, int planes, int h, int w, int oh, int ow, void* stream);
/* F.pad(x, (pl, ow-w-pl, pt, oh-h-pt), mode="replicate") (RAFT InputPadder, raft_arch.py:27-28) */
int mgld_replicate_pad(const float* x, float* y, int planes, int h, int w, int oh, int ow, int pt, int pl, void* stream);
/* [n,c,H,W] fp32 in [0,1] -> uint8 [n,h,w,c] of the top-left h x w window: (x*255).astype(uint8), i.e. truncation, as the
 * reference writes its PNGs (oldcanvas_tile.py:532-543) */
int mgld_to_uint8_hwc(const float* x, void* y, int n, int c, int H, int W, int h, int w, void* stream);


/* ---- RAFT_SR flow estimator (basicsr/archs/raft_arch.py:668-807, called from ddpm.py:3404-3429) ------------
 * Entirely fp32 (round 5): the flows feed a thresholded forward/backward consistency check (util_flow.py:114-136) and sub-pixel
 * warps of the latents, and RAFT is < 0.1 % of a segment's arithmetic.  Activations are fp32 NHWC matrices [n*h*w, ld]. */
/* Every nn.Conv2d of RAFT_SR (7x7 s2, 3x3 s1/s2, 1x1 s1/s2, 1x5, 5x1; raft_arch.py:95-97,216,246,383-389,430-434,354-355,471-473)
 * and the all-pairs correlation matmul (:82-85, batched LINEAR form) as an implicit GEMM on the f32-input MFMA:
 *   C[m][n] = post_relu( act( alpha * (sum_k A[m][k] * W[n][k] + bias[n]) ) + R[m][n] )
 * m = (image, oy, ox) over n_img*Hout*Wout = M rows, k = (tap, channel); A is gathered from the NHWC input with zero padding.
 * W: fp32 [N][kh*kw][Cin4], Cin4 = Cin rounded up to 4 (zero filled).  act: MGLD_ACT_NONE / RELU / SIGMOID / TANH.
 * R (optional, leading dimension ldr) is added AFTER the activation; post_relu then clamps (ResidualBlock tail, :138).
 * batch > 1 (kh = kw = 1 only): operand b starts at A + b*strideA, W + b*strideW, C (and R) + b*strideC (floats). */
typedef struct MgldConvF32 {
  const float* A;
  const float* W;
  const float* bias; /* [N] or NULL */
  const float* R;    /* [M, ldr] or NULL */
  float* C;
  int64_t M;
  int32_t N, Cin, lda, ldc, ldr;
  int32_t Hin, Win, Hout, Wout, kh, kw, stride, pad_t, pad_l;
  int32_t act, post_relu;
  float alpha;
  int32_t batch;
  int64_t strideA, strideW, strideC;
} MgldConvF32;
int mgld_conv_f32(const MgldConvF32* p, void* stream);
/* nn.InstanceNorm2d (no affine, eps 1e-5; raft_arch.py:115-119,211) on fp32 NHWC [n*hw, ldx] -> y, optional ReLU, optional
 * ResidualBlock tail y = relu(skip + y) (:138).  fp64 statistics; `part` is scratch of n * mgld_instnorm_chunks(hw) * C * 2 doubles. */
int mgld_instnorm_chunks(int hw);
int mgld_instnorm_f32(const float* x, int ldx, double* part, const float* skip, int lds, float* y, int ldy, int n, int hw, int C,
                      float eps, int relu, void* stream);
/* [n,c,h,w] fp32 -> fp32 NHWC [n*hw, ld >= c], pad columns zeroed (the encoders' RGB input, :248-250) */
int mgld_nchw_to_nhwc_f32(const float* x, float* y, int n, int c, int hw, int ld, void* stream);
/* F.avg_pool2d(x, 2, stride=2) on fp32 planes [planes, h, w] -> [planes, h/2, w/2]  (correlation pyramid, :47-49) */
int mgld_avgpool2(const float* x, float* y, int64_t planes, int h, int w, void* stream);
/* CorrBlock.__call__ (:54-75): windowed bilinear lookup of the pyramid at coords [B,2,H,W] (pixel units, x then y);
 * levels[l] = fp32 [B*H*W, hs[l], ws[l]]; out fp32 [B*H*W, ldo], column l*(2r+1)^2 + i*(2r+1) + j samples
 * (cx/2^l + i - r, cy/2^l + j - r) with zeros outside (grid_sample, align_corners=True). */
int mgld_corr_lookup(const float* const* levels, const int* hs, const int* ws, int nlev, const float* coords, int B, int H, int W,
                     int radius, float* out, int ldo, void* stream);
/* SepConvGRU arithmetic (:390-405), fp32 [M, *] views: rhx = cat([r*h, x]) from hx = cat([h, x]);  h = (1-z)*h + z*q */
int mgld_gru_rh(const float* r, int ldr, const float* hx, int ldhx, float* rhx, int ldo, int64_t M, int Ch, int Cx, void* stream);
int mgld_gru_gate(const float* z, int ldz, const float* q, int ldq, float* h, int ldh, int64_t M, int Ch, void* stream);
/* coords1 += delta (fp32 NHWC [B*HW, ldd] columns 0,1; NULL = no update); flow = coords1 - coords0 (fp32 NCHW), also
 * written as two fp32 columns at `mot` / `fin` (NULL to skip): the motion-feature tail and the flow-conv input (:444,:778-783) */
int mgld_flow_update(float* coords1, const float* coords0, const float* delta, int ldd, float* flow, float* mot, int ldm, float* fin,
                     int ldf, int B, int HW, void* stream);
/* convex 8x upsampling (:720-731): mask fp32 NHWC [B*H*W, ldm >= 576] (already scaled by 0.25), flow fp32 [B,2,H,W] ->
 * out fp32 [B,2,8H,8W] */
int mgld_convex_upsample(const float* flow, const float* mask, int ldm, float* out, int B, int H, int W, void* stream);

/* ---- high-precision first-stage encoder (round 5; model.py:473-572 Encoder behind encode_first_stage, ddpm.py:3906-3943) --------
 * The first-stage latent conditions every sampling step, so its error does not average out: fp32 activations between the kernels,
 * split-fp16 operands inside the contractions (activation row [ah | 16 al | ah/256] against weight row [wh | wh/16 | 256 wl]: three
 * times the channels through the ordinary mgld_igemm with out_f32 / r_f32), see csrc/hpenc.hip. */
/* row chunks the two kernels below split a frame of `rows` rows into */
int mgld_hp_chunks(int rows);
/* GroupNorm statistics of an fp32 NHWC tensor [frames*rows, ldx] (model.py:42-43 Normalize): gsums[frame][chunk][group][2] fp64
 * (sum, sumsq), chunk < mgld_hp_chunks(rows); deterministic.  C % 4 == 0, C <= 1024. */
int mgld_hp_gn_stats(const float* x, int ldx, int frames, int rows, int C, int groups, double* gsums, void* stream);
/* y = [silu]((x - mean) * rstd * gamma + beta) from those statistics (gsums == NULL: y = x), written either as the split-fp16
 * contraction operand fp16 [frames*rows, ldo >= 3C] = [yh | 16 yl | yh/256] (out_f32 = 0) or as fp32 [frames*rows, ldo >= C]
 * (out_f32 = 1; the attention block's input).  model.py:134-139 (norm1 / swish), :474 ff. */
int mgld_hp_gn_split(const float* x, int ldx, const double* gsums, float eps, const float* gamma, const float* beta, int silu, void* out,
                     int ldo, int out_f32, int frames, int rows, int C, int groups, void* stream);
/* in-place softmax over the `cols` entries of every fp32 row (the mid attention's key axis, model.py:226-229) */
int mgld_hp_softmax_rows(float* S, int64_t rows, int cols, int ld, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MGLD_HIP_H */
