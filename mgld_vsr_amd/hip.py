"""ctypes binding of libmgld_hip.so (include/mgld_hip.h) + thin tensor-level wrappers.

PyTorch is used here only as plumbing: device memory (torch tensors own the buffers), the current HIP stream,
and dtype bookkeeping.  Every compute call goes through the C ABI; if the library is missing the import of
this module's `lib()` fails loudly — there is no eager/CPU fallback on the product path.
"""
import ctypes as C
import threading
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmgld_hip.so")
_lib = None

MODE_LINEAR, MODE_CONV3X3, MODE_TCONV3 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LRELU02, ACT_SILU, ACT_GEGLU, ACT_SIGMOID, ACT_TANH, ACT_GELU = 0, 1, 2, 3, 4, 5, 6, 7


class MgldIGemm(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p),
        ("bias", C.c_void_p), ("bias_m", C.c_void_p), ("rowvec", C.c_void_p), ("R", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldw", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32), ("ld_rowvec", C.c_int32),
        ("mode", C.c_int32), ("Cin", C.c_int32),
        ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32),
        ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32), ("up2", C.c_int32),
        ("T", C.c_int32), ("HW", C.c_int32), ("rows_per_frame", C.c_int32),
        ("act", C.c_int32), ("out_f32", C.c_int32),
        ("alpha", C.c_float), ("beta", C.c_float),
        ("batch", C.c_int32), ("tap_inner", C.c_int32),
        ("strideA", C.c_int64), ("strideW", C.c_int64), ("strideC", C.c_int64), ("strideR", C.c_int64),
        ("t_off", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("tune", C.c_int32),
        ("w2_scale", C.c_float), ("W2", C.c_void_p), ("gn_part", C.c_void_p), ("r_f32", C.c_int32),
        ("Rlo", C.c_void_p), ("Clo", C.c_void_p),
        ("row_part", C.c_void_p), ("ln_part", C.c_void_p), ("ln_s", C.c_void_p), ("ln_chunks", C.c_int32), ("ln_eps", C.c_float),
    ]


class MgldAttn(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("Vt", C.c_void_p), ("O", C.c_void_p),
        ("batch", C.c_int32), ("heads", C.c_int32), ("Nq", C.c_int32), ("Nkv", C.c_int32), ("head_dim", C.c_int32),
        ("q_sb", C.c_int64), ("q_si", C.c_int64), ("q_sh", C.c_int64),
        ("k_sb", C.c_int64), ("k_si", C.c_int64), ("k_sh", C.c_int64),
        ("vt_sb", C.c_int64), ("vt_sh", C.c_int64), ("vt_sd", C.c_int64),
        ("o_sb", C.c_int64), ("o_si", C.c_int64), ("o_sh", C.c_int64),
        ("scale", C.c_float), ("v_rowmajor", C.c_int32),
    ]


# every symbol include/mgld_hip.h declares (tests check the library exports all of them)
EXPORTS = [
    "mgld_version", "mgld_last_error", "mgld_device_info",
    "mgld_graph_begin", "mgld_graph_end", "mgld_graph_launch", "mgld_graph_destroy",
    "mgld_event_create", "mgld_event_record", "mgld_event_sync", "mgld_event_elapsed_ms", "mgld_event_destroy",
    "mgld_igemm", "mgld_igemm_config", "mgld_igemm_kernel_name", "mgld_igemm_gn_chunks", "mgld_igemm_row_chunks", "mgld_set_workspace", "mgld_gn_chunks", "mgld_gn_stats", "mgld_gn_apply", "mgld_spade_apply",
    "mgld_gn_apply_chunks", "mgld_gn_apply2", "mgld_spade_apply2", "mgld_gn_fused_applies", "mgld_gn_fused", "mgld_layernorm",
    "mgld_gn_apply_lo", "mgld_spade_apply_lo", "mgld_gn_fused_lo", "mgld_layernorm_lo",
    "mgld_attention", "mgld_attention_kernel_name", "mgld_temporal_attention", "mgld_softmax_rows", "mgld_softmax_rows_masked",
    "mgld_linear_small", "mgld_timestep_embedding",
    "mgld_nchw_to_nhwc", "mgld_nhwc_to_nchw", "mgld_copy2d", "mgld_axpby", "mgld_axpby_lo", "mgld_tile_conv3p",
    "mgld_ddpm_step", "mgld_flow_warp", "mgld_guidance", "mgld_guidance_loss", "mgld_step_advance",
    "mgld_step_timestep", "mgld_fb_consistency", "mgld_resize_flow",
    "mgld_adain", "mgld_wavelet_reconstruction", "mgld_init_latent", "mgld_to01",
    "mgld_crop", "mgld_tile_accumulate", "mgld_tile_normalize", "mgld_copy_step",
    "mgld_resize_bicubic", "mgld_resize_bilinear_crop", "mgld_reflect_pad", "mgld_replicate_pad", "mgld_to_uint8_hwc",
    "mgld_hp_chunks", "mgld_hp_gn_stats", "mgld_hp_gn_split", "mgld_hp_softmax_rows",
    "mgld_conv_f32", "mgld_instnorm_chunks", "mgld_instnorm_f32", "mgld_nchw_to_nhwc_f32",
    "mgld_avgpool2", "mgld_corr_lookup", "mgld_gru_rh", "mgld_gru_gate", "mgld_flow_update", "mgld_convex_upsample",
]


def lib_path():
    return _LIB_PATH


def lib():
    """Load libmgld_hip.so; raise if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} is missing: the MGLD-VSR hot path has no CPU/eager fallback. "
                "Build it with `python -m mgld_vsr_amd.build` (hipcc --offload-arch=gfx950).")
        # MGLD_HIP_LIB: load another build of the same library (kernel A/B tuning only)
        _lib = C.CDLL(os.environ.get("MGLD_HIP_LIB") or _LIB_PATH)
        _lib.mgld_last_error.restype = C.c_char_p
        _lib.mgld_gn_chunks.restype = C.c_int
    return _lib


def _chk(rc, what):
    if rc != 0:
        msg = lib().mgld_last_error().decode(errors="replace")
        raise RuntimeError(f"libmgld_hip {what} failed (rc={rc}): {msg}")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libmgld_hip ops need device tensors (no CPU fallback on the product path)")


# --------------------------------------------------------------------------------------------------------------
# tensor-level wrappers.  "Token matrices" are 2-D views [rows, C] with stride (ld, 1) — fp16 unless noted.
# --------------------------------------------------------------------------------------------------------------
def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, f"expected a [rows, C] view with unit channel stride, got {t.shape} {t.stride()}"
    return t.stride(0)


W2_SCALE = 2.0 ** -11     # scale of the weight-residual matrices (MgldIGemm.W2): fp16(residual / W2_SCALE) stays in fp16's normal range


def igemm(a, w, out, *, mode=MODE_LINEAR, bias=None, bias_m=None, rowvec=None, rows_per_frame=0, resid=None,
          act=ACT_NONE, alpha=1.0, beta=1.0, conv=None, tconv=None, batch=1, strideA=0, strideW=0, strideC=0, strideR=0,
          M=None, N=None, K=None, tap_inner=0, t_off=0, ksize=None, tune=0, w2=None, w2_scale=W2_SCALE, gn_part=None,
          resid_lo=None, out_lo=None, row_part=None, ln=None, query_row_chunks=False):
    """out[M,N] = alpha*act(gather(a) @ w^T + bias + rowvec) + beta*resid   (see include/mgld_hip.h).
    w2: the scaled fp16 rounding residual of the weights (same layout as w; engine.split_residual): the product then uses weights
    exact to ~2^-21 at twice the MFMA work.
    resid_lo / out_lo: low planes of the residual / of the output (the residual stream as two fp16 planes, value = hi + 2^-11 lo:
    MgldIGemm.Rlo / Clo); each shares the leading dimension of its hi plane.
    LayerNorm folded into the projection (MgldIGemm.ln_part / row_part): `ln` = (part, chunks, s, eps) — the row sums of a's rows written by
    the launch that produced them, the column sums of w (= W diag(gamma)) and the norm's eps; `row_part(chunks)` -> float32 [chunks, M, 2]
    is called when the kernel picked for this problem writes the row sums of ITS output (the caller keeps what it returned).
    query_row_chunks: no launch; returns mgld_igemm_row_chunks of the problem (> 0: the kernel takes ln / row_part)."""
    _req_cuda(a, w, out, resid_lo, out_lo)
    if not getattr(_TLS, "touched", False):
        ensure_workspace()          # a thread other than the one that built the Engine: give it its own split-K scratch
    p = MgldIGemm()
    p.A, p.W, p.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else None
    p.bias_m = bias_m.data_ptr() if bias_m is not None else None
    p.rowvec = rowvec.data_ptr() if rowvec is not None else None
    p.R = resid.data_ptr() if resid is not None else None
    p.M = M if M is not None else out.shape[0]
    p.N = N if N is not None else w.shape[0]
    p.K = K if K is not None else w.shape[1]
    p.lda, p.ldw, p.ldc = _ld(a), _ld(w), _ld(out)
    p.ldr = _ld(resid) if resid is not None else 0
    p.ld_rowvec = _ld(rowvec) if rowvec is not None else 0
    p.mode = mode
    p.rows_per_frame = rows_per_frame
    p.act = act
    p.out_f32 = 1 if out.dtype == torch.float32 else 0
    p.r_f32 = 1 if (resid is not None and resid.dtype == torch.float32) else 0
    p.alpha, p.beta = alpha, beta
    if resid_lo is not None:
        assert resid is not None and resid_lo.dtype == torch.float16 and resid.dtype == torch.float16 and _ld(resid_lo) == _ld(resid), "resid_lo mirrors resid"
        p.Rlo = resid_lo.data_ptr()
    if out_lo is not None:
        assert out_lo.dtype == torch.float16 and out.dtype == torch.float16 and _ld(out_lo) == _ld(out), "out_lo mirrors out"
        p.Clo = out_lo.data_ptr()
    p.batch = batch
    p.tap_inner = tap_inner
    p.t_off = t_off
    p.tune = tune
    if w2 is not None:
        assert w2.is_cuda and w2.dtype == w.dtype and w2.shape == w.shape and w2.stride() == w.stride(), "w2 must mirror w"
        p.W2, p.w2_scale = w2.data_ptr(), float(w2_scale)
    if ksize is not None:
        p.kh, p.kw = ksize
    p.strideA, p.strideW, p.strideC, p.strideR = strideA, strideW, strideC, strideR
    if mode == MODE_CONV3X3:
        p.Cin, p.Hin, p.Win, p.Hout, p.Wout, p.stride, p.pad_t, p.pad_l, p.up2 = conv
    elif mode == MODE_TCONV3:
        p.Cin, p.T, p.HW = tconv
    if gn_part is not None:
        # gn_part(chunks) -> float32 device tensor [frames * chunks, 2, N] (called only when the kernel picked for this problem writes
        # the statistics of its output: chunks = mgld_igemm_gn_chunks > 0); the caller keeps what it returned
        chunks = lib().mgld_igemm_gn_chunks(C.byref(p))
        if chunks > 0:
            t = gn_part(chunks)
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
            p.gn_part = t.data_ptr()
    if query_row_chunks:
        return lib().mgld_igemm_row_chunks(C.byref(p))
    if ln is not None:
        part, chunks, sv, eps = ln
        assert part.is_cuda and part.dtype == torch.float32 and part.is_contiguous() and part.shape == (chunks, p.M, 2), "ln: row sums [chunks, M, 2]"
        assert sv.is_cuda and sv.dtype == torch.float32 and sv.numel() == p.N
        p.ln_part, p.ln_s, p.ln_chunks, p.ln_eps = part.data_ptr(), sv.data_ptr(), int(chunks), float(eps)
    if row_part is not None:
        chunks = lib().mgld_igemm_row_chunks(C.byref(p))
        if chunks > 0:
            t = row_part(chunks)
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == (chunks, p.M, 2)
            p.row_part = t.data_ptr()
    if IGEMM_LOG is not None:
        IGEMM_LOG.append(MgldIGemm.from_buffer_copy(p))
    if TIMED is not None:
        with timed("igemm", MgldIGemm.from_buffer_copy(p)):
            _chk(lib().mgld_igemm(C.byref(p), stream_ptr()), "igemm")
        return out
    _chk(lib().mgld_igemm(C.byref(p), stream_ptr()), "igemm")
    return out


def conv3p_applies(frames, cin, cout, h, w, up2=False):
    """does the launcher route this 3x3 / stride 1 / pad 1 convolution ((h, w) = INPUT size; up2: nearest-2x upsample folded into
    the gather) to a patch-staged kernel that takes the tiled weight layout (tap_inner = 2)?  Asks the library's own planner
    (mgld_igemm_config), no launch."""
    p = MgldIGemm()
    sc = 2 if up2 else 1
    p.mode, p.M, p.N, p.K, p.batch, p.tap_inner = MODE_CONV3X3, frames * h * w * sc * sc, cout, 9 * cin, 1, 2
    p.Cin, p.Hin, p.Win, p.Hout, p.Wout, p.stride, p.pad_t, p.pad_l, p.up2 = cin, h, w, sc * h, sc * w, 1, 1, 1, 1 if up2 else 0
    return igemm_config(p) % 1000000 >= 300000


TIMED = None       # bench.py: list collecting (kind, info, start Event, stop Event) of launches made inside `timed(...)`


class timed:
    """bracket one launch with hipEvents on the launch stream when hip.TIMED is a list (bench.py's in-sequence kernel timing:
    every launch is measured where it sits in the real launch sequence, operands in the cache state the pipeline leaves them)"""

    def __init__(self, kind, info):
        self.kind, self.info = kind, info

    def __enter__(self):
        if TIMED is not None:
            self.e0, self.e1 = Event(), Event()
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if TIMED is not None:
            self.e1.record()
            TIMED.append((self.kind, self.info, self.e0, self.e1))
        return False


IGEMM_LOG = None   # bench.py sets this to a list to collect the igemm problems of one pass (roofline bookkeeping)


def igemm_relaunch(p):
    _chk(lib().mgld_igemm(C.byref(p), stream_ptr()), "igemm")


def igemm_config(p):
    return lib().mgld_igemm_config(C.byref(p))


def igemm_kernel_name(p):
    """(kernel instantiation name as rocprofv3 prints it, K splits) the launcher picks for this problem"""
    buf = C.create_string_buffer(128)
    splits = lib().mgld_igemm_kernel_name(C.byref(p), buf, 128)
    if splits < 0:
        _chk(splits, "igemm_kernel_name")
    return buf.value.decode(), splits


_TLS = threading.local()     # the library keeps the split-K scratch per host thread; so does this mirror


def set_workspace(t):
    """register a device uint8/float tensor as the calling thread's split-K scratch (the caller keeps it alive); None unregisters"""
    _chk(lib().mgld_set_workspace(_p(t), C.c_int64(t.numel() * t.element_size() if t is not None else 0)), "set_workspace")
    _TLS.touched = True


def ensure_workspace(nbytes=256 << 20):
    """thread-lifetime split-K scratch: one per host thread (a thread drives one stream; concurrent streams must not share slabs)"""
    ws = getattr(_TLS, "ws", None)
    if ws is None or ws.numel() < nbytes:
        ws = _TLS.ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    set_workspace(ws)
    return ws


def igemm_flops(p):
    n_batch = max(1, p.batch)
    return 2.0 * p.M * p.N * p.K * n_batch


def gn_chunks(rows):
    return lib().mgld_gn_chunks(int(rows))


def gn_stats(x, frames, rows, groups, gsums):
    """gsums: float64 [frames, gn_chunks(rows), groups, 2] per-chunk group (sum, sumsq)."""
    _req_cuda(x, gsums)
    assert gsums.dtype == torch.float64 and gsums.numel() >= frames * gn_chunks(rows) * groups * 2
    with timed("gn_stats", {"bytes": 2.0 * frames * rows * x.shape[1]}):
        _chk(lib().mgld_gn_stats(_p(x), frames, rows, x.shape[1], _ld(x), groups, _p(gsums), stream_ptr()), "gn_stats")
    return gsums


class MgldGnStats(C.Structure):
    """include/mgld_hip.h: where a GroupNorm consumer finds the statistics of its input"""
    _fields_ = [("sums", C.c_void_p), ("kind", C.c_int), ("chunks", C.c_int)]


GN_GROUP_SUMS, GN_CHANNEL_SUMS = 0, 1


def _gn_src(sums, kind, chunks, rows):
    st = MgldGnStats()
    st.sums, st.kind, st.chunks = sums.data_ptr(), int(kind), int(chunks if chunks else gn_chunks(rows))
    return st


def gn_apply_chunks(frames, rows, channels, groups):
    """row chunks per frame of the apply kernels' grid = the chunk count of their stats_out"""
    return lib().mgld_gn_apply_chunks(int(frames), int(rows), int(channels), int(groups))


def _lo_of(lo, hi):
    assert lo.is_cuda and lo.dtype == torch.float16 and hi.dtype == torch.float16 and lo.shape == hi.shape and _ld(lo) == _ld(hi), "a low plane mirrors its hi plane"
    return _p(lo)


def gn_apply(x, gsums, eps, gamma, beta, y, frames, rows, groups, silu, kind=GN_GROUP_SUMS, chunks=0, stats_out=None, x_lo=None):
    """kind / chunks: format of `gsums` (default: mgld_gn_stats' output); stats_out: float64 [frames, gn_apply_chunks, groups, 2] that
    receives the per-group sums of y; x_lo: low plane of x (the residual stream as two fp16 planes)"""
    _req_cuda(x, gsums, gamma, beta, y)
    st = _gn_src(gsums, kind, chunks, rows)
    if x_lo is not None:
        assert stats_out is None
        with timed("gn_apply", {"bytes": 6.0 * frames * rows * x.shape[1]}):
            _chk(lib().mgld_gn_apply_lo(_p(x), _lo_of(x_lo, x), _ld(x), C.byref(st), C.c_float(eps), _p(gamma), _p(beta), _p(y), _ld(y), frames,
                                        rows, x.shape[1], groups, int(silu), stream_ptr()), "gn_apply_lo")
        return y
    with timed("gn_apply", {"bytes": 4.0 * frames * rows * x.shape[1]}):
        _chk(lib().mgld_gn_apply2(_p(x), _ld(x), C.byref(st), C.c_float(eps), _p(gamma), _p(beta), _p(y), _ld(y), frames, rows,
                                  x.shape[1], groups, int(silu), _p(stats_out), stream_ptr()), "gn_apply")
    return y


def spade_apply(h, gsums, eps, gamma, beta, gb, skip, y, frames, rows, groups, step_idx=None, step_stride=0, kind=GN_GROUP_SUMS, chunks=0,
                stats_out=None, skip_lo=None, y_lo=None):
    """gb: [frames*rows, 2C] modulation, or (step_idx given) the first slice of a per-step table with `step_stride` elements
    between consecutive steps.  kind / chunks / stats_out as in gn_apply.  skip_lo / y_lo: low planes of skip and y (together)."""
    _req_cuda(h, gsums, gamma, beta, gb, skip, y)
    st = _gn_src(gsums, kind, chunks, rows)
    if y_lo is not None:
        with timed("spade_apply", {"bytes": 14.0 * frames * rows * h.shape[1]}):
            _chk(lib().mgld_spade_apply_lo(_p(h), _ld(h), C.byref(st), C.c_float(eps), _p(gamma), _p(beta), _p(gb), _ld(gb), _p(skip),
                                           _lo_of(skip_lo, skip), _ld(skip), _p(y), _lo_of(y_lo, y), _ld(y), frames, rows, h.shape[1], groups,
                                           _p(step_idx), C.c_int64(step_stride), _p(stats_out), stream_ptr()), "spade_apply_lo")
        return y
    with timed("spade_apply", {"bytes": 10.0 * frames * rows * h.shape[1]}):
        _chk(lib().mgld_spade_apply2(_p(h), _ld(h), C.byref(st), C.c_float(eps), _p(gamma), _p(beta), _p(gb), _ld(gb), _p(skip),
                                     _ld(skip), _p(y), _ld(y), frames, rows, h.shape[1], groups, _p(step_idx), C.c_int64(step_stride),
                                     _p(stats_out), stream_ptr()), "spade_apply")
    return y


def gn_fused_applies(rows, channels, groups):
    """small frames (<= 256 rows, whole-group windows of <= 128 channels): statistics + apply run as one launch"""
    return bool(lib().mgld_gn_fused_applies(int(rows), int(channels), int(groups)))


def gn_fused(x, eps, gamma, beta, y, frames, rows, groups, silu=0, gb=None, skip=None, step_idx=None, step_stride=0, lo_in=None, lo_out=None):
    """y = act(GN(x)) or, with gb / skip, the SPADE formula of spade_apply; one launch (statistics + apply).  lo_in / lo_out: plain form
    — the low plane of x; SPADE form — the low planes of skip and of y"""
    _req_cuda(x, gamma, beta, y)
    spade = gb is not None
    if spade:
        _req_cuda(gb, skip)
    if lo_in is not None:
        with timed("spade_apply" if spade else "gn_apply", {"bytes": (14.0 if spade else 6.0) * frames * rows * x.shape[1]}):
            _chk(lib().mgld_gn_fused_lo(_p(x), _ld(x), C.c_float(eps), _p(gamma), _p(beta), _p(gb) if spade else None, _ld(gb) if spade else 0,
                                        _p(skip) if spade else None, _ld(skip) if spade else 0, _p(y), _ld(y), frames, rows, x.shape[1],
                                        groups, int(silu), _p(step_idx), C.c_int64(step_stride), _lo_of(lo_in, skip if spade else x),
                                        _lo_of(lo_out, y) if spade else None, stream_ptr()), "gn_fused_lo")
        return y
    with timed("spade_apply" if spade else "gn_apply", {"bytes": (10.0 if spade else 4.0) * frames * rows * x.shape[1]}):
        _chk(lib().mgld_gn_fused(_p(x), _ld(x), C.c_float(eps), _p(gamma), _p(beta), _p(gb) if spade else None, _ld(gb) if spade else 0,
                                 _p(skip) if spade else None, _ld(skip) if spade else 0, _p(y), _ld(y), frames, rows, x.shape[1],
                                 groups, int(silu), _p(step_idx), C.c_int64(step_stride), stream_ptr()), "gn_fused")
    return y


def layernorm(x, gamma, beta, y, eps=1e-5, x_lo=None):
    _req_cuda(x, gamma, beta, y)
    if x_lo is not None:
        with timed("layernorm", {"bytes": 6.0 * x.shape[0] * x.shape[1]}):
            _chk(lib().mgld_layernorm_lo(_p(x), _lo_of(x_lo, x), _ld(x), _p(gamma), _p(beta), _p(y), _ld(y), x.shape[0], x.shape[1],
                                         C.c_float(eps), stream_ptr()), "layernorm_lo")
        return y
    with timed("layernorm", {"bytes": 4.0 * x.shape[0] * x.shape[1]}):
        _chk(lib().mgld_layernorm(_p(x), _ld(x), _p(gamma), _p(beta), _p(y), _ld(y), x.shape[0], x.shape[1], C.c_float(eps),
                                  stream_ptr()), "layernorm")
    return y


def attention(q, k, vt, o, *, batch, heads, Nq, Nkv, head_dim, q_strides, k_strides, vt_strides, o_strides, scale, v_rowmajor=False):
    """v_rowmajor: `vt` is V itself, laid out like k (vt_strides = (batch, key row, head)) — the third column block of a fused q|k|v
    projection; otherwise V^T [head][d][key] with vt_strides = (batch, head, d)."""
    _req_cuda(q, k, vt, o)
    p = MgldAttn()
    p.Q, p.K, p.Vt, p.O = q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr()
    p.batch, p.heads, p.Nq, p.Nkv, p.head_dim = batch, heads, Nq, Nkv, head_dim
    p.q_sb, p.q_si, p.q_sh = q_strides
    p.k_sb, p.k_si, p.k_sh = k_strides
    if v_rowmajor:
        p.vt_sb, p.vt_sd, p.vt_sh = vt_strides
        p.v_rowmajor = 1
    else:
        p.vt_sb, p.vt_sh, p.vt_sd = vt_strides
    p.o_sb, p.o_si, p.o_sh = o_strides
    p.scale = scale
    info = None
    if TIMED is not None:
        buf = C.create_string_buffer(96)
        _chk(lib().mgld_attention_kernel_name(C.byref(p), buf, 96), "attention_kernel_name")
        info = {"flops": 4.0 * batch * heads * Nq * Nkv * head_dim, "bytes": 2.0 * batch * heads * head_dim * (2 * Nq + 2 * Nkv), "d": head_dim,
                "vrm": bool(v_rowmajor), "kernel": buf.value.decode()}
    with timed("attention", info):
        _chk(lib().mgld_attention(C.byref(p), stream_ptr()), "attention")
    return o


def temporal_attention(q, k, v, o, T, HW, heads, head_dim, scale):
    _req_cuda(q, k, v, o)
    assert _ld(q) == _ld(k) == _ld(v)
    _chk(lib().mgld_temporal_attention(_p(q), _p(k), _p(v), _ld(q), _p(o), _ld(o), T, HW, heads, head_dim, C.c_float(scale),
                                       stream_ptr()), "temporal_attention")
    return o


def softmax_rows(S, P, rows, cols):
    _req_cuda(S, P)
    _chk(lib().mgld_softmax_rows(_p(S), C.c_int64(S.stride(0)), _p(P), C.c_int64(P.stride(0)), C.c_int64(rows), cols,
                                 stream_ptr()), "softmax_rows")
    return P


def softmax_rows_masked(S, P, rows, cols, cols_pad, causal_period):
    _req_cuda(S, P)
    _chk(lib().mgld_softmax_rows_masked(_p(S), C.c_int64(S.stride(0)), _p(P), C.c_int64(P.stride(0)), C.c_int64(rows), cols,
                                        cols_pad, causal_period, stream_ptr()), "softmax_rows_masked")
    return P


def linear_small(a, w, b, y, silu_in=False, silu_out=False):
    _req_cuda(a, w, y)
    _chk(lib().mgld_linear_small(_p(a), _ld(a), _p(w), _ld(w), _p(b), _p(y), _ld(y), a.shape[0], w.shape[0], w.shape[1],
                                 int(silu_in), int(silu_out), stream_ptr()), "linear_small")
    return y


def timestep_embedding(tvals, out, t_stride=1):
    _req_cuda(tvals, out)
    _chk(lib().mgld_timestep_embedding(_p(tvals), t_stride, _p(out), out.shape[0], out.shape[1], stream_ptr()),
         "timestep_embedding")
    return out


def nchw_to_nhwc(x, y, cpad):
    """x fp32 [n,c,h,w] -> y fp16 [n*h*w, ld] (channels c..cpad-1 zero)."""
    _req_cuda(x, y)
    n, c, h, w = x.shape
    _chk(lib().mgld_nchw_to_nhwc(_p(x), _p(y), n, c, h, w, cpad, _ld(y), stream_ptr()), "nchw_to_nhwc")
    return y


def nhwc_to_nchw(x, y):
    """x fp16/fp32 [n*h*w, ld] -> y fp32 [n,c,h,w]."""
    _req_cuda(x, y)
    n, c, h, w = y.shape
    _chk(lib().mgld_nhwc_to_nchw(_p(x), 1 if x.dtype == torch.float32 else 0, _ld(x), _p(y), n, c, h, w, stream_ptr()),
         "nhwc_to_nchw")
    return y


def tile_conv3p(wp, cin, tap_inner):
    """device form of engine.tile_conv3p: packed fp16 [N, 9*cin] -> the tiled [., 32] image (MgldIGemm.tap_inner = 2)"""
    _req_cuda(wp)
    assert wp.dtype == torch.float16 and wp.is_contiguous() and wp.shape[1] == 9 * cin
    n = wp.shape[0]
    out = torch.empty(((n + 63) // 64 * 64) * 9 * cin // 32, 32, dtype=torch.float16, device=wp.device)
    _chk(lib().mgld_tile_conv3p(_p(wp), n, cin, 1 if tap_inner else 0, _p(out), stream_ptr()), "tile_conv3p")
    return out


def copy2d(src, dst):
    _req_cuda(src, dst)
    _chk(lib().mgld_copy2d(_p(src), _ld(src), _p(dst), _ld(dst), C.c_int64(src.shape[0]), src.shape[1], stream_ptr()), "copy2d")
    return dst


def axpby_lo(x, x_lo, y, y_lo, a, b):
    """(y, y_lo) <- a (x, x_lo) + b (y, y_lo) on two-plane residual-stream tensors (x_lo may be None)"""
    _req_cuda(x, y, y_lo)
    _chk(lib().mgld_axpby_lo(_p(x), _lo_of(x_lo, x) if x_lo is not None else None, _ld(x), _p(y), _lo_of(y_lo, y), _ld(y), C.c_int64(x.shape[0]),
                             x.shape[1], C.c_float(a), C.c_float(b), stream_ptr()), "axpby_lo")


def axpby(x, y, a, b):
    _req_cuda(x, y)
    _chk(lib().mgld_axpby(_p(x), _ld(x), _p(y), _ld(y), C.c_int64(x.shape[0]), x.shape[1], C.c_float(a), C.c_float(b),
                          stream_ptr()), "axpby")
    return y


def ddpm_step(x, eps, noise, coef, step_idx, z, noise_step_stride=0):
    """noise: one [n,c,h,w] tensor (stride 0) or a stack indexed by the schedule index (stride = n*c*h*w)."""
    _req_cuda(x, eps, noise, coef, step_idx, z)
    n, c, h, w = x.shape
    ld_eps = 0 if eps.dim() == 4 else _ld(eps)   # 4-D eps = NCHW canvas
    _chk(lib().mgld_ddpm_step(_p(x), _p(eps), ld_eps, _p(noise), C.c_int64(noise_step_stride), _p(coef), _p(step_idx),
                              _p(z), n, c, h, w, stream_ptr()), "ddpm_step")
    return z


def flow_warp(x, flow, out):
    """x [n,c,h,w] fp32, flow [n,2,h,w] fp32 (dx,dy)."""
    _req_cuda(x, flow, out)
    n, c, h, w = x.shape
    _chk(lib().mgld_flow_warp(_p(x), _p(flow), _p(out), n, c, h, w, stream_ptr()), "flow_warp")
    return out


def guidance_work_bytes(T, c, h, w):
    return T * c * h * w * (8 + 4 + 4)


def guidance(z, ff, fb, focc, bocc, coef, step_idx, gscale, x_out, work):
    _req_cuda(z, ff, fb, focc, bocc, coef, step_idx, x_out, work)
    T, c, h, w = z.shape
    _chk(lib().mgld_guidance(_p(z), _p(ff), _p(fb), _p(focc), _p(bocc), _p(coef), _p(step_idx), C.c_float(gscale), _p(x_out),
                             _p(work), T, c, h, w, stream_ptr()), "guidance")
    return x_out


def guidance_loss(z, ff, fb, focc, bocc, loss_out, work):
    _req_cuda(z, ff, fb, focc, bocc, loss_out, work)
    T, c, h, w = z.shape
    _chk(lib().mgld_guidance_loss(_p(z), _p(ff), _p(fb), _p(focc), _p(bocc), _p(loss_out), _p(work), T, c, h, w, stream_ptr()),
         "guidance_loss")
    return loss_out


def step_advance(step_idx, delta):
    _chk(lib().mgld_step_advance(_p(step_idx), delta, stream_ptr()), "step_advance")


def step_timestep(coef, step_idx, tvals):
    _chk(lib().mgld_step_timestep(_p(coef), _p(step_idx), _p(tvals), tvals.numel(), stream_ptr()), "step_timestep")
    return tvals


def fb_consistency(fwd, bwd, alpha, beta, focc, bocc):
    _req_cuda(fwd, bwd, focc, bocc)
    n, _, h, w = fwd.shape
    _chk(lib().mgld_fb_consistency(_p(fwd), _p(bwd), C.c_float(alpha), C.c_float(beta), _p(focc), _p(bocc), n, h, w,
                                   stream_ptr()), "fb_consistency")
    return focc, bocc


def resize_flow(flow, out):
    _req_cuda(flow, out)
    n, _, h, w = flow.shape
    _chk(lib().mgld_resize_flow(_p(flow), _p(out), n, h, w, out.shape[2], out.shape[3], stream_ptr()), "resize_flow")
    return out


def adain(content, style, out, work):
    _req_cuda(content, style, out, work)
    n, c, h, w = content.shape
    with timed("adain", {"bytes": 4.0 * 3 * content.numel()}):
        _chk(lib().mgld_adain(_p(content), _p(style), _p(out), n * c, C.c_int64(h * w), C.c_float(1e-5), _p(work), stream_ptr()),
             "adain")
    return out


def init_latent(moments, noise, scale, n0=None, sqrt_ac=0.0, sqrt_one_minus_ac=0.0):
    """-> (init_latent, x_T or None): posterior sample * scale and, with n0, its q_sample at one timestep (include/mgld_hip.h)"""
    _req_cuda(moments, noise)
    n, c2, h, w = moments.shape
    c = c2 // 2
    assert moments.dtype == torch.float32 and moments.is_contiguous() and noise.shape == (n, c, h, w) and noise.is_contiguous()
    init = torch.empty_like(noise)
    xT = None
    if n0 is not None:
        _req_cuda(n0)
        assert n0.shape == noise.shape and n0.is_contiguous()
        xT = torch.empty_like(noise)
    _chk(lib().mgld_init_latent(_p(moments), _p(noise), _p(n0), _p(init), _p(xT), n, c, C.c_int64(h * w), C.c_float(scale),
                                C.c_float(sqrt_ac), C.c_float(sqrt_one_minus_ac), stream_ptr()), "init_latent")
    return init, xT


def to01(x, out=None):
    _req_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty_like(x) if out is None else out
    _chk(lib().mgld_to01(_p(x), _p(out), C.c_int64(x.numel()), stream_ptr()), "to01")
    return out


def wavelet_reconstruction(content, style, out, work):
    _req_cuda(content, style, out, work)
    n, c, h, w = content.shape
    _chk(lib().mgld_wavelet_reconstruction(_p(content), _p(style), _p(out), n * c, h, w, _p(work), stream_ptr()), "wavelet")
    return out


def crop(src, dst, y0, x0):
    _req_cuda(src, dst)
    n, c, H, W = src.shape
    _chk(lib().mgld_crop(_p(src), _p(dst), n, c, H, W, y0, x0, dst.shape[2], dst.shape[3], stream_ptr()), "crop")
    return dst


def tile_accumulate(tile, wgt, acc, cnt, y0, x0):
    _req_cuda(tile, wgt, acc, cnt)
    n, c, H, W = acc.shape
    _chk(lib().mgld_tile_accumulate(_p(tile), _p(wgt), _p(acc), _p(cnt), n, c, H, W, y0, x0, tile.shape[2], tile.shape[3],
                                    stream_ptr()), "tile_accumulate")


def copy_step(src_all, dst, step_idx):
    """dst <- src_all[step_idx[0]] (device-side index); src_all [S, ...], dst shaped like one slice"""
    _req_cuda(src_all, dst, step_idx)
    nbytes = dst.numel() * dst.element_size()
    assert src_all[0].numel() * src_all.element_size() == nbytes and src_all.is_contiguous() and dst.is_contiguous()
    _chk(lib().mgld_copy_step(_p(src_all), _p(dst), C.c_int64(nbytes), _p(step_idx), stream_ptr()), "copy_step")
    return dst


def tile_normalize(acc, cnt, out):
    _req_cuda(acc, cnt, out)
    _chk(lib().mgld_tile_normalize(_p(acc), _p(cnt), _p(out), C.c_int64(acc.numel()), stream_ptr()), "tile_normalize")
    return out


# ---- graph / events ---------------------------------------------------------------------------------------------
class Graph:
    """hipGraph captured from launches enqueued on the current torch stream between begin() and end()."""

    def __init__(self):
        self.exec = C.c_void_p(0)

    def begin(self):
        _chk(lib().mgld_graph_begin(stream_ptr()), "graph_begin")

    def end(self):
        _chk(lib().mgld_graph_end(stream_ptr(), C.byref(self.exec)), "graph_end")

    def launch(self):
        _chk(lib().mgld_graph_launch(self.exec, stream_ptr()), "graph_launch")

    def __del__(self):
        try:
            if self.exec and self.exec.value:
                lib().mgld_graph_destroy(self.exec)
        except Exception:
            pass


class Event:
    def __init__(self):
        self.ev = C.c_void_p(0)
        _chk(lib().mgld_event_create(C.byref(self.ev)), "event_create")

    def record(self):
        _chk(lib().mgld_event_record(self.ev, stream_ptr()), "event_record")

    def sync(self):
        _chk(lib().mgld_event_sync(self.ev), "event_sync")

    def elapsed_ms(self, stop):
        ms = C.c_float(0)
        _chk(lib().mgld_event_elapsed_ms(self.ev, stop.ev, C.byref(ms)), "event_elapsed")
        return ms.value

    def __del__(self):
        try:
            if self.ev and self.ev.value:
                lib().mgld_event_destroy(self.ev)
        except Exception:
            pass


def device_info(device=0):
    out = (C.c_int64 * 4)()
    _chk(lib().mgld_device_info(device, out), "device_info")
    return {"cus": out[0], "lds_per_cu": out[1], "clock_khz": out[2], "gfx": out[3]}


# ---- K11: pre/post-processing on the device ------------------------------------------------------------------------
def resize_bicubic(x, size, clamp=None):
    """x [n,c,h,w] fp32 -> [n,c,oh,ow]; F.interpolate(mode="bicubic", align_corners=False) semantics (+ optional clamp)."""
    _req_cuda(x)
    x = x.contiguous()
    n, c, h, w = x.shape
    oh, ow = int(size[0]), int(size[1])
    y = torch.empty(n, c, oh, ow, dtype=torch.float32, device=x.device)
    lo, hi = (-float("inf"), float("inf")) if clamp is None else clamp
    _chk(lib().mgld_resize_bicubic(_p(x), _p(y), n * c, h, w, oh, ow, C.c_float(lo), C.c_float(hi), stream_ptr()),
         "resize_bicubic")
    return y


def resize_center_crop(x, size):
    """torchvision.transforms.Resize(size) + CenterCrop(size) on a tensor [n,c,h,w] (fp32, device): the smaller edge is resized to
    `size` with bilinear interpolation (align_corners=False, no antialias: what torchvision 0.13/0.14 — the reference's pin —
    does for tensors), the larger edge to int(size * long / short), then the central size x size window is cut (round half to even
    on the offsets, as torchvision's center_crop)."""
    _req_cuda(x)
    x = x.contiguous()
    n, c, h, w = x.shape
    if w <= h:
        rw, rh = size, int(size * h / w)
    else:
        rh, rw = size, int(size * w / h)
    cy, cx = int(round((rh - size) / 2.0)), int(round((rw - size) / 2.0))
    y = torch.empty(n, c, size, size, dtype=torch.float32, device=x.device)
    _chk(lib().mgld_resize_bilinear_crop(_p(x), _p(y), n * c, h, w, rh, rw, size, size, cy, cx, stream_ptr()), "resize_bilinear_crop")
    return y


def reflect_pad(x, oh, ow):
    """F.pad(x, (0, ow-w, 0, oh-h), mode="reflect") for x [n,c,h,w] fp32"""
    _req_cuda(x)
    x = x.contiguous()
    n, c, h, w = x.shape
    y = torch.empty(n, c, oh, ow, dtype=torch.float32, device=x.device)
    _chk(lib().mgld_reflect_pad(_p(x), _p(y), n * c, h, w, oh, ow, stream_ptr()), "reflect_pad")
    return y


def replicate_pad(x, pad):
    """F.pad(x, pad=(left, right, top, bottom), mode="replicate") for x [n,c,h,w] fp32"""
    _req_cuda(x)
    x = x.contiguous()
    n, c, h, w = x.shape
    l, r, t, b = pad
    y = torch.empty(n, c, h + t + b, w + l + r, dtype=torch.float32, device=x.device)
    _chk(lib().mgld_replicate_pad(_p(x), _p(y), n * c, h, w, h + t + b, w + l + r, t, l, stream_ptr()), "replicate_pad")
    return y


def to_uint8_hwc(x, h=None, w=None):
    """x [n,c,H,W] fp32 in [0,1] -> uint8 [n,h,w,c] (top-left crop)"""
    _req_cuda(x)
    x = x.contiguous()
    n, c, H, W = x.shape
    h, w = h or H, w or W
    y = torch.empty(n, h, w, c, dtype=torch.uint8, device=x.device)
    _chk(lib().mgld_to_uint8_hwc(_p(x), _p(y), n, c, H, W, h, w, stream_ptr()), "to_uint8_hwc")
    return y


# ---- RAFT_SR (all fp32) -------------------------------------------------------------------------------------------------
class MgldConvF32(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("R", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int64),
        ("N", C.c_int32), ("Cin", C.c_int32), ("lda", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32),
        ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
        ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32),
        ("act", C.c_int32), ("post_relu", C.c_int32),
        ("alpha", C.c_float), ("batch", C.c_int32),
        ("strideA", C.c_int64), ("strideW", C.c_int64), ("strideC", C.c_int64),
    ]


def _f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError(f"RAFT kernels take fp32 tensors, got {t.dtype}")


def conv_f32(a, w, out, n, hin, win, cin, ksize=(1, 1), stride=1, pad=(0, 0), bias=None, act=ACT_NONE, alpha=1.0, resid=None,
             post_relu=False, batch=1, strideA=0, strideW=0, strideC=0, n_out=None):
    """fp32 implicit-GEMM convolution (mgld_conv_f32): a [n*hin*win, lda] NHWC view, w [N, kh*kw*Cin4] (pack_conv_f32),
    out [n*ho*wo, ldc] view.  batch > 1: LINEAR only, the operands of batch b start b*stride{A,W,C} floats further."""
    _req_cuda(a, w, out, bias, resid)
    _f32(a, w, out, bias, resid)
    kh, kw = ksize
    ho = (hin + 2 * pad[0] - kh) // stride + 1
    wo = (win + 2 * pad[1] - kw) // stride + 1
    p = MgldConvF32()
    p.A, p.W, p.C, p.bias, p.R = a.data_ptr(), w.data_ptr(), out.data_ptr(), _p(bias).value, _p(resid).value
    p.M, p.N, p.Cin = n * ho * wo, n_out or w.shape[0], cin
    assert w.shape[1] == kh * kw * ((cin + 3) // 4 * 4), (w.shape, ksize, cin)
    assert batch > 1 or out.shape[0] == p.M, (out.shape, p.M)
    p.lda, p.ldc, p.ldr = _ld(a), _ld(out), _ld(resid) if resid is not None else 0
    p.Hin, p.Win, p.Hout, p.Wout, p.kh, p.kw, p.stride, p.pad_t, p.pad_l = hin, win, ho, wo, kh, kw, stride, pad[0], pad[1]
    p.act, p.post_relu, p.alpha, p.batch = act, int(post_relu), alpha, batch
    p.strideA, p.strideW, p.strideC = strideA, strideW, strideC
    _chk(lib().mgld_conv_f32(C.byref(p), stream_ptr()), "conv_f32")
    return out


def instnorm_chunks(hw):
    return int(lib().mgld_instnorm_chunks(int(hw)))


def instnorm_f32(x, part, out, n, hw, eps, relu, skip=None):
    """InstanceNorm2d (no affine) on fp32 NHWC [n*hw, C] (+ ReLU, + relu(skip + y)); part: fp64 scratch [n, instnorm_chunks(hw), C, 2]"""
    _req_cuda(x, part, out, skip)
    _f32(x, out, skip)
    assert part.dtype == torch.float64 and part.numel() >= n * instnorm_chunks(hw) * x.shape[1] * 2
    _chk(lib().mgld_instnorm_f32(_p(x), _ld(x), _p(part), _p(skip), _ld(skip) if skip is not None else 0, _p(out), _ld(out), n, hw,
                                 x.shape[1], C.c_float(eps), int(relu), stream_ptr()), "instnorm_f32")
    return out


def nchw_to_nhwc_f32(x, out):
    """x fp32 [n,c,h,w] -> out fp32 [n*h*w, ld] (columns >= c zeroed)"""
    _req_cuda(x, out)
    _f32(x, out)
    x = x.contiguous()
    n, c, h, w = x.shape
    assert out.is_contiguous() and out.shape[0] == n * h * w
    _chk(lib().mgld_nchw_to_nhwc_f32(_p(x), _p(out), n, c, h * w, out.shape[1], stream_ptr()), "nchw_to_nhwc_f32")
    return out


def avgpool2(x):
    """x fp32 [planes, h, w] -> [planes, h//2, w//2]"""
    _req_cuda(x)
    pl, h, w = x.shape
    y = torch.empty(pl, h // 2, w // 2, dtype=torch.float32, device=x.device)
    _chk(lib().mgld_avgpool2(_p(x), _p(y), C.c_int64(pl), h, w, stream_ptr()), "avgpool2")
    return y


def corr_lookup(levels, coords, radius, out):
    """levels: list of fp32 [B*H*W, h_l, w_l]; coords fp32 [B,2,H,W]; out fp32 [B*H*W, >= nlev*(2r+1)^2]"""
    _req_cuda(coords, out, *levels)
    _f32(coords, out, *levels)
    n = len(levels)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in levels])
    hs = (C.c_int * n)(*[t.shape[1] for t in levels])
    ws = (C.c_int * n)(*[t.shape[2] for t in levels])
    B, _, H, W = coords.shape
    _chk(lib().mgld_corr_lookup(ptrs, hs, ws, n, _p(coords), B, H, W, radius, _p(out), _ld(out), stream_ptr()), "corr_lookup")
    return out


def gru_rh(r, hx, rhx, Ch):
    _req_cuda(r, hx, rhx)
    _f32(r, hx, rhx)
    _chk(lib().mgld_gru_rh(_p(r), _ld(r), _p(hx), _ld(hx), _p(rhx), _ld(rhx), C.c_int64(hx.shape[0]), Ch, hx.shape[1] - Ch,
                           stream_ptr()), "gru_rh")
    return rhx


def gru_gate(z, q, h):
    _req_cuda(z, q, h)
    _f32(z, q, h)
    _chk(lib().mgld_gru_gate(_p(z), _ld(z), _p(q), _ld(q), _p(h), _ld(h), C.c_int64(h.shape[0]), h.shape[1], stream_ptr()), "gru_gate")
    return h


def flow_update(coords1, coords0, delta, flow, mot=None, fin=None):
    _req_cuda(coords1, coords0, flow)
    _f32(coords1, coords0, delta, flow, mot, fin)
    B, _, H, W = coords1.shape
    _chk(lib().mgld_flow_update(_p(coords1), _p(coords0), _p(delta), _ld(delta) if delta is not None else 0, _p(flow),
                                _p(mot), _ld(mot) if mot is not None else 0, _p(fin), _ld(fin) if fin is not None else 0, B,
                                H * W, stream_ptr()), "flow_update")
    return flow


def convex_upsample(flow, mask):
    _req_cuda(flow, mask)
    _f32(flow, mask)
    B, _, H, W = flow.shape
    out = torch.empty(B, 2, 8 * H, 8 * W, dtype=torch.float32, device=flow.device)
    _chk(lib().mgld_convex_upsample(_p(flow), _p(mask), _ld(mask), _p(out), B, H, W, stream_ptr()), "convex_upsample")
    return out


# ---- high-precision first-stage encoder glue (csrc/hpenc.hip) -----------------------------------------------------------------
def hp_chunks(rows):
    return int(lib().mgld_hp_chunks(int(rows)))


def hp_gn_stats(x, frames, rows, groups, gsums):
    """x fp32 [frames*rows, C] view -> gsums fp64 [frames, hp_chunks(rows), groups, 2]"""
    _req_cuda(x, gsums)
    _f32(x)
    assert gsums.dtype == torch.float64 and gsums.numel() >= frames * hp_chunks(rows) * groups * 2
    _chk(lib().mgld_hp_gn_stats(_p(x), _ld(x), frames, rows, x.shape[1], groups, _p(gsums), stream_ptr()), "hp_gn_stats")
    return gsums


def hp_gn_split(x, gsums, eps, gamma, beta, silu, out, frames, rows, groups):
    """GroupNorm (+ SiLU) of fp32 x from hp_gn_stats' sums (gsums None: identity), written as the split-fp16 operand
    [yh | 16 yl | yh/256] (out fp16 [frames*rows, 3C]) or as fp32 (out fp32 [frames*rows, C])"""
    _req_cuda(x, gsums, gamma, beta, out)
    _f32(x, gamma, beta)
    of32 = out.dtype == torch.float32
    assert of32 or (out.dtype == torch.float16 and out.shape[1] == 3 * x.shape[1])
    _chk(lib().mgld_hp_gn_split(_p(x), _ld(x), _p(gsums), C.c_float(eps), _p(gamma), _p(beta), int(bool(silu)), _p(out), _ld(out),
                                int(of32), frames, rows, x.shape[1], groups, stream_ptr()), "hp_gn_split")
    return out


def hp_softmax_rows(S):
    """in-place row softmax of an fp32 [rows, cols] view"""
    _req_cuda(S)
    _f32(S)
    _chk(lib().mgld_hp_softmax_rows(_p(S), C.c_int64(S.shape[0]), S.shape[1], _ld(S), stream_ptr()), "hp_softmax_rows")
    return S
