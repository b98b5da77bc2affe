"""Host-side execution engine: weight packing for the kernel layouts, a bump-allocated activation arena,
and launch recording (eager first run -> replayable launch list -> hipGraph).

The reference drives one CUDA stream from a Python loop through nn.Module.forward.  Here the
nn.Modules only own parameters (checkpoint-compatible names); their forward() emits C-ABI launches on token-major
(NHWC) fp16 buffers.
"""
import torch

from . import hip


# ------------------------------------------------------------------------------------------------------------------
# weight packing (done once, on the host, in fp32 -> fp16)
# ------------------------------------------------------------------------------------------------------------------
def pack_conv3x3(w, cin_pad=None):
    """[Cout, Cin, 3, 3] -> [Cout, 9*Cin_pad] with K index = (ky*3+kx)*Cin_pad + c  (tap-major, channel contiguous)."""
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3
    cp = cin_pad or ((cin + 7) // 8 * 8)
    out = torch.zeros(cout, 9, cp, dtype=w.dtype)
    out[:, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, 9, cin)
    return out.reshape(cout, 9 * cp).contiguous()


def pack_conv1x1(w, cin_pad=None):
    """[Cout, Cin, 1, 1] or [Cout, Cin, 1] or [Cout, Cin] -> [Cout, Cin_pad]."""
    cout, cin = w.shape[0], w.shape[1]
    cp = cin_pad or ((cin + 7) // 8 * 8)
    out = torch.zeros(cout, cp, dtype=w.dtype)
    out[:, :cin] = w.reshape(cout, cin)
    return out.contiguous()


def pack_tconv3(w):
    """Conv3d weight [C, C, 3, 1, 1] -> [C, 3*C] with K index = dt*C + c."""
    c_out, c_in = w.shape[0], w.shape[1]
    return w[:, :, :, 0, 0].permute(0, 2, 1).reshape(c_out, 3 * c_in).contiguous()


def pack_geglu(w, b):
    """GEGLU projection [2*inner, dim]: interleave value/gate rows in blocks of 32 so that the igemm epilogue finds
    value j and gate j in the same lane (rows [64g, 64g+32) = values 32g.., rows [64g+32, 64g+64) = gates 32g..)."""
    two_inner, dim = w.shape
    inner = two_inner // 2
    assert inner % 32 == 0
    g = inner // 32
    wv, wg = w[:inner].reshape(g, 32, dim), w[inner:].reshape(g, 32, dim)
    wp = torch.stack([wv, wg], 1).reshape(two_inner, dim).contiguous()
    bp = None
    if b is not None:
        bp = torch.stack([b[:inner].reshape(g, 32), b[inner:].reshape(g, 32)], 1).reshape(two_inner).contiguous()
    return wp, bp
