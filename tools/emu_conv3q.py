#!/usr/bin/env python
"""CPU emulation of conv3q_kernel's ADDRESS arithmetic (csrc/igemm.hip): the patch DMA image in LDS (lane-linear 1-KiB pieces, source-side
chunk swizzle), the nine per-lane tap offsets (incl. the nearest-2x fold), the tiled weight pieces, and the fragment reads — checked
element by element against the definition of the convolution.  No GPU needed; scratch tool + used by tests/test_dryrun_cpu.py."""
import itertools
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mgld_vsr_amd.engine import pack_conv3x3, tile_conv3p  # noqa: E402

PB = 64


def check(TY, TX, BN, WM, WN, UP2, frames, Hin, Win, Cin, N, tile_m, tile_n, h=1):
    sc = 2 if UP2 else 1
    Hout, Wout = sc * Hin, sc * Win
    BM = TY * TX
    WAVES_N = BN // WN
    NW = (BM // WM) * WAVES_N
    MI, NI = WM // 32, WN // 32
    PW, PH = (TX // 2 + 2, TY // 2 + 2) if UP2 else (TX + 2, TY + 2)
    PR = PW * PH
    NPA = (PR + 15) // 16
    A_BYTES = NPA * 1024
    BSUB = BN * PB
    B_BYTES = 3 * BSUB
    NPB = 3 * BN // 16
    nh = Cin // 32
    tiles_x, tiles_y = -(-Wout // TX), -(-Hout // TY)
    tpf = tiles_x * tiles_y
    frame, trem = divmod(tile_m, tpf)
    tyi, txi = divmod(trem, tiles_x)
    y0, x0 = tyi * TY, txi * TX
    bn0 = tile_n * BN
    # identities: A element id = flat index + 1 (0 = zero page); weight id likewise on the [N, 3, 3, Cin] tensor
    A = (np.arange(frames * Hin * Win * Cin, dtype=np.int64) + 1).reshape(frames, Hin, Win, Cin)
    Wt = torch.arange(N * Cin * 9, dtype=torch.float64).reshape(N, Cin, 3, 3) + 1          # [Cout, Cin, ky, kx]
    wp = pack_conv3x3(Wt, Cin, tap_inner=False)
    tiled = tile_conv3p(wp, Cin, False).reshape(-1).numpy().astype(np.int64)
    lds = np.zeros((2 * A_BYTES + 2 * B_BYTES) // 2, dtype=np.int64)   # one slot per fp16 element
    # ---- patch DMA (buffer 0), slice h ----
    yb, xb = ((y0 >> 1) if UP2 else y0) - 1, ((x0 >> 1) if UP2 else x0) - 1
    for q in range(NPA):
        for lane in range(64):
            j = q * 16 + (lane >> 2)
            pr, pc = divmod(j, PW)
            y, x = yb + pr, xb + pc
            ok = j < PR and 0 <= y < Hin and 0 <= x < Win
            cl = (lane & 3) ^ ((j >> 2) & 3)
            src = A[frame, y, x, h * 32 + cl * 8: h * 32 + cl * 8 + 8] if ok else np.zeros(8, np.int64)
            dst = (q * 1024 + lane * 16) // 2
            lds[dst:dst + 8] = src
    # ---- weight DMA (buffer 0), stage dyi ----
    errs = 0
    for dyi in range(3):
        for b in range(NPB):
            dxi, rb = divmod(b, BN // 16)
            g64 = (bn0 >> 6) + (rb >> 2)
            ok = g64 * 64 < ((N + 63) & ~63)
            for lane in range(64):
                off = (((g64 * nh * 3 * 4 + (rb & 3)) * 3 + dxi) * 512 + lane * 8) + (h * 3 + dyi) * (12 * 512)
                src = tiled[off:off + 8] if ok else np.zeros(8, np.int64)
                dst = (2 * A_BYTES + b * 1024 + lane * 16) // 2
                lds[dst:dst + 8] = src
        # ---- fragment reads of stage dyi ----
        for wave, l31, lhi in itertools.product(range(NW), range(32), range(2)):
            wm, wn = divmod(wave, WAVES_N)
            for mi in range(MI):
                r = wm * WM + mi * 32 + l31
                ty, tx = divmod(r, TX)
                for dxi in range(3):
                    if UP2:
                        j = (((ty + dyi - 1) >> 1) + 1) * PW + ((tx + dxi - 1) >> 1) + 1
                    else:
                        j = (ty + dyi) * PW + tx + dxi
                    a_off = j * PB + ((lhi ^ ((j >> 2) & 3)) << 4)
                    for ks in range(2):
                        got = lds[(a_off ^ (ks << 5)) // 2:(a_off ^ (ks << 5)) // 2 + 8]
                        oy, ox = y0 + ty + dyi - 1, x0 + tx + dxi - 1          # coordinate in the (virtually upsampled) input
                        c0 = h * 32 + (2 * ks + lhi) * 8
                        if 0 <= oy < Hout and 0 <= ox < Wout:
                            want = A[frame, oy // sc, ox // sc, c0:c0 + 8]
                        else:
                            want = np.zeros(8, np.int64)
                        if y0 + ty < Hout and x0 + tx < Wout and not np.array_equal(got, want):   # rows of ragged tiles are masked later
                            errs += 1
            for ni in range(NI):
                r = wn * WN + ni * 32 + l31
                w_off = r * PB + ((lhi ^ ((r >> 2) & 3)) << 4)
                for dxi in range(3):
                    for ks in range(2):
                        a = (2 * A_BYTES + dxi * BSUB + (w_off ^ (ks << 5))) // 2
                        got = lds[a:a + 8]
                        n = bn0 + r
                        c0 = h * 32 + (2 * ks + lhi) * 8
                        want = Wt[n, c0:c0 + 8, dyi, dxi].numpy().astype(np.int64) if n < N else np.zeros(8, np.int64)
                        if not np.array_equal(got, want):
                            errs += 1
    return errs


def lds_conflicts(TY, TX, WM, UP2):
    """extra LDS cycles of the activation fragment ds_read_b128 (16-lane service groups of MI355X_MICROARCH: {0-3,12-15,20-27},
    {4-11,16-19,28-31} per half-wave); returns the worst (max distinct 16-B slots hit more than once) over taps"""
    PW = TX // 2 + 2 if UP2 else TX + 2
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    worst = 0
    for dyi, dxi in itertools.product(range(3), range(3)):
        for grp in groups:
            slots = {}
            for l31 in grp:
                ty, tx = divmod(l31, TX)
                j = ((((ty + dyi - 1) >> 1) + 1) * PW + ((tx + dxi - 1) >> 1) + 1) if UP2 else ((ty + dyi) * PW + tx + dxi)
                addr = j * PB + ((0 ^ ((j >> 2) & 3)) << 4)
                slots.setdefault((addr // 16) % 16, set()).add(addr)
            worst = max(worst, max(len(v) for v in slots.values()) - 1)
    return worst


if __name__ == "__main__":
    cases = [(8, 16, 64, 32, 32, False), (16, 16, 64, 64, 32, False), (8, 16, 128, 64, 32, False), (16, 16, 128, 64, 64, False),
             (8, 32, 64, 64, 32, False), (8, 16, 64, 64, 32, False), (8, 16, 64, 32, 32, True), (16, 16, 64, 64, 32, True),
             (8, 8, 128, 32, 32, False)]
    for c in cases:
        TY, TX, BN, WM, WN, UP2 = c
        sc = 2 if UP2 else 1
        Hin, Win = (24 // sc) * 1, (40 // sc) * 1          # ragged in both directions for every tile shape
        if TX == 8:
            Hin = Win = 8
        tiles = (-(-Hin * sc // TY)) * (-(-Win * sc // TX))
        e = sum(check(*c, frames=2, Hin=Hin, Win=Win, Cin=64, N=96, tile_m=tm, tile_n=tn) for tm in (0, tiles - 1, tiles, 2 * tiles - 1)
                for tn in range(-(-96 // BN)))
        print(c, "address errors:", e, " worst extra LDS cycles per A read:", lds_conflicts(TY, TX, WM, UP2))
