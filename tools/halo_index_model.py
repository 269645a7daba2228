#!/usr/bin/env python3
"""Host model of the data movement of csrc/igemm_halo.hip (experimental): replays the slab DMA addressing (pixel / slot /
swizzled source chunk, out-of-image fill, K tail) and the per-tap A-fragment addressing of the kernel in numpy and checks
that every fragment a wave would feed to the MFMA holds exactly the bytes the im2col definition prescribes.
    python tools/halo_index_model.py          (no GPU needed; exits non-zero on the first mismatch)"""
import sys

import numpy as np


def check(B, H, W, C, c0, clen, ldx, seed=0, ups=False):
    rng = np.random.default_rng(seed)
    xs = rng.integers(-128, 128, size=(B * H * W // (4 if ups else 1), ldx), dtype=np.int8)
    # `x` below is the LOGICAL input the definition uses; with ups the kernel reads the half-resolution map xs instead
    x = (xs.reshape(B, H // 2, W // 2, ldx).repeat(2, axis=1).repeat(2, axis=2).reshape(B * H * W, ldx) if ups else xs)
    fill = np.full(16, 7, dtype=np.int8)                      # the "true zero" byte of out-of-image taps
    lw = {16: 4, 32: 5, 64: 6}[W]
    Wp, R = W + 2, 128 >> lw
    slabpix = (R + 2) * Wp
    nq = (slabpix + 15) >> 4
    nst = (clen + 63) // 64
    M = B * H * W
    for mb in range(M // 128):
        m0 = mb * 128
        b, y0 = m0 // (H * W), (m0 % (H * W)) >> lw
        for cs in range(nst):
            krem = clen - cs * 64
            lds = np.zeros(17 * 1024, dtype=np.int8)
            for q in range(nq):                               # the union of all waves' DMA instructions
                for lane in range(64):
                    pix, slot = q * 16 + (lane >> 2), lane & 3
                    py, px = divmod(pix, Wp)
                    iy, ix = y0 - 1 + py, px - 1
                    inslab = pix < slabpix
                    inimg = inslab and 0 <= iy < H and 0 <= ix < W
                    c = (slot ^ ((pix >> 2) & 3)) * 16
                    if c >= krem:
                        src = np.zeros(16, dtype=np.int8)
                    elif inimg:
                        a = c0 + cs * 64 + c
                        spix = (b * (H * W >> 2) + (iy >> 1) * (W >> 1) + (ix >> 1)) if ups else ((b * H + iy) * W + ix)
                        src = xs[spix, a:a + 16]
                    else:
                        src = fill if inslab else np.zeros(16, dtype=np.int8)
                    lds[q * 1024 + lane * 16: q * 1024 + lane * 16 + 16] = src
            for tap in range(9):
                dy, dx = divmod(tap, 3)
                toff = dy * Wp + dx
                for r in range(128):
                    p0 = (r >> lw) * Wp + (r & (W - 1))
                    pix = p0 + toff
                    sw = (pix >> 2) & 3
                    for L in range(4):                        # logical 16-byte chunk = ks * 2 + fhalf
                        got = lds[pix * 64 + ((L ^ sw) << 4): pix * 64 + ((L ^ sw) << 4) + 16]
                        iy, ix = y0 + (r >> lw) + dy - 1, (r & (W - 1)) + dx - 1
                        if L * 16 >= krem:
                            want = np.zeros(16, dtype=np.int8)
                        elif 0 <= iy < H and 0 <= ix < W:
                            a = c0 + cs * 64 + L * 16
                            want = x[(b * H + iy) * W + ix, a:a + 16]
                        else:
                            want = fill
                        if not np.array_equal(got, want):
                            print(f"MISMATCH B{B} H{H} W{W} block {mb} slab {cs} tap {tap} row {r} chunk {L}")
                            return False
    return True


if __name__ == "__main__":
    ok = True
    for cfg in [(1, 16, 16, 96, 0, 96, 96), (2, 32, 32, 64, 16, 48, 80), (1, 4, 64, 128, 0, 128, 128), (1, 8, 16, 80, 0, 80, 80)]:
        for ups in (False, True):
            r = check(*cfg, ups=ups)
            print(cfg, "ups" if ups else "   ", "ok" if r else "FAILED")
            ok &= r
    sys.exit(0 if ok else 1)
