// igemm_i8.hip — K3/K4: integer implicit-GEMM convolution / linear on MFMA_I32_32x32x32_I8.
//
// Replaces F.conv2d / F.conv1d(k=1) / F.linear on fake-quantised fp32 operands
// (reference qdiff/quant_layer.py:276) with an exact int32 contraction of the stored codes and a
// fused dequantising epilogue (per-out-channel scale, zero-point restoration, bias, timestep-
// embedding row bias, residual, split-shortcut second segment: quant_layer.py:257-269).
//
// GEMM view:  M = B*Ho*Wo output pixels, N = Cout, K = taps * channels (K-contiguous for both
// operands: activations are NHWC int8, weights are [Cout][tap][channel]).
// Tile: BM x BN x 64 per K-step, 256 threads = 4 waves in a 2x2 grid, each wave owns
// (BM/2)x(BN/2) as 32x32 MFMA tiles.  A K-step never straddles a tap or a segment, so the
// im2col gather is one (ih,iw) computation per row per tap; out-of-image taps load the
// "true zero" byte z' (conv padding happens in the dequantised domain in the reference).
// Staging: global -> VGPR -> LDS (register staging is forced: A needs the padding fill and the
// row-sum side computation, B may need the int4 unpack), double-buffered LDS, one barrier per
// K-step; loads of step i+1 are issued before the MFMAs of step i.
// LDS rows are 64 B; the 16-B chunk index is XORed with (row>>2)&3 so that the four rows a
// ds_read_b128 lane group maps to one 256-B bank row land on distinct 16-B slots (conflict-free).
#include "common.h"

namespace {

struct SegK {
    int c0, clen, kofs, nsteps_tap;
    const int8_t* wzp;
    const float* scale;
    const int*   zc;
    const int*   zw;
    const int*   zfill;
};

struct ConvK {
    const int8_t*  x;
    const uint8_t* w;
    void*          out;
    int32_t*       iout;
    const float*   bias;
    const float*   rowbias;
    const void*    residual;
    long ldx, ldk, ldo, ldr, ldrb;
    int B, H, W, Ho, Wo, Cout, kh, kw, stride, pad_t, pad_l;
    int M, taps, nseg, nblk_m, nblk_n;
    SegK seg[2];
};

enum { OUT_F32 = 0, OUT_F16 = 1, OUT_I32 = 2 };

__device__ __forceinline__ int bytesum16(const v4i& v) {
    int s = __builtin_amdgcn_sdot4(v.x, 0x01010101, 0, false);
    s = __builtin_amdgcn_sdot4(v.y, 0x01010101, s, false);
    s = __builtin_amdgcn_sdot4(v.z, 0x01010101, s, false);
    s = __builtin_amdgcn_sdot4(v.w, 0x01010101, s, false);
    return s;
}

template <int BM, int BN, int WB, bool HAS_ZW, bool SPLIT, int OUT>
__global__ __launch_bounds__(256) void igemm_kernel(const ConvK p) {
    constexpr int WM = BM / 2, WN = BN / 2;      // wave tile
    constexpr int MT = WM / 32, NT = WN / 32;    // 32x32 MFMA tiles per wave
    constexpr int AL = BM / 64, BL = BN / 64;    // 16-B chunks per thread per K-step
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_BYTES + 2 * B_BYTES + 3 * BM * 4];
    unsigned char* sA   = smem;
    unsigned char* sB   = smem + 2 * A_BYTES;
    int*           sRowB = reinterpret_cast<int*>(smem + 2 * A_BYTES + 2 * B_BYTES);
    int*           sAsum = sRowB + BM;  // [2][BM]

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int nblk    = p.nblk_m * p.nblk_n;
    const int logical = qd_xcd_remap(blockIdx.x, nblk);
    const int mb = logical / p.nblk_n, nb = logical % p.nblk_n;
    const int m0 = mb * BM, n0 = nb * BN;

    // ---- per-thread loader rows -------------------------------------------------------------
    const int kc   = tid & 3;
    const int lrow = tid >> 2;  // 0..63
    long a_base[AL];
    int  a_ih0[AL], a_iw0[AL];
    bool a_valid[AL];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        int m      = m0 + lrow + i * 64;
        a_valid[i] = m < p.M;
        int mm = a_valid[i] ? m : 0;
        int b  = mm / HoWo;
        int r  = mm - b * HoWo;
        int ho = r / p.Wo;
        int wo = r - ho * p.Wo;
        a_base[i] = (long)b * p.H * p.W * p.ldx;
        a_ih0[i]  = ho * p.stride - p.pad_t;
        a_iw0[i]  = wo * p.stride - p.pad_l;
        if (kc == 0) sRowB[lrow + i * 64] = b;
    }
    long b_base[BL];
    bool b_valid[BL];
    unsigned b_zp4[BL];
    auto load_wzp = [&](int s) {
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            int n = n0 + lrow + i * 64;
            b_zp4[i] = 0;
            if (WB == 4 && p.seg[s].wzp && n < p.Cout)
                b_zp4[i] = (unsigned)(unsigned char)p.seg[s].wzp[n] * 0x01010101u;
        }
    };
#pragma unroll
    for (int i = 0; i < BL; ++i) {
        int n      = n0 + lrow + i * 64;
        b_valid[i] = n < p.Cout;
        int nn     = b_valid[i] ? n : 0;
        b_base[i]  = (long)nn * p.taps * p.ldk;
    }
    load_wzp(0);

    // ---- uniform loader state ---------------------------------------------------------------
    int ls = 0, ltap = 0, lr = 0, lq = 0, lc = 0;
    int fillw = 0;  // z' replicated into 4 bytes for the current loader segment
    {
        const int* zf = p.seg[0].zfill;
        int z = zf ? zf[0] : 0;
        fillw = (int)((unsigned)(z & 0xff) * 0x01010101u);
    }
    int asum_part[AL], asum_done[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) { asum_part[i] = 0; asum_done[i] = 0; }

    v4i areg[AL], breg[BL];

    auto load_step = [&]() {
        const SegK& sg = p.seg[ls];
        const int kcol = lc * 64 + kc * 16;
        const bool kvalid = kcol < sg.clen;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            v4i v = {0, 0, 0, 0};
            if (kvalid && a_valid[i]) {
                int ih = a_ih0[i] + lr, iw = a_iw0[i] + lq;
                if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) {
                    const int8_t* src = p.x + a_base[i] + ((long)ih * p.W + iw) * p.ldx + sg.c0 + kcol;
                    v = *reinterpret_cast<const v4i*>(src);
                } else {
                    v = (v4i){fillw, fillw, fillw, fillw};
                }
            }
            areg[i] = v;
            if (HAS_ZW) asum_part[i] += bytesum16(v);
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            v4i v = {0, 0, 0, 0};
            if (kvalid && b_valid[i]) {
                long off = b_base[i] + (long)ltap * p.ldk + sg.kofs + kcol;
                if (WB == 8) {
                    v = *reinterpret_cast<const v4i*>(p.w + off);
                } else {
                    uint2 pk = *reinterpret_cast<const uint2*>(p.w + (off >> 1));
                    unsigned o0 = pk.x & 0x0F0F0F0Fu, o1 = (pk.x >> 4) & 0x0F0F0F0Fu;
                    unsigned o2 = pk.y & 0x0F0F0F0Fu, o3 = (pk.y >> 4) & 0x0F0F0F0Fu;
                    const unsigned z4 = b_zp4[i];
                    o0 = ((o0 | 0x80808080u) - z4) ^ 0x80808080u;
                    o1 = ((o1 | 0x80808080u) - z4) ^ 0x80808080u;
                    o2 = ((o2 | 0x80808080u) - z4) ^ 0x80808080u;
                    o3 = ((o3 | 0x80808080u) - z4) ^ 0x80808080u;
                    v = (v4i){(int)o0, (int)o1, (int)o2, (int)o3};
                }
            }
            breg[i] = v;
        }
    };

    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            int row = lrow + i * 64;
            int slot = kc ^ ((row >> 2) & 3);
            *reinterpret_cast<v4i*>(sA + buf * A_BYTES + row * 64 + slot * 16) = areg[i];
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            int row = lrow + i * 64;
            int slot = kc ^ ((row >> 2) & 3);
            *reinterpret_cast<v4i*>(sB + buf * B_BYTES + row * 64 + slot * 16) = breg[i];
        }
    };

    auto advance = [&]() {
        ++lc;
        if (lc == p.seg[ls].nsteps_tap) {
            lc = 0; ++ltap; ++lq;
            if (lq == p.kw) { lq = 0; ++lr; }
            if (ltap == p.taps) {
                ltap = 0; lr = 0; lq = 0; ++ls;
                if (ls < p.nseg) {
                    const int* zf = p.seg[ls].zfill;
                    int z = zf ? zf[0] : 0;
                    fillw = (int)((unsigned)(z & 0xff) * 0x01010101u);
                    load_wzp(ls);
#pragma unroll
                    for (int i = 0; i < AL; ++i) { asum_done[i] = asum_part[i]; asum_part[i] = 0; }
                }
            }
        }
    };

    auto publish_asum = [&](int slot, const int (&part)[AL]) {
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            int v = part[i];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            if (kc == 0) sAsum[slot * BM + lrow + i * 64] = v;
        }
    };

    // ---- accumulators -----------------------------------------------------------------------
    v16i acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    float facc[SPLIT ? MT : 1][SPLIT ? NT : 1][16];
    if (SPLIT) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) facc[SPLIT ? i : 0][SPLIT ? j : 0][r] = 0.f;
    }

    const int nst0  = p.taps * p.seg[0].nsteps_tap;
    const int total = nst0 + (p.nseg == 2 ? p.taps * p.seg[1].nsteps_tap : 0);

    // contribution of one finished segment: float(acc - zc - zw*(asum-kz)) * scale  (or raw int)
    auto seg_value = [&](int s, int mt, int nt, int r, int n, bool nok, float scale_n, int zc_n, int zw_n,
                         int kz) -> int {
        (void)s; (void)n; (void)nok; (void)scale_n;
        int I = acc[mt][nt][r] - zc_n;
        if (HAS_ZW) {
            int rowl = wm * WM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            I -= zw_n * (sAsum[s * BM + rowl] - kz);
        }
        return I;
    };

    // ---- main loop --------------------------------------------------------------------------
    load_step();
    store_lds(0);
    advance();
    __syncthreads();

    const int frow = lane & 31, fhalf = lane >> 5;
    for (int it = 0; it < total; ++it) {
        const bool has_next = it + 1 < total;
        if (has_next) load_step();

        const unsigned char* cA = sA + (it & 1) * A_BYTES;
        const unsigned char* cB = sB + (it & 1) * B_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v4i af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                int row = wm * WM + i * 32 + frow;
                int slot = (ks * 2 + fhalf) ^ ((row >> 2) & 3);
                af[i] = *reinterpret_cast<const v4i*>(cA + row * 64 + slot * 16);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                int row = wn * WN + j * 32 + frow;
                int slot = (ks * 2 + fhalf) ^ ((row >> 2) & 3);
                bf[j] = *reinterpret_cast<const v4i*>(cB + row * 64 + slot * 16);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
        }

        if (SPLIT && p.nseg == 2 && it == nst0 - 1) {
            // end of segment 0: fold it into the float accumulator with its own scales
            if (HAS_ZW) {
                publish_asum(0, asum_done);
                __syncthreads();
            }
            const SegK& sg = p.seg[0];
            const int kz = sg.zfill ? sg.zfill[1] : 0;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                int n = n0 + wn * WN + j * 32 + frow;
                bool nok = n < p.Cout;
                float sc = nok ? sg.scale[n] : 0.f;
                int zc_n = (nok && sg.zc) ? sg.zc[n] : 0;
                int zw_n = (HAS_ZW && nok && sg.zw) ? sg.zw[n] : 0;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int I = seg_value(0, i, j, r, n, nok, sc, zc_n, zw_n, kz);
                        facc[SPLIT ? i : 0][SPLIT ? j : 0][r] = (float)I * sc;
                        acc[i][j][r] = 0;
                    }
            }
        }

        if (has_next) {
            store_lds((it + 1) & 1);
            advance();
        }
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------
    const int last = p.nseg - 1;
    if (HAS_ZW) {
        publish_asum(last, asum_part);
        __syncthreads();
    }
    const SegK& sg = p.seg[last];
    const int kz = sg.zfill ? sg.zfill[1] : 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int n = n0 + wn * WN + j * 32 + frow;
        bool nok = n < p.Cout;
        float sc = nok ? sg.scale[n] : 0.f;
        int zc_n = (nok && sg.zc) ? sg.zc[n] : 0;
        int zw_n = (HAS_ZW && nok && sg.zw) ? sg.zw[n] : 0;
        float bias_n = (nok && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int rowl = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                int m = m0 + rowl;
                if (!nok || m >= p.M) continue;
                int I = seg_value(last, i, j, r, n, nok, sc, zc_n, zw_n, kz);
                if (OUT == OUT_I32) {
                    p.iout[(long)m * p.Cout + n] = I;
                    continue;
                }
                float v = (float)I * sc;
                if (SPLIT) v += facc[SPLIT ? i : 0][SPLIT ? j : 0][r];
                v += bias_n;
                if (p.rowbias) v += p.rowbias[(long)sRowB[rowl] * p.ldrb + n];
                if (OUT == OUT_F32) {
                    if (p.residual) v += reinterpret_cast<const float*>(p.residual)[(long)m * p.ldr + n];
                    reinterpret_cast<float*>(p.out)[(long)m * p.ldo + n] = v;
                } else {
                    if (p.residual) v += __half2float(reinterpret_cast<const __half*>(p.residual)[(long)m * p.ldr + n]);
                    reinterpret_cast<__half*>(p.out)[(long)m * p.ldo + n] = __float2half(v);
                }
            }
    }
}

template <int BM, int BN, int WB, bool HAS_ZW, bool SPLIT, int OUT>
void launch(const ConvK& k, hipStream_t st) {
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WB, HAS_ZW, SPLIT, OUT>), dim3(k.nblk_m * k.nblk_n), dim3(256), 0, st, k);
}

template <int BM, int BN>
int dispatch(ConvK& k, int wbits, bool has_zw, bool split, int out, hipStream_t st) {
    k.nblk_m = (k.M + BM - 1) / BM;
    k.nblk_n = (k.Cout + BN - 1) / BN;
#define QD_CASE(WB, ZW, SP, O)                                                                  \
    if (wbits == WB && has_zw == ZW && split == SP && out == O) {                               \
        launch<BM, BN, WB, ZW, SP, O>(k, st);                                                   \
        return 0;                                                                               \
    }
    QD_CASE(8, false, false, OUT_F32) QD_CASE(8, true, false, OUT_F32)
    QD_CASE(8, false, true, OUT_F32)  QD_CASE(8, true, true, OUT_F32)
    QD_CASE(4, false, false, OUT_F32) QD_CASE(4, false, true, OUT_F32)
    QD_CASE(8, false, false, OUT_F16) QD_CASE(8, true, false, OUT_F16)
    QD_CASE(8, false, true, OUT_F16)  QD_CASE(8, true, true, OUT_F16)
    QD_CASE(4, false, false, OUT_F16) QD_CASE(4, false, true, OUT_F16)
    QD_CASE(8, false, false, OUT_I32) QD_CASE(8, true, false, OUT_I32)
    QD_CASE(4, false, false, OUT_I32)
#undef QD_CASE
    qd_set_error("qd_conv2d_i8: unsupported variant wbits=%d zw=%d split=%d out=%d", wbits, (int)has_zw, (int)split, out);
    return 1;
}

int run(const qd_conv_desc* d, int32_t* iout, void* stream) {
    QD_REQUIRE(d != nullptr, "qd_conv2d_i8: null descriptor");
    QD_REQUIRE(d->x && d->w && (d->out || iout), "qd_conv2d_i8: null tensor pointer");
    QD_REQUIRE(d->wbits == 8 || d->wbits == 4, "qd_conv2d_i8: wbits must be 4 or 8 (got %d)", d->wbits);
    QD_REQUIRE(d->out_dtype == QD_F32 || d->out_dtype == QD_F16, "qd_conv2d_i8: out_dtype must be f32/f16");
    QD_REQUIRE(d->nseg == 1 || d->nseg == 2, "qd_conv2d_i8: nseg must be 1 or 2");
    QD_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0, "qd_conv2d_i8: bad shape");
    QD_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0, "qd_conv2d_i8: bad kernel/stride");
    QD_REQUIRE((long)d->B * d->Ho * d->Wo < (1L << 31), "qd_conv2d_i8: M overflows int32");
    if (d->w_tiled) return qd_conv2d_i8_tiled(d, iout, stream);
    QD_REQUIRE(d->nseg == 1 || d->nseg == 2, "qd_conv2d_i8: nseg must be 1 or 2");
    QD_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0, "qd_conv2d_i8: bad shape");
    QD_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0, "qd_conv2d_i8: bad kernel/stride");
    QD_REQUIRE(d->ldx % 16 == 0 && d->ldk % 16 == 0, "qd_conv2d_i8: ldx and ldk must be multiples of 16 (ldx=%ld ldk=%ld)", (long)d->ldx, (long)d->ldk);
    QD_REQUIRE(qd_aligned(d->x, 16) && qd_aligned(d->w, 16), "qd_conv2d_i8: x/w must be 16-byte aligned");
    QD_REQUIRE((long)d->B * d->Ho * d->Wo < (1L << 31), "qd_conv2d_i8: M overflows int32");
    QD_REQUIRE(d->out_dtype == QD_F32 || d->out_dtype == QD_F16, "qd_conv2d_i8: out_dtype must be f32/f16");
    ConvK k{};
    k.x = d->x; k.w = d->w; k.out = d->out; k.iout = iout;
    k.bias = d->bias; k.rowbias = d->rowbias; k.residual = d->residual;
    k.ldx = d->ldx; k.ldk = d->ldk; k.ldo = d->ldo; k.ldr = d->ldr; k.ldrb = d->ld_rowbias;
    k.B = d->B; k.H = d->H; k.W = d->W; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout;
    k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad_t = d->pad_t; k.pad_l = d->pad_l;
    k.M = d->B * d->Ho * d->Wo; k.taps = d->kh * d->kw; k.nseg = d->nseg;
    bool has_zw = false;
    for (int s = 0; s < d->nseg; ++s) {
        const qd_conv_seg& g = d->seg[s];
        QD_REQUIRE(g.clen > 0 && g.clen % 16 == 0 && g.c0 % 16 == 0 && g.kofs % 16 == 0,
                   "qd_conv2d_i8: segment %d needs c0/clen/kofs multiples of 16 (c0=%d clen=%d kofs=%d)", s, g.c0, g.clen, g.kofs);
        QD_REQUIRE(g.c0 + g.clen <= d->ldx && g.kofs + g.clen <= d->ldk, "qd_conv2d_i8: segment %d exceeds row", s);
        QD_REQUIRE(g.scale != nullptr, "qd_conv2d_i8: segment %d has no scale vector", s);
        k.seg[s].c0 = g.c0; k.seg[s].clen = g.clen; k.seg[s].kofs = g.kofs;
        k.seg[s].nsteps_tap = (g.clen + 63) / 64;
        k.seg[s].wzp = (d->wbits == 4) ? g.wzp : nullptr;
        k.seg[s].scale = g.scale; k.seg[s].zc = g.zc; k.seg[s].zw = g.zw; k.seg[s].zfill = g.zfill;
        has_zw = has_zw || (g.zw != nullptr);
    }
    QD_REQUIRE(!(d->wbits == 4 && has_zw), "qd_conv2d_i8: int4 weights carry their zero point in wzp, not zw");
    const bool split = d->nseg == 2;
    const int out = iout ? OUT_I32 : (d->out_dtype == QD_F16 ? OUT_F16 : OUT_F32);
    QD_REQUIRE(!(iout && split), "qd_conv2d_i8_acc: single segment only");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // tile choice: the big tile when it still yields at least ~one wave of workgroups per CU pair
    const long blocks_big = (long)((k.M + 127) / 128) * ((k.Cout + 127) / 128);
    int rc;
    if (blocks_big >= 192 && k.Cout > 64)
        rc = dispatch<128, 128>(k, d->wbits, has_zw, split, out, st);
    else
        rc = dispatch<64, 64>(k, d->wbits, has_zw, split, out, st);
    if (rc) return rc;
    QD_LAUNCH_CHECK("qd_conv2d_i8");
    return 0;
}

}  // namespace

extern "C" int qd_conv2d_i8(const qd_conv_desc* d, void* stream) { return run(d, nullptr, stream); }
extern "C" int qd_conv2d_i8_acc(const qd_conv_desc* d, int32_t* iout, void* stream) {
    if (!iout) { qd_set_error("qd_conv2d_i8_acc: null iout"); return 1; }
    return run(d, iout, stream);
}
