// bmm_i8.hip — the two attention contractions as STANDALONE batched integer GEMMs.
//
// The fused kernel (attn_i8.hip) never materialises the T x S score matrix.  The reference's API, however,
// also exposes the two matmuls as separate modules — QuantQKMatMul.forward(q, k) returns the scores,
// the caller applies softmax, QuantSMVMatMul.forward(weight, v) consumes the probabilities
// (qdiff/quant_block.py:114-160, ldm openaimodel.py:384-406) — and a drop-in has to honour that contract when
// the modules are used on their own.  Same operand layouts, zero-point algebra and MFMA tiling as attn_i8.hip:
//   qd_bmm_qk_i8 : out[bh][t][s] = cs * sum_d (q'-zq')(k'-zk')            (exact int32, one float multiply)
//   qd_bmm_pv_i8 : out[bh][c][t] = dw*dv * sum_s (u[t][s]-zpw)(v'[c][s]-zv'),  u = clamp(rint(w/dw)+zpw)
#include "common.h"
#include <type_traits>

namespace {

struct QkK {
    const int8_t* q;
    const int8_t* k;
    const float* prm;       // 0 cs | 1 zq' | 2 zk'
    float* out;
    long ldo, bstride;
    int T, S, d, Tpad, Spad, dpad;
};

// all bytes = c, for c in [-128, 127]; values outside are split by the caller
__device__ __forceinline__ v4i splat_bytes(int c) {
    const int w = (c & 0xff) * 0x01010101;
    return v4i{w, w, w, w};
}

// one wave = 32 queries x all keys.  C = A(query rows) x B(key rows): col = lane&31 = key, rows = queries.
// The zero-point terms ride on the matrix pipe: (-zk') * qsum_i = MFMA(q rows, const), (-zq') * ksum_j =
// MFMA(const, k rows); the K-independent d*zq'*zk' seeds the accumulator.
template <int DT>
__global__ __launch_bounds__(256) void bmm_qk_kernel(const QkK p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (q0 >= p.T) return;
    const float cs = p.prm[0];
    const int zq = (int)p.prm[1], zk = (int)p.prm[2];
    // -z in [-127, 128]: 128 does not fit a signed byte -> two constants
    const int nq1 = -zq > 127 ? 64 : -zq, nq2 = -zq - nq1;
    const int nk1 = -zk > 127 ? 64 : -zk, nk2 = -zk - nk1;
    const v4i cq1 = splat_bytes(nq1), cq2 = splat_bytes(nq2), ck1 = splat_bytes(nk1), ck2 = splat_bytes(nk2);
    const int seed = p.d * zq * zk;

    v4i qf[DT];
    const int8_t* qrow = p.q + ((long)bh * p.Tpad + q0 + frow) * p.dpad + half * 16;
#pragma unroll
    for (int kk = 0; kk < DT; ++kk) qf[kk] = *reinterpret_cast<const v4i*>(qrow + kk * 32);
    const int8_t* kbase = p.k + ((long)bh * p.Spad + frow) * p.dpad + half * 16;
    float* obase = p.out + (long)bh * p.bstride;
    const int ntile = p.Spad >> 5;
    for (int jt = 0; jt < ntile; ++jt) {
        v16i acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = seed;
#pragma unroll
        for (int kk = 0; kk < DT; ++kk) {
            const v4i kf = *reinterpret_cast<const v4i*>(kbase + (long)jt * 32 * p.dpad + kk * 32);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(qf[kk], kf, acc, 0, 0, 0);
            if (zq != 0) {
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(cq1, kf, acc, 0, 0, 0);
                if (nq2 != 0) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(cq2, kf, acc, 0, 0, 0);
            }
            if (zk != 0) {
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(qf[kk], ck1, acc, 0, 0, 0);
                if (nk2 != 0) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(qf[kk], ck2, acc, 0, 0, 0);
            }
        }
        const int j = jt * 32 + frow;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = q0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (i < p.T && j < p.S) obase[(long)i * p.ldo + j] = (float)acc[r] * cs;
        }
    }
}

struct PvK {
    const float* w;         // [BH][T][ldw] probabilities (fp32)
    const int8_t* vt;       // [BH][dpad][Spad] key-permuted stored bytes of v
    const int32_t* vsum;    // [BH][dpad]
    const float* prm;       // 3 dw | 4 zpw | 5 dw*dv | 6 zv'
    float* out;             // [BH][d][ldo]  ("bct")
    long ldw, wbstride, ldo, obstride;
    int T, S, d, Spad, dpad;
    float wmin, wmax;
    int iwmin;
};

// one wave = 32 queries; P codes are built from the fp32 probabilities with the reference's formula (true
// division, round-half-even) directly in the A-operand layout of the P.V MFMA (key slot p = half*16 + r <->
// key (r&3) + 8*(r>>2) + 4*half of the tile), 16-bit codes as hi/lo bytes; epilogue as in attn_i8.hip.
template <int DT, bool P16>
__global__ __launch_bounds__(256) void bmm_pv_kernel(const PvK p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (q0 >= p.T) return;
    const float dw = p.prm[3], zpw = p.prm[4], oscale = p.prm[5];
    const int zv = (int)p.prm[6], izpw = (int)zpw;

    v16i ol[DT], oh[P16 ? DT : 1];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ol[t][r] = 0;
            if (P16) oh[P16 ? t : 0][r] = 0;
        }
    int uusum = 0, nvalid = 0;
    const int qi = q0 + frow;
    const float* wrow = p.w + (long)bh * p.wbstride + (long)(qi < p.T ? qi : 0) * p.ldw;
    const int8_t* vbase = p.vt + ((long)bh * p.dpad + frow) * p.Spad + half * 16;
    const int ntile = p.Spad >> 5;
    for (int jt = 0; jt < ntile; ++jt) {
        int uu[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const bool ok = j < p.S;
            const float x = ok ? wrow[j] : 0.f;
            const int code = qd_code(x, dw, zpw, p.wmin, p.wmax) - p.iwmin;     // uu = u - wmin in [0, 65535]
            uu[r] = ok ? code : 0x8080;                                         // bytes the ^0x80 below turns into 0
            uusum += ok ? code : 0;
            nvalid += ok ? 1 : 0;
        }
        v4i plo, phi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const unsigned t01 = (unsigned)uu[4 * g] | ((unsigned)uu[4 * g + 1] << 16);
            const unsigned t23 = (unsigned)uu[4 * g + 2] | ((unsigned)uu[4 * g + 3] << 16);
            plo[g] = (int)(__builtin_amdgcn_perm(t23, t01, 0x06040200u) ^ 0x80808080u);
            phi[g] = P16 ? (int)(__builtin_amdgcn_perm(t23, t01, 0x07050301u) ^ 0x80808080u) : 0;
        }
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const v4i vf = *reinterpret_cast<const v4i*>(vbase + (long)t * 32 * p.Spad + jt * 32);
            ol[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(plo, vf, ol[t], 0, 0, 0);
            if (P16) oh[P16 ? t : 0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(phi, vf, oh[P16 ? t : 0], 0, 0, 0);
        }
    }
    uusum += __shfl_xor(uusum, 32);
    nvalid += __shfl_xor(nvalid, 32);
    const int usum = uusum + nvalid * p.iwmin;                     // sum over valid keys of the codes u
    const long kconst = (P16 ? 256L * 128L : 0L) + 128L + (long)p.iwmin - (long)izpw;  // multiplies vsum
    float* obase = p.out + (long)bh * p.obstride;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        const int dd = t * 32 + frow;
        const long vs = (dd < p.d) ? p.vsum[(long)bh * p.dpad + dd] : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int il = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int us = __shfl(usum, il);
            const int i = q0 + il;
            if (dd >= p.d || i >= p.T) continue;
            long I = (long)ol[t][r] + kconst * vs - (long)zv * us + (long)p.S * izpw * zv;
            if (P16) I += 256L * (long)oh[P16 ? t : 0][r];
            obase[(long)dd * p.ldo + i] = (float)I * oscale;
        }
    }
}

template <int DT>
void launch_pv(const PvK& k, bool p16, dim3 grid, hipStream_t st) {
    if (p16) hipLaunchKernelGGL((bmm_pv_kernel<DT, true>), grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL((bmm_pv_kernel<DT, false>), grid, dim3(256), 0, st, k);
}

}  // namespace

extern "C" int qd_bmm_qk_i8(const int8_t* q, const int8_t* k, int BH, int T, int S, int d, int Tpad, int Spad, int dpad,
                            const float* prm, float* out, int64_t ldo, int64_t bstride, void* stream) {
    QD_REQUIRE(q && k && prm && out, "qd_bmm_qk_i8: null pointer");
    QD_REQUIRE(BH > 0 && BH < 65536 && T > 0 && S > 0 && d > 0, "qd_bmm_qk_i8: bad shape");
    QD_REQUIRE(Tpad % 32 == 0 && Spad % 32 == 0 && dpad % 32 == 0 && Tpad >= T && Spad >= S && dpad >= d, "qd_bmm_qk_i8: padded dims must be multiples of 32");
    QD_REQUIRE(qd_aligned(q, 16) && qd_aligned(k, 16) && ldo >= S, "qd_bmm_qk_i8: operands must be 16-byte aligned, ldo >= S");
    QkK a{q, k, prm, out, (long)ldo, (long)bstride, T, S, d, Tpad, Spad, dpad};
    dim3 grid((unsigned)((T + 127) / 128), (unsigned)BH);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (dpad / 32) {
        case 1: hipLaunchKernelGGL(bmm_qk_kernel<1>, grid, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(bmm_qk_kernel<2>, grid, dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL(bmm_qk_kernel<3>, grid, dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL(bmm_qk_kernel<4>, grid, dim3(256), 0, st, a); break;
        case 5: hipLaunchKernelGGL(bmm_qk_kernel<5>, grid, dim3(256), 0, st, a); break;
        case 8: hipLaunchKernelGGL(bmm_qk_kernel<8>, grid, dim3(256), 0, st, a); break;
        default: qd_set_error("qd_bmm_qk_i8: head dim pad %d unsupported (32,64,96,128,160,256)", dpad); return 1;
    }
    QD_LAUNCH_CHECK("qd_bmm_qk_i8");
    return 0;
}

extern "C" int qd_bmm_pv_i8(const float* w, int64_t ldw, int64_t wbstride, const int8_t* vt, const int32_t* vsum, int BH, int T,
                            int S, int d, int Spad, int dpad, const float* prm, int wbits, int wmin, int wmax, float* out,
                            int64_t ldo, int64_t obstride, void* stream) {
    QD_REQUIRE(w && vt && vsum && prm && out, "qd_bmm_pv_i8: null pointer");
    QD_REQUIRE(BH > 0 && BH < 65536 && T > 0 && S > 0 && d > 0, "qd_bmm_pv_i8: bad shape");
    QD_REQUIRE(Spad % 32 == 0 && dpad % 32 == 0 && Spad >= S && dpad >= d && qd_aligned(vt, 16), "qd_bmm_pv_i8: padded dims must be multiples of 32, vt 16-byte aligned");
    QD_REQUIRE(wbits == 8 || wbits == 16, "qd_bmm_pv_i8: probability bits must be 8 or 16 (got %d)", wbits);
    QD_REQUIRE(wmax - wmin <= (wbits == 16 ? 65535 : 255), "qd_bmm_pv_i8: probability grid [%d,%d] wider than %d bits", wmin, wmax, wbits);
    QD_REQUIRE(ldw >= S && ldo >= T, "qd_bmm_pv_i8: ldw >= S and ldo >= T required");
    PvK a{w, vt, vsum, prm, out, (long)ldw, (long)wbstride, (long)ldo, (long)obstride, T, S, d, Spad, dpad, (float)wmin, (float)wmax, wmin};
    dim3 grid((unsigned)((T + 127) / 128), (unsigned)BH);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool p16 = wbits == 16;
    switch (dpad / 32) {
        case 1: launch_pv<1>(a, p16, grid, st); break;
        case 2: launch_pv<2>(a, p16, grid, st); break;
        case 3: launch_pv<3>(a, p16, grid, st); break;
        case 4: launch_pv<4>(a, p16, grid, st); break;
        case 5: launch_pv<5>(a, p16, grid, st); break;
        case 8: launch_pv<8>(a, p16, grid, st); break;
        default: qd_set_error("qd_bmm_pv_i8: head dim pad %d unsupported (32,64,96,128,160,256)", dpad); return 1;
    }
    QD_LAUNCH_CHECK("qd_bmm_pv_i8");
    return 0;
}
