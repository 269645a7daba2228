// attn_i8.hip — K7/K8: quantised attention, fused QK^T -> softmax -> quantise(P) -> P.V on MFMA-i8.
//
// Replaces (reference qdiff/quant_block.py:190-221, :123-157, :354-386): fake-quant of q,k ->
// einsum/bmm -> *scale -> fp32 softmax -> fake-quant of P (8- or 16-bit, sm_abit) and v -> einsum.
// The T x S score matrix (1.07 GB fp32 for SD at batch 2) is never materialised.
//
// One wave owns 32 queries.  Scores are computed TRANSPOSED (A = key rows, B = query rows), so the
// 32x32x32 MFMA C layout leaves every lane with ONE query (lane&31) and 16 of the 32 keys of the
// tile: softmax row reductions are lane-local plus one cross-half shuffle, and the quantised
// probabilities are already in A-operand layout for the P.V MFMA (row = query, 16 K-bytes per lane).
// The key order inside a 32-key tile is whatever the C layout gives; V^T is stored pre-permuted by
// qd_quantize_heads so that both operands agree (a contraction is invariant to a K permutation).
// Softmax needs the final row max / sum before P can be quantised with its static delta, hence two
// sweeps over the keys (sweep 1: online max/sum; sweep 2: recompute S, quantise P, accumulate P.V).
// 16-bit probabilities are split into hi/lo bytes: two exact int32 accumulators, combined in int64.
#include "common.h"
#include <climits>

namespace {

struct AttnK {
    const int8_t* q;
    const int8_t* k;
    const int8_t* vt;
    const int32_t* qsum;
    const int32_t* ksum;
    const int32_t* vsum;
    const float* prm;
    float* out;
    long ldo;
    int BH, H, T, S, d, Tpad, Spad, dpad;
    float wmin, wmax;
    int iwmin;
};

// prm layout (device floats): 0 cs = dq*dk*scale | 1 zq' | 2 zk' | 3 dw | 4 zpw | 5 dw*dv | 6 zv'
//
// VALU budget (the kernel is VALU-bound: ~2x more vector ops than MFMA cycles): the softmax runs in
// the exp2 domain with every constant folded (cs*log2e; inv_l/dw), the running max is taken on the
// integer scores (monotone in the float score), probabilities are packed to MFMA operand bytes with
// two v_perm_b32 per four values ((x-128)&0xff == x^0x80), and key masking exists only in the last
// (ragged) tile.
template <int DT, bool P16, bool ASYM>
__global__ __launch_bounds__(256, (DT * (P16 ? 2 : 1) <= 4) ? 2 : 1) void attn_kernel(const AttnK p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (q0 >= p.T) return;

    const float cs2 = p.prm[0] * 1.4426950408889634f;            // scores -> log2 domain
    const int zq = (int)p.prm[1], zk = (int)p.prm[2];
    const float dw = p.prm[3], zpw = p.prm[4], oscale = p.prm[5];
    const int zv = (int)p.prm[6];
    const int izpw = (int)zpw;
    const float urange = p.wmax - p.wmin;                         // codes are handled as uu = u - wmin in [0, urange]
    const float ubias = zpw - p.wmin;

    v4i qf[DT];
    const int8_t* qrow = p.q + ((long)bh * p.Tpad + q0 + frow) * p.dpad + half * 16;
#pragma unroll
    for (int kk = 0; kk < DT; ++kk) qf[kk] = *reinterpret_cast<const v4i*>(qrow + kk * 32);
    int qs_term = 0;                                              // -zk*qsum[i] + d*zq*zk (per query)
    if (ASYM) qs_term = -zk * p.qsum[(long)bh * p.Tpad + q0 + frow] + p.d * zq * zk;

    const int8_t* kbase = p.k + (long)bh * p.Spad * p.dpad + (long)frow * p.dpad + half * 16;
    const int32_t* ksum = p.ksum + (long)bh * p.Spad + 4 * half;
    const int ntile = p.Spad >> 5;
    const int tail_tile = (p.S & 31) ? ntile - 1 : ntile;         // index of the ragged tile (or none)

    // integer scores of key tile jt (zero points restored), C layout: si[4g+e] <-> key jt*32 + e + 8g + 4*half
    auto int_scores = [&](int jt, int (&si)[16]) __attribute__((always_inline)) {
        v16i acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0;
        const int8_t* kp = kbase + (long)jt * 32 * p.dpad;
#pragma unroll
        for (int kk = 0; kk < DT; ++kk) {
            v4i kf = *reinterpret_cast<const v4i*>(kp + kk * 32);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf, qf[kk], acc, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            v4i ks = {0, 0, 0, 0};
            if (ASYM) ks = *reinterpret_cast<const v4i*>(ksum + jt * 32 + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) si[g * 4 + e] = ASYM ? acc[g * 4 + e] + qs_term - zq * ks[e] : acc[g * 4 + e];
        }
        if (jt == tail_tile) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (j >= p.S) si[r] = INT_MIN;                    // masked: exp2(-inf) = 0
            }
        }
    };
    auto to_log2 = [&](int v) __attribute__((always_inline)) { return v == INT_MIN ? -INFINITY : (float)v * cs2; };

    // ---- sweep 1: row max and normaliser (online, log2 domain) ------------------------------------
    float m = -INFINITY, l = 0.f;
    for (int jt = 0; jt < ntile; ++jt) {
        int si[16];
        int_scores(jt, si);
        int tmax = si[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = max(tmax, si[r]);
        const float mn = fmaxf(m, to_log2(tmax));
        if (mn > -INFINITY) {
            float a = 0.f;
            if (jt == tail_tile) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a += __builtin_amdgcn_exp2f(to_log2(si[r]) - mn);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) a += __builtin_amdgcn_exp2f(__builtin_fmaf((float)si[r], cs2, -mn));
            }
            l = l * __builtin_amdgcn_exp2f(m - mn) + a;
            m = mn;
        }
    }
    {
        const float mo = __shfl_xor(m, 32), lo = __shfl_xor(l, 32);
        const float mf = fmaxf(m, mo);
        l = (m > -INFINITY ? l * __builtin_amdgcn_exp2f(m - mf) : 0.f) + (mo > -INFINITY ? lo * __builtin_amdgcn_exp2f(mo - mf) : 0.f);
        m = mf;
    }
    const float inv = 1.0f / (l * dw);                            // p/dw = e * inv

    // ---- sweep 2: quantise P, accumulate P.V ------------------------------------------------------
    v16i ol[DT], oh[P16 ? DT : 1];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ol[t][r] = 0;
            if (P16) oh[P16 ? t : 0][r] = 0;
        }
    int uusum = 0, nvalid = 0;
    const int8_t* vbase = p.vt + ((long)bh * p.dpad + frow) * p.Spad + half * 16;
    for (int jt = 0; jt < ntile; ++jt) {
        int si[16];
        int_scores(jt, si);
        int uu[16];
        const bool tail = jt == tail_tile;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = tail ? __builtin_amdgcn_exp2f(to_log2(si[r]) - m) : __builtin_amdgcn_exp2f(__builtin_fmaf((float)si[r], cs2, -m));
            float t = __builtin_rintf(__builtin_fmaf(e, inv, ubias));
            t = fminf(fmaxf(t, 0.f), urange);
            uu[r] = (int)t;
        }
        if (tail) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = si[r] != INT_MIN;
                uusum += ok ? uu[r] : 0;
                nvalid += ok ? 1 : 0;
                if (!ok) uu[r] = 0x8080;                          // bytes that the ^0x80 below turns into 0
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) uusum += uu[r];
            nvalid += 16;
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
            v4i vf = *reinterpret_cast<const v4i*>(vbase + (long)t * 32 * p.Spad + jt * 32);
            ol[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(plo, vf, ol[t], 0, 0, 0);
            if (P16) oh[P16 ? t : 0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(phi, vf, oh[P16 ? t : 0], 0, 0, 0);
        }
    }
    uusum += __shfl_xor(uusum, 32);
    nvalid += __shfl_xor(nvalid, 32);
    const int usum = uusum + nvalid * p.iwmin;                     // sum over valid keys of the codes u = uu + wmin

    // ---- epilogue: restore zero points (exact, int64), scale, store merged-head rows ------------
    const int b = bh / p.H, hh = bh % p.H;
    const long kconst = (P16 ? 256L * 128L : 0L) + 128L + (long)p.iwmin - (long)izpw;  // multiplies vsum
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
            p.out[((long)b * p.T + i) * p.ldo + hh * p.d + dd] = (float)I * oscale;
        }
    }
}

template <int DT>
int launch_dt(const AttnK& k, bool p16, bool asym, hipStream_t st) {
    dim3 grid((unsigned)((k.T + 127) / 128), (unsigned)k.BH);
    if (p16 && asym) hipLaunchKernelGGL((attn_kernel<DT, true, true>), grid, dim3(256), 0, st, k);
    else if (p16) hipLaunchKernelGGL((attn_kernel<DT, true, false>), grid, dim3(256), 0, st, k);
    else if (asym) hipLaunchKernelGGL((attn_kernel<DT, false, true>), grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL((attn_kernel<DT, false, false>), grid, dim3(256), 0, st, k);
    return 0;
}

}  // namespace

extern "C" int qd_attn_i8(const int8_t* q, const int8_t* k, const int8_t* vt, const int32_t* qsum, const int32_t* ksum,
                          const int32_t* vsum, int BH, int H, int T, int S, int d, int Tpad, int Spad, int dpad,
                          const float* prm, int wbits, int wmin, int wmax, float* out, int64_t ldo, void* stream) {
    QD_REQUIRE(q && k && vt && vsum && prm && out, "qd_attn_i8: null pointer");
    QD_REQUIRE(BH > 0 && H > 0 && BH % H == 0 && T > 0 && S > 0 && d > 0, "qd_attn_i8: bad shape");
    QD_REQUIRE(Tpad % 32 == 0 && Spad % 32 == 0 && dpad % 32 == 0 && Tpad >= T && Spad >= S && dpad >= d, "qd_attn_i8: padded dims must be multiples of 32");
    QD_REQUIRE(BH < 65536, "qd_attn_i8: too many heads for grid.y");
    QD_REQUIRE(wbits == 8 || wbits == 16, "qd_attn_i8: probability bits must be 8 or 16 (got %d)", wbits);
    QD_REQUIRE(wmax - wmin <= (wbits == 16 ? 65535 : 255), "qd_attn_i8: probability grid [%d,%d] wider than %d bits", wmin, wmax, wbits);
    QD_REQUIRE(qd_aligned(q, 16) && qd_aligned(k, 16) && qd_aligned(vt, 16), "qd_attn_i8: operands must be 16-byte aligned");
    const bool asym = qsum != nullptr && ksum != nullptr;
    QD_REQUIRE(asym || (qsum == nullptr && ksum == nullptr), "qd_attn_i8: pass both qsum and ksum or neither");
    AttnK a{q, k, vt, qsum, ksum, vsum, prm, out, (long)ldo, BH, H, T, S, d, Tpad, Spad, dpad, (float)wmin, (float)wmax, wmin};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool p16 = wbits == 16;
    switch (dpad / 32) {
        case 1: launch_dt<1>(a, p16, asym, st); break;
        case 2: launch_dt<2>(a, p16, asym, st); break;
        case 3: launch_dt<3>(a, p16, asym, st); break;
        case 4: launch_dt<4>(a, p16, asym, st); break;
        case 5: launch_dt<5>(a, p16, asym, st); break;
        case 8: launch_dt<8>(a, p16, asym, st); break;
        default:
            qd_set_error("qd_attn_i8: head dim pad %d unsupported (32,64,96,128,160,256)", dpad);
            return 1;
    }
    QD_LAUNCH_CHECK("qd_attn_i8");
    return 0;
}
