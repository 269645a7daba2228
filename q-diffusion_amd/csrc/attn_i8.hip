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
#include <cstdlib>
#include <type_traits>

namespace {


struct AttnK {
    const int8_t* q;
    const int8_t* k;
    const int8_t* vt;
    const int32_t* qsum;
    const int32_t* kterm;     // lean / LDS-staged kernels, KT == 2: [BH][Spad] accumulator seeds 0x4B400000 - zq' * sum_d k'[j][d]
                              // (qd_attn_keyterm); NULL: the per-key term comes from constant-operand MFMAs (KT == 1)
    const int32_t* vsum;
    const float* prm;
    float* out;
    long ldo;
    int BH, H, T, S, d, Tpad, Spad, dpad;
    float wmin, wmax;
    int iwmin;
    // optional quantised output (the act quantiser of the Linear that consumes the attention output)
    int8_t* out8;
    long ldo8;
    const float* oq;
    float oqmin, oqmax;
    int oqoff;
    int xcd;                  // 1: XCD-aware block order (default); 0: plain (QD_ATTN_XCD=0, A/B measurements)
    int gx;                   // query blocks (of 128) per head: the grid is 1-D, gx * BH blocks, remapped so that the blocks of one
                              // head share an XCD (block b runs on XCD b % 8: with a (gx, BH) grid every XCD's L2 fetched every head's
                              // K / V — 21 % L2 misses, profiles/r03_pmc_attn.txt)
};

// Epilogue of all three kernels: I = sum_j (u_j - zpw)(v'_j - zv'), restored exactly from the operand-byte accumulators
// (ol / oh: lo / hi bytes of the codes against v'), the V^T column sums and the code sums us[r] of the lane's 16 queries;
// o = float(I) * dw*dv, stored as fp32 rows or as the int8 input rows of the consuming Linear.
//   * |I| <= 255 * sum_j |u_j - zpw| and sum_j round(p_j / dw) <= 1/dw + S/2: with the softmax quantiser's zero point at 0
//     (always_zero, quant_block.py:240-252) the TRUE value is below 2^25 however large the individual terms are, so the
//     restoration runs in wrapping 32-bit arithmetic (exact whenever the result fits, which the bound guarantees) and converts
//     with v_cvt_f32_i32; the 64-bit form (emulated i64 -> f32 conversion: a third of the epilogue's instructions, and the
//     epilogue is half of a 77-key cross-attention call) stays for grids where the bound does not hold.  Same value either
//     way: one correctly rounded conversion of the same integer.
template <int DT, bool P16>
__device__ __forceinline__ void attn_write_rows(const AttnK& p, const v16i (&ol)[DT], const v16i (&oh)[P16 ? DT : 1], const int (&us)[16],
                                                bool hi_live, int bh, int q0, int frow, int half, float dw, float zpw, float oscale, int zv) {
    const int b = bh / p.H, hh = bh % p.H;
    const int izpw = (int)zpw;
    const QP oqp = p.out8 ? qd_load_qp(p.oq) : QP{1.f, 0.f, 1.f, false};
    const long kconst = (hi_live ? 256L * 128L : 0L) + 128L + (long)p.iwmin - (long)izpw;     // multiplies vsum
    const float code_sum = (1.0f / dw) * 1.01f + 0.5f * (float)p.S + (float)p.S * fmaxf(0.f, p.wmin - zpw) + 16.f;
    const bool small = code_sum * 255.f < 2.0e9f;                 // grid-uniform
    auto epi = [&](auto ft, auto st) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(ft)::value, SMALL = decltype(st)::value;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        const int dd = t * 32 + frow;
        const long vs = (dd < p.d) ? p.vsum[(long)bh * p.dpad + dd] : 0;
        const long ct = kconst * vs + (long)p.S * izpw * zv;      // the column's share
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int il = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int i = q0 + il;
            if (dd >= p.d || i >= p.T) continue;
            float o;
            if (SMALL) {
                unsigned I = (unsigned)ol[t][r] + (unsigned)ct - (unsigned)zv * (unsigned)us[r];
                if (P16 && hi_live) I += (unsigned)oh[P16 ? t : 0][r] << 8;
                o = (float)(int)I * oscale;
            } else {
                long I = (long)ol[t][r] + ct - (long)zv * us[r];
                if (P16 && hi_live) I += 256L * (long)oh[P16 ? t : 0][r];
                o = (float)I * oscale;
            }
            if (p.out8) p.out8[((long)b * p.T + i) * p.ldo8 + hh * p.d + dd] = (int8_t)(qd_code_t<FAST>(o, oqp, p.oqmin, p.oqmax) - p.oqoff);
            else p.out[((long)b * p.T + i) * p.ldo + hh * p.d + dd] = o;
        }
    }
    };
    if (small) QD_FAST_DISPATCH(oqp.fast, [&](auto ft) __attribute__((always_inline)) { epi(ft, std::true_type{}); });
    else QD_FAST_DISPATCH(oqp.fast, [&](auto ft) __attribute__((always_inline)) { epi(ft, std::false_type{}); });
}

// prm layout (device floats): 0 cs = dq*dk*scale | 1 zq' | 2 zk' | 3 dw | 4 zpw | 5 dw*dv | 6 zv'
//
// VALU budget.  The kernel is VALU-bound (every score costs an int->float convert, an exp2 at quarter
// rate and, in the second sweep, the quantisation of P), so the score path is kept to the minimum:
//   * per-QUERY terms of the zero-point restoration (-zk*qsum_i + d*zq*zk) are dropped: a constant added
//     to a softmax row cancels exactly.  The per-KEY term -zq*sum_d k'[j][d] is computed by the matrix
//     pipe itself: one more MFMA per K fragment against a constant operand whose bytes are all -zq
//     (the MFMA pipe has slack, the VALU does not), so no key row sums are needed at all;
//   * the row maximum is subtracted in the INTEGER domain for free: the MFMA accumulator is initialised
//     with -max instead of 0, so exp2 sees cs*log2e*(s - max) with an exact difference;
//   * rounding to the probability grid is one float add of 1.5*2^23 (round-half-even lands in the low
//     mantissa bits, which v_perm_b32 then scatters into the hi/lo operand bytes);
//   * code sums come from v_dot4 on the packed operand bytes, not from per-score adds;
//   * key masking exists only in the last (ragged) tile;
// and the next K tile / this tile's V tile are prefetched into registers ahead of the softmax math, so
// the L2 latency of the (tiny, shared) K/V stream hides behind ~1k VALU cycles.
template <int DT, bool P16, bool ASYM>
// (3 blocks per CU was tried for DT=2/P16: 168 VGPRs + 100 B of scratch, 16% slower end to end.)
__global__ __launch_bounds__(256, (DT * (P16 ? 2 : 1) <= 6) ? 2 : 1) void attn_kernel(const AttnK p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 31, half = lane >> 5;
    const int lblk = p.xcd ? qd_xcd_remap(blockIdx.x, p.gx * p.BH) : (int)blockIdx.x;
    const int bh = lblk / p.gx;
    const int q0 = ((lblk - bh * p.gx) * 4 + wave) * 32;
    if (q0 >= p.T) return;

    const float cs2 = p.prm[0] * 1.4426950408889634f;            // scores -> log2 domain
    const int nzq = -(int)p.prm[1];
    const float dw = p.prm[3], zpw = p.prm[4], oscale = p.prm[5];
    const int zv = (int)p.prm[6];
    const int izpw = (int)zpw;
    const float urange = p.wmax - p.wmin;                         // codes are handled as uu = u - wmin in [0, urange]
    const float ubias = zpw - p.wmin;
    constexpr float MAGIC = 12582912.f;                           // 1.5 * 2^23: float add == round-half-even to integer
    constexpr int   MASKED = -(1 << 30);

    v4i qf[DT];
    const int8_t* qrow = p.q + ((long)bh * p.Tpad + q0 + frow) * p.dpad + half * 16;
#pragma unroll
    for (int kk = 0; kk < DT; ++kk) qf[kk] = *reinterpret_cast<const v4i*>(qrow + kk * 32);

    const int8_t* kbase = p.k + (long)bh * p.Spad * p.dpad + (long)frow * p.dpad + half * 16;
    // -zq' in [-127, 128] as one or two int8 constants (128 does not fit a signed byte)
    const int c1 = nzq > 127 ? 64 : nzq, c2 = nzq - c1;
    const int c1w = (c1 & 0xff) * 0x01010101, c2w = (c2 & 0xff) * 0x01010101;
    const v4i c1v = {c1w, c1w, c1w, c1w}, c2v = {c2w, c2w, c2w, c2w};
    const int ntile = p.Spad >> 5;
    const int tail_tile = (p.S & 31) ? ntile - 1 : ntile;         // index of the ragged tile (or none)

    // K-tile registers (double-buffered by hand: `kf/ks` = current tile, loaded one iteration ahead)
    auto load_k = [&](int jt, v4i (&kf)[DT]) __attribute__((always_inline)) {
        const int8_t* kp = kbase + (long)jt * 32 * p.dpad;
#pragma unroll
        for (int kk = 0; kk < DT; ++kk) kf[kk] = *reinterpret_cast<const v4i*>(kp + kk * 32);
    };
    // d[4g+e] = (score of key jt*32 + e + 8g + 4*half) - base, per-query constants dropped
    auto scores = [&](const v4i (&kf)[DT], int base, int (&d)[16]) __attribute__((always_inline)) {
        v16i acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = -base;
#pragma unroll
        for (int kk = 0; kk < DT; ++kk) {
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[kk], qf[kk], acc, 0, 0, 0);
            if (ASYM) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[kk], c1v, acc, 0, 0, 0);
        }
        if (ASYM && c2 != 0) {                                    // wave-uniform, only when zq' == -128
#pragma unroll
            for (int kk = 0; kk < DT; ++kk) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[kk], c2v, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = acc[r];
    };
    auto key_ok = [&](int jt, int r) __attribute__((always_inline)) { return jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < p.S; };

    // ---- sweep 1: row max (integer) and normaliser (online) ---------------------------------------
    int mi;                                                       // running integer row max (shifted scores)
    float l = 0.f;
    {
        v4i kf[DT], kfn[DT];
        load_k(0, kf);
        {   // seed the running max with tile 0's (so every later difference s - mi is small and exact in fp32)
            int d[16];
            scores(kf, 0, d);
            if (tail_tile == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) if (!key_ok(0, r)) d[r] = MASKED;
            }
            mi = d[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mi = max(mi, d[r]);
        }
        // The ragged last tile is peeled out of the loop (TAIL is a compile-time flag): with a run-time `tail` the
        // compiler kept 56 masking selects per tile in the hot loop (a third of sweep 1's VALU work).
        auto s1_tile = [&](int jt, auto tail_tag) __attribute__((always_inline)) {
            constexpr bool tail = decltype(tail_tag)::value;
            if (jt + 1 < ntile) load_k(jt + 1, kfn);
            int d[16];
            scores(kf, mi, d);                                // d = s - mi
            if (tail) {
#pragma unroll
                for (int r = 0; r < 16; ++r) if (!key_ok(jt, r)) d[r] = MASKED;
            }
            int tmax = d[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = max(tmax, d[r]);
            const int up = max(tmax, 0);                          // the row max moves up by `up`
            const float shift = (float)up * cs2;
            // packed fp32 (v_pk_fma_f32 / v_pk_add_f32: two scores per instruction) around the scalar cvt / exp2
            v2f a2 = {0.f, 0.f};
            const v2f cs2v = {cs2, cs2}, nshift = {-shift, -shift};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const v2f df = {(float)d[r], (float)d[r + 1]};
                const v2f x = __builtin_elementwise_fma(df, cs2v, nshift);
                v2f e = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                if (tail) {
                    if (d[r] == MASKED) e.x = 0.f;
                    if (d[r + 1] == MASKED) e.y = 0.f;
                }
                a2 += e;
            }
            l = l * __builtin_amdgcn_exp2f(-shift) + (a2.x + a2.y);   // first tile: l == 0, the factor is irrelevant
            mi += up;
            if (jt + 1 < ntile) {
#pragma unroll
                for (int kk = 0; kk < DT; ++kk) kf[kk] = kfn[kk];
            }
        };
        for (int jt = 0; jt < tail_tile; ++jt) s1_tile(jt, std::false_type{});
        if (tail_tile < ntile) s1_tile(tail_tile, std::true_type{});
    }
    {
        const int mo = __shfl_xor(mi, 32);
        const float lo = __shfl_xor(l, 32);
        const int mf = max(mi, mo);
        l = l * __builtin_amdgcn_exp2f((float)(mi - mf) * cs2) + lo * __builtin_amdgcn_exp2f((float)(mo - mf) * cs2);
        mi = mf;
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
    int dlo = 0, dhi = 0, nvalid = 0;                             // signed operand-byte sums, valid-key count
    const int8_t* vbase = p.vt + ((long)bh * p.dpad + frow) * p.Spad + half * 16;
    {
        v4i kf[DT], kfn[DT];
        load_k(0, kf);
        auto s2_tile = [&](int jt, auto tail_tag, auto clamp_tag) __attribute__((always_inline)) {
            constexpr bool tail = decltype(tail_tag)::value, CLAMP = decltype(clamp_tag)::value;
            v4i vf[DT];
#pragma unroll
            for (int t = 0; t < DT; ++t) vf[t] = *reinterpret_cast<const v4i*>(vbase + (long)t * 32 * p.Spad + jt * 32);
            if (jt + 1 < ntile) load_k(jt + 1, kfn);
            int d[16];
            scores(kf, mi, d);                                // d = s - rowmax <= 0
            unsigned ub[16];                                      // float bits of uu + MAGIC: low 16 bits == uu
            const v2f cs2v = {cs2, cs2}, invv = {inv, inv}, ubv = {ubias, ubias}, magic = {MAGIC, MAGIC};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const v2f df = {(float)d[r], (float)d[r + 1]};
                const v2f x = df * cs2v;
                const v2f e = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                v2f t = __builtin_elementwise_fma(e, invv, ubv);                        // e*inv + ubias >= 0 always
                if (CLAMP) {
                    t.x = fminf(t.x, urange);
                    t.y = fminf(t.y, urange);
                }
                t += magic;
                ub[r] = __float_as_uint(t.x);
                ub[r + 1] = __float_as_uint(t.y);
            }
            if (tail) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = key_ok(jt, r);
                    nvalid += ok ? 1 : 0;
                    if (!ok) ub[r] = 0x8080u;                     // bytes that the ^0x80 below turns into 0
                }
            } else {
                nvalid += 16;
            }
            v4i plo, phi;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // bytes 0/1 of each word are the lo/hi byte of the code
                const unsigned a01 = __builtin_amdgcn_perm(ub[4 * g + 1], ub[4 * g], 0x05010400u);      // lo0 lo1 hi0 hi1
                const unsigned a23 = __builtin_amdgcn_perm(ub[4 * g + 3], ub[4 * g + 2], 0x05010400u);  // lo2 lo3 hi2 hi3
                plo[g] = (int)(__builtin_amdgcn_perm(a23, a01, 0x05040100u) ^ 0x80808080u);
                dlo = __builtin_amdgcn_sdot4(plo[g], 0x01010101, dlo, false);
                if (P16) {
                    phi[g] = (int)(__builtin_amdgcn_perm(a23, a01, 0x07060302u) ^ 0x80808080u);
                    dhi = __builtin_amdgcn_sdot4(phi[g], 0x01010101, dhi, false);
                } else {
                    phi[g] = 0;
                }
            }
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                ol[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(plo, vf[t], ol[t], 0, 0, 0);
                if (P16) oh[P16 ? t : 0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(phi, vf[t], oh[P16 ? t : 0], 0, 0, 0);
            }
            if (jt + 1 < ntile) {
#pragma unroll
                for (int kk = 0; kk < DT; ++kk) kf[kk] = kfn[kk];
            }
        };
        // a wave whose largest possible code (e = 1 -> inv + ubias) stays on the probability grid needs no upper clamp:
        // with thousands of keys the normaliser is >> 1 and this is the common case (16 of ~100 VALU instructions per tile)
        if (__any(!(inv + ubias <= urange))) {
            for (int jt = 0; jt < tail_tile; ++jt) s2_tile(jt, std::false_type{}, std::true_type{});
        } else {
            for (int jt = 0; jt < tail_tile; ++jt) s2_tile(jt, std::false_type{}, std::false_type{});
        }
        if (tail_tile < ntile) s2_tile(tail_tile, std::true_type{}, std::true_type{});
    }
    // sum over valid keys of uu = 256*hi + lo, from the signed operand bytes (masked keys hold 0)
    int uusum = dlo + 128 * nvalid + (P16 ? 256 * (dhi + 128 * nvalid) : 0);
    uusum += __shfl_xor(uusum, 32);
    nvalid += __shfl_xor(nvalid, 32);
    const int usum = uusum + nvalid * p.iwmin;                     // sum over valid keys of the codes u = uu + wmin

    // ---- epilogue: restore zero points (exact integers), scale, store merged-head rows ------------
    int us[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) us[r] = __shfl(usum, (r & 3) + 8 * (r >> 2) + 4 * half);
    attn_write_rows<DT, P16>(p, ol, oh, us, P16, bh, q0, frow, half, dw, zpw, oscale, zv);
}

// ---- lean variant (head dims that are NOT a multiple of 32 and < 64: Stable Diffusion's 4096-token, d = 40 level) -----
// Same mathematics, ~25 % fewer VALU instructions per score and (round 4) half the score MFMAs.  The 32x32 score tile costs
// VALU AND matrix time (the two pipes of a SIMD do not overlap in this instruction mix, profiles/r03_attn_ablation.md), so:
//   * no int->float converts and no per-tile accumulator initialisation: the MFMA accumulator starts at the bit pattern of
//     1.5*2^23 plus an INTEGER, so its bits ARE the float 1.5*2^23 + s, exactly, as long as |s| < 2^22 — the launcher
//     guarantees that from d and the operand ranges (d * 255 * 128 < 2^22 <=> d < 64).  The row maximum m is NOT subtracted
//     in the accumulator (round 3 did: a per-lane seed, which ties the seed to the lane) but in the constant of the one
//     packed FMA that maps the score to the log2 domain: x = fma(F, cs2, nc(m)), nc(m) = -fl((1.5*2^23 + m) * cs2).  The FMA
//     is exact inside, so x = cs2*(s - m) + eps(m) with ONE rounding; eps(m), the rounding error of nc(m), is an exactly
//     representable float (computed as fma(1.5*2^23 + m, cs2, nc(m))) common to every score of the row: the normaliser is
//     corrected by 2^(eps(m_sweep2) - eps(m_sweep1)), see attn_rowref / attn_finish_stats;
//   * the per-KEY zero-point term -zq' * sum_d k'[j][d] is therefore free (KT == 2): the seed of accumulator register r is
//     0x4B400000 - zq'*ksum[key(r)], the same for every lane of a half-wave, read as four 16-byte words per tile from a
//     table that qd_attn_keyterm builds once per K operand — no constant-operand MFMAs (round 3: 2 of the 4 score MFMAs of
//     a tile carried no data), no VALU.  KT == 1 keeps the constant-operand MFMAs for callers without a table (bit-identical
//     results: the accumulators hold the same integers); KT == 0: symmetric q, no term;
//   * sweep 1 does not maintain a running maximum in the float domain (16 accumulator re-initialisations, a rescale and
//     an extra exp2 per tile): it accumulates sum(exp2(cs2*(s - m0))) against the maximum m0 of the FIRST tile (taken over
//     both half-waves, so both halves of a query row share one reference and one eps) and tracks the integer maximum with
//     v_max3_i32 only.  fp32 has the exponent range for that (the relative precision of a floating-point sum does not
//     depend on the common scale) unless the maximum rises by more than 64 octaves, which is detected (wave-uniform) and
//     answered by one more pass against the true maximum;
//   * rounding to the probability grid is folded into the normalising FMA: fma(e, inv, ubias + 1.5*2^23);
//   * the sums of the probability codes (needed for the v zero point) come out of the P.V MFMA itself: the lane that
//     holds V^T row `d` (a padding row: d is not a multiple of 32) reads a constant row of ones instead, so column d
//     of the output tile is sum_j code_j — no v_dot4 in the loop.
__device__ __attribute__((aligned(16))) int qd_ones16[4] = {0x01010101, 0x01010101, 0x01010101, 0x01010101};

constexpr float QD_MAGIC  = 12582912.f;                        // 1.5 * 2^23
constexpr int   QD_MAGICI = 0x4B400000;                        // its bit pattern

struct AttnRowRef { float nc, eps; };                          // x = fma(F, cs2, nc);  eps = (1.5*2^23 + m)*cs2 + nc, exactly
__device__ __forceinline__ AttnRowRef attn_rowref(int m, float cs2) {
#pragma clang fp contract(off)
    const float fm = QD_MAGIC + (float)m;                      // exact: |m| < 2^22
    AttnRowRef r;
    r.nc = -(fm * cs2);
    r.eps = __builtin_fmaf(fm, cs2, r.nc);                     // the rounding error of a product is a float
    return r;
}
// End of sweep 1 (shared by the lean and the LDS-staged kernel so that both evaluate the SAME float expressions): `l` = this
// half-wave's sum of exp2(cs2*(s - m0) + eps0), `mxn` = how far this half's maximum lies above m0 (may be negative: m0 is the
// maximum of tile 0 over BOTH halves).  Returns the row maximum mi, the reference of sweep 2, and 1 / (normaliser * dw) such
// that e2 * inv = p / dw for e2 = exp2(fma(F, cs2, ref2.nc)).
struct AttnNorm { int mi; AttnRowRef ref; float inv, emax; };
__device__ __forceinline__ AttnNorm attn_finish_stats(float l, int mxn, int m0, const AttnRowRef& ref0, float cs2, float dw) {
#pragma clang fp contract(off)
    AttnNorm n;
    const float lo = __shfl_xor(l, 32);
    const int mx = max(mxn, __shfl_xor(mxn, 32));               // >= 0: the half that supplied m0 has mxn >= 0
    n.mi = m0 + mx;
    n.ref = attn_rowref(n.mi, cs2);
    const float lt = l + lo;                                    // both halves are sums against the same m0 (and the same eps0)
    const float ls = lt * __builtin_amdgcn_exp2f((n.ref.eps - ref0.eps) - (float)mx * cs2);   // = 2^eps(mi) * sum 2^(cs2*(s - mi))
    n.inv = 1.0f / (ls * dw);
    n.emax = __builtin_amdgcn_exp2f(n.ref.eps);                 // e2 of the row maximum
    return n;
}

template <int DT, bool P16, int KT>
__device__ __forceinline__ void attn_lean_body(const AttnK& p) {
    // a*b+c written as such stays unfused: hipcc's default -ffp-contract=fast lets the optimiser fuse (or not) depending on the
    // surrounding code, and the lean and the LDS-staged bodies must produce the same normaliser bit for bit
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 31, half = lane >> 5;
    const int lblk = p.xcd ? qd_xcd_remap(blockIdx.x, p.gx * p.BH) : (int)blockIdx.x;
    const int bh = lblk / p.gx;
    const int q0 = ((lblk - bh * p.gx) * 4 + wave) * 32;
    if (q0 >= p.T) return;

    const float cs2 = p.prm[0] * 1.4426950408889634f;
    const int nzq = -(int)p.prm[1];
    const float dw = p.prm[3], zpw = p.prm[4], oscale = p.prm[5];
    const int zv = (int)p.prm[6];
    const int izpw = (int)zpw;
    const float urange = p.wmax - p.wmin;
    const float ubias = zpw - p.wmin;
    constexpr float MAGIC = QD_MAGIC;
    constexpr int   MAGICI = QD_MAGICI;

    v4i qf[DT];
    const int8_t* qrow = p.q + ((long)bh * p.Tpad + q0 + frow) * p.dpad + half * 16;
#pragma unroll
    for (int kk = 0; kk < DT; ++kk) qf[kk] = *reinterpret_cast<const v4i*>(qrow + kk * 32);
    const int8_t* kbase = p.k + (long)bh * p.Spad * p.dpad + (long)frow * p.dpad + half * 16;
    const int32_t* tbase = KT == 2 ? p.kterm + (long)bh * p.Spad + half * 4 : nullptr;
    const int c1 = nzq > 127 ? 64 : nzq, c2 = nzq - c1;           // KT == 1: -zq' in [-127, 128] as one or two int8 constants
    const int c1w = (c1 & 0xff) * 0x01010101, c2w = (c2 & 0xff) * 0x01010101;
    const v4i c1v = {c1w, c1w, c1w, c1w}, c2v = {c2w, c2w, c2w, c2w};
    const int ntile = p.Spad >> 5;
    const int tail_tile = (p.S & 31) ? ntile - 1 : ntile;

    auto load_k = [&](int jt, v4i (&kf)[DT]) __attribute__((always_inline)) {
        const int8_t* kp = kbase + (long)jt * 32 * p.dpad;
#pragma unroll
        for (int kk = 0; kk < DT; ++kk) kf[kk] = *reinterpret_cast<const v4i*>(kp + kk * 32);
    };
    // seeds of the 16 accumulator registers of this lane for tile jt: register 4g+e <-> key jt*32 + e + 8g + 4*half
    auto load_t = [&](int jt, v16i& ti) __attribute__((always_inline)) {
        if (KT == 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i w = *reinterpret_cast<const v4i*>(tbase + jt * 32 + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) ti[4 * g + e] = w[e];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) ti[r] = MAGICI;
        }
    };
    // acc[4g+e] = 0x4B400000 + (score of key jt*32 + e + 8g + 4*half), per-query constants dropped
    auto scores = [&](const v4i (&kf)[DT], const v16i& ti, v16i& acc) __attribute__((always_inline)) {
        acc = ti;
#pragma unroll
        for (int kk = 0; kk < DT; ++kk) {
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[kk], qf[kk], acc, 0, 0, 0);
            if (KT == 1) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[kk], c1v, acc, 0, 0, 0);
        }
        if (KT == 1 && c2 != 0) {                                  // wave-uniform, only when zq' == -128
#pragma unroll
            for (int kk = 0; kk < DT; ++kk) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[kk], c2v, acc, 0, 0, 0);
        }
    };
    auto key_ok = [&](int jt, int r) __attribute__((always_inline)) { return jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < p.S; };
    const v2f cs2v = {cs2, cs2};

    // ---- sweep 1 ----------------------------------------------------------------------------------------------------
    AttnNorm nrm;
    {
        v4i kf[DT], kfn[DT];
        v16i ti;
        load_k(0, kf);
        load_t(0, ti);
        int m0;
        {
            v16i acc;
            scores(kf, ti, acc);
            m0 = 0;                                               // raw accumulators are > 0; 0 = masked
#pragma unroll
            for (int r = 0; r < 16; ++r) m0 = max(m0, (tail_tile == 0 && !key_ok(0, r)) ? 0 : acc[r]);
            m0 -= MAGICI;
            m0 = max(m0, __shfl_xor(m0, 32));                     // one reference for both halves of the query row
        }
        float l = 0.f;
        int mxn = 0;
        AttnRowRef ref0;
        for (int pass = 0; pass < 2; ++pass) {
            ref0 = attn_rowref(m0, cs2);
            const v2f ncv = {ref0.nc, ref0.nc};
            int mxa = 0;                                          // max of the raw accumulators (all > 0: 0 is "masked")
            v2f a2 = {0.f, 0.f};
            auto s1_tile = [&](int jt, auto tail_tag) __attribute__((always_inline)) {
                constexpr bool tail = decltype(tail_tag)::value;
                if (jt + 1 < ntile) load_k(jt + 1, kfn);
                v16i acc;
                scores(kf, ti, acc);
                if (KT == 2 && jt + 1 < ntile) load_t(jt + 1, ti);
                if (tail) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) if (!key_ok(jt, r)) acc[r] = 0;
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    mxa = max(max(mxa, acc[r]), acc[r + 1]);              // v_max3_i32
                    const v2f F = {__int_as_float(acc[r]), __int_as_float(acc[r + 1])};
                    const v2f x = __builtin_elementwise_fma(F, cs2v, ncv);
                    v2f e = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                    if (tail) {
                        if (acc[r] == 0) e.x = 0.f;
                        if (acc[r + 1] == 0) e.y = 0.f;
                    }
                    a2 += e;
                }
                if (jt + 1 < ntile) {
#pragma unroll
                    for (int kk = 0; kk < DT; ++kk) kf[kk] = kfn[kk];
                }
            };
            for (int jt = 0; jt < tail_tile; ++jt) s1_tile(jt, std::false_type{});
            if (tail_tile < ntile) s1_tile(tail_tile, std::true_type{});
            l = a2.x + a2.y;
            mxn = mxa ? mxa - MAGICI - m0 : 0;                    // this half's maximum relative to m0; a half without any valid key: 0, l = 0
            const int mxm = max(mxn, __shfl_xor(mxn, 32));
            if (!__any((float)mxm * cs2 > 64.f)) break;
            m0 += mxm;                                            // (rare) start again against the true maximum (the same in both halves)
            load_k(0, kf);
            load_t(0, ti);
        }
        nrm = attn_finish_stats(l, mxn, m0, ref0, cs2, dw);
    }
    const int mi = nrm.mi;
    const float inv = nrm.inv, emax = nrm.emax;
    (void)mi;

    // ---- sweep 2 ----------------------------------------------------------------------------------------------------
    v16i ol[DT], oh[P16 ? DT : 1];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ol[t][r] = 0;
            if (P16) oh[P16 ? t : 0][r] = 0;
        }
    const int t1 = p.d >> 5, frow1 = p.d & 31;                     // where the row of ones sits
    const int8_t* vp[DT];
    int vstep[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        const bool ones = (t == t1) && (frow == frow1);
        vp[t] = ones ? reinterpret_cast<const int8_t*>(qd_ones16) : p.vt + ((long)bh * p.dpad + t * 32 + frow) * p.Spad + half * 16;
        vstep[t] = ones ? 0 : 32;
    }
    {
        v4i kf[DT], kfn[DT];
        v16i ti;
        load_k(0, kf);
        load_t(0, ti);
        const v2f ncv = {nrm.ref.nc, nrm.ref.nc};
        const v2f invv = {inv, inv}, ubv = {ubias, ubias}, magic = {MAGIC, MAGIC}, ubm = {ubias + MAGIC, ubias + MAGIC};
        auto s2_tile = [&](int jt, auto tail_tag, auto clamp_tag) __attribute__((always_inline)) {
            constexpr bool tail = decltype(tail_tag)::value, CLAMP = decltype(clamp_tag)::value;
            v4i vf[DT];
#pragma unroll
            for (int t = 0; t < DT; ++t) vf[t] = *reinterpret_cast<const v4i*>(vp[t] + (long)jt * vstep[t]);
            if (jt + 1 < ntile) load_k(jt + 1, kfn);
            v16i acc;
            scores(kf, ti, acc);
            if (KT == 2 && jt + 1 < ntile) load_t(jt + 1, ti);
            unsigned ub[16];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const v2f F = {__int_as_float(acc[r]), __int_as_float(acc[r + 1])};
                const v2f x = __builtin_elementwise_fma(F, cs2v, ncv);
                const v2f e = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                v2f t;
                if (CLAMP) {
                    t = __builtin_elementwise_fma(e, invv, ubv);
                    t.x = fminf(t.x, urange);
                    t.y = fminf(t.y, urange);
                    t += magic;
                } else {
                    t = __builtin_elementwise_fma(e, invv, ubm);  // one rounding: half-even on the exact e*inv
                }
                ub[r] = __float_as_uint(t.x);
                ub[r + 1] = __float_as_uint(t.y);
            }
            if (tail) {
#pragma unroll
                for (int r = 0; r < 16; ++r) if (!key_ok(jt, r)) ub[r] = 0x8080u;
            }
            v4i plo, phi;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned a01 = __builtin_amdgcn_perm(ub[4 * g + 1], ub[4 * g], 0x05010400u);
                const unsigned a23 = __builtin_amdgcn_perm(ub[4 * g + 3], ub[4 * g + 2], 0x05010400u);
                plo[g] = (int)(__builtin_amdgcn_perm(a23, a01, 0x05040100u) ^ 0x80808080u);
                phi[g] = P16 ? (int)(__builtin_amdgcn_perm(a23, a01, 0x07060302u) ^ 0x80808080u) : 0;
            }
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                ol[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(plo, vf[t], ol[t], 0, 0, 0);
                if (P16) oh[P16 ? t : 0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(phi, vf[t], oh[P16 ? t : 0], 0, 0, 0);
            }
            if (jt + 1 < ntile) {
#pragma unroll
                for (int kk = 0; kk < DT; ++kk) kf[kk] = kfn[kk];
            }
        };
        // (an odd ubias would send exact ties of e*inv to the even SUM: keep the two-step rounding of the clamp path then)
        if (__any(!(emax * inv * 1.0001f + ubias + 0.5f <= urange)) || (((int)ubias) & 1)) {
            for (int jt = 0; jt < tail_tile; ++jt) s2_tile(jt, std::false_type{}, std::true_type{});
        } else {
            for (int jt = 0; jt < tail_tile; ++jt) s2_tile(jt, std::false_type{}, std::false_type{});
        }
        if (tail_tile < ntile) s2_tile(tail_tile, std::true_type{}, std::true_type{});
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    // code sums of query il(r, half): column d of the output tile, i.e. register r of lane frow1 + 32*half
    int us[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int sl = 0, sh = 0;
#pragma unroll
        for (int t = 0; t < DT; ++t)
            if (t == t1) {
                sl = ol[t][r];
                if (P16) sh = oh[P16 ? t : 0][r];
            }
        sl = __shfl(sl, frow1 + 32 * half);
        if (P16) sh = __shfl(sh, frow1 + 32 * half);
        // sum over the valid keys of uu = 256*hi + lo from the signed operand bytes, then of u = uu + wmin
        us[r] = sl + 128 * p.S + (P16 ? 256 * (sh + 128 * p.S) : 0) + p.S * p.iwmin;
    }
    attn_write_rows<DT, P16>(p, ol, oh, us, P16, bh, q0, frow, half, dw, zpw, oscale, zv);
}

#ifndef QD_ATTN_LEAN_OCC
#define QD_ATTN_LEAN_OCC 3
#endif
template <int DT, bool P16, int KT>
__global__ __launch_bounds__(256, DT <= 2 ? QD_ATTN_LEAN_OCC : 2) void attn_lean_kernel(const AttnK p) { attn_lean_body<DT, P16, KT>(p); }

constexpr int QD_ONES_ROW = 16384;                             // longest padded key axis the LDS-staged kernel takes (bytes of ones)
struct OnesRow {                                               // constant-initialised: lives in the code object's data segment
    int v[QD_ONES_ROW / 4 + 4];
    constexpr OnesRow() : v() {
        for (int i = 0; i < QD_ONES_ROW / 4 + 4; ++i) v[i] = 0x01010101;
    }
};
__device__ __attribute__((aligned(16))) OnesRow qd_ones_row_obj = OnesRow();   // not `const`: a constant-address-space pointer would turn the V loads into flat loads
#define qd_ones_row (qd_ones_row_obj.v)

// ---- LDS-staged variant (round 3; measurements: profiles/r03_attn_ablation.md) ---------------------------------------------
// The 4 waves of a block share each K / V^T tile through LDS instead of loading every fragment themselves: one 16-byte-per-
// lane DMA instruction per wave per tile into a 4-stage ring (global_load_lds_dwordx4, no VGPR round trip), one barrier per
// tile; fragments come from LDS by conflict-free ds_read_b128 (XOR swizzle applied to the DMA's SOURCE chunk, as in
// igemm_dma.hip).  The vector-memory return path of a CU (~46 B/clk measured: re-reading tile 0 for every tile costs the same
// as the real stream) carries a quarter of the bytes.  The tile loop is software-pipelined one tile deep: the P.V MFMAs of
// tile j-1 (independent accumulators) sit between the probability VALU of tile j, the score chain of tile j+1 follows back
// to back.  On top (P16 only, exact): when no row of the wave can produce a probability code >= 256 (emax * inv + ubias <
// 255.5 for all lanes — the common case with thousands of keys and unpeaked rows), every hi operand byte is the constant
// 0x80: the hi MFMAs and their v_perm / v_xor are skipped and the epilogue uses the 8-bit zero-point constants.
// Arithmetic, operand order and summation order are those of attn_lean_kernel: bit-identical results
// (tests/test_hip_kernels.py::test_attention_lds_equals_lean).
// What it buys, honestly: 64x64 self-attention of SD (T = S = 4096, d = 40, 128 heads) 1170 -> 1047 us on unpeaked rows
// (-10 %, most of it the skipped hi MFMAs), 1180 -> 1143 us on peaked rows; short key axes (77-token cross-attention) are
// SLOWER (ring start-up, barriers) and stay on attn_lean_kernel.  What did NOT help, all measured on the same shape: a
// register-fed pipeline with every MFMA spread between VALU slices at 2 waves per SIMD (1200 us), the same spreading in
// this kernel at 3 waves per SIMD with the hi part computed in a second sweep (1020 us unpeaked, 1650 us peaked), a one-sweep
// formulation (needs the 16 x 4096 scores of a block in registers and 7 bytes of K / V per score from L1: > 64 B/clk per CU).
// Counters of every variant say the same thing: the matrix pipe is busy ~32 % and the vector pipe ~27 % of a SIMD's cycles and
// the two barely overlap — the kernel is bound by in-order instruction issue of the ~135 non-MFMA instructions per tile pair.
__device__ __forceinline__ void attn_glds16(const void* gsrc, unsigned lds_base) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_base)
        : "memory");
}
template <int N>
__device__ __forceinline__ void attn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void attn_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ---- round 6: the LDS-staged path as TWO kernels (statistics, then P.V) -------------------------------------------------------
// What bounds the 4096-token call is in-order VALU ISSUE at low occupancy, not a pipe: one wave issues a vector instruction every
// ~5.5 cycles, two waves per SIMD one every ~3.2 - 3.9 (v_exp_f32 6.5), three one every ~2.5 (profiles/r03_ubench_issue.txt) —
// and the one-kernel form (sweep 1 + sweep 2 in one body: 233 VGPRs for the two score sets, the hi + lo output accumulators and
// the one-tile-deep software pipeline) ran at TWO.  Registers are allocated for the worst moment of a kernel, so the sweeps are
// separate kernels now:
//   * attn_stats_kernel  — sweep 1 only: one score set, no output accumulators (~64 VGPRs -> up to 8 waves per SIMD).  Writes
//     per query {nc, inv, emax} (what attn_finish_stats returns: the sweep-2 reference, 1 / (normaliser * dw), e2 of the row
//     maximum) and per block whether ANY of its waves needs the upper clamp or the hi operand bytes of the 16-bit codes;
//   * attn_pv_kernel<.., FULL = false> — sweep 2 for blocks where no wave does (the common case with thousands of keys): lo
//     bytes only, ONE 32 x 64 accumulator pair, no software pipeline -> <= 128 VGPRs, 4 waves per SIMD;
//   * attn_pv_kernel<.., FULL = true>  — sweep 2 for the other blocks (hi + lo accumulators, <= 168 VGPRs, 3 waves per SIMD); each
//     wave picks clamp / hi exactly as the one-kernel form did.  A block runs in exactly one of the two launches (the other one
//     returns after reading its flag).
// The latency of a wave's MFMA -> v_exp -> pack -> MFMA chain is covered by the OTHER waves of the SIMD instead of by a second
// accumulator set of the same wave.  Arithmetic, operand order and summation order are attn_lean_kernel's: bit-identical
// (tests/test_hip_kernels.py::test_attention_lds_equals_lean).
struct AttnStat { float nc, inv, emax; int pad; };            // 16 bytes per query: [BH][Tpad]

template <int DT, int KT, bool WITH_V>
struct AttnRing {
    static constexpr int NST = 4;                             // ring stages (tiles)
    static constexpr int PD = 3;                              // tiles in flight ahead of the one being consumed
    static constexpr int KB = 1024 * DT;                      // K tile: 32 keys x dpad bytes
    static constexpr int VB = WITH_V ? 1024 * DT : 0;         // V^T tile: dpad rows x 32 keys
    static constexpr int TB = KT == 2 ? 128 : 0;              // the tile's 32 accumulator seeds (qd_attn_keyterm)
    static constexpr int STAGE = KB + VB + TB;
    static constexpr int CPRK = 2 * DT;                       // 16-byte chunks per K row (dpad / 16)
    unsigned char* smem;
    unsigned lds0, dma_dst;
    const int8_t* ksrc;
    const int8_t* vsrc;
    const int32_t* tsrc;
    bool dma_k, dma_v, dma_t, lane8;
    int ntile;
    unsigned koff[DT], voff[DT], toff;

    // DMA role of a wave: slot s < DT copies 1 KB of the K tile, DT <= s < 2*DT 1 KB of the V^T tile; wave 3 also copies the 128
    // bytes of accumulator seeds (lanes 0-7, issued BEFORE its V^T copy: vmcnt retires in order, so the uniform "all but the
    // newest PD - 1 tiles" wait over-waits on wave 3, which is safe).  The LDS side is lane-linear (chunk c = slot*64 + lane);
    // the bank swizzle lives in the SOURCE chunk index (igemm_dma.hip does the same).
    __device__ __forceinline__ AttnRing(const AttnK& p, unsigned char* sm, int bh, int wave, int lane) {
        smem = sm;
        ntile = p.Spad >> 5;
        const int frow = lane & 31, half = lane >> 5;
        dma_k = wave < DT;
        dma_v = WITH_V && wave >= DT && wave < 2 * DT;
        dma_t = KT == 2 && wave == 3;
        lane8 = lane < 8;
        tsrc = KT == 2 ? p.kterm + (long)bh * p.Spad + (lane & 7) * 4 : nullptr;
        {
            const int c = (dma_k ? wave : 0) * 64 + lane;
            const int row = c / CPRK, pos = c % CPRK;
            const int sw = CPRK == 4 ? ((row >> 2) & 3) : ((row >> 3) & 1);
            ksrc = p.k + ((long)bh * p.Spad + row) * p.dpad + (pos ^ sw) * 16;
            const int cv = (dma_v ? wave - DT : 0) * 64 + lane;
            const int vrow = cv >> 1, vpos = cv & 1;
            const int vsw = (vrow >> 3) & 1;
            vsrc = (vrow == p.d) ? reinterpret_cast<const int8_t*>(qd_ones_row) + (vpos ^ vsw) * 16        // the padding row d reads ones
                                 : p.vt + ((long)bh * p.dpad + vrow) * p.Spad + (vpos ^ vsw) * 16;
        }
        lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)(sm));
        dma_dst = dma_k ? wave * 1024 : KB + (wave - DT) * 1024;
#pragma unroll
        for (int kk = 0; kk < DT; ++kk) {
            const int sw = CPRK == 4 ? ((frow >> 2) & 3) : ((frow >> 3) & 1);
            koff[kk] = frow * (16 * CPRK) + (((kk * 2 + half) ^ sw) * 16);
            const int vrow = kk * 32 + frow;
            voff[kk] = KB + vrow * 32 + ((half ^ ((vrow >> 3) & 1)) * 16);
        }
        toff = KB + VB + half * 16;
    }
    // Stage of tile jt: jt & (NST - 1) — a compile-time constant ST when the caller unrolls the tile loop NST-fold (the stage
    // offset then folds into the ds_read / m0 immediates: no address arithmetic per tile), ST = -1: computed.
    template <int ST>
    __device__ __forceinline__ unsigned stage_of(int jt) const { return ST >= 0 ? (unsigned)ST * STAGE : (unsigned)(jt & (NST - 1)) * STAGE; }
    // this wave's copies for tile `jt`; past the end: tile ntile-1 again (harmless, keeps the vmcnt bookkeeping uniform)
    template <int ST = -1>
    __device__ __forceinline__ void issue(int jt) const {
        const int j = min(jt, ntile - 1);
        const unsigned st = lds0 + stage_of<ST>(jt);
        if (dma_t && lane8) attn_glds16(tsrc + (long)j * 32, st + KB + VB);
        if (dma_k) attn_glds16(ksrc + (long)j * KB, st + dma_dst);
        else if (dma_v) attn_glds16(vsrc + (long)j * 32, st + dma_dst);
    }
    template <int ST = -1>
    __device__ __forceinline__ void read_k(int jt, v4i (&kf)[DT]) const {
        const unsigned char* sp = smem + stage_of<ST>(jt);
#pragma unroll
        for (int kk = 0; kk < DT; ++kk) kf[kk] = *reinterpret_cast<const v4i*>(sp + koff[kk]);
    }
    template <int ST = -1>
    __device__ __forceinline__ void read_v(int jt, v4i (&vf)[DT]) const {
        const unsigned char* sp = smem + stage_of<ST>(jt);
#pragma unroll
        for (int t = 0; t < DT; ++t) vf[t] = *reinterpret_cast<const v4i*>(sp + voff[t]);
    }
    // seeds of this lane's 16 accumulator registers for tile jt: register 4g+e <-> key e + 8g + 4*half of the tile, i.e. the
    // 16-byte word 2g + half of the tile's table (two distinct addresses per ds_read_b128: broadcast)
    template <int ST = -1>
    __device__ __forceinline__ void read_t(int jt, v16i& ti) const {
        if (KT == 2) {
            const unsigned char* sp = smem + stage_of<ST>(jt) + toff;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i w = *reinterpret_cast<const v4i*>(sp + g * 32);
#pragma unroll
                for (int e = 0; e < 4; ++e) ti[4 * g + e] = w[e];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) ti[r] = QD_MAGICI;
        }
    }
    // start of a sweep / pass: tiles 0 .. PD-1 in flight, tile 0 landed and visible to every wave
    __device__ __forceinline__ void prologue() const {
        attn_wait_vmcnt<0>();                                 // run-ahead copies of a previous pass must not land behind the new tiles
        __syncthreads();                                      // nobody still reads the ring of the previous pass
#pragma unroll
        for (int j = 0; j < PD; ++j) issue(j);
        attn_wait_vmcnt<PD - 1>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                        // no LDS read of the new tiles is hoisted above the barrier
    }
    // head of iteration jt: tile jt has landed for every wave (the PD - 1 newer ones may still be in flight) and every wave is
    // past iteration jt-1, whose stage tile jt+PD now overwrites
    template <int ST = -1>
    __device__ __forceinline__ void step_sync(int jt) const {
        attn_wait_vmcnt<PD - 1>();
        attn_wait_lgkm0();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue<(ST >= 0 ? (ST + PD) & (NST - 1) : -1)>(jt + PD);
    }
};

template <int DT, int KT>
__device__ __forceinline__ void attn_qk(const v4i (&kf)[DT], const v4i (&qf)[DT], v16i& acc) {
    static_assert(KT != 1, "the LDS-staged kernels take the per-key term from the table (or none)");
#pragma unroll
    for (int kk = 0; kk < DT; ++kk) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[kk], qf[kk], acc, 0, 0, 0);
}

#ifndef QD_ATTN_STATS_OCC
#define QD_ATTN_STATS_OCC 4                                   // waves per SIMD the statistics kernel is compiled for
#endif
#ifndef QD_ATTN_PV_OCC
#define QD_ATTN_PV_OCC 4                                      // ... the lo-only P.V kernel (<= 128 VGPRs)
#endif
#ifndef QD_ATTN_PVFULL_OCC
#define QD_ATTN_PVFULL_OCC 3                                  // ... the hi + lo P.V kernel (<= 168 VGPRs)
#endif
// hi + lo P.V kernel: vector instructions between two of the previous tile's P.V MFMAs (measured on one box, us per 4096-token call:
// undeferred 961, all six MFMAs of a tile in one cluster 961, 8 / 15 / 22 instructions apart 940 / 935 / 940 — r06_c18)
constexpr int QD_ATTN_PV_DEFER_VALU = 15;

template <int DT, int KT>
__global__ __launch_bounds__(256, QD_ATTN_STATS_OCC) void attn_stats_kernel(const AttnK p, AttnStat* __restrict__ stat, int* __restrict__ blkflag, int p16) {
    // a*b+c written as such stays unfused: the lean and the LDS-staged bodies must produce the same normaliser bit for bit
#pragma clang fp contract(off)
    using Ring = AttnRing<DT, KT, false>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[Ring::NST * Ring::STAGE];
    __shared__ int s_flag[2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int frow = lane & 31, half = lane >> 5;
    const int lblk = p.xcd ? qd_xcd_remap(blockIdx.x, p.gx * p.BH) : (int)blockIdx.x;
    const int bh = lblk / p.gx;
    const int q0r = ((lblk - bh * p.gx) * 4 + wave) * 32;
    const bool live = q0r < p.T;                              // a wave past the last query still serves the DMA ring and the barriers
    const int q0 = live ? q0r : 0;
    const float cs2 = p.prm[0] * 1.4426950408889634f;
    const float dw = p.prm[3], zpw = p.prm[4];
    const float urange = p.wmax - p.wmin;
    const float ubias = zpw - p.wmin;
    constexpr int MAGICI = QD_MAGICI;

    v4i qf[DT];
    const int8_t* qrow = p.q + ((long)bh * p.Tpad + q0 + frow) * p.dpad + half * 16;
#pragma unroll
    for (int kk = 0; kk < DT; ++kk) qf[kk] = *reinterpret_cast<const v4i*>(qrow + kk * 32);
    const int ntile = p.Spad >> 5;
    const int nfull = (p.S & 31) ? ntile - 1 : ntile;
    const Ring ring(p, smem, bh, wave, lane);
    auto key_ok = [&](int jt, int r) __attribute__((always_inline)) { return jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < p.S; };
    const v2f cs2v = {cs2, cs2};

    v4i kf[DT];
    v16i acc;
    if (threadIdx.x == 0) s_flag[1] = 0;
    ring.prologue();
    ring.read_k(0, kf);
    ring.read_t(0, acc);
    attn_qk<DT, KT>(kf, qf, acc);
    int m0 = 0;                                               // raw accumulators are > 0; 0 = masked
#pragma unroll
    for (int r = 0; r < 16; ++r) m0 = max(m0, (nfull == 0 && !key_ok(0, r)) ? 0 : acc[r]);
    m0 -= MAGICI;
    m0 = max(m0, __shfl_xor(m0, 32));                         // one reference for both halves of the query row
    float l = 0.f;
    int mxn = 0;
    AttnRowRef ref0 = attn_rowref(m0, cs2);
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) ring.prologue();
        const AttnRowRef refp = attn_rowref(m0, cs2);
        const v2f ncv = {refp.nc, refp.nc};
        int mxa = 0;
        v2f a2 = {0.f, 0.f};
        auto tile = [&](int jt, auto tail_tag, auto st_tag) __attribute__((always_inline)) {
            constexpr bool tail = decltype(tail_tag)::value;
            constexpr int ST = decltype(st_tag)::value;
            ring.template step_sync<ST>(jt);
            ring.template read_k<ST>(jt, kf);
            ring.template read_t<ST>(jt, acc);
            attn_qk<DT, KT>(kf, qf, acc);
            if (tail) {
#pragma unroll
                for (int r = 0; r < 16; ++r) if (!key_ok(jt, r)) acc[r] = 0;
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                mxa = max(max(mxa, acc[r]), acc[r + 1]);
                const v2f F = {__int_as_float(acc[r]), __int_as_float(acc[r + 1])};
                const v2f x = __builtin_elementwise_fma(F, cs2v, ncv);
                v2f e = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                if (tail) {
                    if (acc[r] == 0) e.x = 0.f;
                    if (acc[r + 1] == 0) e.y = 0.f;
                }
                a2 += e;
            }
        };
        int jt = 0;
        for (; jt + 4 <= nfull; jt += 4) {                    // four tiles = one turn of the ring: stage offsets are immediates
            tile(jt, std::false_type{}, std::integral_constant<int, 0>{});
            tile(jt + 1, std::false_type{}, std::integral_constant<int, 1>{});
            tile(jt + 2, std::false_type{}, std::integral_constant<int, 2>{});
            tile(jt + 3, std::false_type{}, std::integral_constant<int, 3>{});
        }
        for (; jt < nfull; ++jt) tile(jt, std::false_type{}, std::integral_constant<int, -1>{});
        if (nfull < ntile) tile(nfull, std::true_type{}, std::integral_constant<int, -1>{});
        const int mxh = mxa ? mxa - MAGICI - m0 : 0;          // this half's maximum relative to m0
        const int mxm = max(mxh, __shfl_xor(mxh, 32));
        const bool again = (float)mxm * cs2 > 64.f;           // this row's maximum rose by > 64 octaves over tile 0's
        if (pass == 0) {
            l = a2.x + a2.y;
            mxn = mxh;
            // the repeat pass runs the ring and the barriers again: the decision is taken for the whole BLOCK; a wave that did
            // not need it keeps its first-pass statistics (what attn_lean_kernel computes for it)
            if (threadIdx.x == 0) s_flag[0] = 0;
            __syncthreads();
            if (__any(again) && lane == 0) s_flag[0] = 1;
            __syncthreads();
            if (!s_flag[0]) break;
            if (__any(again)) {                                // this wave repeats against the true maximum
                m0 += mxm;
                l = -1.f;                                      // marker: take the second pass's statistics
            }
        } else if (l < 0.f) {
            l = a2.x + a2.y;
            mxn = mxh;
            ref0 = refp;
        }
    }
    attn_wait_vmcnt<0>();                                     // the ring's run-ahead copies: nothing may land in LDS after the block retires
    const AttnNorm nrm = attn_finish_stats(l, mxn, m0, ref0, cs2, dw);
    // what sweep 2 decides per WAVE (attn_pv_kernel evaluates the same two expressions on the stored values)
    const bool need_clamp = __any(!(nrm.emax * nrm.inv * 1.0001f + ubias + 0.5f <= urange)) || (((int)ubias) & 1);
    const bool hi_live = p16 && (need_clamp || __any(!(nrm.emax * nrm.inv * 1.0001f + ubias + 0.5f < 256.f)));
    if (live && (need_clamp || hi_live) && lane == 0) s_flag[1] = 1;
    if (live && half == 0) stat[(long)bh * p.Tpad + q0 + frow] = AttnStat{nrm.ref.nc, nrm.inv, nrm.emax, 0};
    __syncthreads();
    if (threadIdx.x == 0) blkflag[lblk] = s_flag[1];
}

template <int DT, bool P16, int KT, bool FULL>
__global__ __launch_bounds__(256, FULL ? QD_ATTN_PVFULL_OCC : QD_ATTN_PV_OCC) void attn_pv_kernel(const AttnK p, const AttnStat* __restrict__ stat,
                                                                                                     const int* __restrict__ blkflag) {
#pragma clang fp contract(off)
    using Ring = AttnRing<DT, KT, true>;
    constexpr bool HIK = FULL && P16;                         // this kernel owns hi accumulators
    __shared__ __attribute__((aligned(16))) unsigned char smem[Ring::NST * Ring::STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int frow = lane & 31, half = lane >> 5;
    const int lblk = p.xcd ? qd_xcd_remap(blockIdx.x, p.gx * p.BH) : (int)blockIdx.x;
    if ((blkflag[lblk] != 0) != FULL) return;                 // block-uniform: the other launch owns this block
    const int bh = lblk / p.gx;
    const int q0r = ((lblk - bh * p.gx) * 4 + wave) * 32;
    const bool live = q0r < p.T;
    const int q0 = live ? q0r : 0;
    const float cs2 = p.prm[0] * 1.4426950408889634f;
    const float dw = p.prm[3], zpw = p.prm[4], oscale = p.prm[5];
    const int zv = (int)p.prm[6];
    const float urange = p.wmax - p.wmin;
    const float ubias = zpw - p.wmin;
    constexpr float MAGIC = QD_MAGIC;

    v4i qf[DT];
    const int8_t* qrow = p.q + ((long)bh * p.Tpad + q0 + frow) * p.dpad + half * 16;
#pragma unroll
    for (int kk = 0; kk < DT; ++kk) qf[kk] = *reinterpret_cast<const v4i*>(qrow + kk * 32);
    const AttnStat st = stat[(long)bh * p.Tpad + q0 + frow];
    const float inv = st.inv, emax = st.emax;
    const int ntile = p.Spad >> 5;
    const int nfull = (p.S & 31) ? ntile - 1 : ntile;
    const Ring ring(p, smem, bh, wave, lane);
    auto key_ok = [&](int jt, int r) __attribute__((always_inline)) { return jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < p.S; };
    const v2f cs2v = {cs2, cs2};

    v16i ol[DT], oh[HIK ? DT : 1];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ol[t][r] = 0;
            if (HIK) oh[HIK ? t : 0][r] = 0;
        }
    const int t1 = p.d >> 5, frow1 = p.d & 31;
    const bool need_clamp = FULL && (__any(!(emax * inv * 1.0001f + ubias + 0.5f <= urange)) || (((int)ubias) & 1));
    const bool hi_live = HIK && (need_clamp || __any(!(emax * inv * 1.0001f + ubias + 0.5f < 256.f)));
    {
        const v2f ncv = {st.nc, st.nc};
        const v2f invv = {inv, inv}, ubv = {ubias, ubias}, magic = {MAGIC, MAGIC}, ubm = {ubias + MAGIC, ubias + MAGIC};
        ring.prologue();
        // tile jt: K / V^T fragments and seeds from the ring, the score chain, the probability chain, the P.V MFMAs — one
        // dependent chain per wave; the other waves of the SIMD fill its gaps
        // (lo bytes only, no clamp: the operand byte is code - 128 = code ^ 0x80 for codes < 256, i.e. the low byte of code + 128 —
        //  the 128 rides in the FMA's constant (even, like MAGIC: ties still go to the even code) and the v_xor disappears)
        const v2f ubm128 = {ubias + MAGIC + 128.f, ubias + MAGIC + 128.f};
        // the probability chain of one 32 x 32 score tile: 16 accumulators of this lane -> four operand words of lo (and hi) bytes
        auto chain = [&](const v16i& acc, int jt, auto tail_tag, auto clamp_tag, auto hi_tag, v4i& plo, v4i& phi) __attribute__((always_inline)) {
            constexpr bool tail = decltype(tail_tag)::value, CLAMP = decltype(clamp_tag)::value, HI = decltype(hi_tag)::value;
            constexpr bool BIAS128 = !tail && !CLAMP && !HI;
            unsigned ub[16];
#pragma unroll
            for (int sidx = 0; sidx < 8; ++sidx) {
                const int r = 2 * sidx;
                const v2f F = {__int_as_float(acc[r]), __int_as_float(acc[r + 1])};
                const v2f x = __builtin_elementwise_fma(F, cs2v, ncv);
                const v2f e = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                v2f t2;
                if (CLAMP) {
                    t2 = __builtin_elementwise_fma(e, invv, ubv);
                    t2.x = fminf(t2.x, urange);
                    t2.y = fminf(t2.y, urange);
                    t2 += magic;
                } else {
                    t2 = __builtin_elementwise_fma(e, invv, BIAS128 ? ubm128 : ubm);     // one rounding: half-even on the exact e*inv
                }
                ub[r] = __float_as_uint(t2.x);
                ub[r + 1] = __float_as_uint(t2.y);
                if (tail) {
                    if (!key_ok(jt, r)) ub[r] = 0x8080u;
                    if (!key_ok(jt, r + 1)) ub[r + 1] = 0x8080u;
                }
                if (sidx & 1) {
                    const int g = sidx >> 1;
                    const unsigned a01 = __builtin_amdgcn_perm(ub[4 * g + 1], ub[4 * g], 0x05010400u);
                    const unsigned a23 = __builtin_amdgcn_perm(ub[4 * g + 3], ub[4 * g + 2], 0x05010400u);
                    const unsigned lo = __builtin_amdgcn_perm(a23, a01, 0x05040100u);
                    plo[g] = BIAS128 ? (int)lo : (int)(lo ^ 0x80808080u);
                    if (HI) phi[g] = (int)(__builtin_amdgcn_perm(a23, a01, 0x07060302u) ^ 0x80808080u);
                }
            }
        };
        auto tile = [&](int jt, auto tail_tag, auto clamp_tag, auto hi_tag, auto st_tag, bool met = false) __attribute__((always_inline)) {
            constexpr bool tail = decltype(tail_tag)::value, CLAMP = decltype(clamp_tag)::value, HI = decltype(hi_tag)::value;
            constexpr int ST = decltype(st_tag)::value;
            v4i kf[DT], vf[DT];
            v16i acc;
            if (!met) ring.template step_sync<ST>(jt);
            ring.template read_k<ST>(jt, kf);
            ring.template read_t<ST>(jt, acc);
            ring.template read_v<ST>(jt, vf);
            __builtin_amdgcn_sched_barrier(0);                 // every fragment read of the tile is in flight before the score chain starts
            attn_qk<DT, KT>(kf, qf, acc);
            v4i plo, phi;
            chain(acc, jt, tail_tag, clamp_tag, hi_tag, plo, phi);
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                ol[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(plo, vf[t], ol[t], 0, 0, 0);
                if (HI) oh[HIK ? t : 0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(phi, vf[t], oh[HIK ? t : 0], 0, 0, 0);
            }
        };
        // The P.V MFMAs of tile j are issued inside tile j+1: the first right behind the score MFMAs (the wave waits for the scores
        // anyway), the others each behind a quarter of tile j+1's probability chain — a wave never queues an MFMA behind its own
        // four, and the matrix pipe works under the vector work of the same wave (sched_group_barrier pins the pattern).
        struct PVOps { v4i plo, phi; v4i vf[DT]; };
        PVOps ops[2];
        auto pv_all = [&](const PVOps& o, auto hi_tag) __attribute__((always_inline)) {
            constexpr bool HI = decltype(hi_tag)::value;
#pragma unroll
            for (int t = 0; t < DT; ++t) ol[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.plo, o.vf[t], ol[t], 0, 0, 0);
            if (HI) {
#pragma unroll
                for (int t = 0; t < DT; ++t) oh[HIK ? t : 0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.phi, o.vf[t], oh[HIK ? t : 0], 0, 0, 0);
            }
        };
        auto tileP = [&](int jt, auto clamp_tag, auto hi_tag, auto st_tag, PVOps& cur, const PVOps& prev, auto prev_tag) __attribute__((always_inline)) {
            constexpr bool HI = decltype(hi_tag)::value, HASP = decltype(prev_tag)::value;
            constexpr int ST = decltype(st_tag)::value;
            constexpr int NPV = HASP ? DT * (HI ? 2 : 1) : 0;
            v4i kf[DT];
            v16i acc;
            __builtin_amdgcn_sched_barrier(0);                 // the previous tile's chain stays on its side
            if (!HASP) ring.template step_sync<ST>(jt);        // (the first tile; every later one was met by its predecessor)
            ring.template read_k<ST>(jt, kf);
            ring.template read_t<ST>(jt, acc);
            ring.template read_v<ST>(jt, cur.vf);
            // the rendezvous for the NEXT tile right behind this tile's fragment reads (where hipcc hoists it to in the undeferred
            // body): a wave waits for the others' reads, not for their vector work
            ring.template step_sync<(ST + 1) & 3>(jt + 1);
            __builtin_amdgcn_sched_barrier(0);
            attn_qk<DT, KT>(kf, qf, acc);
            if (HASP) pv_all(prev, hi_tag);
            chain(acc, jt, std::false_type{}, clamp_tag, hi_tag, cur.plo, cur.phi);
            // the pattern of this region: [score MFMAs + the first deferred one] then a quarter of the chain per further MFMA
            __builtin_amdgcn_sched_group_barrier(0x008, DT + (NPV > 0 ? 1 : 0), 0);
#pragma unroll
            for (int i = 1; i < NPV; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x002, QD_ATTN_PV_DEFER_VALU, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x002, 96, 0);
            // the operand words are consumed one tile later: without a use HERE the optimiser sinks the whole chain down to that use,
            // i.e. behind the next rendezvous and its fragment reads, and the MFMAs are alone again
            asm volatile("" : "+v"(cur.plo));
            if (HI) asm volatile("" : "+v"(cur.phi));
        };
        auto run = [&](auto clamp_tag, auto hi_tag) __attribute__((always_inline)) {
            int jt = 0;
            bool met = false;
            if (FULL && nfull >= 5) {
                using std::integral_constant;
                tileP(0, clamp_tag, hi_tag, integral_constant<int, 0>{}, ops[0], ops[1], std::false_type{});
                for (jt = 1; jt + 4 <= nfull; jt += 4) {       // tiles 1 .. : stage and operand set are compile-time again
                    tileP(jt, clamp_tag, hi_tag, integral_constant<int, 1>{}, ops[1], ops[0], std::true_type{});
                    tileP(jt + 1, clamp_tag, hi_tag, integral_constant<int, 2>{}, ops[0], ops[1], std::true_type{});
                    tileP(jt + 2, clamp_tag, hi_tag, integral_constant<int, 3>{}, ops[1], ops[0], std::true_type{});
                    tileP(jt + 3, clamp_tag, hi_tag, integral_constant<int, 0>{}, ops[0], ops[1], std::true_type{});
                }
                pv_all(ops[0], hi_tag);                        // the last deferred tile; what is left of the key axis runs undeferred
                met = true;                                    // ... and tile jt has been met
            }
            for (; jt + 4 <= nfull && (jt & 3) == 0; jt += 4) {                 // four tiles = one turn of the ring: stage offsets are immediates
                tile(jt, std::false_type{}, clamp_tag, hi_tag, std::integral_constant<int, 0>{});
                tile(jt + 1, std::false_type{}, clamp_tag, hi_tag, std::integral_constant<int, 1>{});
                tile(jt + 2, std::false_type{}, clamp_tag, hi_tag, std::integral_constant<int, 2>{});
                tile(jt + 3, std::false_type{}, clamp_tag, hi_tag, std::integral_constant<int, 3>{});
            }
            for (; jt < nfull; ++jt, met = false) tile(jt, std::false_type{}, clamp_tag, hi_tag, std::integral_constant<int, -1>{}, met);
            if (nfull < ntile) tile(nfull, std::true_type{}, std::true_type{}, hi_tag, std::integral_constant<int, -1>{}, met);
        };
        if constexpr (FULL) {
            if (need_clamp) run(std::true_type{}, std::integral_constant<bool, P16>{});
            else if (hi_live) run(std::false_type{}, std::true_type{});
            else run(std::false_type{}, std::false_type{});
        } else {
            run(std::false_type{}, std::false_type{});
        }
        attn_wait_vmcnt<0>();                                  // the ring's run-ahead copies: nothing may land in LDS after the block retires
    }
    if (!live) return;

    // ---- epilogue (attn_lean_kernel's, with the 8-bit constants when the hi bytes were provably all zero) --------------
    int us[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int sl = 0, sh = 0;
#pragma unroll
        for (int t = 0; t < DT; ++t)
            if (t == t1) {
                sl = ol[t][r];
                if (HIK) sh = oh[HIK ? t : 0][r];
            }
        sl = __shfl(sl, frow1 + 32 * half);
        if (HIK) sh = __shfl(sh, frow1 + 32 * half);
        us[r] = sl + 128 * p.S + (hi_live ? 256 * (sh + 128 * p.S) : 0) + p.S * p.iwmin;
    }
    attn_write_rows<DT, HIK>(p, ol, oh, us, hi_live, bh, q0, frow, half, dw, zpw, oscale, zv);
}

// any row length (a multiple of 16 bytes): one thread per K row
__global__ __launch_bounds__(256) void attn_keyterm_rows_kernel(const int8_t* __restrict__ k, int32_t* __restrict__ kterm, const float* __restrict__ prm,
                                                                long nrows, int dpad) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    const int nzq = -(int)prm[1];
    int s = 0;
    for (int c = 0; c < dpad; c += 16) {
        const v4i w = *reinterpret_cast<const v4i*>(k + r * dpad + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) s = __builtin_amdgcn_sdot4(w[i], 0x01010101, s, false);
    }
    kterm[r] = QD_MAGICI + nzq * s;
}

// The LDS-staged path: statistics, then the two P.V launches (every block runs in exactly one of them).  kt: 0 = symmetric q
// (no per-key term), 2 = key-term table (AttnK::kterm).  ws: [BH][Tpad] AttnStat + one int per block.
template <int DT>
int launch_lds(const AttnK& k, bool p16, int kt, void* ws, hipStream_t st) {
    dim3 grid((unsigned)(k.gx * k.BH));
    AttnStat* stat = reinterpret_cast<AttnStat*>(ws);
    int* flag = reinterpret_cast<int*>(stat + (long)k.BH * k.Tpad);
    if (kt == 2) hipLaunchKernelGGL((attn_stats_kernel<DT, 2>), grid, dim3(256), 0, st, k, stat, flag, p16 ? 1 : 0);
    else hipLaunchKernelGGL((attn_stats_kernel<DT, 0>), grid, dim3(256), 0, st, k, stat, flag, p16 ? 1 : 0);
#define QD_PV_CASE(P, K)                                                                                          \
    if (p16 == P && kt == K) {                                                                                    \
        hipLaunchKernelGGL((attn_pv_kernel<DT, P, K, false>), grid, dim3(256), 0, st, k, (const AttnStat*)stat, (const int*)flag); \
        hipLaunchKernelGGL((attn_pv_kernel<DT, P, K, true>), grid, dim3(256), 0, st, k, (const AttnStat*)stat, (const int*)flag);  \
    }
    QD_PV_CASE(true, 0) QD_PV_CASE(true, 2) QD_PV_CASE(false, 0) QD_PV_CASE(false, 2)
#undef QD_PV_CASE
    return 0;
}

template <int DT>
int launch_lean(const AttnK& k, bool p16, int kt, hipStream_t st) {
    dim3 grid((unsigned)(k.gx * k.BH));
#define QD_LEAN_CASE(P, K) if (p16 == P && kt == K) hipLaunchKernelGGL((attn_lean_kernel<DT, P, K>), grid, dim3(256), 0, st, k);
    QD_LEAN_CASE(true, 0) QD_LEAN_CASE(true, 1) QD_LEAN_CASE(true, 2) QD_LEAN_CASE(false, 0) QD_LEAN_CASE(false, 1) QD_LEAN_CASE(false, 2)
#undef QD_LEAN_CASE
    return 0;
}

// one thread per 16-byte chunk of a K row: row sums by v_dot4 + a butterfly over the CPR lanes of a row
template <int CPR>
__global__ __launch_bounds__(256) void attn_keyterm_kernel(const int8_t* __restrict__ k, int32_t* __restrict__ kterm, const float* __restrict__ prm,
                                                           long nchunks) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    const int nzq = -(int)prm[1];
    int s = 0;
    if (c < nchunks) {
        const v4i w = *reinterpret_cast<const v4i*>(k + c * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) s = __builtin_amdgcn_sdot4(w[i], 0x01010101, s, false);
    }
#pragma unroll
    for (int o = 1; o < CPR; o <<= 1) s += __shfl_xor(s, o);
    if (c < nchunks && (c % CPR) == 0) kterm[c / CPR] = QD_MAGICI + nzq * s;
}

template <int DT>
int launch_dt(const AttnK& k, bool p16, bool asym, hipStream_t st) {
    dim3 grid((unsigned)(k.gx * k.BH));
    if (p16 && asym) hipLaunchKernelGGL((attn_kernel<DT, true, true>), grid, dim3(256), 0, st, k);
    else if (p16) hipLaunchKernelGGL((attn_kernel<DT, true, false>), grid, dim3(256), 0, st, k);
    else if (asym) hipLaunchKernelGGL((attn_kernel<DT, false, true>), grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL((attn_kernel<DT, false, false>), grid, dim3(256), 0, st, k);
    return 0;
}

}  // namespace

// Run-time knobs of the attention launcher: read from the environment ONCE (first call), changed afterwards only through
// qd_attn_config (tests and A/B runs flip the kernel choice inside one process).
struct AttnKnobs { int lean, pipe, xcd, ktab; };
static AttnKnobs& attn_knobs() {
    static AttnKnobs k = [] {
        auto env = [](const char* n, int dflt) { const char* v = getenv(n); return v ? atoi(v) : dflt; };
        return AttnKnobs{env("QD_ATTN_LEAN", 1), env("QD_ATTN_PIPE", 2), env("QD_ATTN_XCD", 1), env("QD_ATTN_KTAB", 1)};
    }();
    return k;
}
// the lean / LDS-staged kernels: a padding row of V^T for the code sums (d not a multiple of 32) and |scores| < 2^22
// (d * 255 * 128 < 2^22 <=> d <= 128).  Default: d < 64 (SD's 4096-token level).  d = 80 (SD's 1024-token level, three
// 32-byte K slabs per key) runs on the register-fed lean kernel only with QD_ATTN_LEAN=3: measured SLOWER than
// attn_kernel<3> (127 vs 113 us per 1024-key self-attention call, profiles/r04_attn_keyterm.md) — without LDS staging the
// three slabs per key come through the vector-memory path per wave, which costs more than the constant-operand MFMAs saved.
static bool attn_lean_shape(int d) {
    const int lean = attn_knobs().lean;
    return lean != 0 && (d & 31) != 0 && (d < 64 || (d < 96 && lean == 3));
}

extern "C" void qd_attn_config(int pipe_mode, int xcd, int ktab, int lean) {
    AttnKnobs& k = attn_knobs();
    if (pipe_mode >= 0) k.pipe = pipe_mode;
    if (xcd >= 0) k.xcd = xcd;
    if (ktab >= 0) k.ktab = ktab;
    if (lean >= 0) k.lean = lean;
}

// shapes of the LDS-staged two-kernel path: a lean shape (d < 64, not a multiple of 32) whose key axis is long enough for
// the ring start-up and the barriers to pay (pipe 2: S >= 512; pipe 3: any) and fits the constant row of ones
static bool attn_lds_shape(int T, int S, int d, int Spad, int dpad) {
    (void)T;
    const AttnKnobs& kn = attn_knobs();
    if (!attn_lean_shape(d) || dpad > 64 || Spad > QD_ONES_ROW) return false;
    return kn.pipe == 3 || (kn.pipe == 2 && S >= 512);
}

extern "C" int64_t qd_attn_ws_bytes(int BH, int T, int S, int d) {
    if (BH <= 0 || T <= 0 || S <= 0 || d <= 0) return 0;
    const int Tpad = (T + 31) / 32 * 32, Spad = (S + 31) / 32 * 32, dpad = (d + 31) / 32 * 32;
    if (!attn_lds_shape(T, S, d, Spad, dpad)) return 0;
    return (int64_t)BH * Tpad * (int64_t)sizeof(AttnStat) + (int64_t)BH * ((T + 127) / 128) * 4;
}

// the table pays where the LDS-staged kernel runs (thousands of keys); the register-fed kernel on short key axes (the 77
// context tokens: 3 tiles per sweep, 3 waves per SIMD) keeps the constant-operand MFMAs — it has no registers to spare
// for 16 seeds per tile and nothing to gain from them; pipe modes 0 / 3 (tests, A/B runs) take a table on every key axis
extern "C" int qd_attn_uses_keyterm(int d, int S, int q_asym) {
    const AttnKnobs& k = attn_knobs();
    return (q_asym != 0 && attn_lean_shape(d) && k.ktab != 0 && (k.pipe != 2 || S >= 512)) ? 1 : 0;
}

extern "C" int qd_attn_keyterm(const int8_t* k, int BH, int Spad, int dpad, const float* prm, int32_t* kterm, void* stream) {
    QD_REQUIRE(k && prm && kterm, "qd_attn_keyterm: null pointer");
    QD_REQUIRE(BH > 0 && Spad > 0 && Spad % 32 == 0 && (dpad == 32 || dpad == 64 || dpad == 96), "qd_attn_keyterm: Spad must be a multiple of 32, dpad 32, 64 or 96 (got %d, %d)", Spad, dpad);
    QD_REQUIRE(qd_aligned(k, 16) && qd_aligned(kterm, 16), "qd_attn_keyterm: operands must be 16-byte aligned");
    const long nchunks = (long)BH * Spad * (dpad / 16);
    QD_REQUIRE((nchunks + 255) / 256 < (1L << 31), "qd_attn_keyterm: too many blocks");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((nchunks + 255) / 256));
    if (dpad == 32) hipLaunchKernelGGL((attn_keyterm_kernel<2>), grid, dim3(256), 0, st, k, kterm, prm, nchunks);
    else if (dpad == 64) hipLaunchKernelGGL((attn_keyterm_kernel<4>), grid, dim3(256), 0, st, k, kterm, prm, nchunks);
    else hipLaunchKernelGGL(attn_keyterm_rows_kernel, dim3((unsigned)(((long)BH * Spad + 255) / 256)), dim3(256), 0, st, k, kterm, prm, (long)BH * Spad, dpad);
    QD_LAUNCH_CHECK("qd_attn_keyterm");
    return 0;
}

extern "C" int qd_attn_i8(const int8_t* q, const int8_t* k, const int8_t* vt, const int32_t* qsum, const int32_t* kterm,
                          const int32_t* vsum, int BH, int H, int T, int S, int d, int Tpad, int Spad, int dpad,
                          const float* prm, int wbits, int wmin, int wmax, int q_asym, float* out, int64_t ldo,
                          int8_t* out8, int64_t ldo8, const float* oq_params, int oq_min, int oq_max, int oq_off,
                          void* ws, int64_t ws_bytes, void* stream) {
    QD_REQUIRE(q && k && vt && vsum && prm && (out || out8), "qd_attn_i8: null pointer");
    QD_REQUIRE(!out8 || (oq_params && ldo8 >= (int64_t)H * d && oq_max - oq_off <= 127 && oq_min - oq_off >= -128),
               "qd_attn_i8: quantised output needs oq_params, ldo8 >= H*d and a grid that fits int8");
    QD_REQUIRE(BH > 0 && H > 0 && BH % H == 0 && T > 0 && S > 0 && d > 0, "qd_attn_i8: bad shape");
    QD_REQUIRE(Tpad % 32 == 0 && Spad % 32 == 0 && dpad % 32 == 0 && Tpad >= T && Spad >= S && dpad >= d, "qd_attn_i8: padded dims must be multiples of 32");
    QD_REQUIRE((long)BH * ((T + 127) / 128) < (1L << 31), "qd_attn_i8: too many blocks");
    QD_REQUIRE(wbits == 8 || wbits == 16, "qd_attn_i8: probability bits must be 8 or 16 (got %d)", wbits);
    QD_REQUIRE(wmax - wmin <= (wbits == 16 ? 65535 : 255), "qd_attn_i8: probability grid [%d,%d] wider than %d bits", wmin, wmax, wbits);
    QD_REQUIRE(qd_aligned(q, 16) && qd_aligned(k, 16) && qd_aligned(vt, 16) && qd_aligned(kterm, 16), "qd_attn_i8: operands must be 16-byte aligned");
    (void)qsum;                                  // per-query constants cancel in the softmax: never needed
    const bool asym = q_asym != 0;
    const AttnKnobs& kn = attn_knobs();
    AttnK a{q, k, vt, qsum, kterm, vsum, prm, out, (long)ldo, BH, H, T, S, d, Tpad, Spad, dpad, (float)wmin, (float)wmax, wmin,
            out8, (long)ldo8, oq_params, (float)oq_min, (float)oq_max, oq_off, kn.xcd != 0 ? 1 : 0, (T + 127) / 128};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool p16 = wbits == 16;
    // lean variant: needs a padding row of V^T (d not a multiple of 32) and |scores| < 2^22 (d < 64)
    // pipe: 2 = LDS-staged kernel wherever it pays (default), 0 = attn_lean_kernel everywhere (A/B runs and the equality
    // test), 3 = LDS-staged kernel on every eligible shape
    if (attn_lean_shape(d)) {
        const int kt = !asym ? 0 : (kterm && qd_attn_uses_keyterm(d, S, q_asym)) ? 2 : 1;     // per-key zero-point term: none / constant-operand MFMAs / table
        // the LDS-staged two-kernel path (qd_attn_ws_bytes says which shapes): without a table for an asymmetric q (kt == 1: the
        // constant-operand MFMAs, A/B runs with ktab = 0) the register-fed kernel runs instead
        if (kt != 1 && attn_lds_shape(T, S, d, Spad, dpad)) {
            const int64_t need = qd_attn_ws_bytes(BH, T, S, d);
            QD_REQUIRE(ws && ws_bytes >= need && qd_aligned(ws, 16), "qd_attn_i8: this shape needs %ld bytes of 16-byte aligned workspace (qd_attn_ws_bytes), got %ld",
                       (long)need, (long)ws_bytes);
            if (dpad == 32) launch_lds<1>(a, p16, kt, ws, st);
            else launch_lds<2>(a, p16, kt, ws, st);
        } else if (dpad == 32) launch_lean<1>(a, p16, kt, st);
        else if (dpad == 64) launch_lean<2>(a, p16, kt, st);
        else launch_lean<3>(a, p16, kt, st);
        QD_LAUNCH_CHECK("qd_attn_i8");
        return 0;
    }
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
