// igemm_dma.hip — second-generation K3/K4 kernel for int4 weights: LDS-DMA fed, 3-stage ring.
//
// Same contraction and epilogue as igemm_i8.hip (reference qdiff/quant_layer.py:256-276), different
// data path.  What limited the first kernel was latency, not the matrix pipe: one K-step of
// register-staged prefetch and a barrier per step.  Here
//   * both operands are copied global -> LDS by the DMA path (global_load_lds, 16 B/lane, no VGPR
//     round trip): activations per (tap, 64-channel step) with the im2col gather expressed in the
//     per-lane SOURCE address — out-of-image taps read a 16-byte buffer of the "true zero" byte z',
//     K-tail / M-tail lanes read zeros — and the LDS destination is lane-linear, so the XOR bank
//     swizzle of the A tile is applied to the source chunk index instead;
//   * weights are pre-tiled at pack time (qd_pack_weights_t4): one K-step x 32 output channels is a
//     contiguous 1-KB block already in fragment order, so the B copy is a straight memcpy and the
//     fragment read is a conflict-free ds_read_b64; the RAW nibbles are unpacked at fragment-read time
//     (two AND/shift per 8 weights — the weight zero point is restored in the epilogue through the
//     activation row sums), which keeps 4-bit weights 4-bit all the way into LDS;
//   * a 3-deep LDS ring keeps two K-steps in flight across the single barrier per step
//     (counted s_waitcnt vmcnt(N), raw s_barrier: cdna_hip_programming.md §5 "Pipelining across
//     barriers").
// Block = 4 waves stacked along M; wave tile = (32*MT) x (32*NT) of 32x32x32 MFMAs, BN = 32*NT
// (NT=5 -> 160 divides every SD-v1 width 320/640/1280/..., NT=7 -> 224 for LDM-4).
#include "common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void* qd_lds_ptr;
typedef const __attribute__((address_space(1))) void* qd_gbl_ptr;

namespace {

__device__ __attribute__((aligned(16))) const int qd_zero16[4] = {0, 0, 0, 0};

struct SegD {
    int c0, clen, kstep0, nsteps_tap;
    const float*  scale;
    const int*    zc;
    const int*    zw;       // [Cout] weight zero point (raw nibbles are the stored operand)
    const int*    zfill;
    const int8_t* fill16;
};

struct ConvD {
    const int8_t*  x;
    const uint8_t* wt;
    void*          out;
    int32_t*       iout;
    const float*   bias;
    const float*   rowbias;
    const void*    residual;
    long ldx, ldo, ldr, ldrb;
    int B, H, W, Ho, Wo, Cout, kh, kw, stride, pad_t, pad_l;
    int M, taps, nseg, nblk_m, nblk_n, ntiles;
    SegD seg[2];
    const float* oq;          // O_GEGLU: {delta, zero_point} of the output quantiser
    float oqmin, oqmax;
    int   oqoff;
    int   it_per;             // O_PART: K-steps per split (blockIdx.y = split index)
    int   hdH, hdd, hdT, hdTpad, hddpad;   // O_HROWS / O_HTR: heads, head dim, tokens per sample, padded dims
    float oqpre;              // O_HROWS / O_HTR: multiplier applied before the output quantiser
    int32_t* hdsum;           // O_HTR: [(b*H+h)][dpad] column sums of the stored bytes (atomically accumulated)
    float* gnpart;            // O_F32, optional: per-(sample, 128-row chunk, channel) {sum, sum of squares} of the output,
    int    gn_nchunk;         //   i.e. the first level of GroupNorm's statistics (layout of gn_partial_kernel); S/128
};

// O_PART: split-K partial.  The block contracts K-steps [y*it_per, (y+1)*it_per) only and stores
// acc - zw[n]*Asum_part[m] (int32, exact) into slice y of the workspace; splitk_finalize_kernel sums
// the slices, restores the K-independent constants and applies the float epilogue.
// O_HROWS / O_HTR: the output feeds the attention kernel: it is quantised with the attention block's
// q/k/v activation quantiser and written as int8 straight into the head-major operand layout of
// attn_i8.hip (rows [(b,h)][t][dpad] for q and k; transposed + key-permuted [(b,h)][dd][t] plus column
// sums for v) — the fp32 projection output and the separate qd_quantize_heads pass disappear.
enum { O_F32 = 0, O_F16 = 1, O_I32 = 2, O_GEGLU = 3, O_PART = 4, O_HROWS = 5, O_HTR = 6 };

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// 16-byte-per-lane global -> LDS DMA, issued through inline asm ON PURPOSE: with the builtin, hipcc
// cannot prove that the pending LDS writes do not alias the next ds_read and inserts
// `s_waitcnt vmcnt(0)` in front of every K-step's first LDS read, which drains the whole ring
// (seen in the ISA of the first version of this kernel).  An asm DMA is invisible to the compiler's
// waitcnt bookkeeping; completion is tracked by hand (wait_vmcnt<N> + s_barrier below).
// LDS destination = lds_base (wave-uniform, bytes) + lane*16; M0 is written in the same statement
// that reads it and restored (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_base) {
    // M0 is not live across statements in this kernel (no movrel / GWS / sendmsg / builtin LDS-DMA), so it is
    // written and consumed inside the one statement and not restored.
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_base))
        : "memory");
}

__device__ __forceinline__ int bytesum16(const v4i& v) {
    int s = __builtin_amdgcn_sdot4(v.x, 0x01010101, 0, false);
    s = __builtin_amdgcn_sdot4(v.y, 0x01010101, s, false);
    s = __builtin_amdgcn_sdot4(v.z, 0x01010101, s, false);
    return __builtin_amdgcn_sdot4(v.w, 0x01010101, s, false);
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(qd_lds_ptr)(p);
}

// WB = weight bits of the tile-ordered operand: 4 (raw nibbles, 1 KB per K-step x 32 channels, qd_pack_weights_t4) or
// 8 (s8 bytes W-128, 2 KB, qd_pack_weights_t8); everything but the B tile size and the fragment read is shared.
template <int MT, int NT, bool SPLIT, int OUT, int WB>
__global__ __launch_bounds__(256, SPLIT ? 1 : 2) void igemm_dma_kernel(const ConvD p) {
    static_assert(WB == 4 || WB == 8, "weight bits");
    constexpr int BM = 128 * MT, BN = 32 * NT;
    constexpr int TB = 256 * WB;                  // bytes of one (K-step, 32-channel) weight tile
    constexpr int A_BYTES = BM * 64, B_BYTES = NT * TB;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NA = 2 * MT;                    // A DMA instructions per wave per stage (16 rows each)
    constexpr int NB = (NT * TB / 64 + 63) / 64;  // B DMA instructions per wave per stage (NT*TB/4 bytes per wave)
    constexpr int PER = NA + NB;

    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * STAGE + 2 * BM * 4 + 4 * BN * 4];
    int* sRowB = reinterpret_cast<int*>(smem + 3 * STAGE);
    int* sAsum = sRowB + BM;
    // per-output-channel epilogue constants of the LAST segment, fetched at kernel start so that their
    // global-load latency overlaps the prologue DMA instead of serialising in front of the stores
    float* sScale = reinterpret_cast<float*>(sAsum + BM);
    int*   sZc    = reinterpret_cast<int*>(sScale + BN);
    int*   sZw    = sZc + BN;
    float* sBias  = reinterpret_cast<float*>(sZw + BN);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;

    const int nblk    = p.nblk_m * p.nblk_n;
    const int logical = qd_xcd_remap(blockIdx.x, nblk);
    const int mb = logical / p.nblk_n, nb = logical % p.nblk_n;
    const int m0 = mb * BM, n0 = nb * BN;

    // ---- loader rows: DMA instruction q covers tile rows q*16 .. q*16+15, lane -> (row, slot) ----
    const int lr16 = lane >> 2, slot = lane & 3;
    const int8_t* a_img[NA];                      // image origin (b, 0, 0, 0) of the row this lane feeds
    int  a_ih0[NA], a_iw0[NA], a_chunk[NA];
    bool a_valid[NA];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = (wave + 4 * i) * 16 + lr16;
        const int m = m0 + r;
        a_valid[i] = m < p.M;
        const int mm = a_valid[i] ? m : 0;
        const int b  = mm / HoWo;
        const int rem = mm - b * HoWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        a_img[i]   = p.x + (long)b * p.H * p.W * p.ldx;
        a_ih0[i]   = ho * p.stride - p.pad_t;
        a_iw0[i]   = wo * p.stride - p.pad_l;
        a_chunk[i] = (slot ^ ((r >> 2) & 3)) * 16;   // source byte offset (within a 64-B K-step) landing in this lane's LDS slot
        if (slot == 0) sRowB[r] = b;
    }

    // ---- loader state (uniform) + per-lane running source pointers ------------------------------
    int it_begin = 0;                             // first K-step of this block (split-K partials only)
    if constexpr (OUT == O_PART) it_begin = blockIdx.y * p.it_per;
    int ls = 0;
    int ltap = it_begin / p.seg[0].nsteps_tap, lc = it_begin - ltap * p.seg[0].nsteps_tap;
    int lrr = ltap / p.kw, lq = ltap - lrr * p.kw;
    const int8_t* a_cur[NA];                      // source of the NEXT K-step for DMA instruction i
    int           a_inc[NA];                      // 64 for real pixels, 0 for fill / zero sources
    const uint8_t* b_cur = p.wt + ((long)(p.seg[0].kstep0 + it_begin) * p.ntiles + (long)nb * NT) * TB + wave * (NT * TB / 4) + lane * 16;
    const long b_inc = (long)p.ntiles * TB;
    const int8_t* zero16 = reinterpret_cast<const int8_t*>(qd_zero16);
    bool b_ok[NB];                                // the last N-block may cover n-tiles that do not exist
#pragma unroll
    for (int r = 0; r < NB; ++r) b_ok[r] = nb * NT + (wave * (NT * TB / 4) + r * 1024 + lane * 16) / TB < p.ntiles;

    auto set_tap = [&]() __attribute__((always_inline)) {
        const SegD& sg = p.seg[ls];
        const int8_t* fill = sg.fill16 ? sg.fill16 : zero16;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int ih = a_ih0[i] + lrr, iw = a_iw0[i] + lq;
            const bool inb = a_valid[i] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            a_cur[i] = inb ? a_img[i] + ((long)ih * p.W + iw) * p.ldx + sg.c0 + a_chunk[i] : (a_valid[i] ? fill : zero16);
            a_inc[i] = inb ? 64 : 0;
        }
    };
    set_tap();
    if constexpr (OUT == O_PART) {
#pragma unroll
        for (int i = 0; i < NA; ++i) a_cur[i] += lc * a_inc[i];
    }

    // issue DMA instruction d of the current loader step into ring stage ST (compile-time LDS addresses)
    auto issue_one = [&](unsigned stage_base, int d) __attribute__((always_inline)) {
        if (d < NA) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                if (i == d) {
                    const int8_t* src = a_cur[i];
                    // K tail: the last 64-wide step of a tap may run past the segment's channels
                    if (lc * 64 + a_chunk[i] >= p.seg[ls].clen) src = zero16;
                    glds16(src, stage_base + (wave + 4 * i) * 1024);
                    a_cur[i] += a_inc[i];
                }
        } else {
            const int r = d - NA;
#pragma unroll
            for (int rr = 0; rr < NB; ++rr)
                if (rr == r && rr * 64 + lane < NT * TB / 64)
                    glds16(b_ok[rr] ? (const void*)(b_cur + rr * 1024) : (const void*)zero16,
                           stage_base + A_BYTES + wave * (NT * TB / 4) + rr * 1024);
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {
        b_cur += b_inc;
        ++lc;
        if (lc == p.seg[ls].nsteps_tap) {
            lc = 0; ++ltap; ++lq;
            if (lq == p.kw) { lq = 0; ++lrr; }
            if (ltap == p.taps) { ltap = 0; lrr = 0; lq = 0; ++ls; }
            if (ls < p.nseg) set_tap();
        }
    };

    // ---- accumulators ---------------------------------------------------------------------------
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
    int asum[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) asum[i] = 0;

    auto publish_asum = [&]() __attribute__((always_inline)) {                 // row sums of this wave's rows -> LDS (both k-halves combined)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int v = asum[i] + __shfl_xor(asum[i], 32);
            if (fhalf == 0) sAsum[wave * (32 * MT) + i * 32 + frow] = v;
            asum[i] = 0;
        }
    };

    const int nst0  = p.taps * p.seg[0].nsteps_tap;
    const int total_all = nst0 + (p.nseg == 2 ? p.taps * p.seg[1].nsteps_tap : 0);
    const int total = OUT == O_PART ? min(p.it_per, total_all - it_begin) : total_all;
    const unsigned lds0 = lds_addr(smem);

    // lane-invariant fragment offsets inside a stage
    int a_off[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = wave * (32 * MT) + i * 32 + frow;
            a_off[i][ks] = row * 64 + (((ks * 2 + fhalf) ^ ((row >> 2) & 3)) * 16);
        }
    const int b_off = (fhalf * 32 + frow) * (WB * 2);   // + ks*(TB/2) + j*TB

    float pr_scale = 0.f, pr_bias = 0.f;
    int   pr_zc = 0, pr_zw = 0;
    {
        const SegD& sgl = p.seg[p.nseg - 1];
        const int pn = n0 + (int)threadIdx.x;
        if ((int)threadIdx.x < BN && pn < p.Cout) {
            pr_scale = sgl.scale[pn];
            if (sgl.zc) pr_zc = sgl.zc[pn];
            if (sgl.zw) pr_zw = sgl.zw[pn];
            if (p.bias) pr_bias = p.bias[pn];
        }
    }

    // ---- prologue: two stages in flight ----------------------------------------------------------
#pragma unroll
    for (int d = 0; d < PER; ++d) issue_one(lds0, d);
    advance();
    if (total > 1) {
#pragma unroll
        for (int d = 0; d < PER; ++d) issue_one(lds0 + STAGE, d);
        advance();
    }
    if ((int)threadIdx.x < BN) {                       // visible to every wave after the main loop's barriers
        sScale[threadIdx.x] = pr_scale;
        sZc[threadIdx.x]    = pr_zc;
        sZw[threadIdx.x]    = pr_zw;
        sBias[threadIdx.x]  = pr_bias;
    }

    auto flush_segment0 = [&]() __attribute__((always_inline)) {
        publish_asum();
        __syncthreads();
        const SegD& sg = p.seg[0];
        const int kz = sg.zfill ? sg.zfill[1] : 0;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + j * 32 + frow;
            const bool nok = n < p.Cout;
            const float sc = nok ? sg.scale[n] : 0.f;
            const int zc_n = (nok && sg.zc) ? sg.zc[n] : 0;
            const int zw_n = (nok && sg.zw) ? sg.zw[n] : 0;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rowl = wave * (32 * MT) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                    const int I = acc[i][j][r] - zc_n - __mul24(zw_n, sAsum[rowl] - kz);
                    facc[SPLIT ? i : 0][SPLIT ? j : 0][r] = (float)I * sc;
                    acc[i][j][r] = 0;
                }
        }
        __syncthreads();                               // sAsum is reused by the second segment
    };

    // one K-step on ring stage ST (compile-time), prefetching step it+2 into stage (ST+2)%3
    auto step = [&](auto st_tag, int it) __attribute__((always_inline)) {
        constexpr int ST = decltype(st_tag)::value;
        constexpr int PST = (ST + 2) % 3;
        if (it + 1 < total) wait_vmcnt<PER>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                  // stage ST landed for every wave; stage ST-1 fully consumed
        const bool prefetch = it + 2 < total;
        const unsigned char* cA = smem + ST * STAGE;
        const unsigned char* cB = cA + A_BYTES;
        int dslot = 0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v4i af[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                af[i] = *reinterpret_cast<const v4i*>(cA + a_off[i][ks]);
                asum[i] += bytesum16(af[i]);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                v4i bf;
                if constexpr (WB == 4) {
                    const uint2 pk = *reinterpret_cast<const uint2*>(cB + b_off + ks * (TB / 2) + j * TB);
                    bf = v4i{(int)(pk.x & 0x0F0F0F0Fu), (int)((pk.x >> 4) & 0x0F0F0F0Fu),
                             (int)(pk.y & 0x0F0F0F0Fu), (int)((pk.y >> 4) & 0x0F0F0F0Fu)};
                } else {
                    bf = *reinterpret_cast<const v4i*>(cB + b_off + ks * (TB / 2) + j * TB);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf, acc[i][j], 0, 0, 0);
                if (dslot < PER) { if (prefetch) issue_one(lds0 + PST * STAGE, dslot); }
                ++dslot;
            }
        }
        if (prefetch) {
#pragma unroll
            for (int d = 2 * NT; d < PER; ++d) issue_one(lds0 + PST * STAGE, d);
            advance();
        }
        if (SPLIT && p.nseg == 2 && it == nst0 - 1) flush_segment0();
    };

    for (int it = 0; it < total; it += 3) {
        step(std::integral_constant<int, 0>{}, it);
        if (it + 1 < total) step(std::integral_constant<int, 1>{}, it + 1);
        if (it + 2 < total) step(std::integral_constant<int, 2>{}, it + 2);
    }

    // ---- epilogue --------------------------------------------------------------------------------
    publish_asum();
    __syncthreads();                                   // sAsum / sRowB visible to every wave
    const SegD& sg = p.seg[p.nseg - 1];
    const int kz = sg.zfill ? sg.zfill[1] : 0;
    // Addressing: every global access of the epilogue is (uniform 64-bit base of this block) + (32-bit lane
    // offset), and the 16 rows a lane owns differ by compile-time multiples of the row stride, so an element
    // costs one v_add_u32 instead of a 64-bit multiply-add.  zw * Asum uses the 24-bit multiplier (both
    // factors fit: |zw| <= 128, |Asum - kz| <= 2 * 128 * K < 2^23, checked on the host).
    if constexpr (OUT == O_GEGLU) {
        // weight rows were packed (value tile, gate tile) interleaved: tiles 2jp / 2jp+1 of this lane hold
        // the value and the gate of output feature nb*(BN/2) + jp*32 + frow.  y = value * gelu(gate)
        // (erf GELU, ldm/modules/attention.py:42-44), then the next Linear's act quantiser, 1 byte out.
        static_assert(NT % 2 == 0, "GEGLU epilogue pairs n-tiles");
        const float od = p.oq[0], oz = p.oq[1];
        int8_t* o8 = reinterpret_cast<int8_t*>(p.out) + (long)m0 * p.ldo + nb * (BN / 2);
        const unsigned ldo = (unsigned)p.ldo;
        const int Fout = p.Cout >> 1;
#pragma unroll
        for (int jp = 0; jp < NT / 2; ++jp) {
            const int lv = (2 * jp) * 32 + frow, lg = lv + 32;            // tile-local channel of value / gate
            const int cl = jp * 32 + frow;
            const bool ok = nb * (BN / 2) + cl < Fout && n0 + lg < p.Cout;
            const float sv = sScale[lv], sgt = sScale[lg];
            const int zcv = sZc[lv], zcg = sZc[lg];
            const int zwv = sZw[lv], zwg = sZw[lg];
            const float bv = sBias[lv], bg = sBias[lg];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int rbase = wave * (32 * MT) + i * 32 + 4 * fhalf;
                const unsigned o0 = (unsigned)rbase * ldo + cl;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cr = (r & 3) + 8 * (r >> 2);
                    const int rowl = rbase + cr;
                    const int as = sAsum[rowl] - kz;
                    const float val = (float)(acc[i][2 * jp][r] - zcv - __mul24(zwv, as)) * sv + bv;
                    const float gate = (float)(acc[i][2 * jp + 1][r] - zcg - __mul24(zwg, as)) * sgt + bg;
                    const float y = val * (0.5f * gate * (1.0f + qd_erff(gate * 0.70710678118654752440f)));
                    const int8_t code = (int8_t)(qd_code(y, od, oz, p.oqmin, p.oqmax) - p.oqoff);
                    if (ok && m0 + rowl < p.M) o8[o0 + (unsigned)cr * ldo] = code;      // only the store is predicated
                }
            }
        }
        return;
    }
    if constexpr (OUT == O_HROWS || OUT == O_HTR) {
        // rows m = b*T + t, columns n = h*d + dd.  The host guarantees T % BM == 0 and M % T == 0: a block
        // lies inside one sample and has no ragged rows.  Pad bytes (dd >= d) are never written: the operand
        // buffers are zero-initialised once and reused.
        const float od = p.oq[0], oz = p.oq[1];
        int8_t* o8 = reinterpret_cast<int8_t*>(p.out);
        const int bidx = m0 / p.hdT, t0 = m0 - bidx * p.hdT;
        int* sPart = reinterpret_cast<int*>(smem);            // [4][BN] column-sum partials (the ring is dead by now)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int cl = j * 32 + frow;
            const bool nok = n0 + cl < p.Cout;
            const int nn = nok ? n0 + cl : 0;
            const int h = nn / p.hdd, dd = nn - h * p.hdd;
            const float sc = sScale[cl];
            const int zc_n = sZc[cl];
            const int zw_n = sZw[cl];
            const float bias_n = sBias[cl];
            if constexpr (OUT == O_HROWS) {
                int8_t* ob = o8 + (((long)bidx * p.hdH + h) * p.hdTpad + t0) * p.hddpad + dd;
                // optional fp32 residual (H = 1, d = Cout turns this epilogue into "Linear + residual -> the next
                // Linear's int8 rows": the FF output of a transformer block feeding SpatialTransformer.proj_out)
                const bool hres = p.residual != nullptr;
                const float* rfh = reinterpret_cast<const float*>(p.residual) + (long)m0 * p.ldr + n0;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int rbase = wave * (32 * MT) + i * 32 + 4 * fhalf;
                    float rs[16];
                    if (hres) {
                        const unsigned r0 = (unsigned)rbase * (unsigned)p.ldr + (nok ? cl : 0);
#pragma unroll
                        for (int r = 0; r < 16; ++r) rs[r] = rfh[r0 + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)p.ldr];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rowl = rbase + (r & 3) + 8 * (r >> 2);
                        const int I = acc[i][j][r] - zc_n - __mul24(zw_n, sAsum[rowl] - kz);
                        float v = (float)I * sc + bias_n;
                        if (hres) v += rs[r];
                        const int8_t code = (int8_t)(qd_code(v * p.oqpre, od, oz, p.oqmin, p.oqmax) - p.oqoff);
                        if (nok) ob[(unsigned)rowl * (unsigned)p.hddpad] = code;
                    }
                }
            } else {
                int8_t* ob = o8 + (((long)bidx * p.hdH + h) * p.hddpad + dd) * p.hdTpad + t0 + fhalf * 16;
                int csum = 0;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int tile0 = wave * (32 * MT) + i * 32;   // first row of this 32-key tile inside the block
                    v4i pk;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        unsigned w = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 4 * g + e;
                            const int rowl = tile0 + 4 * fhalf + (r & 3) + 8 * (r >> 2);
                            const int I = acc[i][j][r] - zc_n - __mul24(zw_n, sAsum[rowl] - kz);
                            const float v = (float)I * sc + bias_n;
                            const int code = qd_code(v * p.oqpre, od, oz, p.oqmin, p.oqmax) - p.oqoff;
                            csum += code;
                            w |= (unsigned)(code & 0xff) << (8 * e);
                        }
                        pk[g] = (int)w;
                    }
                    // key slot p = half*16 + r  <->  key (r&3) + 8*(r>>2) + 4*half of the tile (attn_i8.hip): the
                    // MFMA C layout IS the permuted order, so a lane's 16 codes are 16 contiguous bytes
                    if (nok) *reinterpret_cast<v4i*>(ob + tile0) = pk;
                }
                csum += __shfl_xor(csum, 32);
                if (fhalf == 0) sPart[wave * BN + cl] = nok ? csum : 0;
            }
        }
        if constexpr (OUT == O_HTR) {
            __syncthreads();
            const int c = threadIdx.x;
            if (c < BN && n0 + c < p.Cout) {
                const int tot = sPart[c] + sPart[BN + c] + sPart[2 * BN + c] + sPart[3 * BN + c];
                const int n = n0 + c, h = n / p.hdd, dd = n - h * p.hdd;
                atomicAdd(&p.hdsum[((long)bidx * p.hdH + h) * p.hddpad + dd], tot);
            }
        }
        return;
    }
    // Branch-free per 32x32 tile: out-of-range rows/columns are handled by CLAMPING the offsets of the
    // residual / row-bias loads (and predicating only the stores), so the 16 loads of a tile are issued
    // back to back.  (With a per-element `if (...) continue;` every load sat in its own basic block and the
    // epilogue paid one memory latency per element: layers with a residual ran 2-3x slower.)
    const bool has_rb = p.rowbias != nullptr, has_res = p.residual != nullptr;
    const unsigned ldo = (unsigned)p.ldo, ldr = (unsigned)p.ldr, ldc = (unsigned)p.Cout;
    float*  const of = reinterpret_cast<float*>(p.out) + (long)m0 * p.ldo + n0;
    __half* const oh = reinterpret_cast<__half*>(p.out) + (long)m0 * p.ldo + n0;
    const float*  const rf = reinterpret_cast<const float*>(p.residual) + (long)m0 * p.ldr + n0;
    const __half* const rh = reinterpret_cast<const __half*>(p.residual) + (long)m0 * p.ldr + n0;
    int32_t* const oi = p.iout + ((OUT == O_PART ? (long)blockIdx.y * p.M : 0L) + m0) * p.Cout + n0;
    // optional GroupNorm statistics of the tensor being written (consumed by qd_groupnorm_silu_quant instead of
    // its own pass over HBM): per-column partials in registers -> cross-half shuffle -> fixed-order LDS reduction
    // over the waves of each 128-row chunk -> one {sum, sumsq} pair per (chunk, channel).  Deterministic.
    const bool gn = OUT == O_F32 && p.gnpart != nullptr;
    float* sGn = reinterpret_cast<float*>(smem);          // [4 waves][BN][2]  (the ring is dead by now)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int cl = j * 32 + frow;
        const bool nok = n0 + cl < p.Cout;
        const int clc = nok ? cl : 0;
        float gs = 0.f, gq = 0.f;
        const float sc = sScale[cl];
        const int zc_n = sZc[cl];
        const int zw_n = sZw[cl];
        const float bias_n = sBias[cl];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rbase = wave * (32 * MT) + i * 32 + 4 * fhalf;        // rowl = rbase + (r&3) + 8*(r>>2)
            if constexpr (OUT == O_PART || OUT == O_I32) {
                const unsigned o0 = (unsigned)rbase * ldc + cl;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cr = (r & 3) + 8 * (r >> 2);
                    const int rowl = rbase + cr;
                    const int v = OUT == O_PART ? acc[i][j][r] - __mul24(zw_n, sAsum[rowl])
                                                : acc[i][j][r] - zc_n - __mul24(zw_n, sAsum[rowl] - kz);
                    if (nok && m0 + rowl < p.M) oi[o0 + (unsigned)cr * ldc] = v;
                }
            } else {
                float rb[16], rs[16];
                if (has_rb) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) rb[r] = p.rowbias[(long)sRowB[rbase + (r & 3) + 8 * (r >> 2)] * p.ldrb + n0 + clc];
                }
                if (has_res) {
                    const unsigned r0 = (unsigned)rbase * ldr + clc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cr = (r & 3) + 8 * (r >> 2);
                        const unsigned off = (m0 + rbase + cr < p.M) ? r0 + (unsigned)cr * ldr : (unsigned)clc;   // row m0 always exists
                        if (OUT == O_F32) rs[r] = rf[off];
                        else rs[r] = __half2float(rh[off]);
                    }
                }
                const unsigned o0 = (unsigned)rbase * ldo + cl;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cr = (r & 3) + 8 * (r >> 2);
                    const int rowl = rbase + cr;
                    const int I = acc[i][j][r] - zc_n - __mul24(zw_n, sAsum[rowl] - kz);
                    float v = (float)I * sc;
                    if (SPLIT) v += facc[SPLIT ? i : 0][SPLIT ? j : 0][r];
                    v += bias_n;
                    if (has_rb) v += rb[r];
                    if (has_res) v += rs[r];
                    if (nok && m0 + rowl < p.M) {
                        if (OUT == O_F32) of[o0 + (unsigned)cr * ldo] = v;
                        else oh[o0 + (unsigned)cr * ldo] = __float2half(v);
                        if (gn) { gs += v; gq += v * v; }
                    }
                }
            }
        }
        if (gn) {
            gs += __shfl_xor(gs, 32);
            gq += __shfl_xor(gq, 32);
            if (fhalf == 0) { sGn[(wave * BN + cl) * 2] = gs; sGn[(wave * BN + cl) * 2 + 1] = gq; }
        }
    }
    if (gn) {
        __syncthreads();
        const int c = threadIdx.x;
        if (c < BN && n0 + c < p.Cout) {
            constexpr int WPC = 4 / MT;                       // waves per 128-row chunk
#pragma unroll
            for (int ch = 0; ch < MT; ++ch) {
                const int mrow = m0 + ch * 128;
                if (mrow >= p.M) break;
                float ts = 0.f, tq = 0.f;
#pragma unroll
                for (int w = 0; w < WPC; ++w) { ts += sGn[((ch * WPC + w) * BN + c) * 2]; tq += sGn[((ch * WPC + w) * BN + c) * 2 + 1]; }
                const int b = mrow / HoWo, chunk = (mrow - b * HoWo) >> 7;
                float* dst = p.gnpart + (((long)b * p.gn_nchunk + chunk) * p.Cout + n0 + c) * 2;
                dst[0] = ts;
                dst[1] = tq;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// tile-ordered nibble packer: thread = one 8-byte unit (row n, 16 consecutive K)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_t4_kernel(const float* __restrict__ w, const float* __restrict__ alpha,
                                                      const float* __restrict__ delta, const float* __restrict__ zp,
                                                      int Cout, int Cin_total, int taps, int c0, int clen, int clen_pad,
                                                      int n_levels, uint8_t* __restrict__ wt, int kstep0, int ntiles,
                                                      int nsteps_tap, int32_t* __restrict__ wsum) {
    // unit index: (((tap*nsteps_tap + cs) * ntiles + jt) * 4 + (ksub*2+half)) * 32 + nn
    long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)taps * nsteps_tap * ntiles * 128;
    if (gid >= total) return;
    const int nn = (int)(gid & 31);
    const int kh4 = (int)((gid >> 5) & 3);
    long rest = gid >> 7;
    const int jt = (int)(rest % ntiles);
    rest /= ntiles;
    const int cs = (int)(rest % nsteps_tap);
    const int t = (int)(rest / nsteps_tap);
    const int n = jt * 32 + nn;
    const int cbase = cs * 64 + kh4 * 16;
    int vals[16];
    int sum = 0;
    float d = 1.f, z = 0.f;
    if (n < Cout) { d = delta[n]; z = zp[n]; }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = cbase + j;
        int code = 0;
        if (n < Cout && c < clen) {
            const float wv = w[((long)n * Cin_total + c0 + c) * taps + t];
            float q;
            if (alpha) q = floorf(wv / d) + (alpha[((long)n * clen + c) * taps + t] >= 0.f ? 1.f : 0.f);
            else q = rintf(wv / d);
            q = fminf(fmaxf(q + z, 0.f), (float)(n_levels - 1));
            code = (int)q;
            sum += code;                       // raw nibble sum: the zero point is restored in the epilogue
        }
        vals[j] = code;
    }
    unsigned w0 = 0, w1 = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        w0 |= (unsigned)((vals[b] & 15) | ((vals[4 + b] & 15) << 4)) << (8 * b);
        w1 |= (unsigned)((vals[8 + b] & 15) | ((vals[12 + b] & 15) << 4)) << (8 * b);
    }
    const long kstep = kstep0 + (long)t * nsteps_tap + cs;
    uint2 pk = {w0, w1};
    *reinterpret_cast<uint2*>(wt + (kstep * ntiles + jt) * 1024 + (kh4 * 32 + nn) * 8) = pk;
    if (wsum && sum != 0) atomicAdd(&wsum[n], sum);
}

// split-K second pass: thread = one output element (n fastest).  Same float sequence as the fused epilogue.
template <typename TO>
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const int32_t* __restrict__ part, int nsplit, long MN, int Cout, int HoWo,
                                                              const float* __restrict__ scale, const int* __restrict__ zc,
                                                              const int* __restrict__ zw, const int* __restrict__ zfill,
                                                              const float* __restrict__ bias, const float* __restrict__ rowbias, long ldrb,
                                                              const TO* __restrict__ residual, long ldr, TO* __restrict__ out, long ldo) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= MN) return;
    const long m = e / Cout;
    const int  n = (int)(e - m * Cout);
    int I = 0;
    for (int s = 0; s < nsplit; ++s) I += part[(long)s * MN + e];
    const int kz = zfill ? zfill[1] : 0;
    I = I - (zc ? zc[n] : 0) + (zw ? zw[n] : 0) * kz;
    float v = (float)I * scale[n];
    v += bias ? bias[n] : 0.f;
    if (rowbias) v += rowbias[(m / HoWo) * ldrb + n];
    if constexpr (std::is_same<TO, float>::value) {
        if (residual) v += residual[m * ldr + n];
        out[m * ldo + n] = v;
    } else {
        if (residual) v += __half2float(residual[m * ldr + n]);
        out[m * ldo + n] = __float2half(v);
    }
}

// tile-ordered s8 packer: thread = one 16-byte unit (row n, 16 consecutive K), stored byte = W - 128
__global__ __launch_bounds__(256) void pack_t8_kernel(const float* __restrict__ w, const float* __restrict__ alpha,
                                                      const float* __restrict__ delta, const float* __restrict__ zp,
                                                      int Cout, int Cin_total, int taps, int c0, int clen,
                                                      int n_levels, uint8_t* __restrict__ wt, int kstep0, int ntiles,
                                                      int nsteps_tap, int32_t* __restrict__ wsum) {
    long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)taps * nsteps_tap * ntiles * 128;
    if (gid >= total) return;
    const int nn = (int)(gid & 31);
    const int kh4 = (int)((gid >> 5) & 3);
    long rest = gid >> 7;
    const int jt = (int)(rest % ntiles);
    rest /= ntiles;
    const int cs = (int)(rest % nsteps_tap);
    const int t = (int)(rest / nsteps_tap);
    const int n = jt * 32 + nn;
    const int cbase = cs * 64 + kh4 * 16;
    int sum = 0;
    float d = 1.f, z = 0.f;
    if (n < Cout) { d = delta[n]; z = zp[n]; }
    v4i pk;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned word = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = cbase + g * 4 + e;
            int stored = 0;
            if (n < Cout && c < clen) {
                const float wv = w[((long)n * Cin_total + c0 + c) * taps + t];
                float q;
                if (alpha) q = floorf(wv / d) + (alpha[((long)n * clen + c) * taps + t] >= 0.f ? 1.f : 0.f);
                else q = rintf(wv / d);
                q = fminf(fmaxf(q + z, 0.f), (float)(n_levels - 1));
                stored = (int)q - 128;
                sum += stored;
            }
            word |= (unsigned)(stored & 0xff) << (8 * e);
        }
        pk[g] = (int)word;
    }
    const long kstep = kstep0 + (long)t * nsteps_tap + cs;
    *reinterpret_cast<v4i*>(wt + (kstep * ntiles + jt) * 2048 + (kh4 * 32 + nn) * 16) = pk;
    if (wsum && sum != 0) atomicAdd(&wsum[n], sum);
}

template <int MT, int NT, int WB = 4>
int dispatch(ConvD& k, bool split, int out, hipStream_t st, int nsplit = 1) {
    constexpr int BM = 128 * MT, BN = 32 * NT;
    k.nblk_m = (k.M + BM - 1) / BM;
    k.nblk_n = (k.Cout + BN - 1) / BN;
    dim3 grid(k.nblk_m * k.nblk_n, nsplit), block(256);
#define QD_CASE(SP, O)                                                                      \
    if (split == SP && out == O) {                                                          \
        hipLaunchKernelGGL((igemm_dma_kernel<MT, NT, SP, O, WB>), grid, block, 0, st, k);   \
        return 0;                                                                           \
    }
    QD_CASE(false, O_F32) QD_CASE(false, O_F16) QD_CASE(false, O_I32)
    if constexpr (WB == 4) { QD_CASE(false, O_HROWS) QD_CASE(false, O_HTR) }
    if constexpr (MT == 1) { QD_CASE(true, O_F32) QD_CASE(true, O_F16) QD_CASE(false, O_PART) }
    if constexpr (NT == 4 && WB == 4) { QD_CASE(false, O_GEGLU) }
#undef QD_CASE
    qd_set_error("qd_conv2d_i8 (tiled): unsupported variant split=%d out=%d MT=%d", (int)split, out, MT);
    return 1;
}

// N-tile count of the MT=1 kernel the dispatcher would pick for this width
int nt_for(int N) { return N % 160 == 0 ? 5 : (N % 224 == 0 ? 7 : (N > 64 ? 4 : 2)); }

// Split-K policy.  Worth it only when the plain launch cannot fill the chip (<= 1 block per CU) AND the
// K loop is long enough that the extra int32 partial traffic (2 * 4 * M * N bytes per split) is paid back.
int choose_splitk(const qd_conv_desc* d, int* it_per) {
    *it_per = 0;
    if (!d->w_tiled || d->nseg != 1 || d->epilogue != QD_EPI_LINEAR) return 1;
    const long M = (long)d->B * d->Ho * d->Wo;
    const int  N = d->Cout, bn = 32 * (d->wbits == 8 ? (N > 64 ? 4 : 2) : nt_for(N));
    const long blocks0 = ((M + 127) / 128) * ((N + bn - 1) / bn);
    const int  total = d->kh * d->kw * ((d->seg[0].clen + 63) / 64);
    if (blocks0 > 256) return 1;
    const long mn = M * N;
    const int min_steps = mn <= (1L << 18) ? 2 : (mn <= (3L << 19) ? 4 : 16);
    long S = 512 / blocks0;
    if (S > total / min_steps) S = total / min_steps;
    if (S > 32) S = 32;
    if (S < 2) return 1;
    *it_per = (int)((total + S - 1) / S);
    return (total + *it_per - 1) / *it_per;
}

}  // namespace

extern "C" int64_t qd_conv2d_i8_splitk_ws_bytes(const qd_conv_desc* d) {
    if (!d) return 0;
    int it_per;
    const int S = choose_splitk(d, &it_per);
    return S < 2 ? 0 : (int64_t)S * d->B * d->Ho * d->Wo * d->Cout * 4;
}

int qd_conv2d_i8_tiled(const qd_conv_desc* d, int32_t* iout, void* stream) {
    QD_REQUIRE(d->wbits == 4 || d->wbits == 8, "tiled weights are int4 or int8");
    const bool w8 = d->wbits == 8;
    QD_REQUIRE(!w8 || d->epilogue == QD_EPI_LINEAR, "qd_conv2d_i8 (tiled): the fused GEGLU / head-layout epilogues are int4-weight only");
    QD_REQUIRE(d->ldx % 16 == 0 && qd_aligned(d->x, 16) && qd_aligned(d->w, 16), "qd_conv2d_i8 (tiled): x/w must be 16-byte aligned, ldx %% 16 == 0");
    ConvD k{};
    k.x = d->x; k.wt = d->w; k.out = d->out; k.iout = iout;
    k.bias = d->bias; k.rowbias = d->rowbias; k.residual = d->residual;
    k.ldx = d->ldx; k.ldo = d->ldo; k.ldr = d->ldr; k.ldrb = d->ld_rowbias;
    k.B = d->B; k.H = d->H; k.W = d->W; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout;
    k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad_t = d->pad_t; k.pad_l = d->pad_l;
    k.M = d->B * d->Ho * d->Wo; k.taps = d->kh * d->kw; k.nseg = d->nseg;
    k.ntiles = (d->Cout + 31) / 32;
    for (int s = 0; s < d->nseg; ++s) {
        const qd_conv_seg& g = d->seg[s];
        QD_REQUIRE(g.clen > 0 && g.clen % 16 == 0 && g.c0 % 16 == 0, "qd_conv2d_i8 (tiled): segment %d c0/clen must be multiples of 16", s);
        QD_REQUIRE(g.scale != nullptr, "qd_conv2d_i8 (tiled): segment %d has no scale vector", s);
        QD_REQUIRE(!g.fill16 || qd_aligned(g.fill16, 16), "qd_conv2d_i8 (tiled): fill16 must be 16-byte aligned");
        k.seg[s] = SegD{g.c0, g.clen, g.kstep0, (g.clen + 63) / 64, g.scale, g.zc, g.zw, g.zfill, g.fill16};
    }
    QD_REQUIRE((long)d->kh * d->kw * (d->seg[0].clen + (d->nseg == 2 ? d->seg[1].clen : 0)) < 32768,
               "qd_conv2d_i8 (tiled): K too long for the 24-bit zero-point multiply");
    QD_REQUIRE(d->ldo < (1 << 22) && d->ldr < (1 << 22) && d->Cout < (1 << 22), "qd_conv2d_i8 (tiled): row strides must be < 2^22 elements");
    const bool split = d->nseg == 2;
    const bool geglu = d->epilogue == QD_EPI_GEGLU_I8;
    const bool heads = d->epilogue == QD_EPI_HEADS_I8 || d->epilogue == QD_EPI_HEADS_T_I8;
    const int out = geglu ? O_GEGLU : heads ? (d->epilogue == QD_EPI_HEADS_I8 ? O_HROWS : O_HTR)
                                            : (iout ? O_I32 : (d->out_dtype == QD_F16 ? O_F16 : O_F32));
    if (heads) {
        QD_REQUIRE(!iout && d->nseg == 1 && d->oq_params && d->out, "qd_conv2d_i8 (tiled): heads epilogue needs one segment, oq_params and out");
        QD_REQUIRE(d->oq_max - d->oq_off <= 127 && d->oq_min - d->oq_off >= -128, "qd_conv2d_i8 (tiled): heads output grid does not fit int8");
        QD_REQUIRE(d->hd_H > 0 && d->hd_d > 0 && d->hd_H * d->hd_d == d->Cout, "qd_conv2d_i8 (tiled): heads epilogue: H*d must equal Cout");
        QD_REQUIRE(d->hd_T > 0 && d->hd_T % 128 == 0 && k.M % d->hd_T == 0, "qd_conv2d_i8 (tiled): heads epilogue: tokens per sample (%d) must be a multiple of 128 dividing M", d->hd_T);
        QD_REQUIRE(d->hd_Tpad % 32 == 0 && d->hd_Tpad >= d->hd_T && d->hd_dpad % 32 == 0 && d->hd_dpad >= d->hd_d, "qd_conv2d_i8 (tiled): heads epilogue: bad padded dims");
        QD_REQUIRE(qd_aligned(d->out, 16) && (d->epilogue != QD_EPI_HEADS_T_I8 || d->hd_sum), "qd_conv2d_i8 (tiled): heads epilogue: out unaligned or hd_sum missing");
        QD_REQUIRE(!d->rowbias && (!d->residual || (d->epilogue == QD_EPI_HEADS_I8 && d->out_dtype == QD_F32)),
                   "qd_conv2d_i8 (tiled): heads epilogue takes no rowbias; an fp32 residual only with QD_EPI_HEADS_I8");
        k.oq = d->oq_params; k.oqmin = (float)d->oq_min; k.oqmax = (float)d->oq_max; k.oqoff = d->oq_off;
        k.hdH = d->hd_H; k.hdd = d->hd_d; k.hdT = d->hd_T; k.hdTpad = d->hd_Tpad; k.hddpad = d->hd_dpad;
        k.oqpre = d->oq_prescale; k.hdsum = d->hd_sum;
    }
    if (d->gn_part) {
        QD_REQUIRE(!iout && !heads && !geglu && d->out_dtype == QD_F32, "qd_conv2d_i8 (tiled): gn_part needs the plain fp32 epilogue");
        QD_REQUIRE((d->Ho * d->Wo) % 128 == 0, "qd_conv2d_i8 (tiled): gn_part needs Ho*Wo %% 128 == 0 (a 128-row chunk stays inside one sample)");
        k.gnpart = d->gn_part;
        k.gn_nchunk = d->Ho * d->Wo / 128;
    }
    const bool mt2_ok = !heads || d->hd_T % 256 == 0;            // a block must stay inside one sample
    if (geglu) {
        QD_REQUIRE(!iout && d->nseg == 1 && d->oq_params && d->Cout % 64 == 0, "qd_conv2d_i8 (tiled): GEGLU epilogue needs one segment, oq_params and Cout %% 64 == 0");
        QD_REQUIRE(d->oq_max - d->oq_off <= 127 && d->oq_min - d->oq_off >= -128, "qd_conv2d_i8 (tiled): GEGLU output grid does not fit int8");
        k.oq = d->oq_params; k.oqmin = (float)d->oq_min; k.oqmax = (float)d->oq_max; k.oqoff = d->oq_off;
    }
    QD_REQUIRE(!(iout && split), "qd_conv2d_i8_acc: single segment only");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int N = d->Cout;
    const long M = k.M;
    int rc;
    auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    int it_per = 0;
    const int nsplit = (iout || !d->splitk_ws || d->gn_part) ? 1 : choose_splitk(d, &it_per);   // gn_part: fused epilogue only
    if (nsplit >= 2 && d->splitk_ws_bytes >= (int64_t)nsplit * M * N * 4) {
        QD_REQUIRE(qd_aligned(d->splitk_ws, 16), "qd_conv2d_i8 (tiled): splitk_ws must be 16-byte aligned");
        k.iout = reinterpret_cast<int32_t*>(d->splitk_ws);
        k.it_per = it_per;
        if (w8) rc = N > 64 ? dispatch<1, 4, 8>(k, false, O_PART, st, nsplit) : dispatch<1, 2, 8>(k, false, O_PART, st, nsplit);
        else switch (nt_for(N)) {
            case 5:  rc = dispatch<1, 5>(k, false, O_PART, st, nsplit); break;
            case 7:  rc = dispatch<1, 7>(k, false, O_PART, st, nsplit); break;
            case 4:  rc = dispatch<1, 4>(k, false, O_PART, st, nsplit); break;
            default: rc = dispatch<1, 2>(k, false, O_PART, st, nsplit); break;
        }
        if (rc) return rc;
        const SegD& sg = k.seg[0];
        const long MN = M * N;
        dim3 grid((unsigned)((MN + 255) / 256)), block(256);
        if (d->out_dtype == QD_F16)
            hipLaunchKernelGGL(splitk_finalize_kernel<__half>, grid, block, 0, st, k.iout, nsplit, MN, N, d->Ho * d->Wo, sg.scale, sg.zc, sg.zw,
                               sg.zfill, k.bias, k.rowbias, k.ldrb, (const __half*)k.residual, k.ldr, (__half*)k.out, k.ldo);
        else
            hipLaunchKernelGGL(splitk_finalize_kernel<float>, grid, block, 0, st, k.iout, nsplit, MN, N, d->Ho * d->Wo, sg.scale, sg.zc, sg.zw,
                               sg.zfill, k.bias, k.rowbias, k.ldrb, (const float*)k.residual, k.ldr, (float*)k.out, k.ldo);
        QD_LAUNCH_CHECK("qd_conv2d_i8 (tiled, split-K)");
        return 0;
    }
    static const int force_mt = getenv("QD_TILE_MT") ? atoi(getenv("QD_TILE_MT")) : 0;       // tuning knob: 1 / 2, 0 = heuristic
    static const int force_gmt = getenv("QD_GEGLU_MT") ? atoi(getenv("QD_GEGLU_MT")) : 0;
    if (w8) {                                      // int8 weights (CIFAR W8A8): 128-wide N tiles
        if (N > 64) {
            if (!split && blocks(256, 128) >= 512) rc = dispatch<2, 4, 8>(k, split, out, st);
            else rc = dispatch<1, 4, 8>(k, split, out, st);
        } else {
            rc = dispatch<1, 2, 8>(k, split, out, st);
        }
    } else if (geglu) {
        // 128-row tiles by default: the erf/quantise epilogue is VALU-heavy and overlaps better with other
        // blocks' main loops at 4 waves per SIMD (measured -0.24 ms per SD evaluation vs 256-row tiles)
        if (force_gmt == 2) rc = dispatch<2, 4>(k, split, out, st);
        else rc = dispatch<1, 4>(k, split, out, st);
    } else if (N % 160 == 0) {
        // 256-row tiles only pay off when the K loop is long: short-K layers are bound by their output stream and
        // run better as twice as many 128-row blocks whose load / store phases interleave (measured on SD, -0.3 ms)
        static const int mt2_mink = getenv("QD_MT2_MINK") ? atoi(getenv("QD_MT2_MINK")) : 2048;
        const long Ktot = (long)k.taps * d->seg[0].clen;
        if (!split && mt2_ok && Ktot >= mt2_mink && (force_mt ? force_mt == 2 : blocks(256, 160) >= 512)) rc = dispatch<2, 5>(k, split, out, st);      // >= 2 blocks per CU
        else rc = dispatch<1, 5>(k, split, out, st);
    } else if (N % 224 == 0) {
        rc = dispatch<1, 7>(k, split, out, st);
    } else if (N > 64) {
        if (!split && mt2_ok && blocks(256, 128) >= 512) rc = dispatch<2, 4>(k, split, out, st);
        else rc = dispatch<1, 4>(k, split, out, st);
    } else {
        rc = dispatch<1, 2>(k, split, out, st);
    }
    if (rc) return rc;
    QD_LAUNCH_CHECK("qd_conv2d_i8 (tiled)");
    return 0;
}

extern "C" int qd_pack_weights_t4(const float* w, const float* alpha, const float* delta, const float* zp, int Cout,
                                  int Cin_total, int taps, int c0, int clen, int clen_pad, int n_levels, uint8_t* wt,
                                  int kstep0, int ntiles, int32_t* wsum, void* stream) {
    QD_REQUIRE(w && delta && zp && wt, "qd_pack_weights_t4: null pointer");
    QD_REQUIRE(Cout > 0 && taps > 0 && clen > 0 && c0 >= 0 && c0 + clen <= Cin_total, "qd_pack_weights_t4: bad shape");
    QD_REQUIRE(clen_pad % 16 == 0 && clen_pad >= clen, "qd_pack_weights_t4: clen_pad must be a multiple of 16");
    QD_REQUIRE(n_levels >= 2 && n_levels <= 16, "qd_pack_weights_t4: n_levels %d does not fit a nibble", n_levels);
    QD_REQUIRE(ntiles == (Cout + 31) / 32 && qd_aligned(wt, 16), "qd_pack_weights_t4: bad tile layout");
    const int nsteps_tap = (clen_pad + 63) / 64;
    const long total = (long)taps * nsteps_tap * ntiles * 128;
    hipLaunchKernelGGL(pack_t4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       w, alpha, delta, zp, Cout, Cin_total, taps, c0, clen, clen_pad, n_levels, wt, kstep0, ntiles, nsteps_tap, wsum);
    QD_LAUNCH_CHECK("qd_pack_weights_t4");
    return 0;
}

extern "C" int qd_pack_weights_t8(const float* w, const float* alpha, const float* delta, const float* zp, int Cout,
                                  int Cin_total, int taps, int c0, int clen, int clen_pad, int n_levels, uint8_t* wt,
                                  int kstep0, int ntiles, int32_t* wsum, void* stream) {
    QD_REQUIRE(w && delta && zp && wt, "qd_pack_weights_t8: null pointer");
    QD_REQUIRE(Cout > 0 && taps > 0 && clen > 0 && c0 >= 0 && c0 + clen <= Cin_total, "qd_pack_weights_t8: bad shape");
    QD_REQUIRE(clen_pad % 16 == 0 && clen_pad >= clen, "qd_pack_weights_t8: clen_pad must be a multiple of 16");
    QD_REQUIRE(n_levels >= 2 && n_levels <= 256, "qd_pack_weights_t8: n_levels %d does not fit a byte", n_levels);
    QD_REQUIRE(ntiles == (Cout + 31) / 32 && qd_aligned(wt, 16), "qd_pack_weights_t8: bad tile layout");
    const int nsteps_tap = (clen_pad + 63) / 64;
    const long total = (long)taps * nsteps_tap * ntiles * 128;
    hipLaunchKernelGGL(pack_t8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       w, alpha, delta, zp, Cout, Cin_total, taps, c0, clen, n_levels, wt, kstep0, ntiles, nsteps_tap, wsum);
    QD_LAUNCH_CHECK("qd_pack_weights_t8");
    return 0;
}
