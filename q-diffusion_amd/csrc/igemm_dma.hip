// igemm_dma.hip — K3/K4: integer implicit-GEMM convolution / linear on v_mfma_i32_32x32x32_i8, LDS-DMA fed.
//
// Replaces F.conv2d / F.conv1d(k=1) / F.linear on fake-quantised fp32 operands (reference
// qdiff/quant_layer.py:256-276) by an exact int32 contraction of the stored codes and a fused dequantising
// epilogue (per-out-channel scale, zero-point restoration, bias, timestep-embedding row bias, residual, the
// split-shortcut second segment of quant_layer.py:257-269, optional fused consumers).
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = taps * channels.  Data path:
//   * both operands are copied global -> LDS by the DMA path (global_load_lds_dwordx4, 16 B/lane, no VGPR round
//     trip): activations per (tap, 64-channel K-step) with the im2col gather expressed in the per-lane SOURCE
//     address — out-of-image taps read a 16-byte buffer of the "true zero" byte z', K-tail / M-tail lanes read
//     zeros — and the LDS destination is lane-linear, so the XOR bank swizzle of the A tile is applied to the
//     source chunk index (cdna_hip_programming.md §5.4 rule 21);
//   * weights are pre-tiled at pack time (qd_pack_weights_t4 / _t8): one K-step x 32 output channels is a
//     contiguous 1-KB (int4) / 2-KB (int8) block already in MFMA B-fragment order, so the B copy is a straight
//     memcpy and the fragment read is one conflict-free ds_read_b64 / b128; RAW nibbles are unpacked at
//     fragment-read time (the weight zero point is restored in the epilogue through activation row sums), so
//     4-bit weights stay 4-bit all the way into LDS;
//   * a 3-deep LDS ring keeps two K-steps in flight across the single barrier per step (counted
//     s_waitcnt vmcnt(N), raw s_barrier).
// Round-2 rewrite of the main loop and the epilogue (what the round-1 counters and ISA showed: 26 % MFMA-busy on
// the long-K convolutions — every B fragment was read, waited for with lgkmcnt(0), unpacked and consumed before
// the next read was issued, every DMA sat behind a scalar kernarg load and its own branch — and 4-byte-per-lane
// epilogue traffic that made the short-K projections store-issue bound):
//   * the K-step body is ONE basic block: the last two steps (which prefetch nothing) are peeled, all loader
//     state lives in registers, a DMA costs ~6 VALU + 2 SALU and no branch, surplus B-tile DMAs of a wave
//     re-copy a tile another wave also copies (same bytes, same destination) instead of being predicated;
//   * fragments are software-pipelined: B fragments are read two MFMA groups ahead, the A fragments of the
//     second K half while the first half is being contracted;
//   * the epilogue goes through LDS: accumulators are dequantised in the MFMA C layout (one output channel per
//     lane: per-channel constants are scalars of the lane), transposed through a per-wave 4-KB LDS tile, and
//     leave row-major with 16 bytes per lane — residual / row-bias loads and output stores are 4x fewer, full
//     128-byte row segments; int8 consumers (GEGLU, attention operand rows) store 4..16 bytes per lane instead of 1.
// Block = 4 waves as WM x WN; wave tile = (32*MT) x (32*NT) of 32x32x32 MFMAs.
#include "common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void* qd_lds_ptr;

namespace {

__device__ __attribute__((aligned(16))) const int qd_zero16[4] = {0, 0, 0, 0};

struct SegD {
    int c0, clen, kstep0, nsteps_tap;
    const float*  scale;
    const int*    zc;
    const int*    zw;       // [Cout] weight zero point of the stored operand (raw nibble: zw; s8 byte W-128: zw-128)
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
    int vec;                  // 1: every row-major access of the epilogue may use 16-byte (4-element) vectors
    SegD seg[2];
    const float* oq;          // O_GEGLU / O_HROWS / O_HTR: {delta, zero_point} of the output quantiser
    float oqmin, oqmax;
    int   oqoff;
    int   it_per;             // O_PART: K-steps per split (blockIdx.y = split index)
    int   hdH, hdd, hdT, hdTpad, hddpad;   // O_HROWS / O_HTR: heads, head dim, tokens per sample, padded dims
    unsigned hdrcp;           // ceil(2^32 / hdd): n / hdd == umulhi(n, hdrcp) for n < 2^16 (host checks Cout)
    float oqpre;              // O_HROWS / O_HTR: multiplier applied before the output quantiser
    int32_t* hdsum;           // O_HTR: [(b*H+h)][dpad] column sums of the stored bytes (atomically accumulated)
    float* gnpart;            // O_F32, optional: per-(sample, 128-row chunk, channel) {sum, sum of squares} of the output,
    int    gn_nchunk;         //   i.e. the first level of GroupNorm's statistics (layout of gn_partial_kernel); S/128
    long   gn_ld;             //   channels per chunk row of gnpart (>= Cout: column range of a wider statistics buffer)
    int    pointwise;         // 1x1 convolution / linear layer without padding or stride: output row m is input pixel m
    int    res_f16;           // O_HROWS: the fp residual is fp16 (else fp32)
    int    ups;               // x is the HALF-resolution map [B][H/2][W/2][ldx]; the convolution runs on its nearest-2x up-sampling
};

// O_PART: split-K partial.  The block contracts K-steps [y*it_per, (y+1)*it_per) only and stores
// acc - zw[n]*Asum_part[m] (int32, exact) into slice y of the workspace; splitk_finalize_kernel sums
// the slices, restores the K-independent constants and applies the float epilogue.
// O_HROWS / O_HTR: the output feeds the attention kernel: it is quantised with the attention block's
// q/k/v activation quantiser and written as int8 straight into the head-major operand layout of
// attn_i8.hip (rows [(b,h)][t][dpad] for q and k; transposed + key-permuted [(b,h)][dd][t] plus column
// sums for v) — the fp32 projection output and the separate qd_quantize_heads pass disappear.
// O_BF16 (WB = 16 only): bf16 rows (+ bf16 residual) — the floating-point mode of the kernel (first-stage decoder, §N1).
enum { O_F32 = 0, O_F16 = 1, O_I32 = 2, O_GEGLU = 3, O_PART = 4, O_HROWS = 5, O_HTR = 6, O_BF16 = 7 };

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// 16-byte-per-lane global -> LDS DMA, issued through inline asm ON PURPOSE: with the builtin, hipcc cannot prove
// that the pending LDS writes do not alias the next ds_read and drains vmcnt(0) in front of every K-step's first
// LDS read.  An asm DMA is invisible to the compiler's waitcnt bookkeeping; completion is tracked by hand
// (wait_vmcnt<N> + s_barrier).  LDS destination = lds_base (wave-uniform, bytes) + lane*16; M0 is written in the
// same statement that reads it (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_base) {
#ifdef QD_ABL_NODMA            // measurement-only build (wrong results): the K-loop without its global -> LDS traffic
    asm volatile("" ::"v"(gsrc), "s"(lds_base) : "memory");
    return;
#endif
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_base)
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

__device__ __forceinline__ constexpr int crow(int r) { return (r & 3) + 8 * (r >> 2); }   // C-layout row of register r (+ 4*half)

// WB = weight bits of the tile-ordered operand: 4 (raw nibbles, 1 KB per K-step x 32 channels) or 8 (s8 bytes W-128,
// 2 KB); everything but the B tile size and the fragment read is shared.
// WB = 16: the same kernel as a bf16 x bf16 -> fp32 convolution (v_mfma_f32_32x32x16_bf16).  A 64-byte K-step is 32 bf16
// channels = two K = 16 MFMAs per tile — the byte geometry of the W8 path exactly (16-byte fragments, 2-KB weight tiles in
// [k-half][lane-half][n % 32][16 B] order), so the DMA loader, the im2col gather, the ring and the barriers are untouched;
// accumulators are fp32, there is no zero-point algebra (no row sums, no per-channel integer constants) and out-of-image
// taps read zeros.  All sizes the loader sees (ldx, c0, clen) are BYTES.
#ifndef QD_MT1_OCC
#define QD_MT1_OCC 2      // waves per SIMD the 128-row tiles are compiled for (3 fits without spills; A/B knob of build.py)
#endif
// LDS bytes of one block: the 3-stage ring (re-used by the epilogue as per-wave transposition tiles + GroupNorm partials) + row
// tables + per-channel constants
template <int MT, int NT, int WM, int WN, int OUT, int WB, int KS = 1>
__host__ __device__ constexpr int igemm_smem_bytes() {
    constexpr bool BF = WB >= 16;
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN, NTB = NT * WN;
    constexpr int TB = BF ? 2048 : 256 * WB;
    constexpr int STAGE = BM * 64 + NTB * TB;
    constexpr int TBW = ((OUT == O_F16 || OUT == O_HROWS) && !BF) ? 8192 : 4096;
    constexpr int EPI_BYTES = 4 * KS * TBW + 4 * (32 * NT) * 8;
    constexpr int RING = 3 * KS * STAGE > EPI_BYTES ? 3 * KS * STAGE : EPI_BYTES;
    return RING + (2 + (KS > 1 ? 1 : 0)) * BM * 4 + 4 * BN * 4;
}

// KS = 2 (round 6): TWO K-groups of four waves in one 512-thread block.  A launch whose tiles number at most one per CU (the
// 16 x 16 and 32 x 32 levels of SD at batch 16: 256 blocks) runs ONE wave per SIMD, and a single wave cannot keep its SIMD
// busy: the same layers at twice the batch — two independent blocks per CU — take 1.40 .. 1.44 x the time, not 2 x
// (profiles/r06_igemm_blocks_per_cu.txt).  Splitting K across two BLOCKS pays that back in int32 partial traffic (rounds 2 - 5);
// here the K range is split inside the block: group g contracts the K-steps s = g (mod 2) out of its own ring stages with its
// own four loader waves — each SIMD hosts one wave of either group — and the two accumulator sets meet in LDS (exact int32
// adds: the same integers, the same epilogue, the same bytes as the four-wave kernel).  The groups share the block's barrier.
template <int MT, int NT, int WM, int WN, bool SPLIT, int OUT, int WB, int KS = 1>
__device__ __forceinline__ void igemm_body(const ConvD& p, unsigned char* smem) {
    static_assert(KS == 1 || (KS == 2 && !SPLIT && WB <= 8 && (OUT == O_F32 || OUT == O_F16 || OUT == O_PART || OUT == O_HROWS || OUT == O_HTR)),
                  "two K-groups: one segment; linear rows, split-K partials or the head-layout epilogues");
    static_assert(WB == 4 || WB == 8 || WB == 16 || WB == 17, "weight bits (16 = bf16 mode, 17 = fp16 mode)");
    constexpr bool BF = WB >= 16;                 // the floating-point mode (either operand type: same bytes, same loop)
    constexpr bool FH = WB == 17;                 // ... on IEEE halves (v_mfma_f32_32x32x16_f16) instead of bf16
    static_assert(!BF || (!SPLIT && (OUT == O_F32 || OUT == (FH ? O_F16 : O_BF16))), "16-bit float mode: one segment, fp32 rows or rows of the operand type");
    static_assert(WM * WN == 4, "four waves per block");
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN, NTB = NT * WN;
    constexpr int TB = BF ? 2048 : 256 * WB;      // bytes of one (K-step, 32-channel) weight tile
    constexpr int A_BYTES = BM * 64, B_BYTES = NTB * TB;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NA = BM / 64;                   // A DMA instructions per wave per stage (16 rows each)
    constexpr int NBI = B_BYTES / 1024;           // B DMA instructions per stage, all waves together
    constexpr int NBW = (NBI + 3) / 4;            // ... per wave (surplus ones duplicate the last instruction: same bytes)
    constexpr int PER = NA + NBW;
    constexpr int WCOLS = 32 * NT;                // columns of one wave

    // The epilogue re-uses the (dead) ring as per-wave transposition tiles + the GroupNorm partials behind them: 4 KB per wave,
    // 8 KB for the fp16 stream (two MFMA tiles = 64 columns = one 128-byte line of halves per row, see the epilogue)
    constexpr int TBW = ((OUT == O_F16 || OUT == O_HROWS) && !BF) ? 8192 : 4096;
    constexpr int EPI_BYTES = 4 * KS * TBW + 4 * WCOLS * 8;
    constexpr int RS = KS * STAGE;                // ring stride: a K-group's stages are KS apart
    constexpr int RING = 3 * RS > EPI_BYTES ? 3 * RS : EPI_BYTES;
    static_assert(RING + (2 + (KS > 1 ? 1 : 0)) * BM * 4 + 4 * BN * 4 == igemm_smem_bytes<MT, NT, WM, WN, OUT, WB, KS>(), "igemm_smem_bytes out of step with the body");
    int* sRowB = reinterpret_cast<int*>(smem + RING);
    int* sAsum = sRowB + BM;
    int* sAsum1 = sAsum + BM;                     // KS == 2: the second K-group's activation row sums
    // per-output-channel epilogue constants of the LAST segment, fetched at kernel start so that their
    // global-load latency overlaps the prologue DMA instead of serialising in front of the stores
    float* sScale = reinterpret_cast<float*>(sAsum + (KS > 1 ? 2 : 1) * BM);
    int*   sZc    = reinterpret_cast<int*>(sScale + BN);
    int*   sZw    = sZc + BN;
    float* sBias  = reinterpret_cast<float*>(sZw + BN);

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kg = KS > 1 ? wave_all >> 2 : 0;    // K-group of this wave
    const int wave = wave_all & 3;                // wave inside its group: everything below is per group
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 31, fhalf = lane >> 5;

    const int nblk    = p.nblk_m * p.nblk_n;
    const int logical = qd_xcd_remap(blockIdx.x, nblk);
    const int mb = logical / p.nblk_n, nb = logical % p.nblk_n;
    const int m0 = mb * BM, n0 = nb * BN;

    // ---- loader: DMA instruction q covers tile rows q*16 .. q*16+15, lane -> (row, 16-B slot) ----------------
    const int lr16 = lane >> 2, slot = lane & 3;
    const int a_chunk = (slot ^ ((lr16 >> 2) & 3)) * 16;   // source byte offset inside a 64-B K-step landing in this lane's slot
    const int8_t* a_org[NA];                      // pixel (ho*stride - pad_t, wo*stride - pad_l) of this lane's row, + a_chunk
    unsigned a_mask[NA];                          // bit t: tap t of that row lies inside the image
    int a_hw0[NA];                                // ups only: (ih0 + 8) << 16 | (iw0 + 8) of the row's first tap on the UP-SAMPLED map
    bool a_valid[NA];
    const int HoWo = p.Ho * p.Wo;
    // 1x1 / linear layers (three quarters of the launches of a UNet evaluation, most of them short-K): output row m IS input
    // pixel m, so the im2col decomposition (two integer divisions per row, two more per tap for the in-image mask: ~40
    // instructions each) is skipped — the prologue is a large share of a block's life when K is 320.
    if (p.pointwise) {                            // set by the host: taps == 1, stride 1, no padding, Ho x Wo == H x W
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = (wave + 4 * i) * 16 + lr16;
            const int m = m0 + r;
            a_valid[i] = m < p.M;
            const int mm = a_valid[i] ? m : 0;
            a_org[i] = p.x + (long)mm * p.ldx + a_chunk;
            a_mask[i] = a_valid[i] ? 1u : 0u;
            a_hw0[i] = 0;
            if (p.rowbias != nullptr && slot == 0) sRowB[r] = mm / HoWo;
        }
    } else {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = (wave + 4 * i) * 16 + lr16;
        const int m = m0 + r;
        a_valid[i] = m < p.M;
        const int mm = a_valid[i] ? m : 0;
        const int b  = mm / HoWo;
        const int rem = mm - b * HoWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const int ih0 = ho * p.stride - p.pad_t, iw0 = wo * p.stride - p.pad_l;
        // nearest-2x folded into the gather (openaimodel.py:105-120 Upsample): tap pixel (ih, iw) of the up-sampled map is
        // pixel (ih >> 1, iw >> 1) of the stored one — not linear in the tap, so set_tap() computes it per row
        a_org[i] = p.ups ? p.x + (long)b * (p.H >> 1) * (p.W >> 1) * p.ldx + a_chunk
                         : p.x + ((long)b * p.H * p.W + (long)ih0 * p.W + iw0) * p.ldx + a_chunk;
        a_hw0[i] = ((ih0 + 8) << 16) | (iw0 + 8);
        unsigned msk = 0;
        for (int t = 0; t < p.taps; ++t) {
            const int ih = ih0 + t / p.kw, iw = iw0 + t % p.kw;
            if (a_valid[i] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) msk |= 1u << t;
        }
        a_mask[i] = msk;
        if (slot == 0) sRowB[r] = b;
    }
    }

    // ---- loader state (uniform) + per-lane running source pointers -------------------------------------------
    int it_begin = 0;                             // first K-step of this block (split-K partials only)
    if constexpr (OUT == O_PART) it_begin = blockIdx.y * p.it_per;
    int ls = 0;
    int seg_c0 = p.seg[0].c0, seg_clen = p.seg[0].clen, seg_nst = p.seg[0].nsteps_tap;
    const int8_t* zero16 = reinterpret_cast<const int8_t*>(qd_zero16);
    const int8_t* seg_fill = p.seg[0].fill16 ? p.seg[0].fill16 : zero16;
    int ltap = 0, lc = 0, lrr = 0, lq = 0;           // only split-K partials start in the middle of the K range
    if constexpr (OUT == O_PART) {
        ltap = it_begin / seg_nst; lc = it_begin - ltap * seg_nst;
        lrr = ltap / p.kw; lq = ltap - lrr * p.kw;
    }
    int krem = seg_clen - lc * 64;                // channels left in this tap from the next K-step on (>= 64: no K tail in it)
    const int8_t* a_cur[NA];                      // source of the NEXT K-step for DMA instruction i
    int           a_inc[NA];                      // 64 for real pixels, 0 for fill / zero sources
    const uint8_t* b_cur[NBW];
    int            b_inc[NBW];
    unsigned       b_dst[NBW];
#pragma unroll
    for (int r = 0; r < NBW; ++r) {
        const int bi = min(wave + 4 * r, NBI - 1);
        const bool ok = nb * NTB + bi * 1024 / TB < p.ntiles;   // the last N-block may cover n-tiles that do not exist
        b_cur[r] = ok ? p.wt + ((long)(p.seg[0].kstep0 + it_begin) * p.ntiles + (long)nb * NTB) * TB + bi * 1024 + lane * 16
                      : reinterpret_cast<const uint8_t*>(zero16);
        b_inc[r] = ok ? p.ntiles * TB : 0;
        b_dst[r] = A_BYTES + bi * 1024;
    }

    auto set_tap = [&]() __attribute__((always_inline)) {
        const long tap_off = ((long)lrr * p.W + lq) * p.ldx + seg_c0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool inb = (a_mask[i] >> ltap) & 1u;
            long off = tap_off;
            if (p.ups) {                              // wave-uniform; once per tap, not per K-step
                const int ih = (a_hw0[i] >> 16) - 8 + lrr, iw = (a_hw0[i] & 0xffff) - 8 + lq;
                off = ((long)(ih >> 1) * (p.W >> 1) + (iw >> 1)) * p.ldx + seg_c0;
            }
            a_cur[i] = inb ? a_org[i] + off : (a_valid[i] ? seg_fill : zero16);
            a_inc[i] = inb ? 64 : 0;
        }
    };
    set_tap();
    if constexpr (OUT == O_PART) {
#pragma unroll
        for (int i = 0; i < NA; ++i) a_cur[i] += lc * a_inc[i];
    }

    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));

    // DMA instruction d of the current loader step into the ring stage at byte offset `stage`
    auto issue_one = [&](unsigned stage, int d) __attribute__((always_inline)) {
        if (d < NA) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                if (i == d) {
                    // K tail: the last 64-wide step of a tap may run past the segment's channels
#ifdef QD_ABL_ZERODMA      // measurement-only build (wrong results): every DMA reads the same 16 bytes (issue cost without traffic)
                    const int8_t* src = zero16;
#else
                    const int8_t* src = a_chunk < krem ? a_cur[i] : zero16;
#endif
                    glds16(src, lds0 + stage + (wave + 4 * i) * 1024);
                    a_cur[i] += a_inc[i];
                }
        } else {
#pragma unroll
            for (int r = 0; r < NBW; ++r)
                if (r == d - NA) {
#ifdef QD_ABL_ZERODMA
                    glds16(zero16, lds0 + stage + b_dst[r]);
#else
                    glds16(b_cur[r], lds0 + stage + b_dst[r]);
#endif
                    b_cur[r] += b_inc[r];
                }
        }
    };
    // Every K-step prefetches — past the end of the K range the sources are switched to the 16 zero bytes (the copies
    // land in a ring stage nobody reads again), which keeps the K-step body free of "is there a step it+2" branches and
    // the vmcnt bookkeeping uniform.
    int lleft = 0;                                // K-steps the loader still has to issue (set below, once `total` is known)
    auto advance = [&]() __attribute__((always_inline)) {
        ++lc;
        krem -= 64;
        if (--lleft <= 0) {
#pragma unroll
            for (int i = 0; i < NA; ++i) { a_cur[i] = zero16; a_inc[i] = 0; }
#pragma unroll
            for (int r = 0; r < NBW; ++r) { b_cur[r] = reinterpret_cast<const uint8_t*>(zero16); b_inc[r] = 0; }
            lc = -(1 << 30);                      // never reaches a tap boundary again
            krem = 64;
        } else if (lc == seg_nst) {
            lc = 0; ++ltap; ++lq;
            if (lq == p.kw) { lq = 0; ++lrr; }
            if (ltap == p.taps) {
                ltap = 0; lrr = 0; lq = 0; ++ls;
                if (ls < p.nseg) {
                    seg_c0 = p.seg[1].c0; seg_clen = p.seg[1].clen; seg_nst = p.seg[1].nsteps_tap;
                    seg_fill = p.seg[1].fill16 ? p.seg[1].fill16 : zero16;
                }
            }
            krem = seg_clen;
            if (ls < p.nseg) set_tap();
        }
    };
    // KS == 2: the K-step between two of this group's steps belongs to the other group — move the loader over it
    auto skip_step = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) a_cur[i] += a_inc[i];
#pragma unroll
        for (int r = 0; r < NBW; ++r) b_cur[r] += b_inc[r];
        advance();
    };

    // ---- accumulators ------------------------------------------------------------------------------------
    typename std::conditional<BF, v16f, v16i>::type acc[MT][NT];
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

    const int wrow0 = wm * (32 * MT);             // first tile row of this wave
    auto publish_asum = [&]() __attribute__((always_inline)) {                 // row sums of this wave's rows -> LDS (both k-halves combined)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int v = asum[i] + __shfl_xor(asum[i], 32);
            if (fhalf == 0) sAsum[wrow0 + i * 32 + frow] = v;               // waves sharing rows (WN > 1) write identical values
            asum[i] = 0;
        }
    };

    const int nst0  = p.taps * p.seg[0].nsteps_tap;
    const int total_all = nst0 + (p.nseg == 2 ? p.taps * p.seg[1].nsteps_tap : 0);
    const int total = OUT == O_PART ? min(p.it_per, total_all - it_begin) : total_all;
    lleft = total;

    // lane-invariant fragment offsets inside a stage
    unsigned a_off[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = wrow0 + i * 32 + frow;
            a_off[i][ks] = row * 64 + (((ks * 2 + fhalf) ^ ((row >> 2) & 3)) * 16);
        }
    const unsigned b_off = A_BYTES + wn * NT * TB + (fhalf * 32 + frow) * (BF ? 16 : WB * 2);   // + ks*(TB/2) + j*TB

    constexpr int NPR = (BN + 255) / 256;              // per-channel constants fetched by each thread (BN may exceed the block size)
    float pr_scale[NPR], pr_bias[NPR];
    int   pr_zc[NPR], pr_zw[NPR];
    {
        const SegD& sgl = p.seg[p.nseg - 1];
#pragma unroll
        for (int u = 0; u < NPR; ++u) {
            const int cl = (int)threadIdx.x + 256 * u, pn = n0 + cl;
            pr_scale[u] = 0.f; pr_bias[u] = 0.f; pr_zc[u] = 0; pr_zw[u] = 0;
            if (cl < BN && pn < p.Cout) {
                if (!BF) pr_scale[u] = sgl.scale[pn];
                if (sgl.zc) pr_zc[u] = sgl.zc[pn];
                if (sgl.zw) pr_zw[u] = sgl.zw[pn];
                if (p.bias) pr_bias[u] = p.bias[pn];
            }
        }
    }

    // ---- prologue: two stages in flight ------------------------------------------------------------------------
    const unsigned gofs = kg * STAGE;                  // this K-group's first ring stage
    if (KS > 1 && kg == 1) skip_step();                // group 1 starts at K-step 1
#pragma unroll
    for (int d = 0; d < PER; ++d) issue_one(gofs, d);
    advance();
    if (KS > 1) skip_step();
#pragma unroll
    for (int d = 0; d < PER; ++d) issue_one(gofs + RS, d);
    advance();
    if (KS > 1) skip_step();
#pragma unroll
    for (int u = 0; u < NPR; ++u) {                    // visible to every wave after the main loop's barriers
        const int cl = (int)threadIdx.x + 256 * u;
        if (cl < BN) {
            sScale[cl] = pr_scale[u];
            sZc[cl]    = pr_zc[u];
            sZw[cl]    = pr_zw[u];
            sBias[cl]  = pr_bias[u];
        }
    }

    auto flush_segment0 = [&]() __attribute__((always_inline)) {
        publish_asum();
        __syncthreads();
        const SegD& sg = p.seg[0];
        const int kz = sg.zfill ? sg.zfill[1] : 0;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + wn * WCOLS + j * 32 + frow;
            const bool nok = n < p.Cout;
            const float sc = nok ? sg.scale[n] : 0.f;
            const int zc_n = (nok && sg.zc) ? sg.zc[n] : 0;
            const int zw_n = (nok && sg.zw) ? sg.zw[n] : 0;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rowl = wrow0 + i * 32 + crow(r) + 4 * fhalf;
                    const int I = (int)acc[i][j][r] - zc_n - __mul24(zw_n, sAsum[rowl] - kz);
                    facc[SPLIT ? i : 0][SPLIT ? j : 0][r] = (float)I * sc;
                    acc[i][j][r] = 0;
                }
        }
        __syncthreads();                               // sAsum is reused by the second segment
    };

    // ---- one K-step on the ring stage at byte offset `cur`, prefetching step it+2 into the stage at `nxt` ------------
    typedef typename std::conditional<WB == 4, uint2, v4i>::type braw_t;
    auto step = [&](unsigned cur, unsigned nxt) __attribute__((always_inline)) {
        wait_vmcnt<PER>();
        // stage `cur` landed for every wave; the stage before it is fully consumed.  (KS == 2: both K-groups meet here.  Letting each
        // group meet on its own LDS counter — ds_add + a bounded s_sleep poll instead of s_barrier, so that the two waves of a SIMD run
        // out of phase like two independent blocks — was built, bit-identical, and SLOWER: 84.6 -> 94.7 us on the 3 x 3 1280 -> 1280
        // layer, 18.01 -> 18.31 ms per SD step, profiles/r06_c9_k2_group_counters_*.txt; deleted.)
        __builtin_amdgcn_s_barrier();
        const unsigned char* cS = smem + cur;
        constexpr int S = 2 * NT;                      // (k-half, n-tile) MFMA groups of this step
        auto bread = [&](int s) __attribute__((always_inline)) {
            return *reinterpret_cast<const braw_t*>(cS + b_off + (s / NT) * (TB / 2) + (s % NT) * TB);
        };
        v4i af[2][MT];
        braw_t raw[S];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[0][i] = *reinterpret_cast<const v4i*>(cS + a_off[i][0]);
        raw[0] = bread(0);
        raw[1] = bread(1);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int ks = s / NT, j = s % NT;
            // software pipeline: the B fragment of group s+2 and (once) the A fragments of the second K half are read
            // here, two MFMA groups ahead of their use; the scheduling barrier keeps the compiler from sinking the reads
            // back next to their consumers (which is what it does to save registers, exposing the LDS latency)
            if (s + 2 < S) raw[s + 2] = bread(s + 2);
            if (s == (NT >= 2 ? NT - 2 : 0)) {
#pragma unroll
                for (int i = 0; i < MT; ++i) af[1][i] = *reinterpret_cast<const v4i*>(cS + a_off[i][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            v4i bf;
            if constexpr (WB == 4) {
#ifdef QD_ABL_NOUNPACK     // measurement-only build (wrong results): how much of a K-step is the nibble unpack?
                bf = v4i{(int)raw[s].x, (int)raw[s].x, (int)raw[s].y, (int)raw[s].y};
#else
                bf = v4i{(int)(raw[s].x & 0x0F0F0F0Fu), (int)((raw[s].x >> 4) & 0x0F0F0F0Fu),
                         (int)(raw[s].y & 0x0F0F0F0Fu), (int)((raw[s].y >> 4) & 0x0F0F0F0Fu)};
#endif
            } else {
                bf = raw[s];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if constexpr (FH) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, af[ks][i]), __builtin_bit_cast(v8h, bf),
                                                                        acc[i][j], 0, 0, 0);
                } else if constexpr (BF) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, af[ks][i]), __builtin_bit_cast(v8bf, bf),
                                                                         acc[i][j], 0, 0, 0);
                } else {
#ifdef QD_ABL_NOMFMA       // measurement-only build (wrong results): the K-step without its matrix instructions
                acc[i][j][0] += af[ks][i].x ^ bf.x ^ af[ks][i].w ^ bf.w;
#else
                acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ks][i], bf, acc[i][j], 0, 0, 0);
#endif
                }
            }
#ifndef QD_ABL_NOASUM        // measurement-only build (wrong results): the K-step without its activation row sums (4 v_dot4 per A fragment)
            if (!BF && j == 0) {
#pragma unroll
                for (int i = 0; i < MT; ++i) asum[i] += bytesum16(af[ks][i]);
            }
#endif
#ifndef QD_DMA_FRONT
            if (s < PER) issue_one(nxt, s);
#else                      // A/B knob: all DMAs of the step right behind the first MFMA group
            if (s == 0) {
#pragma unroll
                for (int d = 0; d < PER && d < S; ++d) issue_one(nxt, d);
            }
#endif
        }
#pragma unroll
        for (int d = S; d < PER; ++d) issue_one(nxt, d);
        advance();
        if (KS > 1) skip_step();
    };

    {
        unsigned cur = gofs, nxt = gofs + 2 * RS;
#ifdef QD_ABL_NOKLOOP          // measurement-only build (wrong results): prologue + epilogue, one K-step
        for (int it = 0; it < 1; ++it) {
#else
        // KS == 2: ceil(total / 2) rounds; with an odd step count the second group's last round contracts a stage of zeros (its
        // loader ran out one step earlier: the copies past the end read the zero block), which adds nothing
        for (int it = 0; it < (total + KS - 1) / KS; ++it) {
#endif
            step(cur, nxt);
            if (SPLIT && p.nseg == 2 && it == nst0 - 1) flush_segment0();
            nxt = cur;
            cur = cur == gofs + 2 * RS ? gofs : cur + RS;
        }
        wait_vmcnt<0>();                               // the last two steps' dummy prefetches: the ring is reused below
    }
    // KS == 2: after the K loop the two groups split the EPILOGUE by column tiles — group 0 owns tiles j < J0, group 1 the rest —
    // and hand each other the partial accumulators of the tiles they do not own (two waves per SIMD in the epilogue as well)
    constexpr int J0 = KS > 1 ? (NT == 7 ? 4 : 2) : NT;          // even: a tile pair of the fp16 epilogue never straddles the groups
    auto owner = [](int j) constexpr { return (KS > 1 && j >= J0) ? 1 : 0; };
    if constexpr (KS > 1) {
        // ---- the two K-groups meet: row sums, then the accumulators in batches through the (dead) ring — exact int32 adds ----
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int v = asum[i] + __shfl_xor(asum[i], 32);
            if (fhalf == 0) (kg == 1 ? sAsum1 : sAsum)[wrow0 + i * 32 + frow] = v;   // waves sharing rows (WN > 1) write identical values
        }
        constexpr int TPB = (RING / 16384) < (MT * NT) ? (RING / 16384) : (MT * NT);     // 32 x 32 tiles of all four waves per batch
        static_assert(TPB >= 1, "the ring holds at least one tile of every wave");
        v4i* xch = reinterpret_cast<v4i*>(smem);       // [tile in batch][wave][quarter][lane] 16 bytes
#pragma unroll
        for (int t0 = 0; t0 < MT * NT; t0 += TPB) {
            __syncthreads();                           // the ring is dead (first batch) / the previous batch has been taken
#pragma unroll
            for (int t = t0; t < t0 + TPB && t < MT * NT; ++t) {
                if (kg != owner(t % NT)) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd)
                        xch[(((t - t0) * 4 + wave) * 4 + qd) * 64 + lane] =
                            v4i{acc[t / NT][t % NT][4 * qd], acc[t / NT][t % NT][4 * qd + 1], acc[t / NT][t % NT][4 * qd + 2], acc[t / NT][t % NT][4 * qd + 3]};
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = t0; t < t0 + TPB && t < MT * NT; ++t) {
                if (kg == owner(t % NT)) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const v4i o = xch[(((t - t0) * 4 + wave) * 4 + qd) * 64 + lane];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[t / NT][t % NT][4 * qd + e] += o[e];
                    }
                }
            }
        }
        // total row sums for both groups (the partials were published before the first barrier above)
        for (int r = threadIdx.x; r < BM; r += 256 * KS) sAsum[r] += sAsum1[r];
        __syncthreads();                               // sAsum complete; the exchange area is the epilogue's transposition space from here on
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------
#ifdef QD_ABL_NOEPI            // measurement-only build (wrong results): prologue + K loop, one store per lane instead of the epilogue
    {
        int t = 0;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) t += acc[i][j][0] ^ acc[i][j][7] ^ acc[i][j][15];
        if (t == 0x7fffffff) reinterpret_cast<int*>(p.out)[threadIdx.x] = t;
        return;
    }
#endif
    if constexpr (KS == 1) {
        publish_asum();
        __syncthreads();                               // sAsum / sRowB visible to every wave; the ring is dead from here on
    }
    const SegD& sg = p.seg[p.nseg - 1];
    const int kz = sg.zfill ? sg.zfill[1] : 0;
    // Per-wave transposition tile (the ring is dead): 32 rows x 32 dwords.  Phase 1 writes one MFMA tile in the C layout
    // (lane = column, 16 rows per lane: conflict-free ds_write_b32), phase 2 reads it back row-major, 16 B per lane:
    // lane -> (row = pass*8 + lane/8, 4 columns from (lane%8)*4); zw * Asum uses the 24-bit multiplier (both factors
    // fit: |zw| <= 128, |Asum - kz| <= 2 * 128 * K < 2^23, checked on the host).
    unsigned* tb = reinterpret_cast<unsigned*>(smem + (kg * 4 + wave) * TBW);
    const int rr0 = lane >> 3, c4 = (lane & 7) * 4;
    const int wcol0 = n0 + wn * WCOLS;                 // first global column of this wave

    // Row terms of the zero-point algebra for one 32-row tile: the lane's 16 rows in the C layout are four runs of four
    // consecutive rows (r = 4g + e <-> row 8g + e + 4*fhalf), i.e. four 16-byte LDS reads instead of sixteen 4-byte ones;
    // kz * zw[n] moves into the per-channel constant (zc2 = zc - zw * kz), so an element costs one v_mad_i32_i24 and one
    // subtraction: I = acc - zc2 - zw * Asum (same integers: the host bounds every term, nothing wraps).
    auto row_terms = [&](int rbase, int (&as)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const v4i t = *reinterpret_cast<const v4i*>(sAsum + rbase + 8 * g4 + 4 * fhalf);
#pragma unroll
            for (int e = 0; e < 4; ++e) as[4 * g4 + e] = t[e];
        }
    };

    if constexpr (OUT == O_GEGLU) {
        // weight rows were packed (value tile, gate tile) interleaved: tiles 2jp / 2jp+1 of this lane hold the value
        // and the gate of output feature (wcol0/2) + jp*32 + frow.  y = value * gelu(gate) (erf GELU,
        // ldm/modules/attention.py:42-44), then the next Linear's act quantiser; bytes leave 16 per lane.
        static_assert(NT % 2 == 0, "GEGLU epilogue pairs n-tiles");
        constexpr int ROWB = 16 * NT;                  // output bytes per row of this wave
        constexpr int LPR = ROWB / 16;                 // lanes per row in phase 2
        constexpr int RPP = 64 / LPR;                  // rows per pass
        const QP oqp = qd_load_qp(p.oq);
        const QB oqb = qd_bytes_setup(oqp, p.oqmin, p.oqmax, p.oqoff);
        int8_t* tb8 = reinterpret_cast<int8_t*>(tb);   // [32 rows][ROWB]  (<= 4 KB for NT <= 8)
        const int Fout = p.Cout >> 1;
        const int f0 = (wcol0 >> 1);                   // first output feature of this wave
        int8_t* o8 = reinterpret_cast<int8_t*>(p.out) + f0;
        auto epi = [&](auto ft) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(ft)::value;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rbase = wrow0 + i * 32;
#pragma unroll
            for (int jp = 0; jp < NT / 2; ++jp) {
                const int lv = wn * WCOLS + (2 * jp) * 32 + frow, lg = lv + 32;     // tile-local channel of value / gate
                const float sv = sScale[lv], sgt = sScale[lg];
                const int zcv = sZc[lv], zcg = sZc[lg];
                const int zwv = sZw[lv], zwg = sZw[lg];
                const float bv = sBias[lv], bg = sBias[lg];
#pragma unroll
                for (int r = 0; r < 16; r += 2) {      // two rows per step: the float math runs packed (v_pk_*_f32)
                    const int rl0 = crow(r) + 4 * fhalf, rl1 = crow(r + 1) + 4 * fhalf;
                    const int as0 = sAsum[rbase + rl0] - kz, as1 = sAsum[rbase + rl1] - kz;
                    const v2f vi = {(float)(acc[i][2 * jp][r] - zcv - __mul24(zwv, as0)), (float)(acc[i][2 * jp][r + 1] - zcv - __mul24(zwv, as1))};
                    const v2f gi = {(float)(acc[i][2 * jp + 1][r] - zcg - __mul24(zwg, as0)), (float)(acc[i][2 * jp + 1][r + 1] - zcg - __mul24(zwg, as1))};
                    const v2f val = qd_fma2(vi, qd_splat2(sv), qd_splat2(bv)), gate = qd_fma2(gi, qd_splat2(sgt), qd_splat2(bg));
                    const v2f y = val * (qd_splat2(0.5f) * gate * (qd_splat2(1.0f) + qd_erff2(gate * qd_splat2(0.70710678118654752440f))));
                    int b0, b1;
                    qd_bytes2_t<FAST>(y, oqp, oqb, b0, b1);
                    tb8[rl0 * ROWB + jp * 32 + frow] = (int8_t)b0;
                    tb8[rl1 * ROWB + jp * 32 + frow] = (int8_t)b1;
                }
            }
#pragma unroll
            for (int ps = 0; ps < 32 / RPP; ++ps) {
                const int rl = ps * RPP + lane / LPR, cb = (lane % LPR) * 16;
                const v4i v = *reinterpret_cast<const v4i*>(tb8 + rl * ROWB + cb);
                const long m = m0 + rbase + rl;
                if (m < p.M && f0 + cb < Fout) *reinterpret_cast<v4i*>(o8 + m * p.ldo + cb) = v;
            }
        }
        };
        QD_FAST_DISPATCH(oqp.fast, epi);
        return;
    }
    if constexpr (OUT == O_HROWS) {
        // rows m = b*T + t, columns n = h*d + dd (d % 4 == 0).  The host guarantees T % BM == 0 and M % T == 0: a block
        // lies inside one sample and has no ragged rows.  Pad bytes (dd >= d) are never written: the operand buffers are
        // zero-initialised once and reused.  Phase 1 writes the fp32 projection, phase 2 adds the optional fp32 residual
        // (H = 1, d = Cout turns this epilogue into "Linear + residual -> the next Linear's int8 rows": the FF output of
        // a transformer block feeding SpatialTransformer.proj_out), quantises and stores 4 codes per lane.
        const QP oqp = qd_load_qp(p.oq);
        const QB oqb = qd_bytes_setup(oqp, p.oqmin, p.oqmax, p.oqoff);
        int8_t* o8 = reinterpret_cast<int8_t*>(p.out);
        const int bidx = m0 / p.hdT, t0 = m0 - bidx * p.hdT;
        const bool hres = p.residual != nullptr;
        const float* rf = reinterpret_cast<const float*>(p.residual);
        const __half* rh16 = reinterpret_cast<const __half*>(p.residual);
        const bool res16 = p.res_f16 != 0;             // the residual stream is fp16 (8-byte loads of four halves)
        auto epi = [&](auto ft) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(ft)::value;
        auto tile1 = [&](const int j) __attribute__((always_inline)) {
            const int cl = wn * WCOLS + j * 32 + frow;
            const float sc = sScale[cl];
            const int zw_n = sZw[cl];
            const int zc2 = sZc[cl] - zw_n * kz;
            const float bias_n = sBias[cl];
            const int n4 = wcol0 + j * 32 + c4;        // first of this lane's 4 columns in phase 2
            const bool nok = n4 < p.Cout;
            const int nn = nok ? n4 : 0;
            const int h = (int)__umulhi((unsigned)nn, p.hdrcp), dd = nn - h * p.hdd;     // nn / hdd (exact: nn < 2^16)
            int8_t* ob = o8 + (((long)bidx * p.hdH + h) * p.hdTpad + t0) * p.hddpad + dd;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int rbase = wrow0 + i * 32;
                int as[16];
                row_terms(rbase, as);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = crow(r) + 4 * fhalf;
                    const int I = acc[i][j][r] - zc2 - __mul24(zw_n, as[r]);
                    tb[rl * 32 + frow] = __float_as_uint(__builtin_fmaf((float)I, sc, bias_n));
                }
                v4f rs[4];
                if (hres) {
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const long ro = (long)(m0 + rbase + ps * 8 + rr0) * p.ldr + nn;
                        if (res16) rs[ps] = qd_ld4h(rh16 + ro);
                        else rs[ps] = *reinterpret_cast<const v4f*>(rf + ro);
                    }
                }
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int rl = ps * 8 + rr0;
                    v4f v = *reinterpret_cast<const v4f*>(tb + rl * 32 + c4);
                    if (hres) v += rs[ps];
                    const unsigned w = qd_pack4_t<FAST>(v[0] * p.oqpre, v[1] * p.oqpre, v[2] * p.oqpre, v[3] * p.oqpre, oqp, oqb);
                    if (nok) *reinterpret_cast<unsigned*>(ob + (long)(rbase + rl) * p.hddpad) = w;
                }
            }
        };
        // head dims that are multiples of 8 (40 / 80 / 160; one "head" as wide as the layer): TWO tiles per transposition
        // (the 8-KB swizzled LDS tile of the fp16 stream's full-line epilogue), 8 columns and ONE 8-byte store per lane — half
        // the phase-2 instructions per code; these launches are instruction-bound (a K = 320 q projection that writes 21 MB of
        // codes took longer than the same GEMM writing 84 MB of fp32).  Same values, same codes.
        auto tile2 = [&](const int j) __attribute__((always_inline)) {
            const int cl0 = wn * WCOLS + j * 32 + frow, cl1 = cl0 + 32;
            const float sc0 = sScale[cl0], sc1 = sScale[cl1];
            const int zw0 = sZw[cl0], zw1 = sZw[cl1];
            const int zc0 = sZc[cl0] - zw0 * kz, zc1 = sZc[cl1] - zw1 * kz;
            const float bias0 = sBias[cl0], bias1 = sBias[cl1];
            const int k8 = lane & 7, sw = (lane >> 3) & 1;
            const int n8 = wcol0 + j * 32 + k8 * 8;
            const bool nok = n8 < p.Cout;              // Cout % 8 == 0 (host)
            const int nn = nok ? n8 : 0;
            const int h = (int)__umulhi((unsigned)nn, p.hdrcp), dd = nn - h * p.hdd;     // 8 columns never straddle a head (hdd % 8 == 0)
            int8_t* ob = o8 + (((long)bidx * p.hdH + h) * p.hdTpad + t0) * p.hddpad + dd;
            const int rd_lo = ((2 * k8) ^ sw) * 4, rd_hi = ((2 * k8 + 1) ^ sw) * 4;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int rbase = wrow0 + i * 32;
                int as[16];
                row_terms(rbase, as);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = crow(r) + 4 * fhalf;
                    const int I0 = acc[i][j][r] - zc0 - __mul24(zw0, as[r]);
                    const int I1 = acc[i][j + 1][r] - zc1 - __mul24(zw1, as[r]);
                    const int pc = frow ^ ((r & 1) << 2);
                    tb[rl * 64 + pc] = __float_as_uint(__builtin_fmaf((float)I0, sc0, bias0));
                    tb[rl * 64 + 32 + pc] = __float_as_uint(__builtin_fmaf((float)I1, sc1, bias1));
                }
#pragma unroll
                for (int pb = 0; pb < 4; pb += 2) {
                    v4f ra[2], rb2[2];
                    if (hres) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const long ro = (long)(m0 + rbase + (pb + u) * 8 + rr0) * p.ldr + nn;
                            if (res16) qd_h8_to_f(*reinterpret_cast<const v4i*>(rh16 + ro), ra[u], rb2[u]);
                            else { ra[u] = *reinterpret_cast<const v4f*>(rf + ro); rb2[u] = *reinterpret_cast<const v4f*>(rf + ro + 4); }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int rl = (pb + u) * 8 + rr0;
                        v4f lo = *reinterpret_cast<const v4f*>(tb + rl * 64 + rd_lo);
                        v4f hi = *reinterpret_cast<const v4f*>(tb + rl * 64 + rd_hi);
                        if (hres) { lo += ra[u]; hi += rb2[u]; }
                        const unsigned w0 = qd_pack4_t<FAST>(lo[0] * p.oqpre, lo[1] * p.oqpre, lo[2] * p.oqpre, lo[3] * p.oqpre, oqp, oqb);
                        const unsigned w1 = qd_pack4_t<FAST>(hi[0] * p.oqpre, hi[1] * p.oqpre, hi[2] * p.oqpre, hi[3] * p.oqpre, oqp, oqb);
                        if (nok) *reinterpret_cast<uint2*>(ob + (long)(rbase + rl) * p.hddpad) = make_uint2(w0, w1);
                    }
                }
            }
        };
        if (p.vec == 2) {
#pragma unroll
            for (int q2 = 0; q2 < NT / 2; ++q2)
                if (owner(2 * q2) == kg) tile2(2 * q2);
            if constexpr (NT % 2 == 1) {
                if (owner(NT - 1) == kg) tile1(NT - 1);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j)
                if (owner(j) == kg) tile1(j);
        }
        };
        QD_FAST_DISPATCH(oqp.fast, epi);
        return;
    }
    if constexpr (OUT == O_HTR) {
        const QP oqp = qd_load_qp(p.oq);
        int8_t* o8 = reinterpret_cast<int8_t*>(p.out);
        const int bidx = m0 / p.hdT, t0 = m0 - bidx * p.hdT;
        int* sPart = reinterpret_cast<int*>(smem);            // [4][WCOLS] column-sum partials (the ring is dead by now)
        auto epi = [&](auto ft) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(ft)::value;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (owner(j) != kg) continue;              // KS == 2: the other K-group's column tile
            const int cl = wn * WCOLS + j * 32 + frow;
            const bool nok = n0 + cl < p.Cout;
            const int nn = nok ? n0 + cl : 0;
            const int h = (int)__umulhi((unsigned)nn, p.hdrcp), dd = nn - h * p.hdd;     // nn / hdd (exact: nn < 2^16)
            const float sc = sScale[cl];
            const int zc_n = sZc[cl], zw_n = sZw[cl];
            const float bias_n = sBias[cl];
            int8_t* ob = o8 + (((long)bidx * p.hdH + h) * p.hddpad + dd) * p.hdTpad + t0 + fhalf * 16;
            int csum = 0;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int tile0 = wrow0 + i * 32;      // first row of this 32-key tile inside the block
                v4i pk;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    unsigned w = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        const int rowl = tile0 + 4 * fhalf + crow(r);
                        const int I = acc[i][j][r] - zc_n - __mul24(zw_n, sAsum[rowl] - kz);
                        const float v = (float)I * sc + bias_n;
                        const int code = qd_code_t<FAST>(v * p.oqpre, oqp, p.oqmin, p.oqmax) - p.oqoff;
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
            if (fhalf == 0) sPart[wave * WCOLS + j * 32 + frow] = nok ? csum : 0;
        }
        };
        QD_FAST_DISPATCH(oqp.fast, epi);
        __syncthreads();
        const int c = threadIdx.x;
        if (c < BN && n0 + c < p.Cout) {
            const int wnc = c / WCOLS, cw = c - wnc * WCOLS;
            int tot = 0;
#pragma unroll
            for (int w = 0; w < WM; ++w) tot += sPart[(w * WN + wnc) * WCOLS + cw];
            const int n = n0 + c, h = n / p.hdd, dd = n - h * p.hdd;
            atomicAdd(&p.hdsum[((long)bidx * p.hdH + h) * p.hddpad + dd], tot);
        }
        return;
    }

    // ---- linear epilogues: fp32 / fp16 rows, raw int32 (test hook), split-K partials ----------------------------------------
    constexpr bool INT_OUT = OUT == O_PART || OUT == O_I32;
    constexpr bool H16_OUT = OUT == O_F16, B16_OUT = OUT == O_BF16;
    const bool has_rb = p.rowbias != nullptr, has_res = p.residual != nullptr;
    const bool vec = p.vec != 0;
    float*  const of = reinterpret_cast<float*>(p.out);
    __half* const oh = reinterpret_cast<__half*>(p.out);
    const float*  const rf = reinterpret_cast<const float*>(p.residual);
    const __half* const rh = reinterpret_cast<const __half*>(p.residual);
    int32_t* const oi = p.iout + (OUT == O_PART ? (long)blockIdx.y * p.M * p.Cout : 0L);
    // optional GroupNorm statistics of the tensor being written (consumed by qd_groupnorm_silu_quant instead of its own
    // pass over HBM): per-column partials of the lane's rows -> butterfly over the 8 lanes that share the columns ->
    // fixed-order LDS reduction over the waves of each 128-row chunk -> one {sum, sumsq} pair per (chunk, channel).
    const bool gn = (OUT == O_F32 || OUT == O_F16 || OUT == O_BF16) && p.gnpart != nullptr;   // statistics of the fp32 values (before a 16-bit store)
    float* sGn = reinterpret_cast<float*>(smem + 4 * KS * TBW);   // [4 waves][WCOLS][2], behind the transposition tiles (K-groups own disjoint columns)
    auto single = [&](const int j) __attribute__((always_inline)) {
        const int cl = wn * WCOLS + j * 32 + frow;
        const float sc = sScale[cl];
        const int zw_n = sZw[cl];
        const int zc2 = sZc[cl] - zw_n * kz;
        const float bias_n = sBias[cl];
        const int n4 = wcol0 + j * 32 + c4;            // this lane's 4 columns in phase 2
        const bool nok4 = n4 + 3 < p.Cout;             // all four exist (the vector path); else per element
        const int n4c = n4 < p.Cout ? n4 : 0;
        float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rbase = wrow0 + i * 32;
            // phase 1: dequantise in the C layout, park the tile in LDS
            int as[16];
            if constexpr (!BF) row_terms(rbase, as);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = crow(r) + 4 * fhalf;
                if constexpr (INT_OUT) {
                    const int v = OUT == O_PART ? (int)acc[i][j][r] - __mul24(zw_n, as[r])
                                                : (int)acc[i][j][r] - zc2 - __mul24(zw_n, as[r]);
                    tb[rl * 32 + frow] = (unsigned)v;
                } else if constexpr (BF) {
                    tb[rl * 32 + frow] = __float_as_uint((float)acc[i][j][r] + bias_n);
                } else {
                    const int I = acc[i][j][r] - zc2 - __mul24(zw_n, as[r]);
                    // one fma(I, scale, bias) — written out: the O_F16 pair form and the split-K finalise
                    // must produce this value bit for bit and may not depend on where the optimiser contracts
                    float v;
                    if (SPLIT) v = __builtin_fmaf((float)I, sc, facc[SPLIT ? i : 0][SPLIT ? j : 0][r]) + bias_n;
                    else v = __builtin_fmaf((float)I, sc, bias_n);
                    tb[rl * 32 + frow] = __float_as_uint(v);
                }
            }
            // phase 2: row-major, 4 columns per lane
            if constexpr (INT_OUT) {
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int rl = ps * 8 + rr0;
                    const long m = m0 + rbase + rl;
                    const v4i v = *reinterpret_cast<const v4i*>(tb + rl * 32 + c4);
                    if (m < p.M) {
                        int32_t* dst = oi + m * p.Cout + n4;
                        if (vec && nok4) *reinterpret_cast<v4i*>(dst) = v;
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (n4 + e < p.Cout) dst[e] = v[e];
                        }
                    }
                }
            } else {
                v4f rb[4], rs[4];
                long mrow[4];
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int rl = ps * 8 + rr0;
                    const long m = m0 + rbase + rl;
                    mrow[ps] = m < p.M ? m : m0;       // row m0 always exists: loads are clamped, only stores predicated
                    if (has_rb) {
                        const float* src = p.rowbias + (long)sRowB[rbase + rl] * p.ldrb + n4c;
                        if (vec && nok4) rb[ps] = *reinterpret_cast<const v4f*>(src);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) rb[ps][e] = n4 + e < p.Cout ? src[e] : 0.f;
                        }
                    }
                    if (has_res) {
                        if (OUT == O_F32) {
                            const float* src = rf + mrow[ps] * p.ldr + n4c;
                            if (vec && nok4) rs[ps] = *reinterpret_cast<const v4f*>(src);
                            else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) rs[ps][e] = n4 + e < p.Cout ? src[e] : 0.f;
                            }
                        } else if (B16_OUT) {
                            const unsigned short* src = reinterpret_cast<const unsigned short*>(p.residual) + mrow[ps] * p.ldr + n4c;
                            if (vec && nok4) rs[ps] = qd_ld4bf(src);
                            else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) rs[ps][e] = n4 + e < p.Cout ? qd_bf2f(src[e]) : 0.f;
                            }
                        } else {
                            const __half* src = rh + mrow[ps] * p.ldr + n4c;
                            if (vec && nok4) rs[ps] = qd_ld4h(src);
                            else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) rs[ps][e] = n4 + e < p.Cout ? __half2float(src[e]) : 0.f;
                            }
                        }
                    }
                }
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int rl = ps * 8 + rr0;
                    const bool mok = m0 + rbase + rl < p.M;
                    v4f v = *reinterpret_cast<const v4f*>(tb + rl * 32 + c4);
                    if (has_rb) v += rb[ps];
                    if (has_res) v += rs[ps];
                    if (mok) {
                        if (OUT == O_F32) {
                            float* dst = of + mrow[ps] * p.ldo + n4;
                            if (vec && nok4) *reinterpret_cast<v4f*>(dst) = v;
                            else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) if (n4 + e < p.Cout) dst[e] = v[e];
                            }
                        } else if (B16_OUT) {
                            unsigned short* dst = reinterpret_cast<unsigned short*>(p.out) + mrow[ps] * p.ldo + n4;
                            if (vec && nok4) qd_st4bf(dst, v);
                            else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (n4 + e < p.Cout) dst[e] = (unsigned short)(qd_pack2bf(v[e], 0.f) & 0xffffu);
                            }
                        } else {
                            __half* dst = oh + mrow[ps] * p.ldo + n4;
                            if (vec && nok4) qd_st4h(dst, v);
                            else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) if (n4 + e < p.Cout) dst[e] = __float2half(v[e]);
                            }
                        }
                        if (gn) {
#pragma unroll
                            // (explicit fma: the 4- and the 8-column forms must round alike — left to the optimiser, one is
                            //  packed into v_pk_mul + v_pk_add and the other contracted)
                            for (int e = 0; e < 4; ++e) { gs[e] += v[e]; gq[e] = __builtin_fmaf(v[e], v[e], gq[e]); }
                        }
                    }
                }
            }
        }
        if (gn) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int sh = 8; sh < 64; sh <<= 1) {
                    gs[e] += __shfl_xor(gs[e], sh);
                    gq[e] += __shfl_xor(gq[e], sh);
                }
            }
            if (lane < 8) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sGn[(wave * WCOLS + j * 32 + c4 + e) * 2] = gs[e];
                    sGn[(wave * WCOLS + j * 32 + c4 + e) * 2 + 1] = gq[e];
                }
            }
        }
    };
    // fp16 stream with 8-element-aligned rows (p.vec == 2): TWO MFMA tiles per transposition, i.e. 64 columns = one 128-byte
    // line of halves per row, and 16 bytes (8 halves) per lane in phase 2 — a store instruction writes eight FULL lines
    // (1 KB) where the 4-halves-per-lane form writes eight half lines: the store path is paced per request, not per byte
    // (profiles/r03_igemm_phase_ablation.md), so the half-line form moved half the bytes of the fp32 stream at the fp32
    // stream's speed.  LDS tile: [32 rows][64 dwords]; a ds_read_b128 group of 16 lanes covers rows (0, 1, 2, 3) x 4 chunks,
    // so the 16-byte chunk index is XOR-ed with (row & 1): conflict-free reads, and the ds_write_b32 side stays a
    // permutation of 32 consecutive banks.  Lane -> (row, columns) is the 4-wide form's (row = pass*8 + lane/8), so every
    // column's GroupNorm partial sum runs over the same rows in the same order: the statistics are bit-identical.
    auto pair = [&](const int j) __attribute__((always_inline)) {
        if constexpr (H16_OUT && !BF) {
        const int cl0 = wn * WCOLS + j * 32 + frow, cl1 = cl0 + 32;
        const float sc0 = sScale[cl0], sc1 = sScale[cl1];
        const int zw0 = sZw[cl0], zw1 = sZw[cl1];
        const int zc0 = sZc[cl0] - zw0 * kz, zc1 = sZc[cl1] - zw1 * kz;
        const float bias0 = sBias[cl0], bias1 = sBias[cl1];
        const int k8 = lane & 7, sw = (lane >> 3) & 1;
        const int n8 = wcol0 + j * 32 + k8 * 8;        // this lane's 8 columns in phase 2
        const bool nok8 = n8 < p.Cout;                 // Cout % 8 == 0 (host): all eight exist or none
        const int n8c = nok8 ? n8 : 0;
        const int rd_lo = ((2 * k8) ^ sw) * 4, rd_hi = ((2 * k8 + 1) ^ sw) * 4;
        // the row bias (one row per SAMPLE: the timestep-embedding projection) joins in phase 1 when a 32-row tile cannot
        // straddle two samples: one scalar per lane per tile instead of eight floats per lane per pass; same float order
        // ((I*scale + bias) + rowbias, then + residual) as the per-row form
        const bool rb_tile = has_rb && (HoWo & 31) == 0;
        const int nc0 = wcol0 + j * 32 + frow, nc1 = nc0 + 32;          // this lane's column in phase 1 (clamped for the load)
        const float* rbp0 = p.rowbias + (nc0 < p.Cout ? nc0 : 0);
        const float* rbp1 = p.rowbias + (nc1 < p.Cout ? nc1 : 0);
        float gs[8], gq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { gs[e] = 0.f; gq[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rbase = wrow0 + i * 32;
            __builtin_amdgcn_sched_barrier(0);         // a tile pair's work stays together: hoisting the next pair's dequantisation above this
                                                       // pair's row-major pass costs 32 more live registers (the MT = 2 tiles spill)
            float rbv0 = 0.f, rbv1 = 0.f;
            if (rb_tile) {
                const long ro = (long)sRowB[rbase] * p.ldrb;
                rbv0 = rbp0[ro];
                rbv1 = rbp1[ro];
            }
            int as[16];
            row_terms(rbase, as);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = crow(r) + 4 * fhalf;
                const int I0 = acc[i][j][r] - zc0 - __mul24(zw0, as[r]);
                const int I1 = acc[i][j + 1][r] - zc1 - __mul24(zw1, as[r]);
                float v0, v1;
                if (SPLIT) {
                    v0 = __builtin_fmaf((float)I0, sc0, facc[SPLIT ? i : 0][SPLIT ? j : 0][r]) + bias0;
                    v1 = __builtin_fmaf((float)I1, sc1, facc[SPLIT ? i : 0][SPLIT ? j + 1 : 0][r]) + bias1;
                } else {
                    v0 = __builtin_fmaf((float)I0, sc0, bias0);
                    v1 = __builtin_fmaf((float)I1, sc1, bias1);
                }
                if (rb_tile) { v0 += rbv0; v1 += rbv1; }
                const int pc = frow ^ ((r & 1) << 2);  // (row & 1) == (r & 1) in the C layout
                tb[rl * 64 + pc] = __float_as_uint(v0);
                tb[rl * 64 + 32 + pc] = __float_as_uint(v1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pb = 0; pb < 4; pb += 2) {        // two passes at a time: their residual loads in flight together, 8 registers
            v4i rs[2];
            long mrow[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long m = m0 + rbase + (pb + u) * 8 + rr0;
                mrow[u] = m < p.M ? m : m0;
                if (has_res) rs[u] = *reinterpret_cast<const v4i*>(rh + mrow[u] * p.ldr + n8c);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int rl = (pb + u) * 8 + rr0;
                const bool mok = m0 + rbase + rl < p.M;
                v4f lo = *reinterpret_cast<const v4f*>(tb + rl * 64 + rd_lo);
                v4f hi = *reinterpret_cast<const v4f*>(tb + rl * 64 + rd_hi);
                if (has_rb && !rb_tile) {
                    const float* src = p.rowbias + (long)sRowB[rbase + rl] * p.ldrb + n8c;
                    lo += *reinterpret_cast<const v4f*>(src);
                    hi += *reinterpret_cast<const v4f*>(src + 4);
                }
                if (has_res) {
                    v4f a, b;
                    qd_h8_to_f(rs[u], a, b);
                    lo += a; hi += b;
                }
                if (mok && nok8) {
                    const v4i pk = {(int)qd_pack2h(lo[0], lo[1]), (int)qd_pack2h(lo[2], lo[3]),
                                    (int)qd_pack2h(hi[0], hi[1]), (int)qd_pack2h(hi[2], hi[3])};
                    *reinterpret_cast<v4i*>(oh + mrow[u] * p.ldo + n8) = pk;
                    if (gn) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            gs[e] += lo[e]; gq[e] = __builtin_fmaf(lo[e], lo[e], gq[e]);
                            gs[4 + e] += hi[e]; gq[4 + e] = __builtin_fmaf(hi[e], hi[e], gq[4 + e]);
                        }
                    }
                }
            }
            }
        }
        if (gn) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int sh = 8; sh < 64; sh <<= 1) {
                    gs[e] += __shfl_xor(gs[e], sh);
                    gq[e] += __shfl_xor(gq[e], sh);
                }
            }
            if (lane < 8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sGn[(wave * WCOLS + j * 32 + k8 * 8 + e) * 2] = gs[e];
                    sGn[(wave * WCOLS + j * 32 + k8 * 8 + e) * 2 + 1] = gq[e];
                }
            }
        }
        }
    };
    bool lines16 = false;
    if constexpr (H16_OUT && !BF) lines16 = p.vec == 2;
    if (lines16) {
        // Pairs start at tile 0 whatever the alignment.  Where the wave's first column is an odd multiple of 32 (every other
        // block of the 160-wide tiles, wave column 1 of the 128 x 320 tile) a pair straddles two lines — two half-line
        // requests, what the 4-halves form issues everywhere.  The variant that peels tile 0 on those waves (every pair one
        // aligned line) was measured SLOWER: 19.33 vs 18.98 ms per SD step (profiles/r05_streams_ab.md) — a second copy of
        // the epilogue with other accumulator indices, 62 -> 109 spilled registers on the MT = 2 tiles.
#pragma unroll
        for (int q = 0; q < NT / 2; ++q)
            if (owner(2 * q) == kg) pair(2 * q);
        if constexpr (NT % 2 == 1) {
            if (owner(NT - 1) == kg) single(NT - 1);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j)
            if (owner(j) == kg) single(j);
    }
    if (gn) {
        __syncthreads();
        for (int c = threadIdx.x; c < BN; c += 256 * KS)
        if (n0 + c < p.Cout) {
            constexpr int WPC = MT >= 4 ? 1 : 4 / MT;         // waves (along M) per 128-row chunk
            const int wnc = c / WCOLS, cw = c - wnc * WCOLS;
#pragma unroll
            for (int ch = 0; ch < BM / 128; ++ch) {
                const int mrow = m0 + ch * 128;
                if (mrow >= p.M) break;
                float ts = 0.f, tq = 0.f;
#pragma unroll
                for (int w = 0; w < WPC; ++w) {
                    const int wv = (ch * WPC + w) * WN + wnc;
                    ts += sGn[(wv * WCOLS + cw) * 2];
                    tq += sGn[(wv * WCOLS + cw) * 2 + 1];
                }
                const int b = mrow / HoWo, chunk = (mrow - b * HoWo) >> 7;
                float* dst = p.gnpart + (((long)b * p.gn_nchunk + chunk) * p.gn_ld + n0 + c) * 2;
                dst[0] = ts;
                dst[1] = tq;
            }
        }
    }
}

template <int MT, int NT, int WM, int WN, bool SPLIT, int OUT, int WB>
__global__ __launch_bounds__(256, (SPLIT || MT * NT > 10) ? 1 : (MT == 1 && NT <= 5 ? QD_MT1_OCC : 2)) void igemm_kernel(const ConvD p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[igemm_smem_bytes<MT, NT, WM, WN, OUT, WB>()];
    igemm_body<MT, NT, WM, WN, SPLIT, OUT, WB>(p, smem);
}

template <int MT, int NT, int WM, int WN, int OUT, int WB>
__global__ __launch_bounds__(512, 1) void igemm_k2_kernel(const ConvD p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[igemm_smem_bytes<MT, NT, WM, WN, OUT, WB, 2>()];
    igemm_body<MT, NT, WM, WN, false, OUT, WB, 2>(p, smem);
}

// Grouped launch of the q / k / v projections of one attention block (reference qdiff/quant_block.py:193-199: three Linears on
// the rows of one LayerNorm): the SAME problem shape three times — member blockIdx.y runs its own descriptor (its own rows,
// weights, quantiser, operand buffer), q and k with the head-row epilogue, v with the transposed one.  One launch fills the
// chip where each of the three left it with one block per CU (256 blocks on 256 CUs: a single wave per SIMD), and two of the
// three prologue ramps and dependent-launch boundaries disappear.  Same code per member as the single launch: same bytes.
struct ConvG {
    ConvD d[3];
    int   out[3];             // O_HROWS / O_HTR per member
};
template <int MT, int NT, int WM, int WN, int WB>
__global__ __launch_bounds__(256, (MT * NT > 10) ? 1 : (MT == 1 && NT <= 5 ? QD_MT1_OCC : 2)) void igemm_heads_group_kernel(const ConvG g) {
    constexpr int SA = igemm_smem_bytes<MT, NT, WM, WN, O_HROWS, WB>(), SB = igemm_smem_bytes<MT, NT, WM, WN, O_HTR, WB>();
    __shared__ __attribute__((aligned(16))) unsigned char smem[SA > SB ? SA : SB];
    const int z = blockIdx.y;
    if (g.out[z] == O_HTR) igemm_body<MT, NT, WM, WN, false, O_HTR, WB>(g.d[z], smem);
    else igemm_body<MT, NT, WM, WN, false, O_HROWS, WB>(g.d[z], smem);
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
    float v = __builtin_fmaf((float)I, scale[n], bias ? bias[n] : 0.f);         // the fused epilogue's fma(I, scale, bias)
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

#ifdef QD_PROBE_INSTANCE
// measurement / inspection builds (tools/kernel_resources.py --probe): ONE instantiation of the kernel, nothing else — seconds
// instead of minutes per compile when the shared body is being edited.  -DQD_PROBE_INSTANCE="2,5,2,2,false,1,4"
template __global__ void igemm_kernel<QD_PROBE_INSTANCE>(const ConvD);
}  // namespace
#else
// tile shapes (MT, NT, WM, WN): block = (32*MT*WM) x (32*NT*WN)
// two K-groups per block where a launch has at most one tile per CU (default) or never (QD_KGROUPS=0 / qd_conv_config(0): A/B
// runs and the equality test)
static int& kgroups_knob() {
    static int v = (getenv("QD_KGROUPS") && atoi(getenv("QD_KGROUPS")) == 0) ? 0 : 1;
    return v;
}

// K loops of fewer than eight 64-byte steps stay on the four-wave block (nothing to split).  Measured with the epilogue divided
// between the groups (profiles/r06_c7_kgroups_igemm_ab.txt, one box, us per launch at batch 16): 3 x 3 1280 -> 1280 at 16 x 16
// 104.8 -> 86.1, 2560 -> 1280 188.9 -> 152.7, 3 x 3 640 -> 640 at 32 x 32 83.4 -> 74.5, 1 x 1 5120 -> 1280 47.9 -> 40.3,
// 1 x 1 1280 -> 1280 22.9 -> 19.8, 1 x 1 640 -> 640 at 32 x 32 (ten steps) 26.0 -> 24.7; SD step 19.84 -> 19.35 ms.
constexpr int kgroups_min_steps() { return 8; }

// qd_conv2d_i8_group runs every member through run() — all its checks, its tile choice — with this set: dispatch() then
// records the finished kernel descriptor and the tile it would have launched instead of launching.
struct GroupCapture { ConvD k; int out, tile; bool split; };
thread_local GroupCapture* g_capture = nullptr;
constexpr int tile_code(int MT, int NT, int WM, int WN, int WB) { return (((MT * 16 + NT) * 8 + WM) * 8 + WN) * 32 + WB; }

template <int MT, int NT, int WM, int WN, int WB = 4>
int dispatch(ConvD& k, bool split, int out, hipStream_t st, int nsplit = 1) {
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    k.nblk_m = (k.M + BM - 1) / BM;
    k.nblk_n = (k.Cout + BN - 1) / BN;
    if (g_capture) {
        *g_capture = GroupCapture{k, out, nsplit == 1 ? tile_code(MT, NT, WM, WN, WB) : -1, split};
        return 0;
    }
    // At most one tile per CU and a K loop worth splitting: two K-groups in a 512-thread block (igemm_body, KS = 2) — linear rows
    // on every tile shape, split-K partials and the head-layout epilogues on the shapes the three UNets launch them with
    if constexpr (WB <= 8 && (WM == 4 || (MT == 2 && NT == 5)) && (NT == 4 || NT == 5 || NT == 7) && (WB == 4 || NT == 4)) {
        const long nb = (long)k.nblk_m * k.nblk_n * nsplit;
        const int ksteps = out == O_PART ? k.it_per : k.taps * k.seg[0].nsteps_tap;
        if (kgroups_knob() && !split && k.nseg == 1 && nb <= 256 && ksteps >= kgroups_min_steps()) {
            dim3 grid2(k.nblk_m * k.nblk_n, nsplit), block2(512);
#define QD_K2(O) { hipLaunchKernelGGL((igemm_k2_kernel<MT, NT, WM, WN, O, WB>), grid2, block2, 0, st, k); return 0; }
            if (out == O_F32) QD_K2(O_F32)
            if constexpr (NT == 5 && MT == 1) { if (out == O_F16) QD_K2(O_F16) }        // (fp16 rows on the 256-row tiles spill: four-wave block)
            if constexpr (MT == 1 && WM == 4 && ((WB == 4 && (NT == 5 || NT == 7)) || (WB == 8 && NT == 4))) { if (out == O_PART) QD_K2(O_PART) }
            if constexpr (WB == 4 && NT == 5 && WM == 4) { if (out == O_HROWS) QD_K2(O_HROWS) if (out == O_HTR) QD_K2(O_HTR) }
#undef QD_K2
        }
    }
    dim3 grid(k.nblk_m * k.nblk_n, nsplit), block(256);
#define QD_CASE(SP, O)                                                                              \
    if (split == SP && out == O) {                                                                  \
        hipLaunchKernelGGL((igemm_kernel<MT, NT, WM, WN, SP, O, WB>), grid, block, 0, st, k);       \
        return 0;                                                                                   \
    }
    if constexpr (WB == 16) {
        QD_CASE(false, O_F32) QD_CASE(false, O_BF16)
    } else if constexpr (WB == 17) {
        QD_CASE(false, O_F32) QD_CASE(false, O_F16)
    } else {
        QD_CASE(false, O_F32)
        QD_CASE(false, O_F16)
        if constexpr (MT == 1 && WM == 4) { QD_CASE(false, O_I32) QD_CASE(true, O_F32) QD_CASE(true, O_F16) QD_CASE(false, O_PART) }
        // (int8 weights: the 128 x 128 tile only — the CIFAR attention block's q / k / v, 256 channels)
        if constexpr ((WB == 4 && WM == 4) || (WB == 8 && MT == 1 && NT == 4)) { QD_CASE(false, O_HROWS) QD_CASE(false, O_HTR) }
        if constexpr (NT == 4 && WB == 4 && WM == 4) { QD_CASE(false, O_GEGLU) }
    }
#undef QD_CASE
    qd_set_error("qd_conv2d_%s: unsupported variant split=%d out=%d tile %dx%d", WB >= 16 ? "bf16" : "i8", (int)split, out, BM, BN);
    return 1;
}

// N-tile count of the 128-row kernel the dispatcher would pick for this width
int nt_for(int N) { return N % 160 == 0 ? 5 : (N % 224 == 0 ? 7 : (N > 64 ? 4 : 2)); }

// Split-K policy.  Worth it only when the plain launch cannot fill the chip (<= 1 block per CU) AND the
// K loop is long enough that the extra int32 partial traffic (2 * 4 * M * N bytes per split) is paid back.
int choose_splitk(const qd_conv_desc* d, int* it_per) {
    *it_per = 0;
    if (d->nseg != 1 || d->epilogue != QD_EPI_LINEAR) return 1;
    const long M = (long)d->B * d->Ho * d->Wo;
    const int  N = d->Cout, bn = 32 * (d->wbits == 8 ? (N > 64 ? 4 : 2) : nt_for(N));
    const long blocks0 = ((M + 127) / 128) * ((N + bn - 1) / bn);
    const int  total = d->kh * d->kw * ((d->seg[0].clen + 63) / 64);
    if (blocks0 > 256) return 1;
    const long mn = M * N;
    const int min_steps = mn <= (1L << 18) ? 2 : (mn <= (3L << 19) ? 4 : 16);
    // blocks the split should reach.  One block per CU (256), not two: measured on one box (profiles/r05_c9_splitk_target_sweep.txt)
    // 512 -> 256 is -7 % on a CIFAR W8A8 evaluation (3.74 -> 3.48 ms: fewer finalise passes, and an unsplit layer hands its
    // GroupNorm statistics to the consumer from its epilogue), -0.12 ms on SD, equal on LDM-4; 128 is worse on SD and LDM-4.
    static const long target = getenv("QD_SPLITK_TARGET") ? atol(getenv("QD_SPLITK_TARGET")) : 256;
    long S = target / blocks0;
    if (S > total / min_steps) S = total / min_steps;
    if (S > 32) S = 32;
    if (S < 2) return 1;
    *it_per = (int)((total + S - 1) / S);
    return (total + *it_per - 1) / *it_per;
}

int run(const qd_conv_desc* d, int32_t* iout, void* stream) {
    QD_REQUIRE(d != nullptr, "qd_conv2d_i8: null descriptor");
    QD_REQUIRE(d->x && d->w && (d->out || iout), "qd_conv2d_i8: null tensor pointer");
    QD_REQUIRE(d->w_tiled, "qd_conv2d_i8: weights must be in the tile order of qd_pack_weights_t4 / _t8 (w_tiled = 1)");
    QD_REQUIRE(d->wbits == 4 || d->wbits == 8, "qd_conv2d_i8: wbits must be 4 or 8 (got %d)", d->wbits);
    QD_REQUIRE(!d->upsample2x || (d->stride == 1 && d->kh * d->kw > 1 && d->H % 2 == 0 && d->W % 2 == 0 && d->pad_t < 8 && d->pad_l < 8),
               "qd_conv2d_i8: upsample2x needs stride 1, more than one tap, even H and W");
    QD_REQUIRE(d->out_dtype == QD_F32 || d->out_dtype == QD_F16, "qd_conv2d_i8: out_dtype must be f32/f16");
    QD_REQUIRE(d->nseg == 1 || d->nseg == 2, "qd_conv2d_i8: nseg must be 1 or 2");
    QD_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0, "qd_conv2d_i8: bad shape");
    QD_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->kh * d->kw <= 32, "qd_conv2d_i8: bad kernel/stride (at most 32 taps)");
    QD_REQUIRE((long)d->B * d->Ho * d->Wo < (1L << 31), "qd_conv2d_i8: M overflows int32");
    const bool w8 = d->wbits == 8;
    QD_REQUIRE(!w8 || d->epilogue == QD_EPI_LINEAR || ((d->epilogue == QD_EPI_HEADS_I8 || d->epilogue == QD_EPI_HEADS_T_I8) && d->Cout > 64),
               "qd_conv2d_i8: int8 weights take the linear epilogue, or the head-layout epilogues on more than 64 output channels");
    QD_REQUIRE(d->ldx % 16 == 0 && qd_aligned(d->x, 16) && qd_aligned(d->w, 16), "qd_conv2d_i8: x/w must be 16-byte aligned, ldx %% 16 == 0");
    ConvD k{};
    k.x = d->x; k.wt = d->w; k.out = d->out; k.iout = iout;
    k.bias = d->bias; k.rowbias = d->rowbias; k.residual = d->residual;
    k.ldx = d->ldx; k.ldo = d->ldo; k.ldr = d->ldr; k.ldrb = d->ld_rowbias;
    k.B = d->B; k.H = d->H; k.W = d->W; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout;
    k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad_t = d->pad_t; k.pad_l = d->pad_l;
    k.M = d->B * d->Ho * d->Wo; k.taps = d->kh * d->kw; k.nseg = d->nseg;
    k.pointwise = k.taps == 1 && d->stride == 1 && d->pad_t == 0 && d->pad_l == 0 && d->H == d->Ho && d->W == d->Wo;
    k.ups = d->upsample2x ? 1 : 0;
    k.ntiles = (d->Cout + 31) / 32;
    for (int s = 0; s < d->nseg; ++s) {
        const qd_conv_seg& g = d->seg[s];
        QD_REQUIRE(g.clen > 0 && g.clen % 16 == 0 && g.c0 % 16 == 0, "qd_conv2d_i8: segment %d c0/clen must be multiples of 16", s);
        QD_REQUIRE(g.c0 + g.clen <= d->ldx, "qd_conv2d_i8: segment %d exceeds the activation row", s);
        QD_REQUIRE(g.scale != nullptr, "qd_conv2d_i8: segment %d has no scale vector", s);
        QD_REQUIRE(!g.fill16 || qd_aligned(g.fill16, 16), "qd_conv2d_i8: fill16 must be 16-byte aligned");
        k.seg[s] = SegD{g.c0, g.clen, g.kstep0, (g.clen + 63) / 64, g.scale, g.zc, g.zw, g.zfill, g.fill16};
    }
    QD_REQUIRE((long)d->kh * d->kw * (d->seg[0].clen + (d->nseg == 2 ? d->seg[1].clen : 0)) < 32768,
               "qd_conv2d_i8: K too long for the 24-bit zero-point multiply");
    const bool split = d->nseg == 2;
    const bool geglu = d->epilogue == QD_EPI_GEGLU_I8;
    const bool heads = d->epilogue == QD_EPI_HEADS_I8 || d->epilogue == QD_EPI_HEADS_T_I8;
    const int out = geglu ? O_GEGLU : heads ? (d->epilogue == QD_EPI_HEADS_I8 ? O_HROWS : O_HTR)
                                            : (iout ? O_I32 : (d->out_dtype == QD_F16 ? O_F16 : O_F32));
    // 16-byte row-major accesses in the epilogue: every base and row stride a multiple of 4 elements
    const size_t esz = d->out_dtype == QD_F16 ? 2 : 4;
    // (fp16 streams: four halves = 8 bytes per access; the row bias is always fp32)
    k.vec = d->Cout % 4 == 0 && (iout ? qd_aligned(iout, 16)
                                      : (d->ldo % 4 == 0 && qd_aligned(d->out, 4 * esz) &&
                                         (!d->residual || (d->ldr % 4 == 0 && qd_aligned(d->residual, 4 * esz))) &&
                                         (!d->rowbias || (d->ld_rowbias % 4 == 0 && qd_aligned(d->rowbias, 16)))));
    // fp16 rows whose every access may also be a 16-byte (8-half) vector: the full-line epilogue (two MFMA tiles per
    // transposition, see the kernel)
    if (k.vec && !iout && d->out_dtype == QD_F16 && d->Cout % 8 == 0 && d->ldo % 8 == 0 && qd_aligned(d->out, 16) &&
        (!d->residual || (d->ldr % 8 == 0 && qd_aligned(d->residual, 16))))
        k.vec = 2;
    if (heads) {
        QD_REQUIRE(!iout && d->nseg == 1 && d->oq_params && d->out, "qd_conv2d_i8: heads epilogue needs one segment, oq_params and out");
        QD_REQUIRE(d->oq_max - d->oq_off <= 127 && d->oq_min - d->oq_off >= -128, "qd_conv2d_i8: heads output grid does not fit int8");
        QD_REQUIRE(d->hd_H > 0 && d->hd_d > 0 && d->hd_H * d->hd_d == d->Cout, "qd_conv2d_i8: heads epilogue: H*d must equal Cout");
        QD_REQUIRE(d->hd_d % 4 == 0 && d->hd_dpad % 4 == 0, "qd_conv2d_i8: heads epilogue: head dim and its padding must be multiples of 4");
        QD_REQUIRE(d->hd_T > 0 && d->hd_T % 128 == 0 && k.M % d->hd_T == 0, "qd_conv2d_i8: heads epilogue: tokens per sample (%d) must be a multiple of 128 dividing M", d->hd_T);
        QD_REQUIRE(d->hd_Tpad % 32 == 0 && d->hd_Tpad >= d->hd_T && d->hd_dpad >= d->hd_d, "qd_conv2d_i8: heads epilogue: bad padded dims");
        QD_REQUIRE(qd_aligned(d->out, 16) && (d->epilogue != QD_EPI_HEADS_T_I8 || d->hd_sum), "qd_conv2d_i8: heads epilogue: out unaligned or hd_sum missing");
        QD_REQUIRE(!d->rowbias && (!d->residual || (d->epilogue == QD_EPI_HEADS_I8 && d->ldr % 4 == 0 && qd_aligned(d->residual, 4 * esz))),
                   "qd_conv2d_i8: heads epilogue takes no rowbias; a residual (fp32 or fp16 as out_dtype says, 4-element aligned rows) only with QD_EPI_HEADS_I8");
        k.res_f16 = d->out_dtype == QD_F16 ? 1 : 0;
        // head rows written 8 codes per lane (two tiles per transposition): head dim, padded head dim and the residual rows 8-aligned
        k.vec = (d->epilogue == QD_EPI_HEADS_I8 && d->hd_d % 8 == 0 && d->hd_dpad % 8 == 0 && d->Cout % 8 == 0 && qd_aligned(d->out, 8) &&
                 (!d->residual || (d->ldr % 8 == 0 && qd_aligned(d->residual, 16)))) ? 2 : 1;
        k.oq = d->oq_params; k.oqmin = (float)d->oq_min; k.oqmax = (float)d->oq_max; k.oqoff = d->oq_off;
        k.hdH = d->hd_H; k.hdd = d->hd_d; k.hdT = d->hd_T; k.hdTpad = d->hd_Tpad; k.hddpad = d->hd_dpad;
        k.oqpre = d->oq_prescale; k.hdsum = d->hd_sum;
        QD_REQUIRE(d->Cout < 65536, "qd_conv2d_i8: heads epilogue: Cout must be < 65536");
        k.hdrcp = (unsigned)((0x100000000ULL + (unsigned)d->hd_d - 1) / (unsigned)d->hd_d);
    }
    if (d->gn_part) {
        QD_REQUIRE(!iout && !heads && !geglu, "qd_conv2d_i8: gn_part needs the plain fp32 / fp16 epilogue");
        QD_REQUIRE((d->Ho * d->Wo) % 128 == 0, "qd_conv2d_i8: gn_part needs Ho*Wo %% 128 == 0 (a 128-row chunk stays inside one sample)");
        k.gnpart = d->gn_part;
        k.gn_nchunk = d->Ho * d->Wo / 128;
        QD_REQUIRE(d->gn_ld == 0 || d->gn_ld >= d->Cout, "qd_conv2d_i8: gn_ld must be 0 or >= Cout");
        k.gn_ld = d->gn_ld ? (long)d->gn_ld : (long)d->Cout;
    }
    const bool mt2_ok = !heads || d->hd_T % 256 == 0;            // a block must stay inside one sample
    if (geglu) {
        QD_REQUIRE(!iout && d->nseg == 1 && d->oq_params && d->Cout % 64 == 0, "qd_conv2d_i8: GEGLU epilogue needs one segment, oq_params and Cout %% 64 == 0");
        QD_REQUIRE(d->oq_max - d->oq_off <= 127 && d->oq_min - d->oq_off >= -128, "qd_conv2d_i8: GEGLU output grid does not fit int8");
        QD_REQUIRE(d->ldo % 16 == 0 && qd_aligned(d->out, 16), "qd_conv2d_i8: GEGLU output rows must be 16-byte aligned");
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
        QD_REQUIRE(qd_aligned(d->splitk_ws, 16), "qd_conv2d_i8: splitk_ws must be 16-byte aligned");
        k.iout = reinterpret_cast<int32_t*>(d->splitk_ws);
        k.it_per = it_per;
        k.vec = N % 4 == 0;
        if (w8) rc = N > 64 ? dispatch<1, 4, 4, 1, 8>(k, false, O_PART, st, nsplit) : dispatch<1, 2, 4, 1, 8>(k, false, O_PART, st, nsplit);
        else switch (nt_for(N)) {
            case 5:  rc = dispatch<1, 5, 4, 1>(k, false, O_PART, st, nsplit); break;
            case 7:  rc = dispatch<1, 7, 4, 1>(k, false, O_PART, st, nsplit); break;
            case 4:  rc = dispatch<1, 4, 4, 1>(k, false, O_PART, st, nsplit); break;
            default: rc = dispatch<1, 2, 4, 1>(k, false, O_PART, st, nsplit); break;
        }
        if (rc) return rc;
        const SegD& sg = k.seg[0];
        const long MN = M * N;
        dim3 grid((unsigned)((MN + 255) / 256)), block(256);
        // (an 8-outputs-per-thread form of this pass for fp16 rows was measured +15 us per launch: the pass is bound by its
        //  nsplit int32 reads per output, of which the one-output-per-thread form keeps eight times more in flight)
        if (d->out_dtype == QD_F16)
            hipLaunchKernelGGL(splitk_finalize_kernel<__half>, grid, block, 0, st, k.iout, nsplit, MN, N, d->Ho * d->Wo, sg.scale, sg.zc, sg.zw,
                               sg.zfill, k.bias, k.rowbias, k.ldrb, (const __half*)k.residual, k.ldr, (__half*)k.out, k.ldo);
        else
            hipLaunchKernelGGL(splitk_finalize_kernel<float>, grid, block, 0, st, k.iout, nsplit, MN, N, d->Ho * d->Wo, sg.scale, sg.zc, sg.zw,
                               sg.zfill, k.bias, k.rowbias, k.ldrb, (const float*)k.residual, k.ldr, (float*)k.out, k.ldo);
        QD_LAUNCH_CHECK("qd_conv2d_i8 (split-K)");
        return 0;
    }
    static const int wide_mink = getenv("QD_WIDE_MINK") ? atoi(getenv("QD_WIDE_MINK")) : 1280;
    static const int wide_minblk = getenv("QD_WIDE_MINBLK") ? atoi(getenv("QD_WIDE_MINBLK")) : 200;
    const long Ktot = (long)k.taps * d->seg[0].clen;
    const bool linear_out = out == O_F32 || out == O_F16;
    // 256-row tiles (two 32-row tiles per wave: half the B-fragment reads and nibble unpacks per MFMA) whenever they still
    // give every CU a block
    auto want_mt2 = [&](int bn) {
        if (split || !mt2_ok || out == O_I32) return false;
        // members of a grouped launch take 128-row tiles: three times the blocks fill the chip anyway, and the smaller tile's
        // epilogue tail is shorter (heads_i8_out 1.82 -> 1.74 ms per SD evaluation, profiles/r06_group_ab.md)
        if (g_capture) return false;
        return blocks(256, bn) >= 256;
    };
    if (w8) {                                      // int8 weights (CIFAR W8A8): 128-wide N tiles
        if (N > 64) {
            if (linear_out && want_mt2(128)) rc = dispatch<2, 4, 4, 1, 8>(k, split, out, st);
            else rc = dispatch<1, 4, 4, 1, 8>(k, split, out, st);
        } else {
            rc = dispatch<1, 2, 4, 1, 8>(k, split, out, st);
        }
    } else if (geglu) {
        rc = dispatch<1, 4, 4, 1>(k, split, out, st);          // (256-row GEGLU tiles: measured slower, profiles/r05_c4_ab_summary.txt)
    } else if (N % 320 == 0 && (out == O_F32 || out == O_F16) && !split && Ktot >= wide_mink && blocks(128, 320) >= wide_minblk) {
        // 2 x 2 waves of 64 x 160: a 128 x 320 block moves 18 KB per K-step into LDS for 40960 MACs per K element where the
        // 256 x 160 block of 4 x 1 waves moves 21 KB — the long-K convolutions are bound by exactly that L2 -> LDS traffic
        // (profiles/r02_igemm_kstep_ablation.md).  Measured (profiles/r02b_igemm_tiles.md): -5 .. -13 % on the 32 x 32 level
        // (M = 16384, N = 640), -5 .. -8 % on the 64 x 64 level (M = 65536, N = 320).  A 256 x 320 block of four 128 x 160
        // waves (one wave per SIMD, 512 VGPRs) was 2.2x SLOWER.
        rc = dispatch<2, 5, 2, 2>(k, split, out, st);
    } else if (N % 160 == 0) {
        if (want_mt2(160)) rc = dispatch<2, 5, 4, 1>(k, split, out, st);
        else rc = dispatch<1, 5, 4, 1>(k, split, out, st);
    } else if (N % 224 == 0) {
        rc = dispatch<1, 7, 4, 1>(k, split, out, st);
    } else if (N > 64) {
        if (want_mt2(128)) rc = dispatch<2, 4, 4, 1>(k, split, out, st);
        else rc = dispatch<1, 4, 4, 1>(k, split, out, st);
    } else {
        rc = dispatch<1, 2, 4, 1>(k, split, out, st);
    }
    if (rc) return rc;
    QD_LAUNCH_CHECK("qd_conv2d_i8");
    return 0;
}

// bf16 mode of the same kernel (first-stage decoder).  The descriptor's ldx / c0 / clen count bf16 ELEMENTS; the loader
// works in bytes.
int run_bf16(const qd_conv_desc* d, void* stream) {
    QD_REQUIRE(d != nullptr, "qd_conv2d_bf16: null descriptor");
    QD_REQUIRE(d->x && d->w && d->out, "qd_conv2d_bf16: null tensor pointer");
    QD_REQUIRE(d->w_tiled && (d->wbits == 16 || d->wbits == 17), "qd_conv2d_bf16: weights must come from qd_pack_weights_bf16 / _h16 (w_tiled = 1, wbits = 16: bf16, 17: fp16)");
    QD_REQUIRE(!d->upsample2x || (d->stride == 1 && d->kh * d->kw > 1 && d->H % 2 == 0 && d->W % 2 == 0 && d->pad_t < 8 && d->pad_l < 8),
               "qd_conv2d_bf16: upsample2x needs stride 1, more than one tap, even H and W");
    const bool fh = d->wbits == 17;                  // operands are IEEE halves
    QD_REQUIRE(d->out_dtype == QD_F32 || d->out_dtype == (fh ? QD_F16 : QD_BF16), "qd_conv2d_bf16: out_dtype must be f32 or the operand type");
    QD_REQUIRE(d->nseg == 1 && d->epilogue == QD_EPI_LINEAR && !d->rowbias, "qd_conv2d_bf16: one segment, linear epilogue, no row bias");
    QD_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0, "qd_conv2d_bf16: bad shape");
    QD_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->kh * d->kw <= 32, "qd_conv2d_bf16: bad kernel/stride (at most 32 taps)");
    QD_REQUIRE((long)d->B * d->Ho * d->Wo < (1L << 31), "qd_conv2d_bf16: M overflows int32");
    QD_REQUIRE((long)d->B * d->H * d->W * d->ldx * 2 < (1L << 32), "qd_conv2d_bf16: activation exceeds the 4-GiB offset range");
    QD_REQUIRE(d->ldx % 8 == 0 && qd_aligned(d->x, 16) && qd_aligned(d->w, 16), "qd_conv2d_bf16: x/w must be 16-byte aligned, ldx %% 8 == 0");
    const qd_conv_seg& g = d->seg[0];
    QD_REQUIRE(g.clen > 0 && g.clen % 8 == 0 && g.c0 % 8 == 0 && g.c0 + g.clen <= d->ldx, "qd_conv2d_bf16: c0/clen must be multiples of 8 inside the row");
    ConvD k{};
    k.x = reinterpret_cast<const int8_t*>(d->x); k.wt = d->w; k.out = d->out;
    k.bias = d->bias; k.residual = d->residual;
    k.ldx = d->ldx * 2; k.ldo = d->ldo; k.ldr = d->ldr;
    k.B = d->B; k.H = d->H; k.W = d->W; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout;
    k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad_t = d->pad_t; k.pad_l = d->pad_l;
    k.M = d->B * d->Ho * d->Wo; k.taps = d->kh * d->kw; k.nseg = 1;
    k.pointwise = k.taps == 1 && d->stride == 1 && d->pad_t == 0 && d->pad_l == 0 && d->H == d->Ho && d->W == d->Wo;
    k.ups = d->upsample2x ? 1 : 0;
    k.ntiles = (d->Cout + 31) / 32;
    k.seg[0] = SegD{g.c0 * 2, g.clen * 2, g.kstep0, (g.clen * 2 + 63) / 64, nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t esz = d->out_dtype == QD_F32 ? 4 : 2;
    k.vec = d->Cout % 4 == 0 && d->ldo % 4 == 0 && qd_aligned(d->out, 4 * esz) &&
            (!d->residual || (d->ldr % 4 == 0 && qd_aligned(d->residual, 4 * esz)));
    if (d->gn_part) {
        QD_REQUIRE((d->Ho * d->Wo) % 128 == 0, "qd_conv2d_bf16: gn_part needs Ho*Wo %% 128 == 0");
        k.gnpart = d->gn_part;
        k.gn_nchunk = d->Ho * d->Wo / 128;
        QD_REQUIRE(d->gn_ld == 0 || d->gn_ld >= d->Cout, "qd_conv2d_bf16: gn_ld must be 0 or >= Cout");
        k.gn_ld = d->gn_ld ? (long)d->gn_ld : (long)d->Cout;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int N = d->Cout, out = d->out_dtype == QD_BF16 ? O_BF16 : d->out_dtype == QD_F16 ? O_F16 : O_F32;
    const long M = k.M;
    const bool mt2 = ((M + 255) / 256) * ((N + 127) / 128) >= 256;
    int rc;
    if (fh) {
        if (N > 64) rc = mt2 ? dispatch<2, 4, 4, 1, 17>(k, false, out, st) : dispatch<1, 4, 4, 1, 17>(k, false, out, st);
        else rc = dispatch<1, 2, 4, 1, 17>(k, false, out, st);
    } else {
        if (N > 64) rc = mt2 ? dispatch<2, 4, 4, 1, 16>(k, false, out, st) : dispatch<1, 4, 4, 1, 16>(k, false, out, st);
        else rc = dispatch<1, 2, 4, 1, 16>(k, false, out, st);
    }
    if (rc) return rc;
    QD_LAUNCH_CHECK("qd_conv2d_bf16");
    return 0;
}

// fp32 OIHW (or [N, K] linear) weights -> bf16 (round to nearest even) in the tile order of the bf16 mode: K-step (32
// channels of one tap) x 32-output-channel tile = 2 KB as [k-half (16 ch)][lane-half (8 ch)][n % 32][8 bf16].
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, int clen_pad,
                                                        uint8_t* __restrict__ wt, int ntiles, int nsteps_tap, int fh) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)taps * nsteps_tap * ntiles * 128;
    if (gid >= total) return;
    const int nn = (int)(gid & 31);
    const int kh4 = (int)((gid >> 5) & 3);           // (k-half, lane-half): 8 channels each
    long rest = gid >> 7;
    const int jt = (int)(rest % ntiles);
    rest /= ntiles;
    const int cs = (int)(rest % nsteps_tap);
    const int t = (int)(rest / nsteps_tap);
    const int n = jt * 32 + nn;
    const int cbase = cs * 32 + kh4 * 8;
    v4i pk;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        float f[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = cbase + gq * 2 + e;
            if (n < Cout && c < Cin) f[e] = w[((long)n * Cin + c) * taps + t];
        }
        pk[gq] = fh ? (int)qd_pack2h(f[0], f[1]) : (int)qd_pack2bf(f[0], f[1]);
    }
    const long kstep = (long)t * nsteps_tap + cs;
    *reinterpret_cast<v4i*>(wt + (kstep * ntiles + jt) * 2048 + (kh4 * 32 + nn) * 16) = pk;
}

}  // namespace

extern "C" int qd_conv2d_bf16(const qd_conv_desc* d, void* stream) { return run_bf16(d, stream); }

extern "C" int64_t qd_pack_weights_bf16_bytes(int Cout, int taps, int clen_pad) {
    return (int64_t)taps * ((clen_pad + 31) / 32) * ((Cout + 31) / 32) * 2048;
}

extern "C" int qd_pack_weights_h16(const float* w, int Cout, int Cin, int taps, int clen_pad, int wbits, uint8_t* wt, void* stream);
extern "C" int qd_pack_weights_bf16(const float* w, int Cout, int Cin, int taps, int clen_pad, uint8_t* wt, void* stream) {
    return qd_pack_weights_h16(w, Cout, Cin, taps, clen_pad, 16, wt, stream);
}

extern "C" int qd_pack_weights_h16(const float* w, int Cout, int Cin, int taps, int clen_pad, int wbits, uint8_t* wt, void* stream) {
    QD_REQUIRE(wbits == 16 || wbits == 17, "qd_pack_weights_h16: wbits must be 16 (bf16) or 17 (fp16)");
    QD_REQUIRE(w && wt, "qd_pack_weights_bf16: null pointer");
    QD_REQUIRE(Cout > 0 && taps > 0 && Cin > 0, "qd_pack_weights_bf16: bad shape");
    QD_REQUIRE(clen_pad % 8 == 0 && clen_pad >= Cin, "qd_pack_weights_bf16: clen_pad must be a multiple of 8 and >= Cin");
    QD_REQUIRE(qd_aligned(wt, 16), "qd_pack_weights_bf16: wt must be 16-byte aligned");
    const int ntiles = (Cout + 31) / 32, nsteps_tap = (clen_pad + 31) / 32;
    const long total = (long)taps * nsteps_tap * ntiles * 128;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       w, Cout, Cin, taps, clen_pad, wt, ntiles, nsteps_tap, wbits == 17 ? 1 : 0);
    QD_LAUNCH_CHECK("qd_pack_weights_bf16");
    return 0;
}

extern "C" int64_t qd_conv2d_i8_splitk_ws_bytes(const qd_conv_desc* d) {
    if (!d) return 0;
    int it_per;
    const int S = choose_splitk(d, &it_per);
    return S < 2 ? 0 : (int64_t)S * d->B * d->Ho * d->Wo * d->Cout * 4;
}

template <int MT, int NT, int WM, int WN, int WB>
void launch_group(const ConvG& g, int n, hipStream_t st) {
    dim3 grid(g.d[0].nblk_m * g.d[0].nblk_n, n), block(256);
    hipLaunchKernelGGL((igemm_heads_group_kernel<MT, NT, WM, WN, WB>), grid, block, 0, st, g);
}

// Up to three projections of ONE shape with head-layout epilogues as one launch (see igemm_heads_group_kernel); whatever does
// not qualify — different shapes / tiles, another epilogue, a tile the group kernel is not built for — runs as
// the n single launches it always was: same bytes either way.
int run_group(const qd_conv_desc* const* descs, int n, void* stream) {
    QD_REQUIRE(descs && n >= 1 && n <= 3, "qd_conv2d_i8_group: 1..3 descriptors");
    GroupCapture cap[3];
    bool ok = n >= 2;
    for (int i = 0; i < n && ok; ++i) {
        QD_REQUIRE(descs[i] != nullptr, "qd_conv2d_i8_group: null descriptor");
        g_capture = &cap[i];
        const int rc = run(descs[i], nullptr, stream);
        g_capture = nullptr;
        if (rc) return rc;
        const ConvD &a = cap[i].k, &b = cap[0].k;
        ok = cap[i].tile >= 0 && !cap[i].split && (cap[i].out == O_HROWS || cap[i].out == O_HTR) && cap[i].tile == cap[0].tile &&
             a.M == b.M && a.Cout == b.Cout && a.nblk_m == b.nblk_m && a.nblk_n == b.nblk_n;
    }
    if (ok) {
        ConvG g{};
        for (int i = 0; i < n; ++i) { g.d[i] = cap[i].k; g.out[i] = cap[i].out; }
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        switch (cap[0].tile) {
            case tile_code(1, 5, 4, 1, 4): launch_group<1, 5, 4, 1, 4>(g, n, st); break;
            case tile_code(1, 7, 4, 1, 4): launch_group<1, 7, 4, 1, 4>(g, n, st); break;
            case tile_code(1, 4, 4, 1, 4): launch_group<1, 4, 4, 1, 4>(g, n, st); break;
            case tile_code(1, 4, 4, 1, 8): launch_group<1, 4, 4, 1, 8>(g, n, st); break;
            default: ok = false; break;
        }
        if (ok) {
            QD_LAUNCH_CHECK("qd_conv2d_i8_group");
            return 0;
        }
    }
    for (int i = 0; i < n; ++i) {
        const int rc = run(descs[i], nullptr, stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" void qd_conv_config(int kgroups) {
    if (kgroups >= 0) kgroups_knob() = kgroups ? 1 : 0;
}
extern "C" int qd_conv2d_i8_group(const qd_conv_desc* const* descs, int n, void* stream) { return run_group(descs, n, stream); }
extern "C" int qd_conv2d_i8(const qd_conv_desc* d, void* stream) { return run(d, nullptr, stream); }
extern "C" int qd_conv2d_i8_acc(const qd_conv_desc* d, int32_t* iout, void* stream) {
    if (!iout) { qd_set_error("qd_conv2d_i8_acc: null iout"); return 1; }
    return run(d, iout, stream);
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
#endif  // QD_PROBE_INSTANCE
