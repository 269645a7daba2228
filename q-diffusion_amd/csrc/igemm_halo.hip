// igemm_halo.hip — EXPERIMENTAL (off by default: QDIFF_HALO=1): 3x3 / stride-1 / pad-1 integer convolution whose activation
// operand stays resident in LDS across the nine taps.
//
// Why (profiles/r02_igemm_kstep_ablation.md, profiles/r02b_igemm_tiles.md): the long-K convolutions of
// csrc/igemm_dma.hip are bound by the bytes they pull from L2 into LDS, and the im2col gather re-reads every activation row
// nine times (once per tap).  Here a block owns R image rows x W pixels of ONE sample (R * W = 128 output pixels, W in
// {16, 32, 64}); per 64-channel slab it copies the (R+2) x (W+2)-pixel input patch into LDS ONCE (global_load_lds_dwordx4,
// out-of-image pixels read the "true zero" fill exactly like the gather kernel) and runs the nine taps against it: a tap is a
// constant pixel offset on the A-fragment address.  L2 -> LDS bytes per nine K-steps: 16.9 KB (slab, W = 64) + 9 x 10 KB
// (weights) = 107 KB against 9 x 8 + 9 x 10 = 162 KB of the 128 x 320 gather tile.
//
// Everything else follows igemm_dma.hip: 2 x 2 waves of 64 x 160 (MT = 2, NT = 5), tile-ordered int4 weights copied through a
// 3-stage ring (weights of step it+2 in flight), counted s_waitcnt vmcnt + one raw s_barrier per step, row sums by v_dot4
// on the A fragments, the same LDS-transposed fp32 epilogue (bias, time-embedding row bias, residual, GroupNorm statistics):
// results are bit-identical to qd_conv2d_i8's (tests/test_hip_kernels.py::test_halo_conv_equals_gather_kernel, enabled
// with QDIFF_HALO=1).  K-step order is [channel slab][tap] (the gather kernel's is [tap][slab]; integer sums commute).
//
// LDS: 2 slab buffers x 17 KB + 3 x 10 KB weight stages + 1 KB dummy target + row sums and per-channel constants = 72 KB
// (two blocks per CU).  The slab's 16-byte chunks are XOR-swizzled by the SLAB pixel index ((p >> 2) & 3): the 32 lanes of a
// fragment read touch 32 consecutive pixels whatever the tap, i.e. all 16 combinations of (p & 3, (p >> 2) & 3) per
// 16-lane group: conflict-free ds_read_b128, as in the gather kernel.
#include "common.h"

typedef __attribute__((address_space(3))) void* qd_lds_ptr_h;

namespace {

__device__ __attribute__((aligned(16))) const int qd_hzero16[4] = {0, 0, 0, 0};

struct HaloD {
    const int8_t*  x;
    const uint8_t* wt;
    float*         out;
    const float*   bias;
    const float*   rowbias;
    const float*   residual;
    long ldx, ldo, ldr, ldrb;
    int B, H, W, lw, Cout, M;
    int ups;                            // 1: x is the HALF-resolution map [B][H/2][W/2]; input pixel (iy, ix) reads (iy >> 1, ix >> 1)
    int c0, clen, kstep0, nst;          // single segment: first channel, channels, first K-step of the packed operand, 64-channel slabs
    int ntiles, nblk_n;
    const float*  scale;
    const int*    zc;
    const int*    zw;
    const int*    zfill;
    const int8_t* fill16;
    float* gnpart;
    int    gn_nchunk;
    long   gn_ld;
};

template <int N>
__device__ __forceinline__ void hwait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void hglds16(const void* gsrc, unsigned lds_base) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_base)
        : "memory");
}

__device__ __forceinline__ int hbytesum16(const v4i& v) {
    int s = __builtin_amdgcn_sdot4(v.x, 0x01010101, 0, false);
    s = __builtin_amdgcn_sdot4(v.y, 0x01010101, s, false);
    s = __builtin_amdgcn_sdot4(v.z, 0x01010101, s, false);
    return __builtin_amdgcn_sdot4(v.w, 0x01010101, s, false);
}

__device__ __forceinline__ constexpr int hcrow(int r) { return (r & 3) + 8 * (r >> 2); }

// a product that is rounded on its own (never fused into a following add)
__device__ __forceinline__ float hmul_rn(float a, float b) {
#pragma clang fp contract(off)
    float r = a * b;
    asm volatile("" : "+v"(r));          // the value is opaque to the optimiser from here on: nothing to contract it with
    return r;
}

constexpr int H_MT = 2, H_NT = 5, H_WM = 2, H_WN = 2;
constexpr int H_BM = 32 * H_MT * H_WM, H_BN = 32 * H_NT * H_WN, H_NTB = H_NT * H_WN;      // 128 x 320, 10 n-tiles
constexpr int H_TB = 1024;                                   // bytes of one (K-step, 32-channel) int4 weight tile
constexpr int H_SLAB_MAX = 17 * 1024;                        // (R+2)(W+2) <= 264 pixels -> 17 DMA instructions of 16 pixels
constexpr int H_BST = H_NTB * H_TB;                          // one weight stage
constexpr int H_OFF_B = 2 * H_SLAB_MAX, H_OFF_DUMMY = H_OFF_B + 3 * H_BST, H_RING = H_OFF_DUMMY + 1024;
constexpr int H_NJ = 5;                                      // slab DMA instructions per wave (17 / 4 rounded up)
constexpr int H_NBW = 3;                                     // weight DMA instructions per wave per stage (10 / 4 rounded up)
constexpr int H_PER = H_NBW + 1;                             // DMAs every wave issues per K-step (3 weight + 1 slab-or-dummy)

__global__ __launch_bounds__(256, 2) void igemm_halo_kernel(const HaloD p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[H_RING + H_BM * 4 + 4 * H_BN * 4];
    int*   sAsum  = reinterpret_cast<int*>(smem + H_RING);
    float* sScale = reinterpret_cast<float*>(sAsum + H_BM);
    int*   sZc    = reinterpret_cast<int*>(sScale + H_BN);
    int*   sZw    = sZc + H_BN;
    float* sBias  = reinterpret_cast<float*>(sZw + H_BN);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / H_WN, wn = wave % H_WN;
    const int frow = lane & 31, fhalf = lane >> 5;
    constexpr int WCOLS = 32 * H_NT;

    const int nblk_m = p.M / H_BM;
    const int logical = qd_xcd_remap(blockIdx.x, nblk_m * p.nblk_n);
    const int mb = logical / p.nblk_n, nb = logical % p.nblk_n;
    const int m0 = mb * H_BM, n0 = nb * H_BN;
    const int W = p.W, Wp = W + 2, R = H_BM >> p.lw;
    const int HW = p.H * W;
    const int b = m0 / HW;                                    // the whole block lies in one sample
    const int y0 = (m0 - b * HW) >> p.lw;                     // first image row of the block
    const int slabpix = (R + 2) * Wp;
    const int nq = (slabpix + 15) >> 4;                       // slab DMA instructions (16 pixels each)
    const int8_t* zero16 = reinterpret_cast<const int8_t*>(qd_hzero16);
    const int8_t* fill = p.fill16 ? p.fill16 : zero16;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(qd_lds_ptr_h)(smem));

    // ---- slab loader: DMA instruction q covers slab pixels 16q .. 16q+15; lane -> (pixel, 16-byte slot) ---------------------
    const int lp = lane >> 2, slot = lane & 3;
    const int8_t* s_src[H_NJ];                                // channel-0 address (+ swizzled chunk) of this lane's pixel, or the fill
    int s_inc[H_NJ];                                          // 64 per slab for real pixels, 0 for fill / zero sources
    int s_chunk[H_NJ];                                        // byte offset of this lane's source chunk inside a 64-channel slab
    unsigned s_dst[H_NJ];
#pragma unroll
    for (int j = 0; j < H_NJ; ++j) {
        const int q = min(wave + 4 * j, nq - 1);              // surplus instructions duplicate the last one (same bytes, same target)
        const int pix = q * 16 + lp;
        const int py = pix / Wp, px = pix - py * Wp;
        const int iy = y0 - 1 + py, ix = px - 1;
        const bool inslab = pix < slabpix;
        const bool inimg = inslab && iy >= 0 && iy < p.H && ix >= 0 && ix < W;
        const int c = (slot ^ ((pix >> 2) & 3)) * 16;
        s_chunk[j] = c;
        // nearest-neighbour 2x up-sampling folded into the gather (openaimodel.py:116 F.interpolate -> conv): the patch is read
        // from the small map, the in-image test runs on the up-sampled coordinates
        const long spix = p.ups ? (long)b * (HW >> 2) + (long)(iy >> 1) * (W >> 1) + (ix >> 1) : (long)b * HW + (long)iy * W + ix;
        s_src[j] = inimg ? p.x + spix * p.ldx + p.c0 + c : (inslab ? fill : zero16);
        s_inc[j] = inimg ? 64 : 0;
        s_dst[j] = q * 1024;
    }
    int krem_next = p.clen;                                   // channels left from the NEXT slab to be copied on
    auto slab_dma = [&](int j, unsigned buf) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < H_NJ; ++jj)
            if (jj == j) {
                const int8_t* src = s_chunk[jj] < krem_next ? s_src[jj] : zero16;   // K tail of the last slab: zeros for EVERY pixel
                hglds16(src, lds0 + buf + s_dst[jj]);
                s_src[jj] += s_inc[jj];
            }
    };

    // ---- weight loader: K-step (slab cs, tap t) is packed step kstep0 + t * nst + cs --------------------------------------
    int pcs = 0, ptap = 0, pleft = p.nst * 9;                 // next step whose weights have to be issued
    const uint8_t* b_base[H_NBW];
    unsigned b_dst[H_NBW];
#pragma unroll
    for (int r = 0; r < H_NBW; ++r) {
        const int bi = min(wave + 4 * r, H_NTB - 1);
        b_base[r] = p.wt + ((long)p.kstep0 * p.ntiles + (long)nb * H_NTB) * H_TB + bi * 1024 + lane * 16;
        b_dst[r] = bi * 1024;
    }
    const long b_step = (long)p.ntiles * H_TB;                // bytes between consecutive packed K-steps
    auto weights_dma = [&](int r, unsigned stage) __attribute__((always_inline)) {
        const long koff = ((long)ptap * p.nst + pcs) * b_step;
#pragma unroll
        for (int rr = 0; rr < H_NBW; ++rr)
            if (rr == r) {
                const void* src = pleft > 0 ? static_cast<const void*>(b_base[rr] + koff) : static_cast<const void*>(zero16);
                hglds16(src, lds0 + H_OFF_B + stage + b_dst[rr]);
            }
    };
    auto weights_advance = [&]() __attribute__((always_inline)) {
        --pleft;
        if (++ptap == 9) { ptap = 0; ++pcs; }
    };

    // ---- accumulators, fragment addressing --------------------------------------------------------------------------------
    v16i acc[H_MT][H_NT];
#pragma unroll
    for (int i = 0; i < H_MT; ++i)
#pragma unroll
        for (int j = 0; j < H_NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    int asum[H_MT] = {0, 0};
    const int wrow0 = wm * (32 * H_MT);
    int p0[H_MT];                                             // slab pixel of this lane's output row for tap (0, 0)
#pragma unroll
    for (int i = 0; i < H_MT; ++i) {
        const int r = wrow0 + i * 32 + frow;
        p0[i] = (r >> p.lw) * Wp + (r & (W - 1));
    }
    const unsigned b_off = H_OFF_B + wn * H_NT * H_TB + (fhalf * 32 + frow) * 8;      // + stage + ks * 512 + j * 1024

    // per-output-channel epilogue constants -> LDS (visible after the main loop's barriers)
    {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cl = (int)threadIdx.x + 256 * u, pn = n0 + cl;
            if (cl < H_BN) {
                const bool ok = pn < p.Cout;
                sScale[cl] = ok ? p.scale[pn] : 0.f;
                sZc[cl]    = (ok && p.zc) ? p.zc[pn] : 0;
                sZw[cl]    = (ok && p.zw) ? p.zw[pn] : 0;
                sBias[cl]  = (ok && p.bias) ? p.bias[pn] : 0.f;
            }
        }
    }

    // ---- prologue: slab 0, weights of steps 0 and 1 -------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < H_NJ; ++j) slab_dma(j, 0);
    krem_next -= 64;
#pragma unroll
    for (int r = 0; r < H_NBW; ++r) weights_dma(r, 0);
    weights_advance();
#pragma unroll
    for (int r = 0; r < H_NBW; ++r) weights_dma(r, H_BST);
    weights_advance();

    // ---- main loop ----------------------------------------------------------------------------------------------------------
    const int total = p.nst * 9;
    unsigned wcur = 0, wnxt = 2 * H_BST;                      // weight stage of this step / of step it+2
    int cs = 0, tap = 0, dy = 0, dx = 0;
    for (int it = 0; it < total; ++it) {
        if (it == 0) hwait_vmcnt<H_NBW>();                    // slab 0 and stage 0 landed (stage 1 may still be in flight)
        else hwait_vmcnt<H_PER>();                            // everything issued two steps ago or earlier landed
        __builtin_amdgcn_s_barrier();
        const unsigned char* slab = smem + (cs & 1) * H_SLAB_MAX;
        const unsigned char* bS = smem + wcur;
        const int toff = dy * Wp + dx;
        unsigned a_addr[H_MT][2];
#pragma unroll
        for (int i = 0; i < H_MT; ++i) {
            const int pix = p0[i] + toff;
            const int sw = (pix >> 2) & 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a_addr[i][ks] = pix * 64 + (((ks * 2 + fhalf) ^ sw) << 4);
        }
        constexpr int S = 2 * H_NT;
        auto bread = [&](int s) __attribute__((always_inline)) {
            return *reinterpret_cast<const uint2*>(bS + b_off + (s / H_NT) * 512 + (s % H_NT) * H_TB);
        };
        v4i af[2][H_MT];
        uint2 raw[S];
#pragma unroll
        for (int i = 0; i < H_MT; ++i) af[0][i] = *reinterpret_cast<const v4i*>(slab + a_addr[i][0]);
        raw[0] = bread(0);
        raw[1] = bread(1);
        const bool next_slab = cs + 1 < p.nst;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int ks = s / H_NT, j = s % H_NT;
            if (s + 2 < S) raw[s + 2] = bread(s + 2);
            if (s == H_NT - 2) {
#pragma unroll
                for (int i = 0; i < H_MT; ++i) af[1][i] = *reinterpret_cast<const v4i*>(slab + a_addr[i][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const v4i bf = {(int)(raw[s].x & 0x0F0F0F0Fu), (int)((raw[s].x >> 4) & 0x0F0F0F0Fu),
                            (int)(raw[s].y & 0x0F0F0F0Fu), (int)((raw[s].y >> 4) & 0x0F0F0F0Fu)};
#pragma unroll
            for (int i = 0; i < H_MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ks][i], bf, acc[i][j], 0, 0, 0);
            if (j == 0) {
#pragma unroll
                for (int i = 0; i < H_MT; ++i) asum[i] += hbytesum16(af[ks][i]);
            }
            // this step's four DMAs, one behind each of the first four MFMA groups: weights of step it+2, then one
            // instruction of the NEXT slab (taps 0..4) or a 16-byte dummy (taps 5..8, last slab) so that the count is uniform
            if (s < H_NBW) weights_dma(s, wnxt);
            if (s == H_NBW) {
                if (next_slab && tap < H_NJ) slab_dma(tap, ((cs + 1) & 1) * H_SLAB_MAX);
                else hglds16(zero16, lds0 + H_OFF_DUMMY);
            }
        }
        weights_advance();
        wnxt = wcur;
        wcur = wcur == 2 * H_BST ? 0 : wcur + H_BST;
        if (++dx == 3) { dx = 0; ++dy; }
        if (++tap == 9) {
            tap = 0; dy = 0; ++cs;
            krem_next -= 64;
        }
    }
    hwait_vmcnt<0>();

    // ---- epilogue (the linear fp32 epilogue of igemm_dma.hip; every row and every 4-column group exists) ------------------------
#pragma unroll
    for (int i = 0; i < H_MT; ++i) {
        const int v = asum[i] + __shfl_xor(asum[i], 32);
        if (fhalf == 0) sAsum[wrow0 + i * 32 + frow] = v;     // the two waves that share these rows write identical values
    }
    __syncthreads();
    const int kz = p.zfill ? p.zfill[1] : 0;
    unsigned* tb = reinterpret_cast<unsigned*>(smem + wave * 4096);
    const int rr0 = lane >> 3, c4 = (lane & 7) * 4;
    const int wcol0 = n0 + wn * WCOLS;
    const bool has_rb = p.rowbias != nullptr, has_res = p.residual != nullptr, gn = p.gnpart != nullptr;
    float* sGn = reinterpret_cast<float*>(smem + 4 * 4096);   // [4 waves][WCOLS][2]
#pragma unroll
    for (int j = 0; j < H_NT; ++j) {
        const int cl = wn * WCOLS + j * 32 + frow;
        const float sc = sScale[cl];
        const int zc_n = sZc[cl], zw_n = sZw[cl];
        const float bias_n = sBias[cl];
        const int n4 = wcol0 + j * 32 + c4;
        float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < H_MT; ++i) {
            const int rbase = wrow0 + i * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = hcrow(r) + 4 * fhalf;
                const int I = acc[i][j][r] - zc_n - __mul24(zw_n, sAsum[rbase + rl] - kz);
                const float v = (float)I * sc;                  // (same statement structure as igemm_dma.hip: identical contraction)
                tb[rl * 32 + frow] = __float_as_uint(v + bias_n);
            }
            v4f rb, rs[4];
            if (has_rb) rb = *reinterpret_cast<const v4f*>(p.rowbias + (long)b * p.ldrb + n4);
            if (has_res) {
#pragma unroll
                for (int ps = 0; ps < 4; ++ps)
                    rs[ps] = *reinterpret_cast<const v4f*>(p.residual + (long)(m0 + rbase + ps * 8 + rr0) * p.ldr + n4);
            }
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int rl = ps * 8 + rr0;
                v4f v = *reinterpret_cast<const v4f*>(tb + rl * 32 + c4);
                if (has_rb) v += rb;
                if (has_res) v += rs[ps];
                *reinterpret_cast<v4f*>(p.out + (long)(m0 + rbase + rl) * p.ldo + n4) = v;
                if (gn) {
#pragma unroll
                    // separate multiply and add (as the gather kernel's epilogue is compiled): the statistics are bit-identical
                    for (int e = 0; e < 4; ++e) { gs[e] += v[e]; gq[e] += hmul_rn(v[e], v[e]); }
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
    }
    if (gn) {
        __syncthreads();
        for (int c = threadIdx.x; c < H_BN; c += 256) {
            const int wnc = c / WCOLS, cw = c - wnc * WCOLS;
            float ts = 0.f, tq = 0.f;                         // one 128-row chunk per block: the two waves stacked along M
#pragma unroll
            for (int w = 0; w < H_WM; ++w) {
                const int wv = w * H_WN + wnc;
                ts += sGn[(wv * WCOLS + cw) * 2];
                tq += sGn[(wv * WCOLS + cw) * 2 + 1];
            }
            const int chunk = (m0 - b * HW) >> 7;
            float* dst = p.gnpart + (((long)b * p.gn_nchunk + chunk) * p.gn_ld + n0 + c) * 2;
            dst[0] = ts;
            dst[1] = tq;
        }
    }
}

}  // namespace

// 1 if qd_conv3x3_halo_i8 covers this descriptor (same results as qd_conv2d_i8, different data path), else 0.
extern "C" int qd_conv3x3_halo_ok(const qd_conv_desc* d) {
    if (!d || !d->x || !d->w || !d->out) return 0;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad_t != 1 || d->pad_l != 1 || d->H != d->Ho || d->W != d->Wo) return 0;
    if (d->nseg != 1 || d->wbits != 4 || !d->w_tiled || d->epilogue != QD_EPI_LINEAR || d->out_dtype != QD_F32) return 0;
    if (!(d->W == 16 || d->W == 32 || d->W == 64) || (d->H * d->W) % 128 != 0 || d->H % (128 / d->W) != 0) return 0;
    if (d->Cout % 320 != 0 || d->seg[0].clen % 16 != 0 || d->seg[0].c0 % 16 != 0) return 0;
    if (d->ldx % 16 != 0 || !qd_aligned(d->x, 16) || !qd_aligned(d->w, 16)) return 0;
    if (d->ldo % 4 != 0 || !qd_aligned(d->out, 16)) return 0;
    if (d->residual && (d->ldr % 4 != 0 || !qd_aligned(d->residual, 16))) return 0;
    if (d->rowbias && (d->ld_rowbias % 4 != 0 || !qd_aligned(d->rowbias, 16))) return 0;
    if (d->gn_part && (d->Ho * d->Wo) % 128 != 0) return 0;
    if ((long)9 * d->seg[0].clen >= 32768) return 0;         // 24-bit zero-point multiply, as qd_conv2d_i8
    if (d->upsample2x && (d->H % 2 || d->W % 2)) return 0;
    return 1;
}

extern "C" int qd_conv3x3_halo_i8(const qd_conv_desc* d, void* stream) {
    QD_REQUIRE(qd_conv3x3_halo_ok(d), "qd_conv3x3_halo_i8: descriptor not covered (3x3 / stride 1 / pad 1, one int4 tile-ordered segment, "
                                      "W in {16,32,64}, H*W %% 128 == 0, Cout %% 320 == 0, fp32 linear epilogue)");
    const qd_conv_seg& g = d->seg[0];
    QD_REQUIRE(g.scale != nullptr, "qd_conv3x3_halo_i8: no scale vector");
    HaloD k{};
    k.x = d->x; k.wt = d->w; k.out = reinterpret_cast<float*>(d->out);
    k.bias = d->bias; k.rowbias = d->rowbias; k.residual = reinterpret_cast<const float*>(d->residual);
    k.ldx = d->ldx; k.ldo = d->ldo; k.ldr = d->ldr; k.ldrb = d->ld_rowbias;
    k.ups = d->upsample2x ? 1 : 0;
    k.B = d->B; k.H = d->H; k.W = d->W; k.lw = d->W == 16 ? 4 : (d->W == 32 ? 5 : 6); k.Cout = d->Cout;
    k.M = d->B * d->H * d->W;
    k.c0 = g.c0; k.clen = g.clen; k.kstep0 = g.kstep0; k.nst = (g.clen + 63) / 64;
    k.ntiles = (d->Cout + 31) / 32; k.nblk_n = d->Cout / H_BN;
    k.scale = g.scale; k.zc = g.zc; k.zw = g.zw; k.zfill = g.zfill; k.fill16 = g.fill16;
    if (d->gn_part) {
        k.gnpart = d->gn_part;
        k.gn_nchunk = d->Ho * d->Wo / 128;
        k.gn_ld = d->gn_ld ? (long)d->gn_ld : (long)d->Cout;
    }
    const int nblk = (k.M / H_BM) * k.nblk_n;
    hipLaunchKernelGGL(igemm_halo_kernel, dim3((unsigned)nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), k);
    QD_LAUNCH_CHECK("qd_conv3x3_halo_i8");
    return 0;
}
