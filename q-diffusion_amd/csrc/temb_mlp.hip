// temb_mlp.hip — K6: the timestep-embedding MLPs as ONE launch per dependency level.
//
// Replaces (reference ldm/modules/diffusionmodules/openaimodel.py:758-759 `time_embed` = Linear -> SiLU -> Linear, and
// qdiff/quant_block.py:98-107 / ddim diffusion.py:127 the per-ResBlock `emb_layers` = SiLU -> Linear on the SAME
// embedding): in the reference every one of those Linears is a QuantModule — its own activation quantiser on
// SiLU(emb), 4/8-bit weights — evaluated on M = batch rows.  As generic GEMM launches they are ~70 tiny dependent
// launches per UNet evaluation (SiLU, row quantiser, split-K contraction, split-K finalise for each of 22 ResBlocks + 2)
// that cost ~0.7 ms of a 24 ms SD evaluation while moving 13 MB.  Here one launch evaluates L Linears that share one
// input: block = (layer, 64 output channels); it quantises SiLU(x) with THAT layer's activation quantiser into LDS
// (exact torch.round(x / delta) codes), contracts against the layer's MFMA-tile-ordered weights (the same packed array
// the generic kernel reads: 32 consecutive channels of one 16-wide K chunk are 256 / 512 contiguous bytes, so lanes =
// channels gives coalesced loads) with v_dot4 (M is 16..64 rows: the matrix pipe has nothing to amortise), and applies the
// generic kernel's epilogue — I = acc - z'*Wsum - zw*(Asum - K*z'), out = float(I)*scale + bias — in the same float
// order, so the results equal the generic path's bit for bit whenever the SiLU inputs do.
// Bound: latency (the launch is ~10 us; 13 MB of weights would take 3 us at HBM speed).
#include "common.h"

namespace {

struct TembLayer {
    const uint8_t* w;        // tile-ordered weights (qd_pack_weights_t4 / _t8), K-step 0 of the layer's only segment
    const float*   scale;    // [Cout] delta_x * delta_w[n]
    const int*     zc;       // [Cout] z' * Wsum[n] or NULL
    const int*     zw;       // [Cout] weight zero point of the stored operand or NULL
    const float*   bias;     // [Cout] or NULL
    const float*   qp;       // float[4] activation quantiser (qd_make_qparams)
    const int*     zfill;    // {z', K*z'} or NULL
    int Cout, out_off;
};

// Round 6: grid = (blocks, row groups).  The (layer, 64-channel) blocks of the 22 embedding projections number 40 .. 80 — a
// quarter of the CUs, one wave per SIMD each, every block quantising ALL rows for itself (SiLU + the exact division: the bulk
// of its time).  blockIdx.y now selects a slice of `rpb` rows: the same work per row, spread over the idle CUs (rows are
// independent: same bytes).
template <int WB, int RB>   // RB = rows per pass (accumulators per thread)
__global__ __launch_bounds__(256) void temb_kernel(const float* __restrict__ x, long ldx, int Btot, int rpb, int K, int silu,
                                                   const TembLayer* __restrict__ layers, const int2* __restrict__ blocks,
                                                   float qmin, float qmax, int off, float* __restrict__ out, long ldo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int b_first = blockIdx.y * rpb;
    const int B = min(rpb, Btot - b_first);
    if (B <= 0) return;
    x += (long)b_first * ldx;
    out += (long)b_first * ldo;
    int8_t* codes = reinterpret_cast<int8_t*>(sm);                       // [B][K]
    int* sAsum = reinterpret_cast<int*>(sm + (size_t)B * K);              // [B]
    int* sRed = sAsum + ((B + 3) & ~3);                                   // [4 parts][RB][64]
    const int2 bl = blocks[blockIdx.x];
    const TembLayer L = layers[bl.x];
    const int n0 = bl.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int part = __builtin_amdgcn_readfirstlane(tid >> 6);           // wave = K partition
    const QP q = qd_load_qp(L.qp);

    // ---- 1. SiLU -> this layer's activation quantiser -> int8 rows in LDS, row sums ---------------------------------
    for (int b = tid; b < B; b += 256) sAsum[b] = 0;                     // B may exceed the block (K = 128: up to 352 rows per launch)
    __syncthreads();
    const int k4n = K >> 2;
    auto quant = [&](auto ft) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(ft)::value;
    for (int idx = tid; idx < B * k4n; idx += 256) {
        const int b = idx / k4n, k4 = idx - b * k4n;
        const float4 v = *reinterpret_cast<const float4*>(x + (long)b * ldx + k4 * 4);
        float e[4] = {v.x, v.y, v.z, v.w};
        unsigned u = 0;
        int s = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y = e[j];
            if (silu) y = y / (1.0f + expf(-y));                   // torch's silu: x / (1 + exp(-x))
            const int c = qd_code_t<FAST>(y, q, qmin, qmax) - off;
            s += c;
            u |= (unsigned)(c & 0xff) << (8 * j);
        }
        *reinterpret_cast<unsigned*>(codes + (long)b * K + k4 * 4) = u;
        atomicAdd(&sAsum[b], s);
    }
    };
    QD_FAST_DISPATCH(q.fast, quant);
    __syncthreads();

    // ---- 2. contraction: lane = output channel, wave = every 4th 16-wide K chunk ---------------------------------------
    const int n = n0 + lane;
    const int ntiles = (L.Cout + 31) >> 5;
    const bool nin = (n >> 5) < ntiles;                                   // the tile exists in the packed array
    constexpr int UB = WB * 2;                                            // bytes of one (channel, 16-K chunk) unit
    const uint8_t* wn = L.w + ((long)(n >> 5) * 4 * 32 + (n & 31)) * UB;  // + (kstep*ntiles*4 + chunk%4) * 32 * UB
    const int nchunk = K >> 4;
    const int kz = L.zfill ? L.zfill[1] : 0;
    float sc = 0.f, bias_n = 0.f;
    int zc_n = 0, zw_n = 0;
    if (n < L.Cout) {
        sc = L.scale[n];
        if (L.zc) zc_n = L.zc[n];
        if (L.zw) zw_n = L.zw[n];
        if (L.bias) bias_n = L.bias[n];
    }
    for (int b0 = 0; b0 < B; b0 += RB) {
        int acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = 0;
        for (int c = part; c < nchunk; c += 4) {
            v4i wv = {0, 0, 0, 0};
            if (nin) {
                const uint8_t* src = wn + ((long)(c >> 2) * ntiles * 4 + (c & 3)) * 32 * UB;
                if constexpr (WB == 4) {
                    const uint2 pk = *reinterpret_cast<const uint2*>(src);
                    wv = v4i{(int)(pk.x & 0x0F0F0F0Fu), (int)((pk.x >> 4) & 0x0F0F0F0Fu),
                             (int)(pk.y & 0x0F0F0F0Fu), (int)((pk.y >> 4) & 0x0F0F0F0Fu)};
                } else {
                    wv = *reinterpret_cast<const v4i*>(src);
                }
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                if (b0 + r < B) {                                          // wave-uniform
                    const v4i a = *reinterpret_cast<const v4i*>(codes + (long)(b0 + r) * K + c * 16);   // broadcast read
                    int s = acc[r];
                    s = __builtin_amdgcn_sdot4(a.x, wv.x, s, false);
                    s = __builtin_amdgcn_sdot4(a.y, wv.y, s, false);
                    s = __builtin_amdgcn_sdot4(a.z, wv.z, s, false);
                    acc[r] = __builtin_amdgcn_sdot4(a.w, wv.w, s, false);
                }
            }
        }
        // ---- 3. combine the four K partitions, epilogue ---------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < RB; ++r) sRed[(part * RB + r) * 64 + lane] = acc[r];
        __syncthreads();
        for (int r = part; r < RB; r += 4) {                               // wave w finishes rows w, w+4, ...
            const int b = b0 + r;
            if (b < B && n < L.Cout) {
                const int a = sRed[r * 64 + lane] + sRed[(RB + r) * 64 + lane] + sRed[(2 * RB + r) * 64 + lane] + sRed[(3 * RB + r) * 64 + lane];
                const int I = a - zc_n - zw_n * (sAsum[b] - kz);
                out[(long)b * ldo + L.out_off + n] = (float)I * sc + bias_n;
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int qd_temb_mlp(const float* x, int64_t ldx, int B, int K, int apply_silu, const void* layers, int n_layers,
                           const int32_t* blocks, int n_blocks, int wbits, int qmin, int qmax, int off, float* out,
                           int64_t ldo, void* stream) {
    QD_REQUIRE(x && layers && blocks && out, "qd_temb_mlp: null pointer");
    QD_REQUIRE(B > 0 && K > 0 && K % 16 == 0 && n_layers > 0 && n_blocks > 0, "qd_temb_mlp: bad shape (K must be a multiple of 16, got %d)", K);
    QD_REQUIRE(wbits == 4 || wbits == 8, "qd_temb_mlp: wbits must be 4 or 8");
    QD_REQUIRE(ldx % 4 == 0 && qd_aligned(x, 16), "qd_temb_mlp: x rows must be 16-byte aligned");
    constexpr int RB = 16;
    // row groups: enough that the launch has about two blocks per CU, at least 4 rows each
    int groups = (512 + n_blocks - 1) / n_blocks;
    if (groups > (B + 3) / 4) groups = (B + 3) / 4;
    if (groups < 1) groups = 1;
    const int rpb = (B + groups - 1) / groups;
    groups = (B + rpb - 1) / rpb;
    const size_t lds = (size_t)rpb * K + (size_t)((rpb + 3) & ~3) * 4 + 4 * RB * 64 * 4;
    QD_REQUIRE(lds <= 64 * 1024, "qd_temb_mlp: %d rows x K = %d do not fit LDS (call it on at most %d rows at a time)", rpb, K, (int)((64 * 1024 - 20000) / K));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const auto* L = reinterpret_cast<const TembLayer*>(layers);
    const auto* bl = reinterpret_cast<const int2*>(blocks);
    if (wbits == 4)
        hipLaunchKernelGGL((temb_kernel<4, RB>), dim3(n_blocks, groups), dim3(256), lds, st, x, (long)ldx, B, rpb, K, apply_silu, L, bl, (float)qmin, (float)qmax, off, out, (long)ldo);
    else
        hipLaunchKernelGGL((temb_kernel<8, RB>), dim3(n_blocks, groups), dim3(256), lds, st, x, (long)ldx, B, rpb, K, apply_silu, L, bl, (float)qmin, (float)qmax, off, out, (long)ldo);
    QD_LAUNCH_CHECK("qd_temb_mlp");
    return 0;
}
