// norm_quant.hip — K5 GroupNorm(+SiLU)->quant and K9a LayerNorm->quant producers.
//
// The reference runs GroupNorm32 (fp32), SiLU and the next module's act quantiser as ~11 separate
// elementwise passes over an fp32 tensor (ddim/models/diffusion.py:121-130, openaimodel.py:201-232,
// quant_layer.py:82-88).  Here: one statistics pass (deterministic two-level reduction, no float
// atomics) and one apply pass that reads the activation once and writes one byte per element.
#include "common.h"

namespace {

// rows per partial-sum block: 32, or 8 when the map is small (S <= 256) so that the statistics pass still
// launches >= 2 blocks per CU at batch 16 (measured: 33.9 -> 28.1 us at 16x16x1280; 8 rows at S=1024 was slower)
static inline __host__ __device__ int gn_rows(long S) { return S > 256 ? 32 : 8; }

// partial sums: grid (nchunk, B); thread owns 4 consecutive channels (one float4 load per row), loops over rows.
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, long S, int C, long ldx,
                                                         float* __restrict__ part, int nchunk, int vec) {
    const int chunk = blockIdx.x;
    const long b = blockIdx.y;
    const int rows = gn_rows(S);
    const long r0 = (long)chunk * rows;
    const long r1 = (r0 + rows < S) ? r0 + rows : S;
    for (int c4 = threadIdx.x; c4 < C / 4; c4 += 256) {
        float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
        const T* p = x + (b * S + r0) * ldx + c4 * 4;
#pragma unroll 4
        for (long r = r0; r < r1; ++r, p += ldx) {
            float v[4];
            qd_ld4(p, vec != 0, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[j] += v[j];
                q[j] += v[j] * v[j];
            }
        }
        float* dst = part + (((b * nchunk + chunk) * (long)C) + c4 * 4) * 2;
        *reinterpret_cast<float4*>(dst)     = make_float4(s[0], q[0], s[1], q[1]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(s[2], q[2], s[3], q[3]);
    }
}

// fp16 stream: thread owns 8 consecutive channels (one 16-byte load per row).  Every channel's sum runs over the same rows in
// the same order as in the 4-channel form: bit-identical partials.
__global__ __launch_bounds__(256) void gn_partial_h8_kernel(const __half* __restrict__ x, long S, int C, long ldx,
                                                            float* __restrict__ part, int nchunk) {
    const int chunk = blockIdx.x;
    const long b = blockIdx.y;
    const int rows = gn_rows(S);
    const long r0 = (long)chunk * rows;
    const long r1 = (r0 + rows < S) ? r0 + rows : S;
    for (int c8 = threadIdx.x; c8 < C / 8; c8 += 256) {
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const __half* p = x + (b * S + r0) * ldx + c8 * 8;
#pragma unroll 4
        for (long r = r0; r < r1; ++r, p += ldx) {
            float v[8];
            qd_ld8h(p, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] += v[j];
                q[j] += v[j] * v[j];
            }
        }
        float* dst = part + (((b * nchunk + chunk) * (long)C) + c8 * 8) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(dst + 4 * j) = make_float4(s[2 * j], q[2 * j], s[2 * j + 1], q[2 * j + 1]);
    }
}

// finalize: grid (groups, B), 64 threads.  Writes per-(b,c) affine a = rstd*gamma, sh = beta - mean*a.
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ part, int nchunk, long ldp, long S, int C,
                                                         int groups, float eps, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ ab) {
    const int g = blockIdx.x;
    const long b = blockIdx.y;
    const int cpg = C / groups;
    double s = 0.0, q = 0.0;
    const int items = nchunk * cpg;
    for (int i = threadIdx.x; i < items; i += 64) {
        int chunk = i / cpg, c = g * cpg + i % cpg;
        const float* p = part + (((b * nchunk + chunk) * ldp) + c) * 2;
        s += (double)p[0];
        q += (double)p[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
    }
    const double n = (double)S * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    for (int i = threadIdx.x; i < cpg; i += 64) {
        int c = g * cpg + i;
        float a = rstd * (gamma ? gamma[c] : 1.f);
        float sh = (beta ? beta[c] : 0.f) - fmean * a;
        ab[(b * C + c) * 2] = a;
        ab[(b * C + c) * 2 + 1] = sh;
    }
}

// finalize for long chunk lists (first-stage decoder: 512 x 512 maps = 2048 chunks of 128 rows, 4-16 channels per group —
// the 64-thread kernel above walks 128-512 strided loads per thread there, 30-130 us): 256 threads, same fp64 sums in a
// fixed order (thread-strided partial sums, wave butterflies, then the four wave results in wave order).
__global__ __launch_bounds__(256) void gn_finalize_wide_kernel(const float* __restrict__ part, int nchunk, long ldp, long S, int C,
                                                               int groups, float eps, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ ab) {
    __shared__ double red[8];
    const int g = blockIdx.x;
    const long b = blockIdx.y;
    const int cpg = C / groups;
    double s = 0.0, q = 0.0;
    const int items = nchunk * cpg;
    for (int i = threadIdx.x; i < items; i += 256) {
        int chunk = i / cpg, c = g * cpg + i % cpg;
        const float* p = part + (((b * nchunk + chunk) * ldp) + c) * 2;
        s += (double)p[0];
        q += (double)p[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; red[4 + (threadIdx.x >> 6)] = q; }
    __syncthreads();
    s = ((red[0] + red[1]) + red[2]) + red[3];
    q = ((red[4] + red[5]) + red[6]) + red[7];
    const double n = (double)S * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    for (int i = threadIdx.x; i < cpg; i += 256) {
        int c = g * cpg + i;
        float a = rstd * (gamma ? gamma[c] : 1.f);
        float sh = (beta ? beta[c] : 0.f) - fmean * a;
        ab[(b * C + c) * 2] = a;
        ab[(b * C + c) * 2 + 1] = sh;
    }
}

// use_scale_shift_norm residual blocks (reference quant_block.py:99-103: `out_norm(h) * (1 + scale) + shift`, scale | shift =
// the two halves of the block's embedding projection, one row per sample): the modulation is folded into the per-(sample,
// channel) affine the apply pass already uses — a' = a (1 + scale), sh' = sh (1 + scale) + shift — so the apply pass and its
// one read of the tensor stay as they are.  mod: [B][>= 2C] fp32 rows.
__global__ __launch_bounds__(256) void gn_modulate_kernel(float* __restrict__ ab, const float* __restrict__ mod, long ldm, int C,
                                                          long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / C;
    const int c = (int)(i - b * C);
    const float sc = 1.0f + mod[b * ldm + c], sf = mod[b * ldm + C + c];
    const float a = ab[2 * i], sh = ab[2 * i + 1];
    ab[2 * i] = a * sc;
    ab[2 * i + 1] = sh * sc + sf;
}

// second, optional output of the apply pass: the RAW input quantised for another consumer of the same tensor — the 1x1
// skip connection of a residual block (reference quant_block.py:108-111: `skip_connection(x, split)` reads the very
// tensor `in_layers` normalises; up to two channel segments with their own activation quantisers, quant_layer.py:257-269).
// One read of x feeds both consumers instead of a second pass (qd_quantize_act) over the largest tensors of the UNet.
struct RawQ {
    int8_t* out;
    long    ldo;
    int     nseg;
    int     c0[2], clen[2], oc0[2], off[2];
    float   qmin[2], qmax[2];
    const float* qp[2];
};

// apply: lane = 4 consecutive channels of one row (float4 in, 4 bytes out; both sides fully coalesced).
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, long rows, long S, int C, long ldx,
                                                       const float* __restrict__ ab, int apply_silu,
                                                       const float* __restrict__ qp, float qmin, float qmax, int off,
                                                       int8_t* __restrict__ out, long ldo, float* __restrict__ yout,
                                                       long ldy, int vec, const RawQ raw) {
    const int chunks = C >> 2;
    long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= rows * chunks) return;
    long row = gid / chunks;
    int c = (int)(gid - row * chunks) * 4;
    long b = row / S;
    float v[4];
    qd_ld4(x + row * ldx + c, vec != 0, v);
    const float4 ab0 = *reinterpret_cast<const float4*>(ab + (b * C + c) * 2);
    const float4 ab1 = *reinterpret_cast<const float4*>(ab + (b * C + c) * 2 + 4);
    const float a4[4] = {ab0.x, ab0.z, ab1.x, ab1.z}, s4[4] = {ab0.y, ab0.w, ab1.y, ab1.w};
    QP q{1.f, 0.f, 1.f, false};
    if (out) q = qd_load_qp(qp);
    unsigned u = 0;
    auto body = [&](auto ft) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y = v[j] * a4[j] + s4[j];
            if (apply_silu) y = y * (1.0f / (1.0f + expf(-y)));
            if (yout) yout[row * ldy + c + j] = y;
            if (out) u |= (unsigned)((qd_code_t<decltype(ft)::value>(y, q, qmin, qmax) - off) & 0xff) << (8 * j);
        }
    };
    QD_FAST_DISPATCH(q.fast, body);
    if (out) *reinterpret_cast<unsigned*>(out + row * ldo + c) = u;
    if (raw.out) {
        // segment boundaries are multiples of 16 channels (checked by the host): a lane's 4 channels share a segment
        const bool s1 = raw.nseg > 1 && c >= raw.c0[1];           // explicit selects: no dynamically indexed kernel argument
        const int rc0 = s1 ? raw.c0[1] : raw.c0[0], rlen = s1 ? raw.clen[1] : raw.clen[0], roc0 = s1 ? raw.oc0[1] : raw.oc0[0];
        if (c >= rc0 && c < rc0 + rlen) {
            const QP rq = qd_load_qp(s1 ? raw.qp[1] : raw.qp[0]);
            const float rmin = s1 ? raw.qmin[1] : raw.qmin[0], rmax = s1 ? raw.qmax[1] : raw.qmax[0];
            const int roff = s1 ? raw.off[1] : raw.off[0];
            unsigned w = 0;
            auto rbody = [&](auto ft) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    w |= (unsigned)((qd_code_t<decltype(ft)::value>(v[j], rq, rmin, rmax) - roff) & 0xff) << (8 * j);
            };
            QD_FAST_DISPATCH(rq.fast, rbody);
            *reinterpret_cast<unsigned*>(raw.out + row * raw.ldo + roc0 + (c - rc0)) = w;
        }
    }
}

// The apply pass with U rows per thread (default U = 2 since round 3: -0.15 ms per SD evaluation, profiles/r03_first_call_ab.md) — the U 16-byte loads of a thread
// are issued back to back before any of them is used (more bytes in flight per wave: the one-row kernel streams at
// 3.8 TB/s of the ~5.5 the HBM sustains), the per-(sample, channel) affine is fetched once per U rows.  fp32 input, 16-byte
// aligned rows, no float output; U consecutive rows always belong to one sample (S % U == 0).  Same arithmetic per element.
template <int U>
__global__ __launch_bounds__(256) void gn_apply_rows_kernel(const float* __restrict__ x, long rows, long S, int C, long ldx,
                                                            const float* __restrict__ ab, int apply_silu,
                                                            const float* __restrict__ qp, float qmin, float qmax, int off,
                                                            int8_t* __restrict__ out, long ldo, const RawQ raw) {
    const int chunks = C >> 2;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (rows / U) * chunks) return;
    const long rg = gid / chunks;
    const int c = (int)(gid - rg * chunks) * 4;
    const long row0 = rg * U;
    const long b = row0 / S;
    float4 xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u] = *reinterpret_cast<const float4*>(x + (row0 + u) * ldx + c);
    const float4 ab0 = *reinterpret_cast<const float4*>(ab + (b * C + c) * 2);
    const float4 ab1 = *reinterpret_cast<const float4*>(ab + (b * C + c) * 2 + 4);
    const float a4[4] = {ab0.x, ab0.z, ab1.x, ab1.z}, s4[4] = {ab0.y, ab0.w, ab1.y, ab1.w};
    const QP q = qd_load_qp(qp);
    const bool s1 = raw.out && raw.nseg > 1 && c >= raw.c0[1];
    const int rc0 = s1 ? raw.c0[1] : raw.c0[0], rlen = s1 ? raw.clen[1] : raw.clen[0], roc0 = s1 ? raw.oc0[1] : raw.oc0[0];
    const bool rawhere = raw.out && c >= rc0 && c < rc0 + rlen;
    QP rq{1.f, 0.f, 1.f, false};
    if (rawhere) rq = qd_load_qp(s1 ? raw.qp[1] : raw.qp[0]);
    const float rmin = s1 ? raw.qmin[1] : raw.qmin[0], rmax = s1 ? raw.qmax[1] : raw.qmax[0];
    const int roff = s1 ? raw.off[1] : raw.off[0];
    const QB qb = qd_bytes_setup(q, qmin, qmax, off), rqb = qd_bytes_setup(rq, rmin, rmax, roff);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float v[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
        const long row = row0 + u;
        unsigned w0 = 0;
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[j] = v[j] * a4[j] + s4[j];
            if (apply_silu) y[j] = y[j] * (1.0f / (1.0f + expf(-y[j])));
        }
        auto body = [&](auto ft) __attribute__((always_inline)) { w0 = qd_pack4_t<decltype(ft)::value>(y[0], y[1], y[2], y[3], q, qb); };
        QD_FAST_DISPATCH(q.fast, body);
        *reinterpret_cast<unsigned*>(out + row * ldo + c) = w0;
        if (rawhere) {
            unsigned w = 0;
            auto rbody = [&](auto ft) __attribute__((always_inline)) { w = qd_pack4_t<decltype(ft)::value>(v[0], v[1], v[2], v[3], rq, rqb); };
            QD_FAST_DISPATCH(rq.fast, rbody);
            *reinterpret_cast<unsigned*>(raw.out + row * raw.ldo + roc0 + (c - rc0)) = w;
        }
    }
}

// (Round 6 measured the finalise pass FOLDED into this pass — a block owning a 32- / 64-channel window of up to 512 rows, one wave
// per overlapped group repeating gn_finalize_kernel's reduction in its prologue: bit-identical codes, 61 launches fewer per SD
// evaluation — and deleted it: 18.64 vs 18.63 ms per SD step, 3.39 vs 3.42 ms CIFAR, 16.24 vs 15.83 ms LDM-4
// (profiles/r06_c5_gn_fuse_ab.txt).  The windowed pass streams 128- / 256-byte row segments behind a dependent prologue and
// loses what the removed launches return; what would pay is group-level partials from the producing epilogue.)
// The same pass for the fp16 activation stream: thread = 8 consecutive channels (ONE 16-byte load per row — the 4-channel form
// reads 8 bytes per lane: half lines, the same request count as the fp32 stream) of U rows, 8 code bytes out per row.  Same
// arithmetic per element, so the codes are those of gn_apply_kernel<__half> bit for bit.
template <int U>
__global__ __launch_bounds__(256) void gn_apply_rows_h8_kernel(const __half* __restrict__ x, long rows, long S, int C, long ldx,
                                                               const float* __restrict__ ab, int apply_silu,
                                                               const float* __restrict__ qp, float qmin, float qmax, int off,
                                                               int8_t* __restrict__ out, long ldo, const RawQ raw) {
    const int chunks = C >> 3;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (rows / U) * chunks) return;
    const long rg = gid / chunks;
    const int c = (int)(gid - rg * chunks) * 8;
    const long row0 = rg * U;
    const long b = row0 / S;
    v4i xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u] = *reinterpret_cast<const v4i*>(x + (row0 + u) * ldx + c);
    float a8[8], s8[8];
    const float* abp = ab + (b * C + c) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(abp + 4 * j);
        a8[2 * j] = t.x; s8[2 * j] = t.y; a8[2 * j + 1] = t.z; s8[2 * j + 1] = t.w;
    }
    const QP q = qd_load_qp(qp);
    // segment bounds are multiples of 16 channels (host): a lane's 8 channels share a segment
    const bool s1 = raw.out && raw.nseg > 1 && c >= raw.c0[1];
    const int rc0 = s1 ? raw.c0[1] : raw.c0[0], rlen = s1 ? raw.clen[1] : raw.clen[0], roc0 = s1 ? raw.oc0[1] : raw.oc0[0];
    const bool rawhere = raw.out && c >= rc0 && c < rc0 + rlen;
    QP rq{1.f, 0.f, 1.f, false};
    if (rawhere) rq = qd_load_qp(s1 ? raw.qp[1] : raw.qp[0]);
    const float rmin = s1 ? raw.qmin[1] : raw.qmin[0], rmax = s1 ? raw.qmax[1] : raw.qmax[0];
    const int roff = s1 ? raw.off[1] : raw.off[0];
    const QB qb = qd_bytes_setup(q, qmin, qmax, off), rqb = qd_bytes_setup(rq, rmin, rmax, roff);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        v4f lo, hi;
        qd_h8_to_f(xv[u], lo, hi);
        const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const long row = row0 + u;
        unsigned w0 = 0, w1 = 0;
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            y[j] = v[j] * a8[j] + s8[j];
            if (apply_silu) y[j] = y[j] * (1.0f / (1.0f + expf(-y[j])));
        }
        auto body = [&](auto ft) __attribute__((always_inline)) {
            w0 = qd_pack4_t<decltype(ft)::value>(y[0], y[1], y[2], y[3], q, qb);
            w1 = qd_pack4_t<decltype(ft)::value>(y[4], y[5], y[6], y[7], q, qb);
        };
        QD_FAST_DISPATCH(q.fast, body);
        *reinterpret_cast<uint2*>(out + row * ldo + c) = make_uint2(w0, w1);
        if (rawhere) {
            unsigned r0 = 0, r1 = 0;
            auto rbody = [&](auto ft) __attribute__((always_inline)) {
                r0 = qd_pack4_t<decltype(ft)::value>(v[0], v[1], v[2], v[3], rq, rqb);
                r1 = qd_pack4_t<decltype(ft)::value>(v[4], v[5], v[6], v[7], rq, rqb);
            };
            QD_FAST_DISPATCH(rq.fast, rbody);
            *reinterpret_cast<uint2*>(raw.out + row * raw.ldo + roc0 + (c - rc0)) = make_uint2(r0, r1);
        }
    }
}

// LayerNorm: one wave per row, row held in registers (C <= 64*4*MAXV).
constexpr int LN_MAXV = 6;  // up to 1536 channels
// NV = float4 chunks per lane (ceil(C/256)), RPW = rows per wave: the loads of RPW rows are issued back to
// back and their reduction chains interleave (a one-row-per-wave version ran at 1.8 TB/s, latency-bound on
// the load -> two dependent butterfly reductions -> store chain); gamma/beta are fetched once per wave.
template <typename T, int NV, int RPW>
__global__ __launch_bounds__(256) void ln_quant_kernel(const T* __restrict__ x, long M, int C, long ldx, float eps,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int nout, const float* qp0, const float* qp1, const float* qp2,
                                                       float3 qmin, float3 qmax, int3 off, int8_t* o0, int8_t* o1,
                                                       int8_t* o2, long ldo, int vec) {
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= M) return;
    const int nv = C >> 2;  // float4 count
    float v[RPW][NV][4];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long row = row0 + r < M ? row0 + r : M - 1;          // clamp: loads stay in bounds, stores are predicated
        const T* src = x + row * ldx;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int idx = lane + 64 * k;
            if (idx < nv) qd_ld4(src + idx * 4, vec != 0, v[r][k]);
        }
    }
    float g[NV][4], bt[NV][4];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int idx = lane + 64 * k;
        if (idx < nv) {
            const float4 a = *reinterpret_cast<const float4*>(gamma + idx * 4), b = *reinterpret_cast<const float4*>(beta + idx * 4);
            g[k][0] = a.x; g[k][1] = a.y; g[k][2] = a.z; g[k][3] = a.w;
            bt[k][0] = b.x; bt[k][1] = b.y; bt[k][2] = b.z; bt[k][3] = b.w;
        }
    }
    float s[RPW], q[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        s[r] = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (lane + 64 * k < nv) {
#pragma unroll
                for (int j = 0; j < 4; ++j) s[r] += v[r][k][j];
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < RPW; ++r) s[r] += __shfl_xor(s[r], o);
    float mean[RPW], rstd[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        mean[r] = s[r] / (float)C;
        q[r] = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (lane + 64 * k < nv) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = v[r][k][j] - mean[r]; q[r] += d * d; }
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < RPW; ++r) q[r] += __shfl_xor(q[r], o);
    const QP qa = qd_load_qp(qp0);
    const QP qb = nout > 1 ? qd_load_qp(qp1) : QP{1.f, 0.f, 1.f, false};
    const QP qc = nout > 2 ? qd_load_qp(qp2) : QP{1.f, 0.f, 1.f, false};
    const QB ba = qd_bytes_setup(qa, qmin.x, qmax.x, off.x), bb = qd_bytes_setup(qb, qmin.y, qmax.y, off.y), bc = qd_bytes_setup(qc, qmin.z, qmax.z, off.z);
    auto lnbody = [&](auto ft) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(ft)::value;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        rstd[r] = 1.0f / sqrtf(q[r] / (float)C + eps);
        const long row = row0 + r;
        if (row >= M) continue;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int idx = lane + 64 * k;
            if (idx < nv) {
                unsigned u0 = 0, u1 = 0, u2 = 0;
                float y[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = (v[r][k][j] - mean[r]) * rstd[r] * g[k][j] + bt[k][j];
                u0 = qd_pack4_t<FAST>(y[0], y[1], y[2], y[3], qa, ba);
                if (nout > 1) u1 = qd_pack4_t<FAST>(y[0], y[1], y[2], y[3], qb, bb);
                if (nout > 2) u2 = qd_pack4_t<FAST>(y[0], y[1], y[2], y[3], qc, bc);
                *reinterpret_cast<unsigned*>(o0 + row * ldo + idx * 4) = u0;
                if (nout > 1) *reinterpret_cast<unsigned*>(o1 + row * ldo + idx * 4) = u1;
                if (nout > 2) *reinterpret_cast<unsigned*>(o2 + row * ldo + idx * 4) = u2;
            }
        }
    }
    };
    QD_FAST_DISPATCH(qa.fast && (nout < 2 || qb.fast) && (nout < 3 || qc.fast), lnbody);
}

// C = 32 * J (J <= 10: the 320-channel level of SD): EIGHT lanes per row, lane c owns columns j*32 + c*4 .. +3 of every
// 32-column group j.  A row is reduced as 4*J values per lane in (group, column) order, then a butterfly over the eight lanes
// (xor 1, 2, 4), two passes, explicit fma.  A wave handles eight rows; per group the eight lanes of a row read 128 (fp32) /
// 64 (fp16) contiguous bytes.  (Round 5 ran the same reduction inside the producing GEMM's epilogue: bit-identical, no
// faster — the pass is VALU-bound, not bound by its re-read — and deleted in round 6, profiles/r05_ln_fuse_ab.md.)
template <typename T, int J>
__global__ __launch_bounds__(256) void ln_quant_rows8_kernel(const T* __restrict__ x, long M, long ldx, float eps,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int nout, const float* qp0, const float* qp1, const float* qp2,
                                                             float3 qmin, float3 qmax, int3 off, int8_t* o0, int8_t* o1,
                                                             int8_t* o2, long ldo) {
#pragma clang fp contract(off)
    constexpr int C = 32 * J;
    const int lane = threadIdx.x & 63, c4 = (lane & 7) * 4;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (lane >> 3);
    const bool ok = row < M;
    const T* src = x + (ok ? row : M - 1) * ldx + c4;
    float v[J][4];
#pragma unroll
    for (int j = 0; j < J; ++j) qd_ld4(src + j * 32, true, v[j]);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += v[j][e];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[j][e] - mean;
            q = __builtin_fmaf(d, d, q);
        }
    q += __shfl_xor(q, 1);
    q += __shfl_xor(q, 2);
    q += __shfl_xor(q, 4);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    const QP qa = qd_load_qp(qp0);
    const QP qb = nout > 1 ? qd_load_qp(qp1) : QP{1.f, 0.f, 1.f, false};
    const QP qc = nout > 2 ? qd_load_qp(qp2) : QP{1.f, 0.f, 1.f, false};
    const QB ba = qd_bytes_setup(qa, qmin.x, qmax.x, off.x), bb = qd_bytes_setup(qb, qmin.y, qmax.y, off.y), bc = qd_bytes_setup(qc, qmin.z, qmax.z, off.z);
    auto lnbody = [&](auto ft) __attribute__((always_inline)) {
#pragma clang fp contract(off)
        constexpr bool FAST = decltype(ft)::value;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float4 g4 = *reinterpret_cast<const float4*>(gamma + j * 32 + c4), b4 = *reinterpret_cast<const float4*>(beta + j * 32 + c4);
            const float g[4] = {g4.x, g4.y, g4.z, g4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
            unsigned u0 = 0, u1 = 0, u2 = 0;
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float tt = (v[j][e] - mean) * rstd;
                y[e] = __builtin_fmaf(tt, g[e], b[e]);
            }
            u0 = qd_pack4_t<FAST>(y[0], y[1], y[2], y[3], qa, ba);
            if (nout > 1) u1 = qd_pack4_t<FAST>(y[0], y[1], y[2], y[3], qb, bb);
            if (nout > 2) u2 = qd_pack4_t<FAST>(y[0], y[1], y[2], y[3], qc, bc);
            if (ok) {
                *reinterpret_cast<unsigned*>(o0 + row * ldo + j * 32 + c4) = u0;
                if (nout > 1) *reinterpret_cast<unsigned*>(o1 + row * ldo + j * 32 + c4) = u1;
                if (nout > 2) *reinterpret_cast<unsigned*>(o2 + row * ldo + j * 32 + c4) = u2;
            }
        }
    };
    QD_FAST_DISPATCH(qa.fast && (nout < 2 || qb.fast) && (nout < 3 || qc.fast), lnbody);
}

// fp16 stream: lane = 8 consecutive channels per chunk (one 16-byte load; 8 code bytes per output), NV = chunks per lane
// (ceil(C / 512)).  Same two-pass statistics per row (sum -> mean, sum of squared deviations -> rstd) with wave butterflies.
template <int NV, int RPW>
__global__ __launch_bounds__(256) void ln_quant_h8_kernel(const __half* __restrict__ x, long M, int C, long ldx, float eps,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int nout, const float* qp0, const float* qp1, const float* qp2,
                                                          float3 qmin, float3 qmax, int3 off, int8_t* o0, int8_t* o1,
                                                          int8_t* o2, long ldo) {
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= M) return;
    const int nv = C >> 3;
    float v[RPW][NV][8];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long row = row0 + r < M ? row0 + r : M - 1;
        const __half* src = x + row * ldx;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int idx = lane + 64 * k;
            if (idx < nv) qd_ld8h(src + idx * 8, v[r][k]);
        }
    }
    float g[NV][8], bt[NV][8];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int idx = lane + 64 * k;
        if (idx < nv) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 a = *reinterpret_cast<const float4*>(gamma + idx * 8 + 4 * h), b = *reinterpret_cast<const float4*>(beta + idx * 8 + 4 * h);
                g[k][4 * h] = a.x; g[k][4 * h + 1] = a.y; g[k][4 * h + 2] = a.z; g[k][4 * h + 3] = a.w;
                bt[k][4 * h] = b.x; bt[k][4 * h + 1] = b.y; bt[k][4 * h + 2] = b.z; bt[k][4 * h + 3] = b.w;
            }
        }
    }
    float s[RPW], q[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        s[r] = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (lane + 64 * k < nv) {
#pragma unroll
                for (int j = 0; j < 8; ++j) s[r] += v[r][k][j];
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < RPW; ++r) s[r] += __shfl_xor(s[r], o);
    float mean[RPW], rstd[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        mean[r] = s[r] / (float)C;
        q[r] = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (lane + 64 * k < nv) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[r][k][j] - mean[r]; q[r] += d * d; }
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < RPW; ++r) q[r] += __shfl_xor(q[r], o);
    const QP qa = qd_load_qp(qp0);
    const QP qb = nout > 1 ? qd_load_qp(qp1) : QP{1.f, 0.f, 1.f, false};
    const QP qc = nout > 2 ? qd_load_qp(qp2) : QP{1.f, 0.f, 1.f, false};
    const QB ba = qd_bytes_setup(qa, qmin.x, qmax.x, off.x), bb = qd_bytes_setup(qb, qmin.y, qmax.y, off.y), bc = qd_bytes_setup(qc, qmin.z, qmax.z, off.z);
    auto lnbody = [&](auto ft) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(ft)::value;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        rstd[r] = 1.0f / sqrtf(q[r] / (float)C + eps);
        const long row = row0 + r;
        if (row >= M) continue;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int idx = lane + 64 * k;
            if (idx < nv) {
                unsigned u0[2] = {0, 0}, u1[2] = {0, 0}, u2[2] = {0, 0};
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = (v[r][k][j] - mean[r]) * rstd[r] * g[k][j] + bt[k][j];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u0[h] = qd_pack4_t<FAST>(y[4 * h], y[4 * h + 1], y[4 * h + 2], y[4 * h + 3], qa, ba);
                    if (nout > 1) u1[h] = qd_pack4_t<FAST>(y[4 * h], y[4 * h + 1], y[4 * h + 2], y[4 * h + 3], qb, bb);
                    if (nout > 2) u2[h] = qd_pack4_t<FAST>(y[4 * h], y[4 * h + 1], y[4 * h + 2], y[4 * h + 3], qc, bc);
                }
                *reinterpret_cast<uint2*>(o0 + row * ldo + idx * 8) = make_uint2(u0[0], u0[1]);
                if (nout > 1) *reinterpret_cast<uint2*>(o1 + row * ldo + idx * 8) = make_uint2(u1[0], u1[1]);
                if (nout > 2) *reinterpret_cast<uint2*>(o2 + row * ldo + idx * 8) = make_uint2(u2[0], u2[1]);
            }
        }
    }
    };
    QD_FAST_DISPATCH(qa.fast && (nout < 2 || qb.fast) && (nout < 3 || qc.fast), lnbody);
}

template <int NV>
void launch_ln_h8(hipStream_t st, const void* x, long M, int C, long ldx, float eps, const float* gamma, const float* beta, int nout,
                  const float* const* qp, float3 mn, float3 mx, int3 of, int8_t* const* o, long ldo) {
    constexpr int RPW = NV <= 2 ? 2 : 1;
    dim3 grid((unsigned)((M + 4 * RPW - 1) / (4 * RPW)));
    hipLaunchKernelGGL((ln_quant_h8_kernel<NV, RPW>), grid, dim3(256), 0, st, (const __half*)x, M, C, ldx, eps, gamma, beta, nout, qp[0], qp[1],
                       qp[2], mn, mx, of, o[0], o[1], o[2], ldo);
}

template <typename T, int NV>
void launch_ln(hipStream_t st, const void* x, long M, int C, long ldx, float eps, const float* gamma, const float* beta, int nout,
               const float* const* qp, float3 mn, float3 mx, int3 of, int8_t* const* o, long ldo, int vec) {
    constexpr int RPW = NV <= 3 ? 2 : 1;
    dim3 grid((unsigned)((M + 4 * RPW - 1) / (4 * RPW)));
    hipLaunchKernelGGL((ln_quant_kernel<T, NV, RPW>), grid, dim3(256), 0, st, (const T*)x, M, C, ldx, eps, gamma, beta, nout, qp[0], qp[1],
                       qp[2], mn, mx, of, o[0], o[1], o[2], ldo, vec);
}

template <typename T>
void dispatch_ln(int nvl, hipStream_t st, const void* x, long M, int C, long ldx, float eps, const float* gamma, const float* beta,
                 int nout, const float* const* qp, float3 mn, float3 mx, int3 of, int8_t* const* o, long ldo, int vec) {
    switch (nvl) {
        case 1:  launch_ln<T, 1>(st, x, M, C, ldx, eps, gamma, beta, nout, qp, mn, mx, of, o, ldo, vec); break;
        case 2:  launch_ln<T, 2>(st, x, M, C, ldx, eps, gamma, beta, nout, qp, mn, mx, of, o, ldo, vec); break;
        case 3:  launch_ln<T, 3>(st, x, M, C, ldx, eps, gamma, beta, nout, qp, mn, mx, of, o, ldo, vec); break;
        case 4:  launch_ln<T, 4>(st, x, M, C, ldx, eps, gamma, beta, nout, qp, mn, mx, of, o, ldo, vec); break;
        case 5:  launch_ln<T, 5>(st, x, M, C, ldx, eps, gamma, beta, nout, qp, mn, mx, of, o, ldo, vec); break;
        default: launch_ln<T, LN_MAXV>(st, x, M, C, ldx, eps, gamma, beta, nout, qp, mn, mx, of, o, ldo, vec); break;
    }
}

// First-stage decoder (bf16 convolutions, fp32 residual stream): the apply pass with a bf16 output — thread = 8 consecutive
// channels of one row (two 16-byte loads, one 16-byte store), same statistics passes and per-(sample, channel) affine.
__global__ __launch_bounds__(256) void gn_apply_bf16_kernel(const float* __restrict__ x, long rows, long S, int C, long ldx,
                                                            const float* __restrict__ ab, int apply_silu,
                                                            unsigned short* __restrict__ out, long ldo, int fh) {
    const int chunks = C >> 3;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= rows * chunks) return;
    const long row = gid / chunks;
    const int c = (int)(gid - row * chunks) * 8;
    const long b = row / S;
    const float4 x0 = *reinterpret_cast<const float4*>(x + row * ldx + c);
    const float4 x1 = *reinterpret_cast<const float4*>(x + row * ldx + c + 4);
    const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    float y[8];
    const float* abp = ab + (b * C + c) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(abp + 4 * j);
        y[2 * j] = v[2 * j] * t.x + t.y;
        y[2 * j + 1] = v[2 * j + 1] * t.z + t.w;
    }
    if (apply_silu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = y[j] * (1.0f / (1.0f + expf(-y[j])));
    }
    v4i pk;
#pragma unroll
    for (int j = 0; j < 4; ++j) pk[j] = fh ? (int)qd_pack2h(y[2 * j], y[2 * j + 1]) : (int)qd_pack2bf(y[2 * j], y[2 * j + 1]);
    *reinterpret_cast<v4i*>(out + row * ldo + c) = pk;
}

}  // namespace

// GroupNorm (+ SiLU) of fp32 NHWC rows into bf16 rows: the producer of the bf16 convolutions of the first-stage decoder
// (reference ldm/modules/diffusionmodules/model.py:38-45 Normalize / nonlinearity in front of every convolution).
// gamma / beta may be null (x * rstd - mean * rstd).
extern "C" int qd_groupnorm_silu_h16(const float* x, int64_t B, int64_t S, int C, int64_t ldx, int groups, float eps,
                                     const float* gamma, const float* beta, int apply_silu, int out_dtype, void* out, int64_t ldo, void* ws,
                                     const float* part_in, int nchunk_in, int64_t part_ld, void* stream);
extern "C" int qd_groupnorm_silu_bf16(const float* x, int64_t B, int64_t S, int C, int64_t ldx, int groups, float eps,
                                      const float* gamma, const float* beta, int apply_silu, void* out, int64_t ldo, void* ws,
                                      const float* part_in, int nchunk_in, int64_t part_ld, void* stream) {
    return qd_groupnorm_silu_h16(x, B, S, C, ldx, groups, eps, gamma, beta, apply_silu, QD_BF16, out, ldo, ws, part_in, nchunk_in, part_ld, stream);
}

// the same with the output type as a parameter: QD_BF16 or QD_F16 (the reference decodes under fp16 autocast, scripts/txt2img.py:231-236)
extern "C" int qd_groupnorm_silu_h16(const float* x, int64_t B, int64_t S, int C, int64_t ldx, int groups, float eps,
                                     const float* gamma, const float* beta, int apply_silu, int out_dtype, void* out, int64_t ldo, void* ws,
                                     const float* part_in, int nchunk_in, int64_t part_ld, void* stream) {
    QD_REQUIRE(out_dtype == QD_BF16 || out_dtype == QD_F16, "qd_groupnorm_silu_h16: out_dtype must be bf16 or f16");
    QD_REQUIRE(x && out && ws, "qd_groupnorm_silu_bf16: null pointer");
    QD_REQUIRE(B > 0 && S > 0 && C > 0 && groups > 0 && C % groups == 0 && C % 8 == 0, "qd_groupnorm_silu_bf16: C=%d must be a multiple of 8 and of groups=%d", C, groups);
    QD_REQUIRE(ldx >= C && ldx % 4 == 0 && qd_aligned(x, 16) && ldo >= C && ldo % 8 == 0 && qd_aligned(out, 16), "qd_groupnorm_silu_bf16: rows must be 16-byte aligned");
    QD_REQUIRE(B < 65536, "qd_groupnorm_silu_bf16: batch too large");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nchunk_own = (int)((S + gn_rows(S) - 1) / gn_rows(S));
    float* part = reinterpret_cast<float*>(ws);
    float* ab = part + (size_t)B * nchunk_own * C * 2;
    QD_REQUIRE(!part_in || (nchunk_in > 0 && (part_ld == 0 || part_ld >= C)), "qd_groupnorm_silu_bf16: part_in needs nchunk_in > 0 and part_ld >= C");
    const long ldp = part_in && part_ld ? (long)part_ld : (long)C;
    const int nchunk = part_in ? nchunk_in : nchunk_own;
    if (!part_in)
        hipLaunchKernelGGL(gn_partial_kernel<float>, dim3(nchunk, (unsigned)B), dim3(256), 0, st, x, (long)S, C, (long)ldx, part, nchunk, 1);
    if ((long)nchunk * (C / groups) >= 2048)
        hipLaunchKernelGGL(gn_finalize_wide_kernel, dim3(groups, (unsigned)B), dim3(256), 0, st, part_in ? part_in : part, nchunk, ldp, (long)S, C, groups, eps, gamma, beta, ab);
    else
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, (unsigned)B), dim3(64), 0, st, part_in ? part_in : part, nchunk, ldp, (long)S, C, groups, eps, gamma, beta, ab);
    const long rows = B * S, total = rows * (C / 8);
    hipLaunchKernelGGL(gn_apply_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, rows, (long)S, C, (long)ldx, ab, apply_silu,
                       reinterpret_cast<unsigned short*>(out), (long)ldo, out_dtype == QD_F16 ? 1 : 0);
    QD_LAUNCH_CHECK("qd_groupnorm_silu_bf16");
    return 0;
}

extern "C" int64_t qd_groupnorm_ws_bytes(int64_t B, int64_t C, int64_t S) {
    int64_t nchunk = (S + gn_rows(S) - 1) / gn_rows(S);
    return (B * nchunk * C * 2 + B * C * 2) * (int64_t)sizeof(float);
}

static int groupnorm_impl(const void* x, int x_dtype, int64_t B, int64_t S, int C, int64_t ldx, int groups,
                          float eps, const float* gamma, const float* beta, int apply_silu,
                          const float* qparams, int qmin, int qmax, int off, int8_t* out, int64_t ldo,
                          float* yout, int64_t ldy, void* ws, const float* part_in, int nchunk_in, int64_t part_ld, const qd_raw_quant* raw,
                          const float* mod, int64_t mod_ld, void* stream) {
    QD_REQUIRE(x && ws && (out || yout), "qd_groupnorm_silu_quant: null pointer");
    QD_REQUIRE(!out || qparams, "qd_groupnorm_silu_quant: quantised output needs qparams");
    QD_REQUIRE(x_dtype == QD_F32 || x_dtype == QD_F16, "qd_groupnorm_silu_quant: dtype must be f32/f16");
    QD_REQUIRE(B > 0 && S > 0 && C > 0 && groups > 0 && C % groups == 0 && C % 16 == 0, "qd_groupnorm_silu_quant: C=%d must be a multiple of 16 and of groups=%d", C, groups);
    QD_REQUIRE(ldx >= C && (!out || (ldo >= C && ldo % 16 == 0 && qd_aligned(out, 16))), "qd_groupnorm_silu_quant: bad leading dimensions");
    QD_REQUIRE(B < 65536, "qd_groupnorm_silu_quant: batch too large");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nchunk_own = (int)((S + gn_rows(S) - 1) / gn_rows(S));
    const int vec = qd_aligned(x, x_dtype == QD_F32 ? 16 : 8) && ldx % 4 == 0;
    // fp16 rows that take 16-byte (8-half) lanes
    const bool vec8 = x_dtype == QD_F16 && qd_aligned(x, 16) && ldx % 8 == 0;      // C % 16 == 0 is required above
    float* part = reinterpret_cast<float*>(ws);
    float* ab = part + (size_t)B * nchunk_own * C * 2;
    QD_REQUIRE(!part_in || (nchunk_in > 0 && (part_ld == 0 || part_ld >= C)), "qd_groupnorm_silu_quant: part_in needs nchunk_in > 0 and part_ld >= C");
    const long ldp = part_in && part_ld ? (long)part_ld : (long)C;
    const int nchunk = part_in ? nchunk_in : nchunk_own;
    if (part_in) {
        // first statistics level came with the tensor (written by the producing GEMM's epilogue)
    } else if (x_dtype == QD_F32)
        hipLaunchKernelGGL(gn_partial_kernel<float>, dim3(nchunk, (unsigned)B), dim3(256), 0, st, (const float*)x, (long)S, C, (long)ldx, part, nchunk, vec);
    else if (vec8)
        hipLaunchKernelGGL(gn_partial_h8_kernel, dim3(nchunk, (unsigned)B), dim3(256), 0, st, (const __half*)x, (long)S, C, (long)ldx, part, nchunk);
    else
        hipLaunchKernelGGL(gn_partial_kernel<__half>, dim3(nchunk, (unsigned)B), dim3(256), 0, st, (const __half*)x, (long)S, C, (long)ldx, part, nchunk, vec);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, (unsigned)B), dim3(64), 0, st, part_in ? part_in : part, nchunk, ldp, (long)S, C, groups, eps, gamma, beta, ab);
    if (mod) {
        QD_REQUIRE(mod_ld >= 2 * (int64_t)C, "qd_groupnorm_mod_silu_quant: modulation rows hold scale | shift: mod_ld >= 2 C");
        const long tot = (long)B * C;
        hipLaunchKernelGGL(gn_modulate_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, ab, mod, (long)mod_ld, C, tot);
    }
    RawQ rq{};
    if (raw && raw->out) {
        QD_REQUIRE(raw->nseg == 1 || raw->nseg == 2, "qd_groupnorm_silu_quant: raw output takes 1 or 2 segments");
        QD_REQUIRE(raw->ldo % 16 == 0 && qd_aligned(raw->out, 16), "qd_groupnorm_silu_quant: raw output rows must be 16-byte aligned");
        rq.out = raw->out; rq.ldo = (long)raw->ldo; rq.nseg = raw->nseg;
        for (int i = 0; i < raw->nseg; ++i) {
            const auto& g = raw->seg[i];
            QD_REQUIRE(g.qparams && g.c0 >= 0 && g.clen > 0 && g.c0 % 16 == 0 && g.clen % 16 == 0 && g.oc0 % 16 == 0 && g.c0 + g.clen <= C &&
                       g.oc0 + g.clen <= raw->ldo, "qd_groupnorm_silu_quant: raw segment %d: c0 / clen / oc0 must be multiples of 16 inside the rows", i);
            QD_REQUIRE(g.qmax - g.off <= 127 && g.qmin - g.off >= -128, "qd_groupnorm_silu_quant: raw segment %d grid does not fit int8", i);
            QD_REQUIRE(i == 0 || g.c0 >= raw->seg[0].c0 + raw->seg[0].clen, "qd_groupnorm_silu_quant: raw segments must be ordered and disjoint");
            rq.c0[i] = g.c0; rq.clen[i] = g.clen; rq.oc0[i] = g.oc0; rq.off[i] = g.off;
            rq.qmin[i] = (float)g.qmin; rq.qmax[i] = (float)g.qmax; rq.qp[i] = g.qparams;
        }
    }
    long rows = B * S;
    long total = rows * (C / 4);
    if (x_dtype == QD_F32 && vec && out && !yout && S % 2 == 0) {       // two rows per thread (four measured no better, round 3)
        const long tot = (rows / 2) * (C / 4);
        dim3 g2((unsigned)((tot + 255) / 256));
        hipLaunchKernelGGL(gn_apply_rows_kernel<2>, g2, dim3(256), 0, st, (const float*)x, rows, (long)S, C, (long)ldx, ab, apply_silu, qparams, (float)qmin, (float)qmax, off, out, (long)ldo, rq);
        QD_LAUNCH_CHECK("qd_groupnorm_silu_quant");
        return 0;
    }
    if (vec8 && out && !yout && S % 2 == 0) {
        const long tot = (rows / 2) * (C / 8);
        hipLaunchKernelGGL(gn_apply_rows_h8_kernel<2>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const __half*)x, rows, (long)S, C, (long)ldx, ab, apply_silu,
                           qparams, (float)qmin, (float)qmax, off, out, (long)ldo, rq);
        QD_LAUNCH_CHECK("qd_groupnorm_silu_quant");
        return 0;
    }
    dim3 grid((unsigned)((total + 255) / 256));
    if (x_dtype == QD_F32)
        hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(256), 0, st, (const float*)x, rows, (long)S, C, (long)ldx, ab, apply_silu, qparams, (float)qmin, (float)qmax, off, out, (long)ldo, yout, (long)ldy, vec, rq);
    else
        hipLaunchKernelGGL(gn_apply_kernel<__half>, grid, dim3(256), 0, st, (const __half*)x, rows, (long)S, C, (long)ldx, ab, apply_silu, qparams, (float)qmin, (float)qmax, off, out, (long)ldo, yout, (long)ldy, vec, rq);
    QD_LAUNCH_CHECK("qd_groupnorm_silu_quant");
    return 0;
}

extern "C" int qd_groupnorm_silu_quant(const void* x, int x_dtype, int64_t B, int64_t S, int C, int64_t ldx, int groups,
                                       float eps, const float* gamma, const float* beta, int apply_silu,
                                       const float* qparams, int qmin, int qmax, int off, int8_t* out, int64_t ldo,
                                       float* yout, int64_t ldy, void* ws, const float* part_in, int nchunk_in, int64_t part_ld, const qd_raw_quant* raw, void* stream) {
    return groupnorm_impl(x, x_dtype, B, S, C, ldx, groups, eps, gamma, beta, apply_silu, qparams, qmin, qmax, off, out, ldo, yout, ldy, ws,
                          part_in, nchunk_in, part_ld, raw, nullptr, 0, stream);
}

extern "C" int qd_groupnorm_mod_silu_quant(const void* x, int x_dtype, int64_t B, int64_t S, int C, int64_t ldx, int groups,
                                           float eps, const float* gamma, const float* beta, int apply_silu,
                                           const float* qparams, int qmin, int qmax, int off, int8_t* out, int64_t ldo,
                                           float* yout, int64_t ldy, void* ws, const float* part_in, int nchunk_in, int64_t part_ld,
                                           const float* mod, int64_t mod_ld, void* stream) {
    QD_REQUIRE(mod, "qd_groupnorm_mod_silu_quant: null modulation rows");
    return groupnorm_impl(x, x_dtype, B, S, C, ldx, groups, eps, gamma, beta, apply_silu, qparams, qmin, qmax, off, out, ldo, yout, ldy, ws,
                          part_in, nchunk_in, part_ld, nullptr, mod, mod_ld, stream);
}

extern "C" int qd_layernorm_quant(const void* x, int x_dtype, int64_t M, int C, int64_t ldx, float eps,
                                  const float* gamma, const float* beta, int nout, const float* const* qparams,
                                  const int* qmin, const int* qmax, const int* off, int8_t* const* out, int64_t ldo,
                                  void* stream) {
    QD_REQUIRE(x && gamma && beta && qparams && qmin && qmax && off && out, "qd_layernorm_quant: null pointer");
    QD_REQUIRE(x_dtype == QD_F32 || x_dtype == QD_F16, "qd_layernorm_quant: dtype must be f32/f16");
    QD_REQUIRE(nout >= 1 && nout <= 3, "qd_layernorm_quant: nout must be 1..3");
    QD_REQUIRE(M > 0 && C > 0 && C % 16 == 0 && C <= 64 * 4 * LN_MAXV, "qd_layernorm_quant: C=%d unsupported (multiple of 16, <= %d)", C, 64 * 4 * LN_MAXV);
    QD_REQUIRE(ldx >= C && ldo >= C && ldo % 16 == 0, "qd_layernorm_quant: bad leading dimensions");
    float3 mn = {0, 0, 0}, mx = {0, 0, 0};
    int3 of = {0, 0, 0};
    const float* qp[3] = {nullptr, nullptr, nullptr};
    int8_t* o[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < nout; ++i) {
        QD_REQUIRE(qparams[i] && out[i] && qd_aligned(out[i], 16), "qd_layernorm_quant: output %d null/unaligned", i);
        qp[i] = qparams[i];
        o[i] = out[i];
        (&mn.x)[i] = (float)qmin[i];
        (&mx.x)[i] = (float)qmax[i];
        (&of.x)[i] = off[i];
    }
    QD_REQUIRE(qd_aligned(gamma, 16) && qd_aligned(beta, 16), "qd_layernorm_quant: gamma/beta must be 16-byte aligned");
    const int vec = qd_aligned(x, x_dtype == QD_F32 ? 16 : 8) && ldx % 4 == 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nvl = (C / 4 + 63) / 64;
    if (C == 320 && vec) {
        dim3 grid((unsigned)((M + 31) / 32));
        if (x_dtype == QD_F32)
            hipLaunchKernelGGL((ln_quant_rows8_kernel<float, 10>), grid, dim3(256), 0, st, (const float*)x, (long)M, (long)ldx, eps, gamma, beta, nout, qp[0], qp[1], qp[2],
                               mn, mx, of, o[0], o[1], o[2], (long)ldo);
        else
            hipLaunchKernelGGL((ln_quant_rows8_kernel<__half, 10>), grid, dim3(256), 0, st, (const __half*)x, (long)M, (long)ldx, eps, gamma, beta, nout, qp[0], qp[1], qp[2],
                               mn, mx, of, o[0], o[1], o[2], (long)ldo);
    } else if (x_dtype == QD_F16 && qd_aligned(x, 16) && ldx % 8 == 0 && C % 8 == 0 && C <= 64 * 8 * 3) {
        const int nv8 = (C / 8 + 63) / 64;
        if (nv8 == 1) launch_ln_h8<1>(st, x, (long)M, C, (long)ldx, eps, gamma, beta, nout, qp, mn, mx, of, o, (long)ldo);
        else if (nv8 == 2) launch_ln_h8<2>(st, x, (long)M, C, (long)ldx, eps, gamma, beta, nout, qp, mn, mx, of, o, (long)ldo);
        else launch_ln_h8<3>(st, x, (long)M, C, (long)ldx, eps, gamma, beta, nout, qp, mn, mx, of, o, (long)ldo);
    } else if (x_dtype == QD_F32) dispatch_ln<float>(nvl, st, x, (long)M, C, (long)ldx, eps, gamma, beta, nout, qp, mn, mx, of, o, (long)ldo, vec);
    else dispatch_ln<__half>(nvl, st, x, (long)M, C, (long)ldx, eps, gamma, beta, nout, qp, mn, mx, of, o, (long)ldo, vec);
    QD_LAUNCH_CHECK("qd_layernorm_quant");
    return 0;
}
