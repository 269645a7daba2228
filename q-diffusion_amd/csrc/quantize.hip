// quantize.hip — K1 activation quantiser, K2 weight packer, K9b GEGLU->quant, head re-layout for
// attention.  All of these are HBM-bound byte movers: one pass, 16-byte stores, no re-reads.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// K1a: channel-contiguous input (channels_last / token-major): thread = 16 channels of one row.
// Replaces the 7-8 elementwise ATen passes of quant_layer.py:82-88 with one read + one 1-byte
// write per element.
// ---------------------------------------------------------------------------------------------
// lane = 4 consecutive channels of one row: a wave reads 1 KB contiguous (fp32) and writes 256 B
// contiguous — fully coalesced on both sides (the first version gave each lane 16 channels, i.e. a
// 64-byte lane stride on the read side, 1/4 of the TA rate).
template <typename T>
__global__ __launch_bounds__(256) void quant_rows_kernel(const T* __restrict__ x, long rows, long row_stride,
                                                         long S, long sb, int c0, int clen, int clen_pad,
                                                         const float* __restrict__ qp, float qmin, float qmax,
                                                         int off, int8_t* __restrict__ out, long ldo, int oc0, int vec) {
    const int groups = clen_pad >> 2;
    long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= rows * groups) return;
    long row = gid / groups;
    int  g   = (int)(gid - row * groups);
    const QP q = qd_load_qp(qp);
    const int ztrue = (int)q.zp - off;
    long b = row / S, s = row - b * S;
    const int c = g * 4;
    unsigned u;
    if (c + 4 <= clen) {
        float v[4];
        qd_ld4(x + b * sb + s * row_stride + c0 + c, vec != 0, v);
        u = 0;
        auto body = [&](auto ft) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) u |= (unsigned)((qd_code_t<decltype(ft)::value>(v[j], q, qmin, qmax) - off) & 0xff) << (8 * j);
        };
        QD_FAST_DISPATCH(q.fast, body);
    } else {
        u = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int code = ztrue;
            if (c + j < clen) code = qd_code(qd_ld(x + b * sb + s * row_stride + c0 + c + j), q, qmin, qmax) - off;
            u |= (unsigned)(code & 0xff) << (8 * j);
        }
    }
    *reinterpret_cast<unsigned*>(out + row * ldo + oc0 + c) = u;
}

// ---------------------------------------------------------------------------------------------
// K1b: arbitrary strides (e.g. NCHW-contiguous tensors handed over by unmodified reference UNet
// code).  64 channels x 64 positions per block through an LDS byte tile: reads are coalesced
// along the input's fast axis, writes are 16-byte NHWC rows.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void quant_strided_kernel(const T* __restrict__ x, long S, long sb, long sc,
                                                            long ss, int c0, int clen, int clen_pad,
                                                            const float* __restrict__ qp, float qmin, float qmax,
                                                            int off, int8_t* __restrict__ out, long ldo, int oc0) {
    __shared__ signed char tile[64][68];
    const long s0 = (long)blockIdx.x * 64;
    const int  ct = blockIdx.y * 64;
    const long b  = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const QP q = qd_load_qp(qp);
    const int ztrue = (int)q.zp - off;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        int c = wave * 16 + i;
        int code = ztrue;
        if (ct + c < clen && s0 + lane < S) {
            float v = qd_ld(x + b * sb + (long)(c0 + ct + c) * sc + (s0 + lane) * ss);
            code = qd_code(v, q, qmin, qmax) - off;
        }
        tile[c][lane] = (signed char)code;
    }
    __syncthreads();
    const int s = threadIdx.x >> 2, ch = threadIdx.x & 3;
    if (s0 + s >= S || ct + ch * 16 >= clen_pad) return;
    v4i v;
#pragma unroll
    for (int wd = 0; wd < 4; ++wd) {
        unsigned u = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) u |= (unsigned)((int)tile[ch * 16 + wd * 4 + j][s] & 0xff) << (8 * j);
        v[wd] = (int)u;
    }
    *reinterpret_cast<v4i*>(out + (b * S + s0 + s) * ldo + oc0 + ct + ch * 16) = v;
}

// ---------------------------------------------------------------------------------------------
// K2: weight packer (one-time).  thread = (out-channel n, tap t, 16-channel chunk).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, const float* __restrict__ alpha,
                                                           const float* __restrict__ delta, const float* __restrict__ zp,
                                                           int Cout, int Cin_total, int taps, int c0, int clen,
                                                           int clen_pad, int n_levels, int mode,
                                                           uint8_t* __restrict__ wq, long ldk, int kofs,
                                                           int32_t* __restrict__ wsum, int32_t* __restrict__ codes) {
    const int chunks = clen_pad >> 4;
    long gid = (long)blockIdx.x * 256 + threadIdx.x;
    long total = (long)Cout * taps * chunks;
    if (gid >= total) return;
    int ch = (int)(gid % chunks);
    long nt = gid / chunks;
    int t = (int)(nt % taps);
    int n = (int)(nt / taps);
    const float d = delta[n], z = zp[n];
    const int zi = (int)z;
    int vals[16];
    int sum = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        int c = ch * 16 + j;
        int stored = 0, code = 0;
        if (c < clen) {
            float wv = w[((long)n * Cin_total + c0 + c) * taps + t];
            float q;
            if (alpha) {
                // adaptive_rounding.py:50-59: floor(w/delta) + (alpha >= 0), then + zero_point, clamp
                float a = alpha[((long)n * clen + c) * taps + t];
                q = floorf(wv / d) + (a >= 0.f ? 1.f : 0.f);
            } else {
                // quant_layer.py:82: round(w/delta)  (round-half-even)
                q = rintf(wv / d);
            }
            q = fminf(fmaxf(q + z, 0.f), (float)(n_levels - 1));
            code = (int)q;
            if (codes) codes[((long)n * clen + c) * taps + t] = code;
            stored = (mode == 8) ? code - 128 : (mode == 0 ? code - zi : code);
            sum += (mode == 4) ? code - zi : stored;
        } else if (mode == 4) {
            sum += -zi;  // pad nibble 0 unpacks to -zw; the activation side holds "true zero" there
        }
        vals[j] = stored;
    }
    if (mode == 4) {
        // nibble layout: word0 byte b = k[b] | k[4+b]<<4 ; word1 byte b = k[8+b] | k[12+b]<<4
        unsigned w0 = 0, w1 = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            w0 |= (unsigned)((vals[b] & 15) | ((vals[4 + b] & 15) << 4)) << (8 * b);
            w1 |= (unsigned)((vals[8 + b] & 15) | ((vals[12 + b] & 15) << 4)) << (8 * b);
        }
        uint2 pk = {w0, w1};
        *reinterpret_cast<uint2*>(wq + (((long)n * taps + t) * ldk + kofs + ch * 16) / 2) = pk;
    } else {
        v4i v;
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
            unsigned u = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) u |= (unsigned)(vals[wd * 4 + j] & 0xff) << (8 * j);
            v[wd] = (int)u;
        }
        *reinterpret_cast<v4i*>(wq + ((long)n * taps + t) * ldk + kofs + ch * 16) = v;
    }
    if (wsum && sum != 0) atomicAdd(&wsum[n], sum);
}

// ---------------------------------------------------------------------------------------------
// K9b: GEGLU -> quant.  out = quant(h[:, :F] * gelu(h[:, F:]))   (attention.py:42-44, erf GELU)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void geglu_quant_kernel(const T* __restrict__ h, long M, int F, long ldh,
                                                          const float* __restrict__ qp, float qmin, float qmax,
                                                          int off, int8_t* __restrict__ out, long ldo) {
    const int chunks = F >> 4;
    long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= M * chunks) return;
    long row = gid / chunks;
    int ch = (int)(gid - row * chunks);
    const QP q = qd_load_qp(qp);
    const T* xa = h + row * ldh + ch * 16;
    const T* xg = xa + F;
    v4i v;
    auto body = [&](auto ft) __attribute__((always_inline)) {
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
            unsigned u = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = qd_ld(xa + wd * 4 + j), g = qd_ld(xg + wd * 4 + j);
                float gl = 0.5f * g * (1.0f + qd_erff(g * 0.70710678118654752440f));
                int code = qd_code_t<decltype(ft)::value>(a * gl, q, qmin, qmax) - off;
                u |= (unsigned)(code & 0xff) << (8 * j);
            }
            v[wd] = (int)u;
        }
    };
    QD_FAST_DISPATCH(q.fast, body);
    *reinterpret_cast<v4i*>(out + row * ldo + ch * 16) = v;
}

// ---------------------------------------------------------------------------------------------
// attention head re-layout + quantise.
//   transpose=0: thread = one (bh, t) row; out[bh][t][dpad], rsum[bh][t]
//   transpose=1: thread = (bh, 16 permuted key slots, dd); out[bh][dd][Tpad], rsum[bh][dd] (atomics)
// key permutation inside a 32-key tile: slot p=half*16+r  <->  key (r&3)+8*(r>>2)+4*half
// ---------------------------------------------------------------------------------------------
// rows layout: a block owns `tpb` whole tokens; lane = 4 consecutive features (d % 4 == 0, so a group
// never straddles two heads): reads are contiguous float4s along the token's feature axis, writes are
// 4-byte pieces of the head rows, row sums are accumulated with LDS atomics (integer, deterministic)
// and stored once.  Pad rows / pad features are zeroed by a memset node ahead of the launch.
template <typename T>
__global__ __launch_bounds__(256) void quant_heads_rows_kernel(const T* __restrict__ x, int B, int Tn, int H, int d,
                                                               long sb, long st, long sh, long sd, float prescale,
                                                               const float* __restrict__ qp, float qmin, float qmax,
                                                               int off, int8_t* __restrict__ out,
                                                               int32_t* __restrict__ rsum, int Tpad, int dpad, int tpb, int vec) {
    __shared__ int ssum[256];
    const int gpr = (H * d) >> 2;                       // float4 groups per token
    const long tok0 = (long)blockIdx.x * tpb;            // first token (b*T + t) of this block
    const long ntok = (long)B * Tn;
    for (int i = threadIdx.x; i < tpb * H; i += 256) ssum[i] = 0;
    __syncthreads();
    const QP q = qd_load_qp(qp);
    for (int idx = threadIdx.x; idx < tpb * gpr; idx += 256) {
        const int lt = idx / gpr, g = idx - lt * gpr;
        const long tok = tok0 + lt;
        if (tok >= ntok) break;
        const int b = (int)(tok / Tn), t = (int)(tok - (long)b * Tn);
        const int f = g * 4, hh = f / d, dd = f - hh * d;
        float v[4];
        const T* src = x + b * sb + (long)t * st + hh * sh + dd * sd;
        if (sd == 1) qd_ld4(src, vec != 0, v);
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = qd_ld(src + j * sd);
        }
        unsigned u = 0;
        int sum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int code = qd_code(v[j] * prescale, q, qmin, qmax) - off;
            sum += code;
            u |= (unsigned)(code & 0xff) << (8 * j);
        }
        *reinterpret_cast<unsigned*>(out + (((long)b * H + hh) * Tpad + t) * dpad + dd) = u;
        if (rsum) atomicAdd(&ssum[lt * H + hh], sum);
    }
    if (!rsum) return;
    __syncthreads();
    for (int i = threadIdx.x; i < tpb * H; i += 256) {
        const long tok = tok0 + i / H;
        if (tok >= ntok) continue;
        const int b = (int)(tok / Tn), t = (int)(tok - (long)b * Tn);
        rsum[((long)b * H + (i % H)) * Tpad + t] = ssum[i];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void quant_heads_tr_kernel(const T* __restrict__ x, int B, int Tn, int H, int d,
                                                             long sb, long st, long sh, long sd, float prescale,
                                                             const float* __restrict__ qp, float qmin, float qmax,
                                                             int off, int8_t* __restrict__ out,
                                                             int32_t* __restrict__ rsum, int Tpad, int dpad) {
    // grid: x = ceil(dpad/64) * (Tpad/16) ; y = B*H.   lane -> dd (coalesced reads along d)
    const int nslot = Tpad >> 4;
    const int slot = blockIdx.x % nslot;          // 16-slot group along the permuted key axis
    const int dblk = blockIdx.x / nslot;
    const int bh = blockIdx.y;
    const int hh = bh % H, b = bh / H;
    // 256 threads: 64 dd x 4 sub-slots of 4 keys
    const int dd = dblk * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;  // which 4 of the 16 slots
    if (dd >= dpad) return;
    const QP q = qd_load_qp(qp);
    const int tile = slot >> 1, half = slot & 1;
    unsigned u = 0;
    int sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int r = sub * 4 + j;
        int t = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        int code = 0;
        if (t < Tn && dd < d) {
            float xv = qd_ld(x + b * sb + (long)t * st + hh * sh + dd * sd) * prescale;
            code = qd_code(xv, q, qmin, qmax) - off;
            sum += code;
        }
        u |= (unsigned)(code & 0xff) << (8 * j);
    }
    *reinterpret_cast<unsigned*>(out + ((long)bh * dpad + dd) * Tpad + slot * 16 + sub * 4) = u;
    if (rsum && sum != 0) atomicAdd(&rsum[(long)bh * dpad + dd], sum);
}

}  // namespace

// qd_make_qparams: {delta, zero_point} -> {delta, zero_point, rinv, fast} (common.h QP).  One thread per mantissa of x
// in [1, 2): the three-instruction quotient must equal the IEEE division for every one of them (scaling x by a power of
// two scales every intermediate exactly, the sign is symmetric), else `fast` is cleared and the kernels divide.
__global__ void qparams_init_kernel(const float* __restrict__ delta, const float* __restrict__ zp, float* __restrict__ out) {
    const float d = delta[0];
    out[0] = d;
    out[1] = zp[0];
    out[2] = (float)(1.0 / (double)d);
    out[3] = (d > 0.f && d < 3.0e38f) ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void qparams_check_kernel(float* __restrict__ out) {
    const unsigned m = blockIdx.x * 256u + threadIdx.x;            // 2^23 mantissas
    const float x = __uint_as_float(0x3f800000u | m);
    const float d = out[0], r = out[2];
    const float y = x * r;
    const float e = __builtin_fmaf(-y, d, x);
    const float q = __builtin_fmaf(e, r, y);
    if (__float_as_uint(q) != __float_as_uint(x / d)) out[3] = 0.f;   // idempotent store: no atomics needed
}

extern "C" int qd_make_qparams(const float* delta, const float* zero_point, float* out4, void* stream) {
    QD_REQUIRE(delta && zero_point && out4 && qd_aligned(out4, 16), "qd_make_qparams: null / unaligned pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(qparams_init_kernel, dim3(1), dim3(1), 0, st, delta, zero_point, out4);
    hipLaunchKernelGGL(qparams_check_kernel, dim3(1u << 15), dim3(256), 0, st, out4);
    QD_LAUNCH_CHECK("qd_make_qparams");
    return 0;
}

extern "C" int qd_quantize_act(const void* x, int x_dtype, int64_t B, int64_t C, int64_t S, int64_t sb, int64_t sc,
                               int64_t ss, int c0, int clen, int clen_pad, const float* qparams, int qmin, int qmax,
                               int off, int8_t* out, int64_t ldo, int oc0, void* stream) {
    QD_REQUIRE(x && out && qparams, "qd_quantize_act: null pointer");
    QD_REQUIRE(x_dtype == QD_F32 || x_dtype == QD_F16, "qd_quantize_act: dtype must be f32/f16");
    QD_REQUIRE(B > 0 && S > 0 && clen > 0 && c0 >= 0 && c0 + clen <= C, "qd_quantize_act: bad shape (C=%ld c0=%d clen=%d)", (long)C, c0, clen);
    QD_REQUIRE(clen_pad % 16 == 0 && clen_pad >= clen && clen_pad - clen < 16, "qd_quantize_act: clen_pad must be clen rounded up to 16");
    QD_REQUIRE(oc0 % 16 == 0 && ldo % 16 == 0 && oc0 + clen_pad <= ldo && qd_aligned(out, 16), "qd_quantize_act: output must be 16-byte aligned (oc0=%d ldo=%ld)", oc0, (long)ldo);
    QD_REQUIRE(qmin >= -128 - 0 && qmax - off <= 127 && qmin - off >= -128, "qd_quantize_act: grid [%d,%d]-%d does not fit int8", qmin, qmax, off);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (sc == 1) {
        long rows = B * S;
        long total = rows * (clen_pad / 4);
        dim3 grid((unsigned)((total + 255) / 256));
        const size_t esz = x_dtype == QD_F32 ? 4 : 2;
        const int vec = qd_aligned(x, 4 * esz) && (ss % 4 == 0) && (sb % 4 == 0) && (c0 % 4 == 0);
        if (x_dtype == QD_F32)
            hipLaunchKernelGGL(quant_rows_kernel<float>, grid, dim3(256), 0, st, (const float*)x, rows, (long)ss, (long)S, (long)sb, c0, clen, clen_pad, qparams, (float)qmin, (float)qmax, off, out, (long)ldo, oc0, vec);
        else
            hipLaunchKernelGGL(quant_rows_kernel<__half>, grid, dim3(256), 0, st, (const __half*)x, rows, (long)ss, (long)S, (long)sb, c0, clen, clen_pad, qparams, (float)qmin, (float)qmax, off, out, (long)ldo, oc0, vec);
    } else {
        QD_REQUIRE(B < 65536, "qd_quantize_act: batch too large for strided kernel");
        dim3 grid((unsigned)((S + 63) / 64), (unsigned)((clen_pad + 63) / 64), (unsigned)B);
        if (x_dtype == QD_F32)
            hipLaunchKernelGGL(quant_strided_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (long)S, (long)sb, (long)sc, (long)ss, c0, clen, clen_pad, qparams, (float)qmin, (float)qmax, off, out, (long)ldo, oc0);
        else
            hipLaunchKernelGGL(quant_strided_kernel<__half>, grid, dim3(256), 0, st, (const __half*)x, (long)S, (long)sb, (long)sc, (long)ss, c0, clen, clen_pad, qparams, (float)qmin, (float)qmax, off, out, (long)ldo, oc0);
    }
    QD_LAUNCH_CHECK("qd_quantize_act");
    return 0;
}

extern "C" int qd_pack_weights(const float* w, const float* alpha, const float* delta, const float* zp, int Cout,
                               int Cin_total, int taps, int c0, int clen, int clen_pad, int n_levels, int mode,
                               uint8_t* wq, int64_t ldk, int kofs, int32_t* wsum, int32_t* codes, void* stream) {
    QD_REQUIRE(w && delta && zp && wq, "qd_pack_weights: null pointer");
    QD_REQUIRE(mode == 8 || mode == 4 || mode == 0, "qd_pack_weights: mode must be 8, 4 or 0");
    QD_REQUIRE(Cout > 0 && taps > 0 && clen > 0 && c0 >= 0 && c0 + clen <= Cin_total, "qd_pack_weights: bad shape");
    QD_REQUIRE(clen_pad % 16 == 0 && clen_pad >= clen, "qd_pack_weights: clen_pad must be a multiple of 16");
    QD_REQUIRE(ldk % 16 == 0 && kofs % 16 == 0 && kofs + clen_pad <= ldk && qd_aligned(wq, 16), "qd_pack_weights: bad packed layout");
    QD_REQUIRE(n_levels >= 2 && n_levels <= 256 && (mode != 4 || n_levels <= 16), "qd_pack_weights: n_levels %d unsupported for mode %d", n_levels, mode);
    long total = (long)Cout * taps * (clen_pad / 16);
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       w, alpha, delta, zp, Cout, Cin_total, taps, c0, clen, clen_pad, n_levels, mode, wq, (long)ldk, kofs, wsum, codes);
    QD_LAUNCH_CHECK("qd_pack_weights");
    return 0;
}

extern "C" int qd_geglu_quant(const void* h, int h_dtype, int64_t M, int F, int64_t ldh, const float* qparams, int qmin,
                              int qmax, int off, int8_t* out, int64_t ldo, void* stream) {
    QD_REQUIRE(h && out && qparams, "qd_geglu_quant: null pointer");
    QD_REQUIRE(h_dtype == QD_F32 || h_dtype == QD_F16, "qd_geglu_quant: dtype must be f32/f16");
    QD_REQUIRE(M > 0 && F > 0 && F % 16 == 0 && ldh >= 2 * F && ldo >= F && ldo % 16 == 0 && qd_aligned(out, 16), "qd_geglu_quant: bad shape (F=%d)", F);
    long total = M * (F / 16);
    dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (h_dtype == QD_F32)
        hipLaunchKernelGGL(geglu_quant_kernel<float>, grid, dim3(256), 0, st, (const float*)h, (long)M, F, (long)ldh, qparams, (float)qmin, (float)qmax, off, out, (long)ldo);
    else
        hipLaunchKernelGGL(geglu_quant_kernel<__half>, grid, dim3(256), 0, st, (const __half*)h, (long)M, F, (long)ldh, qparams, (float)qmin, (float)qmax, off, out, (long)ldo);
    QD_LAUNCH_CHECK("qd_geglu_quant");
    return 0;
}

// Zero-fill as a KERNEL, not hipMemsetAsync: under stream capture the latter becomes a memset node, and two captured
// evaluations of one model that hold such nodes and are replayed alternately (A, B, A) came back wrong on ROCm 7.2 from
// the first block with memset nodes on — reproducibly when the captured graph is a single chain (a prepared context leaves
// no forked branch in it), never with kernel nodes only (tools/probes/pin_dbg6.py, pin_dbg8.py).
__global__ __launch_bounds__(256) void zero16_kernel(v4i* __restrict__ p, long n16) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) p[i] = v4i{0, 0, 0, 0};
}
static void zero_async(void* p, size_t bytes, hipStream_t st) {     // bytes % 16 == 0, p 16-byte aligned (callers check)
    const long n16 = (long)(bytes / 16);
    hipLaunchKernelGGL(zero16_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<v4i*>(p), n16);
}

extern "C" int qd_quantize_heads(const void* x, int x_dtype, int B, int T, int H, int d, int64_t sb, int64_t st_,
                                 int64_t sh, int64_t sd, float prescale, const float* qparams, int qmin, int qmax,
                                 int off, int transpose, int8_t* out, int32_t* rsum, int Tpad, int dpad, void* stream) {
    QD_REQUIRE(x && out && qparams, "qd_quantize_heads: null pointer");
    QD_REQUIRE(x_dtype == QD_F32 || x_dtype == QD_F16, "qd_quantize_heads: dtype must be f32/f16");
    QD_REQUIRE(B > 0 && T > 0 && H > 0 && d > 0, "qd_quantize_heads: bad shape");
    QD_REQUIRE(Tpad % 32 == 0 && Tpad >= T && dpad % 32 == 0 && dpad >= d && qd_aligned(out, 16), "qd_quantize_heads: Tpad/dpad must be multiples of 32 covering T/d");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!transpose) {
        QD_REQUIRE(d % 4 == 0 && H <= 256, "qd_quantize_heads: head dim must be a multiple of 4 and H <= 256 (d=%d H=%d)", d, H);
        QD_REQUIRE(!rsum || qd_aligned(rsum, 16), "qd_quantize_heads: rsum must be 16-byte aligned");
        zero_async(out, (size_t)B * H * Tpad * dpad, st);
        if (rsum) zero_async(rsum, sizeof(int32_t) * (size_t)B * H * Tpad, st);
        const int gpr = (H * d) / 4;
        int tpb = 1024 / gpr;                                   // ~4 float4 groups per thread
        if (tpb < 1) tpb = 1;
        while (tpb * H > 256) --tpb;
        const size_t esz = x_dtype == QD_F32 ? 4 : 2;
        const int vec = qd_aligned(x, 4 * esz) && sb % 4 == 0 && st_ % 4 == 0 && sh % 4 == 0 && sd == 1;
        dim3 grid((unsigned)(((long)B * T + tpb - 1) / tpb));
        if (x_dtype == QD_F32)
            hipLaunchKernelGGL(quant_heads_rows_kernel<float>, grid, dim3(256), 0, st, (const float*)x, B, T, H, d, (long)sb, (long)st_, (long)sh, (long)sd, prescale, qparams, (float)qmin, (float)qmax, off, out, rsum, Tpad, dpad, tpb, vec);
        else
            hipLaunchKernelGGL(quant_heads_rows_kernel<__half>, grid, dim3(256), 0, st, (const __half*)x, B, T, H, d, (long)sb, (long)st_, (long)sh, (long)sd, prescale, qparams, (float)qmin, (float)qmax, off, out, rsum, Tpad, dpad, tpb, vec);
    } else {
        QD_REQUIRE((long)B * H < 65536, "qd_quantize_heads: too many heads for grid.y");
        if (rsum) {
            QD_REQUIRE(qd_aligned(rsum, 16), "qd_quantize_heads: rsum must be 16-byte aligned");
            zero_async(rsum, sizeof(int32_t) * (size_t)B * H * dpad, st);
        }
        dim3 grid((unsigned)(((dpad + 63) / 64) * (Tpad / 16)), (unsigned)(B * H));
        if (x_dtype == QD_F32)
            hipLaunchKernelGGL(quant_heads_tr_kernel<float>, grid, dim3(256), 0, st, (const float*)x, B, T, H, d, (long)sb, (long)st_, (long)sh, (long)sd, prescale, qparams, (float)qmin, (float)qmax, off, out, rsum, Tpad, dpad);
        else
            hipLaunchKernelGGL(quant_heads_tr_kernel<__half>, grid, dim3(256), 0, st, (const __half*)x, B, T, H, d, (long)sb, (long)st_, (long)sh, (long)sd, prescale, qparams, (float)qmin, (float)qmax, off, out, rsum, Tpad, dpad);
    }
    QD_LAUNCH_CHECK("qd_quantize_heads");
    return 0;
}
