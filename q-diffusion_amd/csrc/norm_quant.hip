// norm_quant.hip — K5 GroupNorm(+SiLU)->quant and K9a LayerNorm->quant producers.
//
// The reference runs GroupNorm32 (fp32), SiLU and the next module's act quantiser as ~11 separate
// elementwise passes over an fp32 tensor (ddim/models/diffusion.py:121-130, openaimodel.py:201-232,
// quant_layer.py:82-88).  Here: one statistics pass (deterministic two-level reduction, no float
// atomics) and one apply pass that reads the activation once and writes one byte per element.
#include "common.h"

namespace {

constexpr int GN_ROWS = 32;  // rows per partial-sum block

// partial sums: grid (nchunk, B); thread owns 4 consecutive channels (one float4 load per row), loops over rows.
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, long S, int C, long ldx,
                                                         float* __restrict__ part, int nchunk, int vec) {
    const int chunk = blockIdx.x;
    const long b = blockIdx.y;
    const long r0 = (long)chunk * GN_ROWS;
    const long r1 = (r0 + GN_ROWS < S) ? r0 + GN_ROWS : S;
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

// finalize: grid (groups, B), 64 threads.  Writes per-(b,c) affine a = rstd*gamma, sh = beta - mean*a.
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ part, int nchunk, long S, int C,
                                                         int groups, float eps, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ ab) {
    const int g = blockIdx.x;
    const long b = blockIdx.y;
    const int cpg = C / groups;
    double s = 0.0, q = 0.0;
    const int items = nchunk * cpg;
    for (int i = threadIdx.x; i < items; i += 64) {
        int chunk = i / cpg, c = g * cpg + i % cpg;
        const float* p = part + (((b * nchunk + chunk) * (long)C) + c) * 2;
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

// apply: lane = 4 consecutive channels of one row (float4 in, 4 bytes out; both sides fully coalesced).
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, long rows, long S, int C, long ldx,
                                                       const float* __restrict__ ab, int apply_silu,
                                                       const float* __restrict__ qp, float qmin, float qmax, int off,
                                                       int8_t* __restrict__ out, long ldo, float* __restrict__ yout,
                                                       long ldy, int vec) {
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
    float delta = 1.f, zp = 0.f;
    if (out) { delta = qp[0]; zp = qp[1]; }
    unsigned u = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float y = v[j] * a4[j] + s4[j];
        if (apply_silu) y = y * (1.0f / (1.0f + expf(-y)));
        if (yout) yout[row * ldy + c + j] = y;
        if (out) u |= (unsigned)((qd_code(y, delta, zp, qmin, qmax) - off) & 0xff) << (8 * j);
    }
    if (out) *reinterpret_cast<unsigned*>(out + row * ldo + c) = u;
}

// LayerNorm: one wave per row, row held in registers (C <= 64*4*MAXV).
constexpr int LN_MAXV = 6;  // up to 1536 channels
template <typename T>
__global__ __launch_bounds__(256) void ln_quant_kernel(const T* __restrict__ x, long M, int C, long ldx, float eps,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int nout, const float* qp0, const float* qp1, const float* qp2,
                                                       float3 qmin, float3 qmax, int3 off, int8_t* o0, int8_t* o1,
                                                       int8_t* o2, long ldo, int vec) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = C >> 2;  // float4 count
    const T* src = x + row * ldx;
    float v[LN_MAXV][4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        int idx = lane + 64 * k;
        if (idx < nv) {
            qd_ld4(src + idx * 4, vec != 0, v[k]);
#pragma unroll
            for (int j = 0; j < 4; ++j) s += v[k][j];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        int idx = lane + 64 * k;
        if (idx < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { float d = v[k][j] - mean; q += d * d; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    const float d0 = qp0[0], z0 = qp0[1];
    const float d1 = nout > 1 ? qp1[0] : 1.f, z1 = nout > 1 ? qp1[1] : 0.f;
    const float d2 = nout > 2 ? qp2[0] : 1.f, z2 = nout > 2 ? qp2[1] : 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        int idx = lane + 64 * k;
        if (idx < nv) {
            unsigned u0 = 0, u1 = 0, u2 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int c = idx * 4 + j;
                float y = (v[k][j] - mean) * rstd * gamma[c] + beta[c];
                u0 |= (unsigned)((qd_code(y, d0, z0, qmin.x, qmax.x) - off.x) & 0xff) << (8 * j);
                if (nout > 1) u1 |= (unsigned)((qd_code(y, d1, z1, qmin.y, qmax.y) - off.y) & 0xff) << (8 * j);
                if (nout > 2) u2 |= (unsigned)((qd_code(y, d2, z2, qmin.z, qmax.z) - off.z) & 0xff) << (8 * j);
            }
            *reinterpret_cast<unsigned*>(o0 + row * ldo + idx * 4) = u0;
            if (nout > 1) *reinterpret_cast<unsigned*>(o1 + row * ldo + idx * 4) = u1;
            if (nout > 2) *reinterpret_cast<unsigned*>(o2 + row * ldo + idx * 4) = u2;
        }
    }
}

}  // namespace

extern "C" int64_t qd_groupnorm_ws_bytes(int64_t B, int64_t C, int64_t S) {
    int64_t nchunk = (S + GN_ROWS - 1) / GN_ROWS;
    return (B * nchunk * C * 2 + B * C * 2) * (int64_t)sizeof(float);
}

extern "C" int qd_groupnorm_silu_quant(const void* x, int x_dtype, int64_t B, int64_t S, int C, int64_t ldx, int groups,
                                       float eps, const float* gamma, const float* beta, int apply_silu,
                                       const float* qparams, int qmin, int qmax, int off, int8_t* out, int64_t ldo,
                                       float* yout, int64_t ldy, void* ws, void* stream) {
    QD_REQUIRE(x && ws && (out || yout), "qd_groupnorm_silu_quant: null pointer");
    QD_REQUIRE(!out || qparams, "qd_groupnorm_silu_quant: quantised output needs qparams");
    QD_REQUIRE(x_dtype == QD_F32 || x_dtype == QD_F16, "qd_groupnorm_silu_quant: dtype must be f32/f16");
    QD_REQUIRE(B > 0 && S > 0 && C > 0 && groups > 0 && C % groups == 0 && C % 16 == 0, "qd_groupnorm_silu_quant: C=%d must be a multiple of 16 and of groups=%d", C, groups);
    QD_REQUIRE(ldx >= C && (!out || (ldo >= C && ldo % 16 == 0 && qd_aligned(out, 16))), "qd_groupnorm_silu_quant: bad leading dimensions");
    QD_REQUIRE(B < 65536, "qd_groupnorm_silu_quant: batch too large");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nchunk = (int)((S + GN_ROWS - 1) / GN_ROWS);
    const int vec = qd_aligned(x, x_dtype == QD_F32 ? 16 : 8) && ldx % 4 == 0;
    float* part = reinterpret_cast<float*>(ws);
    float* ab = part + (size_t)B * nchunk * C * 2;
    if (x_dtype == QD_F32)
        hipLaunchKernelGGL(gn_partial_kernel<float>, dim3(nchunk, (unsigned)B), dim3(256), 0, st, (const float*)x, (long)S, C, (long)ldx, part, nchunk, vec);
    else
        hipLaunchKernelGGL(gn_partial_kernel<__half>, dim3(nchunk, (unsigned)B), dim3(256), 0, st, (const __half*)x, (long)S, C, (long)ldx, part, nchunk, vec);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, (unsigned)B), dim3(64), 0, st, part, nchunk, (long)S, C, groups, eps, gamma, beta, ab);
    long rows = B * S;
    long total = rows * (C / 4);
    dim3 grid((unsigned)((total + 255) / 256));
    if (x_dtype == QD_F32)
        hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(256), 0, st, (const float*)x, rows, (long)S, C, (long)ldx, ab, apply_silu, qparams, (float)qmin, (float)qmax, off, out, (long)ldo, yout, (long)ldy, vec);
    else
        hipLaunchKernelGGL(gn_apply_kernel<__half>, grid, dim3(256), 0, st, (const __half*)x, rows, (long)S, C, (long)ldx, ab, apply_silu, qparams, (float)qmin, (float)qmax, off, out, (long)ldo, yout, (long)ldy, vec);
    QD_LAUNCH_CHECK("qd_groupnorm_silu_quant");
    return 0;
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
    dim3 grid((unsigned)((M + 3) / 4));
    const int vec = qd_aligned(x, x_dtype == QD_F32 ? 16 : 8) && ldx % 4 == 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (x_dtype == QD_F32)
        hipLaunchKernelGGL(ln_quant_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (long)M, C, (long)ldx, eps, gamma, beta, nout, qp[0], qp[1], qp[2], mn, mx, of, o[0], o[1], o[2], (long)ldo, vec);
    else
        hipLaunchKernelGGL(ln_quant_kernel<__half>, grid, dim3(256), 0, st, (const __half*)x, (long)M, C, (long)ldx, eps, gamma, beta, nout, qp[0], qp[1], qp[2], mn, mx, of, o[0], o[1], o[2], (long)ldo, vec);
    QD_LAUNCH_CHECK("qd_layernorm_quant");
    return 0;
}
