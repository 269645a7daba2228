// fakequant.hip — fused fake-quantisation forward / backward for CALIBRATION.
//
// The activation phase of block / layer reconstruction (reference qdiff/block_recon.py:72-110) trains the step sizes
// `delta` of the activation quantisers through autograd on the simulation graph
//     codes = clamp(round_ste(x / delta) + zp, lo, hi);   y = (codes - zp) * delta            (quant_layer.py:82-88)
// which PyTorch runs as 6 elementwise kernels forward and ~12 backward over every quantised activation of the unit.
// Here: one kernel forward (4 B read, 4 B written per element) and one backward (8 B read, 4 B written) that also
// produces the block-level partial sums of d(loss)/d(delta) (deterministic two-level reduction, no float atomics).
// The arithmetic follows the autograd graph operation by operation:
//     forward:   d = x / delta (IEEE division), r = rint(d) + zp, q = min(max(r, lo), hi), y = (q - zp) * delta
//     backward:  t = g * delta; t = (lo <= r && r <= hi) ? t : 0 (clamp passes the gradient on the closed interval);
//                gx = t / delta;                        (round_ste is the identity, d(x/delta)/dx = 1/delta)
//                gdelta = sum( g * (q - zp) - t * ((x / delta) / delta) )      (ATen's div backward for the divisor)
// so that y and gx are bit-identical to the composition and gdelta differs by summation order only.
#include "common.h"

namespace {

constexpr int FQ_VEC = 4;

__global__ __launch_bounds__(256) void fakequant_fwd_kernel(const float* __restrict__ x, long n, const float* __restrict__ delta,
                                                            const float* __restrict__ zpp, float lo, float hi, float* __restrict__ y) {
    const float d = delta[0], zp = zpp[0];
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * FQ_VEC;
    if (i0 >= n) return;
    if (i0 + FQ_VEC <= n) {
        const float4 v = *reinterpret_cast<const float4*>(x + i0);
        float in[4] = {v.x, v.y, v.z, v.w}, out[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float r = rintf(in[j] / d) + zp;
            out[j] = (fminf(fmaxf(r, lo), hi) - zp) * d;
        }
        *reinterpret_cast<float4*>(y + i0) = make_float4(out[0], out[1], out[2], out[3]);
    } else {
        for (long i = i0; i < n; ++i) {
            const float r = rintf(x[i] / d) + zp;
            y[i] = (fminf(fmaxf(r, lo), hi) - zp) * d;
        }
    }
}

__device__ __forceinline__ void fq_bwd_one(float x, float g, float d, float zp, float lo, float hi, float& gx, float& gd) {
    const float dv = x / d;
    const float r = rintf(dv) + zp;
    const float q = fminf(fmaxf(r, lo), hi);
    float t = g * d;
    t = (r >= lo && r <= hi) ? t : 0.f;
    gx = t / d;
    gd += g * (q - zp) - t * (dv / d);
}

__global__ __launch_bounds__(256) void fakequant_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, long n,
                                                            const float* __restrict__ delta, const float* __restrict__ zpp, float lo,
                                                            float hi, float* __restrict__ gx, float* __restrict__ part) {
    __shared__ float red[4];
    const float d = delta[0], zp = zpp[0];
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * FQ_VEC;
    float gd = 0.f;
    if (i0 + FQ_VEC <= n) {
        const float4 v = *reinterpret_cast<const float4*>(x + i0), g = *reinterpret_cast<const float4*>(gy + i0);
        float o[4];
        fq_bwd_one(v.x, g.x, d, zp, lo, hi, o[0], gd);
        fq_bwd_one(v.y, g.y, d, zp, lo, hi, o[1], gd);
        fq_bwd_one(v.z, g.z, d, zp, lo, hi, o[2], gd);
        fq_bwd_one(v.w, g.w, d, zp, lo, hi, o[3], gd);
        *reinterpret_cast<float4*>(gx + i0) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        for (long i = i0; i < n; ++i) {
            float o;
            fq_bwd_one(x[i], gy[i], d, zp, lo, hi, o, gd);
            gx[i] = o;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gd += __shfl_xor(gd, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gd;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

extern "C" int64_t qd_fakequant_blocks(int64_t n) { return (n + 256 * FQ_VEC - 1) / (256 * FQ_VEC); }

extern "C" int qd_fakequant_fwd(const float* x, int64_t n, const float* delta, const float* zero_point, int qmin, int qmax, float* y,
                                void* stream) {
    QD_REQUIRE(x && y && delta && zero_point && n > 0, "qd_fakequant_fwd: null pointer / empty tensor");
    QD_REQUIRE(qd_aligned(x, 16) && qd_aligned(y, 16), "qd_fakequant_fwd: tensors must be 16-byte aligned");
    hipLaunchKernelGGL(fakequant_fwd_kernel, dim3((unsigned)qd_fakequant_blocks(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x, (long)n, delta, zero_point, (float)qmin, (float)qmax, y);
    QD_LAUNCH_CHECK("qd_fakequant_fwd");
    return 0;
}

extern "C" int qd_fakequant_bwd(const float* x, const float* gy, int64_t n, const float* delta, const float* zero_point, int qmin,
                                int qmax, float* gx, float* gdelta_part, void* stream) {
    QD_REQUIRE(x && gy && gx && gdelta_part && delta && zero_point && n > 0, "qd_fakequant_bwd: null pointer / empty tensor");
    QD_REQUIRE(qd_aligned(x, 16) && qd_aligned(gy, 16) && qd_aligned(gx, 16), "qd_fakequant_bwd: tensors must be 16-byte aligned");
    hipLaunchKernelGGL(fakequant_bwd_kernel, dim3((unsigned)qd_fakequant_blocks(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x, gy, (long)n, delta, zero_point, (float)qmin, (float)qmax, gx, gdelta_part);
    QD_LAUNCH_CHECK("qd_fakequant_bwd");
    return 0;
}
