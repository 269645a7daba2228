// boxcal.hip — qd_box_probe: two ~10 ms issue loops that tell how fast THIS box runs the two instruction classes the UNet
// evaluation is bound by, so that a benchmark line can separate box speed from code speed (the same library measured
// 18.8 .. 20.5 ms per SD step across boxes of the pool in round 5, almost all of it in the exp-bound attention kernel):
//   kind 0  dense v_mfma_i32_32x32x32_i8, four independent accumulators per wave, two waves per SIMD — the int8 matrix rate the
//           chip SUSTAINS (power-managed clock), the ceiling of qd_conv2d_i8's K loop;
//   kind 1  v_exp_f32 on four independent registers per wave, four waves per SIMD — the transcendental issue rate that bounds
//           qd_attn_i8's two softmax sweeps.
// Each block's wave 0 reports its elapsed shader-clock ticks (s_memtime); the caller times the launch with HIP events:
// ticks / microseconds = the shader clock under that load.
#include "common.h"

namespace {

template <int KIND>
__global__ __launch_bounds__(256) void box_probe_kernel(int iters, long long* __restrict__ ticks, int* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    v16i acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0;
    v4i a = {lane, lane + 1, lane + 2, lane + 3}, b = {lane * 3, lane * 5, lane * 7, lane * 11};
    float f[4] = {0.001f * (float)lane, 0.002f * (float)lane, -0.003f * (float)lane, -0.004f * (float)lane};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[u & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_exp_f32 %0, %0" : "+v"(f[u & 3]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    int s = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) s += acc[u][lane & 15] + (int)f[u];
    if (s == 0x7fffffff) sink[0] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

}  // namespace

extern "C" int qd_box_probe(int kind, int blocks, int iters, long long* ticks, int* sink, void* stream) {
    QD_REQUIRE((kind == 0 || kind == 1) && blocks > 0 && iters > 0 && ticks && sink, "qd_box_probe: kind 0 / 1, blocks > 0, iters > 0, ticks[blocks] and sink[1] on the device");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (kind == 0) hipLaunchKernelGGL(box_probe_kernel<0>, dim3(blocks), dim3(256), 0, st, iters, ticks, sink);
    else hipLaunchKernelGGL(box_probe_kernel<1>, dim3(blocks), dim3(256), 0, st, iters, ticks, sink);
    QD_LAUNCH_CHECK("qd_box_probe");
    return 0;
}
