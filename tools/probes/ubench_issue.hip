// Probe (measurement only, not part of the library): issue cost of the instructions the attention / contraction kernels are
// made of, on gfx950, as shader cycles per wave-instruction at 1, 2 and 3 waves per SIMD — and how well a VALU stream and an
// MFMA stream of DIFFERENT waves (or of the same wave) overlap on one SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_issue tools/probes/ubench_issue.hip && ./ubench_issue
// Every wave times its own loop with s_memtime (shader clock); the table prints the mean over waves divided by the number of
// instructions one wave issued, i.e. the AVERAGE INTERVAL between two instructions of one wave; the SIMD-level cost of an
// instruction is that number divided by the waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define ITERS 400

enum { T_FMA, T_PKFMA, T_EXP, T_MAX3, T_PERM, T_CVT, T_ADDU, T_DOT4, T_MUL24, T_MFMA32_IND, T_MFMA32_DEP, T_MFMA16_IND, T_MFMA16_DEP,
       T_MIX_ATTN1, T_MIX_MFMA_FMA4, T_MIX_MFMA_FMA8, T_MIX_MFMA_EXP4, T_SPLIT_ROLES, T_LDSR128, T_MIX_MFMA_LDS,
       T_S2_MIX, T_S2_NOEXP, T_S2_NOPK, T_S2_NOPERM, T_S2_ALLFMA, T_S2_VALUONLY, T_S2_MIX_DEP, T_MFMA32K16_IND, T_MFMA32K16_DEP, T_COUNT };

static const char* NAMES[T_COUNT] = {"v_fma_f32 x32", "v_pk_fma_f32 x32", "v_exp_f32 x32", "v_max3_i32 x32", "v_perm_b32 x32", "v_cvt_f32_i32 x32",
    "v_add_u32 x32", "v_dot4_i32_i8 x32", "v_mul_i32_i24 x32", "mfma_i32_32x32x32_i8 x8 (4 independent acc)", "mfma_i32_32x32x32_i8 x8 (1 dependent acc)",
    "mfma_i32_16x16x64_i8 x8 (4 independent acc)", "mfma_i32_16x16x64_i8 x8 (1 dependent acc)",
    "attention sweep-1 tile: 4 mfma32 (dep) + 8x(max3, pk_fma, 2 exp, pk_add)", "4 mfma32 (ind) + 16 v_fma interleaved", "4 mfma32 (ind) + 32 v_fma interleaved",
    "4 mfma32 (ind) + 16 v_exp interleaved", "waves 0,1: 8 mfma32 only | waves 2,3 (other SIMDs) ... see note", "ds_read_b128 x16", "4 mfma32 (ind) + 8 ds_read_b128",
    "sweep-2 slice x4: mfma(ind) + 2 pk_fma, 2 exp, 3 perm, 2 xor", "  ... exp -> v_fma", "  ... pk_fma -> 2 v_fma", "  ... perm/xor -> v_fma",
    "  ... every VALU -> v_fma (9 per mfma)", "  ... the VALU mix without the MFMAs", "  ... the mix with a DEPENDENT mfma chain (one accumulator)",
    "mfma_i32_32x32x16_i8 (gfx942 shape, K = 16) x8 (4 independent acc)", "mfma_i32_32x32x16_i8 (gfx942 shape, K = 16) x8 (1 dependent acc)"};
static const int NINSTR[T_COUNT] = {32, 32, 32, 32, 32, 32, 32, 32, 32, 8, 8, 8, 8, 4 + 8 * 5, 20, 36, 20, 8, 16, 12, 40, 40, 48, 40, 40, 36, 40, 8, 8};

template <int T>
__global__ __launch_bounds__(256) void bench(long long* out, int seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    float f[8];
    int   n[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { f[i] = 0.001f * (lane + i + seed); n[i] = lane * 7 + i + seed; }
    v4i a = {n[0], n[1], n[2], n[3]}, b = {n[4], n[5], n[6], n[7]};
    v16i acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    v4i acc4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    v2f p2[4] = {{f[0], f[1]}, {f[2], f[3]}, {f[4], f[5]}, {f[6], f[7]}};
    for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<int*>(lds)[i] = i;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (T == T_FMA) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u & 7]) : "v"(f[(u + 1) & 7]), "v"(1.0f));
        } else if constexpr (T == T_PKFMA) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2[u & 3]) : "v"(p2[(u + 1) & 3]));
        } else if constexpr (T == T_EXP) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_exp_f32 %0, %0" : "+v"(f[u & 7]));
        } else if constexpr (T == T_MAX3) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(n[u & 7]) : "v"(n[(u + 1) & 7]), "v"(n[(u + 2) & 7]));
        } else if constexpr (T == T_PERM) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(n[u & 7]) : "v"(n[(u + 1) & 7]), "v"(0x05010400));
        } else if constexpr (T == T_CVT) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[u & 7]) : "v"(n[u & 7]));
        } else if constexpr (T == T_ADDU) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[u & 7]) : "v"(n[(u + 1) & 7]));
        } else if constexpr (T == T_DOT4) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(n[u & 7]) : "v"(n[(u + 1) & 7]), "v"(0x01010101));
        } else if constexpr (T == T_MUL24) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(n[u & 7]) : "v"(n[(u + 1) & 7]));
        } else if constexpr (T == T_MFMA32_IND) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[u & 3], 0, 0, 0);
        } else if constexpr (T == T_MFMA32_DEP) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[0], 0, 0, 0);
        } else if constexpr (T == T_MFMA32K16_IND) {
            const long a8 = ((long)a.x << 32) | (unsigned)a.y, b8 = ((long)b.x << 32) | (unsigned)b.y;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, acc[u & 3], 0, 0, 0);
        } else if constexpr (T == T_MFMA32K16_DEP) {
            const long a8 = ((long)a.x << 32) | (unsigned)a.y, b8 = ((long)b.x << 32) | (unsigned)b.y;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[0] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, acc[0], 0, 0, 0);
        } else if constexpr (T == T_MFMA16_IND) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc4[u & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc4[u & 3], 0, 0, 0);
        } else if constexpr (T == T_MFMA16_DEP) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc4[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc4[0], 0, 0, 0);
        } else if constexpr (T == T_MIX_ATTN1) {
            // the shape of attn_lean sweep 1: a dependent MFMA chain on one accumulator, then VALU that consumes it
            v16i c;
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = n[0];
#pragma unroll
            for (int u = 0; u < 4; ++u) c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(n[1]) : "v"(c[r]), "v"(c[r + 1]));
                v2f F = {__int_as_float(c[r]), __int_as_float(c[r + 1])};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(F) : "v"(p2[0]), "v"(p2[1]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(F.x));
                asm volatile("v_exp_f32 %0, %0" : "+v"(F.y));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2[2]) : "v"(F));
            }
        } else if constexpr (T == T_MIX_MFMA_FMA4 || T == T_MIX_MFMA_FMA8 || T == T_MIX_MFMA_EXP4) {
            constexpr int PER = T == T_MIX_MFMA_FMA8 ? 8 : 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[u], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < PER; ++v) {
                    if constexpr (T == T_MIX_MFMA_EXP4) asm volatile("v_exp_f32 %0, %0" : "+v"(f[v & 7]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[v & 7]) : "v"(f[(v + 1) & 7]), "v"(1.0f));
                }
            }
        } else if constexpr (T == T_SPLIT_ROLES) {
            // blocks alternate roles by parity: even blocks issue only MFMAs, odd blocks only v_exp — with >= 2 blocks per CU a
            // SIMD hosts one wave of each kind: do the two pipes run concurrently across waves?
            if (blockIdx.x & 1) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    asm volatile("v_exp_f32 %0, %0" : "+v"(f[u & 7]));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[(u + 3) & 7]) : "v"(f[(u + 1) & 7]), "v"(1.0f));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[(u + 4) & 7]) : "v"(f[(u + 2) & 7]), "v"(1.0f));
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[u & 3], 0, 0, 0);
            }
        } else if constexpr (T >= T_S2_MIX && T <= T_S2_MIX_DEP) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (T == T_S2_MIX_DEP) acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[0], 0, 0, 0);
                else if constexpr (T != T_S2_VALUONLY) acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[u], 0, 0, 0);
                // 2 pk_fma
                if constexpr (T == T_S2_NOPK || T == T_S2_ALLFMA) {
#pragma unroll
                    for (int v = 0; v < (T == T_S2_NOPK ? 4 : 2); ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[v & 7]) : "v"(f[(v + 1) & 7]), "v"(1.0f));
                } else {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2[0]) : "v"(p2[1]));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2[2]) : "v"(p2[3]));
                }
                // 2 exp
                if constexpr (T == T_S2_NOEXP || T == T_S2_ALLFMA) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[4]) : "v"(f[5]), "v"(1.0f));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[6]) : "v"(f[7]), "v"(1.0f));
                } else {
                    asm volatile("v_exp_f32 %0, %0" : "+v"(f[4]));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(f[6]));
                }
                // 3 perm + 2 xor
                if constexpr (T == T_S2_NOPERM || T == T_S2_ALLFMA) {
#pragma unroll
                    for (int v = 0; v < 5; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[v & 3]) : "v"(f[(v + 1) & 3]), "v"(1.0f));
                } else {
                    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(n[0]) : "v"(n[1]), "v"(0x05010400));
                    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(n[2]) : "v"(n[3]), "v"(0x05010400));
                    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(n[4]) : "v"(n[5]), "v"(0x05040100));
                    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(n[6]) : "v"(0x80808080));
                    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(n[7]) : "v"(0x80808080));
                }
            }
        } else if constexpr (T == T_LDSR128) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                v4i t = *reinterpret_cast<const volatile v4i*>(lds + ((lane * 16 + u * 1024) & 16383));
                n[u & 7] += t.x;
            }
        } else if constexpr (T == T_MIX_MFMA_LDS) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[u], 0, 0, 0);
                v4i t0 = *reinterpret_cast<const volatile v4i*>(lds + ((lane * 16 + u * 2048) & 16383));
                v4i t1 = *reinterpret_cast<const volatile v4i*>(lds + ((lane * 16 + u * 2048 + 1024) & 16383));
                n[u & 7] += t0.x + t1.y;
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    int sink = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) sink += n[i] + (int)f[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) sink += acc[i][lane & 15] + acc4[i][lane & 3] + (int)p2[i].x;
    if (sink == 0x7fffffff) out[0] = sink;
    if (lane == 0) out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int T>
void run(long long* dout, std::vector<long long>& h) {
    for (int occ = 1; occ <= 3; ++occ) {
        const int lds_bytes = occ == 1 ? 100 * 1024 : (occ == 2 ? 70 * 1024 : 50 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(bench<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        const int nblk = 256 * occ;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(bench<T>, dim3(nblk), dim3(256), lds_bytes, 0, dout, rep);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), dout, sizeof(long long) * (1 + nblk * 4), hipMemcpyDeviceToHost);
        double sum = 0, mx = 0;
        double se = 0, so = 0;
        for (int i = 0; i < nblk * 4; ++i) {
            sum += (double)h[1 + i];
            if ((double)h[1 + i] > mx) mx = (double)h[1 + i];
            if ((i / 4) & 1) so += (double)h[1 + i]; else se += (double)h[1 + i];
        }
        const double per = sum / (nblk * 4) / ITERS;
        if (T == T_SPLIT_ROLES)
            printf("%-74s occ %d: MFMA-only waves %7.1f ticks/iter (8 mfma), VALU-only waves %7.1f ticks/iter (8 exp + 16 fma)\n", NAMES[T], occ,
                   se / (nblk * 2) / ITERS, so / (nblk * 2) / ITERS);
        else
            printf("%-74s occ %d: %7.1f ticks/iter/wave  = %6.2f ticks per instruction of a wave, %6.2f per SIMD slot\n", NAMES[T], occ, per,
                   per / NINSTR[T], per / NINSTR[T] / occ);
    }
}

int main(int argc, char** argv) {
    const bool only_s2 = argc > 1 && std::string(argv[1]) == "s2";
    const bool only_mfma = argc > 1 && std::string(argv[1]) == "mfma";
    long long* dout;
    hipMalloc(&dout, sizeof(long long) * (1 + 256 * 3 * 4));
    std::vector<long long> h(1 + 256 * 3 * 4);
    // calibrate the tick of __builtin_readcyclecounter against wall time
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipFuncSetAttribute(reinterpret_cast<const void*>(bench<T_MFMA32_IND>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        hipLaunchKernelGGL(bench<T_MFMA32_IND>, dim3(256), dim3(256), 100 * 1024, 0, dout, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(bench<T_MFMA32_IND>, dim3(256), dim3(256), 100 * 1024, 0, dout, 0);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), dout, sizeof(long long) * 5, hipMemcpyDeviceToHost);
        printf("calibration: kernel wall %.1f us, wave ticks %lld -> %.1f ticks/us; %d mfma32 per wave -> %.1f ns per MFMA\n", ms * 1000.0, h[1],
               (double)h[1] / (ms * 1000.0), ITERS * 8, ms * 1e6 / (ITERS * 8));
    }
    if (only_mfma) {
        run<T_MFMA32_IND>(dout, h); run<T_MFMA32_DEP>(dout, h); run<T_MFMA32K16_IND>(dout, h); run<T_MFMA32K16_DEP>(dout, h);
        run<T_MFMA16_IND>(dout, h); run<T_MFMA16_DEP>(dout, h);
        return 0;
    }
    if (!only_s2) {
    run<T_FMA>(dout, h); run<T_PKFMA>(dout, h); run<T_EXP>(dout, h); run<T_MAX3>(dout, h); run<T_PERM>(dout, h); run<T_CVT>(dout, h);
    run<T_ADDU>(dout, h); run<T_DOT4>(dout, h); run<T_MUL24>(dout, h);
    run<T_MFMA32_IND>(dout, h); run<T_MFMA32_DEP>(dout, h); run<T_MFMA16_IND>(dout, h); run<T_MFMA16_DEP>(dout, h);
    run<T_MIX_ATTN1>(dout, h); run<T_MIX_MFMA_FMA4>(dout, h); run<T_MIX_MFMA_FMA8>(dout, h); run<T_MIX_MFMA_EXP4>(dout, h);
    run<T_SPLIT_ROLES>(dout, h); run<T_LDSR128>(dout, h); run<T_MIX_MFMA_LDS>(dout, h);
    }
    run<T_S2_MIX>(dout, h); run<T_S2_NOEXP>(dout, h); run<T_S2_NOPK>(dout, h); run<T_S2_NOPERM>(dout, h); run<T_S2_ALLFMA>(dout, h);
    run<T_S2_VALUONLY>(dout, h); run<T_S2_MIX_DEP>(dout, h);
    return 0;
}
