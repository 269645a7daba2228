// common.h — shared helpers for libqdiff_hip.so (gfx950 only; no portability shims).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/qdiff_hip.h"

typedef int   v4i  __attribute__((ext_vector_type(4)));
typedef int   v16i __attribute__((ext_vector_type(16)));
typedef float v4f  __attribute__((ext_vector_type(4)));

void qd_set_error(const char* fmt, ...);

#define QD_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            qd_set_error(__VA_ARGS__);        \
            return 1;                         \
        }                                     \
    } while (0)

#define QD_LAUNCH_CHECK(name)                                                       \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            qd_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
            return 2;                                                               \
        }                                                                           \
    } while (0)

static inline bool qd_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// MI355X: 8 XCDs, block b is dispatched to XCD b % 8 (observed; used for L2 locality only).
// Bijective remap so that consecutive logical ids share an XCD (cdna_hip_programming.md §5 T1).
__device__ __forceinline__ int qd_xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int xcd = bid % nx, idx = bid / nx;
    int q = nwg / nx, r = nwg % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// load a float-like element and widen
template <typename T> __device__ __forceinline__ float qd_ld(const T* p);
template <> __device__ __forceinline__ float qd_ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float qd_ld<__half>(const __half* p) { return __half2float(*p); }

// 4 consecutive elements; `vec` = the caller proved 4-element alignment (one 16-/8-byte load)
template <typename T>
__device__ __forceinline__ void qd_ld4(const T* p, bool vec, float (&v)[4]) {
    if (vec) {
        if constexpr (sizeof(T) == 4) {
            const float4 f = *reinterpret_cast<const float4*>(p);
            v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
        } else {
            const uint2 u = *reinterpret_cast<const uint2*>(p);
            const __half2 a = *reinterpret_cast<const __half2*>(&u.x), b = *reinterpret_cast<const __half2*>(&u.y);
            v[0] = __low2float(a); v[1] = __high2float(a); v[2] = __low2float(b); v[3] = __high2float(b);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = qd_ld(p + j);
    }
}

// erf with < 1 ulp polynomial error on both ranges, evaluated branch-free (both ranges, then a select):
// the GEGLU epilogue is VALU-bound and the library erff's divergent branches cost ~2x this
// (coefficients checked against scipy.special.erf over [-6, 6]: max error 0.97 ulp).
//   |a| <= 0.9277: a + a*P(a^2);   else: sign(a) * (1 - exp(Q(|a|)))
__device__ __forceinline__ float qd_erff(float a) {
    const float t = __builtin_fabsf(a), s = a * a;
    float r = __builtin_fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = __builtin_fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = __builtin_fmaf(r, s, u);
    r = __builtin_fmaf(r, t, -1.06777877e-1f);
    r = __builtin_fmaf(r, t, -6.34846687e-1f);
    r = __builtin_fmaf(r, t, -1.28717512e-1f);
    r = __builtin_fmaf(r, t, -t);
    const float big = __builtin_copysignf(1.0f - __builtin_amdgcn_exp2f(r * 1.4426950408889634f), a);
    float q = -5.96761703e-4f;
    q = __builtin_fmaf(q, s, 4.99119423e-3f);
    q = __builtin_fmaf(q, s, -2.67681349e-2f);
    q = __builtin_fmaf(q, s, 1.12819925e-1f);
    q = __builtin_fmaf(q, s, -3.76125336e-1f);
    q = __builtin_fmaf(q, s, 1.28379166e-1f);
    const float small = __builtin_fmaf(q, a, a);
    return t > 0.927734375f ? big : small;
}

// Quantiser parameters as the kernels receive them: device float[4] = {delta, zero_point, rinv, fast} written by
// qd_make_qparams.  `fast` != 0 certifies (exhaustively, over all 2^23 mantissas of x) that the three-instruction
// quotient  y = x*rinv;  e = fma(-y, delta, x);  q = fma(e, rinv, y)   equals the IEEE division x / delta BIT FOR BIT for
// this delta (Markstein's correction: rinv is the correctly rounded reciprocal, e is exact); the kernels then skip the
// ~10-instruction division sequence that torch.round(x / delta) semantics (quant_layer.py:82) otherwise cost per element.
struct QP { float delta, zp, rinv; bool fast; };
__device__ __forceinline__ QP qd_load_qp(const float* q) { return QP{q[0], q[1], q[2], q[3] != 0.f}; }
__device__ __forceinline__ float qd_quot(float x, const QP& q) {
    if (q.fast) {
        const float y = x * q.rinv;
        const float e = __builtin_fmaf(-y, q.delta, x);
        return __builtin_fmaf(e, q.rinv, y);
    }
    return x / q.delta;
}
__device__ __forceinline__ int qd_code(float x, const QP& q, float qmin, float qmax) {
    float r = rintf(qd_quot(x, q)) + q.zp;
    r = fminf(fmaxf(r, qmin), qmax);
    return (int)r;
}
// The same with the fast / exact choice as a COMPILE-TIME flag.  The run-time form above puts a (uniform) branch around
// the division of every element, which splits an unrolled epilogue into one basic block per element and serialises its
// dependency chains (seen in the ISA of the GEGLU epilogue: 167 us of a 234 us launch); kernels therefore test q.fast once
// and run a branch-free body:  QD_FAST_DISPATCH(q.fast, body)  with  body = [&](auto fast_tag) { ... qd_code_t<FAST>(...) }.
template <bool FAST>
__device__ __forceinline__ int qd_code_t(float x, const QP& q, float qmin, float qmax) {
    float d;
    if constexpr (FAST) {
        const float y = x * q.rinv;
        d = __builtin_fmaf(__builtin_fmaf(-y, q.delta, x), q.rinv, y);
    } else {
        d = x / q.delta;
    }
    float r = rintf(d) + q.zp;
    r = fminf(fmaxf(r, qmin), qmax);
    return (int)r;
}
#define QD_FAST_DISPATCH(flag, body)                     \
    do {                                                 \
        if (flag) body(std::true_type{});                \
        else body(std::false_type{});                    \
    } while (0)

// quantise one value to its stored byte: clamp(rint(x/delta)+zp, qmin, qmax) - off
// (true IEEE division + round-half-even, as torch.round(x / delta): quant_layer.py:82)
__device__ __forceinline__ int qd_code(float x, float delta, float zp, float qmin, float qmax) {
    float r = rintf(x / delta) + zp;
    r = fminf(fmaxf(r, qmin), qmax);
    return (int)r;
}
