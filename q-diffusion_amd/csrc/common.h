// common.h — shared helpers for libqdiff_hip.so (gfx950 only; no portability shims).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
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

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

// four bf16 <-> four floats through one 8-byte access (first-stage decoder: bf16 activations, fp32 accumulators).
// bf16 -> fp32 is a 16-bit shift; fp32 -> bf16 rounds to nearest even (v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ v4f qd_ld4bf(const void* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return v4f{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
}
__device__ __forceinline__ unsigned qd_pack2bf(float a, float b) {
    const __hip_bfloat16 x = __float2bfloat16(a), y = __float2bfloat16(b);
    return (unsigned)(*reinterpret_cast<const unsigned short*>(&x)) | ((unsigned)(*reinterpret_cast<const unsigned short*>(&y)) << 16);
}
__device__ __forceinline__ void qd_st4bf(void* p, const v4f& v) {
    uint2 u;
    u.x = qd_pack2bf(v[0], v[1]);
    u.y = qd_pack2bf(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ float qd_bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// four halves <-> four floats through one 8-byte access (fp16 activation streams)
// two floats -> two IEEE halves in one word (round to nearest even), low half first
__device__ __forceinline__ unsigned qd_pack2h(float a, float b) {
    const __half2 h = __floats2half2_rn(a, b);
    return *reinterpret_cast<const unsigned*>(&h);
}
__device__ __forceinline__ v4f qd_ld4h(const __half* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    const __half2 a = *reinterpret_cast<const __half2*>(&u.x), b = *reinterpret_cast<const __half2*>(&u.y);
    const float2 fa = __half22float2(a), fb = __half22float2(b);
    return v4f{fa.x, fa.y, fb.x, fb.y};
}
__device__ __forceinline__ void qd_st4h(__half* p, const v4f& v) {
    const __half2 a = __floats2half2_rn(v[0], v[1]), b = __floats2half2_rn(v[2], v[3]);
    uint2 u;
    u.x = *reinterpret_cast<const unsigned*>(&a);
    u.y = *reinterpret_cast<const unsigned*>(&b);
    *reinterpret_cast<uint2*>(p) = u;
}

// eight halves (one 16-byte access) <-> two float quads: the 16-byte lanes of the fp16 activation stream
__device__ __forceinline__ float2 qd_h2_to_f(int w) {
    const unsigned u = (unsigned)w;
    return __half22float2(*reinterpret_cast<const __half2*>(&u));
}
__device__ __forceinline__ void qd_h8_to_f(const v4i& u, v4f& lo, v4f& hi) {
    const float2 a = qd_h2_to_f(u.x), b = qd_h2_to_f(u.y), c = qd_h2_to_f(u.z), d = qd_h2_to_f(u.w);
    lo = v4f{a.x, a.y, b.x, b.y};
    hi = v4f{c.x, c.y, d.x, d.y};
}
__device__ __forceinline__ void qd_ld8h(const __half* p, float (&v)[8]) {
    const v4i u = *reinterpret_cast<const v4i*>(p);
    v4f lo, hi;
    qd_h8_to_f(u, lo, hi);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
}

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
// ---- packed forms for the VALU-bound epilogues (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 lanes per issue) ----
// Same IEEE operations in the same order as the scalar forms above, so the codes are those of qd_code_t bit for bit.
// The byte itself comes from one float add instead of rint + add + min + max + cvt + sub: clamping BEFORE rounding is the
// same function when the bounds are integers (qmin - zp, qmax - zp; zero points are integers, checked by the host), and
// adding 1.5*2^23 rounds half-to-even into the low mantissa bits; (zp - off) is then added as an INTEGER (folding it into
// the float constant would send ties to the even SUM, i.e. to the odd code whenever zp - off is odd), which leaves the
// two's complement of (code - off) in the low byte.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f qd_splat2(float a) { return v2f{a, a}; }
__device__ __forceinline__ v2f qd_fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
struct QB { float lo, hi; int n; };
__device__ __forceinline__ QB qd_bytes_setup(const QP& q, float qmin, float qmax, int off) {
    return QB{qmin - q.zp, qmax - q.zp, (int)q.zp - off};
}
template <bool FAST>
__device__ __forceinline__ v2f qd_quot2_t(v2f x, const QP& q) {
    if constexpr (FAST) {
        const v2f r = qd_splat2(q.rinv), dl = qd_splat2(q.delta);
        const v2f y = x * r;
        return qd_fma2(qd_fma2(-y, dl, x), r, y);
    } else {
        return v2f{x.x / q.delta, x.y / q.delta};
    }
}
// two values -> two ints whose LOW BYTES are the stored codes clamp(rint(x / delta) + zp, qmin, qmax) - off
template <bool FAST>
__device__ __forceinline__ void qd_bytes2_t(v2f x, const QP& q, const QB& b, int& b0, int& b1) {
    const v2f d = qd_quot2_t<FAST>(x, q);
    v2f c = {__builtin_amdgcn_fmed3f(d.x, b.lo, b.hi), __builtin_amdgcn_fmed3f(d.y, b.lo, b.hi)};
    c += qd_splat2(12582912.f);
    b0 = __float_as_int(c.x) + b.n;
    b1 = __float_as_int(c.y) + b.n;
}
// four values -> the four stored code bytes of ONE quantiser in a word (byte e = value e): qd_code_t's bytes bit for bit, at
// ~5 instructions per value instead of ~12 (the producers — GroupNorm apply, LayerNorm with up to three quantisers — are
// VALU-bound once the activation stream is fp16, and sit at the edge of it in fp32: profiles/r05_streams_ab.md)
template <bool FAST>
__device__ __forceinline__ unsigned qd_pack4_t(float y0, float y1, float y2, float y3, const QP& q, const QB& b) {
    int b0, b1, b2, b3;
    qd_bytes2_t<FAST>(v2f{y0, y1}, q, b, b0, b1);
    qd_bytes2_t<FAST>(v2f{y2, y3}, q, b, b2, b3);
    return __builtin_amdgcn_perm(__builtin_amdgcn_perm((unsigned)b3, (unsigned)b2, 0x0c0c0400u),
                                 __builtin_amdgcn_perm((unsigned)b1, (unsigned)b0, 0x0c0c0400u), 0x05040100u);
}
__device__ __forceinline__ v2f qd_erff2(v2f a) {             // qd_erff on two values (same operations, packed)
    const v2f t = {__builtin_fabsf(a.x), __builtin_fabsf(a.y)}, s = a * a;
    v2f r = qd_fma2(qd_splat2(-1.72853470e-5f), t, qd_splat2(3.83197126e-4f));
    const v2f u = qd_fma2(qd_splat2(-3.88396438e-3f), t, qd_splat2(2.42546219e-2f));
    r = qd_fma2(r, s, u);
    r = qd_fma2(r, t, qd_splat2(-1.06777877e-1f));
    r = qd_fma2(r, t, qd_splat2(-6.34846687e-1f));
    r = qd_fma2(r, t, qd_splat2(-1.28717512e-1f));
    r = qd_fma2(r, t, -t);
    const v2f rl = r * qd_splat2(1.4426950408889634f);
    const v2f om = qd_splat2(1.0f) - v2f{__builtin_amdgcn_exp2f(rl.x), __builtin_amdgcn_exp2f(rl.y)};
    const v2f big = {__builtin_copysignf(om.x, a.x), __builtin_copysignf(om.y, a.y)};
    v2f q = qd_splat2(-5.96761703e-4f);
    q = qd_fma2(q, s, qd_splat2(4.99119423e-3f));
    q = qd_fma2(q, s, qd_splat2(-2.67681349e-2f));
    q = qd_fma2(q, s, qd_splat2(1.12819925e-1f));
    q = qd_fma2(q, s, qd_splat2(-3.76125336e-1f));
    q = qd_fma2(q, s, qd_splat2(1.28379166e-1f));
    const v2f small = qd_fma2(q, a, a);
    return v2f{t.x > 0.927734375f ? big.x : small.x, t.y > 0.927734375f ? big.y : small.y};
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
