// 1200-point mixed-radix FFT building blocks shared by the feature kernels (features.hip: FOA log-mel + intensity vector;
// features_mic.hip: GCC-PHAT): radix 10 x 10 x 12 as in-register (5x2), (5x2), (4x3) composite butterflies, and the LDS image
// of the in-place decimation-in-frequency transform.
#pragma once
#include <hip/hip_runtime.h>

namespace adyolo {

constexpr int FN = 1200, FBINS = 601, FHOP = 600;
// the twiddle table handed in by the caller: [0, 1200) exp(-2 pi i n / 1200); then the entries of passes 1 and 2 in the order
// their lanes read them -- FTW1 + (k-1) 120 + st = W^{st k} (k = 1..9, st < 120) and FTW2 + (k-1) 12 + n3 = W^{10 n3 k}: a
// wave's load of one k touches 4 cache lines instead of up to 36 (the gather tw[(st k) mod 1200] kept the CU's address unit busy
// for a third of the kernel)
constexpr int FTW1 = FN, FTW2 = FN + 9 * 120, FTW_LEN = FN + 9 * 120 + 9 * 12;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmi(float2 a) { return make_float2(a.y, -a.x); }      // -i * a
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }

template <int R>
__device__ __forceinline__ void butterfly(float2 *v);
template <>
__device__ __forceinline__ void butterfly<4>(float2 *v) {
    const float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = cmi(csub(v[1], v[3]));
    v[0] = cadd(a, c); v[1] = cadd(b, d); v[2] = csub(a, c); v[3] = csub(b, d);
}
template <>
__device__ __forceinline__ void butterfly<3>(float2 *v) {
    const float2 t = cadd(v[1], v[2]);
    const float2 q = cscale(cmi(csub(v[1], v[2])), 0.86602540378443864676f);
    const float2 m = csub(v[0], cscale(t, 0.5f));
    v[0] = cadd(v[0], t); v[1] = cadd(m, q); v[2] = csub(m, q);
}
template <>
__device__ __forceinline__ void butterfly<5>(float2 *v) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    const float2 a1 = make_float2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
    const float2 a2 = make_float2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
    const float2 b1 = cmi(make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
    const float2 b2 = cmi(make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
    v[0] = cadd(v[0], cadd(t1, t2));
    v[1] = cadd(a1, b1); v[4] = csub(a1, b1);
    v[2] = cadd(a2, b2); v[3] = csub(a2, b2);
}

template <>
__device__ __forceinline__ void butterfly<2>(float2 *v) {
    const float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}

// exp(-2 pi i m / R) tables for the in-register composite butterflies (folded to immediates after unrolling)
__device__ constexpr float RC10[10] = {1.f, 0.809016994f, 0.309016994f, -0.309016994f, -0.809016994f, -1.f, -0.809016994f, -0.309016994f, 0.309016994f, 0.809016994f};
__device__ constexpr float RS10[10] = {0.f, -0.587785252f, -0.951056516f, -0.951056516f, -0.587785252f, 0.f, 0.587785252f, 0.951056516f, 0.951056516f, 0.587785252f};
__device__ constexpr float RC12[12] = {1.f, 0.866025404f, 0.5f, 0.f, -0.5f, -0.866025404f, -1.f, -0.866025404f, -0.5f, 0.f, 0.5f, 0.866025404f};
__device__ constexpr float RS12[12] = {0.f, -0.5f, -0.866025404f, -1.f, -0.866025404f, -0.5f, 0.f, 0.5f, 0.866025404f, 1.f, 0.866025404f, 0.5f};
template <int R>
__device__ __forceinline__ float2 root(int m);
template <>
__device__ __forceinline__ float2 root<10>(int m) { return make_float2(RC10[m], RS10[m]); }
template <>
__device__ __forceinline__ float2 root<12>(int m) { return make_float2(RC12[m], RS12[m]); }

// radix R1*R2 butterfly entirely in registers (Cooley-Tukey: n = n2 + R2 n1, k = k1 + R1 k2)
template <int R1, int R2>
__device__ __forceinline__ void butterfly_composite(float2 *v) {
    constexpr int R = R1 * R2;
    float2 y[R2][R1];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) {
        float2 t[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) t[n1] = v[n2 + R2 * n1];
        butterfly<R1>(t);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) y[n2][k1] = (n2 * k1 == 0) ? t[k1] : cmul(t[k1], root<R>((n2 * k1) % R));
    }
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
        float2 t[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) t[n2] = y[n2][k1];
        butterfly<R2>(t);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = t[k2];
    }
}
template <>
__device__ __forceinline__ void butterfly<10>(float2 *v) { butterfly_composite<5, 2>(v); }
template <>
__device__ __forceinline__ void butterfly<12>(float2 *v) { butterfly_composite<4, 3>(v); }

// LDS image of one packed signal: position p = k1*120 + k2*12 + k3 lives at p + 2*(p/12) = k1*140 + k2*14 + k3 (complex
// units): rows of 12 padded to 14, so the twelve-point stage's float4 reads step 28 dwords from lane to lane (every
// 16-lane ds_read_b128 group hits 16 distinct bank quads)
constexpr int FROW = 14, FSIG = 10 * 10 * FROW;          // 1400 complex per signal
__device__ __forceinline__ int fpos(int k) {             // where bin k ends up after the three in-place passes
    const int k1 = k % 10, q = k / 10;
    return k1 * (10 * FROW) + (q % 10) * FROW + q / 10;
}


}  // namespace adyolo
