// EXPLORATORY micro-benchmark (round 3; nothing in libadyolo_hip.so uses it): could the fp32 GEMM-shaped work
// of this path run on the bf16 matrix pipe with split operands and still be "fp32"?
//   x = hi + mid + lo  (three bf16 terms, 24 mantissa bits),  a*b ~ hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid
// Six v_mfma_f32_32x32x16_bf16 (fp32 accumulate) stand for eight v_mfma_f32_32x32x2_f32.  Reported:
//   1. rates of the two MFMA pipes from register operands (all CUs, 2 waves per SIMD, 4 independent accumulators);
//   2. the rate of the six-product form INCLUDING the operand split (VALU), with the A fragment reused for R = 1 and R = 4 B
//      fragments, as "fp32-equivalent" TFLOP/s;
//   3. accuracy on C = A B (M = N = 128, K = 4096, values ~ N(0,1)): max |error| / (sum_k |a||b|) against a float64 sum for
//      the fp32 MFMA, the six-product split, a three-product split (two terms, 16 bits) and plain bf16.
// build + run (GPU box): hipcc -O3 --offload-arch=gfx950 tools/micro/split_bf16_gemm.hip -o /tmp/split_bf16 && /tmp/split_bf16
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &hi, bf16x8 &mid, bf16x8 &lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float r1 = x[i] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        hi[i] = h;
        mid[i] = m;
        lo[i] = (__bf16)r2;
    }
}

// ---------------------------------------------------------------------------------------------- rates
__global__ __launch_bounds__(256) void rate_bf16(float *out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(threadIdx.x * 0.001f + i);
        b[i] = (__bf16)(1.0f - i * 0.01f);
    }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ __launch_bounds__(256) void rate_f32(float *out, int iters) {
    float a = threadIdx.x * 0.001f, b = 1.0f;
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

__device__ __forceinline__ f32x16 mma6(bf16x8 ah, bf16x8 am, bf16x8 al, bf16x8 bh, bf16x8 bm, bf16x8 bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);      // smallest terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    return c;
}

// one K = 16 block per iteration: fresh fp32 operand registers (perturbed so nothing is hoisted), split on the VALU, 6 MFMAs
// per (A, B) fragment pair; R B fragments share one A fragment
template <int R>
__global__ __launch_bounds__(256) void rate_split6(float *out, int iters) {
    float xa[8], xb[R][8];
    for (int i = 0; i < 8; ++i) {
        xa[i] = threadIdx.x * 0.37f + i;
        for (int r = 0; r < R; ++r) xb[r][i] = 1.0f + 0.01f * i + r;
    }
    f32x16 c[R];
    for (int r = 0; r < R; ++r) c[r] = (f32x16){0};
    for (int it = 0; it < iters; ++it) {
        bf16x8 ah, am, al;
        split3(xa, ah, am, al);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            bf16x8 bh, bm, bl;
            split3(xb[r], bh, bm, bl);
            c[r] = mma6(ah, am, al, bh, bm, bl, c[r]);
#pragma unroll
            for (int i = 0; i < 8; ++i) xb[r][i] += 0.5f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) xa[i] += 0.25f;
    }
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += c[r][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------- accuracy
// one wave per 32 x 32 tile of C = A (M x K, row-major) B (K x N, row-major).  Fragment layouts assumed for 32x32x16:
// A: lane l holds A[l & 31][k0 + 8 (l >> 5) + i], B: lane l holds B[k0 + 8 (l >> 5) + i][l & 31], i = 0..7
// (checked by the plain-bf16 mode: a wrong layout gives O(1) errors, not 1e-2)
template <int MODE>       // 0: fp32 MFMA, 1: six-product split, 2: three-product split (hi/lo of two terms), 3: plain bf16
__global__ __launch_bounds__(64) void gemm_tile(const float *A, const float *B, float *C, int M, int N, int K) {
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    f32x16 c = {0};
    if (MODE == 0) {
        for (int k = 0; k < K; k += 2)
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)(m0 + li) * K + k + lh], B[(size_t)(k + lh) * N + n0 + li], c, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 16) {
            float xa[8], xb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xa[i] = A[(size_t)(m0 + li) * K + k0 + 8 * lh + i];
                xb[i] = B[(size_t)(k0 + 8 * lh + i) * N + n0 + li];
            }
            bf16x8 ah, am, al, bh, bm, bl;
            split3(xa, ah, am, al);
            split3(xb, bh, bm, bl);
            if (MODE == 1) {
                c = mma6(ah, am, al, bh, bm, bl, c);
            } else if (MODE == 2) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
            } else {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        C[(size_t)(m0 + row) * N + n0 + li] = c[r];
    }
}

static double time_ms(void (*launch)(float *, int, int), float *out, int blocks, int iters) {
    hipEvent_t s, e;
    hipEventCreate(&s);
    hipEventCreate(&e);
    launch(out, blocks, 16);
    hipDeviceSynchronize();
    hipEventRecord(s);
    launch(out, blocks, iters);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    return ms;
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 4096 * 4);
    const int blocks = 256 * 2;                 // 2 workgroups of 4 waves per CU = 2 waves per SIMD
    const int iters = 2048;
    {
        const double ms = time_ms([](float *o, int b, int it) { rate_bf16<<<b, 256>>>(o, it); }, out, blocks, iters);
        const double fl = (double)blocks * 4 * iters * 32 * 2.0 * 32 * 32 * 16;
        printf("bf16 MFMA 32x32x16 from registers : %8.1f TFLOP/s\n", fl / (ms * 1e-3) / 1e12);
    }
    {
        const double ms = time_ms([](float *o, int b, int it) { rate_f32<<<b, 256>>>(o, it); }, out, blocks, iters);
        const double fl = (double)blocks * 4 * iters * 32 * 2.0 * 32 * 32 * 2;
        printf("fp32 MFMA 32x32x2 from registers  : %8.1f TFLOP/s\n", fl / (ms * 1e-3) / 1e12);
    }
    {
        const double ms = time_ms([](float *o, int b, int it) { rate_split6<1><<<b, 256>>>(o, it); }, out, blocks, iters * 4);
        const double fl = (double)blocks * 4 * iters * 4 * 1 * 2.0 * 32 * 32 * 16;
        printf("six-product split, split included, A reused x1 : %8.1f TFLOP/s fp32-equivalent\n", fl / (ms * 1e-3) / 1e12);
    }
    {
        const double ms = time_ms([](float *o, int b, int it) { rate_split6<4><<<b, 256>>>(o, it); }, out, blocks, iters);
        const double fl = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
        printf("six-product split, split included, A reused x4 : %8.1f TFLOP/s fp32-equivalent\n", fl / (ms * 1e-3) / 1e12);
    }
    // accuracy
    const int M = 128, N = 128, K = 4096;
    std::vector<float> hA((size_t)M * K), hB((size_t)K * N), hC((size_t)M * N);
    srand(1234);
    auto gauss = []() {
        const double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
        return (float)(sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v));
    };
    for (auto &x : hA) x = gauss();
    for (auto &x : hB) x = gauss();
    std::vector<double> ref((size_t)M * N), mag((size_t)M * N);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0.0, a = 0.0;
            for (int k = 0; k < K; ++k) {
                const double p = (double)hA[(size_t)m * K + k] * (double)hB[(size_t)k * N + n];
                s += p;
                a += fabs(p);
            }
            ref[(size_t)m * N + n] = s;
            mag[(size_t)m * N + n] = a;
        }
    float *dA, *dB, *dC;
    hipMalloc(&dA, hA.size() * 4);
    hipMalloc(&dB, hB.size() * 4);
    hipMalloc(&dC, hC.size() * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    const char *names[4] = {"fp32 MFMA (v_mfma_f32_32x32x2_f32)", "six-product bf16 split (3 terms)", "three-product bf16 split (2 terms)",
                            "plain bf16"};
    for (int mode = 0; mode < 4; ++mode) {
        dim3 grid(N / 32, M / 32);
        if (mode == 0) gemm_tile<0><<<grid, 64>>>(dA, dB, dC, M, N, K);
        if (mode == 1) gemm_tile<1><<<grid, 64>>>(dA, dB, dC, M, N, K);
        if (mode == 2) gemm_tile<2><<<grid, 64>>>(dA, dB, dC, M, N, K);
        if (mode == 3) gemm_tile<3><<<grid, 64>>>(dA, dB, dC, M, N, K);
        hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0.0, worst_abs = 0.0;
        for (size_t i = 0; i < hC.size(); ++i) {
            const double e = fabs((double)hC[i] - ref[i]);
            if (e / mag[i] > worst) worst = e / mag[i];
            if (e > worst_abs) worst_abs = e;
        }
        printf("%-38s: max |err| %.3e, max |err| / sum|a b| %.3e\n", names[mode], worst_abs, worst);
    }
    return 0;
}
