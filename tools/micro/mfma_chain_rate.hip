// Rate of v_mfma_f32_32x32x16_bf16 as a function of the number of INDEPENDENT accumulators a wave cycles through (1 = every
// MFMA depends on the previous one) and of the waves per SIMD (1 or 2).  Question behind it (wino_b3.hip): do the six
// back-to-back MFMAs on one accumulator of a bf16x3 product run at the full rate of the matrix pipe?
// build + run (GPU box): hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_chain_rate.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int CHAIN>
__global__ __launch_bounds__(256) void rate(float *out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(threadIdx.x * 0.001f + i);
        b[i] = (__bf16)(1.0f - i * 0.01f);
    }
    f32x16 c[NACC];
    for (int n = 0; n < NACC; ++n) c[n] = (f32x16){0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n)
#pragma unroll
            for (int k = 0; k < CHAIN; ++k) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) s += c[n][n];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int CHAIN>
static void run(const char *name, int wgs_per_cu, float *d) {
    const int iters = 2000, cus = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rate<NACC, CHAIN><<<cus * wgs_per_cu, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate<NACC, CHAIN><<<cus * wgs_per_cu, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)cus * wgs_per_cu * 4 * iters * NACC * CHAIN;
    const double tf = n_mfma * 32768.0 / (ms * 1e-3) / 1e12;
    const double ns_per = ms * 1e6 / ((double)iters * NACC * CHAIN * wgs_per_cu);       // per MFMA per SIMD
    printf("%-44s %d wave(s)/SIMD: %8.1f TFLOP/s   %.2f ns per MFMA and SIMD\n", name, wgs_per_cu, tf, ns_per);
}

int main() {
    float *d;
    hipMalloc(&d, 256 * 2 * 256 * sizeof(float));
    for (int w = 1; w <= 2; ++w) {
        run<1, 6>("1 accumulator (all dependent)", w, d);
        run<2, 6>("2 accumulators, 6-long chains alternating", w, d);
        run<4, 6>("4 accumulators, 6-long chains", w, d);
        run<8, 6>("8 accumulators, 6-long chains", w, d);
        run<8, 1>("8 accumulators, round robin", w, d);
    }
    return 0;
}
