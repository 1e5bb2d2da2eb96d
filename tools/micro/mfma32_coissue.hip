// Can one wave hide other instructions under v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles per instruction and SIMD)?
// Per MFMA the loop inserts K independent v_fma_f32 (inline asm: no SLP pairing, no elimination), or a ds_read_b128, or a
// coalesced 1 KB global_load_dwordx4.  One wave per SIMD (256-thread workgroup per CU) or two.  Question behind it (wino4.hip):
// the F(4x4) kernel's non-MFMA work costs its full issue time ON TOP of the matrix time.
// build + run (GPU box): hipcc -O3 --offload-arch=gfx950 tools/micro/mfma32_coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int CHAIN, int MODE>
__global__ __launch_bounds__(256) void kern(float *out, const float4 *src, int iters) {
    __shared__ float4 lds[1024];
    float a = threadIdx.x * 0.001f, b = 1.0f;
    lds[threadIdx.x] = make_float4(a, b, a, b);
    lds[threadIdx.x + 256] = lds[threadIdx.x];
    __syncthreads();
    f32x16 c[4];
    for (int n = 0; n < 4; ++n) c[n] = (f32x16){0};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    float4 ld = make_float4(0.f, 0.f, 0.f, 0.f), acc4 = ld;
    const float4 *p = src + (blockIdx.x * 256 + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int k = 0; k < CHAIN; ++k) {
                c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[n], 0, 0, 0);
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < K; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j % 16]) : "v"(b));
                } else if (MODE == 1) {
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        ld = lds[(threadIdx.x + 64 * j + it) & 511];
                        acc4.x += ld.x;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        ld = p[(size_t)((it * 4 + n) & 63) * 65536];
                        acc4.x += ld.y;
                    }
                }
            }
    }
    float s = acc4.x;
    for (int n = 0; n < 4; ++n) s += c[n][n];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int K, int CHAIN, int MODE>
static void run(const char *name, int wgs_per_cu, float *d, float4 *src) {
    const int iters = 2000, cus = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<K, CHAIN, MODE><<<cus * wgs_per_cu, 256>>>(d, src, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<K, CHAIN, MODE><<<cus * wgs_per_cu, 256>>>(d, src, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)iters * 4 * CHAIN;                // per wave
    const double ns_per = ms * 1e6 / (n_mfma * wgs_per_cu);          // per MFMA and SIMD
    printf("%-34s K=%2d chain %d, %d wave(s)/SIMD: %7.2f ns per MFMA and SIMD (%.1f TFLOP/s)\n", name, K, CHAIN, wgs_per_cu, ns_per,
           4096.0 / ns_per * 1024 / 1e3);
}

int main() {
    float *d;
    float4 *src;
    hipMalloc(&d, 256 * 2 * 256 * sizeof(float));
    hipMalloc(&src, (size_t)64 * 65536 * 16 + 512 * 256 * 16);
    hipMemset(src, 0, (size_t)64 * 65536 * 16 + 512 * 256 * 16);
    for (int w = 1; w <= 2; ++w) {
        run<0, 4, 0>("v_fma_f32 per MFMA", w, d, src);
        run<2, 4, 0>("v_fma_f32 per MFMA", w, d, src);
        run<4, 4, 0>("v_fma_f32 per MFMA", w, d, src);
        run<8, 4, 0>("v_fma_f32 per MFMA", w, d, src);
        run<12, 4, 0>("v_fma_f32 per MFMA", w, d, src);
        run<16, 4, 0>("v_fma_f32 per MFMA", w, d, src);
        run<4, 1, 0>("v_fma_f32 per MFMA", w, d, src);
        run<8, 1, 0>("v_fma_f32 per MFMA", w, d, src);
        run<1, 4, 1>("ds_read_b128 per MFMA", w, d, src);
        run<2, 4, 1>("ds_read_b128 per MFMA", w, d, src);
        run<1, 4, 2>("global_load_dwordx4 per MFMA", w, d, src);
    }
    return 0;
}
