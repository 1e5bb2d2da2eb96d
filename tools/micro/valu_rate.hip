// VALU issue-rate probe for gfx950: N waves per SIMD run a chain-free stream of v_fma_f32; reports wave-instructions per
// cycle per SIMD (2 cycles per wave64 instruction = 0.5 would be the 157 TFLOP/s vector peak).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void fma_kernel(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
            x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main() {
    float *out;
    hipMalloc(&out, 256 * 8192 * 4);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 4096;
    for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
        const int blocks = 256 * wg_per_cu;
        fma_kernel<<<blocks, 256>>>(out, 64, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(s);
        fma_kernel<<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms;
        hipEventElapsedTime(&ms, s, e);
        const double inst = (double)blocks * 4 * iters * 128;          // wave-instructions
        const double tf = inst * 64 * 2 / (ms * 1e-3) / 1e12;
        printf("%d waves/SIMD: %.3f ms, %.1f TFLOP/s, %.3f wave-instr/ns/SIMD (at 2.4 GHz: %.2f cycles per instr per SIMD)\n",
               wg_per_cu, ms, tf, inst / 1024 / (ms * 1e6), 2.4 / (inst / 1024 / (ms * 1e6)));
    }
    return 0;
}
