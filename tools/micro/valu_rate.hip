// VALU issue-rate probe for gfx950 (round 3: SLP-proof).  Round 2's version was written with fmaf() and the compiler packed
// the whole stream into v_pk_fma_f32, so its "cycles per wave64 instruction" described PACKED
// instructions.  Here the two streams are inline assembly: N waves per SIMD run a chain-free stream of
//   (a) v_fma_f32      (one fp32 FMA per lane and instruction)
//   (b) v_pk_fma_f32   (two)
// and the probe reports wave-instructions per ns per SIMD, cycles per instruction at the measured rate and TFLOP/s.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void fma_scalar(float *out, int iters, float a, float b) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_packed(float *out, int iters, float a, float b) {
    v2f x[8], va = {a, a}, vb = {b, b};
    for (int i = 0; i < 8; ++i) x[i] = v2f{(float)threadIdx.x + i, (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(va), "v"(vb));
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 8192 * 4);
    hipEvent_t s, e;
    hipEventCreate(&s);
    hipEventCreate(&e);
    const int iters = 4096;
    for (int packed = 0; packed < 2; ++packed)
        for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
            const int blocks = 256 * wg_per_cu;
            auto launch = [&](int it) {
                if (packed) fma_packed<<<blocks, 256>>>(out, it, 1.0001f, 0.5f);
                else fma_scalar<<<blocks, 256>>>(out, it, 1.0001f, 0.5f);
            };
            launch(64);
            hipDeviceSynchronize();
            hipEventRecord(s);
            launch(iters);
            hipEventRecord(e);
            hipEventSynchronize(e);
            float ms;
            hipEventElapsedTime(&ms, s, e);
            const double inst = (double)blocks * 4 * iters * 128;          // wave-instructions
            const double rate = inst / 1024 / (ms * 1e6);                  // per ns per SIMD
            printf("%-13s %d waves/SIMD: %.3f ms, %.3f wave-instr/ns/SIMD (%.2f cycles per instruction at 2.4 GHz), %.1f TFLOP/s\n",
                   packed ? "v_pk_fma_f32" : "v_fma_f32", wg_per_cu, ms, rate, 2.4 / rate,
                   inst * 64 * 2 * (packed ? 2 : 1) / (ms * 1e-3) / 1e12);
        }
    return 0;
}
