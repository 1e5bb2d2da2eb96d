// ds_write_addtid_b32 with an AccVGPR data operand: where do the stored words come from?  (round 6 probe; see
// tools/micro/lds_write_rate.hip, whose addtid image held wrong words.)  One wave, three ways of filling a0..a3:
//   case 0: v_accvgpr_write from VGPRs holding known values; case 1: global_load_dword straight into the AccVGPRs;
//   case 2: an MFMA result (A = ones, B = lane-dependent) -- the epilogue's case.
// Each followed by four ds_write_addtid_b32 a[i] and, for comparison, four ds_write_b32 v_addr, a[i] into a second region.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/addtid_dbg2.hip -o tools/micro/addtid_dbg2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CASE>
__global__ __launch_bounds__(64) void k(float *out, const float *in) {
    __shared__ float lds[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) lds[i] = -1.f;
    __syncthreads();
    float a0, a1, a2, a3;
    if (CASE == 0) {
        a0 = (float)(lane * 4 + 0); a1 = (float)(lane * 4 + 1); a2 = (float)(lane * 4 + 2); a3 = (float)(lane * 4 + 3);
        asm volatile("" : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3));
    } else if (CASE == 1) {
        asm volatile("global_load_dword %0, %4, off\n\tglobal_load_dword %1, %4, off offset:256\n\t"
                     "global_load_dword %2, %4, off offset:512\n\tglobal_load_dword %3, %4, off offset:768\n\ts_waitcnt vmcnt(0)"
                     : "=a"(a0), "=a"(a1), "=a"(a2), "=a"(a3) : "v"(in + lane) : "memory");
    } else {
        f32x16 acc = {0};
        const float one = 1.0f, bv = (float)(lane & 31);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(one, bv, acc, 0, 0, 0);      // D[i][j] = sum_k 1 * B[k][j] = 2 * (j) for both k halves -> 2 j
        asm volatile("" : "+a"(acc));
        a0 = acc[0]; a1 = acc[1]; a2 = acc[2]; a3 = acc[3];
        asm volatile("" : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3));
    }
    const unsigned base = (unsigned)(size_t)lds;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %2 offset:0\n\tds_write_addtid_b32 %3 offset:256\n\t"
                 "ds_write_addtid_b32 %4 offset:512\n\tds_write_addtid_b32 %5 offset:768\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(base), "a"(a0), "a"(a1), "a"(a2), "a"(a3) : "memory");
    float *p = lds + 1024 + lane;
    asm volatile("ds_write_b32 %0, %1 offset:0\n\tds_write_b32 %0, %2 offset:256\n\tds_write_b32 %0, %3 offset:512\n\tds_write_b32 %0, %4 offset:768\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 :: "v"((unsigned)(size_t)p), "a"(a0), "a"(a1), "a"(a2), "a"(a3) : "memory");
    __syncthreads();
    for (int i = lane; i < 2048; i += 64) out[i] = lds[i];
}

int main() {
    float *out, *in;
    hipMalloc(&out, 2048 * 4);
    hipMalloc(&in, 4096 * 4);
    std::vector<float> h(4096), r(2048);
    for (int i = 0; i < 4096; ++i) h[i] = 1000.f + i;
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int c = 0; c < 3; ++c) {
        if (c == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, in);
        else if (c == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, in);
        else hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, in);
        hipDeviceSynchronize();
        hipMemcpy(r.data(), out, 2048 * 4, hipMemcpyDeviceToHost);
        int same = 0;
        for (int i = 0; i < 256; ++i) same += r[i] == r[1024 + i];
        printf("case %d: addtid image equals the ds_write_b32 image in %d of 256 words; lane 0..3 of register 0: addtid %.1f %.1f %.1f %.1f | ds_write_b32 %.1f %.1f %.1f %.1f;"
               " register 1 lane 0: %.1f | %.1f\n", c, same, r[0], r[1], r[2], r[3], r[1024], r[1025], r[1026], r[1027], r[64], r[1024 + 64]);
    }
    return 0;
}
