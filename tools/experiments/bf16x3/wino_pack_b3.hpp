// filter packing of the bf16x3 experiment (was in csrc/wino_common.hpp and called from wino.hip's pack_many kernel until round 5)
#pragma once
namespace adyolo {
// U = G g G^T, split into three bf16 terms, in fragment order [16 pos][Nn/32][K/16][3][64 lanes][8]: lane (n, h) k slot i =
// U_pos[cin 16 G + 8 (i >> 2) + 4 h + (i & 3)][cout 32 nb + n].  mode 0: forward filter, mode 1: data-gradient filter (see
// wino_pack_one in wino.hip; same arithmetic for U)
__device__ __forceinline__ void wino_pack_b3_one(const float *__restrict__ w, unsigned short *__restrict__ u, int Cin_real,
                                                 int K, int Nn, int mode, long idx) {
    const int i = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    const long rest = idx >> 9;
    const int nG = K / 16;
    const int G = (int)(rest % nG), nb = (int)(rest / nG);
    const int k = G * 16 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3), nn = nb * 32 + (lane & 31);
    float f[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float v = 0.f;
            if (mode == 0) {
                if (k < Cin_real) v = w[((size_t)nn * Cin_real + k) * 9 + a * 3 + b];
            } else {
                if (nn < Cin_real) v = w[((size_t)k * Cin_real + nn) * 9 + (2 - a) * 3 + (2 - b)];
            }
            f[a][b] = v;
        }
    float tt[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        tt[0][b] = f[0][b];
        tt[1][b] = 0.5f * (f[0][b] + f[1][b] + f[2][b]);
        tt[2][b] = 0.5f * (f[0][b] - f[1][b] + f[2][b]);
        tt[3][b] = f[2][b];
    }
    const size_t pos_stride = (size_t)(Nn / 32) * nG * 3 * 512;           // bf16 elements per transform position
    const size_t base = (((size_t)nb * nG + G) * 3) * 512 + (size_t)lane * 8 + i;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float uu[4];
        uu[0] = tt[a][0];
        uu[1] = 0.5f * (tt[a][0] + tt[a][1] + tt[a][2]);
        uu[2] = 0.5f * (tt[a][0] - tt[a][1] + tt[a][2]);
        uu[3] = tt[a][2];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const __bf16 h = (__bf16)uu[b];
            const float r1 = uu[b] - (float)h;
            const __bf16 m = (__bf16)r1;
            const float r2 = r1 - (float)m;
            const __bf16 l = (__bf16)r2;
            const size_t o = (size_t)(a * 4 + b) * pos_stride + base;
            u[o] = __builtin_bit_cast(unsigned short, h);
            u[o + 512] = __builtin_bit_cast(unsigned short, m);
            u[o + 1024] = __builtin_bit_cast(unsigned short, l);
        }
    }
}


}  // namespace adyolo
