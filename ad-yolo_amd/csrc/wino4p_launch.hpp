// launcher interface of the persistent F(4x4,3x3) kernels (wino4p.hpp), shared with the dispatcher in wino4.hip
#pragma once
#include "common.hpp"

namespace adyolo {
namespace w4 {

struct W4Launch {                                         // everything adyolo_wino4_fwd hands to a launcher
    const float *x, *u, *bias, *addend, *addend_mask, *in_scale, *in_shift;
    float *y, *stats;
    const float *stat_aux, *stat_mean, *stat_invstd, *stat_mask;
    int H, W, Cin, Cout, patchesW, patchesH, nsp, ncb, xcd_div, relu, mask_bits, tc, grid, nb;
    hipStream_t st;
};
// EPI: bit 0 per-patch statistics, 1 addend, 2 addend mask (bits), 3 statistics against a BatchNorm input (stat_aux), 4 statistics
// mask (bits).  Compile-time, one translation unit per value (wino4p_e<EPI>.hip): the register allocation of a 512-register
// kernel does not survive run-time operand combinations (conditionally loaded operand arrays were merged through scratch
// memory).  Instantiated: the combinations the SE-ResNet block launches (functional.py) -- 0 plain, 1 forward convolutions,
// 9 data-gradient of conv2, 2 / 27 / 31 data-gradient of conv1 (projection shortcut after a pooled / un-pooled stage boundary,
// identity shortcut), 15 the same for the first block (statistics against the stem's BatchNorm input: no mask); every other combination, and
// masks given as float tensors, take the one-patch kernel (wino4.hip)
template <int EPI>
void launch_wino4p(const W4Launch &a);

}  // namespace w4
}  // namespace adyolo
