// one epilogue variant of the persistent F(4x4,3x3) kernel (wino4p.hpp): EPI = 15 (round 6: the data-gradient of the FIRST block's
// conv1 -- addend + mask bits, statistics against the stem's BatchNorm input, no statistics mask -- the last block launch of a
// bench step that still ran on the F(2x2) kernel)
#include "wino4p.hpp"
namespace adyolo {
namespace w4 {
template void launch_wino4p<15>(const W4Launch &);
}
}
