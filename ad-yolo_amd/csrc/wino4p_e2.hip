// one epilogue variant of the persistent F(4x4,3x3) kernel (wino4p.hpp): EPI = 2
#include "wino4p.hpp"
namespace adyolo {
namespace w4 {
template void launch_wino4p<2>(const W4Launch &);
}
}
