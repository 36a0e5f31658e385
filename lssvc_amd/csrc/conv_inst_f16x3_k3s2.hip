// explicit instantiation of the f16x3 conv kernels for 3x3, stride 2
#include "conv_f16x3_kernel.h"
namespace lssvc {
template int dispatch_tile_f16x3_s2<3>(const ConvP &, int, int, hipStream_t);
}  // namespace lssvc
