// explicit instantiation of the f16x3 conv kernels for 3x3, stride 1
#include "conv_f16x3_kernel.h"
namespace lssvc {
template int dispatch_tile_f16x3<3, 1>(const ConvP &, int, int, hipStream_t);
}  // namespace lssvc
