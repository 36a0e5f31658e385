// explicit instantiation of the conv kernels for 3x3, stride 2
#include "conv_mfma_kernel.h"
namespace lssvc {
template int dispatch_tile<3, 2, true>(const ConvP &, int, int, hipStream_t);
template int dispatch_tile<3, 2, false>(const ConvP &, int, int, hipStream_t);
}  // namespace lssvc
