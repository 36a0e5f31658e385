// explicit instantiation of the conv kernels for 1x1, stride 2
#include "conv_mfma_kernel.h"
namespace lssvc {
template int dispatch_tile<1, 2, true>(const ConvP &, int, int, hipStream_t);
template int dispatch_tile<1, 2, false>(const ConvP &, int, int, hipStream_t);
}  // namespace lssvc
