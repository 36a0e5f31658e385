// explicit instantiation of the conv kernels for 7x7, stride 1
#include "conv_mfma_kernel.h"
namespace lssvc {
template int dispatch_tile<7, 1, true>(const ConvP &, int, int, hipStream_t);
template int dispatch_tile<7, 1, false>(const ConvP &, int, int, hipStream_t);
}  // namespace lssvc
