// 2-bit instantiations of the decode GEMV (one TU per width so the three compile in parallel).
#include "gemv_impl.h"
namespace gbx {
GBX_DEFINE_LAUNCH_GEMV(2)
}  // namespace gbx
