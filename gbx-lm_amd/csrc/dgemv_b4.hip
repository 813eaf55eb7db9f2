// 4-bit instantiations of the decode-block GEMV (one TU per width so the three compile in parallel).
#include "dgemv_impl.h"
namespace gbx {
GBX_DEFINE_LAUNCH_DGEMV(4)
}  // namespace gbx
