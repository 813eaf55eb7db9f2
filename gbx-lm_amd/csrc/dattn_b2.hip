// dattn_kernel, 2-bit q|k|v rows (one lane pass up to K = 4096)
#include "dattn_kernel.h"
namespace gbx {
GBX_DEFINE_LAUNCH_DATTN_MX(2, 1)
GBX_DEFINE_LAUNCH_DATTN(2, 1)
}
