// dattn_kernel, 3-bit q|k|v rows (two lane passes up to K = 4096)
#include "dattn_kernel.h"
namespace gbx {
GBX_DEFINE_LAUNCH_DATTN(3, 2)
}
