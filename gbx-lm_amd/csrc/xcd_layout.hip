// What the launches with XCD-local hand-offs (dattn_kernel, dmlp_kernel, dblock_kernel, slices_norm_xcd_kernel) require of the DEVICE,
// checked once per device instead of assumed:
//   * 256 CUs in ONE partition (SPX: 8 XCDs x 32 CUs) -- a CPX / DPX / QPX partition shows up as a device with fewer CUs;
//   * no CU mask on the process (HSA_CU_MASK / ROC_GLOBAL_CU_MASK leave multiProcessorCount alone but take CUs away: the 256 one-per-CU
//     workgroups would no longer be resident together and every hand-off would spin to its limit);
//   * the dispatcher deals consecutive workgroups round-robin over the XCDs, so that blocks 8 r .. 8 r + 7 sit on 8 DIFFERENT XCDs and
//     (group = XCC_ID, member = blockIdx / 8) is a bijection -- HIP promises no such thing (MI355X_MICROARCH.md, "Contract"), so a census
//     launch of the same geometry (256 x 1024 threads, one workgroup per CU) MEASURES it on this device before the first fused launch.
// A device that fails any of these keeps the five-launch decode block (the plan functions return "unsupported"); a launch that loses a
// member at run time (another tenant on the GPU) gives up after its spin limit and poisons its output (dattn_impl.h).
#include "kernels.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

namespace gbx {

namespace {

constexpr int XC_WG = 256, XC_NT = 1024, XC_LDS = 84 * 1024, XC_LAUNCHES = 6;

__global__ __launch_bounds__(XC_NT) void xcd_census_kernel(unsigned* out, int launch) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 0) {
    smem[0] = 1;
    out[(launch * XC_WG + blockIdx.x) * 2] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu;      // HW_REG_XCC_ID
    out[(launch * XC_WG + blockIdx.x) * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);           // HW_REG_HW_ID (cu 11:8, sh 12, se 15:13)
  }
}

struct DevState {
  std::atomic<int> static_done{0}, census_done{0};
  char static_why[160];
  char census_why[160];
};
DevState g_dev[64];

DevState* cur(int* devp) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) return nullptr;
  if (devp) *devp = d;
  return &g_dev[d & 63];
}

void static_check(DevState* s, int dev) {
  s->static_why[0] = 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    snprintf(s->static_why, sizeof(s->static_why), "hipGetDeviceProperties failed");
  } else if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    snprintf(s->static_why, sizeof(s->static_why), "device is %s, not gfx950", prop.gcnArchName);
  } else if (prop.multiProcessorCount != XC_WG) {
    snprintf(s->static_why, sizeof(s->static_why), "device reports %d CUs, not 256 (a CPX / DPX / QPX partition?)", prop.multiProcessorCount);
  } else {
    for (const char* n : {"HSA_CU_MASK", "ROC_GLOBAL_CU_MASK", "HSA_CU_MASK_SKIP_INIT"}) {
      const char* v = getenv(n);
      if (v && *v) { snprintf(s->static_why, sizeof(s->static_why), "%s is set: not every CU is available to this process", n); break; }
    }
  }
  s->static_done.store(1, std::memory_order_release);
}

}  // namespace

// "" when the device passes the static checks (and the census, once one has run); otherwise the reason.  Never launches anything.
const char* xcd_layout_reason() {
  int dev = 0;
  DevState* s = cur(&dev);
  if (!s) return "no HIP device";
  if (!s->static_done.load(std::memory_order_acquire)) static_check(s, dev);
  if (s->static_why[0]) return s->static_why;
  if (s->census_done.load(std::memory_order_acquire) && s->census_why[0]) return s->census_why;
  return "";
}

bool xcd_layout_ok() { return xcd_layout_reason()[0] == 0; }

// The census: XC_LAUNCHES launches of the fused kernels' geometry on the null stream, synchronous (call it OUTSIDE stream capture: the plan
// entry points do, the forward entry points never do).  Runs once per device; later calls return the cached verdict.
hipError_t xcd_census() {
  int dev = 0;
  DevState* s = cur(&dev);
  if (!s) return hipErrorNoDevice;
  if (!s->static_done.load(std::memory_order_acquire)) static_check(s, dev);
  if (s->static_why[0] || s->census_done.load(std::memory_order_acquire)) return hipSuccess;
  s->census_why[0] = 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(xcd_census_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, XC_LDS);
  if (e != hipSuccess) return e;
  int per_cu = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, xcd_census_kernel, XC_NT, XC_LDS);
  if (e != hipSuccess) return e;
  if (per_cu < 1) {
    snprintf(s->census_why, sizeof(s->census_why), "a 1024-thread workgroup with %d KB of LDS does not fit a CU (occupancy query: %d)", XC_LDS / 1024, per_cu);
    s->census_done.store(1, std::memory_order_release);
    return hipSuccess;
  }
  unsigned* buf = nullptr;
  e = hipMalloc(&buf, XC_LAUNCHES * XC_WG * 2 * sizeof(unsigned));
  if (e != hipSuccess) return e;
  for (int l = 0; l < XC_LAUNCHES; ++l) hipLaunchKernelGGL(xcd_census_kernel, dim3(XC_WG), dim3(XC_NT), XC_LDS, nullptr, buf, l);
  std::vector<unsigned> h(XC_LAUNCHES * XC_WG * 2);
  e = hipMemcpy(h.data(), buf, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  (void)hipFree(buf);
  if (e != hipSuccess) return e;
  for (int l = 0; l < XC_LAUNCHES && !s->census_why[0]; ++l) {
    std::vector<unsigned> place;
    for (int r = 0; r < XC_WG / 8 && !s->census_why[0]; ++r) {
      unsigned seen = 0;
      for (int j = 0; j < 8; ++j) {
        const unsigned x = h[(l * XC_WG + 8 * r + j) * 2];
        if (x > 7 || (seen & (1u << x))) {
          snprintf(s->census_why, sizeof(s->census_why), "census launch %d: blocks %d..%d do not sit on 8 different XCDs (XCC_ID %u twice or out of range)",
                   l, 8 * r, 8 * r + 7, x);
          break;
        }
        seen |= 1u << x;
      }
    }
    for (int b = 0; b < XC_WG; ++b) place.push_back((h[(l * XC_WG + b) * 2] << 16) | ((h[(l * XC_WG + b) * 2 + 1] >> 8) & 0xFFu));
    std::sort(place.begin(), place.end());
    if (!s->census_why[0] && std::unique(place.begin(), place.end()) != place.end())
      snprintf(s->census_why, sizeof(s->census_why), "census launch %d: two of the 256 workgroups shared a CU (not one workgroup per CU)", l);
  }
  s->census_done.store(1, std::memory_order_release);
  return hipSuccess;
}

}  // namespace gbx
