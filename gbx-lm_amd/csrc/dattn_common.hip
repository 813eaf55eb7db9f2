// Host side of the fused attention launch (dattn_impl.h): which shapes it covers, and the dispatch over the q|k|v bit width.
#include "dattn_impl.h"
#include "gbx_env.h"
#include "kernels.h"

namespace gbx {

bool dattn_plan(int D, int n_heads, int n_kv, int head_dim, int qkv_bits, int qkv_group, int o_bits, int o_group, DattnPlan* pl) {
  if (!xcd_layout_ok()) return false;                                     // 8 XCDs x 32 CUs, all available, round-robin placement (xcd_layout.hip)
  if (n_kv != DA_GROUPS || head_dim != DA_HD) return false;               // one kv head per XCD
  if (n_heads <= 0 || (n_heads % DA_GROUPS) != 0) return false;
  const int rep = n_heads / DA_GROUPS;
  if (rep > 8) return false;
  if (D <= 0 || (D % 128) != 0 || (D % DA_MEMBERS) != 0) return false;
  if (qkv_bits < 2 || qkv_bits > 4 || o_bits < 2 || o_bits > 4) return false;
  auto okg = [](int g) { return g == 32 || g == 64 || g == 128; };
  if (!okg(qkv_group) || !okg(o_group)) return false;
  const int vq = qkv_bits == 2 ? 64 : 32, vo = o_bits == 2 ? 64 : 32;
  if (qkv_group < vq || o_group < vo) return false;                       // one (scale, zero) per lane slot
  // phase 1: the rows are one (2 bits) or two (3 / 4 bits) lane passes long
  const int nslots = D / vq;
  const int P = qkv_bits == 2 ? 1 : 2;
  if (nslots > 64 * P) return false;
  const int nb = (rep + 2) * (DA_HD / 2), nprod = DA_MEMBERS - rep;
  const int bpw = (nb + nprod - 1) / nprod;
  if (bpw > DA_NW) return false;
  if (static_cast<long long>(n_heads + 2 * n_kv) * DA_HD * (D / 32 * qkv_bits) * 4 >= 0x7FFFFFFFll) return false;
  // phase 3: rows of a workgroup x slots of a row slice must fit DA_OP passes of 16 waves
  const int KC = rep * DA_HD;
  if ((KC % o_group) != 0) return false;
  const int spr = KC / vo;
  if (spr > 32) return false;
  const int lpr = spr <= 8 ? 8 : (spr <= 16 ? 16 : 32);
  const int rpp = 64 / lpr, RPW = D / DA_MEMBERS;
  if (RPW > DA_OP * DA_NW * rpp) return false;
  if (static_cast<long long>(D) * (KC * DA_GROUPS / 8) * o_bits >= 0x7FFFFFFFll) return false;
  *pl = DattnPlan{P, bpw, static_cast<size_t>(84 * 1024)};
  return true;
}

// the matrix-core form of the q|k|v rows and the o_proj slice (dattn_kernel<2, 1, true, 2>): bf16, D = 4096 (two spans), group 128 for
// both matrices, 2-bit o_proj, four query heads per kv head
bool dattn_mx_wanted(const DattnArgs& a, bool bf16) {
  static const int on = env_switch("GBX_DATTN_MX");
  return on && bf16 && a.D == 4096 && a.rep == 4 && a.qkv_log2g == 7 && a.o_bits == 2 && a.o_log2g == 7;
}

hipError_t launch_dattn(const DattnArgs& a, const DattnPlan& pl, int qkv_bits, bool bf16, hipStream_t stream) {
  switch (qkv_bits) {
    case 2: return launch_dattn_bits<2>(a, pl, bf16, stream);
    case 3: return launch_dattn_bits<3>(a, pl, bf16, stream);
    default: return launch_dattn_bits<4>(a, pl, bf16, stream);
  }
}

}  // namespace gbx
