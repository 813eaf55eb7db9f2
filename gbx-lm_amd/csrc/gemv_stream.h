// Device-format program entry of the streamed decode GEMV / persistent decode engine (gemv_stream.hip).
// Built on the host by gbx_stream_encode (capi.hip) from the public descriptors of include/gbx_qlinear.h.
#pragma once
#include <stdint.h>

namespace gbx {

constexpr int ST_NW = 8;                 // wavefronts per workgroup; one workgroup per CU
constexpr int ST_NT = ST_NW * 64;
constexpr int ST_RS = 5;                 // ring slots per wave
constexpr int ST_PRE = 2;                // slots the first op of a launch requests before its activations have arrived
constexpr int ST_SLOT = 2560;            // bytes per slot: two row wave-loads (1 KiB each) + 512 B of (scale|zero) pairs
constexpr int ST_BPO_MAX = 16;           // batches per wave and op

// LDS carve of the workgroup (bytes)
constexpr int ST_LDS_XIMG = 40960;       // activation image of one op (K <= 16384 values at 2 bits, 14336 at 3/4 bits)
constexpr int ST_LDS_RED = ST_LDS_XIMG;
constexpr int ST_LDS_PART = ST_LDS_RED + 64;
constexpr int ST_LDS_RESL = ST_LDS_PART + (ST_NW / 4) * ST_BPO_MAX * 8 * 4;
constexpr int ST_LDS_RING = ST_LDS_RESL + ST_BPO_MAX * ST_NW * 2 * 4;
constexpr int ST_LDS_BYTES = ST_LDS_RING + ST_NW * ST_RS * ST_SLOT;

enum : int { ST_GEMV = 0, ST_ATTN = 1 };
enum : int { ST_F_SWIGLU = 1, ST_F_NORM = 2, ST_F_RESID = 4 };

struct StOp {
  int type, bits, ks, P;
  int K, gshift, nseg, flags;
  int n0, n01, N, nbatch;            // N: rows over the segments (SwiGLU: rows of one segment = batches)
  int W4, G8, nslots, pad0;          // packed row bytes, bytes of one (scale|zero) row pair, slots per row
  const uint32_t* qw[3];
  const uint32_t* sz;
  const unsigned short* x;
  const unsigned short* norm_w;
  const unsigned short* resid;
  unsigned short* y[3];
  float eps;
  int pad1[3];
};
static_assert(sizeof(StOp) == 160, "StOp layout is part of the host <-> device contract");

}  // namespace gbx
