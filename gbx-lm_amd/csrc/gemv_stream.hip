// Streamed decode GEMV for GBA packed weights on gfx950: the same arithmetic as gemv_impl.h (lane <-> slot mapping, in-place
// field masks + v_dot2, per-sub-slot scale/zero, two-row DPP reduction: results are bit-identical), but the packed rows
// reach the CU through a per-wave LDS ring filled by LDS-DMA (`buffer_load_dwordx4 ... lds`) that runs AHEAD of the
// dependency: a wave keeps ST_RS slots (2 row wave-loads + their scale|zero pairs each: four DMA instructions) in flight without holding a
// single VGPR for them, across batches, across the activation prologue and -- in a multi-op program -- across ops:
// weights never depend on activations, so the rows of op i+1 stream while op i's edge resolves.
//
// One workgroup (8 waves) per CU walks a PROGRAM of ops (device-resident array of StOp); a launch runs ops
// [op_begin, op_end).  One op per launch = the deeper-prefetch form of gbx_qlinear_fused_forward (the decode block's
// four GEMV launches, gbx_lm/models/qllama.py:76,96,115); a whole layer (or token) per launch = the persistent decode
// engine, whose ops hand activations to each other through tagged 8-byte granules (MI355X_MICROARCH.md, allgather).
//
// Replaces the M == 1 case of mx.quantized_matmul at gbx_lm/models/quantized_linear_gba.py:195-203.
#include "gemv_impl.h"
#include "gemv_stream.h"
#include "kernels.h"

namespace gbx {

namespace {

__device__ __forceinline__ constexpr int vmcnt_imm(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0xF << 8); }

#define GBX_LDSP(p) ((__attribute__((address_space(3))) void*)(p))

// The program is read-only for the whole launch: reading it through the CONSTANT address space makes hipcc fetch the
// fields with scalar loads (lgkmcnt, not vmcnt: the LDS-DMA queue of the ring is never drained for them) and keeps
// everything derived from them in SGPRs (descriptors built from VGPR-held pointers become waterfall loops).
using CStOp = const __attribute__((address_space(4))) StOp;
// The launch's FIRST op also travels by value in the kernel arguments (the kernarg segment is constant memory too): its
// fields are in SGPRs ~1.3 us earlier than a cold fetch of the device-resident program would deliver them.
__device__ __forceinline__ CStOp* cprog(const StOp* p, CStOp* first, int op_begin, int i) {
  const int k = __builtin_amdgcn_readfirstlane(i);
  return k == op_begin ? first : reinterpret_cast<CStOp*>(reinterpret_cast<uintptr_t>(p)) + k;
}

// Wave-uniform issue cursor over this wave's slot stream: (op, batch of the wave, pass).  The fields of the op it
// points at are cached in SGPRs; cur_load re-reads them when the cursor enters another op.
struct IssueCur {
  int op, j, p;          // next slot to request
  int jn;                // batches this wave owns in `op`
  // cached op fields
  int ks, P, slotb, nslots, sshift, rs, n0, n01, W4, G8, owners, owner;
  // segment bases as base + deltas: a select among three pointer MEMBERS is turned into an indexed load of the struct by
  // hipcc, which pins the whole cursor in scratch memory (and every scratch access is a vmcnt event in the ring's queue)
  const unsigned char* q0; long long d1, d2; const unsigned char* sz;
};

__device__ __forceinline__ int batches_of(int nbatch, int owner, int owners) {
  return owner < nbatch ? (nbatch - 1 - owner) / owners + 1 : 0;
}
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T> __device__ __forceinline__ const unsigned char* sgpr_ptr(T p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
  return reinterpret_cast<const unsigned char*>((static_cast<unsigned long long>(hi) << 32) | lo);
}

__device__ __forceinline__ void cur_load(IssueCur& c, const StOp* prog, CStOp* first, int op_begin, int op_end, int cu, int wave, int ncu) {
  // skip ops without weights of this wave
  while (c.op < op_end) {
    CStOp* o = cprog(prog, first, op_begin, c.op);
    if (o->type == ST_GEMV) {
      c.ks = o->ks;
      c.owners = ncu * (c.ks == 1 ? ST_NW : ST_NW / 4);
      c.owner = c.ks == 1 ? cu * ST_NW + wave : cu * (ST_NW / 4) + (wave >> 2);
      c.jn = batches_of(o->nbatch, c.owner, c.owners);
      if (c.jn > 0) {
        const int bits = o->bits;
        c.P = o->P; c.rs = (o->flags & ST_F_SWIGLU) ? 0 : 1; c.n0 = o->n0; c.n01 = o->n01;
        c.W4 = o->W4; c.G8 = o->G8; c.nslots = o->nslots;
        c.slotb = bits == 3 ? 12 : 16;
        c.sshift = o->gshift - (bits == 2 ? 1 : 0);        // slot -> group: 64-value slots at 2 bits, 32-value slots otherwise
        c.q0 = sgpr_ptr(o->qw[0]); c.sz = sgpr_ptr(o->sz);
        c.d1 = sgpr_ptr(o->qw[1]) - c.q0; c.d2 = sgpr_ptr(o->qw[2]) - c.q0;
        return;
      }
    }
    c.op = sgpr(c.op + 1);
  }
}

constexpr int RS_FLAGS = 0x00020000, RS_ALL = 0x7FFFFFFF;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* p, int bytes = RS_ALL) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, RS_FLAGS);
}

// Request the cursor's slot into ring position `dst` (LDS byte address, wave-uniform) and advance the cursor: three DMA
// instructions -- the two rows' wave-loads and ONE 16-byte-per-lane load of the slot's (scale|zero) pairs (lane l fetches
// the pairs of groups g0 + 2l, g0 + 2l + 1 of the row pair; LDS-DMA moves 4, 12 or 16 bytes per lane, never 8).
// Every batch of a wave is a whole one (row counts are even, SwiGLU segments equal), so no row index needs clamping.
// Returns false (nothing requested) once the program is exhausted.
__device__ __forceinline__ bool issue_slot(IssueCur& c, const StOp* prog, CStOp* first, int op_begin, int op_end,
                                           unsigned char* dst, int lane, int wave, int cu, int ncu) {
  if (c.op >= op_end) return false;
  const int b = sgpr(c.owner + c.j * c.owners);
  const int r0 = sgpr(b << c.rs);
  const int slot_first = sgpr(c.p * 64 * c.ks + (c.ks == 1 ? 0 : (wave & 3) * 64));
  const int slot = min(slot_first + lane, c.nslots - 1);
  const int voff_w = slot * c.slotb;
  // rows: SwiGLU = row r0 of segment 0 and of segment 1; otherwise rows r0, r0 + 1 of ONE segment (every N is even)
  const unsigned char* qa; const unsigned char* qb;
  if (c.rs == 0) {
    const long long off = static_cast<long long>(r0) * c.W4;
    qa = c.q0 + off; qb = qa + c.d1;
  } else {
    const int sg = (r0 >= c.n0) + (r0 >= c.n01);
    const int row = r0 - (sg == 0 ? 0 : (sg == 1 ? c.n0 : c.n01));
    qa = c.q0 + ((sg == 0 ? 0ll : (sg == 1 ? c.d1 : c.d2)) + static_cast<long long>(row) * c.W4);
    qb = qa + c.W4;
  }
  const __amdgpu_buffer_rsrc_t da = mk_rsrc(sgpr_ptr(qa)), db = mk_rsrc(sgpr_ptr(qb));
  if (c.slotb == 12) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(da, GBX_LDSP(dst), 12, voff_w, 0, 0, 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(db, GBX_LDSP(dst + 1024), 12, voff_w, 0, 0, 2);
  } else {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(da, GBX_LDSP(dst), 16, voff_w, 0, 0, 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(db, GBX_LDSP(dst + 1024), 16, voff_w, 0, 0, 2);
  }
  {
    const int g0 = sgpr(slot_first >> c.sshift);
    const int gl = min(g0 + 2 * lane, c.G8 / 8 - 2);                    // lanes past the row pair re-read its last 16 bytes
    const unsigned char* ps = c.sz + static_cast<long long>(b) * c.G8;
    // 32 lanes only (EXEC-masked: the slot has 512 bytes for the pairs; a full wave would write 1 KiB into the NEXT slot)
    if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(mk_rsrc(sgpr_ptr(ps)), GBX_LDSP(dst + 2048), 16, gl * 8, 0, 0, 0);
  }
  c.p = sgpr(c.p + 1);
  if (c.p == c.P) {
    c.p = 0;
    c.j = sgpr(c.j + 1);
    if (c.j == c.jn) {
      c.j = 0;
      c.op = sgpr(c.op + 1);
      cur_load(c, prog, first, op_begin, op_end, cu, wave, ncu);
    }
  }
  return true;
}

// Wait until the OLDEST slot in flight has landed: all but the 3 * `ahead` youngest vector-memory operations done
// (`ahead` = ring slots requested after it; anything else issued since only makes the wait more conservative).
__device__ __forceinline__ void wait_oldest(int ahead) {
  switch (ahead) {
    case 0: __builtin_amdgcn_s_waitcnt(vmcnt_imm(0)); break;
    case 1: __builtin_amdgcn_s_waitcnt(vmcnt_imm(3)); break;
    case 2: __builtin_amdgcn_s_waitcnt(vmcnt_imm(6)); break;
    case 3: __builtin_amdgcn_s_waitcnt(vmcnt_imm(9)); break;
    case 4: __builtin_amdgcn_s_waitcnt(vmcnt_imm(12)); break;
    default: __builtin_amdgcn_s_waitcnt(vmcnt_imm(15)); break;
  }
  asm volatile("" ::: "memory");
}

#ifdef GBX_STAMPS
#define ST_STAMP(w, opl, i) do { if ((w).lane == 0 && (opl) < 5) (w).stamps[(static_cast<size_t>((w).cu) * ST_NW + (w).wave) * 64 + (opl) * 12 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ST_STAMP(w, opl, i) do {} while (0)
#endif

struct WaveCtx {
#ifdef GBX_STAMPS
  unsigned long long* stamps;
  int opl;
#endif
  unsigned char* smem;
  unsigned char* ring;     // this wave's ring
  int ri;                  // ring position of the oldest slot in flight (the one consumed next)
  int wi;                  // ring position the next request goes to
  int inflight;            // slots requested and not yet consumed
  const StOp* prog; CStOp* first; int op_begin, op_end;
  int tid, lane, wave, cu, ncu;
};

// One op of the program, all eight waves of the workgroup.
template <int BITS, int P, int KS, bool BF16>
__device__ __forceinline__ void run_gemv_op(int opi, IssueCur& cur, WaveCtx& w) {
  using F = Fmt<BITS>;
  using L = Lds<BITS>;
  using WT = typename SlotWords<BITS>::type;
  constexpr int NT = ST_NT, NW = ST_NW;
  constexpr int PAIRS = L::PAIRS, SSTR = L::SSTR, NSUB = F::NSUB;
  constexpr int UPS = F::VPL / F::UNIT_VALS;
  constexpr int UPT = (P * KS * 64 * UPS + NT - 1) / NT;
  constexpr int UW = F::UNIT_VALS / 2;
  constexpr int UPSUB = (32 / F::UNIT_VALS) ? (32 / F::UNIT_VALS) : 1;
  constexpr int PSTRIDE = 64 * KS;

  CStOp* o = cprog(w.prog, w.first, w.op_begin, opi);
  const int K = o->K, N = o->N;
  const int nslots = o->nslots;
  const int tid = w.tid, lane = w.lane, wave = w.wave;
  const bool swiglu = (o->flags & ST_F_SWIGLU) != 0;
  const bool norm = (KS == 1) && (o->flags & ST_F_NORM) != 0;
  const int rowstep = swiglu ? 1 : 2;
  const int owners = w.ncu * (KS == 1 ? NW : NW / 4);
  const int owner = KS == 1 ? w.cu * NW + wave : w.cu * (NW / 4) + (wave >> 2);
  const int jn = batches_of(o->nbatch, owner, owners);
  const int kq = wave & 3, rg = wave >> 2;
  const int slot0 = (KS == 1 ? 0 : kq * 64) + lane;
  const int sshift = o->gshift - (BITS == 2 ? 1 : 0);      // slot -> group

  h2* xp_lds = reinterpret_cast<h2*>(w.smem);
  float2* xs_lds = reinterpret_cast<float2*>(w.smem + static_cast<size_t>(nslots) * SSTR * 4);
  float* red_lds = reinterpret_cast<float*>(w.smem + ST_LDS_RED);
  float* part_lds = reinterpret_cast<float*>(w.smem + ST_LDS_PART);
  float* resl_lds = reinterpret_cast<float*>(w.smem + ST_LDS_RESL);

#ifdef GBX_STAMPS
  const int opl = w.opl++;
#endif
  ST_STAMP(w, opl, 0);
  // ---------------- 1. activation (and norm weight) loads: issued BEFORE the first ring fill (vmcnt retires in order)
  const unsigned short* __restrict__ xg = o->x;
  const unsigned short* __restrict__ wg = norm ? o->norm_w : xg;
  const int nunits = K / F::UNIT_VALS;
  constexpr int WPT = (KS == 1) ? UPT : 1;
  unsigned xr[UPT][UW], wr[WPT][UW];
#pragma unroll
  for (int j = 0; j < UPT; ++j) {
    const int u = min(tid + NT * j, nunits - 1);
    const unsigned* src = reinterpret_cast<const unsigned*>(xg + static_cast<size_t>(u) * F::UNIT_VALS);
    const unsigned* wsrc = reinterpret_cast<const unsigned*>(wg + static_cast<size_t>(u) * F::UNIT_VALS);
#pragma unroll
    for (int q = 0; q < UW; q += 4) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(src + q);
      xr[j][q] = v.x; xr[j][q + 1] = v.y; xr[j][q + 2] = v.z; xr[j][q + 3] = v.w;
      if constexpr (KS == 1) {
        const u32x4 t = *reinterpret_cast<const u32x4*>(wsrc + q);
        wr[j][q] = t.x; wr[j][q + 1] = t.y; wr[j][q + 2] = t.z; wr[j][q + 3] = t.w;
      }
    }
  }
  // ---------------- 2. keep the ring full.  A later op of a launch finds it full already (its rows were requested while the
  // previous op was being consumed).  The first op requests ST_PRE slots now and the rest once x has arrived: a wave
  // that queues its whole ring here is held at the issue port (back-pressure) while its activations wait.
  auto top_up = [&](int limit) {
#pragma unroll 1
    while (w.inflight < limit) {
      if (!issue_slot(cur, w.prog, w.first, w.op_begin, w.op_end, w.ring + w.wi * ST_SLOT, lane, wave, w.cu, w.ncu)) break;
      w.wi = (w.wi + 1 == ST_RS) ? 0 : w.wi + 1;
      w.inflight = sgpr(w.inflight + 1);
    }
  };
  top_up(ST_PRE);
  ST_STAMP(w, opl, 1);
  // ---------------- 3a. optional fused RMSNorm
  float rstd = 1.f;
  if (norm) {
    float ssq = 0.f;
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
      const bool live = (tid + NT * j) < nunits;
#pragma unroll
      for (int q = 0; q < UW; ++q) {
        const float lo = act_lo<BF16>(xr[j][q]), hi = act_hi<BF16>(xr[j][q]);
        ssq += live ? fmaf(hi, hi, lo * lo) : 0.f;   // explicit: -ffp-contract=on may pick either product for the fma, per instantiation
      }
    }
    ssq = wave_sum(ssq);
    __builtin_amdgcn_sched_barrier(0);
    top_up(ST_RS);                                   // x is here: the rest of the ring goes out before the barrier
    if (lane == 0) red_lds[wave] = ssq;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) tot += red_lds[i];
    rstd = rsqrtf(tot / static_cast<float>(K) + o->eps);
  }
  ST_STAMP(w, opl, 2);
  if (!norm) {
    // touch x (forces its wait here), then queue the rest of the ring
    unsigned t = 0;
#pragma unroll
    for (int j = 0; j < UPT; ++j) t |= xr[j][0];
    asm volatile("" :: "v"(t));
    __builtin_amdgcn_sched_barrier(0);
    top_up(ST_RS);
  }
  // ---------------- 3b. activations -> pre-shifted fp16 pairs, one block exponent per 32-value sub-slot
#pragma unroll
  for (int j = 0; j < UPT; ++j) {
    const int u = tid + NT * j;
    float xf[F::UNIT_VALS];
    unsigned amax = 0;
#pragma unroll
    for (int q = 0; q < UW; ++q) {
      float lo = act_lo<BF16>(xr[j][q]), hi = act_hi<BF16>(xr[j][q]);
      if constexpr (KS == 1) {
        if (norm) {
          lo = round_bf16_or_f16(lo * rstd * act_lo<BF16>(wr[j][q]), BF16);
          hi = round_bf16_or_f16(hi * rstd * act_hi<BF16>(wr[j][q]), BF16);
        }
      }
      xf[2 * q] = lo;
      xf[2 * q + 1] = hi;
      amax = max(amax, max(__float_as_uint(lo) & 0x7FFFFFFFu, __float_as_uint(hi) & 0x7FFFFFFFu));
    }
    float usum = 0.f;
#pragma unroll
    for (int q = 0; q < F::UNIT_VALS; ++q) usum += xf[q];
    float ssum;
    if constexpr (BITS == 2) { ssum = pair_sum(usum); amax = max(amax, dpp_u32<0xB1>(amax)); }
    else if constexpr (BITS == 4) {
      ssum = quad_sum(usum);
      amax = max(amax, dpp_u32<0xB1>(amax));
      amax = max(amax, dpp_u32<0x4E>(amax));
    } else ssum = usum;
    int E = 141 - static_cast<int>(amax >> 23);
    E = min(max(E, -100), 110);
    if (u < nunits) {
      const int slot = u / UPS, uin = u % UPS;
      h2* dst = xp_lds + static_cast<size_t>(slot) * SSTR + uin * F::UNIT_PAIRS;
#pragma unroll
      for (int i = 0; i < F::UNIT_PAIRS; ++i) {
        const float sc = exp2i(E - F::pos(i));
        h2 v;
        v.x = static_cast<_Float16>(xf[F::ka(i)] * sc);
        v.y = static_cast<_Float16>(xf[F::kb(i)] * sc);
        dst[i] = v;
      }
      if ((u % UPSUB) == 0) xs_lds[u / UPSUB] = make_float2(ssum, exp2i(24 - E));
    }
  }
  __syncthreads();
  ST_STAMP(w, opl, 3);

  // ---------------- 4. lane-resident activations
  h2 xv[P][PAIRS];
  float xs[P][NSUB], cf[P][NSUB];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int slot = p * PSTRIDE + slot0;
    const bool live = slot < nslots;
    const int sl = min(slot, nslots - 1);
    const u32x4* src = reinterpret_cast<const u32x4*>(xp_lds + static_cast<size_t>(sl) * SSTR);
#pragma unroll
    for (int q = 0; q < PAIRS / 4; ++q) {
      const u32x4 v = src[q];
      xv[p][4 * q + 0] = as_h2(live ? v.x : 0u);
      xv[p][4 * q + 1] = as_h2(live ? v.y : 0u);
      xv[p][4 * q + 2] = as_h2(live ? v.z : 0u);
      xv[p][4 * q + 3] = as_h2(live ? v.w : 0u);
    }
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      const float2 t = xs_lds[sl * NSUB + s];
      xs[p][s] = live ? t.x : 0.f;
      cf[p][s] = live ? t.y : 0.f;
    }
  }

  ST_STAMP(w, opl, 4);
  // ---------------- 5. stream: consume the oldest ring slot, refill it with the cursor's next slot (any later op)
#pragma unroll 1
  for (int j = 0; j < jn; ++j) {
    float acc[2] = {0.f, 0.f};
#pragma unroll
    for (int p = 0; p < P; ++p) {
      unsigned char* sl = w.ring + w.ri * ST_SLOT;
      wait_oldest(w.inflight - 1);
      if (j == 0 && p == 0) ST_STAMP(w, opl, 5);
      WT wv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr (BITS == 3) {
          // `buffer_load_dwordx3 ... lds` lands lane l's 12 bytes at l * 16 (measured, scratch/probe3/probe_dma3.hip)
          const u32x4 t3 = *reinterpret_cast<const u32x4*>(sl + i * 1024 + lane * 16);
          wv[i].x = t3.x; wv[i].y = t3.y; wv[i].z = t3.z;
        } else {
          wv[i] = *reinterpret_cast<const u32x4*>(sl + i * 1024 + lane * 16);
        }
      }
      // group of this lane's slot relative to the first group of the pass (lanes past the row: group 0, nulled by x = 0)
      const int gi = max(((min(p * PSTRIDE + slot0, nslots - 1)) >> sshift) - ((p * PSTRIDE + (KS == 1 ? 0 : kq * 64)) >> sshift), 0);
      const u32x2 szv = *reinterpret_cast<const u32x2*>(sl + 2048 + gi * 8);
      const unsigned szp[2] = {szv.x, szv.y};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float d[NSUB];
        dot_slot<BITS>(wv[i], xv[p], d);
        const float sc0 = sz_scale<BF16>(szp[i]), ze0 = sz_zero<BF16>(szp[i]);
        if constexpr (NSUB == 2) {
          const float t = fmaf(cf[p][1], d[1], cf[p][0] * d[0]);
          acc[i] = fmaf(sc0, t, acc[i]);
          acc[i] = fmaf(ze0, xs[p][0] + xs[p][1], acc[i]);
        } else {
          const float sc = sc0 * cf[p][0];
          acc[i] = fmaf(sc, d[0], acc[i]);
          acc[i] = fmaf(ze0, xs[p][0], acc[i]);
        }
      }
      // the slot's words are in registers (the dots above consumed them): hand the ring position back to the stream
      asm volatile("" ::: "memory");
      if (j == 0 && p == 0) ST_STAMP(w, opl, 6);
      w.ri = (w.ri + 1 == ST_RS) ? 0 : w.ri + 1;
      w.inflight = sgpr(w.inflight - 1);
      top_up(ST_RS);
      if (j == 0 && p == 0) ST_STAMP(w, opl, 7);
    }
    const float tot = wave_sum2(acc[0], acc[1]);
    const float t0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tot), 31));
    const float t1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tot), 63));
    if (lane == 0) {
      if constexpr (KS == 1) {
        resl_lds[(j * NW + wave) * 2 + 0] = t0;
        resl_lds[(j * NW + wave) * 2 + 1] = t1;
      } else {
        float* pp = part_lds + (rg * ST_BPO_MAX + j) * 8 + kq;
        pp[0] = t0;
        pp[4] = t1;
      }
    }
  }
  ST_STAMP(w, opl, 8);
  __syncthreads();
  ST_STAMP(w, opl, 9);

  // ---------------- 6. epilogue of the CU's rows (residual add / SwiGLU, the reference's op-by-op roundings) + store
  {
    const int opw = KS == 1 ? NW : NW / 4;                       // owners per workgroup
    const int jmax = batches_of(o->nbatch, w.cu * opw, owners);   // owner 0 of the CU has the most batches
    const int n0 = o->n0, n01 = o->n01;
    const unsigned short* __restrict__ resid = (o->flags & ST_F_RESID) ? o->resid : nullptr;
    for (int t = tid; t < jmax * opw * 2; t += NT) {
      const int i = t & 1, ow = (t >> 1) % opw, j = (t >> 1) / opw;
      const int b = w.cu * opw + ow + j * owners;
      if (b >= o->nbatch) continue;
      float v0, v1 = 0.f;
      if constexpr (KS == 1) {
        v0 = resl_lds[(j * NW + ow) * 2 + i];
        if (swiglu) v1 = resl_lds[(j * NW + ow) * 2 + 1];
      } else {
        const float4 pp = *reinterpret_cast<const float4*>(part_lds + (ow * ST_BPO_MAX + j) * 8 + i * 4);
        v0 = (pp.x + pp.y) + (pp.z + pp.w);
      }
      if (swiglu) {
        if (i == 0 && b < N) {
          const float g = round_bf16_or_f16(v0, BF16), up = round_bf16_or_f16(v1, BF16);
          const float sl = round_bf16_or_f16(g / (1.f + __expf(-g)), BF16);
          o->y[0][b] = f32_to_act<BF16>(sl * up);
        }
        continue;
      }
      const int r = b * 2 + i;
      if (r >= N) continue;
      const int sg = (r >= n0) + (r >= n01);
      const int row = r - (sg == 0 ? 0 : (sg == 1 ? n0 : n01));
      float v = v0;
      if (resid) v = act_to_f32<BF16>(resid[row]) + round_bf16_or_f16(v, BF16);
      unsigned short* ysel = sg == 0 ? o->y[0] : (sg == 1 ? o->y[1] : o->y[2]);
      ysel[row] = f32_to_act<BF16>(v);
    }
  }
  (void)rowstep;
  ST_STAMP(w, opl, 10);
}

template <bool BF16>
__global__ __launch_bounds__(ST_NT) void stream_kernel(const StOp first_op, const StOp* __restrict__ prog, int op_begin, int op_end
#ifdef GBX_STAMPS
                                                        , unsigned long long* stamps
#endif
                                                        ) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  WaveCtx w;
  w.smem = smem;
  w.tid = threadIdx.x;
  w.lane = w.tid & 63;
  w.wave = __builtin_amdgcn_readfirstlane(w.tid >> 6);
  w.cu = blockIdx.x;
  w.ncu = gridDim.x;
  w.ring = smem + ST_LDS_RING + w.wave * (ST_RS * ST_SLOT);
  w.ri = w.wi = w.inflight = 0;
  w.prog = prog;
  w.first = (CStOp*)__builtin_amdgcn_kernarg_segment_ptr();                        // first_op is the first kernel argument
  w.op_begin = op_begin;
  w.op_end = op_end;
  (void)first_op;
#ifdef GBX_STAMPS
  w.stamps = stamps;
  w.opl = 0;
  if (w.lane == 0) stamps[(static_cast<size_t>(w.cu) * ST_NW + w.wave) * 64 + 63] = __builtin_amdgcn_s_memrealtime();
#endif
  IssueCur cur{};
  cur.op = op_begin;
  cur_load(cur, prog, w.first, op_begin, op_end, w.cu, w.wave, w.ncu);
#pragma unroll 1
  for (int opi = op_begin; opi < op_end; ++opi) {
    CStOp* o = cprog(prog, w.first, op_begin, opi);
    if (o->type != ST_GEMV) continue;
    const int v = o->bits * 8 + o->P * 2 + (o->ks == 4 ? 1 : 0);
    switch (v) {
      case 2 * 8 + 1 * 2 + 0: run_gemv_op<2, 1, 1, BF16>(opi, cur, w); break;
      case 2 * 8 + 2 * 2 + 0: run_gemv_op<2, 2, 1, BF16>(opi, cur, w); break;
      case 2 * 8 + 1 * 2 + 1: run_gemv_op<2, 1, 4, BF16>(opi, cur, w); break;
      case 2 * 8 + 2 * 2 + 1: run_gemv_op<2, 2, 4, BF16>(opi, cur, w); break;
      case 3 * 8 + 1 * 2 + 0: run_gemv_op<3, 1, 1, BF16>(opi, cur, w); break;
      case 3 * 8 + 2 * 2 + 0: run_gemv_op<3, 2, 1, BF16>(opi, cur, w); break;
      case 3 * 8 + 1 * 2 + 1: run_gemv_op<3, 1, 4, BF16>(opi, cur, w); break;
      case 3 * 8 + 2 * 2 + 1: run_gemv_op<3, 2, 4, BF16>(opi, cur, w); break;
      case 4 * 8 + 1 * 2 + 0: run_gemv_op<4, 1, 1, BF16>(opi, cur, w); break;
      case 4 * 8 + 2 * 2 + 0: run_gemv_op<4, 2, 1, BF16>(opi, cur, w); break;
      case 4 * 8 + 1 * 2 + 1: run_gemv_op<4, 1, 4, BF16>(opi, cur, w); break;
      default: run_gemv_op<4, 2, 4, BF16>(opi, cur, w); break;
    }
    __syncthreads();        // the next op rewrites the activation image and the result slots
  }
  __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));     // no LDS-DMA may outlive the workgroup's LDS allocation
#ifdef GBX_STAMPS
  if (w.lane == 0) stamps[(static_cast<size_t>(w.cu) * ST_NW + w.wave) * 64 + 62] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace

hipError_t launch_stream(const void* dev_prog, const void* host_first_op, int op_begin, int op_end, bool bf16, hipStream_t s) {
  static int ncu = 0;
  static DeviceOnce attr_done;
  if (ncu == 0) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return e;
    ncu = prop.multiProcessorCount;
  }
  if (attr_done.pending()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done.done();
  }
  const StOp* prog = static_cast<const StOp*>(dev_prog);
  const StOp first = *static_cast<const StOp*>(host_first_op);
#ifdef GBX_STAMPS
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(strtoull(getenv("GBX_STAMP_PTR") ? getenv("GBX_STAMP_PTR") : "0", nullptr, 0));
  if (bf16) hipLaunchKernelGGL((stream_kernel<true>), dim3(ncu), dim3(ST_NT), ST_LDS_BYTES, s, first, prog, op_begin, op_end, stamps);
  else hipLaunchKernelGGL((stream_kernel<false>), dim3(ncu), dim3(ST_NT), ST_LDS_BYTES, s, first, prog, op_begin, op_end, stamps);
#else
  if (bf16) hipLaunchKernelGGL((stream_kernel<true>), dim3(ncu), dim3(ST_NT), ST_LDS_BYTES, s, first, prog, op_begin, op_end);
  else hipLaunchKernelGGL((stream_kernel<false>), dim3(ncu), dim3(ST_NT), ST_LDS_BYTES, s, first, prog, op_begin, op_end);
#endif
  return hipGetLastError();
}

}  // namespace gbx
