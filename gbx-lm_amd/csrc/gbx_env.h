// The ONE reader of the library's environment switches: every switch is an entry of the table below (name, default, what it does);
// gbx::env_switch refuses names that are not in it, so a switch cannot exist without its line of documentation (DESIGN.md section 8
// is this table).  None is needed for normal use: they are A/B and diagnostic aids.  Values are read once per call site.
#pragma once
#include <cstdlib>
#include <cstring>

namespace gbx {

struct EnvSwitch {
  const char* name;
  int dflt;
  const char* doc;
};

inline const EnvSwitch* env_table(int* n) {
  static const EnvSwitch t[] = {
      {"GBX_FORCE_GENERIC", 0, "1: every gbx_qlinear_forward call on the shape-generic kernel (diagnostic)"},
      {"GBX_CHAIN_FLOOR", 0, "measurement only, WRONG results: 2 = every decode-GEMV row read redirected to the first rows (no weight stream), 1 = one ring round per wave"},
      {"GBX_GEMV_UNEVEN", 1, "gemv_kernel: uneven wave split of the fat geometry (A/B)"},
      {"GBX_GEMV_CEPI", 1, "gemv_kernel: compile-time one-segment epilogue, MODE 5 (A/B)"},
      {"GBX_DG_R", 0, "dgemv: force rows per batch (2 / 4); 0 = per-launch default"},
      {"GBX_DG_MIN_NW", 4, "dgemv: minimum waves per workgroup of the one-round launches (4 / 8 / 16)"},
      {"GBX_DG_MR_NW", 16, "dgemv: workgroup width (4 / 8 / 16) of the one-pass launches with more than one batch per wave (gate|up at 2 bits)"},
      {"GBX_DG_P2_NW", 8, "dgemv: workgroup width (4 / 8) of the launches whose rows take two lane passes (3 / 4 bits at K = 4096); 0 = as the one-pass launches; two 8-wave workgroups per CU measured +3.9 % on the all-4-bit strategy"},
      {"GBX_DG_DEPTH", 2, "dgemv: register buffers of the multi-batch launches: 2 (default), 3, or 4 = all of a wave's (up to four) batches in flight (gate|up)"},
      {"GBX_DG_MX", 1, "dgemv: dot products on the FP4 x FP8 matrix cores (dgemv_mx.hip; 2 bits, group 128, bf16): 0 off, 1 the SwiGLU (gate|up) launches, 2 every eligible launch"},
      {"GBX_DATTN_MX", 1, "fused attention launch: q|k|v rows and the o_proj slice on the FP4 x FP8 matrix cores where the shapes allow (2 bits, group 128, bf16, D = 4096, 4 query heads per kv head); 0: vector-ALU dots (A/B)"},
      {"GBX_DATTN_COOP_MIN", 384, "fused attention launch: contexts beyond this many cached positions are swept by the kv group's 28 producer workgroups in chunks (K / V read once for all the group's heads) instead of by the head's own workgroup (tuning aid)"},
      {"GBX_DMLP_MX3", 1, "gbx_dmlp_weight_layout: 3-bit blocks (group 128, Llama-3 shapes) want their rows as a 2-bit and a 1-bit plane (gbx_repack_planes) for the matrix-core launch; 0: the reference layout and the vector form (dmlpv.hip) (A/B)"},
      {"GBX_DMLP_MX4", 1, "gbx_dmlp_forward: 4-bit blocks (group 128, Llama-3-8B shapes) on the matrix-core launch -- a 4-bit field as two 2-bit fields of one nibble (dmlp.hip); 0: the vector form (dmlpv.hip) (A/B)"},
      {"GBX_DATTN_SPIN_LOG2", 22, "fused attention / MLP launches: log2 of the polls of one hand-off before a launch gives up, raises its status word and poisons its accumulator (test hook: a small value makes the time-out path reachable in a test)"},
      {"GBX_DG_LONG_PRIO", 0, "dgemv: waves that own a batch in the last (partial) round run at s_setprio 1 (A/B)"},
      {"GBX_DG_LATE", 1, "dgemv: request batch 0 behind the activation image in the launches with an RMSNorm prologue (A/B)"},
      {"GBX_NORM_WIDE", 0, "slice-sum + residual + RMSNorm launch of batched decode: 512 threads per row (measured within noise; the 256-thread default is bit-identical to gbx_add_rmsnorm)"},
      {"GBX_ATTN_WIDE", 1, "decode attention at head_dim 128: 16 waves per head with the in-wave state merge (0: 4 waves) (A/B)"},
      {"GBX_ATTN_GQA_WIDE", 1, "batched decode attention (head_dim 128, 4 query heads per kv head, unsplit context): 8 waves per workgroup (0: 4) (A/B)"},
      {"GBX_ATTN_SPLIT_R", 0, "decode attention: force the context split (tuning aid)"},
      {"GBX_ATTN_SPLIT_MIN", 1024, "decode attention: smallest cache capacity that is split over workgroups (tuning aid)"},
      {"GBX_ATTN_SPLIT_DIV", 128, "decode attention: cached positions per split (tuning aid)"},
      {"GBX_ATTN_GROUPED", 0, "1: the grouped-query decode attention kernel at batch 1 too (tuning aid)"},
      {"GBX_BATCH_KSPLIT_MINK", 8192, "batch GEMM: smallest K that is split across workgroups (tuning aid)"},
      {"GBX_DENSE_SKINNY", 1, "dense GEMM at M <= 32 (the head of a decode batch): the row-streaming kernel (0: the 256 x 256 tile) (A/B)"},
      {"GBX_BATCH_ROWS", 0, "batch GEMM: force the activation rows staged per pass (tuning aid)"},
      {"GBX_NO_BATCH_KERNEL", 0, "1: 2 <= M <= 32 on the split-K dequantising kernel instead of the no-dequant batch GEMM (diagnostic)"},
      {"GBX_PREFILL_TILE", 0, "prefill GEMM: force the tile (11 / 12 / 13) (tuning aid)"},
      {"GBX_DENSE_NO_8PHASE", 0, "1: the register-staged round-1 tile on dequantised weights instead of the 8-phase dense GEMM (A/B)"},
      {"GBX_P2P_SPIN_LOG2", 26, "one-shot P2P all-reduce: log2 of the polls of one granule before a launch gives up and raises its error word (test hook: a small value makes the time-out path reachable in a test)"},
      {"GBX_P2P_FORCE_COARSE", 0, "test hook: gbx_p2p_inbox_create behaves as if fine-grained memory were unavailable"},
  };
  *n = static_cast<int>(sizeof(t) / sizeof(t[0]));
  return t;
}

// value of switch `name` (atoi of the variable; the table's default when unset or empty); aborts on a name the table does not know
inline int env_switch(const char* name) {
  int n;
  const EnvSwitch* t = env_table(&n);
  for (int i = 0; i < n; ++i) {
    if (strcmp(t[i].name, name) == 0) {
      const char* e = getenv(name);
      return (e && *e) ? atoi(e) : t[i].dflt;
    }
  }
  abort();      // an undocumented switch is a bug
}

}  // namespace gbx
