// The attention half of a decoder block at M == 1 as ONE launch whose hand-offs never leave an XCD ("dattn", round 5):
//
//   RMSNorm(h) -> q | k | v rows of ONE kv group -> RoPE + KV-cache append -> attention of the group's query heads
//              -> o_proj over that group's rep * 128 input columns -> fixed-point partial sums into acc[D]
//
// replaces, per block, three launches of the round-4 decode path (q|k|v GEMV, attention, o_proj GEMV + residual):
//   mx.quantized_matmul at M == 1 (gbx_lm/models/quantized_linear_gba.py:195-203) at the call sites gbx_lm/models/qllama.py:76
//   (q, k, v projections) and :96 (o_proj), with the attention between them (qllama.py:78-95) and the input RMSNorm (:137).
//
// Mapping: Llama-3's 8 kv heads = the chip's 8 XCDs.  A workgroup reads its XCD from HW_REG_XCC_ID -- that IS its kv group g -- and
// its member index r = blockIdx.x / 8 (256 workgroups, one per CU; 32 members per group under the round-robin placement measured in
// profiles/r05_xcd_barrier.txt; a placement that breaks the bijection leaves a group short of a member and the launch reports a
// time-out in its status word instead of a wrong result).  Roles inside a group (rep = query heads per kv head):
//   members [0, rep)   : attention of head g * rep + r.  They skip phase 1, prefetch 128 cached K / V rows meanwhile, poll the group's
//                        q / k / v granules, rotate, append (member 0), sweep, merge, publish 64 granules of output.
//   members [rep, 32)  : phase 1 -- dgemv's prologue (carried statistic, fp16 image, weights requested behind the image) and ONE
//                        two-row batch per wave over the group's (rep + 2) * 128 rows of the concatenated q|k|v matrix; the two
//                        rounded row totals leave as one 8-byte {tag, value} granule.
//   everyone           : phase 3 -- o_proj rows [r * D/32, (r + 1) * D/32), the group's column range only; the packed words were
//                        requested at the start of the launch and sit in registers; rep waves poll the attention granules, the
//                        workgroup builds the operand image once, every lane owns one slot of one row; partial row sums go to
//                        acc[row] as 2^-32 fixed-point integer adds (order-independent: the eight groups may finish in any order).
// A hand-off that does not arrive within spin_limit polls (a placement that broke the bijection, a member that never ran) ends the wait:
// the launch raises its status word AND the accumulator's non-finite marker acc[D], so the stream every later launch forms from these
// sums is NaN -- the failure is in-band (NaN logits, NaN logprob of the token) even for a caller that never reads the status word.
// Hand-off form (measured first, profiles/r05_xcd_barrier.txt P3): granule stores are workgroup-scope (sc0: write-through to the
// XCD's L2, where the line stays), polls are agent-scope loads (sc1: past the L1, served by that same L2) -- 0.45 us per hand-off
// against 1.05 us for the device-scope form, and correct by construction because producer and consumer share the L2 they
// meet in (the group is DEFINED by XCC_ID).  Tags come from a per-group generation word in device memory that member 0 advances at
// the end of the launch, so no memset node and no host-side epoch is needed and a graph replay is self-contained.
// The consumer of acc is the gate|up launch (dgemv x_acc / x_out: x = round(h + round(acc * 2^-32)), one rounding each like the
// reference's o_proj output and residual add), the down_proj launch clears it (dgemv zero_buf).
#pragma once
#include "dgemv_impl.h"

namespace gbx {

constexpr int DA_GROUPS = 8;                 // kv groups = XCDs
constexpr int DA_MEMBERS = 32;               // workgroups per group
constexpr int DA_NW = 16, DA_NT = DA_NW * 64;
constexpr int DA_HD = 128;
constexpr int DA_OP = 2;                     // o_proj lane passes held in registers
constexpr int DA_HAND_STRIDE = 2048;         // granules (8 bytes) per group: [0, 1024) phase 1, [1024, 1536) attention output, [1536, 1792) rotated q (cooperative sweep)
constexpr int DA_HAND_QROT = 1536;
constexpr int DA_STATE_WORDS = 64 * 16;      // u32: gen[g] at g * 16 (one 64-byte line each), status at 8 * 16, placement at 9 * 16
constexpr unsigned DA_SPIN_LIMIT = 1u << 22;   // default of DattnArgs / DmlpArgs.spin_limit
constexpr int DA_PART_STRIDE = 132;          // granules per partial softmax state: {tag, max}, {tag, denominator}, 128 x {tag, numerator}, 2 pad
constexpr int DA_PART_GRANULES = DA_GROUPS * DA_MEMBERS * 4 * DA_PART_STRIDE;      // [group][member][head of the group <= 4]

struct DattnArgs {
  const unsigned short* x;          // [D] residual stream
  const unsigned short* norm_w;     // [D] input RMSNorm weight
  const float* ssq_in;              // [DG_SSQ_SLOTS] partial sums of x^2 from the producer of x, or null
  const uint32_t* qkv_qw;           // [(n_heads + 16) * 128, D * bq / 32]: q | k | v rows concatenated
  const uint32_t* qkv_sz;           // (scale | zero << 16) companion of that matrix in batch order, two rows per batch
  unsigned short* kcache;           // [8, capacity, 128]
  unsigned short* vcache;
  const float* inv_freq;            // [64]
  const long long* pos;             // device scalar: index of the new token
  const uint32_t* o_qw;             // [D, n_heads * 128 * bo / 32]
  const unsigned short* o_s;        // [D, n_heads * 128 / go]
  const unsigned short* o_z;
  unsigned long long* acc;          // [D + 1] 2^-32 fixed point (word D: non-finite marker)
  unsigned long long* hand;         // [8][DA_HAND_STRIDE] granules
  unsigned* state;                  // [DA_STATE_WORDS]
  unsigned short* attn_out;         // optional [n_heads * 128]: the attention output (tests)
  unsigned long long* part;         // [8][32][4][DA_PART_STRIDE] granules: partial softmax states of the cooperative sweep (long contexts)
  // the previous block's MLP launch (dmlp.hip) leaves down_proj's output as fixed-point sums too:
  const long long* x_acc;           // [D + 1] or null: x <- round(x + round(x_acc * 2^-32)) in the producers' prologue (no ssq_in then)
  unsigned short* x_out;            // with x_acc: the new x, written once
  unsigned long long* zero_buf;     // or null: zero_n 8-byte words this launch clears
  int zero_n;
  int D, rep, capacity, bpw;        // bpw: phase-1 batches per producer workgroup
  int qkv_log2g, o_bits, o_log2g, traditional;
  int coop_min;                     // contexts beyond this many cached positions: cooperative sweep (the group's producers, in chunks)
  unsigned spin_limit;              // polls of one hand-off before the launch gives up (GBX_DATTN_SPIN_LOG2)
  float eps, rope_scale, sm_scale;
#ifdef GBX_STAMPS
  unsigned long long* stamps;
  // round-6 step 0 (ii): an LDS-DMA weight stream issued beside the hand-off chain (scratch/probe7, profiles/r06_block_probe.txt)
  const unsigned char* probe_src;
  int probe_mode, probe_kb_p, probe_kb_a, probe_kb_a0;
#endif
};

struct DattnPlan {
  int P;            // lane passes of a q|k|v row
  int bpw;
  size_t shmem;
};
// false: this launch does not cover the shape (the caller keeps the five-launch block)
bool dattn_plan(int D, int n_heads, int n_kv, int head_dim, int qkv_bits, int qkv_group, int o_bits, int o_group, DattnPlan* pl);
hipError_t launch_dattn(const DattnArgs& a, const DattnPlan& pl, int qkv_bits, bool bf16, hipStream_t stream);
// dmlp.hip: the MLP half of the block as one launch (gate | up + SwiGLU + XCD-local hand-off + down_proj partial sums)
struct DmlpArgs {
  const unsigned short* x;          // [K] the stream before o_proj's output is added
  const long long* x_acc;           // [K + 1] o_proj sums, 2^-32 fixed point (word K: non-finite marker)
  unsigned short* x_out;            // [K] x' = round(x + round(x_acc * 2^-32)), written once (workgroup 0)
  const unsigned short* norm_w;     // [K] post-attention RMSNorm weight
  const uint32_t* gate_qw;          // [I, K / 16]
  const uint32_t* up_qw;            // [I, K / 16]
  const uint32_t* gu_sz;            // [I][K / 128][2]: (scale | zero << 16) of (gate row n, up row n) (gbx_dgemv_pack, SwiGLU, 2 rows per batch)
  const uint32_t* down_qw;          // [D, I / 16]
  const uint32_t* down_sz;          // [D / 2][I / 128][2]: gbx_dgemv_pack of down_proj, 2 rows per batch
  unsigned long long* acc_out;      // [D + 1] down_proj sums, 2^-32 fixed point (zero on entry)
  unsigned long long* zero_buf;     // or null: zero_n words this launch clears
  unsigned long long* hand;         // [8][DA_HAND_STRIDE] granules
  unsigned* state;                  // [DA_STATE_WORDS]
  unsigned short* act_out;          // optional [I]: the SwiGLU output (tests)
  int K, I, D, zero_n, bpw, a4;     // bpw: gate|up batches per workgroup; waves [0, a4) own 4 of them, the others 2
  int gu_log2g, dn_log2g;           // dmlpv.hip (3 / 4 bits): log2 of the group sizes
  unsigned spin_limit;              // polls of the hand-off before the launch gives up (GBX_DATTN_SPIN_LOG2)
  float eps;
#ifdef GBX_STAMPS
  unsigned long long* stamps;
#endif
};

bool dmlp_plan(int D, int I, int bits_gu, int group_gu, int bits_dn, int group_dn, bool bf16, int* bpw, int* a4);
hipError_t launch_dmlp(const DmlpArgs& a, int bits, hipStream_t stream);
// dmlpv.hip: the same launch on the vector-ALU arithmetic for 3- and 4-bit rows (Llama-3-8B and Llama-3.2-3B MLP shapes)
bool dmlpv_plan(int D, int I, int bits_gu, int group_gu, int bits_dn, int group_dn, bool bf16, int* bpw, int* a4);
hipError_t launch_dmlpv(const DmlpArgs& a, int bits, hipStream_t stream);
hipError_t launch_acc_finish(const void* x, void* acc, void* out, int D, bool bf16, void* z0, void* z1, void* z2, hipStream_t stream);
// dblock.hip (round 6): the whole 2-bit block -- both launches above -- as ONE launch with a chip-wide release between o_proj's sums and the MLP
struct DblockArgs {
  // attention half (DattnArgs)
  const unsigned short* x;          // [D] residual stream
  const unsigned short* norm_w;     // [D] input RMSNorm weight
  const float* ssq_in;              // carried statistic of x, or null
  const uint32_t* qkv_qw;           // q | k | v rows concatenated
  const uint32_t* qkv_sz;
  unsigned short* kcache;           // [8, capacity, 128]
  unsigned short* vcache;
  const float* inv_freq;
  const long long* pos;
  const uint32_t* o_qw;
  const unsigned short* o_s;
  const unsigned short* o_z;
  unsigned long long* acc_o;        // [D + 1] o_proj sums (zero on entry)
  const long long* x_acc;           // [D + 1] or null: the previous block's down_proj sums
  unsigned short* x_out;            // with x_acc: x <- round(x + round(x_acc * 2^-32)), written once (write-through: the leaders read it back)
  unsigned short* attn_out;         // optional (tests)
  // MLP half (DmlpArgs)
  const unsigned short* norm2_w;    // [D] post-attention RMSNorm weight
  const uint32_t* gate_qw;
  const uint32_t* up_qw;
  const uint32_t* gu_sz;
  const uint32_t* down_qw;
  const uint32_t* down_sz;
  unsigned long long* acc_d;        // [D + 1] down_proj sums (zero on entry)
  unsigned short* x2_out;           // [D] x' = round(h + round(o_proj sums)): the stream the next launch adds down_proj's output to
  unsigned short* act_out;          // optional (tests)
  // plumbing
  unsigned long long* hand;         // [8][DB_HAND_STRIDE] granules
  unsigned* state;
  unsigned long long* zero0;        // or null: zero_n0 words this launch clears (write-through stores)
  unsigned long long* zero1;
  int zero_n0, zero_n1;
  int capacity, traditional, max_ctx;
  unsigned spin_limit;
  float eps1, eps2, rope_scale, sm_scale;
#ifdef GBX_STAMPS
  unsigned long long* stamps;
#endif
};
bool dblock_plan(int D, int I, int n_heads, int n_kv, int head_dim, int qkv_bits, int qkv_group, int o_bits, int o_group, int gu_bits, int gu_group,
                 int dn_bits, int dn_group, bool bf16);
size_t dblock_state_bytes();
size_t dblock_status_offset();
int dblock_max_ctx();
hipError_t launch_dblock(const DblockArgs& a, hipStream_t stream);

template <int BQ>
hipError_t launch_dattn_bits(const DattnArgs& a, const DattnPlan& pl, bool bf16, hipStream_t stream);

}  // namespace gbx
