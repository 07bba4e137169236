// l2o_lstm_bx3.h -- the LSTM gate GEMM on the bf16 matrix cores, at fp32 accuracy.
//
// Why: on gfx950 the fp32-input MFMA (v_mfma_f32_16x16x4_f32) executes at the fp32 VECTOR
// rate and does not overlap with VALU work of the same SIMD (profiles/archive_r01_r03/r01_b_microbench_*:
// 32 cycles each, 80 per tile-step = 35 % of the fused kernel), while bf16 MFMAs run on the
// matrix pipe concurrently with the VALU.  Every fp32 value is therefore split into three
// bf16 terms  x = x1 + x2 + x3  (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2):
// 24 mantissa bits), the weights likewise on the host (from float64), and
//      x.w  =  x1w1 + x1w2 + x1w3 + x2w1 + x2w2 + x3w1      (dropped terms < 2^-24)
// is accumulated on the matrix pipe: bf16 x bf16 products are exact in fp32 and the accumulator
// is fp32, so the result carries an fp32-level error (measured rms 1.7e-8 vs 7.7e-8 of a
// sequential fp32 fmaf chain).
//
// Layout.  Same tile decomposition as l2o_common.h: lane (c, q) owns the units u = 4t + q of
// coordinate c; the D operand row rho = 4q + r of M-tile t is gate r of unit 4t + q.  A chunk
// (one 20-vector times the 80 gate rows of a layer) has K = 20 inputs x 6 products + 3 bias
// levels = 123 (term, input) pairs.  A lane group q supplies 8 K-slots per MFMA, all of them
// values the lane already owns (its five units) -- no cross-lane traffic -- so the 6 x 5 = 30
// (product, unit) pairs of a lane group are PACKED into the 32 slots of FOUR
// v_mfma_f32_16x16x32_bf16 per M-tile (round 1: one MFMA per product with 5 of 8 slots used,
// six per M-tile) and the two slots left over carry the bias (B = 1.0: levels 1, 2 on q == 0,
// level 3 on q == 1).  Register r of MFMA j holds slots (2r, 2r + 1) = the table slot_*() below:
//      j = 0: (u0,u1) x1w1 | (u2,u3) x1w1 | u4: x1w1, x1w2 | bias
//      j = 1: (u0,u1) x1w2 | (u2,u3) x1w2 | (u0,u1) x1w3   | (u2,u3) x1w3
//      j = 2: (u0,u1) x2w1 | (u2,u3) x2w1 | u4: x1w3, x2w1 | (u0,u1) x2w2
//      j = 3: (u2,u3) x2w2 | (u0,u1) x3w1 | (u2,u3) x3w1   | u4: x2w2, x3w1
// (u_i = unit 4i + q; a unit pair of one split level is one v_cvt_pk_bf16_f32 result, so the B
// operand is 16 registers built with the same conversions as before plus 6 copies; MFMA 0 and 1
// need the first split level only and start while the residuals are still being computed.)
// Chunks: L1H = h1(t-1) -> layer 1 (+ b_gates1), L2A = h1(t) -> layer 2 (+ b_gates2),
//         L2B = h2(t-1) -> layer 2, L1X = the 20 ELU features of RNNProp -> layer 1.
// The 1-2 gradient features of the DM nets are applied with 20-40 VALU FMAs instead (they
// arrive last; a chunk of their own would put 20 MFMAs on the critical path).
//
// The gate weights are pre-scaled on the host so that the accumulators ARE the exp2
// arguments: rows i, f, o by -log2(e) (f including forget_bias = 1), rows j by +2 log2(e);
// the nonlinearity block works on unit pairs with v_pk_{add,mul,fma}_f32 (9 instead of 19
// VALU instructions per unit; scripts/microbench/gates_variants.hip: 317 -> 215 ns).
#pragma once
#include <type_traits>
#include "l2o_common.h"

namespace l2o {
namespace bx {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kChL1H = 0, kChL2A = 1, kChL2B = 2, kChL1X = 3;
constexpr int kFragWords = 256;     // 64 lanes x 4 dwords
// PK (template parameter below): true = the packed form (4 MFMAs per chunk and M-tile, 80 fragment registers per
// chunk), false = one MFMA per product (6 per chunk and M-tile, 60 fragment registers per chunk).  The packed form is
// the default wherever its fragments fit the 256 AGPRs next to the kernel's own registers: the DM nets (3 chunks =
// 240 registers); RNNProp (4 chunks) and the kernels that hold a problem slice in registers as well (k_unroll_cu with
// two column blocks, k_mlp_unroll) keep the 6-product form -- spills cost them more than the MFMAs (config 3 with
// spilling packed fragments: 4.8 -> 2.8 G coordinate-steps/s).
constexpr int kPack = 4;            // packed: MFMAs per (chunk, M-tile): 32 K-slots per lane group, 30 products + bias
constexpr int kProducts = 6;        // 6-product form: (x level, w level) pairs below
__host__ __device__ constexpr bool packed_default(int pre) { return pre != L2O_PRE_FC_ELU; }
__host__ __device__ constexpr int frags(bool pk) { return pk ? kPack : 3; }                 // fragments per (chunk, M-tile)
__host__ __device__ constexpr int chunk_mfmas(bool pk) { return (pk ? kPack : kProducts) * kNT; }   // 20 | 30

__host__ __device__ constexpr int nchunks(int pre) { return pre == L2O_PRE_FC_ELU ? 4 : 3; }
// word offsets inside wpack (the bf16 section follows the fp32 rows of l2o_common.h): packed fragments, 6-product
// fragments, then the input rows.  (Round 4: RNNProp's wpack carries the packed section too -- 80 KB; its register-resident
// kernels keep the 6-product form, packed_default above, but k_unroll_lds reads the packed fragments from LDS.)
__host__ __device__ constexpr int base(int pre) { return wp_rows(pre) * 64; }
__host__ __device__ constexpr int packed_words(int pre) { return nchunks(pre) * kNT * kPack * kFragWords; }
__host__ __device__ constexpr int level_words(int pre) { return nchunks(pre) * kNT * 3 * kFragWords; }
__host__ __device__ constexpr int frag_off(int pre, bool pk, int ch, int t, int j) {
  return base(pre) + (pk ? 0 : packed_words(pre)) + ((ch * kNT + t) * frags(pk) + j) * kFragWords;
}
__host__ __device__ constexpr int win_off(int pre) { return base(pre) + packed_words(pre) + level_words(pre); }
// DM nets: 2 inputs x 5 tiles rows of 64 lanes x 4 floats (the lane's four gate rows)
__host__ __device__ constexpr int win_words(int pre) { return pre == L2O_PRE_FC_ELU ? 0 : 2 * kNT * 256; }
// The gate BIASES (pre-scaled like the rows, forget_bias folded in) as fp32: [layer 0 / 1][M-tile t][lane group q][gate
// r] -- the accumulator INIT of the chunks that start a layer's accumulation (L1H, L2B), staged into LDS by every kernel
// (stage_bias) and read as one ds_read_b128 per M-tile.  Round 3: the bias used to ride in two spare K-slots of the
// gate GEMM (B = 1.0); v_mfma_f32_16x16x32_bf16 chops every product of an 8-slot group at 2^-24 of the group's LARGEST
// (scripts/microbench/mfma_round_probe.hip) and the O(1) bias was that largest: it cost the unit products next to it
// their low bits, one-sidedly -- the ~1e-5 drift of every bf16x3 kernel at T = 1000 (profiles/archive_r01_r03/r03c_drift_forms.txt).
constexpr int kBiasWords = 2 * kNT * 4 * 4;
__host__ __device__ constexpr int bias_off(int pre) { return win_off(pre) + win_words(pre); }
__host__ __device__ constexpr int words(int pre) { return packed_words(pre) + level_words(pre) + win_words(pre) + kBiasWords; }
__host__ __device__ constexpr int prod_x(int p) { return p < 3 ? 0 : (p < 5 ? 1 : 2); }
__host__ __device__ constexpr int prod_w(int p) { return p < 3 ? p : (p < 5 ? p - 3 : 0); }
// K-slot table: slot h (0 = low, 1 = high half) of register r of MFMA j carries unit slot_unit (0..4 of the lane
// group, 5 = the bias slot) at activation split level slot_x times weight split level slot_w
struct SlotDesc { int unit, x, w; };
__host__ __device__ constexpr SlotDesc slot_desc(int j, int r, int h) {
  // pairs: {first unit, x level, w level}; unit-4 registers: {x, w} of the low and of the high slot
  switch (j * 4 + r) {
    case 0: return {0 + h, 0, 0};
    case 1: return {2 + h, 0, 0};
    case 2: return {4, 0, h};               // x1w1, x1w2
    case 3: return {5, 0, h};               // bias (w = slot index; the level depends on the lane group)
    case 4: return {0 + h, 0, 1};
    case 5: return {2 + h, 0, 1};
    case 6: return {0 + h, 0, 2};
    case 7: return {2 + h, 0, 2};
    case 8: return {0 + h, 1, 0};
    case 9: return {2 + h, 1, 0};
    case 10: return {4, h, h ? 0 : 2};      // x1w3, x2w1
    case 11: return {0 + h, 1, 1};
    case 12: return {2 + h, 1, 1};
    case 13: return {0 + h, 2, 0};
    case 14: return {2 + h, 2, 0};
    default: return {4, 1 + h, h ? 0 : 1};  // x2w2, x3w1
  }
}
// bias level carried by slot h of the bias register on lane group kq (-1: none)
__host__ __device__ constexpr int bias_level(int kq, int h) { return kq == 0 ? h : (kq == 1 && h == 0 ? 2 : -1); }

// a 5-value activation vector split into the B operands of its chunk
template <bool PK>
struct BOp {
  u32x4 m[frags(PK)];              // packed: per MFMA; 6-product form: per split level
};

__device__ __forceinline__ unsigned cvt_pk(f32x2 v) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ f32x2 widen(unsigned p) {
  f32x2 r;
  r.x = __uint_as_float(p << 16);
  r.y = __uint_as_float(p & 0xffff0000u);
  return r;
}
__device__ __forceinline__ f32x2 mk2(float a, float b) {
  f32x2 r;
  r.x = a; r.y = b;
  return r;
}

// the B = 1.0 slots of the bias row.  Packed: both slots of the bias register on q == 0, its low slot on q == 1;
// 6-product form: slot 7 (the high half of register 3 of the first level) on q == 0
template <bool PK>
__device__ __forceinline__ unsigned bias_one(int q) {
  if (PK) return q == 0 ? 0x3f803f80u : (q == 1 ? 0x00003f80u : 0u);
  return q == 0 ? 0x3f800000u : 0u;
}

// one = bias_one<PK>(q) for the chunks that carry a bias row, 0 otherwise
template <bool PK>
__device__ __forceinline__ void split5(const float (&v)[kNT], unsigned one, BOp<PK>& o) {
  if constexpr (PK) {
#ifdef L2O_ABLATE_SPLIT
#pragma unroll
    for (int j = 0; j < kPack; ++j) {
      o.m[j][0] = __float_as_uint(v[0]) ^ j; o.m[j][1] = __float_as_uint(v[1]); o.m[j][2] = __float_as_uint(v[2 + (j & 1)]);
      o.m[j][3] = j ? __float_as_uint(v[4]) : one;
    }
    return;
#endif
    f32x2 a = mk2(v[0], v[1]), b = mk2(v[2], v[3]);
    const unsigned pa0 = cvt_pk(a), pb0 = cvt_pk(b), ca = cvt_pk(mk2(v[4], v[4]));
    o.m[0][0] = pa0; o.m[0][1] = pb0; o.m[0][2] = ca; o.m[0][3] = one;
    o.m[1][0] = pa0; o.m[1][1] = pb0; o.m[1][2] = pa0; o.m[1][3] = pb0;
    a -= widen(pa0); b -= widen(pb0);
    const float r1 = v[4] - __uint_as_float(ca << 16);
    const unsigned pa1 = cvt_pk(a), pb1 = cvt_pk(b), cb = cvt_pk(mk2(v[4], r1));
    o.m[2][0] = pa1; o.m[2][1] = pb1; o.m[2][2] = cb; o.m[2][3] = pa1;
    a -= widen(pa1); b -= widen(pb1);
    const float r2 = r1 - __uint_as_float(cb & 0xffff0000u);
    o.m[3][0] = pb1; o.m[3][1] = cvt_pk(a); o.m[3][2] = cvt_pk(b); o.m[3][3] = cvt_pk(mk2(r1, r2));
  } else {
    // K-slots 0..4 of the lane group = its five units, 5, 6 zero, 7 = the bias slot; one register set per level
    f32x2 a = mk2(v[0], v[1]), b = mk2(v[2], v[3]), c = mk2(v[4], 0.0f);
    unsigned pa = cvt_pk(a), pb = cvt_pk(b), pc = cvt_pk(c);
    o.m[0][0] = pa; o.m[0][1] = pb; o.m[0][2] = pc; o.m[0][3] = one;
    a -= widen(pa); b -= widen(pb); c -= widen(pc);
    pa = cvt_pk(a); pb = cvt_pk(b); pc = cvt_pk(c);
    o.m[1][0] = pa; o.m[1][1] = pb; o.m[1][2] = pc; o.m[1][3] = 0u;
    a -= widen(pa); b -= widen(pb); c -= widen(pc);
    o.m[2][0] = cvt_pk(a); o.m[2][1] = cvt_pk(b); o.m[2][2] = cvt_pk(c); o.m[2][3] = 0u;
  }
}

template <int PRE, bool PK = packed_default(PRE)>
struct NetWB {
  static constexpr int NCH = nchunks(PRE);
  static constexpr bool kPacked = PK;
  static constexpr int NF = frags(PK);
  u32x4 a[NCH][kNT][NF];           // weight fragments (A operands): per packed MFMA | per split level
  f32x4 win0[kNT], win1[kNT];      // DM: pre-scaled weights of input 0 / 1 for this lane's 4 gate rows
  float wl[kNT];                   // output Linear
  float bl;
  float fcw0[kNT], fcw1[kNT], fcb[kNT];   // RNNProp input projection (2 -> 20)
  const __attribute__((address_space(3))) f32x4* bias;   // LDS: this lane group's [layer][t] accumulator inits (set_bias)
  static constexpr bool kLdsFrags = false;               // (NetWBL: the A operands are read from LDS at issue time)
  static constexpr int kFragFence = 0;
  static constexpr int kFragDepth = 0;                   // (LDS fragments: reads kept in flight ahead of their MFMA, issue_pipelined)
  static constexpr bool kLdsWin = false;                 // (NetWBLF: win0 / win1 are read from LDS where they are used)
  static constexpr int kLdsChunk = -1;                   // (NetWBH: ONE chunk's A operands are read from LDS)
};
// The 6-product network with ONE chunk's fragments in LDS (round 5, k_mlp_unroll): the four chunks of the RNNProp net are
// 240 AGPRs, which leaves the step's VALU operands 256 registers and the allocator spilling per-thread addresses to
// scratch -- every reload then drains the whole memory queue (s_waitcnt vmcnt(0): the granule stores and the prefetched
// image loads in flight with it).  Chunk LDSCH costs 15 ds_read_b128 per step instead of 60 registers; lch points at this
// lane's 16 bytes of its fragment (M-tile 0, level 0), fragment (t, level) sits 1 KB x (3 t + level) further.
template <int PRE, int LDSCH>
struct NetWBH : NetWB<PRE, false> {
  static constexpr int kLdsChunk = LDSCH;
  static constexpr int kChunkFloats = kNT * 3 * 256;
  const __attribute__((address_space(3))) u32x4* lch;
};
// The packed DM network with its weight FRAGMENTS IN LDS (round 4, k_unroll_lds): `a` is never loaded (no registers), an
// MFMA's A operand is one ds_read_b128 from the workgroup's 60 KB fragment image -- what lets TWO waves share a SIMD
// (256 registers each) where the register-resident form needs 240 AGPRs per wave.  lfr points at this lane's 16 bytes of
// fragment 0; fragment (chunk, M-tile, MFMA j) sits at a compile-time offset (ds_read_b128 immediate, < 64 KB).
template <int PRE>
// Measured (profiles/r06_frag_pipeline_ab.txt; results bit-identical at every depth):
//   k_mlp_xcd     depth 0 / 2 / 3 / 4: 5.66 / 6.32 / 6.28 / 6.39 G (config 5, 8 replicas)       -> 4 (L2O_MX_FRAG_DEPTH, l2o_mlp_xcd.h)
//   k_unroll_cu8  depth 0 / 2 / 3:     5.69 / 6.05 / 6.03 G (config 3)                          -> 2
//   k_unroll_lds  depth 0 / 2 / 4 / 6: 8.28 / 7.23 / 7.47 / 7.60 G (config 4 on one GPU)         -> 0: there hipcc's own schedule
//                 (the reads interleaved with the GEMV passes' LDS traffic) beats the pinned alternation
#ifndef L2O_LDS_FRAG_DEPTH
#define L2O_LDS_FRAG_DEPTH 0      // k_unroll_lds (NetWBL)
#endif
#ifndef L2O_CU8_FRAG_DEPTH
#define L2O_CU8_FRAG_DEPTH 2      // k_unroll_cu8 (NetWBLF)
#endif
struct NetWBL : NetWB<PRE, true> {
  static constexpr bool kLdsFrags = true;
  static constexpr int kFragFence = 0;
  static constexpr int kFragDepth = L2O_LDS_FRAG_DEPTH;
  const __attribute__((address_space(3))) u32x4* lfr;
};

// every thread of the workgroup copies its share of the bias table into LDS (160 floats); the caller's barrier follows
__device__ __forceinline__ void stage_bias(float* lds, const float* __restrict__ wp, int pre, int tid, int nthreads) {
  for (int i = tid; i < kBiasWords; i += nthreads) lds[i] = wp[bias_off(pre) + i];
}
template <class W>
__device__ __forceinline__ void set_bias(W& w, const float* lds, int q) {
  w.bias = reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(
               (const __attribute__((address_space(3))) float*)lds) + q;
}

// FRAGS = false: the caller fills w.a itself (k_cwlstm_step stages the fragments through LDS once
// per workgroup instead of 4 x 61 KB of L2 reads)
// LDSWIN: the input-weight rows win0 / win1 are read from LDS where they are used (NetWBLF) -- passed explicitly by the
// caller: `w` is seen as its base class here, whose kLdsWin is always false
template <int PRE, bool FRAGS = true, bool PK = packed_default(PRE), int SKIP = -1, bool LDSWIN = false>
__device__ __forceinline__ void load_netw(NetWB<PRE, PK>& w, const float* __restrict__ wp, int lane) {
  const unsigned* wu = reinterpret_cast<const unsigned*>(wp);
  if (FRAGS) {
#pragma unroll
    for (int ch = 0; ch < NetWB<PRE, PK>::NCH; ++ch)
#pragma unroll
      for (int t = 0; t < kNT; ++t)
#pragma unroll
        for (int j = 0; j < frags(PK); ++j)
          if (ch != SKIP) w.a[ch][t][j] = *reinterpret_cast<const u32x4*>(wu + frag_off(PRE, PK, ch, t, j) + lane * 4);
  }
  const float* p = wp + lane;
#pragma unroll
  for (int t = 0; t < kNT; ++t) {
    if constexpr (PRE != L2O_PRE_FC_ELU && !LDSWIN) {
      w.win0[t] = *reinterpret_cast<const f32x4*>(wp + win_off(PRE) + t * 256 + lane * 4);
      if (PRE == L2O_PRE_LOGSIGN)
        w.win1[t] = *reinterpret_cast<const f32x4*>(wp + win_off(PRE) + (kNT + t) * 256 + lane * 4);
    }
    w.wl[t] = p[(wp_row_wl(PRE) + t) * 64];
    if (PRE == L2O_PRE_FC_ELU) {
      w.fcw0[t] = p[(wp_row_fc(PRE) + t) * 64];
      w.fcw1[t] = p[(wp_row_fc(PRE) + kNT + t) * 64];
      w.fcb[t] = p[(wp_row_fc(PRE) + 2 * kNT + t) * 64];
    }
  }
  w.bl = p[wp_row_bl(PRE) * 64];
}

__device__ __forceinline__ f32x4 mfma_bf(u32x4 a, u32x4 b, f32x4 c) {
#ifdef L2O_ABLATE_MFMA
  c[0] = __builtin_fmaf(__uint_as_float(a[0]), __uint_as_float(b[0]), c[0]);
  return c;
#endif
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                  0, 0);
}

// the accumulator init of LAYER (0 / 1): its pre-scaled bias, one ds_read_b128 per M-tile.  Issued as soon as the
// accumulator is dead (after the gate block that consumed it, or at kernel start) so that its latency is never exposed;
// the asm pins the loads HERE (the scheduler otherwise sinks them in front of the first MFMA that reads them)
// PIN = false: no pinning asm -- that empty asm is a USE of the loaded registers, so the compiler waits for the five reads
// right there (s_waitcnt lgkmcnt(4..0) directly behind them: one exposed LDS round trip per call).  A caller that issues
// the reads in front of a point that drains the LDS queue anyway (the step's barrier) fences them with
// __builtin_amdgcn_sched_barrier(0) instead and pays nothing (round 4, k_unroll_pair).
template <int LAYER, class W, bool PIN = true>
__device__ __forceinline__ void preload_bias(const W& w, f32x4 (&acc)[kNT]) {
#pragma unroll
  for (int t = 0; t < kNT; ++t) acc[t] = w.bias[(LAYER * kNT + t) * 4];
  if constexpr (PIN) {
#pragma unroll
    for (int t = 0; t < kNT; ++t) asm volatile("" : "+v"(acc[t]));
  }
}

// MFMAs [LO, HI) of chunk CH (index n: packed MFMA | product n / 5, M-tile n % 5 -- consecutive MFMAs hit
// different accumulators).  ZERO: the first one starts the accumulator (C = the layer's bias, see bias_off).
// LDS-resident fragments, software-pipelined (round 6): W::kFragDepth > 0 keeps that many fragment reads IN FLIGHT ahead of
// the MFMA that consumes them.  Left to itself hipcc -- in a kernel near its register limit -- reuses ONE register quad for
// every fragment: `ds_read_b128 v[114:117]; s_waitcnt lgkmcnt(0); v_mfma ... v[114:117]`, eighty times per tile step, i.e.
// every MFMA pays an LDS round trip (found in k_mlp_xcd's ISA: 55 % of its wave cycles in s_waitcnt).  A rotating set of
// kFragDepth quads, the read of fragment n + depth issued right behind MFMA n (which has read its operand by then), and
// sched_group_barriers that pin the (MFMA, DS read) alternation.
template <int PRE, int CH, int LO, int HI, bool ZERO, bool PK, class W>
__device__ __forceinline__ void issue_pipelined(const W& w, const BOp<PK>& b, f32x4 (&acc)[kNT]) {
  static_assert(PK && W::kLdsFrags, "issue_pipelined: the packed LDS-resident fragments");
  constexpr int DEP = W::kFragDepth;
  u32x4 f[DEP];
  static_for<0, DEP>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (LO + i < HI) {
      constexpr int n = LO + i;
      f[i] = w.lfr[((CH * kNT + n % kNT) * kPack + n / kNT) * 64];
    }
  });
  // (the first `depth` reads go out TOGETHER, ahead of the first MFMA: without this group the scheduler sinks each of them
  //  into the read slot of a later iteration and the depth collapses to one)
  __builtin_amdgcn_sched_group_barrier(0x100, (HI - LO) < DEP ? (HI - LO) : DEP, 0);
  static_for<LO, HI>([&](auto nc) {
    constexpr int n = decltype(nc)::value;
    constexpr int p = n / kNT, t = n % kNT, slot = (n - LO) % DEP;
    acc[t] = mfma_bf(f[slot], b.m[p], acc[t]);
    if constexpr (n + DEP < HI) {
      constexpr int m = n + DEP;
      f[slot] = w.lfr[((CH * kNT + m % kNT) * kPack + m / kNT) * 64];
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA ...
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // ... then the read that refills its quad
  });
}

template <int PRE, int CH, int LO, int HI, bool ZERO, bool PK, class W>
__device__ __forceinline__ void issue(const W& w, const BOp<PK>& b, f32x4 (&acc)[kNT]) {
  if constexpr (W::kLdsFrags && W::kFragDepth > 0) {
    issue_pipelined<PRE, CH, LO, HI, ZERO, PK, W>(w, b, acc);
    return;
  }
  static_for<LO, HI>([&](auto nc) {
    constexpr int n = decltype(nc)::value;
    constexpr int p = n / kNT, t = n % kNT;
    // ZERO: this MFMA starts the layer's accumulator -- from the layer's bias, which preload_bias() put INTO acc[t]
    // long before (the ds_read_b128 rides behind the gate block that consumed the accumulator), not from zero
    if constexpr (W::kLdsFrags) {
      static_assert(PK, "LDS-resident fragments: the packed form");
#ifdef L2O_LDS_ABL_NOFRAG   // (timing ablation only: no fragment reads -- wrong numerics)
      acc[t] = mfma_bf(b.m[(p + 1) % kPack], b.m[p], acc[t]);
#else
      acc[t] = mfma_bf(w.lfr[((CH * kNT + t) * kPack + p) * 64], b.m[p], acc[t]);
#endif
      // W::kFragFence > 0: a scheduling fence every kFragFence MFMAs -- left alone the scheduler issues ALL fragment reads of
      // a chunk (20 x 4 registers) ahead of the first MFMA, which a kernel with several tiles' state in registers cannot hold
      if constexpr (W::kFragFence > 0) {
        if ((n - LO) % W::kFragFence == W::kFragFence - 1) __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (PK) acc[t] = mfma_bf(w.a[CH][t][p], b.m[p], acc[t]);
    else if constexpr (CH == W::kLdsChunk) acc[t] = mfma_bf(w.lch[(t * 3 + prod_w(p)) * 64], b.m[prod_x(p)], acc[t]);
    else acc[t] = mfma_bf(w.a[CH][t][prod_w(p)], b.m[prod_x(p)], acc[t]);
  });
}

// acc = [-log2e*i, 2log2e*j, -log2e*(f+1), -log2e*o] of the unit pair (T0, T0+1)
template <int T0>
__device__ __forceinline__ void gates_pair(const f32x4 (&acc)[kNT], float (&c)[kNT], float (&h)[kNT]) {
  constexpr int T1 = T0 + 1;
  constexpr float k2 = 2.0f * 1.4426950408889634f;
  const f32x2 one = mk2(1.0f, 1.0f);
  const f32x2 e_i = mk2(fast_exp2(acc[T0][0]), fast_exp2(acc[T1][0]));
  const f32x2 E_j = mk2(fast_exp2(-__builtin_fabsf(acc[T0][1])), fast_exp2(-__builtin_fabsf(acc[T1][1])));
  const f32x2 e_f = mk2(fast_exp2(acc[T0][2]), fast_exp2(acc[T1][2]));
  const f32x2 e_o = mk2(fast_exp2(acc[T0][3]), fast_exp2(acc[T1][3]));
  const f32x2 dij = (one + e_i) * (one + E_j);
  const f32x2 df = one + e_f;
  const f32x2 ij = mk2(fast_rcp(dij.x), fast_rcp(dij.y));
  const f32x2 rf = mk2(fast_rcp(df.x), fast_rcp(df.y));
  f32x2 tj = (one - E_j) * ij;                                   // sigmoid(i) * tanh|j|
  tj = mk2(__builtin_copysignf(tj.x, acc[T0][1]), __builtin_copysignf(tj.y, acc[T1][1]));
  const f32x2 cn = __builtin_elementwise_fma(rf, mk2(c[T0], c[T1]), tj);
  const f32x2 ca = cn * mk2(k2, k2);
  const f32x2 E_c = mk2(fast_exp2(-__builtin_fabsf(ca.x)), fast_exp2(-__builtin_fabsf(ca.y)));
  const f32x2 dro = (one + E_c) * (one + e_o);
  const f32x2 ro = mk2(fast_rcp(dro.x), fast_rcp(dro.y));
  const f32x2 hh = (one - E_c) * ro;                             // tanh|c'| * sigmoid(o)
  c[T0] = cn.x; c[T1] = cn.y;
  h[T0] = __builtin_copysignf(hh.x, cn.x);
  h[T1] = __builtin_copysignf(hh.y, cn.y);
}

// Sonnet LSTM nonlinearities (gates i, j, f, o; forget_bias folded into the weights) for the
// lane's five units; same algebra as l2o::lstm_gates5 (5 v_exp + 3 v_rcp per unit).
__device__ __forceinline__ void gates5(const f32x4 (&acc)[kNT], float (&c)[kNT], float (&h)[kNT]) {
#ifdef L2O_ABLATE_TRANS
#pragma unroll
  for (int t = 0; t < kNT; ++t) {
    c[t] = (acc[t][2] * 0.25f + 0.75f) * c[t] + (acc[t][0] * 0.25f + 0.5f) * (acc[t][1] * 0.5f);
    h[t] = (c[t] * 0.5f) * (acc[t][3] * 0.25f + 0.5f);
  }
  return;
#endif
#ifdef L2O_ABLATE_GATES
#pragma unroll
  for (int t = 0; t < kNT; ++t) { c[t] = acc[t][2] + acc[t][0]; h[t] = acc[t][1] + acc[t][3]; }
  return;
#endif
  gates_pair<0>(acc, c, h);
  gates_pair<2>(acc, c, h);
  constexpr float k2 = 2.0f * 1.4426950408889634f;
  const float e_i = fast_exp2(acc[4][0]), E_j = fast_exp2(-__builtin_fabsf(acc[4][1]));
  const float e_f = fast_exp2(acc[4][2]), e_o = fast_exp2(acc[4][3]);
  const float ij = fast_rcp((1.0f + e_i) * (1.0f + E_j)), rf = fast_rcp(1.0f + e_f);
  const float cn = __builtin_fmaf(rf, c[4], __builtin_copysignf((1.0f - E_j) * ij, acc[4][1]));
  const float E_c = fast_exp2(-__builtin_fabsf(cn * k2));
  const float ro = fast_rcp((1.0f + E_c) * (1.0f + e_o));
  c[4] = cn;
  h[4] = __builtin_copysignf((1.0f - E_c) * ro, cn);
}

// The same nonlinearities with plain (non-packed) fp32 VALU instructions, stage-major over the five units.  Used
// where bf16 MFMAs are interleaved with the gate math: beside MFMAs a v_pk_*_f32 costs its issue slot plus ~13
// cycles (MI355X_MICROARCH.md, "price of one filler beside MFMAs"), a plain VALU instruction only its slot.
__device__ __forceinline__ void gates5_scalar(const f32x4 (&acc)[kNT], float (&c)[kNT], float (&h)[kNT]) {
  constexpr float k2 = 2.0f * 1.4426950408889634f;
  float e_i[kNT], E_j[kNT], e_f[kNT], e_o[kNT], ij[kNT], rf[kNT], cn[kNT], E_c[kNT], ro[kNT];
#pragma unroll
  for (int t = 0; t < kNT; ++t) e_i[t] = fast_exp2(acc[t][0]);
#pragma unroll
  for (int t = 0; t < kNT; ++t) E_j[t] = fast_exp2(-__builtin_fabsf(acc[t][1]));
#pragma unroll
  for (int t = 0; t < kNT; ++t) e_f[t] = fast_exp2(acc[t][2]);
#pragma unroll
  for (int t = 0; t < kNT; ++t) e_o[t] = fast_exp2(acc[t][3]);
#pragma unroll
  for (int t = 0; t < kNT; ++t) ij[t] = fast_rcp((1.0f + e_i[t]) * (1.0f + E_j[t]));
#pragma unroll
  for (int t = 0; t < kNT; ++t) rf[t] = fast_rcp(1.0f + e_f[t]);
#pragma unroll
  for (int t = 0; t < kNT; ++t)
    cn[t] = __builtin_fmaf(rf[t], c[t], __builtin_copysignf((1.0f - E_j[t]) * ij[t], acc[t][1]));
#pragma unroll
  for (int t = 0; t < kNT; ++t) E_c[t] = fast_exp2(-__builtin_fabsf(cn[t] * k2));
#pragma unroll
  for (int t = 0; t < kNT; ++t) ro[t] = fast_rcp((1.0f + E_c[t]) * (1.0f + e_o[t]));
#pragma unroll
  for (int t = 0; t < kNT; ++t) {
    c[t] = cn[t];
    h[t] = __builtin_copysignf((1.0f - E_c[t]) * ro[t], cn[t]);
  }
}

// Everything that needs this step's gradient (see l2o::lstm_finish for the contract):
// acc1 must hold chunk L1H (h1(t-1), bias), acc2 chunk L2B (h2(t-1)).  On return s holds the
// new state, b1 / b2 the split h1(t) / h2(t) (the next step's L1H / L2B operands) and, with
// NEXT, acc1 the next step's chunk L1H.  Returns the Linear output (before tanh / scale).
// shadow(): caller work that does not feed the network (loss reductions, stores), placed behind the issue of chunk
// L2A -- the one stretch of the step where the wave only waits for the matrix pipe.
struct NoShadow { __device__ __forceinline__ void operator()() const {} };
// REARM: leave acc1 / acc2 re-initialised with the biases for the NEXT step (the persistent unroll kernels); tile_step
// (a fresh pair of accumulators per call) switches it off -- the pinned loads would be 10 dead ds_read_b128 per tile
template <int PRE, bool NEXT, bool PK, class Shadow = NoShadow, bool REARM = true, class W = NetWB<PRE, PK>>
__device__ __forceinline__ float finish(const W& w, TileState& s, BOp<PK>& b1, BOp<PK>& b2,
                                        f32x4 (&acc1)[kNT], f32x4 (&acc2)[kNT], float in0, float in1, unsigned one,
                                        int q, PhaseClock& pc, Shadow&& shadow = Shadow()) {
  constexpr int kN = chunk_mfmas(PK);
  if constexpr (PRE == L2O_PRE_FC_ELU) {   // (constexpr: chunk L1X does not exist for the DM nets)
    float fc[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t)
      fc[t] = eluf_(__builtin_fmaf(w.fcw1[t], in1, __builtin_fmaf(w.fcw0[t], in0, w.fcb[t])));
    BOp<PK> bf;
    split5<PK>(fc, 0u, bf);
    issue<PRE, kChL1X, 0, kN, false>(w, bf, acc1);
  } else if constexpr (W::kLdsWin) {
    // the DM nets' input-weight rows from LDS (round 5, k_unroll_cu8: 20 / 40 registers per wave that the 256-register
    // budget of two waves per SIMD does not have): one ds_read_b128 per (feature, M-tile), like the bias table
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
      acc1[t] += w.lwin[t * 64] * in0;
      if (PRE == L2O_PRE_LOGSIGN) acc1[t] += w.lwin[(kNT + t) * 64] * in1;
    }
  } else {
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
      acc1[t] += w.win0[t] * in0;
      if (PRE == L2O_PRE_LOGSIGN) acc1[t] += w.win1[t] * in1;
    }
  }
  pc.drain(acc1);
  pc.mark(5);
  gates5(acc1, s.c1, s.h1);
#ifndef L2O_FINISH_PINNED_LDSFRAGS
#define L2O_FINISH_PINNED_LDSFRAGS 0
#endif
  // (LDS-resident fragments: the pinning asm makes the wave wait for the five bias reads on the spot -- behind the
  //  fragment reads of eight waves; the default leaves the wait to the first MFMA that needs acc1: -1..2 %, profiles/archive_r04/r04x_*)
  constexpr bool kPinBias = !W::kLdsFrags || L2O_FINISH_PINNED_LDSFRAGS;
  if (REARM || NEXT) preload_bias<0, W, kPinBias>(w, acc1);   // acc1 is dead: the next step's layer-1 accumulator init
  pc.mark(6);
  split5<PK>(s.h1, one, b1);
  issue<PRE, kChL2A, 0, kN, false>(w, b1, acc2);
  if constexpr (!std::is_same<typename std::decay<Shadow>::type, NoShadow>::value) {
    // the caller's work between the MFMAs: in program order the wave would issue the kN MFMAs back to back
    // (16 cycles each, 4 needed) and only then the shadow
    shadow();
#ifndef L2O_SHADOW_VALU_PER_MFMA
#define L2O_SHADOW_VALU_PER_MFMA 3
#endif
#pragma unroll
    for (int i = 0; i < kN; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, L2O_SHADOW_VALU_PER_MFMA, 0);     // VALU
    }
  }
  pc.mark(7);
  pc.drain(acc2);
  pc.mark(10);
  // the next step's chunk L1H rides on the matrix pipe underneath the layer-2 nonlinearities:
  // a single wave issues in order, so the MFMAs must be interleaved with the VALU stream in
  // program order (20 back-to-back MFMAs stall the wave for 20 x 17 cycles)
  __builtin_amdgcn_sched_barrier(0);
  if (NEXT) {
#ifndef L2O_NEXT_VALU_PER_MFMA
#define L2O_NEXT_VALU_PER_MFMA 3
#endif
    issue<PRE, kChL1H, 0, kN, true>(w, b1, acc1);
    gates5_scalar(acc2, s.c2, s.h2);
#pragma unroll
    for (int i = 0; i < kN; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x402, L2O_NEXT_VALU_PER_MFMA, 0);    // VALU | transcendental
    }
  } else {
    gates5(acc2, s.c2, s.h2);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (REARM) preload_bias<1, W, kPinBias>(w, acc2);          // acc2 is dead: the next step's layer-2 accumulator init
  pc.mark(8);
  if (NEXT) split5<PK>(s.h2, one, b2);
  float d0 = s.h2[0] * w.wl[0], d1 = s.h2[1] * w.wl[1];
  d0 = __builtin_fmaf(s.h2[2], w.wl[2], d0);
  d1 = __builtin_fmaf(s.h2[3], w.wl[3], d1);
  d0 = __builtin_fmaf(s.h2[4], w.wl[4], d0);
  const float d = quad_q_sum(d0 + d1);
  return d + w.bl;
}

// One optimizer-network evaluation for a 16-coordinate tile (step-granular kernel).
// (W: NetWB<PRE, PK> -- fragments in registers -- or NetWBL<PRE> -- packed fragments read from LDS at issue time)
template <int PRE, bool PK, class W>
__device__ __forceinline__ float tile_step_w(const W& w, TileState& s, float in0, float in1, int q) {
  const unsigned one = bias_one<PK>(q);
  constexpr int kN = chunk_mfmas(PK);
  BOp<PK> b1, b2;
  f32x4 acc1[kNT], acc2[kNT];
#ifdef L2O_TILE_STEP_PINNED
  preload_bias<1>(w, acc2);
  preload_bias<0>(w, acc1);
  split5<PK>(s.h2, one, b2);
#else
  // round 4: the ten bias reads go out first, UNPINNED (the pinning asm made the wave wait for them on the spot: one
  // exposed LDS round trip per tile-step, eight per step and wave in the streaming unroll), the split of h2 hides their
  // latency; the group barriers keep the scheduler from sinking the reads to the first MFMA that needs them
  preload_bias<1, W, false>(w, acc2);
  preload_bias<0, W, false>(w, acc1);
  split5<PK>(s.h2, one, b2);
  __builtin_amdgcn_sched_group_barrier(0x100, 2 * kNT, 0);
  __builtin_amdgcn_sched_group_barrier(0x002, 32, 0);
#endif
  issue<PRE, kChL2B, 0, kN, true>(w, b2, acc2);
  split5<PK>(s.h1, one, b1);
  issue<PRE, kChL1H, 0, kN, true>(w, b1, acc1);
  PhaseClock pc;
  return finish<PRE, false, PK, NoShadow, false, W>(w, s, b1, b2, acc1, acc2, in0, in1, one, q, pc);
}
template <int PRE, bool PK>
__device__ __forceinline__ float tile_step(const NetWB<PRE, PK>& w, TileState& s, float in0, float in1, int q) {
  return tile_step_w<PRE, PK, NetWB<PRE, PK>>(w, s, in0, in1, q);
}

// ---- TWO tiles stepped together on LDS-resident packed fragments (round 6, k_mlp_xcd's four-wave form) -----------------
// One wave per SIMD issues in order and has nobody to issue from while it waits for a dependent result (an accumulator
// out of the matrix pipe, an exp ahead of its rcp): 40 % of a lone wave's tile step is such waiting.  Two tiles are two
// INDEPENDENT dependent chains in one instruction stream; every fragment read (one ds_read_b128) feeds two MFMAs.  Same
// arithmetic per tile, in the same order per accumulator, as tile_step_w: results are bit-identical.
template <int PRE, int CH, class W>
__device__ __forceinline__ void issue2(const W& w, const BOp<true>& ba, const BOp<true>& bb, f32x4 (&acca)[kNT], f32x4 (&accb)[kNT]) {
  static_assert(W::kLdsFrags, "issue2: the LDS-resident packed fragments");
  static_for<0, chunk_mfmas(true)>([&](auto nc) {
    constexpr int n = decltype(nc)::value;
    constexpr int p = n / kNT, t = n % kNT;
    const u32x4 fr = w.lfr[((CH * kNT + t) * kPack + p) * 64];
    acca[t] = mfma_bf(fr, ba.m[p], acca[t]);
    accb[t] = mfma_bf(fr, bb.m[p], accb[t]);
  });
}
__device__ __forceinline__ float linear_out(const float (&h2)[kNT], const float (&wl)[kNT]) {
  float d0 = h2[0] * wl[0], d1 = h2[1] * wl[1];
  d0 = __builtin_fmaf(h2[2], wl[2], d0);
  d1 = __builtin_fmaf(h2[3], wl[3], d1);
  d0 = __builtin_fmaf(h2[4], wl[4], d0);
  return d0 + d1;
}
template <int PRE, class W>
__device__ __forceinline__ void tile_step2_w(const W& w, TileState& sa, TileState& sb, float ina0, float ina1, float inb0,
                                             float inb1, int q, float& da, float& db) {
  const unsigned one = bias_one<true>(q);
  f32x4 a1a[kNT], a2a[kNT], a1b[kNT], a2b[kNT];
  preload_bias<1, W, false>(w, a2a);
  preload_bias<1, W, false>(w, a2b);
  preload_bias<0, W, false>(w, a1a);
  preload_bias<0, W, false>(w, a1b);
  {
    BOp<true> ba, bb;
    split5<true>(sa.h2, one, ba);
    split5<true>(sb.h2, one, bb);
    issue2<PRE, kChL2B>(w, ba, bb, a2a, a2b);
  }
  BOp<true> b1a, b1b;
  split5<true>(sa.h1, one, b1a);
  split5<true>(sb.h1, one, b1b);
  issue2<PRE, kChL1H>(w, b1a, b1b, a1a, a1b);
  if constexpr (PRE == L2O_PRE_FC_ELU) {
    float fca[kNT], fcb_[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
      fca[t] = eluf_(__builtin_fmaf(w.fcw1[t], ina1, __builtin_fmaf(w.fcw0[t], ina0, w.fcb[t])));
      fcb_[t] = eluf_(__builtin_fmaf(w.fcw1[t], inb1, __builtin_fmaf(w.fcw0[t], inb0, w.fcb[t])));
    }
    BOp<true> bfa, bfb;
    split5<true>(fca, 0u, bfa);
    split5<true>(fcb_, 0u, bfb);
    issue2<PRE, kChL1X>(w, bfa, bfb, a1a, a1b);
  } else if constexpr (W::kLdsWin) {
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
      const f32x4 w0 = w.lwin[t * 64];
      a1a[t] += w0 * ina0;
      a1b[t] += w0 * inb0;
      if (PRE == L2O_PRE_LOGSIGN) {
        const f32x4 w1 = w.lwin[(kNT + t) * 64];
        a1a[t] += w1 * ina1;
        a1b[t] += w1 * inb1;
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
      a1a[t] += w.win0[t] * ina0;
      a1b[t] += w.win0[t] * inb0;
      if (PRE == L2O_PRE_LOGSIGN) { a1a[t] += w.win1[t] * ina1; a1b[t] += w.win1[t] * inb1; }
    }
  }
  gates5(a1a, sa.c1, sa.h1);
  gates5(a1b, sb.c1, sb.h1);
  split5<true>(sa.h1, one, b1a);
  split5<true>(sb.h1, one, b1b);
  issue2<PRE, kChL2A>(w, b1a, b1b, a2a, a2b);
  gates5(a2a, sa.c2, sa.h2);
  gates5(a2b, sb.c2, sb.h2);
  da = quad_q_sum(linear_out(sa.h2, w.wl)) + w.bl;
  db = quad_q_sum(linear_out(sb.h2, w.wl)) + w.bl;
}

}  // namespace bx

// ---- one interface over the two gate-GEMM forms, for the fused unroll kernels -----------
// BX = true : bf16x3 on the matrix pipe; ~200-260 weight registers -> kernels that run one
//             wave per SIMD (<= 4 waves per workgroup: 512 VGPR + AGPR per lane)
// BX = false: the fp32 MFMA form of l2o_common.h (<= 256 registers, two waves per SIMD)
// kTotal = MFMAs of one recurrent chunk (h(t-1) part of a layer), kHalf = where the fused
// kernels split them between their two GEMV passes.
// PK (BX only): packed | 6-product form of the gate GEMM (l2o_lstm_bx3.h)
template <int PRE, bool BX, bool PK = bx::packed_default(PRE)>
struct LstmCore;

template <int PRE, bool PK>
struct LstmCore<PRE, false, PK> {
  static constexpr int kTotal = 25, kHalf = 12;
  NetW<PRE> w;
  __device__ __forceinline__ void load(const float* __restrict__ wpack, int lane) { load_netw<PRE>(w, wpack, lane); }
  static constexpr int kBiasFloats = 4;
  __device__ __forceinline__ void stage_bias(float*, const float* __restrict__, int, int, int) {}
  __device__ __forceinline__ void pin() {}
  __device__ __forceinline__ void preload(f32x4 (&)[kNT], f32x4 (&)[kNT]) {}
  __device__ __forceinline__ void preload_unpinned(f32x4 (&)[kNT], f32x4 (&)[kNT]) {}
  __device__ __forceinline__ void init(const TileState&, int) {}
  template <int LO, int HI>
  __device__ __forceinline__ void issue_l1_prev(const TileState& s, f32x4 (&acc1)[kNT]) {
    lstm_issue_l1_prev<PRE, LO, HI>(w, s, acc1);
  }
  template <int LO, int HI>
  __device__ __forceinline__ void issue_l2_prev(const TileState& s, f32x4 (&acc2)[kNT]) {
    lstm_issue_l2_prev<PRE, LO, HI>(w, s, acc2);
  }
  template <bool NEXT, class Shadow = bx::NoShadow, bool REARM = true>
  __device__ __forceinline__ float finish(TileState& s, f32x4 (&acc1)[kNT], f32x4 (&acc2)[kNT], float in0, float in1,
                                          int q, PhaseClock& pc, Shadow&& shadow = Shadow()) {
    shadow();
    return lstm_finish<PRE, NEXT>(w, s, acc1, acc2, in0, in1, q, pc);
  }
  // after finish<false>: make the recurrent operands of the NEXT step current (nothing to do here)
  __device__ __forceinline__ void refresh(const TileState&) {}
};

template <int PRE, bool PK, class WT>
struct LstmCoreRegs {
  static constexpr int kTotal = bx::chunk_mfmas(PK), kHalf = kTotal / 2;
  WT w;
  bx::BOp<PK> b1, b2;      // split h1(t-1), h2(t-1): the recurrent chunks' B operands
  unsigned one;
  __device__ __forceinline__ void load(const float* __restrict__ wpack, int lane) { bx::load_netw<PRE, true, PK, WT::kLdsChunk>(w, wpack, lane); }
  // the bias table -> LDS (all threads; the caller puts a barrier between this and the first issue_*), then the lane's view
  static constexpr int kBiasFloats = bx::kBiasWords;
  __device__ __forceinline__ void stage_bias(float* lds, const float* __restrict__ wpack, int tid, int nthreads, int q) {
    bx::stage_bias(lds, wpack, PRE, tid, nthreads);
    bx::set_bias(w, lds, q);
  }
  // Pin the 180-240 fragment registers to the accumulation half of the register file.  MFMA reads its A operand
  // from AGPRs directly; left to itself the allocator parks whatever does not fit the 256 VGPRs (fragments AND
  // VALU operands) there and pays a v_accvgpr_read per use (126 of the 618 VALU instructions of a config-2 step).
  __device__ __forceinline__ void pin() {
#ifndef L2O_NO_AGPR_PIN
#pragma unroll
    for (int ch = 0; ch < WT::NCH; ++ch)
#pragma unroll
      for (int t = 0; t < kNT; ++t)
#pragma unroll
        for (int j = 0; j < bx::frags(PK); ++j)
          if (ch != WT::kLdsChunk) asm volatile("" : "+a"(w.a[ch][t][j]));
#endif
  }
  __device__ __forceinline__ void init(const TileState& s, int q) {
    one = bx::bias_one<PK>(q);
    bx::split5<PK>(s.h1, one, b1);
    bx::split5<PK>(s.h2, one, b2);
  }
  // the accumulators of the FIRST step start from the biases too (after the barrier that follows stage_bias); every
  // later step's are re-armed by finish()
  __device__ __forceinline__ void preload(f32x4 (&acc1)[kNT], f32x4 (&acc2)[kNT]) {
    bx::preload_bias<0>(w, acc1);
    bx::preload_bias<1>(w, acc2);
  }
  // the same ten reads without the pinning asm (see preload_bias): the caller fences them and drains the LDS queue later
  __device__ __forceinline__ void preload_unpinned(f32x4 (&acc1)[kNT], f32x4 (&acc2)[kNT]) {
    bx::preload_bias<0, WT, false>(w, acc1);
    bx::preload_bias<1, WT, false>(w, acc2);
  }
  template <int LO, int HI>
  __device__ __forceinline__ void issue_l1_prev(const TileState&, f32x4 (&acc1)[kNT]) {
    bx::issue<PRE, bx::kChL1H, LO, HI, true>(w, b1, acc1);
  }
  template <int LO, int HI>
  __device__ __forceinline__ void issue_l2_prev(const TileState&, f32x4 (&acc2)[kNT]) {
    bx::issue<PRE, bx::kChL2B, LO, HI, true>(w, b2, acc2);
  }
  // REARM = false: the caller re-arms the accumulators itself (preload()) at a point where the bias reads' latency is
  // free -- finish<.., true> issues them right behind the gate blocks and the pinning asm WAITS for them there, i.e.
  // in front of the split of h1 and in front of the output Linear / the x update (round 4, k_unroll_pair)
  template <bool NEXT, class Shadow = bx::NoShadow, bool REARM = true>
  __device__ __forceinline__ float finish(TileState& s, f32x4 (&acc1)[kNT], f32x4 (&acc2)[kNT], float in0, float in1,
                                          int q, PhaseClock& pc, Shadow&& shadow = Shadow()) {
    return bx::finish<PRE, NEXT, PK, Shadow, REARM, WT>(w, s, b1, b2, acc1, acc2, in0, in1, one, q, pc, static_cast<Shadow&&>(shadow));
  }
  // after finish<false>: b1 already holds split h1(t) (finish builds it for chunk L2A); split h2(t) for chunk L2B
  __device__ __forceinline__ void refresh(const TileState& s) { bx::split5<PK>(s.h2, one, b2); }
};
template <int PRE, bool PK>
struct LstmCore<PRE, true, PK> : LstmCoreRegs<PRE, PK, bx::NetWB<PRE, PK>> {};
// the 6-product form with chunk LDSCH in LDS (bx::NetWBH); stage_chunk: every thread copies its share, a barrier follows
template <int PRE, int LDSCH>
struct LstmCoreHyb : LstmCoreRegs<PRE, false, bx::NetWBH<PRE, LDSCH>> {
  static constexpr int kChunkFloats = bx::NetWBH<PRE, LDSCH>::kChunkFloats;
  __device__ __forceinline__ void stage_chunk(float* lds, const float* __restrict__ wpack, int tid, int nthreads, int lane) {
    const unsigned* wu = reinterpret_cast<const unsigned*>(wpack);
    bx::u32x4* dst = reinterpret_cast<bx::u32x4*>(lds);
    for (int i = tid; i < kChunkFloats / 4; i += nthreads) {
      const int f = i >> 6, l = i & 63;                               // fragment 3 t + level, lane
      dst[i] = *reinterpret_cast<const bx::u32x4*>(wu + bx::frag_off(PRE, false, LDSCH, f / 3, f % 3) + l * 4);
    }
    this->w.lch = reinterpret_cast<const __attribute__((address_space(3))) bx::u32x4*>(
                      (const __attribute__((address_space(3))) float*)lds) + lane;
  }
};

namespace bx {
#ifndef L2O_CU8_FRAG_FENCE
#define L2O_CU8_FRAG_FENCE 0
#endif
#ifndef L2O_CU8_LDS_WIN
#define L2O_CU8_LDS_WIN 1
#endif
template <int PRE, int DEPTH = L2O_CU8_FRAG_DEPTH>
struct NetWBLF : NetWBL<PRE> {                                                            // (k_unroll_cu8, k_mlp_xcd)
  static constexpr int kFragFence = L2O_CU8_FRAG_FENCE;
  static constexpr int kFragDepth = DEPTH;               // fragment reads in flight ahead of their MFMA (issue_pipelined)
  static constexpr bool kLdsWin = L2O_CU8_LDS_WIN && PRE != L2O_PRE_FC_ELU;
  const __attribute__((address_space(3))) f32x4* lwin;   // LDS: this lane's 16 bytes of input-weight row 0 ([rows][64 lanes][4])
  static constexpr int kWinFloats = PRE == L2O_PRE_FC_ELU ? 0 : (PRE == L2O_PRE_LOGSIGN ? 2 : 1) * kNT * 256;
};
// every thread copies its share of the input-weight rows of wpack into LDS; the caller's barrier follows
template <int PRE, int DEPTH>
__device__ __forceinline__ void stage_win(NetWBLF<PRE, DEPTH>& w, float* lds, const float* __restrict__ wp, int tid, int nthreads, int lane) {
  if constexpr (NetWBLF<PRE, DEPTH>::kLdsWin) {
    const f32x4* src = reinterpret_cast<const f32x4*>(wp + win_off(PRE));
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    for (int i = tid; i < NetWBLF<PRE, DEPTH>::kWinFloats / 4; i += nthreads) dst[i] = src[i];
    w.lwin = reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((const __attribute__((address_space(3))) float*)lds) + lane;
  }
}
}  // namespace bx

// BX packed form with the fragments in LDS (bx::NetWBL): the same interface as LstmCore, <= 256 registers per lane
template <int PRE, class WT = bx::NetWBL<PRE>>
struct LstmCoreLds {
  static constexpr int kTotal = bx::chunk_mfmas(true), kHalf = kTotal / 2;
  static constexpr int kFragWords = bx::packed_words(PRE);            // 3 chunks x 5 M-tiles x 4 MFMAs x 256 words = 60 KB
  WT w;                                                              // (bx::NetWBL<PRE>, or its fenced variant NetWBLF)
  bx::BOp<true> b1, b2;
  unsigned one;
  __device__ __forceinline__ void load(const float* __restrict__ wpack, int lane) { bx::load_netw<PRE, false, true, -1, WT::kLdsWin>(w, wpack, lane); }
  // every thread copies its share of the packed fragment section of wpack into LDS (16-byte pieces); a barrier follows
  __device__ __forceinline__ void stage_frags(float* lds, const float* __restrict__ wpack, int tid, int nthreads, int lane) {
    const bx::u32x4* src = reinterpret_cast<const bx::u32x4*>(wpack + bx::base(PRE));
    bx::u32x4* dst = reinterpret_cast<bx::u32x4*>(lds);
    for (int i = tid; i < kFragWords / 4; i += nthreads) dst[i] = src[i];
    w.lfr = reinterpret_cast<const __attribute__((address_space(3))) bx::u32x4*>(
                (const __attribute__((address_space(3))) float*)lds) + lane;
  }
  static constexpr int kBiasFloats = bx::kBiasWords;
  __device__ __forceinline__ void stage_bias(float* lds, const float* __restrict__ wpack, int tid, int nthreads, int q) {
    bx::stage_bias(lds, wpack, PRE, tid, nthreads);
    bx::set_bias(w, lds, q);
  }
  __device__ __forceinline__ void pin() {}
  __device__ __forceinline__ void init(const TileState& s, int q) {
    one = bx::bias_one<true>(q);
    bx::split5<true>(s.h1, one, b1);
    bx::split5<true>(s.h2, one, b2);
  }
  __device__ __forceinline__ void preload(f32x4 (&acc1)[kNT], f32x4 (&acc2)[kNT]) {
    bx::preload_bias<0>(w, acc1);
    bx::preload_bias<1>(w, acc2);
  }
  __device__ __forceinline__ void preload_unpinned(f32x4 (&acc1)[kNT], f32x4 (&acc2)[kNT]) {
    bx::preload_bias<0, WT, false>(w, acc1);
    bx::preload_bias<1, WT, false>(w, acc2);
  }
  template <int LO, int HI>
  __device__ __forceinline__ void issue_l1_prev(const TileState&, f32x4 (&acc1)[kNT]) {
    bx::issue<PRE, bx::kChL1H, LO, HI, true>(w, b1, acc1);
  }
  template <int LO, int HI>
  __device__ __forceinline__ void issue_l2_prev(const TileState&, f32x4 (&acc2)[kNT]) {
    bx::issue<PRE, bx::kChL2B, LO, HI, true>(w, b2, acc2);
  }
  template <bool NEXT, class Shadow = bx::NoShadow, bool REARM = true>
  __device__ __forceinline__ float finish(TileState& s, f32x4 (&acc1)[kNT], f32x4 (&acc2)[kNT], float in0, float in1,
                                          int q, PhaseClock& pc, Shadow&& shadow = Shadow()) {
    return bx::finish<PRE, NEXT, true, Shadow, REARM, WT>(w, s, b1, b2, acc1, acc2, in0, in1, one, q, pc, static_cast<Shadow&&>(shadow));
  }
  __device__ __forceinline__ void refresh(const TileState& s) { bx::split5<true>(s.h2, one, b2); }
};

}  // namespace l2o
