// l2o_atb.h -- G = A^T B for tall-skinny fp32 operands: A [R][KA], B [R][KB], R up to millions of rows,
// KA <= 112, KB <= 192.  Every weight gradient of a recorded unroll is a block of this one product
// (meta_minimize, DM/meta.py:398-414: A = [act1 | act2 | h2 | feats | 1], B = [dz1 | dz2 | dd | du] as emitted by
// l2o_cwlstm_bwd_unroll; KA x KB = 82 x 161 (L2O-DM), 83 x 161 (LogAndSign), 103 x 181 (RNNProp)).
// Included by l2o_kernels.hip.
//
// Split-K over the rows: a persistent workgroup walks row blocks of 32, keeps the WHOLE KA x KB result in its
// accumulators (MT x NT tiles of 16 x 16 over 4 waves, v_mfma_f32_16x16x4_f32: exact fp32 products and sums, like
// the library GEMM it replaces) and writes one partial per workgroup; a second kernel adds the partials in a
// fixed order.  Operands: each 32-row block of A and B is contiguous in memory; it is read once, coalesced, staged
// in registers while the previous block is being multiplied, and laid out in LDS with a row stride = 16 mod 32
// floats so that the per-lane operand reads (ds_read_b32: lanes (m, k) -> row k, column 16 tile + m) are
// bank-conflict free.  Both operands stream from HBM exactly once: 4 (KA + KB) bytes per row.
#pragma once

#ifndef L2O_ATB_ROWS
#define L2O_ATB_ROWS 32
#endif
constexpr int kAtbRows = L2O_ATB_ROWS;     // rows per block (8 k-steps of the 16x16x4 MFMA per 32)
#ifndef L2O_ATB_NBUF
#define L2O_ATB_NBUF 2                      // LDS buffers per operand (1: single-buffered with 3 / 4 workgroups per CU:
                                           // measured slower, train step T = 100 1.62 -> 1.74 / 1.70 ms)
#endif
#ifndef L2O_ATB_WGS_PER_CU
#define L2O_ATB_WGS_PER_CU 2
#endif
constexpr int kAtbMaxGroups = 256 * 3;   // split-K partials (persistent workgroups: 2 per CU for k_atb, up to 3 for k_atb_bx3)

__host__ __device__ constexpr int atb_ld(int tiles) { return (16 * tiles) % 32 == 16 ? 16 * tiles : 16 * tiles + 16; }

// Which 16 x 16 tiles of the product a caller reads.  MASK 0: all of them (l2o_atb).  MASK 1 / 2 (l2o_cwlstm_wgrad):
// only the weight-gradient BLOCKS of A^T Bm for the DM nets (KA = 82 | 83, KB = 161) / RNNProp (103 x 181) --
//   [in | h1_prev]^T dz1,  [h1 | h2_prev]^T dz2,  h2^T dd,  (RNNProp) feats^T du,  and the bias row 1^T [dz1 | dz2 | dd | du]
// (DM/meta.py:398-414 differentiates w.r.t. exactly these; the cross blocks act1^T dz2, ... are nobody's gradient):
// 38 of 66 / 43 of 84 tiles.
__host__ __device__ constexpr bool atb_needed(int mask, int mt, int nt) {
  return mask == 0 ? true
       : mask == 1 ? ((mt <= 1 && nt <= 4) || (mt >= 1 && mt <= 3 && nt >= 5 && nt <= 9) || (mt >= 3 && nt == 10) || mt == 5)
                   : ((mt <= 2 && nt <= 4) || (mt >= 2 && mt <= 4 && nt >= 5 && nt <= 9) || (mt >= 5 && nt == 10) || mt == 6);
}
__host__ __device__ constexpr int atb_count(int mask, int MT, int NT) {
  int n = 0;
  for (int t = 0; t < MT * NT; ++t) n += atb_needed(mask, t / NT, t % NT) ? 1 : 0;
  return n;
}
// the k-th needed tile (row-major order), -1 past the end
__host__ __device__ constexpr int atb_nth(int mask, int MT, int NT, int k) {
  for (int t = 0; t < MT * NT; ++t)
    if (atb_needed(mask, t / NT, t % NT)) {
      if (k == 0) return t;
      --k;
    }
  return -1;
}

template <int MT, int NT, int MASK = 0>
__global__ __launch_bounds__(256) void k_atb(const float* __restrict__ A, const float* __restrict__ B, long R, int KA,
                                             int KB, float* __restrict__ part) {
  constexpr int LDA = atb_ld(MT), LDB = atb_ld(NT);
  constexpr int NTILES = atb_count(MASK, MT, NT), TPW = (NTILES + 3) / 4;   // (needed) tiles, per wave
  __shared__ float sA[L2O_ATB_NBUF][kAtbRows * LDA];
  __shared__ float sB[L2O_ATB_NBUF][kAtbRows * LDB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ml = lane & 15, kq = lane >> 4;
  f32x4 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // zero the padding columns once (never overwritten)
  for (int e = tid; e < L2O_ATB_NBUF * kAtbRows * LDA; e += 256) (&sA[0][0])[e] = 0.0f;
  for (int e = tid; e < L2O_ATB_NBUF * kAtbRows * LDB; e += 256) (&sB[0][0])[e] = 0.0f;
  __syncthreads();
  const long nblk = (R + kAtbRows - 1) / kAtbRows;
  // global -> registers -> LDS: wave w fetches rows w * 8 .. w * 8 + 7 of the block, one dword per (row, column)
  // and lane (coalesced 256-byte pieces of a row); measured against dwordx4 pieces of the contiguous block scattered
  // through a multiply-high row/column split: 723 vs 894 us at R = 1.6 M rows (the scatter costs more than it saves)
  constexpr int NLA = (16 * MT + 63) / 64, NLB = (16 * NT + 63) / 64;   // dword loads per lane and row
  constexpr int RPW = kAtbRows / 4;                                 // rows of a block loaded by one wave
  float ra[RPW][NLA], rb[RPW][NLB];
  auto fetch = [&](long blk) {                                      // this wave's 8 rows of the block -> registers
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const long row = blk * kAtbRows + wv * RPW + i;
      const bool ok = row < R;
#pragma unroll
      for (int u = 0; u < NLA; ++u) {
        const int col = lane + 64 * u;
        ra[i][u] = (ok && col < KA) ? A[row * KA + col] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < NLB; ++u) {
        const int col = lane + 64 * u;
        rb[i][u] = (ok && col < KB) ? B[row * KB + col] : 0.0f;
      }
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wv * RPW + i;
#pragma unroll
      for (int u = 0; u < NLA; ++u) {
        const int col = lane + 64 * u;
        if (col < 16 * MT) sA[buf][r * LDA + col] = ra[i][u];
      }
#pragma unroll
      for (int u = 0; u < NLB; ++u) {
        const int col = lane + 64 * u;
        if (col < 16 * NT) sB[buf][r * LDB + col] = rb[i][u];
      }
    }
  };
  long blk = blockIdx.x;
  int buf = 0;
  if (blk < nblk) { fetch(blk); stage(0); }
  __syncthreads();
  for (; blk < nblk; blk += gridDim.x) {
    const long nxt = blk + gridDim.x;
    if (nxt < nblk) fetch(nxt);                                     // in flight while this block is multiplied
    const float* pa = sA[buf] + kq * LDA + ml;
    const float* pb = sB[buf] + kq * LDB + ml;
    // the tile list of a wave is static once the wave index is a compile-time constant: tile W + 4 i -> (mt, nt)
    // fold into immediate LDS offsets, and operands shared by consecutive tiles are read once
    auto compute = [&](auto wc) {
      constexpr int W = decltype(wc)::value;
#pragma unroll
      for (int ks = 0; ks < kAtbRows / 4; ++ks) {
        l2o::static_for<0, TPW>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          constexpr int ti = atb_nth(MASK, MT, NT, W + 4 * i);      // the wave's i-th tile: a compile-time constant
          if constexpr (ti >= 0) {
            constexpr int mt = ti / NT, nt = ti - mt * NT;
            const float av = pa[(4 * ks) * LDA + 16 * mt];
            const float bv = pb[(4 * ks) * LDB + 16 * nt];
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
          }
        });
      }
    };
    switch (wv) {
      case 0: compute(std::integral_constant<int, 0>{}); break;
      case 1: compute(std::integral_constant<int, 1>{}); break;
      case 2: compute(std::integral_constant<int, 2>{}); break;
      default: compute(std::integral_constant<int, 3>{});
    }
#if L2O_ATB_NBUF == 2
    if (nxt < nblk) stage(buf ^ 1);
    __syncthreads();
    buf ^= 1;
#else
    __syncthreads();                                               // every wave is done with the block in LDS
    if (nxt < nblk) stage(0);
    __syncthreads();
#endif
  }
  // ---- this workgroup's partial: D[16 mt + 4 kq + r][16 nt + ml] (needed tiles only: the reduction never reads the rest)
  float* out = part + (size_t)blockIdx.x * KA * KB;
  auto store = [&](auto wc) {
    constexpr int W = decltype(wc)::value;
    l2o::static_for<0, TPW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int ti = atb_nth(MASK, MT, NT, W + 4 * i);
      if constexpr (ti >= 0) {
        constexpr int mt = ti / NT, nt = ti - mt * NT;
        const int col = 16 * nt + ml;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * mt + 4 * kq + r;
          if (row < KA && col < KB) out[(size_t)row * KB + col] = acc[i][r];
        }
      }
    });
  };
  switch (wv) {
    case 0: store(std::integral_constant<int, 0>{}); break;
    case 1: store(std::integral_constant<int, 1>{}); break;
    case 2: store(std::integral_constant<int, 2>{}); break;
    default: store(std::integral_constant<int, 3>{});
  }
}

// ---- the bf16 x 3 form (l2o_cwlstm_wgrad's default) ---------------------------------------------------------------
// Same split-K structure, same partials and reduction, but the products run on the bf16 matrix pipe at fp32 accuracy:
// every operand value is split x = x1 + x2 + x3 (three bf16 levels, RNE: v_cvt_pk_bf16_f32) on its way INTO LDS and a
// tile takes the six products  x1 y1, x1 y2, x2 y1, x2 y2, x1 y3, x3 y1  (the three dropped ones are < 2^-24 of
// |x||y|) as v_mfma_f32_16x16x32_bf16 -- 6 x 16 cycles per tile and 32-row block against 8 x 32 for the fp32 pipe
// (l2o_lstm_bx3.h 2 has the pipe's rounding model; a weight gradient is a sum of up to millions of products of
// either sign, the truncation inside a K-group is unbiased noise at 2^-24 of the group's largest product here).
//   * K of the MFMA = the ROW index of the block: lane (m, kq) of an operand holds rows 8 kq .. 8 kq + 7 of column
//     16 tile + m as one bf16x8.  A wave fetches 8 consecutive rows, lane = column, so a lane already holds exactly
//     one such octet per 64-column chunk: three ds_write_b128 (one per level) per chunk, no transpose pass.
//   * LDS: [level][column][4 octets of 16 bytes], the octet index XORed with (column >> 1) & 3.  ds_read_b128 is
//     served in four groups of 16 NON-contiguous lanes ({0-3, 12-15, 20-27}, ... MI355X_MICROARCH.md, LDS) over 64
//     banks, ds_write_b128 in eight groups of 8 contiguous lanes over 32 banks: with this XOR the reads (lane = column
//     16 tile + m, octet kq) and the writes (lanes = consecutive columns, one octet per wave) are both conflict-free
//     (the obvious (column >> 2) & 3 is a 2-way conflict on every read: +50 us of 340 at R = 1.6 M rows).
//     52 KB for 82 x 161 (3 workgroups per CU), 58 KB for 103 x 181 (2); single-buffered: the next block
//     waits in registers while this one is multiplied, other workgroups of the CU cover the barriers.
//   * tiles of a wave: CONTIGUOUS pieces of the row-major list of needed tiles (a wave stays on one or two tile rows:
//     the A-side operands are read once per tile row).
//   * (tried: s_setprio 1 / 3 around the MFMA phase: 304 / 310 vs 305 us -- nothing)
//   * fetch: clamped addresses instead of predicated loads (straight-line code, scalar base + 32-bit lane offset; the
//     predicated form of k_atb compiles to a branch per load).
template <int MT, int NT>
__host__ __device__ constexpr int atb_bx3_lds_bytes() { return 3 * 16 * (MT + NT) * 64; }
template <int MT, int NT>
__host__ __device__ constexpr int atb_bx3_wgs_per_cu() {
  return 160 * 1024 / atb_bx3_lds_bytes<MT, NT>() >= 3 ? 3 : (160 * 1024 / atb_bx3_lds_bytes<MT, NT>() >= 2 ? 2 : 1);
}

// L2O_ATB_ABLATE (scripts/microbench/atb_bx3_bench.hip only; results are wrong): 1 = no MFMAs / operand reads,
// 2 = no global loads, 3 = no split arithmetic, 4 = global loads only, 5 = per-phase clock (s_memtime sums of wave 0
// of every workgroup in g_atb_phase[]: fetch issue, multiply, barrier, split + stage, barrier, iterations)
#ifndef L2O_ATB_ABLATE
#define L2O_ATB_ABLATE 0
#endif
#if L2O_ATB_ABLATE == 5
__device__ unsigned long long g_atb_phase[8];
#define L2O_ATB_TICK(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); ph[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define L2O_ATB_TICK(k) do { } while (0)
#endif
#if L2O_ATB_ABLATE == 2
#define L2O_ATB_LOAD(p) __uint_as_float((unsigned)(size_t)(p))
#elif defined(L2O_ATB_NT)
#define L2O_ATB_LOAD(p) __builtin_nontemporal_load(reinterpret_cast<const float*>(p))
#else
#define L2O_ATB_LOAD(p) (*reinterpret_cast<const float*>(p))
#endif
// The COMPACT A of l2o_cwlstm_bwd_unroll_compact (round 5): rows [in (P) | h1(t) | h2(t) | feats | 1] in T + 1 blocks of
// R / T rows (block ts + 1 = step ts, block 0 = the state before step 0).  The product is still taken over the VIRTUAL
// columns [in | h1(t-1) | h1(t) | h2(t-1) | h2(t) | feats | 1]: a virtual column is a physical column of the row in block ts
// (the previous step's outputs) or in block ts + 1 -- byte offset of virtual column v relative to row g of block 0 (row g of
// Bm is step g / rows-per-block): h1(t-1), h2(t-1) at their physical column, everything else `slice_bytes` further on.
// A lane computes its <= 2 offsets once per block; everything downstream of the loads is unchanged.
__device__ __forceinline__ unsigned atb_compact_off(unsigned v, unsigned P, unsigned slice_bytes) {
  if (v < P) return 4u * v + slice_bytes;
  const unsigned w = v - P;
  if (w < 20u) return 4u * (P + w);                          // h1(t-1): block ts
  if (w < 40u) return 4u * (P + w - 20u) + slice_bytes;      // h1(t)
  if (w < 60u) return 4u * (P + 20u + w - 40u);              // h2(t-1): block ts
  if (w < 80u) return 4u * (P + 20u + w - 60u) + slice_bytes;   // h2(t)
  return 4u * (P + 40u + w - 80u) + slice_bytes;             // feats, 1
}

// KAs: floats per row of A in memory (KA; compact: KA - 40).  a_slice_bytes: 0 = the plain [R][KA] operand.
template <int MT, int NT, int MASK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(atb_bx3_wgs_per_cu<MT, NT>())))
void k_atb_bx3(const float* __restrict__ A, const float* __restrict__ B, long R, int KA, int KB,
               float* __restrict__ part, int KAs, unsigned a_slice_bytes, int P) {
  using l2o::bx::u32x4;
  using l2o::bx::f32x2;
  constexpr int CA = 16 * MT, CB = 16 * NT;
  constexpr int NTILES = atb_count(MASK, MT, NT), TPW = (NTILES + 3) / 4;
  static_assert(kAtbRows == 32, "one bf16 MFMA spans the 32 rows of a block");
  __shared__ u32x4 sA[3 * CA * 4];
  __shared__ u32x4 sB[3 * CB * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ml = lane & 15, kq = lane >> 4;
  f32x4 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long nblk = (R + kAtbRows - 1) / kAtbRows;
  constexpr int NLA = (CA + 63) / 64, NLB = (CB + 63) / 64;         // 64-column chunks per row
  constexpr int RPW = 8;                                            // rows of a block per wave = one K octet
  float ra[RPW][NLA], rb[RPW][NLB];
  // clamped column (byte) offsets instead of predicated loads: every load is unconditional and at a valid address
  // (a `cond ? load : 0` is compiled into a branch per load, with a wait behind each)
  const unsigned lane4 = 4u * lane, enda = 4u * (KA - 1), endb = 4u * (KB - 1);
  // loads only -- nothing here touches a loaded value, so the wave does not wait for them before it multiplies the
  // current block (the masks are applied in stage())
  auto fetch = [&](long blk) {
    // (the asm keeps the lane offsets and their 32 -> 64 bit extension in this block: scalar base + 32-bit lane
    // offset addressing instead of a hoisted 64-bit offset pair per chunk, and one live register instead of ten)
    unsigned oa[NLA], ob[NLB];
    unsigned l4 = lane4;
    asm volatile("" : "+v"(l4));
#pragma unroll
    for (int u = 0; u < NLA; ++u) {
      oa[u] = min(l4 + 256u * u, enda);
      if (a_slice_bytes) oa[u] = atb_compact_off(oa[u] >> 2, (unsigned)P, a_slice_bytes);   // (wave-uniform branch)
    }
#pragma unroll
    for (int u = 0; u < NLB; ++u) ob[u] = min(l4 + 256u * u, endb);
    // row pointers: scalar, advanced by one row unless that would pass the end (the rows past R re-read row R - 1,
    // any valid address; stage() zeroes them) -- a handful of scalar instructions per row instead of two 64-bit
    // multiplies
    const long row0 = blk * kAtbRows + wv * RPW;                    // (wave-uniform)
    const long r0 = row0 < R ? row0 : R - 1;
    const long left = R - row0;
    const int nv = left >= RPW ? RPW : (left > 0 ? (int)left : 0);  // valid rows of the octet
    const unsigned stepa = 4u * KAs, stepb = 4u * KB;
    const char* pa = reinterpret_cast<const char*>(A + r0 * KAs);
    const char* pb = reinterpret_cast<const char*>(B + r0 * KB);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
#pragma unroll
      for (int u = 0; u < NLA; ++u) ra[i][u] = L2O_ATB_LOAD(pa + oa[u]);
#pragma unroll
      for (int u = 0; u < NLB; ++u) rb[i][u] = L2O_ATB_LOAD(pb + ob[u]);
      const bool adv = i + 1 < nv;
      pa += adv ? stepa : 0u;
      pb += adv ? stepb : 0u;
    }
  };
  // 8 rows of one column -> the three bf16 levels of the octet.  rows < 8 (the last, ragged block only): the rows past
  // R are zeroed.  Columns past KA | KB need no mask: they only reach output rows | columns that are never stored.
  auto split8 = [&](const float (&v)[RPW], int rows, u32x4& l0, u32x4& l1, u32x4& l2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x2 x = l2o::bx::mk2(2 * j < rows ? v[2 * j] : 0.0f, 2 * j + 1 < rows ? v[2 * j + 1] : 0.0f);
#if L2O_ATB_ABLATE == 3
      l0[j] = __float_as_uint(x.x); l1[j] = __float_as_uint(x.y); l2[j] = 0u;
      continue;
#endif
      const unsigned c0 = l2o::bx::cvt_pk(x);
      x -= l2o::bx::widen(c0);
      const unsigned c1 = l2o::bx::cvt_pk(x);
      x -= l2o::bx::widen(c1);
      l0[j] = c0; l1[j] = c1; l2[j] = l2o::bx::cvt_pk(x);
    }
  };
  auto stage_rows = [&](int rows) {
#pragma unroll
    for (int u = 0; u < NLA; ++u) {
      const int col = lane + 64 * u;
      float v[RPW];
#pragma unroll
      for (int i = 0; i < RPW; ++i) v[i] = ra[i][u];
      u32x4 l0, l1, l2;
      split8(v, rows, l0, l1, l2);
      if (col < CA) {
        const int o = col * 4 + (wv ^ ((col >> 1) & 3));
        sA[o] = l0; sA[CA * 4 + o] = l1; sA[2 * CA * 4 + o] = l2;
      }
    }
#pragma unroll
    for (int u = 0; u < NLB; ++u) {
      const int col = lane + 64 * u;
      float v[RPW];
#pragma unroll
      for (int i = 0; i < RPW; ++i) v[i] = rb[i][u];
      u32x4 l0, l1, l2;
      split8(v, rows, l0, l1, l2);
      if (col < CB) {
        const int o = col * 4 + (wv ^ ((col >> 1) & 3));
        sB[o] = l0; sB[CB * 4 + o] = l1; sB[2 * CB * 4 + o] = l2;
      }
    }
  };
  auto stage = [&](long blk) {
    const long left = R - (blk * kAtbRows + wv * RPW);              // valid rows of this wave's octet (wave-uniform)
    if (left >= RPW) stage_rows(RPW);                                // (constant-folded: no selects)
    else stage_rows(left > 0 ? (int)left : 0);
  };
  long blk = blockIdx.x;
#if L2O_ATB_ABLATE == 4
  for (; blk < nblk; blk += gridDim.x) {
    fetch(blk);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
#pragma unroll
      for (int u = 0; u < NLA; ++u) acc[0][0] += ra[i][u];
#pragma unroll
      for (int u = 0; u < NLB; ++u) acc[0][1] += rb[i][u];
    }
  }
#endif
  if (blk < nblk) { fetch(blk); stage(blk); }
  __syncthreads();
  const u32x4* pa = sA + ml * 4 + (kq ^ ((ml >> 1) & 3));            // + (level * CA + 16 mt) * 4
  const u32x4* pb = sB + ml * 4 + (kq ^ ((ml >> 1) & 3));
#if L2O_ATB_ABLATE == 5
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#endif
  for (; blk < nblk; blk += gridDim.x) {
    const long nxt = blk + gridDim.x;
    if (nxt < nblk) fetch(nxt);                                     // in flight while this block is multiplied
    L2O_ATB_TICK(0);
    // the B-side operands of tile i + 1 are requested before the six MFMAs of tile i (two sets in registers), the A
    // side is re-read only when the tile row changes; the scheduling barriers keep the compiler from hoisting more
    // reads than that (it otherwise keeps ~20 in flight and spills at the 168 registers of 3 waves per SIMD -- and a
    // scratch reload in this loop is a vmcnt wait on the prefetched block)
    auto compute = [&](auto wc) {
      constexpr int W = decltype(wc)::value;
      u32x4 a[3], b[2][3];
      auto request_b = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int ti = atb_nth(MASK, MT, NT, W * TPW + i);
        if constexpr (i < TPW && ti >= 0) {
          constexpr int nt = ti % NT;
#pragma unroll
          for (int l = 0; l < 3; ++l) b[i & 1][l] = pb[(l * CB + 16 * nt) * 4];
        }
      };
      request_b(std::integral_constant<int, 0>{});
      l2o::static_for<0, TPW>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int ti = atb_nth(MASK, MT, NT, W * TPW + i);      // the wave's i-th tile: a compile-time constant
        if constexpr (ti >= 0) {
          constexpr int mt = ti / NT;
          if constexpr (i == 0 || atb_nth(MASK, MT, NT, W * TPW + i - 1) / NT != mt) {   // a new tile row (<= 3 per wave)
#pragma unroll
            for (int l = 0; l < 3; ++l) a[l] = pa[(l * CA + 16 * mt) * 4];
          }
          request_b(std::integral_constant<int, i + 1>{});
          __builtin_amdgcn_sched_barrier(0);
          f32x4 c = acc[i];
          c = l2o::bx::mfma_bf(a[2], b[i & 1][0], c);               // small products first
          c = l2o::bx::mfma_bf(a[0], b[i & 1][2], c);
          c = l2o::bx::mfma_bf(a[1], b[i & 1][1], c);
          c = l2o::bx::mfma_bf(a[1], b[i & 1][0], c);
          c = l2o::bx::mfma_bf(a[0], b[i & 1][1], c);
          c = l2o::bx::mfma_bf(a[0], b[i & 1][0], c);
          acc[i] = c;
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    };
#if L2O_ATB_ABLATE != 1
    switch (wv) {
      case 0: compute(std::integral_constant<int, 0>{}); break;
      case 1: compute(std::integral_constant<int, 1>{}); break;
      case 2: compute(std::integral_constant<int, 2>{}); break;
      default: compute(std::integral_constant<int, 3>{});
    }
#endif
    L2O_ATB_TICK(1);
    __syncthreads();                                                // every wave is done with the block in LDS
    L2O_ATB_TICK(2);
    if (nxt < nblk) stage(nxt);
    L2O_ATB_TICK(3);
    __syncthreads();
    L2O_ATB_TICK(4);
#if L2O_ATB_ABLATE == 5
    ph[5] += 1;
#endif
  }
#if L2O_ATB_ABLATE == 5
  if (lane == 0)
    for (int k = 0; k < 6; ++k) atomicAdd(&g_atb_phase[k], ph[k]);
#endif
  float* out = part + (size_t)blockIdx.x * KA * KB;
  auto store = [&](auto wc) {
    constexpr int W = decltype(wc)::value;
    l2o::static_for<0, TPW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int ti = atb_nth(MASK, MT, NT, W * TPW + i);
      if constexpr (ti >= 0) {
        constexpr int mt = ti / NT, nt = ti - mt * NT;
        const int col = 16 * nt + ml;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * mt + 4 * kq + r;
          if (row < KA && col < KB) out[(size_t)row * KB + col] = acc[i][r];
        }
      }
    });
  };
  switch (wv) {
    case 0: store(std::integral_constant<int, 0>{}); break;
    case 1: store(std::integral_constant<int, 1>{}); break;
    case 2: store(std::integral_constant<int, 2>{}); break;
    default: store(std::integral_constant<int, 3>{});
  }
}

// out[e] = sum_g part[g][e] in a FIXED order (bit-reproducible): a block of 256 threads owns 32 consecutive entries,
// thread (entry e, slice k) adds the partials g = k, k + 8, k + 16, ... in four interleaved accumulators, the eight
// slice sums are combined as ((0+1)+(2+3))+((4+5)+(6+7)) through LDS.  (One thread per entry walking all <= 512
// partials was latency-bound: 61 us of a 510 us training step at T = 20; now 8 loads in flight per thread and
// 1/8 of the trip count.)
// mask / KB: entries outside the needed tiles are written as zeros without reading the partials
__global__ __launch_bounds__(256) void k_atb_reduce(const float* __restrict__ part, int groups, int n,
                                                    float* __restrict__ out, int mask, int NT, int KB) {
  __shared__ float red[8][32];
  const int el = threadIdx.x & 31, k = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  bool need = e < n;
  if (need && mask) {
    const int row = e / KB, col = e - row * KB;
    need = atb_needed(mask, row >> 4, col >> 4);
  }
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  if (need) {
    int g = k;
    for (; g + 24 < groups; g += 32) {
      s0 += part[(size_t)g * n + e];
      s1 += part[(size_t)(g + 8) * n + e];
      s2 += part[(size_t)(g + 16) * n + e];
      s3 += part[(size_t)(g + 24) * n + e];
    }
    for (; g < groups; g += 8) s0 += part[(size_t)g * n + e];
  }
  red[k][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (k == 0 && e < n)
    out[e] = need ? ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) +
                        ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]))
                  : 0.0f;
}
