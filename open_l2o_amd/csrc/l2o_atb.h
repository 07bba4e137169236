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
constexpr int kAtbMaxGroups = 256 * L2O_ATB_WGS_PER_CU;   // split-K partials (persistent workgroups)

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
