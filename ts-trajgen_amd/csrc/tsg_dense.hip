// Dense contractions of the path on the gfx950 matrix cores (exact-fp32 v_mfma_f32_32x32x2_f32):
//   * tsg_gemm_f32            one-off table products (road_emb.W_ih^T, node_emb.emb_fc^T, GAT projection)
//   * tsg_lstm_step_batched   the per-expansion recurrent cell: gathered state rows x W^T + fused gate epilogue
//
// Tiling (both kernels): 256 threads = 4 waves; block tile 128 rows x 128 columns, BK = 32.
// Wave w owns rows [32w, 32w+32) and all four 32-column sub-tiles -> 4 accumulators of 16 VGPRs.
// For the LSTM the 128 columns of a tile are the 4 gates (i,f,g,o) of 32 hidden units (the host
// interleaves the weight columns that way), so a lane holds all four gate pre-activations of one
// (row, unit) in the same accumulator register index and the cell update needs no cross-lane traffic.
// A tile in LDS as [row][33] (pad 1: conflict-free ds_read_b32 of 32 rows at fixed k),
// B tile as [k][128] (lanes read consecutive columns).
#include "tsg_engine_dev.h"

#ifndef TSG_ABLATE
#define TSG_ABLATE 0  // timing-only ablation builds (tools/build_variant.sh); 0 in the shipped library
#endif

#define BM 128
#define BN 128
#define BK 32
#define A_LD 33

struct alignas(16) TileLds {  // b rows are written as float4 (ds_write_b128): 16-byte aligned by construction
  float a[BM][A_LD];
  float b[BK][BN];
};

__device__ __forceinline__ void mma_tile(const TileLds& t, int wave, int lane, tsg_f32x16 acc[4]) {
  const int r = wave * 32 + (lane & 31);
  const int kh = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < BK; kk += 2) {
    const float av = t.a[r][kk + kh];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float bv = t.b[kk + kh][g * 32 + (lane & 31)];
      acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[g], 0, 0, 0);
    }
  }
}

// ------------------------------------------------------------------------------------ plain GEMM
// 128 x 128 tile per workgroup, k in steps of 32.  The next k-step's operands are requested from global memory
// (16-byte loads where the leading dimensions and base addresses allow: `vec` bit 0 for A, bit 1 for Bt) before
// the matrix-core work of the current one, so that their latency runs under it.
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda,
                                                       const float* __restrict__ Bt, int ldb,
                                                       const float* __restrict__ bias, float* __restrict__ C,
                                                       int ldc, int M, int N, int K, int elu, int vec) {
  __shared__ TileLds t;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int row0 = blockIdx.y * BM, col0 = blockIdx.x * BN;
  tsg_f32x16 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[g][i] = 0.0f;
  float ra[4][4], rb[4][4];
  const int a_r = tid >> 3, a_k = (tid & 7) * 4;   // A tile: 128 rows x 32 k, rows a_r + 32 p
  const int b_k = tid >> 5, b_c = (tid & 31) * 4;  // B tile: 32 k x 128 cols, k rows b_k + 8 p
  auto fetch = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int gr = row0 + a_r + 32 * p, gk = k0 + a_k;
      const float* src = A + (size_t)gr * lda + gk;
      if ((vec & 1) && gr < M && gk + 3 < K) {
        const float4 v = *reinterpret_cast<const float4*>(src);
        ra[p][0] = v.x, ra[p][1] = v.y, ra[p][2] = v.z, ra[p][3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[p][i] = (gr < M && gk + i < K) ? src[i] : 0.0f;
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int gk = k0 + b_k + 8 * p, gc = col0 + b_c;
      const float* src = Bt + (size_t)gk * ldb + gc;
      if ((vec & 2) && gk < K && gc + 3 < N) {
        const float4 v = *reinterpret_cast<const float4*>(src);
        rb[p][0] = v.x, rb[p][1] = v.y, rb[p][2] = v.z, rb[p][3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[p][i] = (gk < K && gc + i < N) ? src[i] : 0.0f;
      }
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int i = 0; i < 4; ++i) t.a[a_r + 32 * p][a_k + i] = ra[p][i];
      *reinterpret_cast<float4*>(&t.b[b_k + 8 * p][b_c]) = make_float4(rb[p][0], rb[p][1], rb[p][2], rb[p][3]);
    }
    __syncthreads();
    if (k0 + BK < K) fetch(k0 + BK);
    mma_tile(t, wave, lane, acc);
    __syncthreads();
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int gc = col0 + g * 32 + (lane & 31);
    if (gc >= N) continue;
    const float bv = bias ? bias[gc] : 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int gr = row0 + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      if (gr < M) {
        float v = acc[g][i] + bv;
        if (elu) v = v > 0.0f ? v : expm1f(v);
        C[(size_t)gr * ldc + gc] = v;
      }
    }
  }
}

// C = act(A . Bt + bias); elu != 0 closes every element with ELU (the GAT layer's activation, gat.py:114).
// Library-internal (tsg_gat.hip); the exported tsg_gemm_f32 is the elu == 0 case.
int tsg_gemm_f32_act(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C, int ldc, int M,
                     int N, int K, int elu, void* stream) {
  if (!A || !Bt || !C || M < 0 || N <= 0 || K <= 0 || lda < K || ldb < N || ldc < N) return TSG_E_INVALID;
  if (M == 0) return 0;
  dim3 grid(tsg_cdiv(N, BN), tsg_cdiv(M, BM));
  if (grid.y > 65535u) return TSG_E_LIMIT;
  const int vec = ((lda % 4 == 0 && (uintptr_t)A % 16 == 0) ? 1 : 0) | ((ldb % 4 == 0 && (uintptr_t)Bt % 16 == 0) ? 2 : 0);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, lda, Bt, ldb, bias, C, ldc, M, N, K,
                     elu, vec);
  TSG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tsg_gemm_f32(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C, int ldc,
                            int M, int N, int K, void* stream) {
  return tsg_gemm_f32_act(A, lda, Bt, ldb, bias, C, ldc, M, N, K, 0, stream);
}

// ------------------------------------------------------------------------------------ LSTM layer step
struct LstmArgs {
  int layer;
  int hp;
  int n_rows_host;
  const int32_t* n_rows_dev;
  // row addressing: either explicit index arrays (tsg_lstm_step_batched) or engine work items
  const int64_t* prev_idx;
  const int64_t* out_idx;
  const int32_t* rid;
  const int32_t* tcode;
  const int32_t* items;  // engine mode when non-null: the superstep's work list (tsg_common.h)
  const int32_t* item_counts;  // engine mode: tsg_engine.item_counts
  const int32_t* pc;     // engine mode: tsg_engine.counters
  int slots;             // engine mode
  int lstm_blocks;       // workgroups of the recurrence proper; the launch may carry cand_dis_block workgroups behind them
  long long state_rows;  // explicit index arrays: rows of h / c (0 = unchecked); indices outside are dropped
  const float* h_below;
  float* h;
  float* c;
  const float* wt;     // [K][4Hp]
  const float* xproj;  // layer 0
  const float* tproj;  // layer 0
  const float* bias;   // layer >= 1
};

// Rows of one 128-row tile.  Explicit index arrays: rows [row0, row0 + 128) of the arrays.  Engine mode: the
// tile is a stretch of 128 positions of one region of the work list (tsg_common.h); returns false when none of
// them is filled in this superstep.  Fills s_prev / s_out (-1 = no row) and, for layer 0, s_rid / s_tc.
static_assert(TSG_REGION_SLOTS % BM == 0, "whole row tiles per region of the work list");
template <bool LAYER0, typename IDX>
__device__ __forceinline__ bool lstm_tile_rows(const LstmArgs& a, int rowtile, IDX* s_prev, IDX* s_out, int* s_rid,
                                               int* s_tc) {
  const int tid = threadIdx.x;
  if (a.items) {
    constexpr int TPR = TSG_REGION_SLOTS / BM;
    const int region = rowtile / TPR, p0 = (rowtile % TPR) * BM;
    if (region >= tsg_n_regions(a.slots)) return false;
    const int rs = tsg_region_size(a.slots, region);
    // item words and counters are loaded together (the words of an unfilled position are stale and dropped)
    const int p = p0 + (tid & (BM - 1));
    const int32_t* it = a.items + ((size_t)region * TSG_REGION_SLOTS + p) * IT_WORDS;
    const int4 w0 = *reinterpret_cast<const int4*>(it);      // IT_SLOT, IT_ROW, IT_PROW, IT_RID
    const int w_tc = it[IT_TCODE];
    int n_long, n_short;
    tsg_item_counts(a.item_counts, a.pc, a.slots, region, n_long, n_short);
    if (!(p0 < n_long || (n_short > 0 && p0 < rs && p0 + BM > rs - n_short))) return false;  // nothing filled here
    if (tid < BM) {
      const bool filled = tsg_item_filled(p, rs, n_long, n_short);
      s_prev[tid] = filled ? (IDX)w0.z : (IDX)-1;  // state rows (page * 64 + offset), -1 = no parent
      s_out[tid] = filled ? (IDX)w0.y : (IDX)-1;
      s_rid[tid] = filled ? w0.w : 0;
      s_tc[tid] = filled ? w_tc : 0;
    }
    return true;
  }
  const int n = a.n_rows_dev ? *a.n_rows_dev : a.n_rows_host;
  const int row0 = rowtile * BM;
  if (row0 >= n) return false;
  if (tid < BM) {
    const int m = row0 + tid;
    long long pv = -1, ov = -1;
    int rd = 0, tc = 0;
    if (m < n) {
      pv = a.prev_idx[m];
      ov = a.out_idx[m];
      if (a.state_rows > 0) {  // tsg_lstm_step.state_rows: never touch a row outside the state arrays
        if (pv >= a.state_rows) pv = -1;
        if (ov >= a.state_rows) ov = -1;
      }
      if (LAYER0) {
        rd = a.rid[m];
        tc = a.tcode[m];
      }
    }
    s_prev[tid] = (IDX)pv;
    s_out[tid] = (IDX)ov;
    s_rid[tid] = rd;
    s_tc[tid] = tc;
  }
  return true;
}

template <bool LAYER0>
__global__ __launch_bounds__(256, 2) void lstm_layer_kernel(LstmArgs a) {
  __shared__ TileLds tt[2];  // double-buffered k-tiles: one barrier per tile
  __shared__ long long s_prev[BM];
  __shared__ long long s_out[BM];
  __shared__ int s_rid[BM];
  __shared__ int s_tc[BM];
  const int hp = a.hp;
  const int ct_n = hp / 32;  // column tiles (32 hidden units each)
  // XCD-aware decode: the ct_n column tiles of one row tile share bid % 8 (one XCD's L2 serves the A rows)
  const int bid = blockIdx.x;
  const int xcd = bid & 7, within = bid >> 3;
  const int rowtile = (within / ct_n) * 8 + xcd;
  const int coltile = within % ct_n;
  if (!lstm_tile_rows<LAYER0, long long>(a, rowtile, s_prev, s_out, s_rid, s_tc)) return;  // uniform over the workgroup
#if TSG_ABLATE == 21
  return;
#endif
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __syncthreads();
  const int K = LAYER0 ? hp : 2 * hp;
  const int ldw = 4 * hp;
  const int u = coltile * 32 + (lane & 31);
  const int gcol = coltile * BN + (lane & 31);
  // The additive gate terms and the previous cell state are fetched in batches of independent gathers
  // (never one dependent round trip per accumulator row).
  tsg_f32x16 acc[4];
  float cprev[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    const long long pv = s_prev[r];
    cprev[i] = a.c[(size_t)(pv >= 0 ? pv : 0) * hp + u];  // unconditional load, masked in the epilogue
  }
  if (LAYER0) {  // accumulators start from the road term xproj[rid]: 64 independent gathers, one wait
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      const float4 xv = *reinterpret_cast<const float4*>(a.xproj + (size_t)s_rid[r] * ldw + 4 * u);  // i, f, g, o of unit u
      acc[0][i] = xv.x;
      acc[1][i] = xv.y;
      acc[2][i] = xv.z;
      acc[3][i] = xv.w;
    }
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float bv = a.bias[gcol + g * 32];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[g][i] = bv;
    }
  }
  // software pipeline: the global loads of k-tile t+1 are issued before the MFMAs of tile t and land in
  // registers while the matrix cores work; they are written to LDS after the tile's last read.
  // (named registers, not arrays: the arrays were left on the stack by the compiler)
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  unsigned am0, am1, am2, am3;
  const int a_r = tid >> 3, a_kc = (tid & 7) * 4;   // A tile: row a_r + 32p, k chunk a_kc
  const int b_kr = tid >> 5, b_cc = (tid & 31) * 4;  // B tile: k row b_kr + 8p, column chunk b_cc
#define TSG_GLOAD_A(P, RA, AM)                                                                       \
  {                                                                                                  \
    const long long ridx = below ? s_out[a_r + 32 * P] : s_prev[a_r + 32 * P];                       \
    AM = ridx >= 0 ? 0xffffffffu : 0u; /* ANDed in at the LDS write: row 0 may hold anything */      \
    RA = *reinterpret_cast<const float4*>(src + (size_t)(ridx >= 0 ? ridx : 0) * hp + kcol + a_kc);  \
  }
#define TSG_GLOAD_B(P, RB) \
  RB = *reinterpret_cast<const float4*>(a.wt + (size_t)(k0 + b_kr + 8 * P) * ldw + coltile * BN + b_cc);
#define TSG_LSTORE_A(P, RA, AM)                                              \
  {                                                                          \
    float* dst = &tt[buf].a[a_r + 32 * P][a_kc];                                   \
    dst[0] = __uint_as_float(__float_as_uint(RA.x) & AM);                    \
    dst[1] = __uint_as_float(__float_as_uint(RA.y) & AM);                    \
    dst[2] = __uint_as_float(__float_as_uint(RA.z) & AM);                    \
    dst[3] = __uint_as_float(__float_as_uint(RA.w) & AM);                    \
  }
#define TSG_LSTORE_B(P, RB) *reinterpret_cast<float4*>(&tt[buf].b[b_kr + 8 * P][b_cc]) = RB;
  auto gload = [&](int k0) {
    const bool below = !LAYER0 && (k0 < hp);
    const int kcol = below ? k0 : (!LAYER0 ? k0 - hp : k0);
    const float* src = below ? a.h_below : a.h;
    TSG_GLOAD_A(0, ra0, am0) TSG_GLOAD_A(1, ra1, am1) TSG_GLOAD_A(2, ra2, am2) TSG_GLOAD_A(3, ra3, am3)
    TSG_GLOAD_B(0, rb0) TSG_GLOAD_B(1, rb1) TSG_GLOAD_B(2, rb2) TSG_GLOAD_B(3, rb3)
  };
  auto lstore = [&](int buf) {
    TSG_LSTORE_A(0, ra0, am0) TSG_LSTORE_A(1, ra1, am1) TSG_LSTORE_A(2, ra2, am2) TSG_LSTORE_A(3, ra3, am3)
    TSG_LSTORE_B(0, rb0) TSG_LSTORE_B(1, rb1) TSG_LSTORE_B(2, rb2) TSG_LSTORE_B(3, rb3)
  };
  gload(0);
  lstore(0);
  __syncthreads();
#if TSG_ABLATE == 22
  for (int k0 = 0, cur = 0; k0 < BK; k0 += BK, cur ^= 1) {
#else
  for (int k0 = 0, cur = 0; k0 < K; k0 += BK, cur ^= 1) {
#endif
#if TSG_ABLATE >= 11
    const bool more = false;
#else
    const bool more = k0 + BK < K;
#endif
    if (more) gload(k0 + BK);
    mma_tile(tt[cur], wave, lane, acc);
    if (more) lstore(cur ^ 1);  // the other buffer was last read before the previous barrier
    __syncthreads();
  }
  // fused cell update: lane holds i,f,g,o of (row, unit) in acc[0..3][i].  Layer 0 still owes the time
  // term tproj[tcode] (an 11 MB, cache-resident table): fetched here four rows ahead of its use.
  float4 tq[4];
  auto tload = [&](int i) {
    const int r = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    return *reinterpret_cast<const float4*>(a.tproj + (size_t)s_tc[r] * ldw + 4 * u);
  };
  if (LAYER0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) tq[i] = tload(i);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    float4 tv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (LAYER0) {
      tv = tq[i & 3];
      if (i + 4 < 16) tq[i & 3] = tload(i + 4);
    }
    const float zi = acc[0][i] + tv.x, zf = acc[1][i] + tv.y;
    const float zg = acc[2][i] + tv.z, zo = acc[3][i] + tv.w;
    const float cp = s_prev[r] >= 0 ? cprev[i] : 0.0f;
    const float cn = tsg_sigmoid(zf) * cp + tsg_sigmoid(zi) * tsg_tanh(zg);
    const float hn = tsg_sigmoid(zo) * tsg_tanh(cn);
    if (s_out[r] < 0) continue;  // no row at this position of the tile
    const size_t o = (size_t)s_out[r] * hp + u;
#if TSG_ABLATE >= 13
    if (cn == 12345.0f) a.c[o] = hn;
#else
    a.c[o] = cn;
    a.h[o] = hn;
#endif
  }
}

// ------------------------------------------------------------------------------------ LSTM layer step, split-f16 form
// Same contraction on the f16 matrix cores with fp32-grade products: every fp32 value x is carried as two
// f16 numbers, major = f16(x) and minor = f16((x - major) * 2^11), i.e. 22 significant bits.  A product
// a*b is accumulated (fp32 accumulators) as  2^11 a_maj b_maj + a_maj b_min + a_min b_maj  and scaled by
// 2^-11 in the epilogue; the dropped a_min b_min term is 2^-22 relative.  v_mfma_f32_32x32x16_f16 runs 16 k
// in 32 cycles where v_mfma_f32_32x32x2_f32 takes 64 cycles for 2 k: three passes cost 3/16 of the fp32 form.
//   * hidden states live in HBM as one 32-bit word per element: major in the low half, minor in the high half
//     (tsg_engine.state_format = 1); the epilogue writes that word, the attention reads it back as
//     major + minor * 2^-11.  Cell states stay fp32.
//   * A fragments come straight from the gathered state rows into registers (a wave's 32 rows are its own);
//     within a 32-k tile lane half h owns k = 16h .. 16h+15 (one 64-byte run per row), the weights' LDS read
//     offset applies the same permutation.
//   * weights are pre-split and pre-tiled on the host as the LDS image [col tile][k tile][major|minor][128][32] f16.
typedef _Float16 tsg_h8 __attribute__((ext_vector_type(8)));
typedef unsigned tsg_u4 __attribute__((ext_vector_type(4)));

// One k-tile of the weights in LDS: [major|minor][128 columns][32 k] f16, 64 bytes per column, no padding: the
// four 16-byte chunks of column c sit at chunk ^ ((c >> 2) & 3), which makes both the stores (one column per
// thread) and the fragment reads (ds_read_b128, 32 columns per half wave) conflict-free.
struct __align__(16) TileB16 {
  _Float16 major[BN][32];
  _Float16 minor[BN][32];
};

__device__ __forceinline__ tsg_h8 tsg_pack_halves(unsigned w0, unsigned w1, unsigned w2, unsigned w3, unsigned w4,
                                                  unsigned w5, unsigned w6, unsigned w7, unsigned sel) {
  tsg_u4 p;
  p.x = __builtin_amdgcn_perm(w1, w0, sel);
  p.y = __builtin_amdgcn_perm(w3, w2, sel);
  p.z = __builtin_amdgcn_perm(w5, w4, sel);
  p.w = __builtin_amdgcn_perm(w7, w6, sel);
  return __builtin_bit_cast(tsg_h8, p);
}

// fp32 -> state word (major | minor << 16)
__device__ __forceinline__ unsigned tsg_split_word(float x) {
  const bool tiny = fabsf(x) < 6.103515625e-05f;  // below the f16 normal range: everything goes to the minor part
  const _Float16 mj = tiny ? (_Float16)0.0f : (_Float16)x;
  const _Float16 mn = (_Float16)((x - (float)mj) * 2048.0f);
  return (unsigned)__builtin_bit_cast(unsigned short, mj) | ((unsigned)__builtin_bit_cast(unsigned short, mn) << 16);
}

#ifndef TSG_TQ
#define TSG_TQ 4  // rows of tproj requested ahead in layer 0's epilogue
#endif
#ifndef TSG_LSTM_STAMP
#define TSG_LSTM_STAMP 0  // diagnostic build only: where a wave of the recurrence spends its time (never shipped)
#endif
#if TSG_LSTM_STAMP
// [layer0 ? 0 : 1][workgroup % 64][phase]: sums of 10-ns ticks over waves (0 work items, 1 prologue, 2 k loop,
// 3 unused, 4 epilogue), [7] = waves.  Kept per wave in registers and added once at the end, spread over 64 lines:
// five same-address atomics per wave halved the whole engine's throughput.
__device__ unsigned long long g_lstm_stamp[2][64][8];
extern "C" int tsg_debug_read_lstm_stamps(unsigned long long* out_host, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_lstm_stamp), sizeof(g_lstm_stamp)) != hipSuccess) return -1;
  if (reset) {
    static unsigned long long z[2 * 64 * 8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lstm_stamp), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#define LSTAMP(i)                          \
  {                                        \
    const long long now_ = wall_clock64(); \
    lacc_[i] = now_ - lst_;                \
    lst_ = now_;                           \
  }
#else
#define LSTAMP(i)
#endif

template <bool LAYER0>
__device__ __forceinline__ void lstm_layer_split_body(const LstmArgs& a) {
#if TSG_LSTM_STAMP
  long long lst_ = wall_clock64();
  long long lacc_[5] = {0, 0, 0, 0, 0};
#endif
  __shared__ TileB16 tb[2];
  // per wave: the gathered state rows of one k-tile, [32 rows][32 words]; the eight 16-byte pieces of row r sit
  // at piece ^ (r & 7)
  __shared__ __align__(16) unsigned s_a[4][32 * 32];
  __shared__ int s_prev[BM];
  __shared__ int s_out[BM];
  __shared__ int s_rid[BM];
  __shared__ int s_tc[BM];
  const int hp = a.hp;
  const int ct_n = hp / 32;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, within = bid >> 3;
  const int rowtile = (within / ct_n) * 8 + xcd;
  const int coltile = within % ct_n;
  if (!lstm_tile_rows<LAYER0, int>(a, rowtile, s_prev, s_out, s_rid, s_tc)) return;  // uniform over the workgroup
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __syncthreads();
  LSTAMP(0)
  const int K = LAYER0 ? hp : 2 * hp;
  const int nkt = K / BK;
  const int ldw = 4 * hp;
  const int u = coltile * 32 + (lane & 31);
  const int gcol = coltile * BN + (lane & 31);
  const int lr = lane & 31, lh = lane >> 5;
  tsg_f32x16 acc[4];
  if (LAYER0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      const float4 xv = *reinterpret_cast<const float4*>(a.xproj + (size_t)s_rid[r] * ldw + 4 * u);  // i, f, g, o of unit u
      acc[0][i] = xv.x * 2048.0f;
      acc[1][i] = xv.y * 2048.0f;
      acc[2][i] = xv.z * 2048.0f;
      acc[3][i] = xv.w * 2048.0f;
    }
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float bv = a.bias[gcol + g * 32] * 2048.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[g][i] = bv;
    }
  }
  // The A operand: row r of the wave's 32 rows is the state row of that work item in the lower layer's output
  // (k < hp) or in its own previous state (k >= hp for layers above 0, all k for layer 0); a k-tile is 128
  // bytes of it.  Lane (lr, lh) of the matrix-core fragment needs 64 bytes of ROW lr, but a load in which every
  // lane reads its own row makes the four lanes of a quad touch four cache lines per instruction, and the
  // address path then takes twice as long as the matrix cores.  So the tile is fetched with eight lanes per
  // row (instruction j: rows (lane >> 3) + 8 j, 16-byte piece lane & 7: whole lines), written to the wave's own
  // LDS stretch, and read back in fragment order.
  const unsigned* hs_own = reinterpret_cast<const unsigned*>(a.h);
  const unsigned* hs_below = reinterpret_cast<const unsigned*>(a.h_below);
  int rprev[4], rout[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rprev[j] = s_prev[wave * 32 + (lane >> 3) + 8 * j];
    rout[j] = s_out[wave * 32 + (lane >> 3) + 8 * j];
  }
  const _Float16* wimg = reinterpret_cast<const _Float16*>(a.wt) + (size_t)coltile * nkt * (2 * BN * BK);
  unsigned* a_st = s_a[wave] + (lane >> 3) * 32 + (((lane & 7) ^ ((lane >> 3) & 7)) << 2);  // + j * 256 words
  const unsigned* a_ld = s_a[wave] + lr * 32;
  tsg_u4 ma0, ma1, ma2, ma3, nb0, nb1, nb2, nb3;  // in flight: the rows of one k-tile (ma*) and the weights of one (nb*)
#define TSG_GLOAD_ROWS(KT)                                                                                  \
  {                                                                                                         \
    const int k0_ = (KT) * BK;                                                                              \
    const bool below_ = !LAYER0 && (k0_ < hp);                                                              \
    const unsigned* b_ = (below_ ? hs_below : hs_own) + ((below_ || LAYER0) ? k0_ : k0_ - hp) + 4 * (lane & 7); \
    const int r0_ = below_ ? rout[0] : rprev[0], r1_ = below_ ? rout[1] : rprev[1];                         \
    const int r2_ = below_ ? rout[2] : rprev[2], r3_ = below_ ? rout[3] : rprev[3];                         \
    ma0 = *reinterpret_cast<const tsg_u4*>(b_ + (size_t)(r0_ >= 0 ? r0_ : 0) * hp);                         \
    ma1 = *reinterpret_cast<const tsg_u4*>(b_ + (size_t)(r1_ >= 0 ? r1_ : 0) * hp);                         \
    ma2 = *reinterpret_cast<const tsg_u4*>(b_ + (size_t)(r2_ >= 0 ? r2_ : 0) * hp);                         \
    ma3 = *reinterpret_cast<const tsg_u4*>(b_ + (size_t)(r3_ >= 0 ? r3_ : 0) * hp);                         \
  }
  // rows without a parent state (row index -1) contribute zeros: masked here, when the data is first touched
#define TSG_ASTORE(KT)                                                                                      \
  {                                                                                                         \
    const bool below_ = !LAYER0 && ((KT) * BK < hp);                                                        \
    const unsigned m0_ = ~(unsigned)((below_ ? rout[0] : rprev[0]) >> 31);                                  \
    const unsigned m1_ = ~(unsigned)((below_ ? rout[1] : rprev[1]) >> 31);                                  \
    const unsigned m2_ = ~(unsigned)((below_ ? rout[2] : rprev[2]) >> 31);                                  \
    const unsigned m3_ = ~(unsigned)((below_ ? rout[3] : rprev[3]) >> 31);                                  \
    *reinterpret_cast<tsg_u4*>(a_st) = ma0 & m0_;                                                           \
    *reinterpret_cast<tsg_u4*>(a_st + 256) = ma1 & m1_;                                                     \
    *reinterpret_cast<tsg_u4*>(a_st + 512) = ma2 & m2_;                                                     \
    *reinterpret_cast<tsg_u4*>(a_st + 768) = ma3 & m3_;                                                     \
  }
#define TSG_ALOAD(C0, C1, C2, C3)                                                                           \
  {                                                                                                         \
    C0 = *reinterpret_cast<const tsg_u4*>(a_ld + (((4 * lh + 0) ^ (lr & 7)) << 2));                         \
    C1 = *reinterpret_cast<const tsg_u4*>(a_ld + (((4 * lh + 1) ^ (lr & 7)) << 2));                         \
    C2 = *reinterpret_cast<const tsg_u4*>(a_ld + (((4 * lh + 2) ^ (lr & 7)) << 2));                         \
    C3 = *reinterpret_cast<const tsg_u4*>(a_ld + (((4 * lh + 3) ^ (lr & 7)) << 2));                         \
  }
  // the 16 KB weight tile is 1 024 pieces of 16 bytes; thread tid takes pieces tid, tid + 256, ...: every load
  // of a wave reads 1 KB of consecutive bytes.  Piece p: plane p / 512, column (p % 512) / 4, chunk p % 4.
  auto gload_w = [&](int kt) {
    const tsg_u4* w4 = reinterpret_cast<const tsg_u4*>(wimg + (size_t)kt * (2 * BN * BK)) + tid;
    nb0 = w4[0]; nb1 = w4[256]; nb2 = w4[512]; nb3 = w4[768];
  };
  const int w_off = (tid >> 2) * 32 + 8 * ((tid & 3) ^ ((tid >> 4) & 3));  // column tid / 4 (+ 64), swizzled chunk
  auto lstore = [&](int buf) {
    *reinterpret_cast<tsg_u4*>(&tb[buf].major[0][0] + w_off) = nb0;
    *reinterpret_cast<tsg_u4*>(&tb[buf].major[64][0] + w_off) = nb1;
    *reinterpret_cast<tsg_u4*>(&tb[buf].minor[0][0] + w_off) = nb2;
    *reinterpret_cast<tsg_u4*>(&tb[buf].minor[64][0] + w_off) = nb3;
  };
  const int b_sw = (lr >> 2) & 3;  // the same swizzle on the read side: column g * 32 + lr
  tsg_u4 ca0, ca1, ca2, ca3;
  TSG_GLOAD_ROWS(0)
  gload_w(0);
  TSG_ASTORE(0)
  if (nkt > 1) TSG_GLOAD_ROWS(1)
  lstore(0);
  TSG_ALOAD(ca0, ca1, ca2, ca3)
  __syncthreads();
  LSTAMP(1)
  const tsg_h8 scale = {(_Float16)2048.0f, (_Float16)2048.0f, (_Float16)2048.0f, (_Float16)2048.0f,
                        (_Float16)2048.0f, (_Float16)2048.0f, (_Float16)2048.0f, (_Float16)2048.0f};
  // one k-tile: the rows of tile KT+1 (requested one step ago) go to the wave's LDS stretch, whose previous
  // content (tile KT) is already in ca*; then the rows of tile KT+2 and the weights of tile KT+1 are requested
#ifndef TSG_IGLP
#define TSG_IGLP -1  // scheduling experiments: 0 / 1 = __builtin_amdgcn_iglp_opt(n) in the k-step, 2 = sched_group_barrier pattern
#endif
#if TSG_IGLP == 0 || TSG_IGLP == 1
#define TSG_KSTEP_SCHED __builtin_amdgcn_iglp_opt(TSG_IGLP);
#elif TSG_IGLP == 2
#define TSG_KSTEP_SCHED                                                                                     \
  _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                                                        \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); /* 2 DS reads */                                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); /* 3 MFMA */                                        \
  }
#else
#define TSG_KSTEP_SCHED
#endif
#define TSG_KSTEP(KT, CUR, MORE)                                                                            \
  {                                                                                                         \
    TSG_KSTEP_SCHED                                                                                         \
    if (MORE) {                                                                                             \
      gload_w((KT) + 1);                                                                                    \
      TSG_ASTORE((KT) + 1)                                                                                  \
      if ((KT) + 2 < nkt) TSG_GLOAD_ROWS((KT) + 2)                                                          \
    }                                                                                                       \
    const TileB16& t_ = tb[CUR];                                                                            \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                         \
      const tsg_u4 w0 = s == 0 ? ca0 : ca2, w1 = s == 0 ? ca1 : ca3;                                        \
      const tsg_h8 amaj = tsg_pack_halves(w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, 0x05040100u);     \
      const tsg_h8 amin = tsg_pack_halves(w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, 0x07060302u);     \
      const tsg_h8 amaj_s = amaj * scale;                                                                   \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                       \
        const int col = g * 32 + lr;                                                                        \
        const tsg_h8 bmaj = *reinterpret_cast<const tsg_h8*>(&t_.major[col][8 * ((2 * lh + s) ^ b_sw)]);    \
        const tsg_h8 bmin = *reinterpret_cast<const tsg_h8*>(&t_.minor[col][8 * ((2 * lh + s) ^ b_sw)]);    \
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(amaj_s, bmaj, acc[g], 0, 0, 0);                     \
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(amaj, bmin, acc[g], 0, 0, 0);                       \
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(amin, bmaj, acc[g], 0, 0, 0);                       \
      }                                                                                                     \
    }                                                                                                       \
    if (MORE) {                                                                                             \
      lstore((CUR) ^ 1);                                                                                    \
      TSG_ALOAD(ca0, ca1, ca2, ca3)                                                                         \
      __syncthreads();                                                                                      \
    }                                                                                                       \
  }
  int kt = 0, cur = 0;
  for (; kt + 1 < nkt; ++kt, cur ^= 1) TSG_KSTEP(kt, cur, true)
  // previous cell states, for the epilogue: requested before the last k-tile, where the prefetch registers of
  // the loop are free (held across the whole loop they are sixteen registers too many)
  float cprev[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    const int pv = s_prev[r];
    cprev[i] = a.c[(size_t)(pv >= 0 ? pv : 0) * hp + u];
  }
  TSG_KSTEP(kt, cur, false)
#undef TSG_KSTEP
#undef TSG_GLOAD_ROWS
#undef TSG_ASTORE
#undef TSG_ALOAD
  LSTAMP(2)
  float4 tq[TSG_TQ];
  auto tload = [&](int i) {
    const int r = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    return *reinterpret_cast<const float4*>(a.tproj + (size_t)s_tc[r] * ldw + 4 * u);
  };
  if (LAYER0) {
#pragma unroll
    for (int i = 0; i < TSG_TQ; ++i) tq[i] = tload(i);
  }
  unsigned* hs_out = reinterpret_cast<unsigned*>(a.h);
  const float inv = 1.0f / 2048.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    float4 tv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (LAYER0) {
      tv = tq[i % TSG_TQ];
      if (i + TSG_TQ < 16) tq[i % TSG_TQ] = tload(i + TSG_TQ);
    }
    const float zi = acc[0][i] * inv + tv.x, zf = acc[1][i] * inv + tv.y;
    const float zg = acc[2][i] * inv + tv.z, zo = acc[3][i] * inv + tv.w;
    const float cp = s_prev[r] >= 0 ? cprev[i] : 0.0f;
    const float cn = tsg_sigmoid(zf) * cp + tsg_sigmoid(zi) * tsg_tanh(zg);
    const float hn = tsg_sigmoid(zo) * tsg_tanh(cn);
    if (s_out[r] < 0) continue;  // no row at this position of the tile
    const size_t o = (size_t)s_out[r] * hp + u;
    a.c[o] = cn;
    hs_out[o] = tsg_split_word(hn);
  }
  LSTAMP(4)
#if TSG_LSTM_STAMP
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* g_ = g_lstm_stamp[LAYER0 ? 0 : 1][blockIdx.x & 63];
    for (int i = 0; i < 5; ++i) atomicAdd(&g_[i], (unsigned long long)lacc_[i]);
    atomicAdd(&g_[7], 1ull);
  }
#endif
}

template <bool LAYER0>
__global__ __launch_bounds__(256, 3) void lstm_layer_split_kernel(LstmArgs a) {
  lstm_layer_split_body<LAYER0>(a);
}

// Layer 0 of an engine superstep: the recurrence's own workgroups, then the superstep's candidate distances and
// control step (cand_dis_block) as extra workgroups, which find room in the half-empty second round of this launch.
__global__ __launch_bounds__(256, 3) void lstm_layer0_split_cand_kernel(LstmArgs a, EngineDev D) {
  if ((int)blockIdx.x >= a.lstm_blocks) {
    cand_dis_block(D, (int)blockIdx.x - a.lstm_blocks);
    return;
  }
  lstm_layer_split_body<true>(a);
}

static int check_model_dims(const tsg_model* m) {
  if (!m) return TSG_E_INVALID;
  if (m->hp <= 0 || (m->hp % 64) != 0 || m->hp > 512) return TSG_E_INVALID;
  if (m->n_layers < 1 || m->n_layers > TSG_MAX_LAYERS) return TSG_E_INVALID;
  return 0;
}

int tsg_launch_lstm_layer(const tsg_model* m, int layer, int max_rows, const int32_t* n_rows_dev, int n_rows_host,
                          const int64_t* prev_idx, const int64_t* out_idx, const int32_t* rid, const int32_t* tcode,
                          const tsg_engine* eng, const float* h_below, float* h, float* c,
                          hipStream_t stream, int split, const EngineDev* tail, long long state_rows) {
  LstmArgs a;
  a.state_rows = state_rows;
  a.layer = layer;
  a.hp = m->hp;
  a.n_rows_host = n_rows_host;
  a.n_rows_dev = n_rows_dev;
  a.prev_idx = prev_idx;
  a.out_idx = out_idx;
  a.rid = rid;
  a.tcode = tcode;
  a.items = eng ? eng->items : nullptr;
  a.item_counts = eng ? eng->item_counts : nullptr;
  a.pc = eng ? eng->counters : nullptr;
  a.slots = eng ? eng->slots : 0;
  if (eng) max_rows = (int)tsg_item_positions(eng->slots);  // row tiles of every region of the work list
  a.h_below = h_below;
  a.h = h;
  a.c = c;
  a.wt = split ? reinterpret_cast<const float*>(layer == 0 ? m->whh0_s : m->wl_s[layer])
               : (layer == 0 ? m->whh0_t : m->wl_t[layer]);
  a.xproj = m->xproj;
  a.tproj = m->tproj;
  a.bias = layer == 0 ? nullptr : m->bl[layer];
  if (!a.wt || !h || !c) return TSG_E_INVALID;
  if (layer == 0 && (!m->xproj || !m->tproj)) return TSG_E_INVALID;
  if (layer > 0 && (!a.bias || !h_below)) return TSG_E_INVALID;
  if (max_rows <= 0) return 0;
  const int rowtiles = tsg_cdiv(max_rows, BM);
  const int grid = 8 * tsg_cdiv(rowtiles, 8) * (m->hp / 32);
  a.lstm_blocks = grid;
  if (split) {
    // `tail` (engine runs, layer 0 of the split form): the candidate-distance workgroups of the superstep go
    // behind the recurrence's own, where the half-empty second round of this launch has room for them
    if (layer == 0 && tail)
      hipLaunchKernelGGL(lstm_layer0_split_cand_kernel, dim3(grid + cand_dis_blocks(tail->e, tail->g)), dim3(256), 0,
                         stream, a, *tail);
    else if (layer == 0)
      hipLaunchKernelGGL(lstm_layer_split_kernel<true>, dim3(grid), dim3(256), 0, stream, a);
    else
      hipLaunchKernelGGL(lstm_layer_split_kernel<false>, dim3(grid), dim3(256), 0, stream, a);
  } else if (layer == 0)
    hipLaunchKernelGGL(lstm_layer_kernel<true>, dim3(grid), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(lstm_layer_kernel<false>, dim3(grid), dim3(256), 0, stream, a);
  TSG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tsg_lstm_step_batched(const tsg_model* m, const tsg_lstm_step* s, void* stream) {
  int rc = check_model_dims(m);
  if (rc) return rc;
  if (!s || s->layer < 0 || s->layer >= m->n_layers || s->n_rows_host < 0) return TSG_E_INVALID;
  if (!s->prev_idx || !s->out_idx) return TSG_E_INVALID;
  if (s->layer == 0 && (!s->rid || !s->tcode)) return TSG_E_INVALID;
  return tsg_launch_lstm_layer(m, s->layer, s->n_rows_host, s->n_rows_dev, s->n_rows_host, s->prev_idx, s->out_idx,
                               s->rid, s->tcode, nullptr, s->h_below, s->h, s->c, (hipStream_t)stream, 0, nullptr,
                               (long long)s->state_rows);
}
