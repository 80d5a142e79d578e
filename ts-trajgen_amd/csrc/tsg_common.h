// Shared host/device helpers for libtsg (gfx950 only: wave64, MFMA f32 32x32x2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tsg.h"

#define TSG_WAVE 64

#define TSG_HIP_TRY(expr)                              \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) return -(int)_e - 1000;      \
  } while (0)

#define TSG_LAUNCH_CHECK()                             \
  do {                                                 \
    hipError_t _e = hipGetLastError();                 \
    if (_e != hipSuccess) return -(int)_e - 1000;      \
  } while (0)

static inline int tsg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef float tsg_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float tsg_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, TSG_WAVE);
  return v;
}

__device__ __forceinline__ float tsg_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, TSG_WAVE));
  return v;
}

// The same reductions without the LDS crossbar (ds_bpermute): four DPP steps inside each row of 16 lanes (quad
// swaps, half-row and row mirrors: v_add_f32 ... dpp), then the four row totals are read as scalars.  The result is
// wave-uniform; all 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ float tsg_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float tsg_lane_value(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float tsg_wave_sum_u(float v) {
  v += tsg_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  v += tsg_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  v += tsg_dpp<0x141>(v);  // row_half_mirror
  v += tsg_dpp<0x140>(v);  // row_mirror
  return (tsg_lane_value(v, 0) + tsg_lane_value(v, 16)) + (tsg_lane_value(v, 32) + tsg_lane_value(v, 48));
}
__device__ __forceinline__ float tsg_wave_max_u(float v) {
  v = fmaxf(v, tsg_dpp<0xB1>(v));
  v = fmaxf(v, tsg_dpp<0x4E>(v));
  v = fmaxf(v, tsg_dpp<0x141>(v));
  v = fmaxf(v, tsg_dpp<0x140>(v));
  return fmaxf(fmaxf(tsg_lane_value(v, 0), tsg_lane_value(v, 16)), fmaxf(tsg_lane_value(v, 32), tsg_lane_value(v, 48)));
}

// v_exp_f32 / v_rcp_f32 forms (about 1e-7 absolute error, far inside the 1e-4 parity budget)
#ifndef TSG_PRECISE_GATES
#define TSG_PRECISE_GATES 0  // 1: libm expf / tanhf and a true division (ablation: do fewer near-ties flip?)
#endif
#if TSG_PRECISE_GATES
__device__ __forceinline__ float tsg_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tsg_tanh(float x) { return tanhf(x); }
#else
__device__ __forceinline__ float tsg_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tsg_tanh(float x) { return 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * x)); }
#endif

// slot-state word indices (engine.slot_state is [B][SS_WORDS = 40] int32).  The words are grouped by how often they
// change, and the frontier kernel writes a group back as whole 16-byte pieces (every store instruction of a lane is
// a request of its own to the L2, and the frontier kernel is bound by its request count):
//   words  0..7   change every superstep
//   words  8..15  the expansion the search waits for (written when a node is expanded); once the search is over
//                 (PH_FINISH) the same words hold its result: SS_FIN_* alias them
//   words 16..19  per-slot statistics, summed on the host
//   words 20..    set when a search starts (OD pair, prefix, fallback snapshot) or rarely
enum {
  SS_PHASE = 0, SS_STEP, SS_HEAP_N, SS_NODE_N, SS_EXP_N, SS_NPOP, SS_PAGES,
  SS_LAST_PAGE,  // the state page taken last (= the page of the next expansion's row once ensure_page succeeded)
  SS_CUR_EXP = 8, SS_CUR_RID, SS_CUR_C, SS_CUR_DATE_TIME, SS_CUR_COST, SS_CUR_LP, SS_CUR_DEPTH, SS_CUR_SPARE,
  SS_FIN_STATUS = SS_CUR_EXP, SS_FIN_PARENT = SS_CUR_RID, SS_FIN_RID = SS_CUR_C, SS_FIN_TCODE = SS_CUR_DATE_TIME,
  SS_FIN_DEPTH = SS_CUR_DEPTH,
  SS_ST_EVALS = 16, SS_ST_CAND, SS_ST_CHAIN, SS_ST_ITEMS,
  SS_ST_SKIPPED = 20, SS_ST_STALLS, SS_OD, SS_GEN, SS_WOFF, SS_DES, SS_DEFLEN, SS_MAXSTEP, SS_ALLOW, SS_WARM_K, SS_PLEN,
  SS_POFF, SS_BEST_LP, SS_BEST_PARENT, SS_BEST_RID, SS_BEST_TCODE, SS_HAVE_BEST,
  SS_WORDS = 40
};
// PH_FINISH: the search is over and its result is parked in the slot (SS_FIN_*); the next superstep writes the
// trajectory and returns the pages, the one after admits a new OD pair: the long chains of a superstep (pops until
// the destination, the write-out, the start of a search) never follow each other inside one launch
enum { PH_IDLE = 0, PH_WARMUP = 1, PH_SEARCH = 2, PH_EXPANDED = 3, PH_DONE = 4, PH_FINISH = 5 };
// shared counters (tsg_engine.shared_counters)
enum { CT_ITEMS0 = 0, CT_ITEMS1 = 1, CT_NEXT_OD = 2, CT_FINISHED = 3, CT_EXPANSIONS = 4, CT_MODEL_EVALS = 5, CT_ERROR = 6, CT_ITEMS_TOTAL = 7 };
// state-page control words (tsg_engine.counters): free ring [head, tail) of page ids, retry ring of aborted OD
// pairs, admission and deadlock control.  *_SNAP = the tail as it stood when the launch began: a launch consumes
// only what earlier launches produced.
enum {
  PC_HEAD = 0, PC_TAIL, PC_TAIL_SNAP, PC_RHEAD, PC_RTAIL, PC_RTAIL_SNAP, PC_STALLS, PC_ABORTS, PC_ABORT_LEVEL,
  PC_ITEM_PLANE, PC_STALL_STEP, PC_SUPERSTEP, PC_MIN_FREE, PC_ADMIT_QUOTA, PC_ABORT_BUDGET, PC_WORDS = 16
};
// work item words ([positions][12] int32); ROW / PROW = state rows (page * 64 + offset) of the node and of its parent
enum { IT_SLOT = 0, IT_ROW, IT_PROW, IT_RID, IT_TCODE, IT_NCAND, IT_DES, IT_LP, IT_DEPTH, IT_WORDS = 12 };
#define TSG_PAGE_ROWS 64
#define TSG_PAGE_SHIFT 6
// Work list of a superstep (tsg_engine.items).  Only part of the searches emit a work item in a superstep (the
// others yielded, wait for a page, or are idle), so the items form a compacted list that the distance,
// recurrence and scoring kernels walk instead of all slots.  The slots are cut into regions of
// TSG_REGION_SLOTS; the searches of a region write into the region's own stretch of the list (one counter pair
// per region and superstep parity in tsg_engine.item_counts, one atomic per wave): items whose ancestor chain
// has at least TSG_LONG_DEPTH rows from the front, the others from the back, so the scoring kernel can start
// the long chains first.  Position p of a region of rs slots is filled iff p < n_long or p >= rs - n_short.
#define TSG_REGION_SLOTS 4096
#ifndef TSG_LONG_DEPTH
#define TSG_LONG_DEPTH 16
#endif
#define TSG_ITEM_CNT_STRIDE 32  // one 128-byte line per region: word 0 = n_long, word 1 = n_short
__host__ __device__ inline int tsg_n_regions(int slots) { return (slots + TSG_REGION_SLOTS - 1) / TSG_REGION_SLOTS; }
__host__ __device__ inline size_t tsg_item_positions(int slots) { return (size_t)tsg_n_regions(slots) * TSG_REGION_SLOTS; }
__host__ __device__ inline size_t tsg_item_count_words(int slots) {  // two planes
  return (size_t)tsg_n_regions(slots) * 2 * TSG_ITEM_CNT_STRIDE;
}
__host__ __device__ inline int tsg_region_size(int slots, int region) {
  const int left = slots - region * TSG_REGION_SLOTS;
  return left < TSG_REGION_SLOTS ? left : TSG_REGION_SLOTS;
}
__host__ __device__ inline size_t tsg_item_cnt_index(int slots, int plane, int region) {
  return ((size_t)plane * tsg_n_regions(slots) + region) * TSG_ITEM_CNT_STRIDE;
}
// Reader side: the counters of the plane the control step published.  Both planes are loaded together with the
// plane word (and the caller loads the item words before it knows whether the position is filled: they are
// then stale, in range, and not used), so finding one's work item costs one round trip.
__device__ __forceinline__ void tsg_item_counts(const int32_t* counts, const int32_t* pc, int slots, int region,
                                                int& n_long, int& n_short) {
  const int32_t* c0 = counts + tsg_item_cnt_index(slots, 0, region);
  const int32_t* c1 = counts + tsg_item_cnt_index(slots, 1, region);
  const int plane = pc[PC_ITEM_PLANE];
  const int l0 = c0[0], s0 = c0[1], l1 = c1[0], s1 = c1[1];
  n_long = plane ? l1 : l0;
  n_short = plane ? s1 : s0;
}
__host__ __device__ inline bool tsg_item_filled(int p, int rs, int n_long, int n_short) {
  return p < n_long || (p >= rs - n_short && p < rs);
}
// node record words ([B][node_cap][8] int32): BEG / C = CSR row of the node's road (indptr[rid], out-degree), DEPTH =
// length of the trajectory that ends in the node (its current parent's depth + 1)
enum { ND_RID = 0, ND_PARENT, ND_LP, ND_DATE_TIME, ND_EXP, ND_BEG, ND_C, ND_DEPTH, ND_WORDS = 8 };
enum { ND_OPEN = -1, ND_CLOSED_NOSLOT = -2 };
// expanded-slot meta words ([B][exp_cap][4] int32)
enum { EM_PARENT = 0, EM_RID, EM_TCODE, EM_DEPTH, EM_WORDS = 4 };

// Queue stretch of a slot (tsg_engine.heap, heap_cap words of 8 bytes): a blocked heap for the first twelve levels
// (entries 1..4095: three levels of the binary heap per 64-byte block, blocks numbered like an 8-ary heap, the
// stretch starting one block late; heap_addr in tsg_search.hip), the entries beyond in plain heap order behind the
// blocks.  tsg_heap_room: the number of entries a stretch of `words` words can hold (one spare block behind the
// last one, so that a sift may read the aligned pair of blocks of an only child).
#define TSG_HEAP_BLOCKED 4095                  /* entries in blocks: levels 0..11 */
#define TSG_HEAP_LIN_BASE (8 * (585 + 2) + 8)  /* word of entry 4096: behind 585 blocks, the offset and the spare one */
__host__ __device__ inline int tsg_heap_room(int words) {
  if (words > TSG_HEAP_LIN_BASE) return TSG_HEAP_BLOCKED + (words - TSG_HEAP_LIN_BASE);
  const long blocks = (long)words / 8 - 2;
  long n = 0, pre = 0, p8 = 1;
  while (p8 <= 512) {
    if (blocks >= pre + p8) {  // every block of this block level fits: entries through three more levels
      n = 8 * p8 - 1;
      pre += p8;
      p8 *= 8;
    } else {
      const long avail = blocks - pre;
      if (avail > 0) n = p8 + avail - 1;
      break;
    }
  }
  return (int)n;
}

#define TSG_GEN_SHIFT 20
#define TSG_NODE_MASK ((1u << TSG_GEN_SHIFT) - 1u)
#define TSG_GEN_MAX 4095
