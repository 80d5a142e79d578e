// Device side of the batched A* engine that more than one translation unit needs: the engine descriptor as the
// kernels take it, the page-pool control step, and the candidate distances of a superstep.  The candidate
// distances are launched either as a kernel of their own (cand_dis_kernel, tsg_search.hip) or as extra
// workgroups at the end of the layer-0 recurrence launch (tsg_dense.hip), whose second round of workgroups leaves
// half of the CUs' places free.
#pragma once
#include "tsg_device.h"

struct EngineDev {
  tsg_engine e;
  tsg_graph g;
  tsg_od_batch od;
};

// One LSTM layer step (tsg_dense.hip).  Rows come either from explicit index arrays (prev_idx / out_idx / rid /
// tcode, checked against state_rows when it is > 0) or, with `eng`, from the engine's work list; `tail`: the
// candidate-distance workgroups of the superstep ride behind the recurrence's own (layer 0, split form).
int tsg_launch_lstm_layer(const tsg_model* m, int layer, int max_rows, const int32_t* n_rows_dev, int n_rows_host,
                          const int64_t* prev_idx, const int64_t* out_idx, const int32_t* rid, const int32_t* tcode,
                          const tsg_engine* eng, const float* h_below, float* h, float* c, hipStream_t stream, int split,
                          const EngineDev* tail, long long state_rows = 0);

// Candidate distances of the superstep: one LANE per (work item, successor) pair instead of one wave per
// item, so the fp64 Vincenty iteration runs on full wavefronts.  Writes kilometres * 10 as f32
// (search.py:355-356) into cand_lp[slot][c], where the scoring kernel picks it up and overwrites it with
// the candidate's log-probability.
// Once per superstep, by one thread, after the frontier kernel: close the books of the state-page pool.
//   * ring heads may have overshot their snapshot (a failed take still increments): clamp them;
//   * what the launch returned becomes available to the next launch (tail snapshots);
//   * new searches are admitted only out of the surplus above admit_reserve (quota = surplus / 2 searches per
//     superstep: a new search takes a page at once and more soon), so the running searches keep first call
//     on the pool; an empty engine admits freely;
//   * if some search waited for a page and no search made progress, the pool is deadlocked: raise the restart
//     level (searches holding at most that many pages may give them back and queue again), doubling until
//     something moves, and hand out a budget of pages to free, so that of several equally young searches
//     only as many restart as it takes; any progress resets the level.
__device__ __forceinline__ void engine_control_step(const EngineDev& D, bool progressed, bool waited) {
  int32_t* pc = D.e.counters;
  const int ss = pc[PC_SUPERSTEP];
  const int tsnap = pc[PC_TAIL_SNAP];
  int head = pc[PC_HEAD];
  if (head > tsnap) head = tsnap;
  pc[PC_HEAD] = head;
  const int tail = pc[PC_TAIL];
  pc[PC_TAIL_SNAP] = tail;
  const int rsnap = pc[PC_RTAIL_SNAP];
  int rhead = pc[PC_RHEAD];
  if (rhead > rsnap) rhead = rsnap;
  pc[PC_RHEAD] = rhead;
  pc[PC_RTAIL_SNAP] = pc[PC_RTAIL];
  const int free_pages = tail - head;
  if (free_pages < pc[PC_MIN_FREE]) pc[PC_MIN_FREE] = free_pages;
  int quota = free_pages >= D.e.n_pages ? D.e.slots : (free_pages - D.e.admit_reserve) / 2;
  pc[PC_ADMIT_QUOTA] = quota > 0 ? quota : 0;
  int level = pc[PC_ABORT_LEVEL];
  int budget = 0;
  if (waited && !progressed) {
    level = level > 0 ? (2 * level < D.e.pages_per_slot ? 2 * level : D.e.pages_per_slot) : 1;
    budget = D.e.admit_reserve / 2 > 1 ? D.e.admit_reserve / 2 : 1;
  } else if (progressed) {
    level = 0;
  }
  pc[PC_ABORT_LEVEL] = level;
  pc[PC_ABORT_BUDGET] = budget;
  pc[PC_SUPERSTEP] = ss + 1;
}

// "Did any search make progress / wait for a page in this superstep?"  Thousands of searches answer per launch;
// one flag word would be thousands of stores to one address (which the memory system serialises), so search b
// stamps word (b % 256) * 32 of its flag plane (256 words, one per 128-byte line) and one workgroup gathers them.
#define TSG_FLAG_PLANE (256 * 32)
__device__ __forceinline__ void engine_control_block(const EngineDev& D) {  // all 256 threads of one workgroup
  const int stamp = D.e.counters[PC_SUPERSTEP] + 1;
  const int p = D.e.page_flags[threadIdx.x * 32] == stamp;
  const int w = D.e.page_flags[TSG_FLAG_PLANE + threadIdx.x * 32] == stamp;
  const int any_p = __syncthreads_or(p), any_w = __syncthreads_or(w);
  // work list: the frontier kernel filled (and published, PC_ITEM_PLANE) the plane of counters that the other
  // kernels of this superstep read; the other plane is cleared here for the next frontier launch
  const int ss = D.e.counters[PC_SUPERSTEP];
  const int nreg = tsg_n_regions(D.e.slots);
  for (int r = threadIdx.x; r < nreg; r += blockDim.x) {
    int32_t* c = D.e.item_counts + tsg_item_cnt_index(D.e.slots, (ss + 1) & 1, r);
    c[0] = 0;
    c[1] = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) engine_control_step(D, any_p != 0, any_w != 0);
}

// Workgroup `block` of the candidate-distance work of a superstep (256 threads): workgroup 0 closes the books of
// the superstep (engine_control_block: the longest dependent chain, so it comes first and does nothing else),
// the others take one (work item, successor) pair per lane.
__device__ __forceinline__ void cand_dis_block(const EngineDev& D, int block) {
  const int md = D.g.max_degree;
  // the first workgroup only closes the books of the superstep (engine_control_block: the longest dependent chain
  // of this launch, so it starts first and does nothing else); the others take the candidates
  if (block == 0) {
    if (D.od.n_od > 0) engine_control_block(D);  // engine run (not the stand-alone scoring entry)
    return;
  }
  const long t = (long)(block - 1) * blockDim.x + threadIdx.x;
  const long q = t / md;  // position in the work list
  const int c = (int)(t % md);
  const int region = (int)(q / TSG_REGION_SLOTS), pos = (int)(q % TSG_REGION_SLOTS);
  if (region >= tsg_n_regions(D.e.slots)) return;
  const int32_t* it = D.e.items + (size_t)q * IT_WORDS;
  const int ncand = it[IT_NCAND];
  int n_long, n_short;
  tsg_item_counts(D.e.item_counts, D.e.counters, D.e.slots, region, n_long, n_short);
  if (!tsg_item_filled(pos, tsg_region_size(D.e.slots, region), n_long, n_short) || ncand < 2 || c >= ncand) return;
  const int cand = D.e.cand_id[(size_t)it[IT_SLOT] * md + c];  // the successor list the frontier kernel left
  float dis;
  if (D.g.dist_table) {  // region graphs: region_dist[c][des] / 100 (search.py:1049)
    dis = (float)(D.g.dist_table[(size_t)cand * D.g.n_roads + it[IT_DES]] / 100.0);
  } else {
    const double dm = tsg_dev_geodesic_m(D.g.geo + 4 * (size_t)it[IT_DES], D.g.geo + 4 * (size_t)cand);
    dis = (float)(dm / 1000.0 * 10.0);
  }
  D.e.cand_lp[(size_t)it[IT_SLOT] * md + c] = dis;
}

// number of workgroups cand_dis_block wants
static inline int cand_dis_blocks(const tsg_engine& e, const tsg_graph& g) {
  const long threads = (long)tsg_item_positions(e.slots) * g.max_degree;
  return (int)tsg_cdiv(threads, 256) + 1;
}
