// Batched A* engine kernels: candidate scoring (one wave64 per expanded node) and the per-OD
// frontier kernel (heapq-exact priority queue, closed set, relax rules of search.py:385-419).
#include <new>

#include "tsg_engine_dev.h"
#include "tsg_heap.h"


// =====================================================================================
// explicit-row scoring: GeneratorV4.predict (generator_v4.py:56-72) for (prefix, candidates, dis) rows
// =====================================================================================
template <int JT>
__global__ __launch_bounds__(256) void score_batch_kernel(tsg_model m, tsg_score_batch sb) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= sb.n_rows) return;
  const int hp = m.hp, J = hp / 64;
  const int c0 = sb.chain_ptr[row], c1 = sb.chain_ptr[row + 1];
  const int k0 = sb.cand_ptr[row], ncand = sb.cand_ptr[row + 1] - k0;
  if (c1 <= c0 || ncand <= 0) return;
  float q[JT];
  load_row<JT>(sb.h_top + (size_t)sb.chain[c1 - 1] * hp, q, J, lane);
  AttnAcc<JT> acc;
  attn_init(acc);
  for (int i = c1 - 1; i >= c0; i -= 4) {
    const float* rows[4];
    unsigned present = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      rows[u] = (i - u >= c0) ? sb.h_top + (size_t)sb.chain[i - u] * hp : nullptr;
      if (rows[u]) present |= 1u << u;
    }
    float key[4][JT];
    attn_load4<JT>(key, rows, J, lane);
    attn_fold4<JT>(acc, key, present, q);
  }
  float pooled[JT];
  const float inv = 1.0f / acc.l;
#pragma unroll
  for (int j = 0; j < JT; ++j) pooled[j] = acc.v[j] * inv;
  const int my_cand = lane < ncand ? sb.cand[k0 + lane] : 0;
  const float my_dis = lane < ncand ? sb.dis[k0 + lane] : 0.0f;
  float pg, ph, lg, lh;
  const float f = score_row<JT>(m, pooled, J, lane, ncand, my_cand, my_dis, sb.des[row], &pg, &ph, &lg, &lh);
  if (lane < ncand) {
    sb.prob[k0 + lane] = f;
    if (sb.prob_g) sb.prob_g[k0 + lane] = pg;
    if (sb.prob_h) sb.prob_h[k0 + lane] = ph;
    if (sb.logit_g) sb.logit_g[k0 + lane] = lg;
    if (sb.logit_h) sb.logit_h[k0 + lane] = lh;
  }
}

static inline int jt_for(int hp) {
  const int J = hp / 64;
  return J <= 1 ? 1 : J <= 2 ? 2 : J <= 4 ? 4 : 8;
}

extern "C" int tsg_score_candidates(const tsg_model* m, const tsg_score_batch* b, void* stream) {
  if (!m || !b) return TSG_E_INVALID;
  if (m->hp <= 0 || (m->hp % 64) != 0 || m->hp > 512) return TSG_E_INVALID;
  if (b->n_rows < 0 || !b->chain_ptr || !b->chain || !b->h_top || !b->cand_ptr || !b->cand || !b->dis || !b->des ||
      !b->prob)
    return TSG_E_INVALID;
  if (!m->out_w || !m->out_b || !m->pfrom || !m->pto || !m->hparams) return TSG_E_INVALID;
  if (b->n_rows == 0) return 0;
  const dim3 grid(tsg_cdiv(b->n_rows, 4)), blk(256);
  hipStream_t s = (hipStream_t)stream;
  switch (jt_for(m->hp)) {
    case 1: hipLaunchKernelGGL(score_batch_kernel<1>, grid, blk, 0, s, *m, *b); break;
    case 2: hipLaunchKernelGGL(score_batch_kernel<2>, grid, blk, 0, s, *m, *b); break;
    case 4: hipLaunchKernelGGL(score_batch_kernel<4>, grid, blk, 0, s, *m, *b); break;
    default: hipLaunchKernelGGL(score_batch_kernel<8>, grid, blk, 0, s, *m, *b); break;
  }
  TSG_LAUNCH_CHECK();
  return 0;
}

// Function H on explicit (from, to, distance) triples: the score before the softmax (distance_gat_fc.py:118-134
// predict_cache; :44-74 forward with the module in inference mode, where the node embeddings are the cached ones).
__global__ void h_pair_scores_kernel(tsg_model m, int n, const int32_t* from, const int32_t* to, const float* dis,
                                     float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = tsg_h_logit(m, from[i], to[i], dis[i]);
}

extern "C" int tsg_h_pair_scores(const tsg_model* m, int32_t n, const int32_t* from, const int32_t* to,
                                 const float* dis, float* out, void* stream) {
  if (!m || n < 0 || !m->pfrom || !m->pto || !m->hparams) return TSG_E_INVALID;
  if (n == 0) return 0;
  if (!from || !to || !dis || !out) return TSG_E_INVALID;
  hipLaunchKernelGGL(h_pair_scores_kernel, dim3(tsg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, *m, n, from, to,
                     dis, out);
  TSG_LAUNCH_CHECK();
  return 0;
}

// The candidate distance on explicit pairs (search.py:355-356): the same device function the engine's
// cand_dis_block calls, exposed so that it can be checked on its own.
__global__ void geodesic_pairs_kernel(const double* __restrict__ geo, int n_rows, int n, const int32_t* __restrict__ a,
                                      const int32_t* __restrict__ b, double* __restrict__ out_m) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int ia = a[i], ib = b[i];
  if (ia < 0 || ia >= n_rows || ib < 0 || ib >= n_rows) {
    out_m[i] = -1.0;  // index outside the table
    return;
  }
  out_m[i] = tsg_dev_geodesic_m(geo + 4 * (size_t)ia, geo + 4 * (size_t)ib);
}

extern "C" int tsg_geodesic_pairs(const double* geo, int32_t n_rows, int32_t n, const int32_t* a, const int32_t* b,
                                  double* out_m, void* stream) {
  if (!geo || n_rows <= 0 || n < 0) return TSG_E_INVALID;
  if (n == 0) return 0;
  if (!a || !b || !out_m) return TSG_E_INVALID;
  hipLaunchKernelGGL(geodesic_pairs_kernel, dim3(tsg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, geo, n_rows, n, a,
                     b, out_m);
  TSG_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================
// relaxation of a scored expansion (search.py:385-419 / 317-350), one search per lane
// =====================================================================================
// The words of the expansion whose successors are relaxed and of the search it belongs to.  The relaxation runs
// at the start of the frontier kernel's next superstep.  TSG_RELAX_IN_SCORE = 1 (measured, not shipped: DESIGN.md
// section 8.2) moves it to where the log-probs are, lane 0 of the scoring wave that has just produced them; the
// frontier kernel then keeps this code for the expansions the scoring wave leaves alone (a workspace cap in reach:
// the error path lives there).  Results are identical either way (same operations in the same order per search);
// the frontier launch gets 15 us shorter and the scoring launch 14-18 us longer, because a scoring wave's life is
// what bounds that launch (four rounds of waves) and the chain of pushes is added to every life.
#ifndef TSG_RELAX_IN_SCORE
#define TSG_RELAX_IN_SCORE 0
#endif
struct RelaxCtx {
  int b, e, C, rid, date_time, cost, child_depth, gen, woff, deflen;
  float best_lp;
  int heap_n, node_n, err;
  int heap_room, node_cap, n_roads;
  uint64_t* heap;
  uint64_t* table;
  int32_t* nodes;
  int32_t* S;
  const int32_t* pair_cost;
};
// successors i0 .. i0+3 of the list (positions >= C repeat the last one and are skipped): ids, scored log-probs,
// their table entries and CSR rows as fetched by the caller; the pushes stay in list order (search.py:385)
__device__ __forceinline__ void relax_batch(RelaxCtx& r, int i0, const int cc[4], const float lps[4], uint64_t tt[4],
                                            const int cb[4], const int ce[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i0 + k >= r.C || r.err) continue;
    const int c = cc[k];
    const float lp = lps[k];
    bool dup = false;  // the same road twice in one list: re-read what the earlier entry may have changed
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < k && cc[j] == c) dup = true;
    if (dup) tt[k] = r.table[c];
    // travel time of the expanded road (search.py:376-382); region graphs fall back to a per-pair
    // cost when the region's time distribution is unknown (search.py:1073-1078)
    const int cost = r.cost >= 0 ? r.cost : r.pair_cost[(size_t)r.rid * r.n_roads + c];
    const int next_time = (r.date_time + cost) % 86400;
    const int tcode_child = next_time / 60 + r.woff;
    const uint32_t t = (uint32_t)tt[k];
    const float tlp = __uint_as_float((uint32_t)(tt[k] >> 32));  // open: the node's log_prob; closed: +inf
    bool pushed = false;
    int idx = -1;
    if ((int)(t >> TSG_GEN_SHIFT) != r.gen) {  // c not in rid2node and not in close_set
      if (r.node_n >= r.node_cap || r.heap_n >= r.heap_room) {
        r.err = 1;
        continue;
      }
      idx = r.node_n++;
      int4* nd = reinterpret_cast<int4*>(r.nodes + (size_t)idx * ND_WORDS);
      nd[0] = make_int4(c, r.e, __float_as_int(lp), next_time);
      nd[1] = make_int4(ND_OPEN, cb[k], ce[k] - cb[k], r.child_depth);
      pushed = true;
    } else {
      idx = (int)(t & TSG_NODE_MASK);
      if (tlp < lp) {  // c in rid2node, strictly better
        if (r.heap_n >= r.heap_room) {
          r.err = 1;
          continue;
        }
        int32_t* nd = r.nodes + (size_t)idx * ND_WORDS;
        nd[ND_PARENT] = r.e;
        nd[ND_LP] = __float_as_int(lp);
        nd[ND_DATE_TIME] = next_time;
        nd[ND_DEPTH] = r.child_depth;
        pushed = true;
      }
    }
    if (pushed) {
      r.table[c] = (uint64_t)(((uint32_t)r.gen << TSG_GEN_SHIFT) | (uint32_t)idx) | ((uint64_t)__float_as_uint(lp) << 32);
      HeapEnt he;
      he.prio = -lp;
      he.node = idx;
      heap_push(r.heap, r.heap_n, he, r.nodes);
      // default-length fallback snapshot (search.py:397-402): np.exp(lp) > best_score, best_score starts at 0
      if (r.child_depth == r.deflen && lp > r.best_lp && lp > -745.13f) {
        r.best_lp = lp;
        r.S[SS_BEST_LP] = __float_as_int(lp);
        r.S[SS_BEST_PARENT] = r.e;
        r.S[SS_BEST_RID] = c;
        r.S[SS_BEST_TCODE] = tcode_child;
        r.S[SS_HAVE_BEST] = 1;
      }
    }
  }
}

// The relaxation as the tail of a scoring wave: lane 0 runs the chain.  What it needs was requested while the key
// rows travelled and waits in three registers: `s_word`, lane i < SS_WORDS = word i of the slot state as the
// frontier kernel left it; (`pa`, `pb`), lanes 0-3 = the table entries of the first four successors, lanes 4-7 =
// their CSR rows (indptr[c], indptr[c + 1]).  Lane k < C holds successor k's id and scored log-prob.  Nothing is
// done (the frontier kernel relaxes as it always did) unless the slot waits for exactly this expansion and every
// push is certain to find room.
__device__ __forceinline__ void relax_in_wave(const EngineDev& D, int b, int C, int lane, int my_cand, float my_lp,
                                              int s_word, unsigned pa, unsigned pb) {
  const int heap_room = tsg_heap_room(D.e.heap_cap);
  const int phase = __shfl(s_word, SS_PHASE, TSG_WAVE), heap_n = __shfl(s_word, SS_HEAP_N, TSG_WAVE);
  const int node_n = __shfl(s_word, SS_NODE_N, TSG_WAVE), cur_c = __shfl(s_word, SS_CUR_C, TSG_WAVE);
  if (phase != PH_EXPANDED || cur_c != C || C < 1) return;
  if (node_n + C > D.e.node_cap || heap_n + C > heap_room) return;
  RelaxCtx r;
  r.b = b;
  r.e = __shfl(s_word, SS_CUR_EXP, TSG_WAVE);
  r.C = C;
  r.rid = __shfl(s_word, SS_CUR_RID, TSG_WAVE);
  r.date_time = __shfl(s_word, SS_CUR_DATE_TIME, TSG_WAVE);
  r.cost = __shfl(s_word, SS_CUR_COST, TSG_WAVE);
  r.child_depth = __shfl(s_word, SS_CUR_DEPTH, TSG_WAVE) + 1;
  r.gen = __shfl(s_word, SS_GEN, TSG_WAVE);
  r.woff = __shfl(s_word, SS_WOFF, TSG_WAVE);
  r.deflen = __shfl(s_word, SS_DEFLEN, TSG_WAVE);
  r.best_lp = __int_as_float(__shfl(s_word, SS_BEST_LP, TSG_WAVE));
  r.heap_n = heap_n;
  r.node_n = node_n;
  r.err = 0;
  r.heap_room = heap_room;
  r.node_cap = D.e.node_cap;
  r.n_roads = D.g.n_roads;
  r.heap = D.e.heap + (size_t)b * D.e.heap_cap;
  r.table = D.e.table + (size_t)b * D.g.n_roads;
  r.nodes = D.e.nodes + (size_t)b * D.e.node_cap * ND_WORDS;
  r.S = D.e.slot_state + (size_t)b * SS_WORDS;
  r.pair_cost = D.g.pair_cost;
  const float lp_parent = __int_as_float(__shfl(s_word, SS_CUR_LP, TSG_WAVE));
  const int step = __shfl(s_word, SS_STEP, TSG_WAVE);
  for (int i0 = 0; i0 < C; i0 += 4) {
    int cc[4], cb[4], ce[4];
    float lps[4];
    uint64_t tt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // all lanes: lane 0 receives what the other lanes hold
      const int src = i0 + k < C ? i0 + k : C - 1;
      cc[k] = __shfl(my_cand, src, TSG_WAVE);
      lps[k] = C == 1 ? lp_parent : __shfl(my_lp, src, TSG_WAVE);
      const int kk = k < C ? k : C - 1;
      tt[k] = (uint64_t)(unsigned)__shfl((int)pa, kk, TSG_WAVE) | ((uint64_t)(unsigned)__shfl((int)pb, kk, TSG_WAVE) << 32);
      cb[k] = __shfl((int)pa, 4 + kk, TSG_WAVE);
      ce[k] = __shfl((int)pb, 4 + kk, TSG_WAVE);
    }
    if (lane == 0) {
      if (i0 > 0) {  // beyond the first four (rare): fetched now, after the pushes that may have changed them
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          tt[k] = r.table[cc[k]];
          cb[k] = D.g.indptr[cc[k]];
          ce[k] = D.g.indptr[cc[k] + 1];
        }
      }
      relax_batch(r, i0, cc, lps, tt, cb, ce);
    }
  }
  if (lane == 0) *reinterpret_cast<int4*>(r.S) = make_int4(PH_SEARCH, step, r.heap_n, r.node_n);
}

// Waves per workgroup of the two kernels whose waves run for very different times (a long ancestor chain, a long
// pop chain): a workgroup gives its registers back only when its last wave ends, so one wave per workgroup
// lets the next work item start as soon as ANY wave ends (scoring kernel: 130 -> 115 us per launch).
#ifndef TSG_SCORE_WPB
#define TSG_SCORE_WPB 1
#endif
#ifndef TSG_SCORE_GROUP
#define TSG_SCORE_GROUP 4  // key rows per group of the scoring kernel's attention loop; two groups are in flight (groups of 8: 143 us per launch against 117)
#endif

// =====================================================================================
// engine scoring: one wave per work item of the current superstep
// =====================================================================================
__global__ __launch_bounds__(256) void engine_control_kernel(EngineDev D) { engine_control_block(D); }

__global__ void engine_init_pages_kernel(tsg_engine e) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < e.n_pages) e.page_ring[i] = i;
  if (i == 0) {
    int32_t* pc = e.counters;
    for (int k = 0; k < PC_WORDS; ++k) pc[k] = 0;
    pc[PC_TAIL] = e.n_pages;
    pc[PC_TAIL_SNAP] = e.n_pages;
    pc[PC_MIN_FREE] = e.n_pages;
    pc[PC_ADMIT_QUOTA] = e.slots;
  }
}

__global__ __launch_bounds__(256) void cand_dis_kernel(EngineDev D, int superstep) { cand_dis_block(D, blockIdx.x); }

// Every expanded slot e keeps the list of its 64 nearest ancestors: anc[e][i] = the (i+1)-th ancestor
// (or -1).  A wave builds the list of its own slot from its parent's list with ONE coalesced 256-byte read
// (anc[e][0] = parent, anc[e][i] = anc[parent][i-1]) and then already holds, one per lane, the slots whose
// top-layer hidden rows the attention needs: no pointer chase, the key rows are independent coalesced
// loads.  Chains longer than 64 hop through anc[.][63] once per 64 rows.
template <int JT, bool SPLIT, bool VEC4>
__global__ __launch_bounds__(256) void score_frontier_kernel(tsg_model m, EngineDev D, int superstep) {
  // wave w of the launch -> position w / R of region w % R of the work list: the fronts of all regions (long
  // ancestor chains) start first, the empty middle exits at once, the short chains fill in behind
  const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int nreg = tsg_n_regions(D.e.slots);
  const int region = w % nreg, pos = w / nreg;
  const int rs = tsg_region_size(D.e.slots, region);
  if (pos >= rs) return;
  // Round trip 1: the counters of the region and the item words together (the words of a position that is not
  // filled in this superstep are stale but in range).  The empty asm statements make the compiler issue the loads
  // HERE: left alone it sinks every load below the early returns, one dependent round trip after the other
  // (counters, item, ancestor list, candidates, function H's rows: five before the first key row was requested).
  const int32_t* it = D.e.items + ((size_t)region * TSG_REGION_SLOTS + pos) * IT_WORDS;
  int4 w0 = reinterpret_cast<const int4*>(it)[0];  // IT_SLOT, IT_ROW, IT_PROW, IT_RID
  int4 w1 = reinterpret_cast<const int4*>(it)[1];  // IT_TCODE, IT_NCAND, IT_DES, IT_LP
  int depth = it[IT_DEPTH];                        // = number of rows on the ancestor chain incl. e
  int n_long, n_short;
  tsg_item_counts(D.e.item_counts, D.e.counters, D.e.slots, region, n_long, n_short);
  float hpar[42];  // function H's small layers and w1: the same for every wave, read through the scalar cache (before
                   // this wave's first store: behind it the compiler would fall back to 42 vector loads per lane)
#pragma unroll
  for (int k = 0; k < 42; ++k) hpar[k] = m.hparams[k];
  asm volatile("" : "+v"(w0.x), "+v"(w1.x), "+v"(depth), "+v"(n_long), "+v"(n_short));
  if (!tsg_item_filled(pos, rs, n_long, n_short)) return;  // not filled in this superstep
  // the item is the same for all lanes: its words go to scalar registers (and what they address with the same
  // offset in all lanes, function H's destination row, comes through the scalar cache)
  const int b = __builtin_amdgcn_readfirstlane(w0.x), e = __builtin_amdgcn_readfirstlane(w0.y);  // e: the state row of the expanded node
  const int parent = __builtin_amdgcn_readfirstlane(w0.z);
  const int ncand = __builtin_amdgcn_readfirstlane(w1.y), des = __builtin_amdgcn_readfirstlane(w1.z);
  const float lp_parent = __int_as_float(__builtin_amdgcn_readfirstlane(w1.w));
  depth = __builtin_amdgcn_readfirstlane(depth);
  constexpr int LV = TSG_UP_LEVELS;  // = 64 ancestors per row
  int32_t* anc = D.e.exp_up;
  const int md = D.g.max_degree;
  const int hp = m.hp, J = hp / 64;
  const float* htop = D.e.h[m.n_layers - 1];
  const bool scored = ncand >= 2;  // warm-up rows and single-successor expansions need no model call (search.py:302)
  const bool on = lane < ncand && scored;
  // Round trip 2, everything the item words address: the parent's ancestor list, and for a scored item the query
  // row, the successor ids and distances (left by the frontier and distance kernels) and function H's
  // destination terms.
  // lane i: the state row of the (i+1)-th ancestor; kept in a register (this wave must not read back its own store)
  int mine = lane == 0 ? parent : (parent >= 0 ? anc[(size_t)parent * LV + lane - 1] : -1);
  float q[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) q[j] = 0.0f;
  int my_cand = 0;
  float my_dis = 0.0f;
  float pt[TSG_H_HIDDEN];
#pragma unroll
  for (int j = 0; j < TSG_H_HIDDEN; ++j) pt[j] = 0.0f;
  if (scored) {
    {  // raw words: decoded below, after the next loads are out
      const float* qrow[1] = {htop + (size_t)e * hp};
      float q1[1][JT];
      attn_load_rows<JT, 1, VEC4>(q1, qrow, J, lane);
#pragma unroll
      for (int j = 0; j < JT; ++j) q[j] = q1[0][j];
    }
    if (on) {
      my_cand = D.e.cand_id[(size_t)b * md + lane];
      my_dis = D.e.cand_lp[(size_t)b * md + lane];  // written by the distance workgroups of this superstep
    }
    const float* ptp = m.pto + (size_t)des * TSG_H_HIDDEN;
#pragma unroll
    for (int j = 0; j < TSG_H_HIDDEN; ++j) pt[j] = ptp[j];
  }
#if TSG_RELAX_IN_SCORE
  if (ncand == 1 && lane == 0) my_cand = D.e.cand_id[(size_t)b * md];
  const int s_word = D.e.slot_state[(size_t)b * SS_WORDS + (lane < SS_WORDS ? lane : 0)];  // for the relaxation at the end
#endif
  asm volatile("" : "+v"(mine), "+v"(my_cand));
  anc[(size_t)e * LV + lane] = mine;
#if TSG_RELAX_IN_SCORE
  // the first successors' table entries (lanes 0-3) and CSR rows (lanes 4-7), for the relaxation: requested here,
  // used at the end
  unsigned pa = 0, pb = 0;
  {
    const int k = lane & 3;
    const int c = __shfl(my_cand, k < ncand ? k : 0, TSG_WAVE);
    if (lane < 8 && k < ncand) {
      const unsigned* src = lane < 4 ? reinterpret_cast<const unsigned*>(D.e.table + (size_t)b * D.g.n_roads + c)
                                     : reinterpret_cast<const unsigned*>(D.g.indptr + c);
      pa = src[0];
      pb = src[1];
    }
  }
  if (!scored) {
    if (ncand == 1) relax_in_wave(D, b, 1, lane, my_cand, 0.0f, s_word, pa, pb);
    return;
  }
#else
  if (!scored) return;
#endif
  // Round trip 3: function H's candidate rows and the out_fc bias (they need the successor id) go out in front of
  // the first group of key rows; H's score is finished before the attention loop is entered and costs one
  // register from there on.
  float pf[TSG_H_HIDDEN];
  float out_bias = 0.0f;
  {
    const float* pfp = m.pfrom + (size_t)my_cand * TSG_H_HIDDEN;
#pragma unroll
    for (int j = 0; j < TSG_H_HIDDEN; ++j) pf[j] = pfp[j];
    out_bias = m.out_b[my_cand];
  }
  if (SPLIT) {
#pragma unroll
    for (int j = 0; j < JT; ++j) q[j] = tsg_state_value(q[j]);
  }
  AttnAcc<JT> acc;
  attn_init(acc);
  int anchor = e;   // the (64 * chunk)-th ancestor
  int list = mine;  // lane i: the (i+1)-th ancestor of `anchor`
  constexpr int G = TSG_SCORE_GROUP;
  float keyA[G][JT], keyB[G][JT];
  bool w_in_a = false;
  // the out_fc rows of candidates 0 .. G-1 (function_g_fc.py:184-186: only the rows that are needed), as one more group
  auto issue_w = [&](float key[G][JT]) {
    const float* rows[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int cid = __shfl(my_cand, u < ncand ? u : ncand - 1, TSG_WAVE);
      rows[u] = m.out_w + (size_t)cid * hp;
    }
    attn_load_rows<JT, G, VEC4>(key, rows, J, lane);
  };
  // groups of TSG_SCORE_GROUP rows, software-pipelined: group g+1 is loading while group g is folded
  int n = depth < 64 ? depth : 64;  // rows of the current chunk of 64 ancestors
  int cur = 0;                      // lane t: the ancestor at distance t of `anchor` (lane 0 = the anchor itself)
  unsigned presA = 0, presB = 0;
  auto issue = [&](int i, float key[G][JT], unsigned& pres) {
    const float* rows[G];
    pres = 0;
#pragma unroll
    for (int u = 0; u < G; ++u) {
      // a position beyond the chain reads the chain's last row again (a row that exists) and is masked in the fold
      const int src_lane = i + u < n ? i + u : n - 1;
      const int id = __shfl(cur, src_lane, TSG_WAVE);
      rows[u] = htop + (size_t)id * hp;
      if (i + u < n) pres |= 1u << u;
    }
    attn_load_rows<JT, G, VEC4>(key, rows, J, lane);
  };
  {
    const int up1 = __shfl_up(list, 1, TSG_WAVE);
    cur = lane == 0 ? anchor : up1;
  }
  issue(0, keyA, presA);
  float h_logit = -INFINITY;
  bool h_pending = true;
  for (int base = 0; base < depth; base += 64) {
    const bool last_chunk = base + 64 >= depth;
    if (base > 0) {
      n = depth - base < 64 ? depth - base : 64;
      const int up1 = __shfl_up(list, 1, TSG_WAVE);
      cur = lane == 0 ? anchor : up1;
      issue(0, keyA, presA);
    }
    for (int i = 0; i < n; i += 2 * G) {
      if (i + G < n) {
        issue(i + G, keyB, presB);
      } else if (last_chunk) {
        issue_w(keyB);
        w_in_a = false;
      }
      if (h_pending) {  // function H's score, once, while the first two groups of rows travel
        h_logit = on ? tsg_h_logit_regs(hpar, pf, pt, my_dis) : -INFINITY;
        h_pending = false;
      }
      attn_fold_n<JT, G, SPLIT>(acc, keyA, presA, q);
      if (i + G < n) {
        if (i + 2 * G < n) {
          issue(i + 2 * G, keyA, presA);
        } else if (last_chunk) {
          issue_w(keyA);
          w_in_a = true;
        }
        attn_fold_n<JT, G, SPLIT>(acc, keyB, presB, q);
      }
    }
    if (!last_chunk) {  // next chunk: hop 64 up, fetch that slot's list
      anchor = __shfl(list, 63, TSG_WAVE);
      list = anc[(size_t)anchor * LV + lane];
    }
  }
  float pooled[JT];
  const float inv = 1.0f / acc.l;
#pragma unroll
  for (int j = 0; j < JT; ++j) pooled[j] = acc.v[j] * inv;
  // G: the candidates' out_fc rows . pooled (+ bias); the first group is already in registers (moved with plain
  // register copies under a uniform branch: a select between the two arrays made the compiler keep both on the stack)
  float g_logit = -INFINITY;
  if (w_in_a) {
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
      for (int j = 0; j < JT; ++j) keyB[u][j] = keyA[u][j];
  }
  {
    float part[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      float p = 0.0f;
#pragma unroll
      for (int j = 0; j < JT; ++j) p += keyB[u][j] * pooled[j];
      part[u] = p;
    }
#pragma unroll
    for (int u = 0; u < G; ++u) part[u] = tsg_wave_sum_u(part[u]);
#pragma unroll
    for (int u = 0; u < G; ++u)
      if (lane == u && u < ncand) g_logit = part[u] + out_bias;
  }
  for (int c0 = G; c0 < ncand; c0 += 4) {  // more than G successors (rare on road graphs)
    float part[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u < ncand ? c0 + u : ncand - 1;
      const int cid = __shfl(my_cand, c, TSG_WAVE);
      float w[JT];
      load_row<JT>(m.out_w + (size_t)cid * hp, w, J, lane);
      part[u] = 0.0f;
#pragma unroll
      for (int j = 0; j < JT; ++j) part[u] += w[j] * pooled[j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) part[u] = tsg_wave_sum_u(part[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (lane == c0 + u && c0 + u < ncand) g_logit = part[u] + out_bias;
  }
  const float f = tsg_mix_softmax(m, g_logit, h_logit, on, nullptr, nullptr);
  const float my_lp = logf(f) + lp_parent;  // search.py:370-373
  if (on) D.e.cand_lp[(size_t)b * md + lane] = my_lp;
#if TSG_RELAX_IN_SCORE
  relax_in_wave(D, b, ncand, lane, my_cand, my_lp, s_word, pa, pb);
#endif
}

// With the split-f16 recurrence the candidate distances (and the control step) ride on the layer-0 launch as extra
// workgroups (tsg_dense.hip); the fp32 form launches them on their own.
static inline bool cand_dis_rides(const tsg_engine* e) { return e->state_format == 1; }

static void launch_cand_dis(const EngineDev& D, int superstep, hipStream_t s) {
  hipLaunchKernelGGL(cand_dis_kernel, dim3(cand_dis_blocks(D.e, D.g)), dim3(256), 0, s, D, superstep);
}

template <bool SPLIT>
static void launch_score_frontier_t(const tsg_model* m, const EngineDev& D, int superstep, hipStream_t s) {
  const dim3 grid(tsg_cdiv(tsg_n_regions(D.e.slots) * TSG_REGION_SLOTS, TSG_SCORE_WPB)), blk(64 * TSG_SCORE_WPB);
  if (m->hp == 256) {  // BJ dims: one 16-byte load per lane and row
    hipLaunchKernelGGL((score_frontier_kernel<4, SPLIT, true>), grid, blk, 0, s, *m, D, superstep);
    return;
  }
  switch (jt_for(m->hp)) {
    case 1: hipLaunchKernelGGL((score_frontier_kernel<1, SPLIT, false>), grid, blk, 0, s, *m, D, superstep); break;
    case 2: hipLaunchKernelGGL((score_frontier_kernel<2, SPLIT, false>), grid, blk, 0, s, *m, D, superstep); break;
    case 4: hipLaunchKernelGGL((score_frontier_kernel<4, SPLIT, false>), grid, blk, 0, s, *m, D, superstep); break;
    default: hipLaunchKernelGGL((score_frontier_kernel<8, SPLIT, false>), grid, blk, 0, s, *m, D, superstep); break;
  }
}

static void launch_score_frontier(const tsg_model* m, const EngineDev& D, int superstep, hipStream_t s) {
  if (D.e.state_format)
    launch_score_frontier_t<true>(m, D, superstep, s);
  else
    launch_score_frontier_t<false>(m, D, superstep, s);
}

// =====================================================================================
// frontier kernel: one thread per in-flight search
// =====================================================================================
// The work of a search in a superstep is a chain of dependent small accesses; what it costs is the number of
// round trips to memory on that chain.  The data structures are laid out so that a look-up is one round trip:
//   * the rid table carries the open node's log_prob next to the node index (+inf once the road is closed), so
//     relaxation and the futile-expansion check never read a node record;
//   * a node record carries the CSR row of its road and its depth, so a popped node needs no indptr / parent read;
//   * the slot keeps the successor list of the expansion it waits for (cand_id, also read by the distance and
//     scoring kernels) and the page of its next state row;
//   * _siftdown after a pop starts from entries the sift just read;
//   * a finished search writes its trajectory from the ancestor lists (exp_up) and row_meta, eight points per
//     two round trips, instead of walking the parent chain one point at a time.
// (One WAVE per search cuts the round trips further, to 10-14 per superstep, but 24 576 waves of ~1 900
// instructions are issue-bound: 90 us per launch against this kernel's time; ablate/frontier_wave_per_search.hip.txt.)
// A trajectory out of the expansion records of one search: (rid, tcode) is its last point (depth = its length),
// `parent` the expanded node before it; the chain of parents fills the positions in front (search.py:291-296:
// the popped node's trace; :421-424: the fallback trace).  Positions >= lmax are dropped.
__device__ __forceinline__ void write_path(const int32_t* em, int parent, int rid, int tcode, int depth, int32_t* orid,
                                           int32_t* otim, int lmax) {
  int i = depth - 1;
  if (i < lmax) {
    orid[i] = rid;
    otim[i] = tcode;
  }
  int cur = parent;
  --i;
  while (cur >= 0 && i >= 0) {
    if (i < lmax) {
      orid[i] = em[(size_t)cur * EM_WORDS + EM_RID];
      otim[i] = em[(size_t)cur * EM_WORDS + EM_TCODE];
    }
    cur = em[(size_t)cur * EM_WORDS + EM_PARENT];
    --i;
  }
}

__global__ void reconstruct_paths_kernel(tsg_engine e, int n, const int32_t* slot, const int32_t* node,
                                         const int32_t* last_rid, const int32_t* last_tcode, int lmax,
                                         int32_t* out_rid, int32_t* out_time, int32_t* out_len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = slot[i], nd = node[i];
  if (b < 0 || b >= e.slots || nd < 0 || nd >= e.exp_cap) {
    out_len[i] = 0;
    return;
  }
  const int32_t* em = e.exp_meta + (size_t)b * e.exp_cap * EM_WORDS;
  const int32_t* r = em + (size_t)nd * EM_WORDS;
  int parent = r[EM_PARENT], rid = r[EM_RID], tcode = r[EM_TCODE], depth = r[EM_DEPTH];
  if (last_rid && last_rid[i] >= 0) {  // one more point behind the expanded node (a successor that was not expanded)
    parent = nd;
    rid = last_rid[i];
    tcode = last_tcode ? last_tcode[i] : 0;
    depth += 1;
  }
  write_path(em, parent, rid, tcode, depth, out_rid + (size_t)i * lmax, out_time + (size_t)i * lmax, lmax);
  out_len[i] = depth;
}

// The same trajectory from the ancestor lists: exp_up[row] names the 64 nearest ancestors' state rows of the
// node in `row`, row_meta[row] = (rid, time code).  Eight points per two round trips (a list read, then the rows).
__device__ __forceinline__ void write_path_rows(const tsg_engine& e, int parent_row, int rid, int tcode, int depth,
                                                int32_t* orid, int32_t* otim, int lmax) {
  if (depth - 1 < lmax) {
    orid[depth - 1] = rid;
    otim[depth - 1] = tcode;
  }
  int anchor = parent_row;  // sits at position `top`
  int top = depth - 2;
  while (anchor >= 0 && top >= 0) {
    {
      const int2 rm = *reinterpret_cast<const int2*>(e.row_meta + 2 * (size_t)anchor);
      if (top < lmax) {
        orid[top] = rm.x;
        otim[top] = rm.y;
      }
    }
    const int32_t* up = e.exp_up + (size_t)anchor * TSG_UP_LEVELS;
    int next_anchor = -1;
    for (int i0 = 0; i0 < TSG_UP_LEVELS && top - 1 - i0 >= 0; i0 += 8) {
      int4 a[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) a[u] = *reinterpret_cast<const int4*>(up + i0 + 4 * u);
      const int rows[8] = {a[0].x, a[0].y, a[0].z, a[0].w, a[1].x, a[1].y, a[1].z, a[1].w};
      int2 rm[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rows[u];
        rm[u] = *reinterpret_cast<const int2*>(e.row_meta + 2 * (size_t)(r >= 0 ? r : 0));
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int pos = top - 1 - i0 - u;
        if (rows[u] >= 0 && pos >= 0 && pos < lmax) {
          orid[pos] = rm[u].x;
          otim[pos] = rm[u].y;
        }
      }
      if (i0 + 8 == TSG_UP_LEVELS) next_anchor = rows[7];
    }
    // the list's last entry is the anchor's 64th ancestor: it sits at position top - 64
    anchor = next_anchor;
    top -= TSG_UP_LEVELS;
  }
}

// TSG_SPW searches per wave (one lane each, the other lanes idle): the work is a chain of dependent small
// accesses, so what matters is how many chains are in flight and how long a search waits for the slowest
// search of its wave (every divergent loop runs to the longest trip count in the wave), not lane use.
#ifndef TSG_SPW
#define TSG_SPW 8
#endif
#ifndef TSG_FRONTIER_WPB
#define TSG_FRONTIER_WPB 4
#endif
static_assert(TSG_REGION_SLOTS % TSG_SPW == 0, "the searches of a wave share one region of the work list");
#if TSG_STAMP
__device__ unsigned long long g_stamp[32768 * 32];  // one record per search slot, of the superstep TSG_STAMP_AT
#define CKPT(i) { const long long _n = clock64(); st_acc[i] += _n - st_last; st_last = _n; }
#define STAMP_ADD(i, v) st_acc[i] += (long long)(v)
extern "C" int tsg_debug_read_stamps(unsigned long long* out_host, int reset) {
  (void)reset;
  if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamp), sizeof(g_stamp)) != hipSuccess) return -1;
  return 0;
}
#else
#define CKPT(i)
#define STAMP_ADD(i, v)
#endif
// The look-ups that follow a pop (search.py:281-300: the popped node's record, then its road's successors, label,
// travel cost and the page of its parent's state row, then the successors' table entries for the futile-expansion
// check), as the chain that rides along with the sift of heap_pop (tsg_heap.h): three dependent round trips that
// used to follow the pop's own four now share them.  Values only; every decision stays where it was.
#ifndef TSG_POP_STAGES
#define TSG_POP_STAGES 2  // look-up stages that ride inside the sift: 1 = node record + successors etc., 2 = + table entries
#endif
struct PopSide {
  const int32_t* nodes;
  const int32_t* indices;
  const int32_t* my_pages;
  const int32_t* label;
  const uint8_t* empty_key;
  const int32_t* cost_tab;
  const uint64_t* table;
  int n_roads, allow;
  int stage;  // 0 idle, 1 node record requested, 2 successors etc. requested, 3 table entries requested / nothing more to do
  int4 nd0, nd1;
  int sc[4];
  int ppage, lab, ekey, cost;
  uint64_t tq[4];
  __device__ __forceinline__ void begin(int node) {
    nd0 = reinterpret_cast<const int4*>(nodes + (size_t)node * ND_WORDS)[0];
    nd1 = reinterpret_cast<const int4*>(nodes + (size_t)node * ND_WORDS)[1];
    stage = 1;
  }
  __device__ __forceinline__ void step() {
    if (stage == 1) {
      if (nd1.x != ND_OPEN) {  // stale duplicate of a closed road: nothing more is looked up
        stage = 3;
        return;
      }
      const int rid = nd0.x, parent = nd0.y, date_time = nd0.w, beg = nd1.y, C = nd1.z;
#pragma unroll
      for (int k = 0; k < 4; ++k) sc[k] = C > 0 ? indices[beg + (k < C ? k : C - 1)] : 0;
      ppage = parent >= 0 ? my_pages[parent >> TSG_PAGE_SHIFT] : 0;
      lab = (allow >= 0 && label) ? label[rid] : -1;
      ekey = (C == 0 && empty_key) ? empty_key[rid] : 0;
      cost = cost_tab[(size_t)(date_time / 3600) * n_roads + rid];  // search.py:376-381 (<0: per pair)
      stage = 2;
    } else if (stage == 2 && TSG_POP_STAGES >= 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) tq[k] = table[sc[k]];
      stage = 3;
    }
  }
  __device__ __forceinline__ void finish() {
    if (stage == 1) step();
    if (stage == 2) {  // (with TSG_POP_STAGES == 1 the table entries are always requested here, after the sift)
#pragma unroll
      for (int k = 0; k < 4; ++k) tq[k] = table[sc[k]];
      stage = 3;
    }
  }
};

__global__ __launch_bounds__(256, 3) void frontier_kernel(EngineDev D, int superstep) {
  const int b = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * TSG_SPW + (threadIdx.x & 63);
  if (b >= D.e.slots || (threadIdx.x & 63) >= TSG_SPW) return;
  const tsg_graph& g = D.g;
  const tsg_od_batch& od = D.od;
  const int md = g.max_degree;
  int32_t* S = D.e.slot_state + (size_t)b * SS_WORDS;
  int32_t* pc = D.e.counters;
  // ---- first round trip: the slot state, the control words and (whether or not they are needed) the first
  // successors and scored log-probs of the expansion this search may be waiting for
  int4 sq[SS_WORDS / 4];
#pragma unroll
  for (int i = 0; i < SS_WORDS / 4; ++i) sq[i] = reinterpret_cast<const int4*>(S)[i];
  const int ss_now = pc[PC_SUPERSTEP];
  const int tail_snap = pc[PC_TAIL_SNAP], rtail_snap = pc[PC_RTAIL_SNAP];
  const int abort_level = pc[PC_ABORT_LEVEL];
#if !TSG_RELAX_IN_SCORE
  int cc0[4];
  float lp0[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = k < md ? k : md - 1;
    cc0[k] = D.e.cand_id[(size_t)b * md + i];
    lp0[k] = D.e.cand_lp[(size_t)b * md + i];
  }
#endif
  // the plane of item_counts this launch fills (superstep parity), for the kernels that follow; nothing reads
  // the word during this launch
  if (b == 0) pc[PC_ITEM_PLANE] = ss_now & 1;
  const int* sv = reinterpret_cast<const int*>(sq);
#if TSG_STAMP
  long long st_acc[32] = {0};
  const long long st_t0 = clock64();
  long long st_last = st_t0;
#endif
  int phase = sv[SS_PHASE];
  if (phase == PH_DONE) return;
  CKPT(0);  // first round trip
  int32_t itw[IT_WORDS] = {0};  // the work item of this superstep, if one is emitted: written to the work list at the end
  int32_t* shared = D.e.shared_counters;  // OD queue + completion, common to all sub-engines of a batch
  // the slot's queue stretch: a blocked heap (tsg_heap.h), heap_cap words
  uint64_t* heap = D.e.heap + (size_t)b * D.e.heap_cap;
  const int heap_room = tsg_heap_room(D.e.heap_cap);
  uint64_t* table = D.e.table + (size_t)b * g.n_roads;
  int32_t* nodes = D.e.nodes + (size_t)b * D.e.node_cap * ND_WORDS;
  int32_t* em = D.e.exp_meta + (size_t)b * D.e.exp_cap * EM_WORDS;

  // The words every superstep works with live in registers and go back at the end; the others (prefix, fallback
  // snapshot, the expansion waited for) are written where they change and read where they are needed, and the
  // statistics are added with atomics nobody waits for: registers held across the whole kernel cost occupancy.
  int odi = sv[SS_OD], step = sv[SS_STEP], heap_n = sv[SS_HEAP_N], node_n = sv[SS_NODE_N], exp_n = sv[SS_EXP_N];
  int gen = sv[SS_GEN], woff = sv[SS_WOFF], des = sv[SS_DES], deflen = sv[SS_DEFLEN], max_step = sv[SS_MAXSTEP];
  int allow = sv[SS_ALLOW], npop = sv[SS_NPOP];
  const int cur_exp = sv[SS_CUR_EXP], cur_rid = sv[SS_CUR_RID], cur_c = sv[SS_CUR_C], cur_date_time = sv[SS_CUR_DATE_TIME];
  const int cur_cost = sv[SS_CUR_COST], cur_lp = sv[SS_CUR_LP], cur_depth = sv[SS_CUR_DEPTH];
  const float best_lp_in = __int_as_float(sv[SS_BEST_LP]);
  int st_evals = sv[SS_ST_EVALS], st_cand = sv[SS_ST_CAND], st_chain = sv[SS_ST_CHAIN], st_items = sv[SS_ST_ITEMS];
  int st_flags = 0;  // rare statistics of this superstep: bit 2 stall, 8.. skipped expansions
  int4 cur_new = make_int4(0, 0, 0, 0), cur_new2 = make_int4(0, 0, 0, 0);  // words 8..15 when they change
  bool cur_dirty = false;
  int err = 0;
  // state pages (include/tsg.h, tsg_engine): this search's page list, the shared free ring, control words
  int pages_held = sv[SS_PAGES], last_page = sv[SS_LAST_PAGE];
  int32_t* my_pages = D.e.slot_pages + (size_t)b * D.e.pages_per_slot;
  const int ss_stamp = ss_now + 1;
  const bool active_at_entry = phase != PH_IDLE;
  const unsigned n_pages_u = (unsigned)D.e.n_pages;
  const unsigned retry_cap_u = (unsigned)(D.e.retry_bytes / 4);
  // make sure the state row of expansion index `idx` exists; false = no free page right now
  auto ensure_page = [&](int idx) -> bool {
    if (pages_held > (idx >> TSG_PAGE_SHIFT)) return true;
    if (pc[PC_HEAD] >= tail_snap) return false;  // cheap look before the atomic
    const int i = atomicAdd(&pc[PC_HEAD], 1);
    if (i >= tail_snap) return false;            // overshoot, clamped by the control step
    last_page = D.e.page_ring[(unsigned)i % n_pages_u];
    my_pages[pages_held++] = last_page;
    return true;
  };
  auto release_pages = [&]() {
    if (pages_held > 0) {
      const int j = atomicAdd(&pc[PC_TAIL], pages_held);
      // sixteen page ids per round trip (a load behind every store would make this one round trip per page,
      // and the whole launch waits for the searches that end in it)
      for (int k0 = 0; k0 < pages_held; k0 += 16) {
        int p[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) p[u] = my_pages[k0 + u < pages_held ? k0 + u : pages_held - 1];
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (k0 + u < pages_held) D.e.page_ring[(unsigned)(j + k0 + u) % n_pages_u] = p[u];
      }
      pages_held = 0;
    }
  };
  // a search that found no free page: wait, or (deadlock control) give the pages back and queue again
  auto stall = [&]() {
    D.e.page_flags[TSG_FLAG_PLANE + (b & 255) * 32] = ss_stamp;
    st_flags |= 4;
    if (abort_level > 0 && pages_held <= abort_level && active_at_entry && pc[PC_ABORT_BUDGET] > 0 &&
        atomicSub(&pc[PC_ABORT_BUDGET], pages_held + 1) > 0) {
      release_pages();
      const int j = atomicAdd(&pc[PC_RTAIL], 1);
      D.e.retry[(unsigned)j % retry_cap_u] = odi;
      atomicAdd(&pc[PC_ABORTS], 1);
      phase = PH_IDLE;
    }
  };
  // close a node: it leaves rid2node for close_set (search.py:288-289); the rid table then answers +inf
  auto close_node = [&](int node, int rid, int exp_word) {
    nodes[(size_t)node * ND_WORDS + ND_EXP] = exp_word;
    reinterpret_cast<uint32_t*>(table + rid)[1] = 0x7f800000u;
  };

  // ---- relax the expansion scored during the previous superstep (search.py:385-419 / 317-350), unless the scoring
  // wave has done it already (it then left PH_SEARCH behind)
  if (phase == PH_EXPANDED) {
    RelaxCtx r;
    r.b = b;
    r.e = cur_exp;
    r.C = cur_c;
    r.rid = cur_rid;
    r.date_time = cur_date_time;
    r.cost = cur_cost;
    r.child_depth = cur_depth + 1;
    r.gen = gen;
    r.woff = woff;
    r.deflen = deflen;
    r.best_lp = best_lp_in;
    r.heap_n = heap_n;
    r.node_n = node_n;
    r.err = 0;
    r.heap_room = heap_room;
    r.node_cap = D.e.node_cap;
    r.n_roads = g.n_roads;
    r.heap = heap;
    r.table = table;
    r.nodes = nodes;
    r.S = S;
    r.pair_cost = g.pair_cost;
    const int C = cur_c;
    const float lp_parent = __int_as_float(cur_lp);
    for (int i0 = 0; i0 < C && !r.err; i0 += 4) {
      // the table entries (and CSR rows, for the nodes that turn out new) of up to four successors are fetched
      // together
      int cc[4];
      float lps[4];
      uint64_t tt[4];
      int cb[4], ce[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k < C ? i0 + k : C - 1;
#if !TSG_RELAX_IN_SCORE
        if (i0 == 0) {  // came with the slot state
          cc[k] = cc0[k < C ? k : C - 1];
          lps[k] = (C == 1) ? lp_parent : lp0[k < C ? k : C - 1];
          continue;
        }
#endif
        cc[k] = D.e.cand_id[(size_t)b * md + i];
        lps[k] = (C == 1) ? lp_parent : D.e.cand_lp[(size_t)b * md + i];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        tt[k] = table[cc[k]];
        cb[k] = g.indptr[cc[k]];
        ce[k] = g.indptr[cc[k] + 1];
      }
      relax_batch(r, i0, cc, lps, tt, cb, ce);
    }
    heap_n = r.heap_n;
    node_n = r.node_n;
    err = r.err;
    phase = PH_SEARCH;
    CKPT(1);  // relax
    STAMP_ADD(12, 1);
  }

  bool emitted = false;
  int item_depth = 0;
  int pops_here = 0;
  bool wrote_out = false;
  if (phase == PH_FINISH) {  // the search ended in the previous superstep: write its result, give the pages back
    int fin_status = sv[SS_FIN_STATUS];
    const int fin_parent = sv[SS_FIN_PARENT], fin_rid = sv[SS_FIN_RID], fin_tcode = sv[SS_FIN_TCODE];
    const int fin_depth = sv[SS_FIN_DEPTH];
    const int lmax = od.lmax;
    if (fin_depth > lmax) fin_status |= TSG_STATUS_OVERFLOW;
    const int parent_row =
        fin_parent >= 0 ? my_pages[fin_parent >> TSG_PAGE_SHIFT] * TSG_PAGE_ROWS + (fin_parent & (TSG_PAGE_ROWS - 1)) : -1;
    write_path_rows(D.e, parent_row, fin_rid, fin_tcode, fin_depth, od.out_rid + (size_t)odi * lmax,
                    od.out_time + (size_t)odi * lmax, lmax);
    od.out_len[odi] = fin_depth;
    od.out_status[odi] = fin_status;
    od.out_steps[odi] = step;
    if (od.poplog_n) od.poplog_n[odi] = npop;
    atomicAdd(&shared[CT_EXPANSIONS], step);
    atomicAdd(&shared[CT_FINISHED], 1);
    release_pages();
    D.e.page_flags[(b & 255) * 32] = ss_stamp;
    phase = PH_IDLE;
    wrote_out = true;
  }
  while (!emitted && !err && !wrote_out) {
    if (phase == PH_IDLE) {
      if (gen >= TSG_GEN_MAX) {  // table generation space exhausted for this slot: stop refilling it
        phase = PH_DONE;
        break;
      }
      // admission: closed while the page pool is short; the first search of slot b waits for its place on the
      // start-up ramp; OD pairs that were sent back come first.  (Plain loads decide whether the atomic is
      // worth issuing: thousands of idle slots must not hammer one address.)
      int take = -1;
      const bool ramp_ok = gen > 0 || (long long)b * D.e.ramp_steps <= (long long)(ss_stamp - 1) * D.e.slots;
      const bool pending = pc[PC_RHEAD] < rtail_snap || shared[CT_NEXT_OD] < od.n_od;
      if (pending && ramp_ok && pc[PC_ADMIT_QUOTA] > 0 && atomicSub(&pc[PC_ADMIT_QUOTA], 1) > 0) {
        if (pc[PC_RHEAD] < rtail_snap) {
          const int i = atomicAdd(&pc[PC_RHEAD], 1);
          if (i < rtail_snap) take = D.e.retry[(unsigned)i % retry_cap_u];
        }
        if (take < 0 && shared[CT_NEXT_OD] < od.n_od) {
          const int i = atomicAdd(&shared[CT_NEXT_OD], 1);
          if (i < od.n_od) take = i;
        }
      }
      if (take < 0) break;  // nothing to start now: stay idle (sent-back OD pairs may still arrive)
      odi = take;
      // ---- start a new search (search.py:255-278)
      ++gen;
      const int poff = od.prefix_ptr[odi];
      const int plen = od.prefix_ptr[odi + 1] - poff;
      des = od.des[odi];
      deflen = od.default_len[odi];
      max_step = od.max_step[odi];
      allow = od.allow_label ? od.allow_label[odi] : -1;
      if (plen < 1 || max_step < 0 || plen - 1 + max_step > D.e.exp_cap) {
        // the expansion records of this engine cannot hold the search the caller asked for: report it (CT_ERROR,
        // the OD pair ends as OVERFLOW) instead of running a different, shorter search
        step = 0;
        err = 1;
        break;
      }
      const int start_time = od.prefix_time[poff + plen - 1];
      woff = start_time >= 1440 ? 1440 : 0;
      for (int k = 0; k + 1 < plen; ++k) {
        em[(size_t)k * EM_WORDS + EM_PARENT] = k - 1;
        em[(size_t)k * EM_WORDS + EM_RID] = od.prefix_rid[poff + k];
        em[(size_t)k * EM_WORDS + EM_TCODE] = od.prefix_time[poff + k];
        em[(size_t)k * EM_WORDS + EM_DEPTH] = k + 1;
      }
      const int r0 = od.prefix_rid[poff + plen - 1];
      {
        const int rb = g.indptr[r0], re = g.indptr[r0 + 1];
        int4* nd = reinterpret_cast<int4*>(nodes);
        nd[0] = make_int4(r0, plen - 2, __float_as_int(0.0f), (start_time - woff) * 60);
        nd[1] = make_int4(ND_OPEN, rb, re - rb, plen);
      }
      table[r0] = (uint64_t)((uint32_t)gen << TSG_GEN_SHIFT);  // node 0, log_prob 0
      HeapEnt he;
      he.prio = 0.0f;
      he.node = 0;
      heap_st(heap, heap_addr(1), he);
      heap_n = 1;
      node_n = 1;
      exp_n = plen - 1;
      step = 0;
      npop = 0;
      S[SS_OD] = odi;
      S[SS_GEN] = gen;
      S[SS_WOFF] = woff;
      S[SS_DES] = des;
      S[SS_DEFLEN] = deflen;
      S[SS_MAXSTEP] = max_step;
      S[SS_ALLOW] = allow;
      S[SS_WARM_K] = 0;
      S[SS_PLEN] = plen;
      S[SS_POFF] = poff;
      S[SS_HAVE_BEST] = 0;
      S[SS_BEST_LP] = __float_as_int(-INFINITY);
      S[SS_BEST_PARENT] = -1;
      S[SS_BEST_RID] = -1;
      S[SS_BEST_TCODE] = 0;
      phase = plen > 1 ? PH_WARMUP : PH_SEARCH;
    }
    if (phase == PH_WARMUP) {  // run the recurrent cell over the known prefix, one element per superstep
      const int prev_page = last_page;  // page of element k - 1 when it sits in the page before
      const int k = S[SS_WARM_K], plen = S[SS_PLEN], poff = S[SS_POFF];
      if (!ensure_page(k)) {
        stall();
        break;
      }
      S[SS_WARM_K] = k + 1;
      int32_t* it = itw;
      it[IT_SLOT] = b;
      it[IT_ROW] = last_page * TSG_PAGE_ROWS + (k & (TSG_PAGE_ROWS - 1));
      it[IT_PROW] = k > 0 ? ((k & (TSG_PAGE_ROWS - 1)) ? last_page : prev_page) * TSG_PAGE_ROWS +
                                ((k - 1) & (TSG_PAGE_ROWS - 1))
                          : -1;
      it[IT_RID] = od.prefix_rid[poff + k];
      it[IT_TCODE] = od.prefix_time[poff + k];
      it[IT_NCAND] = 0;
      it[IT_DES] = des;
      it[IT_LP] = 0;
      it[IT_DEPTH] = k + 1;
      *reinterpret_cast<int2*>(D.e.row_meta + 2 * (size_t)it[IT_ROW]) = make_int2(it[IT_RID], it[IT_TCODE]);
      item_depth = k + 1;
      st_items += 1;
      D.e.page_flags[(b & 255) * 32] = ss_stamp;
      if (k + 1 >= plen - 1) phase = PH_SEARCH;
      emitted = true;
      break;
    }
    // the next expansion needs a state row: have its page before anything is popped
    if (heap_n > 0 && step < max_step && !ensure_page(exp_n)) {
      stall();
      break;
    }
    // ---- PH_SEARCH: pop until a node can be expanded (search.py:279-300)
    int fin_status = -1, fin_parent = -1, fin_rid = -1, fin_tcode = 0, fin_depth = 0;
    bool yielded = false;
    PopSide side;
    side.nodes = nodes;
    side.indices = g.indices;
    side.my_pages = my_pages;
    side.label = g.label;
    side.empty_key = g.empty_key;
    side.cost_tab = g.cost;
    side.table = table;
    side.n_roads = g.n_roads;
    side.allow = allow;
    side.stage = 0;
    while (heap_n > 0 && step < max_step) {
      // The launch lasts as long as its slowest search, and the number of pops a search needs before it
      // finds a node worth expanding has a long tail (stale duplicates, expansions that cannot push
      // anything).  After max_pops pops the search yields: it emits no work item and carries on in the next
      // superstep.  The operations of a search and their order do not change, only the superstep they land in.
      if (D.e.max_pops > 0 && pops_here >= D.e.max_pops) {
        yielded = true;
        break;
      }
      ++pops_here;
      CKPT(2);
      side.stage = 0;
      const HeapEnt top = heap_pop(heap, heap_n, nodes, side);
      side.finish();
      CKPT(3);  // pop
      STAMP_ADD(13, 1);
      // the node record: rid, parent, log_prob, date_time | state, CSR row of the road, depth
      const int4 nd0 = side.nd0, nd1 = side.nd1;
      CKPT(4);  // node record
      if (nd1.x != ND_OPEN) continue;  // stale duplicate of a closed rid (search.py:284-287)
      const int rid = nd0.x, parent = nd0.y, nd_lp = nd0.z, date_time = nd0.w;
      const int beg = nd1.y, C = nd1.z, depth = nd1.w;
      const int tcode = date_time / 60 + woff;
      if (od.poplog_cap > 0 && npop < od.poplog_cap) {
        od.poplog_rid[(size_t)odi * od.poplog_cap + npop] = rid;
        od.poplog_prio[(size_t)odi * od.poplog_cap + npop] = top.prio;
      }
      ++npop;
      if (rid == des) {  // search.py:291-297
        close_node(top.node, rid, ND_CLOSED_NOSLOT);
        fin_status = TSG_STATUS_REACHED;
        fin_parent = parent;
        fin_rid = rid;
        fin_tcode = tcode;
        fin_depth = depth;
        break;
      }
      // the first successors, the parent's page, the road's label and cost: requested during the sift (PopSide)
      int sc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) sc[k] = side.sc[k];
      const int ppage = side.ppage, lab = side.lab, ekey = side.ekey, cost = side.cost;
      const bool gate = C > 0 && (allow < 0 || (g.label && lab == allow));    // search.py:300 / :487
      CKPT(5);  // successors etc
      if (ekey) {  // key present, no successors: `continue` (search.py:993-995)
        close_node(top.node, rid, ND_CLOSED_NOSLOT);
        continue;
      }
      ++step;  // search.py:420
      if (!gate) {
        close_node(top.node, rid, ND_CLOSED_NOSLOT);
        continue;
      }
      {  // Expansions that cannot change anything.  The reference still evaluates the model (search.py:359-367)
         // but pushes nothing when every successor is either closed (:388 fails, :403 needs rid2node) or open
         // with a log_prob that no outcome of this evaluation can beat: f = softmax(mix) with every mix_c
         // inside an interval of width s = |1-w1| + |w1| gives f_c <= e^s / (e^s + C - 1).  Nothing pushed
         // means no child will ever need this node's recurrent state, so it is closed without an LSTM step
         // or a scoring pass; results are unchanged.
        const float lp_node = __int_as_float(nd_lp);
        float ub = lp_node;  // C == 1: the child inherits the log_prob (search.py:305)
        bool use_bound = C == 1;
        if (C > 1 && D.e.prune_spread > 0.0f) {
          const float es = __expf(D.e.prune_spread);
          ub = lp_node + logf(es / (es + (float)(C - 1))) + 1e-4f;
          use_bound = true;
        }
        bool futile = true;
        for (int i0 = 0; i0 < C && futile; i0 += 4) {
          uint64_t tq[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int i = i0 + k < C ? i0 + k : C - 1;
            tq[k] = i0 == 0 ? side.tq[k] : table[g.indices[beg + i]];  // the first four came with the sift
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool seen = (int)((uint32_t)tq[k] >> TSG_GEN_SHIFT) == gen;
            const float qlp = __uint_as_float((uint32_t)(tq[k] >> 32));  // +inf: closed
            const bool dead = seen && (use_bound ? qlp >= ub : qlp == INFINITY);
            if (!dead) futile = false;
          }
        }
        CKPT(6);  // table
        if (futile) {
          close_node(top.node, rid, ND_CLOSED_NOSLOT);
          st_flags += 256;
          continue;
        }
      }
      if (exp_n >= D.e.exp_cap) {
        err = 1;
        break;
      }
      const int e = exp_n++;
      close_node(top.node, rid, e);
      *reinterpret_cast<int4*>(em + (size_t)e * EM_WORDS) = make_int4(parent, rid, tcode, depth);
      // the successor list, for the distance and scoring kernels and for the relaxation in the next superstep
      for (int i0 = 0; i0 < C; i0 += 4) {
        int s[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = i0 == 0 ? sc[k] : g.indices[beg + (i0 + k < C ? i0 + k : C - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (i0 + k < C) D.e.cand_id[(size_t)b * md + i0 + k] = s[k];
      }
      cur_new = make_int4(e, rid, C, date_time);
      cur_new2 = make_int4(cost, nd_lp, depth, 0);
      cur_dirty = true;
      if (C > 1) {  // per-slot statistics (summed on the host; no same-address atomics on the hot path)
        st_evals += 1;
        st_cand += C;
        st_chain += depth;
      }
      int32_t* it = itw;
      it[IT_SLOT] = b;
      it[IT_ROW] = last_page * TSG_PAGE_ROWS + (e & (TSG_PAGE_ROWS - 1));
      st_items += 1;
      it[IT_PROW] = parent >= 0 ? ppage * TSG_PAGE_ROWS + (parent & (TSG_PAGE_ROWS - 1)) : -1;
      it[IT_RID] = rid;
      it[IT_TCODE] = tcode;
      it[IT_NCAND] = C;
      it[IT_DES] = des;
      it[IT_LP] = nd_lp;
      it[IT_DEPTH] = depth;
      *reinterpret_cast<int2*>(D.e.row_meta + 2 * (size_t)it[IT_ROW]) = make_int2(rid, tcode);
      item_depth = C > 1 ? depth : 0;  // single-successor expansions are not scored: short
      D.e.page_flags[(b & 255) * 32] = ss_stamp;
      phase = PH_EXPANDED;
      emitted = true;
      CKPT(7);  // expansion stores
      break;
    }
    if (emitted || err) break;
    if (yielded) break;
    // ---- the search is over (search.py:421-424): destination popped, queue empty, or budget spent
    if (fin_status < 0) {
      const int plen = S[SS_PLEN], poff = S[SS_POFF];
      if (S[SS_HAVE_BEST]) {
        fin_status = TSG_STATUS_FALLBACK;
        fin_parent = S[SS_BEST_PARENT];
        fin_rid = S[SS_BEST_RID];
        fin_tcode = S[SS_BEST_TCODE];
        fin_depth = deflen;
      } else {
        fin_status = TSG_STATUS_UNTOUCHED;
        fin_parent = plen - 2;
        fin_rid = od.prefix_rid[poff + plen - 1];
        fin_tcode = od.prefix_time[poff + plen - 1];
        fin_depth = plen;
      }
    }
    cur_new = make_int4(fin_status, fin_parent, fin_rid, fin_tcode);  // SS_FIN_* (the words of the expansion waited for)
    cur_new2 = make_int4(0, 0, fin_depth, 0);
    cur_dirty = true;
    D.e.page_flags[(b & 255) * 32] = ss_stamp;
    phase = PH_FINISH;
    break;
  }
  if (err) {
    // a workspace cap was hit: report, finish the OD as OVERFLOW so that the batch still terminates
    atomicExch(&shared[CT_ERROR], 1);
    if (odi >= 0 && odi < od.n_od) {
      od.out_len[odi] = 0;
      od.out_status[odi] = TSG_STATUS_OVERFLOW;
      od.out_steps[odi] = step;
      atomicAdd(&shared[CT_FINISHED], 1);
    }
    release_pages();
    phase = PH_IDLE;
  }
  // ---- work list of the superstep (tsg_common.h): the wave's emitting searches take their positions with one
  // atomic per class; the lanes have reconverged here (a wave that has not is still correct: every group of
  // lanes then runs this block for itself)
  {
    const bool emit = emitted && !err;
    const bool is_long = emit && item_depth >= TSG_LONG_DEPTH;
    const unsigned long long m_long = __ballot(is_long), m_short = __ballot(emit && !is_long);
    const unsigned long long m_any = m_long | m_short;
    if (m_any) {
      const int lane = threadIdx.x & 63;
      const int leader = __ffsll((long long)m_any) - 1;
      const int region = b / TSG_REGION_SLOTS;  // TSG_REGION_SLOTS is a multiple of TSG_SPW: one region per wave
      int base_l = 0, base_s = 0;
      if (lane == leader) {
        int32_t* oc = D.e.item_counts + tsg_item_cnt_index(D.e.slots, (ss_stamp - 1) & 1, region);
        if (m_long) base_l = atomicAdd(&oc[0], __popcll(m_long));
        if (m_short) base_s = atomicAdd(&oc[1], __popcll(m_short));
      }
      base_l = __shfl(base_l, leader, TSG_WAVE);
      base_s = __shfl(base_s, leader, TSG_WAVE);
      if (emit) {
        const unsigned long long below = (1ull << lane) - 1ull;
        const int p = is_long ? base_l + __popcll(m_long & below)
                              : tsg_region_size(D.e.slots, region) - 1 - (base_s + __popcll(m_short & below));
        int4* dst = reinterpret_cast<int4*>(D.e.items + ((size_t)region * TSG_REGION_SLOTS + p) * IT_WORDS);
        dst[0] = make_int4(itw[0], itw[1], itw[2], itw[3]);
        dst[1] = make_int4(itw[4], itw[5], itw[6], itw[7]);
        dst[2] = make_int4(itw[8], 0, 0, 0);
      }
    }
  }
  CKPT(8);  // emit (work-list atomic)
#if TSG_STAMP
  st_acc[9] = clock64() - st_t0; st_acc[14] = 1; st_acc[15] = st_t0; st_acc[11] = wall_clock64();
  if (ss_now == TSG_STAMP_AT && b < 32768) for (int i = 0; i < 32; ++i) g_stamp[(size_t)b * 32 + i] = (unsigned long long)st_acc[i];
#endif
  // slot state back, in whole 16-byte pieces (tsg_common.h): the words of every superstep, the expansion waited for
  // (or the result) when it changed, the statistics; the rare counters with atomics nobody waits for
  {
    int4* S4 = reinterpret_cast<int4*>(S);
    S4[0] = make_int4(phase, step, heap_n, node_n);
    S4[1] = make_int4(exp_n, npop, pages_held, last_page);
    if (cur_dirty) {
      S4[2] = cur_new;
      S4[3] = cur_new2;
    }
    if (st_items != sv[SS_ST_ITEMS]) S4[4] = make_int4(st_evals, st_cand, st_chain, st_items);
  }
  if (st_flags & 4) atomicAdd(&S[SS_ST_STALLS], 1);
  if (st_flags >> 8) atomicAdd(&S[SS_ST_SKIPPED], st_flags >> 8);
}

// =====================================================================================
// host side
// =====================================================================================
static int check_engine(const tsg_model* m, const tsg_graph* g, const tsg_od_batch* od, const tsg_engine* e) {
  if (!g || !e) return TSG_E_INVALID;
  if (g->n_roads <= 0 || g->max_degree <= 0 || !g->indptr || !g->indices || !g->cost) return TSG_E_INVALID;
  if (g->max_degree > TSG_MAX_DEGREE) return TSG_E_LIMIT;
  if (g->dist_table && !g->pair_cost) return TSG_E_INVALID;
  if (e->slots <= 0 || e->exp_cap <= 0 || e->node_cap <= 0 || e->heap_cap < 32 || (e->heap_cap % 16) != 0) return TSG_E_INVALID;
  if ((unsigned)e->node_cap > TSG_NODE_MASK) return TSG_E_LIMIT;
  const size_t B = (size_t)e->slots;
  if (!e->slot_state || e->slot_state_bytes < B * SS_WORDS * 4) return TSG_E_TOO_SMALL;
  if (!e->heap || e->heap_bytes < B * e->heap_cap * 8) return TSG_E_TOO_SMALL;
  if (!e->table || e->table_bytes < B * (size_t)g->n_roads * 8) return TSG_E_TOO_SMALL;
  if (!e->cand_id || e->cand_id_bytes < B * (size_t)g->max_degree * 4) return TSG_E_TOO_SMALL;
  if (!e->exp_up || e->exp_up_bytes < (size_t)e->n_pages * TSG_PAGE_ROWS * TSG_UP_LEVELS * 4) return TSG_E_TOO_SMALL;
  if (!e->row_meta || e->row_meta_bytes < (size_t)e->n_pages * TSG_PAGE_ROWS * 8) return TSG_E_TOO_SMALL;
  if (!e->nodes || e->nodes_bytes < B * e->node_cap * ND_WORDS * 4) return TSG_E_TOO_SMALL;
  if (!e->exp_meta || e->exp_meta_bytes < B * e->exp_cap * EM_WORDS * 4) return TSG_E_TOO_SMALL;
  if (!e->items || e->items_bytes < tsg_item_positions(e->slots) * IT_WORDS * 4) return TSG_E_TOO_SMALL;
  if (!e->cand_lp || e->cand_lp_bytes < B * (size_t)g->max_degree * 4) return TSG_E_TOO_SMALL;
  if (!e->counters || e->counters_bytes < PC_WORDS * 4) return TSG_E_TOO_SMALL;
  if (!e->shared_counters || e->shared_counters == e->counters) return TSG_E_INVALID;
  if (e->pages_per_slot != tsg_cdiv(e->exp_cap, TSG_PAGE_ROWS)) return TSG_E_INVALID;
  if (e->n_pages < e->pages_per_slot || e->n_pages > (1 << 24)) return TSG_E_INVALID;  // one search must always fit
  if (e->admit_reserve < 0 || e->ramp_steps < 0) return TSG_E_INVALID;
  if (!e->slot_pages || e->slot_pages_bytes < B * e->pages_per_slot * 4) return TSG_E_TOO_SMALL;
  if (!e->page_ring || e->page_ring_bytes < (size_t)e->n_pages * 4) return TSG_E_TOO_SMALL;
  if (!e->retry || e->retry_bytes < B * 4) return TSG_E_TOO_SMALL;
  if (!e->page_flags || e->page_flags_bytes < 2 * TSG_FLAG_PLANE * 4) return TSG_E_TOO_SMALL;
  if (!e->item_counts || e->item_counts_bytes < tsg_item_count_words(e->slots) * 4) return TSG_E_TOO_SMALL;
  if (m) {
    if (m->hp <= 0 || (m->hp % 64) != 0 || m->hp > 512) return TSG_E_INVALID;
    if (m->n_layers < 1 || m->n_layers > TSG_MAX_LAYERS) return TSG_E_INVALID;
    if (m->n_roads < g->n_roads) return TSG_E_INVALID;
    if (!g->geo && !g->dist_table) return TSG_E_INVALID;
    if (e->up_levels != TSG_UP_LEVELS) return TSG_E_INVALID;
    if (e->state_format != 0 && e->state_format != 1) return TSG_E_INVALID;
    if (e->max_pops < 0) return TSG_E_INVALID;
    if (e->state_format == 1) {
      if (!m->whh0_s) return TSG_E_INVALID;
      for (int l = 1; l < m->n_layers; ++l)
        if (!m->wl_s[l]) return TSG_E_INVALID;
    }
    const size_t rows = (size_t)e->n_pages * TSG_PAGE_ROWS;
    if (!e->exp_up || e->exp_up_bytes < rows * (size_t)e->up_levels * 4) return TSG_E_TOO_SMALL;
    for (int l = 0; l < m->n_layers; ++l)
      if (!e->h[l] || !e->c[l] || e->state_bytes < rows * (size_t)m->hp * 4) return TSG_E_TOO_SMALL;
  }
  if (od) {
    if (od->n_od < 0 || od->lmax <= 0) return TSG_E_INVALID;
    if (!od->prefix_ptr || !od->prefix_rid || !od->prefix_time || !od->des || !od->default_len || !od->max_step)
      return TSG_E_INVALID;
    if (!od->out_rid || !od->out_time || !od->out_len || !od->out_status || !od->out_steps) return TSG_E_INVALID;
    if (od->poplog_cap > 0 && (!od->poplog_rid || !od->poplog_prio || !od->poplog_n)) return TSG_E_INVALID;
    if (od->allow_label && !g->label) return TSG_E_INVALID;
    if (e->retry_bytes < ((size_t)od->n_od + B) * 4) return TSG_E_TOO_SMALL;
  }
  return 0;
}

extern "C" int tsg_engine_reset(const tsg_engine* e, const tsg_graph* g, void* stream) {
  int rc = check_engine(nullptr, g, nullptr, e);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  TSG_HIP_TRY(hipMemsetAsync(e->slot_state, 0, (size_t)e->slots * SS_WORDS * 4, s));  // phase IDLE, gen 0
  TSG_HIP_TRY(hipMemsetAsync(e->table, 0, (size_t)e->slots * g->n_roads * 8, s));
  TSG_HIP_TRY(hipMemsetAsync(e->page_flags, 0, 2 * TSG_FLAG_PLANE * 4, s));
  TSG_HIP_TRY(hipMemsetAsync(e->items, 0, tsg_item_positions(e->slots) * IT_WORDS * 4, s));
  TSG_HIP_TRY(hipMemsetAsync(e->item_counts, 0, tsg_item_count_words(e->slots) * 4, s));
  hipLaunchKernelGGL(engine_init_pages_kernel, dim3(tsg_cdiv(e->n_pages, 256)), dim3(256), 0, s, *e);
  TSG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tsg_reconstruct_paths(const tsg_engine* e, int32_t n, const int32_t* slot, const int32_t* node,
                                     const int32_t* last_rid, const int32_t* last_tcode, int32_t lmax,
                                     int32_t* out_rid, int32_t* out_time, int32_t* out_len, void* stream) {
  if (!e || e->slots <= 0 || e->exp_cap <= 0 || !e->exp_meta) return TSG_E_INVALID;
  if (e->exp_meta_bytes < (size_t)e->slots * e->exp_cap * EM_WORDS * 4) return TSG_E_TOO_SMALL;
  if (n < 0 || lmax <= 0 || !slot || !node || !out_rid || !out_time || !out_len) return TSG_E_INVALID;
  if (n == 0) return 0;
  hipLaunchKernelGGL(reconstruct_paths_kernel, dim3(tsg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, *e, n, slot,
                     node, last_rid, last_tcode, lmax, out_rid, out_time, out_len);
  TSG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tsg_frontier_expand_select(const tsg_graph* g, const tsg_od_batch* od, const tsg_engine* e,
                                          int32_t superstep, void* stream) {
  int rc = check_engine(nullptr, g, od, e);
  if (rc) return rc;
  EngineDev D;
  D.e = *e;
  D.g = *g;
  D.od = *od;
  hipLaunchKernelGGL(frontier_kernel, dim3(tsg_cdiv(e->slots, TSG_FRONTIER_WPB * TSG_SPW)), dim3(64 * TSG_FRONTIER_WPB), 0, (hipStream_t)stream, D, superstep);
  TSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(engine_control_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, D);  // tsg_engine_run does this in cand_dis
  TSG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tsg_score_frontier(const tsg_model* m, const tsg_graph* g, const tsg_engine* e, int32_t superstep,
                                  void* stream) {
  int rc = check_engine(m, g, nullptr, e);
  if (rc) return rc;
  if (!m->out_w || !m->out_b || !m->pfrom || !m->pto || !m->hparams) return TSG_E_INVALID;
  EngineDev D;
  D.e = *e;
  D.g = *g;
  D.od = tsg_od_batch{};
  if (!cand_dis_rides(e)) {
    launch_cand_dis(D, superstep, (hipStream_t)stream);
    TSG_LAUNCH_CHECK();
  }
  for (int l = 0; l < m->n_layers; ++l) {
    rc = tsg_launch_lstm_layer(m, l, e->slots, nullptr, e->slots, nullptr, nullptr, nullptr, nullptr, e,
                               l > 0 ? e->h[l - 1] : nullptr, e->h[l], e->c[l], (hipStream_t)stream, e->state_format,
                               cand_dis_rides(e) ? &D : nullptr);
    if (rc) return rc;
  }
  launch_score_frontier(m, D, superstep, (hipStream_t)stream);
  TSG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tsg_engine_run(const tsg_model* m, const tsg_graph* g, const tsg_od_batch* od, const tsg_engine* e,
                              int32_t first_superstep, int32_t n_supersteps, void* stream) {
  int rc = check_engine(m, g, od, e);
  if (rc) return rc;
  if (!m->out_w || !m->out_b || !m->pfrom || !m->pto || !m->hparams) return TSG_E_INVALID;
  hipStream_t s = (hipStream_t)stream;
  EngineDev D;
  D.e = *e;
  D.g = *g;
  D.od = *od;
  const dim3 fgrid(tsg_cdiv(e->slots, TSG_FRONTIER_WPB * TSG_SPW));
  for (int k = 0; k < n_supersteps; ++k) {
    const int ss = first_superstep + k;
    hipLaunchKernelGGL(frontier_kernel, fgrid, dim3(64 * TSG_FRONTIER_WPB), 0, s, D, ss);
    TSG_LAUNCH_CHECK();
    if (!cand_dis_rides(e)) {
      launch_cand_dis(D, ss, s);
      TSG_LAUNCH_CHECK();
    }
    for (int l = 0; l < m->n_layers; ++l) {
      rc = tsg_launch_lstm_layer(m, l, e->slots, nullptr, e->slots, nullptr, nullptr, nullptr, nullptr, e,
                                 l > 0 ? e->h[l - 1] : nullptr, e->h[l], e->c[l], s, e->state_format,
                                 cand_dis_rides(e) ? &D : nullptr);
      if (rc) return rc;
    }
    launch_score_frontier(m, D, ss, s);
    TSG_LAUNCH_CHECK();
  }
  return 0;
}

// HIP-graph form of tsg_engine_run: the launches of `n_supersteps` supersteps are captured once from
// `stream` (which must not be the legacy default stream) and replayed with one graph launch per chunk,
// which removes most of the per-kernel launch gaps (five short kernels per superstep).
extern "C" int tsg_engine_graph_create(const tsg_model* m, const tsg_graph* g, const tsg_od_batch* od,
                                       const tsg_engine* e, int32_t n_supersteps, void* stream, void** exec_out) {
  if (!exec_out || n_supersteps <= 0) return TSG_E_INVALID;
  *exec_out = nullptr;
  int rc = check_engine(m, g, od, e);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (s == nullptr) return TSG_E_INVALID;
  TSG_HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  rc = tsg_engine_run(m, g, od, e, 0, n_supersteps, stream);
  hipGraph_t graph = nullptr;
  hipError_t he = hipStreamEndCapture(s, &graph);
  if (rc) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  if (he != hipSuccess || !graph) return -(int)he - 1000;
  hipGraphExec_t exec = nullptr;
  he = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (he != hipSuccess) return -(int)he - 1000;
  *exec_out = (void*)exec;
  return 0;
}

extern "C" int tsg_engine_graph_launch(void* exec, void* stream) {
  if (!exec) return TSG_E_INVALID;
  TSG_HIP_TRY(hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream));
  return 0;
}

extern "C" int tsg_engine_graph_destroy(void* exec) {
  if (!exec) return 0;
  TSG_HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)exec));
  return 0;
}

// Profiling variant: same launches, each bracketed by HIP events on `stream`; synchronises the stream
// before returning (the caller polls the completion counter after every chunk anyway).
extern "C" int tsg_engine_run_profiled(const tsg_model* m, const tsg_graph* g, const tsg_od_batch* od,
                                       const tsg_engine* e, int32_t first_superstep, int32_t n_supersteps,
                                       void* stream, float* ms_sum_host, int32_t* launches_host) {
  int rc = check_engine(m, g, od, e);
  if (rc) return rc;
  if (!ms_sum_host || !launches_host || n_supersteps <= 0 || n_supersteps > 4096) return TSG_E_INVALID;
  if (!m->out_w || !m->out_b || !m->pfrom || !m->pto || !m->hparams) return TSG_E_INVALID;
  hipStream_t s = (hipStream_t)stream;
  EngineDev D;
  D.e = *e;
  D.g = *g;
  D.od = *od;
  const int nk = 2 + m->n_layers;  // frontier, score, lstm layers
  const int nev = n_supersteps * nk * 2;
  hipEvent_t* ev = new (std::nothrow) hipEvent_t[nev];
  if (!ev) return TSG_E_INVALID;
  for (int i = 0; i < nev; ++i)
    if (hipEventCreate(&ev[i]) != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipEventDestroy(ev[j]);
      delete[] ev;
      return TSG_E_INVALID;
    }
  const dim3 fgrid(tsg_cdiv(e->slots, TSG_FRONTIER_WPB * TSG_SPW));
  int q = 0;
  rc = 0;
  for (int k = 0; k < n_supersteps && rc == 0; ++k) {
    const int ss = first_superstep + k;
    (void)hipEventRecord(ev[q++], s);
    hipLaunchKernelGGL(frontier_kernel, fgrid, dim3(64 * TSG_FRONTIER_WPB), 0, s, D, ss);
    if (!cand_dis_rides(e)) launch_cand_dis(D, ss, s);  // timed with the frontier kernel
    (void)hipEventRecord(ev[q++], s);
    for (int l = 0; l < m->n_layers && rc == 0; ++l) {
      (void)hipEventRecord(ev[q++], s);
      rc = tsg_launch_lstm_layer(m, l, e->slots, nullptr, e->slots, nullptr, nullptr, nullptr, nullptr, e,
                                 l > 0 ? e->h[l - 1] : nullptr, e->h[l], e->c[l], s, e->state_format,
                                 cand_dis_rides(e) ? &D : nullptr);
      (void)hipEventRecord(ev[q++], s);
    }
    (void)hipEventRecord(ev[q++], s);
    launch_score_frontier(m, D, ss, s);
    (void)hipEventRecord(ev[q++], s);
  }
  hipError_t he = hipStreamSynchronize(s);
  if (rc == 0 && he == hipSuccess) {
    q = 0;
    for (int k = 0; k < n_supersteps; ++k) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, ev[q], ev[q + 1]);
      q += 2;
      ms_sum_host[0] += ms;
      launches_host[0] += 1;
      for (int l = 0; l < m->n_layers; ++l) {
        (void)hipEventElapsedTime(&ms, ev[q], ev[q + 1]);
        q += 2;
        ms_sum_host[2 + l] += ms;
        launches_host[2 + l] += 1;
      }
      (void)hipEventElapsedTime(&ms, ev[q], ev[q + 1]);
      q += 2;
      ms_sum_host[1] += ms;
      launches_host[1] += 1;
    }
  }
  for (int i = 0; i < nev; ++i) (void)hipEventDestroy(ev[i]);
  delete[] ev;
  if (rc) return rc;
  if (he != hipSuccess) return -(int)he - 1000;
  return 0;
}
