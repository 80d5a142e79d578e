/*
 * tsg.h -- C ABI of libtsg.so: gfx950 (MI355X) HIP kernels for the A*-guided trajectory
 * generation path of TS-TrajGen.
 *
 * The reference is pure Python on stock PyTorch ops and has NO FFI / plugin layer for this
 * path, so there is no existing binding to match symbol-for-symbol.
 * Each entry point below therefore cites the reference *Python* interface it replaces; the
 * host-side mirror of that interface lives in ts-trajgen_amd/ and is what a maintainer binds
 * (INTEGRATION.md shows the ctypes stubs).
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer unless its name ends in _host.
 *   - The caller allocates and owns every buffer, workspaces included; nothing here calls
 *     hipMalloc/hipFree.  Descriptors carry the byte size of every workspace and the library
 *     checks them (and all dims) on the host before any launch.
 *   - All functions only enqueue work on `stream` (a hipStream_t passed as void*); no hidden
 *     synchronisation, no global mutable state, re-entrant across streams/devices.
 *   - Return value: 0 on success, a negative TSG_E_* code or -(hipError_t) - 1000 otherwise.
 *     No C++ exception crosses the boundary.
 *   - Arithmetic: fp32 (int32 ids/time codes, fp64 only inside the geodesic distance), with ONE documented
 *     exception: the engine's default recurrence (tsg_engine.state_format == 1, what Searcher uses unless told
 *     otherwise) carries every LSTM operand as two f16 numbers (22 significant bits, two fewer than fp32), runs
 *     three f16 matrix-core passes per product with fp32 accumulation, and stores hidden states in that form;
 *     gates use v_exp_f32 / v_rcp_f32.  state_format == 0 is the plain fp32 matrix-core path.  Both forms give the
 *     oracle's trajectories on every golden and full-size test; a near-tie (|df| < 1e-5) may fall differently.
 */
#ifndef TSG_H_
#define TSG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSG_ABI_VERSION 2

#define TSG_E_INVALID (-1)   /* bad dims / null pointer / misaligned */
#define TSG_E_TOO_SMALL (-2) /* a caller-provided buffer is smaller than required */
#define TSG_E_LIMIT (-3)     /* a documented limit is exceeded (e.g. out-degree > 64) */

#define TSG_MAX_LAYERS 4
#define TSG_MAX_DEGREE 64 /* successors per road the scoring kernel handles in one wave */
#define TSG_UP_LEVELS 64  /* ancestors listed per expanded node (exp_up row width) */
#define TSG_H_HIDDEN 10   /* emb_fc / dis_fc width, hard-coded at distance_gat_fc.py:38-40 */

/* per-OD result status (mirrors the three exits of search.py:421-424) */
#define TSG_STATUS_UNTOUCHED 0 /* no best_trace: the input prefix is returned */
#define TSG_STATUS_REACHED 1   /* destination popped (search.py:291-297) */
#define TSG_STATUS_FALLBACK 2  /* best path of length default_len (search.py:397-402) */
#define TSG_STATUS_OVERFLOW 0x100 /* OR-ed in: path longer than lmax (truncated) or a workspace cap hit */

int tsg_version(void);

/* ------------------------------------------------------------------------------------------
 * Dense helper:  C[M,N] = A[rows,K] . Bt[K,N] (+ bias[N])      (fp32 MFMA, exact-fp32 fma chains)
 * Replaces the one-off `nn.Linear`/`nn.Embedding`-product tables the engine pre-factors:
 * road_emb.W_ih_l0^T, time_emb.W_ih_l0^T (function_g_fc.py:155-169) and node_emb.emb_fc^T
 * (distance_gat_fc.py:129-131).  lda/ldb/ldc in elements.  K and N need not be tile multiples.
 * ------------------------------------------------------------------------------------------ */
int tsg_gemm_f32(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C, int ldc, int M,
                 int N, int K, void* stream);

/* ------------------------------------------------------------------------------------------
 * GAT road-graph encoder.  Replaces GATLayerImp3.forward + GATLayer.skip_concat_bias
 * (generator/gat.py:138-200, :84-114) as called from DistanceGatFC._setup_node_emb /
 * update_node_emb (generator/distance_gat_fc.py:189-223), eval mode, ELU activation.
 * ------------------------------------------------------------------------------------------ */
typedef struct tsg_gat_desc {
  int32_t n_nodes;     /* N */
  int32_t n_edges;     /* E */
  int32_t fin;         /* FIN  (input features per node) */
  int32_t heads;       /* NH */
  int32_t fout;        /* F per head */
  int32_t concat;      /* 0: mean over heads (out is N x F), 1: concat (out is N x NH*F) */
  int32_t skip_identity; /* 1 when FIN == F: skip adds x itself (gat.py:94-97) */
  int32_t reserved;
  const float* x;        /* [N][FIN] */
  const float* w_proj_t; /* [FIN][NH*F]  = linear_proj.weight^T */
  const float* w_skip_t; /* [FIN][NH*F]  = skip_proj.weight^T (ignored when skip_identity) */
  const float* a_src;    /* [NH][F] scoring_fn_source */
  const float* a_trg;    /* [NH][F] scoring_fn_target */
  const float* bias;     /* [F] or [NH*F] (concat), may be NULL */
  const int32_t* in_ptr; /* [N+1] CSR by TARGET node, edges kept in original COO order */
  const int32_t* in_src; /* [E] source node of every incoming edge */
  float* proj;           /* workspace [N][NH*F]: the projection x . W^T, or (input-space form, below) the
                            attention-weighted input rows [N][(NH+1)*FIN] */
  float* s_src;          /* workspace [N][NH] */
  float* s_trg;          /* workspace [N][NH] */
  float* gmax;           /* workspace [1]: global max edge score (gat.py:221) */
  float* out;            /* [N][F] or [N][NH*F] */
  size_t proj_bytes, s_bytes, out_bytes;
  /* skip connection (gat.py:93-102) as a dense product; unused (may be NULL) when skip_identity */
  float* skip;           /* workspace, same shape as out: x . Ws^T + bias, with Ws averaged over heads when
                            concat == 0 (the head mean of gat.py:109 commutes with the skip projection) */
  float* w_skip_mean;    /* workspace [FIN][F], concat == 0 only */
  size_t skip_bytes, w_skip_mean_bytes;
  float* w_att;          /* optional workspace [FIN][2*NH]: W_h a_src[h] | W_h a_trg[h]; when given, the per-node
                            scores are computed from x (FIN multiply-adds) instead of re-reading proj */
  size_t w_att_bytes;
} tsg_gat_desc;

/* Two forms of the same layer.  With concat == 0, a learned skip projection (FIN != F), FIN <= 64, w_att given
 * and N >= (NH+1)*FIN (so that the stacked weights fit `skip`), the layer runs in the INPUT SPACE: per head the
 * attention-weighted sum of the FIN-wide source rows, then one product [N][(NH+1)*FIN] . [(NH+1)*FIN][F] with
 * bias and ELU in its epilogue; the N x NH x F projection is never written.  sum_j att_j (x_j W) becomes
 * (sum_j att_j x_j) W: equal as real numbers, rounded in another order (both forms are held to 1e-4 of the
 * reference's output by G1/G2).  Every other case materialises the projection and adds the gathered rows in
 * the reference's COO order. */

int tsg_gat_forward(const tsg_gat_desc* d, void* stream);

/* tsg_gat_forward without any allocation, by the caller or the library.  No counterpart in the reference: it
 * is for callers that cannot allocate device memory (code under HIP-graph capture, tight serving loops,
 * harnesses that forbid hipMalloc), e.g. re-encoding a region graph (our_model_generate.py:62-83: ~100-1000
 * regions) after a weight update.  Same descriptor, same kernels, same results as tsg_gat_forward, but the
 * WORKSPACE fields of the descriptor (proj, s_src, s_trg, gmax, skip, w_skip_mean, w_att and their *_bytes)
 * are ignored: the workspaces live in a file-scope device array of TSG_GAT_STATIC_SCRATCH_BYTES that every GPU
 * the library runs on holds until the process ends.  Inputs (x, weights, CSR) and `out` are still the caller's.
 *   - Limit: the workspaces, each rounded up to 256 bytes, must fit the array:
 *       4 N NH F (proj) + 2 * 4 N NH (scores) + 256 (max) + 4 N outw (skip; 0 if skip_identity)
 *       + 4 FIN F (folded skip weights; concat == 0 only) + 8 FIN NH (attention weights)  <=  8 MiB,
 *     e.g. N <= 2 700 nodes at NH = 5, F = 128.  Larger graphs return TSG_E_LIMIT and launch nothing.
 *   - The first call on a device looks up that device's address of the array (hipGetSymbolAddress); make it
 *     once outside graph capture.  No call ever uses hipMalloc / hipFree or an allocating library routine.
 *   - All calls on one GPU share the array: calls that may overlap on one GPU (different streams, threads or
 *     processes' graphs replayed together) race; serialise them.  Nothing in it has to survive between calls. */
#define TSG_GAT_STATIC_SCRATCH_BYTES (8u << 20)
int tsg_gat_forward_static(const tsg_gat_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * GAT layer in TRAINING mode (row f4 of the scope table): GATLayerImp3.forward with live dropout
 * (generator/gat.py:149,155,180) and its backward, as DistanceGatFC.forward / calculate_loss run them
 * (generator/distance_gat_fc.py:44-95) and train_gan.py:284 re-encodes after every optimiser step.
 * The dropouts are caller-supplied masks, already scaled by 1 / (1 - p) (NULL = no dropout), so a run is
 * reproducible from the caller's seed and comparable with the reference's autograd under the same masks.
 * tsg_gat_train_forward: x_d = x * m_in; p_d = (x_d . W^T) * m_proj; scores, edge softmax with the global max,
 * att_d = att * m_att; aggregation + skip + head mean / concat + bias + ELU -> out.  Keeps x_d, p_d, s_src, s_trg,
 * att, att_d for the backward.
 * tsg_gat_train_backward: from d_out, the gradients of the sparse part: d_o (gradient of the per-head sums, also the
 * gradient of the skip projection's output), d_p (gradient of the projection x_d . W^T), d_a_src, d_a_trg, d_bias.
 * The dense products that remain (d_W = d_p^T . x_d, d_x_d = d_p . W, and the same for the skip projection) are
 * tsg_gemm_f32 calls of the caller (ts-trajgen_amd/generator/gat.py).  Gradient through the global max
 * (gat.py:221) is dropped: it is O(1e-16 / neighbourhood sum) of the attention gradients.
 * Edge "slots" j are the edges sorted stably by target (in_ptr / in_src as in tsg_gat_desc); m_att, att, att_d and
 * d_pre are indexed by slot.  out_ptr / out_slot list, per SOURCE node, the slots of its outgoing edges;
 * slot_trg[j] is the target of slot j.
 * ------------------------------------------------------------------------------------------ */
typedef struct tsg_gat_train {
  int32_t n_nodes, n_edges, fin, heads, fout, concat, skip_identity, reserved;
  const int32_t* in_ptr;   /* [N+1] */
  const int32_t* in_src;   /* [E] */
  const int32_t* out_ptr;  /* [N+1] (backward) */
  const int32_t* out_slot; /* [E]   (backward) */
  const int32_t* slot_trg; /* [E]   (backward) */
  const float* x;          /* [N][FIN] */
  const float* m_in;       /* [N][FIN] or NULL */
  const float* m_proj;     /* [N][NH*F] or NULL */
  const float* m_att;      /* [E][NH] by slot, or NULL */
  const float* w_proj_t;   /* [FIN][NH*F] linear_proj.weight^T */
  const float* w_skip_t;   /* [FIN][NH*F] skip_proj.weight^T (unused when skip_identity) */
  const float* a_src;      /* [NH][F] */
  const float* a_trg;      /* [NH][F] */
  const float* bias;       /* [F] or [NH*F] (concat), or NULL */
  float* x_d;              /* [N][FIN]   saved: x * m_in */
  float* p_d;              /* [N][NH*F]  saved: projection after its dropout */
  float* skip;             /* [N][NH*F]  workspace: x_d . Ws^T (unused when skip_identity) */
  float* s_src;            /* [N][NH]    saved */
  float* s_trg;            /* [N][NH]    saved */
  float* att;              /* [E][NH]    saved: attention before its dropout */
  float* att_d;            /* [E][NH]    saved: after */
  float* gmax;             /* [1] */
  float* out;              /* [N][F] or [N][NH*F] */
  const float* d_out;      /* backward in: gradient of out */
  float* d_o;              /* [N][NH*F] out: gradient of the per-head sums (= of the skip projection's output) */
  float* d_p;              /* [N][NH*F] out: gradient of x_d . W^T */
  float* d_pre;            /* [E][NH]   workspace */
  float* ds_src;           /* [N][NH]   workspace */
  float* ds_trg;           /* [N][NH]   workspace */
  float* d_a_src;          /* [NH][F] out */
  float* d_a_trg;          /* [NH][F] out */
  float* d_bias;           /* like bias, out (NULL when bias is NULL) */
  size_t in_bytes;         /* of x_d */
  size_t node_feat_bytes;  /* of EACH of p_d, skip, d_o, d_p */
  size_t score_bytes;      /* of EACH of s_src, s_trg, ds_src, ds_trg */
  size_t edge_bytes;       /* of EACH of att, att_d, d_pre */
  size_t out_bytes;        /* of out */
} tsg_gat_train;

int tsg_gat_train_forward(const tsg_gat_train* d, void* stream);
int tsg_gat_train_backward(const tsg_gat_train* d, void* stream);

/* GAT input features: x[n] = [lon_embed[nf[n][-2]], lat_embed[nf[n][-1]], nf[n][:-2]]
 * (generator/distance_gat_fc.py:191-197).  node_features [N][fraw] f32 (last two columns hold the
 * integer lon/lat grid cells), lon_embed [n_lon][gps], lat_embed [n_lat][gps], x [N][2*gps + fraw - 2].
 * Out-of-range grid cells set *err_flag (device int32) to 1 and read row 0. */
int tsg_gat_build_input(const float* node_features, int n_nodes, int fraw, const float* lon_embed, int n_lon,
                        const float* lat_embed, int n_lat, int gps_dim, float* x, int32_t* err_flag, void* stream);

/* ------------------------------------------------------------------------------------------
 * Model tables shared by the scoring entry points (function G + function H of
 * generator/generator_v4.py:31-72, pre-factored on the host by ts-trajgen_amd/engine.py).
 * Hp = hidden size padded to a multiple of 64 (padding units have zero weights).
 * Gate columns of the weights (whh0_t, wl_t, bl) are interleaved: col = (u/32)*128 + gate*32 + u%32, gates in
 * torch order i,f,g,o (one 128-column matrix-core tile holds the four gates of 32 units); the additive tables
 * xproj / tproj keep the four gates of a unit side by side instead, col = 4*u + gate (one 16-byte read per unit).
 * ------------------------------------------------------------------------------------------ */
typedef struct tsg_model {
  int32_t n_roads;   /* rows of out_w / pfrom / pto */
  int32_t n_time;    /* rows of tproj */
  int32_t hp;        /* padded hidden size, multiple of 64, <= 512 */
  int32_t n_layers;  /* LSTM layers, 1..TSG_MAX_LAYERS */
  const float* xproj;                /* [n_roads+1][Hp][4] road_emb . W_ih_l0[:, :Er]^T, 16-byte aligned */
  const float* tproj;                /* [n_time][Hp][4]   time_emb . W_ih_l0[:, Er:]^T + b_ih_l0 + b_hh_l0, 16-byte aligned */
  const float* whh0_t;               /* [Hp][4Hp]       W_hh_l0^T */
  const float* wl_t[TSG_MAX_LAYERS]; /* l>=1: [2Hp][4Hp] = [W_ih_l^T ; W_hh_l^T]; entry 0 unused */
  const float* bl[TSG_MAX_LAYERS];   /* l>=1: [4Hp] b_ih_l + b_hh_l; entry 0 unused */
  const float* out_w;                /* [n_roads][Hp] function_g.out_fc.weight */
  const float* out_b;                /* [n_roads] */
  const float* pfrom;                /* [n_roads][10]  node_emb . emb_fc.weight[:, :F]^T */
  const float* pto;                  /* [n_roads][10]  node_emb . emb_fc.weight[:, F:]^T + emb_fc.bias */
  const float* hparams;              /* [42]: dis_fc.w[10], dis_fc.b[10], out_fc.w[20], out_fc.b, w1 */
  /* Split-f16 images of whh0_t / wl_t for tsg_engine.state_format == 1 (may be null otherwise): every weight w
   * as major = f16(w), minor = f16((w - major) * 2^11), tiled as [4Hp/128][K/32][major|minor][128 cols][32 k]
   * f16 (the LDS image of one k-tile of one column tile is contiguous). */
  const void* whh0_s;
  const void* wl_s[TSG_MAX_LAYERS];
} tsg_model;

/* ------------------------------------------------------------------------------------------
 * One LSTM layer step for a batch of rows (the per-expansion recurrent cell).  Replaces the
 * `nn.LSTM` call of function_g_fc.py:169, incrementally: row i continues from the state rows
 * prev_idx[i] and writes state rows out_idx[i] (row indices into the [rows][Hp] state arrays;
 * prev_idx < 0 = zero initial state).  n_rows_dev (device int) may be NULL -> n_rows_host.
 * layer 0 adds xproj[rid[i]] + tproj[tcode[i]]; layer l>=1 reads h_below rows out_idx[i].
 * ------------------------------------------------------------------------------------------ */
typedef struct tsg_lstm_step {
  int32_t layer;
  int32_t n_rows_host;
  const int32_t* n_rows_dev;
  const int64_t* prev_idx; /* [n] */
  const int64_t* out_idx;  /* [n] */
  const int32_t* rid;      /* [n] (layer 0) */
  const int32_t* tcode;    /* [n] (layer 0) */
  const float* h_below;    /* [rows][Hp] (layer >= 1): new hidden of layer-1 */
  float* h;                /* [rows][Hp] this layer's hidden states (read prev, write out) */
  float* c;                /* [rows][Hp] this layer's cell states */
  size_t state_rows;       /* rows in h / c: a row whose out_idx is outside [0, state_rows) is skipped, a prev_idx
                              outside it counts as "no previous state" (checked on the device; 0 = unchecked) */
} tsg_lstm_step;

int tsg_lstm_step_batched(const tsg_model* m, const tsg_lstm_step* s, void* stream);

/* ------------------------------------------------------------------------------------------
 * Candidate scoring for explicit (prefix, candidates, distances) rows: the drop-in for
 * GeneratorV4.predict (generator/generator_v4.py:56-72) == softmax((1-w1) softmax(G) + w1 softmax(H)).
 * Row i attends over the top-layer hidden rows chain[chain_ptr[i] .. chain_ptr[i+1]) (last = query; a row may be
 * listed more than once, e.g. an all-zero row standing for the padded positions of a masked batch, which the
 * reference keeps in the attention softmax, function_g_fc.py:31-35),
 * scores candidates cand[cand_ptr[i] .. cand_ptr[i+1]) with distances dis[...] towards des[i] and
 * writes prob[...] (and, if non-NULL, the separate G and H softmaxes).  <= TSG_MAX_DEGREE candidates.
 * ------------------------------------------------------------------------------------------ */
typedef struct tsg_score_batch {
  int32_t n_rows;
  const int32_t* chain_ptr; /* [n_rows+1] */
  const int64_t* chain;     /* state-row indices, oldest first */
  const float* h_top;       /* [rows][Hp] top-layer hidden states */
  const int32_t* cand_ptr;  /* [n_rows+1] */
  const int32_t* cand;      /* candidate road ids */
  const float* dis;         /* candidate distances (units of 100 m, search.py:355-356) */
  const int32_t* des;       /* [n_rows] */
  float* prob;              /* out, same layout as cand */
  float* prob_g;            /* out or NULL */
  float* prob_h;            /* out or NULL */
  float* logit_g;           /* out or NULL: function G's candidate scores BEFORE its softmax, out_fc(attn)[cand]
                               (FunctionGFC.forward, function_g_fc.py:184-187) */
  float* logit_h;           /* out or NULL: function H's scores before its softmax (distance_gat_fc.py:129-134) */
} tsg_score_batch;

int tsg_score_candidates(const tsg_model* m, const tsg_score_batch* b, void* stream);

/* ------------------------------------------------------------------------------------------
 * Batched A* engine.  Replaces Searcher.astar_search (search.py:240-424) and
 * Searcher.astar_search_in_region (search.py:426-612) for MANY OD pairs at once; the call
 * our_model_generate.py:417-426 makes one OD at a time.  `slots` searches are in flight; a
 * finished slot pulls the next pending OD.  One superstep = frontier kernel (relax the previous
 * expansion with heapq-exact ordering, pop the next node, goal/closed/region checks, time update
 * search.py:376-384) -> LSTM layer kernels -> scoring kernel (attention over the ancestor chain,
 * gathered out_fc rows, factored H, on-device WGS-84 geodesic, double softmax, log).
 * ------------------------------------------------------------------------------------------ */
typedef struct tsg_graph {
  int32_t n_roads;
  int32_t max_degree;
  const int32_t* indptr;  /* [n_roads+1] successors CSR, order of adjacent_list preserved */
  const int32_t* indices; /* [E] */
  const double* geo;      /* [n_roads][4] {lon_deg, lat_deg, sinU, cosU}, U = atan((1-f) tan(lat)): road_center_gps
                             with the reduced-latitude terms of the WGS-84 inverse geodesic precomputed (fp64) */
  const int32_t* cost;    /* [24][n_roads] seconds: rtd if != 1 else int(round(len/8.334)) (search.py:376-381) */
  const int32_t* label;   /* [n_roads] region label per road, or NULL (no in-region gating) */
  /* region-level graphs (DoubleLayerSearcher.astar_search_in_region, search.py:936-1117); NULL for roads: */
  const double* dist_table;   /* [n][n] metres: candidate distance = dist_table[c][des] / 100 (search.py:1049) */
  const int32_t* pair_cost;   /* [n][n] seconds int(round(dist[now][c] / 8.334)), used where cost[h][now] < 0 */
  const uint8_t* empty_key;   /* [n] 1 = node has a key with no successors: closed without a step (:993-995) */
} tsg_graph;

typedef struct tsg_od_batch {
  int32_t n_od;
  int32_t lmax;                /* row length of out_rid / out_time */
  const int32_t* prefix_ptr;   /* [n_od+1] */
  const int32_t* prefix_rid;   /* known trajectory prefix (trace_loc), >= 1 element per OD */
  const int32_t* prefix_time;  /* trace_tim */
  const int32_t* des;          /* [n_od] */
  const int32_t* default_len;  /* [n_od] */
  const int32_t* max_step;     /* [n_od] */
  const int32_t* allow_label;  /* [n_od] label the expansion is confined to, <0 = unrestricted; or NULL */
  int32_t* out_rid;            /* [n_od][lmax] */
  int32_t* out_time;           /* [n_od][lmax] */
  int32_t* out_len;            /* [n_od] true length (may exceed lmax -> OVERFLOW bit) */
  int32_t* out_status;         /* [n_od] TSG_STATUS_* */
  int32_t* out_steps;          /* [n_od] expansions (`step`, search.py:420) */
  int32_t poplog_cap;          /* entries per OD in the pop log, 0 = off */
  int32_t* poplog_rid;         /* [n_od][poplog_cap] rid of every non-stale pop, in order */
  float* poplog_prio;          /* [n_od][poplog_cap] its priority (-log_prob) */
  int32_t* poplog_n;           /* [n_od] */
} tsg_od_batch;

typedef struct tsg_engine {
  int32_t slots;     /* B: searches in flight */
  int32_t exp_cap;   /* expanded nodes per slot  >= max prefix + max max_step */
  int32_t node_cap;  /* discovered nodes per slot (<= 2^20 - 1) */
  int32_t heap_cap;  /* 8-byte words of a slot's queue stretch, a multiple of 16 (>= 32).  The queue is a blocked heap for
                        its first twelve levels (entries 1..4095): three levels of CPython's binary heap per 64-byte
                        block, blocks numbered like an 8-ary heap, so a sift touches one cache line per three levels;
                        entries beyond lie behind the blocks in plain heap order.  A stretch for n entries (n = 2 +
                        max_step * max_degree covers every search) has 4704 + (n - 4095) words for n > 4095, else
                        8 * (blocks + 2) with blocks = (8^L - 1) / 7 + (d == 3L ? n - 8^L + 1 : 8^L), d = floor(log2 n),
                        L = d / 3 (ts-trajgen_amd/engine.py::heap_words); the heap base must be 128-byte aligned */
  /* workspaces (caller-allocated); *_bytes are checked against the dims above */
  int32_t* slot_state;  size_t slot_state_bytes;  /* [B][40] int32 */
  uint64_t* heap;       size_t heap_bytes;        /* [B][heap_cap] {float prio; int32 node} */
  uint64_t* table;      size_t table_bytes;       /* [B][n_roads] rid -> low word: generation << 20 | node index (generation
                                                     0 = never seen, so tsg_engine_reset zeroes it once per OD batch); high
                                                     word: the node's current log_prob as f32 while it is open (rid2node),
                                                     +inf once it is closed (close_set), so relaxation and the
                                                     futile-expansion check need no node record */
  int32_t* nodes;       size_t nodes_bytes;       /* [B][node_cap][8] int32 {rid, parent, log_prob, date_time, state,
                                                     indptr[rid], out-degree, depth} */
  int32_t* exp_meta;    size_t exp_meta_bytes;    /* [B][exp_cap][4] {parent, rid, tcode, depth} */
  int32_t* exp_up;      size_t exp_up_bytes;      /* [n_pages*64][64]: exp_up[row][i] = state row of the (i+1)-th ancestor
                                                     of the node in state row `row`, -1 = none */
  int32_t up_levels;    float prune_spread;       /* |1-w1| + |w1| of the model (0 = only the all-closed rule): bound used to
                                                     skip expansions that cannot push anything, see tsg_search.hip */
  /* up_levels must be TSG_UP_LEVELS */
  int32_t state_format; int32_t max_pops;         /* state_format 0: h[l] holds fp32 and the recurrence runs on the fp32 matrix-core
                                                     path; 1: h[l] holds one 32-bit word per element, low half
                                                     major = f16(x), high half minor = f16((x - major) * 2^11), and the
                                                     recurrence runs as three f16 matrix-core passes with fp32
                                                     accumulation (products exact to 2^-22; see tsg_dense.hip).
                                                     max_pops: a search that has popped this many queue entries in one
                                                     superstep without finding a node to expand yields until the next
                                                     superstep (0 = no limit); bounds the launch time by the typical,
                                                     not the worst, search; results do not depend on it */
  float* h[TSG_MAX_LAYERS];                       /* [n_pages*64][Hp] per layer */
  float* c[TSG_MAX_LAYERS];
  size_t state_bytes;                             /* bytes of EACH h[l] / c[l] */
  /* State pages.  The recurrent state, and the ancestor list, of an expanded node live in a state ROW; rows
   * come in pages of 64 from a pool shared by all searches (a search that ends after 300 expansions holds 5
   * pages, not the worst case of exp_cap rows), so many more searches fit in HBM than exp_cap * slots rows
   * would allow.  n_pages == slots * pages_per_slot reserves the worst case (then nothing below can trigger).
   * With a smaller pool: new searches are admitted only out of the surplus above admit_reserve free pages; a
   * search that finds no free page waits (emits no work item) until pages are returned; if every search that
   * wants to run is waiting, the youngest ones (fewest pages) give their pages back and their OD pair is
   * queued again (`retry`), restarting from scratch later: results do not depend on any of this. */
  int32_t n_pages;        int32_t pages_per_slot; /* pool size; ceil(exp_cap / 64) */
  int32_t admit_reserve;  int32_t ramp_steps;     /* slot b starts its first search at superstep b * ramp_steps / slots:
                                                     spreads the ages of the first generation of searches */
  int32_t* slot_pages;  size_t slot_pages_bytes;  /* [B][pages_per_slot] page ids of each search, in order */
  int32_t* page_ring;   size_t page_ring_bytes;   /* [n_pages] free ring */
  int32_t* retry;       size_t retry_bytes;       /* [>= n_od + B] ring of OD indices waiting to restart */
  int32_t* page_flags;  size_t page_flags_bytes;  /* [2][256][32] progress / wait stamps of the superstep, spread over
                                                     256 cache lines each (gathered by the control step) */
  int32_t* items;       size_t items_bytes;       /* [R * 4096][12], R = ceil(B / 4096): the work items of the current
                                                     superstep, a compacted list.  The searches of slots [4096 r,
                                                     4096 r + 4096) that emit an item write it into stretch r: items
                                                     with 16 or more rows on their ancestor chain from the front, the
                                                     others from the back; the distance, recurrence and scoring kernels
                                                     walk the filled positions only */
  int32_t* item_counts; size_t item_counts_bytes; /* [2][R][32]: n_long, n_short of each stretch (one 128-byte line each),
                                                     one plane per superstep parity */
  float* cand_lp;       size_t cand_lp_bytes;     /* [B][max_degree] */
  int32_t* cand_id;     size_t cand_id_bytes;     /* [B][max_degree] successors of the expansion the search waits for (left by
                                                     the frontier kernel for the distance and scoring kernels and for its own
                                                     relaxation in the next superstep) */
  int32_t* row_meta;    size_t row_meta_bytes;    /* [n_pages*64][2] {rid, time code} of every state row: with exp_up, a finished
                                                     search's trajectory is written 64 points per read */
  int32_t* shared_counters;                       /* [16] words 2,3,4,6 below; == counters for a lone engine.  Several
                                                     engines (disjoint workspaces, one HIP stream each) that work on
                                                     the SAME tsg_od_batch share this array: one OD queue */
  int32_t* counters;    size_t counters_bytes;    /* [16] state-page control of THIS engine: 0 ring head, 1 ring tail,
                                                     2 tail snapshot, 3-5 the same for `retry`, 6 waits, 7 restarts,
                                                     8 restart level, 9 plane of item_counts just filled, 10 reserved, 11 superstep,
                                                     12 fewest free pages seen, 13 admission quota, 14 restart budget.
                                                     shared_counters: 2 next OD, 3 finished ODs, 4 total expansions,
                                                     6 error flag.  Per-slot statistics live in slot_state words
                                                     16..21: model evaluations, scored candidates, attention rows, work
                                                     items, skipped expansions, waits for a page */
} tsg_engine;

/* Reset slot states, the rid table, the page ring and `counters` for a new OD batch (enqueues memsets and one
 * small kernel).  The caller zeroes `shared_counters` itself when it is a separate array. */
int tsg_engine_reset(const tsg_engine* e, const tsg_graph* g, void* stream);

/* Function H on explicit pairs: out[i] = out_fc(cat[relu(emb_fc(cat[emb[from[i]], emb[to[i]]])), relu(dis_fc(dis[i]))]),
 * the score BEFORE the softmax (DistanceGatFC.predict_cache, distance_gat_fc.py:118-134; DistanceGatFC.forward,
 * :44-74, for a module in inference mode), from the factored tables pfrom / pto / hparams of tsg_model.
 * from[i] < n_roads and to[i] < rows of pto are the caller's to guarantee (the Python wrapper checks them). */
int tsg_h_pair_scores(const tsg_model* m, int32_t n, const int32_t* from, const int32_t* to, const float* dis,
                      float* out, void* stream);

/* The candidate distance on explicit pairs: out_m[i] = WGS-84 inverse geodesic (Vincenty, fp64, metres) between
 * rows a[i] and b[i] of `geo` ([n_rows][4] as tsg_graph.geo); -1 for an index outside the table.  Replaces
 * geopy.distance.distance((lat, lon), (lat, lon)) at search.py:355-356 (the engine feeds the model
 * kilometres * 10 = out_m / 100 as f32); the engine computes the same function inside its supersteps, this
 * entry point exposes it for callers of GeneratorV4.predict and for the parity test. */
int tsg_geodesic_pairs(const double* geo, int32_t n_rows, int32_t n, const int32_t* a, const int32_t* b,
                       double* out_m, void* stream);

/* Trajectories out of the expansion records of the searches (the path the reference carries in every SearchNode,
 * search.py:14-52: trace_loc / trace_tim, and returns at :291-296 / :421-424).  Request i names a search slot and
 * one of its expanded nodes (index into the slot's exp_meta); with last_rid[i] >= 0 one more point (last_rid[i],
 * last_tcode[i]) is appended behind it (a successor that was not expanded, e.g. the destination).  Writes
 * out_rid / out_time [n][lmax] (positions >= lmax are dropped) and out_len[n] (0 for a request out of range).
 * tsg_engine_run does the same for every search that ends; this entry point serves searches still in flight and
 * finished slots whose records have not been reused.  last_rid / last_tcode may be NULL. */
int tsg_reconstruct_paths(const tsg_engine* e, int32_t n, const int32_t* slot, const int32_t* node,
                          const int32_t* last_rid, const int32_t* last_tcode, int32_t lmax, int32_t* out_rid,
                          int32_t* out_time, int32_t* out_len, void* stream);

/* Enqueue `n_supersteps` supersteps.  `first_superstep` is the index of the first one (parity
 * selects the ping-pong item counter).  Poll counters[3] == n_od for completion. */
int tsg_engine_run(const tsg_model* m, const tsg_graph* g, const tsg_od_batch* od, const tsg_engine* e,
                   int32_t first_superstep, int32_t n_supersteps, void* stream);

/* HIP-graph form of tsg_engine_run (no counterpart in the reference): capture the launches of `n_supersteps`
 * supersteps once from `stream` (not the legacy default stream), replay them with one graph launch per
 * chunk.  The graph holds the pointer values of m / g / od / e at capture time: create one per OD batch. */
int tsg_engine_graph_create(const tsg_model* m, const tsg_graph* g, const tsg_od_batch* od, const tsg_engine* e,
                            int32_t n_supersteps, void* stream, void** exec_out);
int tsg_engine_graph_launch(void* exec, void* stream);
int tsg_engine_graph_destroy(void* exec);

/* Measurement variant of tsg_engine_run: identical launches, each bracketed by HIP events on `stream`.
 * SYNCHRONISES the stream before returning.  Adds per-kernel elapsed ms and launch counts to
 * ms_sum_host / launches_host, both HOST arrays of 2 + TSG_MAX_LAYERS entries:
 * [0] frontier kernel, [1] scoring kernel, [2 + l] LSTM layer l.  (No counterpart in the reference.) */
int tsg_engine_run_profiled(const tsg_model* m, const tsg_graph* g, const tsg_od_batch* od, const tsg_engine* e,
                            int32_t first_superstep, int32_t n_supersteps, void* stream, float* ms_sum_host,
                            int32_t* launches_host);

/* One superstep in two calls (tests, profiling, callers that interleave work of their own):
 * tsg_frontier_expand_select = the frontier kernel and the page-pool control step (search.py:279-300 pop loop,
 * :385-419 relaxation of the previous expansion); tsg_score_frontier = everything that evaluates the model on the
 * work items it left (candidate distances :352-357, the recurrent cell, GeneratorV4.predict :359-373).  Calling
 * the pair in a loop is tsg_engine_run. */
int tsg_frontier_expand_select(const tsg_graph* g, const tsg_od_batch* od, const tsg_engine* e, int32_t superstep,
                               void* stream);
int tsg_score_frontier(const tsg_model* m, const tsg_graph* g, const tsg_engine* e, int32_t superstep,
                       void* stream);

/* ---- stochastic rollouts --------------------------------------------------------------------------------
 * Replaces the per-trajectory loops of Searcher.road_random_sample (search.py:614-716) and, with a region
 * graph (tsg_graph.dist_table set), DoubleLayerSearcher.region_random_sample (search.py:1163-1256): every
 * row is one (prefix, destination, default_len) rollout; one step appends at most one point per row.
 * The reference's `multinomial(1)` (search.py:686,1233) is an inverse-CDF draw here: the k-th model
 * evaluation of row b takes uniform[b][k] and picks the first candidate i with u < f_0 + ... + f_i
 * (fp32, candidate order), else the last.  All arrays are device memory owned by the caller. */
typedef struct tsg_rollout {
  int32_t n_traj, lmax;       /* rows; row width of out_rid / out_time and state rows per trajectory */
  int32_t n_draws, reserved;  /* columns of `uniform` (>= max(20, default_len) covers every row) */
  int32_t* out_rid;           /* [n_traj][lmax] prefix pre-filled; grows in place */
  int32_t* out_time;          /* [n_traj][lmax] time codes, prefix pre-filled */
  int32_t* len;               /* [n_traj] in: prefix length >= 1; out: returned length (after the
                                 trace[:default_len] cut of search.py:716 when des was not reached) */
  int32_t* n_state;           /* [n_traj] zero at start: tokens already run through the LSTM */
  int32_t* n_drawn;           /* [n_traj] zero at start: uniforms consumed */
  int32_t* status;            /* [n_traj] zero at start: 0 running, TSG_STATUS_REACHED, TSG_STATUS_FALLBACK
                                 (budget max(20, default_len) spent or dead end), | TSG_STATUS_OVERFLOW when lmax or
                                 n_draws ran out */
  const int32_t* des;         /* [n_traj] */
  const int32_t* default_len; /* [n_traj] */
  const int32_t* start_code;  /* [n_traj] time code of the last prefix point (search.py:632) */
  const float* uniform;       /* [n_traj][n_draws] in [0, 1) */
  float* prob_log;            /* optional [n_traj][n_draws][max_degree]: the probability vector of every draw */
  float* h[TSG_MAX_LAYERS];   /* [n_traj * lmax][hp] per layer; row b * lmax + i = state after point i of row b */
  float* c[TSG_MAX_LAYERS];
  size_t state_bytes;         /* bytes of each h[l] / c[l] */
  int64_t* prev_idx;          /* [n_traj] work rows of the LSTM step */
  int64_t* out_idx;           /* [n_traj] */
  int32_t* rid;               /* [n_traj] */
  int32_t* tcode;             /* [n_traj] */
} tsg_rollout;

/* Enqueue `n_steps` rollout steps.  A row with a P-point prefix needs P - 1 warm-up steps, then one step per
 * appended point, then one to notice that it is finished: P + max(20, default_len) steps finish every row. */
int tsg_rollout_run(const tsg_model* m, const tsg_graph* g, const tsg_rollout* r, int32_t n_steps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TSG_H_ */
