// Batched stochastic rollouts: Searcher.road_random_sample (search.py:614-716) and
// DoubleLayerSearcher.region_random_sample (search.py:1163-1256) for many (prefix, destination) rows at once.
// One rollout step = feed kernel (which token each trajectory runs through the LSTM) -> the LSTM layer
// kernels of tsg_dense.hip -> step kernel (one wave64 per trajectory: stop tests, candidate scoring,
// inverse-CDF draw, append).  The reference's multinomial(1) becomes an inverse-CDF draw from caller-supplied
// uniforms so that a rollout is reproducible on any device.
#include "tsg_engine_dev.h"

struct RolloutDev {
  tsg_rollout r;
  tsg_graph g;
};

// thread per trajectory: the next unprocessed token (if any) becomes this step's LSTM row
__global__ __launch_bounds__(256) void rollout_feed_kernel(RolloutDev D) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= D.r.n_traj) return;
  long long pv = -1, ov = -1;
  int rd = 0, tc = 0;
  const int ns = D.r.n_state[b];
  if (D.r.status[b] == 0 && ns < D.r.len[b]) {
    const long long base = (long long)b * D.r.lmax;
    pv = ns > 0 ? base + ns - 1 : -1;
    ov = base + ns;
    rd = D.r.out_rid[base + ns];
    tc = D.r.out_time[base + ns];
    D.r.n_state[b] = ns + 1;
  }
  D.r.prev_idx[b] = pv;
  D.r.out_idx[b] = ov;
  D.r.rid[b] = rd;
  D.r.tcode[b] = tc;
}

// wave per trajectory: one iteration of the while loop at search.py:640-711 / :1189-1251
template <int JT>
__global__ __launch_bounds__(256) void rollout_step_kernel(tsg_model m, RolloutDev D) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= D.r.n_traj) return;
  if (D.r.status[b] != 0) return;
  const int len = D.r.len[b];
  if (D.r.n_state[b] != len) return;  // still running the prefix through the LSTM
  const size_t base = (size_t)b * D.r.lmax;
  int32_t* rid_row = D.r.out_rid + base;
  int32_t* tim_row = D.r.out_time + base;
  const int now = rid_row[len - 1], des = D.r.des[b];
  const int dl = D.r.default_len[b];
  const int max_step = dl > 20 ? dl : 20;  // search.py:636
  const int N = D.g.n_roads;
  if (now == des) {  // loop condition :640; found -> full trace (:712-713)
    if (lane == 0) D.r.status[b] = TSG_STATUS_REACHED;
    return;
  }
  const int p0 = D.g.indptr[now];
  const int C = D.g.indptr[now + 1] - p0;
  int stop = 0;
  if (len - 1 >= max_step || C == 0) stop = TSG_STATUS_FALLBACK;       // step budget / dead end (:640,:646-648,:708-710)
  else if (len >= D.r.lmax) stop = TSG_STATUS_FALLBACK | TSG_STATUS_OVERFLOW;
  int idx = 0;
  if (!stop && C > 1) {
    const int nd = D.r.n_drawn[b];
    if (nd >= D.r.n_draws) {
      stop = TSG_STATUS_FALLBACK | TSG_STATUS_OVERFLOW;
    } else {
      const int hp = m.hp, J = hp / 64;
      const float* htop = D.r.h[m.n_layers - 1] + base * hp;
      float q[JT];
      load_row<JT>(htop + (size_t)(len - 1) * hp, q, J, lane);
      AttnAcc<JT> acc;
      attn_init(acc);
      for (int i = len - 1; i >= 0; i -= 4) {
        const float* rows[4];
        unsigned present = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          rows[u] = (i - u >= 0) ? htop + (size_t)(i - u) * hp : nullptr;
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
      int my_cand = 0;
      float my_dis = 0.0f;
      if (lane < C) {
        my_cand = D.g.indices[p0 + lane];
        if (D.g.dist_table) {  // search.py:1220
          my_dis = (float)(D.g.dist_table[(size_t)my_cand * N + des] / 100.0);
        } else {  // search.py:668-672
          const double dm = tsg_dev_geodesic_m(D.g.geo + 4 * (size_t)des, D.g.geo + 4 * (size_t)my_cand);
          my_dis = (float)(dm / 1000.0 * 10.0);
        }
      }
      const float f = score_row<JT>(m, pooled, J, lane, C, my_cand, my_dis, des, nullptr, nullptr);
      if (D.r.prob_log && lane < C) D.r.prob_log[((size_t)b * D.r.n_draws + nd) * D.g.max_degree + lane] = f;
      // inverse CDF in candidate order: first i with u < f_0 + ... + f_i (fp32, sequential), else the last
      const float u = D.r.uniform[(size_t)b * D.r.n_draws + nd];
      float cum = 0.0f;
      idx = C - 1;
      bool found = false;
      for (int i = 0; i < C; ++i) {
        cum += __shfl(f, i, TSG_WAVE);
        if (!found && u < cum) {
          idx = i;
          found = true;
        }
      }
      if (lane == 0) D.r.n_drawn[b] = nd + 1;
    }
  }
  if (stop) {  // not found: the first default_len points (:714-716)
    if (lane == 0) {
      D.r.status[b] = stop;
      D.r.len[b] = len < dl ? len : (dl > 0 ? dl : 0);
    }
    return;
  }
  const int next = D.g.indices[p0 + idx];
  // time: start time + the cost of the current road; the reference never advances now_date_time (:638,:661,:703)
  const int sc = D.r.start_code[b];
  const int woff = sc >= 1440 ? 1440 : 0;
  const int dt0 = (sc - woff) * 60;
  const int hour = dt0 / 3600;
  int cost;
  if (D.g.dist_table) {  // regions: :1207-1211 (single successor: the NEXT region's entry), :1238-1242 (current region's)
    const int who = C == 1 ? next : now;
    cost = D.g.cost[(size_t)hour * N + who];
    if (cost < 0) cost = D.g.pair_cost[(size_t)now * N + next];
  } else {
    cost = D.g.cost[(size_t)hour * N + now];
  }
  const int next_time = (dt0 + cost) % 86400;
  if (lane == 0) {
    rid_row[len] = next;
    tim_row[len] = next_time / 60 + woff;
    D.r.len[b] = len + 1;
  }
}

extern "C" int tsg_rollout_run(const tsg_model* m, const tsg_graph* g, const tsg_rollout* r, int n_steps,
                               void* stream) {
  if (!m || !g || !r) return TSG_E_INVALID;
  if (m->hp <= 0 || (m->hp % 64) != 0 || m->hp > 512) return TSG_E_INVALID;
  if (m->n_layers < 1 || m->n_layers > TSG_MAX_LAYERS) return TSG_E_INVALID;
  if (!m->out_w || !m->out_b || !m->pfrom || !m->pto || !m->hparams) return TSG_E_INVALID;
  if (r->n_traj < 0 || r->lmax < 1 || r->n_draws < 0 || n_steps < 0) return TSG_E_INVALID;
  if (g->n_roads <= 0 || g->max_degree < 1 || g->max_degree > TSG_MAX_DEGREE) return TSG_E_INVALID;
  if (!g->indptr || !g->indices || !g->cost) return TSG_E_INVALID;
  if (!g->geo && !g->dist_table) return TSG_E_INVALID;
  if (g->dist_table && !g->pair_cost) return TSG_E_INVALID;
  if (!r->out_rid || !r->out_time || !r->len || !r->n_state || !r->n_drawn || !r->status || !r->des ||
      !r->default_len || !r->start_code || !r->prev_idx || !r->out_idx || !r->rid || !r->tcode)
    return TSG_E_INVALID;
  if (r->n_draws > 0 && !r->uniform) return TSG_E_INVALID;
  const size_t need = (size_t)r->n_traj * r->lmax * m->hp * sizeof(float);
  if (r->state_bytes < need) return TSG_E_TOO_SMALL;
  for (int l = 0; l < m->n_layers; ++l)
    if (!r->h[l] || !r->c[l]) return TSG_E_INVALID;
  if (r->n_traj == 0 || n_steps == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  RolloutDev D;
  D.r = *r;
  D.g = *g;
  const int J = m->hp / 64;
  const int jt = J <= 1 ? 1 : J <= 2 ? 2 : J <= 4 ? 4 : 8;
  const dim3 blk(256), grid_t(tsg_cdiv(r->n_traj, 256)), grid_w(tsg_cdiv(r->n_traj, 4));
  for (int t = 0; t < n_steps; ++t) {
    hipLaunchKernelGGL(rollout_feed_kernel, grid_t, blk, 0, s, D);
    for (int l = 0; l < m->n_layers; ++l) {
      const int rc = tsg_launch_lstm_layer(m, l, r->n_traj, nullptr, r->n_traj, r->prev_idx, r->out_idx, r->rid,
                                           r->tcode, nullptr, l > 0 ? r->h[l - 1] : nullptr, r->h[l], r->c[l], s, 0, nullptr);
      if (rc) return rc;
    }
    switch (jt) {
      case 1: hipLaunchKernelGGL(rollout_step_kernel<1>, grid_w, blk, 0, s, *m, D); break;
      case 2: hipLaunchKernelGGL(rollout_step_kernel<2>, grid_w, blk, 0, s, *m, D); break;
      case 4: hipLaunchKernelGGL(rollout_step_kernel<4>, grid_w, blk, 0, s, *m, D); break;
      default: hipLaunchKernelGGL(rollout_step_kernel<8>, grid_w, blk, 0, s, *m, D); break;
    }
    TSG_LAUNCH_CHECK();
  }
  return 0;
}
