// Device-side building blocks of the scoring kernels (one wave64 per scored row).
#pragma once
#include "tsg_common.h"

// WGS-84 inverse geodesic (Vincenty), fp64, metres.  Same iteration as oracle/geodesic.py; stands
// in for geopy.distance.distance(...) at search.py:355-356.  The per-road reduced-latitude terms
// sinU = sin(atan((1-f) tan(lat))), cosU are precomputed on the host (tsg_graph.geo rows are
// {lon_deg, lat_deg, sinU, cosU}).
__device__ inline double tsg_dev_geodesic_m(const double* __restrict__ p1, const double* __restrict__ p2) {
  const double a = 6378137.0;
  const double f = 1.0 / 298.257223563;
  const double b = a * (1.0 - f);
  const double lon1 = p1[0], lat1 = p1[1], sinU1 = p1[2], cosU1 = p1[3];
  const double lon2 = p2[0], lat2 = p2[1], sinU2 = p2[2], cosU2 = p2[3];
  if (lat1 == lat2 && lon1 == lon2) return 0.0;
  const double rad = 3.14159265358979323846 / 180.0;
  const double L = (lon2 - lon1) * rad;
  double lam = L, sin_sigma = 0.0, cos_sigma = 0.0, sigma = 0.0, cos_sq_alpha = 0.0, cos2sm = 0.0;
  for (int it = 0; it < 20; ++it) {
    double sin_lam, cos_lam;
    sincos(lam, &sin_lam, &cos_lam);
    const double t1 = cosU2 * sin_lam;
    const double t2 = cosU1 * sinU2 - sinU1 * cosU2 * cos_lam;
    sin_sigma = sqrt(t1 * t1 + t2 * t2);
    if (sin_sigma == 0.0) return 0.0;
    cos_sigma = sinU1 * sinU2 + cosU1 * cosU2 * cos_lam;
    sigma = atan2(sin_sigma, cos_sigma);
    const double sin_alpha = cosU1 * cosU2 * sin_lam / sin_sigma;
    cos_sq_alpha = 1.0 - sin_alpha * sin_alpha;
    cos2sm = (cos_sq_alpha != 0.0) ? cos_sigma - 2.0 * sinU1 * sinU2 / cos_sq_alpha : 0.0;
    const double C = f / 16.0 * cos_sq_alpha * (4.0 + f * (4.0 - 3.0 * cos_sq_alpha));
    const double lam_new =
        L + (1.0 - C) * f * sin_alpha * (sigma + C * sin_sigma * (cos2sm + C * cos_sigma * (-1.0 + 2.0 * cos2sm * cos2sm)));
    const double delta = fabs(lam_new - lam);
    lam = lam_new;
    if (delta < 1e-12) break;
  }
  const double u_sq = cos_sq_alpha * (a * a - b * b) / (b * b);
  const double A = 1.0 + u_sq / 16384.0 * (4096.0 + u_sq * (-768.0 + u_sq * (320.0 - 175.0 * u_sq)));
  const double B = u_sq / 1024.0 * (256.0 + u_sq * (-128.0 + u_sq * (74.0 - 47.0 * u_sq)));
  const double d_sigma =
      B * sin_sigma *
      (cos2sm + B / 4.0 * (cos_sigma * (-1.0 + 2.0 * cos2sm * cos2sm) -
                           B / 6.0 * cos2sm * (-3.0 + 4.0 * sin_sigma * sin_sigma) * (-3.0 + 4.0 * cos2sm * cos2sm)));
  return b * A * (sigma - d_sigma);
}

#define TSG_MAXJ 8  // hp <= 512 -> <= 8 values per lane

// A state row (hp floats) spread over the wave.  JT = compile-time register count per lane.
// hp == 256 (JT == 4, J == 4): lane owns elements 4*lane .. 4*lane+3 (one dwordx4 per row);
// otherwise lane owns elements lane + 64*j.  Every vector that meets in a dot product uses the same map.
// SPLIT rows hold the engine's state words (tsg_engine.state_format == 1): low half major = f16(x), high half
// minor = f16((x - major) * 2^11); the value read back is major + minor * 2^-11.
typedef _Float16 tsg_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float tsg_state_value(float word_as_float) {
  // one v_dot2_f32_f16: major * 1 + minor * 2^-11 with fp32 accumulation (both products are exact)
  const tsg_h2 w = __builtin_bit_cast(tsg_h2, word_as_float);
  const tsg_h2 k = {(_Float16)1.0f, (_Float16)0.00048828125f};
  return __builtin_amdgcn_fdot2(w, k, 0.0f, false);
}

template <int JT, bool SPLIT = false>
__device__ __forceinline__ void load_row(const float* __restrict__ row, float out[JT], int J, int lane) {
  if (JT == 4 && J == 4) {
    const float4 v = *reinterpret_cast<const float4*>(row + 4 * lane);
    out[0] = v.x;
    out[1] = v.y;
    out[2] = v.z;
    out[3] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < JT; ++j) out[j] = (j < J) ? row[lane + 64 * j] : 0.0f;
  }
  if (SPLIT) {
#pragma unroll
    for (int j = 0; j < JT; ++j) out[j] = tsg_state_value(out[j]);
  }
}

// Online-softmax attention state of one wave (IntraAttention.forward, function_g_fc.py:31-35).
template <int JT>
struct AttnAcc {
  float m, l;
  float v[JT];
};

template <int JT>
__device__ __forceinline__ void attn_init(AttnAcc<JT>& s) {
  s.m = -INFINITY;
  s.l = 0.0f;
#pragma unroll
  for (int j = 0; j < JT; ++j) s.v[j] = 0.0f;
}

// Attention over key rows in groups of 4: attn_load4 issues the (independent, coalesced) row loads of a
// group, attn_fold4 folds a loaded group into the running online softmax.  The caller loads group g+1
// before folding group g, so eight rows per wave are in flight.  rows[u] == nullptr marks an absent row.
// (split state rows are loaded RAW here and decoded in attn_fold4, so that decoding never pulls the wait for a
// group's loads in front of the previous group's fold)
template <int JT>
__device__ __forceinline__ void attn_load4(float key[4][JT], const float* const rows[4], int J, int lane) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (rows[u]) {
      load_row<JT>(rows[u], key[u], J, lane);
    } else {
#pragma unroll
      for (int j = 0; j < JT; ++j) key[u][j] = 0.0f;
    }
  }
}

template <int JT, bool SPLIT = false>
__device__ __forceinline__ void attn_fold4(AttnAcc<JT>& s, float key[4][JT], unsigned present,
                                           const float q[JT]) {
  if (SPLIT) {  // decode in place
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < JT; ++j) key[u][j] = tsg_state_value(key[u][j]);
  }
  float part[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    part[u] = 0.0f;
#pragma unroll
    for (int j = 0; j < JT; ++j) part[u] += key[u][j] * q[j];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) part[u] = tsg_wave_sum_u(part[u]);
  float mn = s.m;
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (present & (1u << u)) mn = fmaxf(mn, part[u]);
  const float scale = __expf(s.m - mn);  // exp(-inf) = 0 on the first fold
  float w[4];
  float wsum = 0.0f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    w[u] = (present & (1u << u)) ? __expf(part[u] - mn) : 0.0f;
    wsum += w[u];
  }
  s.l = s.l * scale + wsum;
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    float a = s.v[j] * scale;
#pragma unroll
    for (int u = 0; u < 4; ++u) a += w[u] * key[u][j];
    s.v[j] = a;
  }
  s.m = mn;
}

// N rows that all exist (the caller names a valid row for every u and masks the absent ones when it folds): no
// branch between the loads.  With a branch per row (attn_load_n's null check, load_row's run-time J test) the compiler
// puts `s_waitcnt vmcnt(0)` in front of every row load (the registers it writes may, on some path through the
// branches, still be the target of a load in flight), and a group's rows then travel one after the other.
// VEC4: hp == 256, one dwordx4 per lane and row; otherwise lane owns elements lane + 64 j, j < J.
template <int JT, int N, bool VEC4>
__device__ __forceinline__ void attn_load_rows(float key[N][JT], const float* const rows[N], int J, int lane) {
  if (VEC4) {
    static_assert(!VEC4 || JT == 4, "one dwordx4 per lane: hp == 256");
#pragma unroll
    for (int u = 0; u < N; ++u) {
      const float4 v = *reinterpret_cast<const float4*>(rows[u] + 4 * lane);
      key[u][0] = v.x;
      key[u][1] = v.y;
      key[u][2] = v.z;
      key[u][3] = v.w;
    }
  } else {
#pragma unroll
    for (int u = 0; u < N; ++u)
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const float v = rows[u][lane + 64 * (j < J ? j : 0)];
        key[u][j] = j < J ? v : 0.0f;  // a select, not a branch
      }
  }
}

// The same in groups of N rows (the engine's scoring kernel keeps two groups of eight in flight).
template <int JT, int N>
__device__ __forceinline__ void attn_load_n(float key[N][JT], const float* const rows[N], int J, int lane) {
#pragma unroll
  for (int u = 0; u < N; ++u) {
    if (rows[u]) {
      load_row<JT>(rows[u], key[u], J, lane);
    } else {
#pragma unroll
      for (int j = 0; j < JT; ++j) key[u][j] = 0.0f;
    }
  }
}

template <int JT, int N, bool SPLIT = false>
__device__ __forceinline__ void attn_fold_n(AttnAcc<JT>& s, float key[N][JT], unsigned present, const float q[JT]) {
  if (SPLIT) {  // decode in place
#pragma unroll
    for (int u = 0; u < N; ++u)
#pragma unroll
      for (int j = 0; j < JT; ++j) key[u][j] = tsg_state_value(key[u][j]);
  }
  float part[N];
#pragma unroll
  for (int u = 0; u < N; ++u) {
    float p = 0.0f;
#pragma unroll
    for (int j = 0; j < JT; ++j) p += key[u][j] * q[j];
    part[u] = tsg_wave_sum_u(p);
  }
  float mn = s.m;
#pragma unroll
  for (int u = 0; u < N; ++u)
    if (present & (1u << u)) mn = fmaxf(mn, part[u]);
  const float scale = __expf(s.m - mn);  // exp(-inf) = 0 on the first fold
  float w[N];
  float wsum = 0.0f;
#pragma unroll
  for (int u = 0; u < N; ++u) {
    w[u] = (present & (1u << u)) ? __expf(part[u] - mn) : 0.0f;
    wsum += w[u];
  }
  s.l = s.l * scale + wsum;
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    float a = s.v[j] * scale;
#pragma unroll
    for (int u = 0; u < N; ++u) a += w[u] * key[u][j];
    s.v[j] = a;
  }
  s.m = mn;
}

// H logit of one candidate: factored emb_fc + dis_fc + out_fc (distance_gat_fc.py:129-133)
__device__ __forceinline__ float tsg_h_logit(const tsg_model& m, int cand, int des, float dis) {
  const float* hpar = m.hparams;
  const float* pf = m.pfrom + (size_t)cand * TSG_H_HIDDEN;
  const float* pt = m.pto + (size_t)des * TSG_H_HIDDEN;
  float acc = hpar[40];
#pragma unroll
  for (int j = 0; j < TSG_H_HIDDEN; ++j) acc += hpar[20 + j] * fmaxf(pf[j] + pt[j], 0.0f);
#pragma unroll
  for (int j = 0; j < TSG_H_HIDDEN; ++j) acc += hpar[30 + j] * fmaxf(hpar[j] * dis + hpar[10 + j], 0.0f);
  return acc;
}

// The same from rows already in registers (pf = pfrom[cand], pt = pto[des]): identical arithmetic, term by term.
__device__ __forceinline__ float tsg_h_logit_regs(const float hpar[42], const float pf[TSG_H_HIDDEN],
                                                  const float pt[TSG_H_HIDDEN], float dis) {
  float acc = hpar[40];
#pragma unroll
  for (int j = 0; j < TSG_H_HIDDEN; ++j) acc += hpar[20 + j] * fmaxf(pf[j] + pt[j], 0.0f);
#pragma unroll
  for (int j = 0; j < TSG_H_HIDDEN; ++j) acc += hpar[30 + j] * fmaxf(hpar[j] * dis + hpar[10 + j], 0.0f);
  return acc;
}

// softmax((1-w1) softmax(G) + w1 softmax(H)) over the lanes with `on` set (generator_v4.py:46-54,71-72);
// one candidate per lane, whole wave participates.
__device__ __forceinline__ float tsg_mix_softmax(const tsg_model& m, float g_logit, float h_logit, bool on, float* pg,
                                                 float* ph) {
  const float w1 = m.hparams[41];
  const float gmax = tsg_wave_max_u(on ? g_logit : -INFINITY);
  const float ge = on ? expf(g_logit - gmax) : 0.0f;
  const float g = ge / tsg_wave_sum_u(ge);
  const float hmax = tsg_wave_max_u(on ? h_logit : -INFINITY);
  const float he = on ? expf(h_logit - hmax) : 0.0f;
  const float h = he / tsg_wave_sum_u(he);
  const float mix = on ? (1.0f - w1) * g + w1 * h : -INFINITY;
  const float mmax = tsg_wave_max_u(mix);
  const float me = on ? expf(mix - mmax) : 0.0f;
  const float f = me / tsg_wave_sum_u(me);
  if (pg) *pg = g;
  if (ph) *ph = h;
  return f;
}

// Score up to 64 candidates of one row.  Lane i < ncand returns the probability f_i of
// softmax((1-w1) softmax(G) + w1 softmax(H))  (generator_v4.py:46-54,71-72); also the two inner
// softmaxes through pg/ph.  `pooled` = attention-pooled hidden in the load_row lane map.
template <int JT>
__device__ __forceinline__ float score_row(const tsg_model& m, const float pooled[JT], int J, int lane, int ncand,
                                           int my_cand, float my_dis, int des, float* pg, float* ph,
                                           float* lg = nullptr, float* lh = nullptr) {
  const int hp = m.hp;
  // G: gathered rows of out_fc (function_g_fc.py:184-186) -- only the ncand rows that are needed
  float g_logit = -INFINITY;
  for (int c0 = 0; c0 < ncand; c0 += 4) {
    float part[4];
    int cid[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u < ncand ? c0 + u : ncand - 1;
      cid[u] = __shfl(my_cand, c, TSG_WAVE);
      float w[JT];
      load_row<JT>(m.out_w + (size_t)cid[u] * hp, w, J, lane);
      part[u] = 0.0f;
#pragma unroll
      for (int j = 0; j < JT; ++j) part[u] += w[j] * pooled[j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) part[u] = tsg_wave_sum_u(part[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (lane == c0 + u && c0 + u < ncand) g_logit = part[u] + m.out_b[cid[u]];
  }
  const float h_logit = lane < ncand ? tsg_h_logit(m, my_cand, des, my_dis) : -INFINITY;
  float g, h;
  const float f = tsg_mix_softmax(m, g_logit, h_logit, lane < ncand, &g, &h);
  if (pg) *pg = g;
  if (ph) *ph = h;
  if (lg) *lg = g_logit;
  if (lh) *lh = h_logit;
  return f;
}
