// Device-side building blocks of the scoring kernels (one wave64 per scored row).
#pragma once
#include "tsg_common.h"

// WGS-84 inverse geodesic (Vincenty), fp64, metres.  Same iteration as oracle/geodesic.py; stands
// in for geopy.distance.distance(...) at search.py:355-356.
__device__ inline double tsg_dev_geodesic_m(double lat1, double lon1, double lat2, double lon2) {
  const double a = 6378137.0;
  const double f = 1.0 / 298.257223563;
  const double b = a * (1.0 - f);
  if (lat1 == lat2 && lon1 == lon2) return 0.0;
  const double rad = 3.14159265358979323846 / 180.0;
  const double U1 = atan((1.0 - f) * tan(lat1 * rad));
  const double U2 = atan((1.0 - f) * tan(lat2 * rad));
  const double L = (lon2 - lon1) * rad;
  const double sinU1 = sin(U1), cosU1 = cos(U1), sinU2 = sin(U2), cosU2 = cos(U2);
  double lam = L, sin_sigma = 0.0, cos_sigma = 0.0, sigma = 0.0, cos_sq_alpha = 0.0, cos2sm = 0.0;
  for (int it = 0; it < 20; ++it) {
    const double sin_lam = sin(lam), cos_lam = cos(lam);
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

// Online-softmax attention state of one wave: lane owns elements lane + 64 j of the pooled vector.
struct AttnAcc {
  float m, l;
  float v[TSG_MAXJ];
};

__device__ __forceinline__ void attn_init(AttnAcc& s) {
  s.m = -INFINITY;
  s.l = 0.0f;
#pragma unroll
  for (int j = 0; j < TSG_MAXJ; ++j) s.v[j] = 0.0f;
}

// Fold one key row (already in registers) into the running softmax(keys . q)-weighted sum
// (IntraAttention.forward, function_g_fc.py:31-35).
__device__ __forceinline__ void attn_fold(AttnAcc& s, const float key[TSG_MAXJ], const float q[TSG_MAXJ], int J) {
  float part = 0.0f;
#pragma unroll
  for (int j = 0; j < TSG_MAXJ; ++j)
    if (j < J) part += key[j] * q[j];
  const float sc = tsg_wave_sum(part);
  const float mn = fmaxf(s.m, sc);
  const float scale = expf(s.m - mn);  // exp(-inf) = 0 on the first fold
  const float w = expf(sc - mn);
  s.l = s.l * scale + w;
#pragma unroll
  for (int j = 0; j < TSG_MAXJ; ++j)
    if (j < J) s.v[j] = s.v[j] * scale + w * key[j];
  s.m = mn;
}

// Score up to 64 candidates of one row.  Lane i < ncand returns log-free probability f_i of
// softmax((1-w1) softmax(G) + w1 softmax(H))  (generator_v4.py:46-54,71-72); also the two inner
// softmaxes through pg/ph.  `pooled` = attention-pooled hidden (lane owns lane + 64 j).
__device__ __forceinline__ float score_row(const tsg_model& m, const float pooled[TSG_MAXJ], int J, int lane,
                                           int ncand, int my_cand, float my_dis, int des, float* pg, float* ph) {
  const int hp = m.hp;
  // G: gathered rows of out_fc (function_g_fc.py:184-186) -- only the ncand rows that are needed
  float g_logit = -INFINITY;
  for (int c = 0; c < ncand; ++c) {
    const int cid = __shfl(my_cand, c, TSG_WAVE);
    const float* w = m.out_w + (size_t)cid * hp;
    float part = 0.0f;
#pragma unroll
    for (int j = 0; j < TSG_MAXJ; ++j)
      if (j < J) part += w[lane + 64 * j] * pooled[j];
    const float tot = tsg_wave_sum(part) + m.out_b[cid];
    if (lane == c) g_logit = tot;
  }
  // H: factored emb_fc + dis_fc + out_fc (distance_gat_fc.py:129-133)
  float h_logit = -INFINITY;
  if (lane < ncand) {
    const float* hpar = m.hparams;
    const float* pf = m.pfrom + (size_t)my_cand * TSG_H_HIDDEN;
    const float* pt = m.pto + (size_t)des * TSG_H_HIDDEN;
    float acc = hpar[40];
#pragma unroll
    for (int j = 0; j < TSG_H_HIDDEN; ++j) acc += hpar[20 + j] * fmaxf(pf[j] + pt[j], 0.0f);
#pragma unroll
    for (int j = 0; j < TSG_H_HIDDEN; ++j) acc += hpar[30 + j] * fmaxf(hpar[j] * my_dis + hpar[10 + j], 0.0f);
    h_logit = acc;
  }
  const float w1 = m.hparams[41];
  const bool on = lane < ncand;
  const float gmax = tsg_wave_max(g_logit);
  const float ge = on ? expf(g_logit - gmax) : 0.0f;
  const float g = ge / tsg_wave_sum(ge);
  const float hmax = tsg_wave_max(h_logit);
  const float he = on ? expf(h_logit - hmax) : 0.0f;
  const float h = he / tsg_wave_sum(he);
  const float mix = on ? (1.0f - w1) * g + w1 * h : -INFINITY;
  const float mmax = tsg_wave_max(mix);
  const float me = on ? expf(mix - mmax) : 0.0f;
  const float f = me / tsg_wave_sum(me);
  if (pg) *pg = g;
  if (ph) *ph = h;
  return f;
}
