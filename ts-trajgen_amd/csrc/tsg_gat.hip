// GAT road-graph encoder (generator/gat.py:138-200 + :84-114, eval mode) on a CSR-by-target graph:
//   1. projection  proj = x . W^T                      (MFMA GEMM, tsg_dense.hip)
//   2. node scores s_src/s_trg = <proj[n,h,:], a[h,:]> (one wave per (node, head))
//   3. global max of LeakyReLU(s_src[src] + s_trg[trg]) over all edges and heads   (gat.py:221)
//   4. per target node: segmented edge softmax (wave = head) + SpMM aggregation of the gathered
//      source rows + skip projection + head mean/concat + bias + ELU, all in one pass.
// Incoming edges are visited in their original COO order, so the per-target sums are accumulated
// in the same order as the reference's sequential scatter_add_.
// With head averaging and a narrow input (concat == 0, FIN <= 64: the reference's road encoder) the layer runs in
// the input space instead (gat_forward_inputspace below): the N x NH x F projection is never written.
#include <mutex>

#include "tsg_common.h"

extern "C" int tsg_gemm_f32(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C, int ldc,
                            int M, int N, int K, void* stream);
int tsg_gemm_f32_act(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C, int ldc, int M,
                     int N, int K, int elu, void* stream);

__device__ __forceinline__ float leaky02(float v) { return v > 0.0f ? v : 0.2f * v; }

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  if (v >= 0.0f)
    atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else
    atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ __launch_bounds__(256) void gat_node_scores_kernel(const float* __restrict__ proj,
                                                              const float* __restrict__ a_src,
                                                              const float* __restrict__ a_trg, float* __restrict__ s_src,
                                                              float* __restrict__ s_trg, int n_nodes, int heads, int fout) {
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= (long)n_nodes * heads) return;
  const int h = (int)(w % heads);
  const float* p = proj + (size_t)w * fout;
  float ps = 0.0f, pt = 0.0f;
  for (int f = lane; f < fout; f += 64) {
    const float v = p[f];
    ps += v * a_src[h * fout + f];
    pt += v * a_trg[h * fout + f];
  }
  ps = tsg_wave_sum(ps);
  pt = tsg_wave_sum(pt);
  if (lane == 0) {
    s_src[w] = ps;
    s_trg[w] = pt;
  }
}

// The per-node scores without re-reading the projection (N x NH x F floats): <proj[n,h,:], a[h,:]> =
// <x[n,:] W_h, a_h> = <x[n,:], W_h a_h>.  w_att[k][h] = sum_f W[k][h*F+f] a_src[h][f] (and a_trg in columns NH..2NH-1)
// is FIN x 2NH numbers; the scores are then FIN multiply-adds per (node, head) on the input features.
__global__ __launch_bounds__(256) void gat_att_weights_kernel(const float* __restrict__ w_proj_t,
                                                              const float* __restrict__ a_src,
                                                              const float* __restrict__ a_trg, float* __restrict__ w_att,
                                                              int fin, int heads, int fout) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per (k, h)
  const int lane = threadIdx.x & 63;
  if (w >= fin * heads) return;
  const int k = w / heads, h = w % heads;
  const float* row = w_proj_t + (size_t)k * heads * fout + (size_t)h * fout;
  float ps = 0.0f, pt = 0.0f;
  for (int f = lane; f < fout; f += 64) {
    ps += row[f] * a_src[h * fout + f];
    pt += row[f] * a_trg[h * fout + f];
  }
  ps = tsg_wave_sum(ps);
  pt = tsg_wave_sum(pt);
  if (lane == 0) {
    w_att[(size_t)k * 2 * heads + h] = ps;
    w_att[(size_t)k * 2 * heads + heads + h] = pt;
  }
}

__global__ __launch_bounds__(256) void gat_node_scores_x_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ w_att,
                                                                float* __restrict__ s_src, float* __restrict__ s_trg,
                                                                int n_nodes, int heads, int fin) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n_nodes * heads) return;
  const int n = (int)(t / heads), h = (int)(t % heads);
  const float* xr = x + (size_t)n * fin;
  float ps = 0.0f, pt = 0.0f;
  for (int k = 0; k < fin; ++k) {
    const float v = xr[k];
    ps += v * w_att[(size_t)k * 2 * heads + h];
    pt += v * w_att[(size_t)k * 2 * heads + heads + h];
  }
  s_src[t] = ps;
  s_trg[t] = pt;
}

__global__ void gat_init_max_kernel(float* gmax) { *gmax = -INFINITY; }

__global__ __launch_bounds__(256) void gat_edge_max_kernel(const int32_t* __restrict__ in_ptr,
                                                           const int32_t* __restrict__ in_src,
                                                           const float* __restrict__ s_src,
                                                           const float* __restrict__ s_trg, float* gmax, int n_nodes,
                                                           int heads) {
  float m = -INFINITY;
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < n_nodes; n += gridDim.x * blockDim.x) {
    const int e0 = in_ptr[n], e1 = in_ptr[n + 1];
    for (int e = e0; e < e1; ++e) {
      const int s = in_src[e];
      for (int h = 0; h < heads; ++h) m = fmaxf(m, leaky02(s_src[(size_t)s * heads + h] + s_trg[(size_t)n * heads + h]));
    }
  }
  m = tsg_wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > -INFINITY) atomic_max_float(gmax, m);
}

#define GAT_MAXJ 8  // fout <= 512

// mean over heads of the skip projection (concat == 0): mean_h(agg_h + x.Ws_h) = mean_h(agg_h) + x.mean_h(Ws_h),
// so the skip term is one [FIN][F] product instead of NH of them (gat.py:101,109)
__global__ __launch_bounds__(256) void gat_skip_mean_kernel(const float* __restrict__ w_skip_t, float* __restrict__ w_mean,
                                                            int fin, int heads, int fout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= fin * fout) return;
  const int k = i / fout, f = i % fout;
  float s = 0.0f;
  for (int h = 0; h < heads; ++h) s += w_skip_t[(size_t)k * heads * fout + (size_t)h * fout + f];
  w_mean[i] = s / (float)heads;
}

// One wave per target node.  Per head: the incoming edges' attention weights are computed one edge per lane
// (segmented softmax over the node's CSR row, gat.py:225-230,246), then the source rows proj[src][h][:]
// (F contiguous floats each) are gathered four at a time, the whole wave reading one row coalesced, and
// accumulated in COO order (mul, then add: gat.py:188,263).  Heads are averaged in registers; the skip
// term (already a dense product, with the bias folded in) and ELU close the row: no LDS, no second pass.
// VEC = 4: F is a multiple of 256 and lane l owns floats 4l..4l+3 of every 256-float segment (dwordx4 loads);
// VEC = 2: F == 128, lane l owns floats 2l, 2l+1 (dwordx2: one load per 512-byte row); VEC = 0: any F, lane l owns
// floats l, l+64, ...
template <int VEC>
__global__ __launch_bounds__(256) void gat_aggregate_kernel(tsg_gat_desc d) {
  constexpr bool VEC4 = VEC == 4;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= d.n_nodes) return;
  const int F = d.fout, NH = d.heads;
  const int e0 = d.in_ptr[n], e1 = d.in_ptr[n + 1];
  const float gmax = *d.gmax;
  const int nv = VEC4 ? F / 64 : VEC == 2 ? 2 : (F + 63) / 64;  // floats per lane
  auto col = [&](int j) { return VEC4 ? (j >> 2) * 256 + 4 * lane + (j & 3) : VEC == 2 ? 2 * lane + j : lane + 64 * j; };
  float mean_acc[GAT_MAXJ];
#pragma unroll
  for (int j = 0; j < GAT_MAXJ; ++j) mean_acc[j] = 0.0f;
  const int outw = d.concat ? NH * F : F;
  for (int h = 0; h < NH; ++h) {
    const float st = d.s_trg[(size_t)n * NH + h];
    // segmented softmax denominator over the incoming edges of n, summed in COO order
    float denom = 0.0f;
    for (int c0 = e0; c0 < e1; c0 += 64) {
      const int cnt = e1 - c0 < 64 ? e1 - c0 : 64;
      float ex = 0.0f;
      if (lane < cnt) ex = expf(leaky02(d.s_src[(size_t)d.in_src[c0 + lane] * NH + h] + st) - gmax);
      for (int i = 0; i < cnt; ++i) denom += __shfl(ex, i, TSG_WAVE);
    }
    const float dinv_base = denom + 1e-16f;  // gat.py:230
    float acc[GAT_MAXJ];
#pragma unroll
    for (int j = 0; j < GAT_MAXJ; ++j) acc[j] = 0.0f;
    for (int c0 = e0; c0 < e1; c0 += 64) {
      const int cnt = e1 - c0 < 64 ? e1 - c0 : 64;
      int src = 0;
      float att = 0.0f;
      if (lane < cnt) {
        src = d.in_src[c0 + lane];
        att = expf(leaky02(d.s_src[(size_t)src * NH + h] + st) - gmax) / dinv_base;
      }
      for (int i0 = 0; i0 < cnt; i0 += 4) {
        float p[4][GAT_MAXJ];
        float a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // four source rows in flight
          const int i = i0 + u < cnt ? i0 + u : cnt - 1;
          const int s = __shfl(src, i, TSG_WAVE);
          a[u] = __shfl(att, i, TSG_WAVE);
          const float* row = d.proj + ((size_t)s * NH + h) * F;
          if (VEC4) {
#pragma unroll
            for (int q = 0; q < GAT_MAXJ / 4; ++q) {
              if (q * 256 < F) {
                const float4 v = *reinterpret_cast<const float4*>(row + q * 256 + 4 * lane);
                p[u][4 * q] = v.x;
                p[u][4 * q + 1] = v.y;
                p[u][4 * q + 2] = v.z;
                p[u][4 * q + 3] = v.w;
              }
            }
          } else if (VEC == 2) {
            const float2 v = *reinterpret_cast<const float2*>(row + 2 * lane);
            p[u][0] = v.x;
            p[u][1] = v.y;
#pragma unroll
            for (int j = 2; j < GAT_MAXJ; ++j) p[u][j] = 0.0f;
          } else {
#pragma unroll
            for (int j = 0; j < GAT_MAXJ; ++j) {
              const int f = lane + 64 * j;
              p[u][j] = f < F ? row[f] : 0.0f;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (i0 + u < cnt) {
#pragma unroll
            for (int j = 0; j < GAT_MAXJ; ++j)
              if (j < nv) acc[j] = __fadd_rn(acc[j], __fmul_rn(p[u][j], a[u]));  // mul then add, no fma
          }
        }
      }
    }
    if (d.concat) {
#pragma unroll
      for (int j = 0; j < GAT_MAXJ; ++j) {
        const int f = col(j);
        if (j < nv && f < F) {
          const size_t o = (size_t)n * outw + (size_t)h * F + f;
          float v = acc[j];
          if (d.skip_identity)
            v += d.x[(size_t)n * d.fin + f] + (d.bias ? d.bias[h * F + f] : 0.0f);  // gat.py:94-97
          else
            v += d.skip[o];  // x . Ws^T + bias
          d.out[o] = v > 0.0f ? v : expm1f(v);  // gat.py:114 ELU
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < GAT_MAXJ; ++j) mean_acc[j] += acc[j];
    }
  }
  if (!d.concat) {
#pragma unroll
    for (int j = 0; j < GAT_MAXJ; ++j) {
      const int f = col(j);
      if (j < nv && f < F) {
        float v = mean_acc[j] / (float)NH;  // gat.py:109
        if (d.skip_identity)
          v += d.x[(size_t)n * d.fin + f] + (d.bias ? d.bias[f] : 0.0f);
        else
          v += d.skip[(size_t)n * F + f];  // x . mean_h(Ws_h)^T + bias (gat.py:101,109,112)
        d.out[(size_t)n * F + f] = v > 0.0f ? v : expm1f(v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------ input-space layer
// concat == 0 (heads averaged, gat.py:109) with a learned skip projection:
//   out[n] = ELU( 1/NH sum_h sum_j att_hj (x[src_j] W_h)  +  x[n] mean_h(Ws_h)  +  b )
//          = ELU( [ sum_j att_1j x[src_j] | ... | sum_j att_NHj x[src_j] | x[n] ] . Wcat + b ),
//   Wcat = [ W_1 / NH ; ... ; W_NH / NH ; mean_h(Ws_h) ]   ((NH+1) FIN rows, F columns).
// The attention-weighted sums are taken over the FIN-wide input rows (120 bytes at the reference's dimensions,
// against NH*F*4 = 5 KB projected), and one matrix product per node closes the layer with bias and ELU in its
// epilogue.  Against the reference this moves the projection behind the edge sum (sum_j att_j (x_j W) becomes
// (sum_j att_j x_j) W): the same real number, rounded in another order; the layer's tests hold both forms to the
// same 1e-4 bound (tests/test_gpu_parity.py, G1/G2).

// One launch for everything that depends on the weights alone: w_att (gat_att_weights_kernel's job, one wave per
// (k, h)) in the first att_blocks blocks, Wcat in the others, and the global maximum's start value.
__global__ __launch_bounds__(256) void gat_prepare_kernel(tsg_gat_desc d, float* __restrict__ wcat, int att_blocks) {
  const int NH = d.heads, F = d.fout, FIN = d.fin;
  if (blockIdx.x == 0 && threadIdx.x == 0) *d.gmax = -INFINITY;
  if ((int)blockIdx.x < att_blocks) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= FIN * NH) return;
    const int k = w / NH, h = w % NH;
    const float* row = d.w_proj_t + (size_t)k * NH * F + (size_t)h * F;
    float ps = 0.0f, pt = 0.0f;
    for (int f = lane; f < F; f += 64) {
      ps += row[f] * d.a_src[h * F + f];
      pt += row[f] * d.a_trg[h * F + f];
    }
    ps = tsg_wave_sum(ps);
    pt = tsg_wave_sum(pt);
    if (lane == 0) {
      d.w_att[(size_t)k * 2 * NH + h] = ps;
      d.w_att[(size_t)k * 2 * NH + NH + h] = pt;
    }
    return;
  }
  const long i = (long)(blockIdx.x - att_blocks) * 256 + threadIdx.x;
  if (i >= (long)(NH + 1) * FIN * F) return;
  const int r = (int)(i / F), f = (int)(i % F);
  float v;
  if (r < NH * FIN) {
    const int h = r / FIN, k = r % FIN;
    v = d.w_proj_t[(size_t)k * NH * F + (size_t)h * F + f] / (float)NH;
  } else {
    const int k = r - NH * FIN;
    float s = 0.0f;
    for (int h = 0; h < NH; ++h) s += d.w_skip_t[(size_t)k * NH * F + (size_t)h * F + f];
    v = s / (float)NH;
  }
  wcat[i] = v;
}

// Scores of all nodes from the input rows: a workgroup (one wave) stages 64 rows of x in LDS (coalesced reads, odd
// row stride), then thread t folds row t against the 2 NH columns of w_att (uniform addresses: scalar loads).
#define GAT_SCORE_ROWS 64
__global__ __launch_bounds__(GAT_SCORE_ROWS) void gat_node_scores_lds_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ w_att,
                                                                  float* __restrict__ s_src, float* __restrict__ s_trg,
                                                                  int n_nodes, int heads, int fin) {
  extern __shared__ float rows[];  // [GAT_SCORE_ROWS][fin | 1]
  const int ld = fin | 1;
  const long n0 = (long)blockIdx.x * GAT_SCORE_ROWS;
  const int lim = (int)((long)n_nodes - n0 < GAT_SCORE_ROWS ? (long)n_nodes - n0 : GAT_SCORE_ROWS) * fin;
  for (int i = threadIdx.x; i < lim; i += GAT_SCORE_ROWS) rows[(i / fin) * ld + (i % fin)] = x[n0 * fin + i];
  __syncthreads();
  const long n = n0 + threadIdx.x;
  if (n >= n_nodes) return;
  const float* r = rows + threadIdx.x * ld;
  for (int h = 0; h < heads; ++h) {
    float ps = 0.0f, pt = 0.0f;
    for (int k = 0; k < fin; ++k) {  // same order of additions as gat_node_scores_x_kernel
      const float v = r[k];
      ps += v * w_att[(size_t)k * 2 * heads + h];
      pt += v * w_att[(size_t)k * 2 * heads + heads + h];
    }
    s_src[n * heads + h] = ps;
    s_trg[n * heads + h] = pt;
  }
}

// Global maximum of the edge scores, one thread per (target node, head); one atomic per workgroup (thousands of
// atomics on the one address would cost more than the scan itself).
__global__ __launch_bounds__(256) void gat_edge_max_nh_kernel(const int32_t* __restrict__ in_ptr,
                                                              const int32_t* __restrict__ in_src,
                                                              const float* __restrict__ s_src,
                                                              const float* __restrict__ s_trg, float* gmax, int n_nodes,
                                                              int heads) {
  __shared__ float wmax[4];
  float m = -INFINITY;
  const long total = (long)n_nodes * heads;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int n = (int)(t / heads), h = (int)(t % heads);
    const int e0 = in_ptr[n], e1 = in_ptr[n + 1];
    const float st = s_trg[t];
    for (int e = e0; e < e1; ++e) m = fmaxf(m, leaky02(s_src[(size_t)in_src[e] * heads + h] + st));
  }
  m = tsg_wave_max(m);
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (m > -INFINITY) atomic_max_float(gmax, m);
  }
}

// One group of G lanes (a wave, or half a wave when FIN <= 32) per target node, lane k = input feature k.  The
// softmax weights of all heads are computed one edge per lane (same expressions, same COO order as
// gat_aggregate_kernel); every source row x[src] is read once, 4 FIN bytes coalesced, and folded into the NH
// per-head sums.  The rows of the first four edges are requested together with the scores, so a node of
// in-degree <= 4 (a road network's usual case) costs three dependent round trips: in_ptr, in_src, {scores, rows}.
// Row n of aggx is [sum_1 | ... | sum_NH | x[n]].
template <int G>
__global__ __launch_bounds__(256) void gat_aggregate_x_kernel(tsg_gat_desc d, float* __restrict__ aggx) {
  const int n = (int)(((long)blockIdx.x * 256 + threadIdx.x) / G);
  const int gl = threadIdx.x & (G - 1);
  if (n >= d.n_nodes) return;
  const int NH = d.heads, FIN = d.fin;
  const int e0 = d.in_ptr[n], e1 = d.in_ptr[n + 1];
  const float gmax = *d.gmax;
  const float own = gl < FIN ? d.x[(size_t)n * FIN + gl] : 0.0f;
  float st[16], denom[16], acc[16], ex0[16];
#pragma unroll
  for (int h = 0; h < 16; ++h) {
    st[h] = h < NH ? d.s_trg[(size_t)n * NH + h] : 0.0f;
    denom[h] = 0.0f;
    acc[h] = 0.0f;
    ex0[h] = 0.0f;
  }
  const int cnt0 = e1 - e0 < G ? e1 - e0 : G;
  const int src0 = gl < cnt0 ? d.in_src[e0 + gl] : 0;
  float xr[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int s = __shfl(src0, u < cnt0 ? u : 0, G);
    xr[u] = (u < cnt0 && gl < FIN) ? d.x[(size_t)s * FIN + gl] : 0.0f;
  }
#pragma unroll
  for (int h = 0; h < 16; ++h)
    if (h < NH && gl < cnt0) ex0[h] = expf(leaky02(d.s_src[(size_t)src0 * NH + h] + st[h]) - gmax);
  // segmented softmax denominators over all incoming edges, summed in COO order
  for (int c0 = e0; c0 < e1; c0 += G) {
    const int cnt = e1 - c0 < G ? e1 - c0 : G;
    const int src = c0 == e0 ? src0 : (gl < cnt ? d.in_src[c0 + gl] : 0);
#pragma unroll
    for (int h = 0; h < 16; ++h) {
      if (h < NH) {
        float ex = ex0[h];
        if (c0 != e0) ex = gl < cnt ? expf(leaky02(d.s_src[(size_t)src * NH + h] + st[h]) - gmax) : 0.0f;
        for (int i = 0; i < cnt; ++i) denom[h] += __shfl(ex, i, G);
      }
    }
  }
  for (int c0 = e0; c0 < e1; c0 += G) {
    const int cnt = e1 - c0 < G ? e1 - c0 : G;
    const bool first = c0 == e0;
    const int src = first ? src0 : (gl < cnt ? d.in_src[c0 + gl] : 0);
    float att[16];
#pragma unroll
    for (int h = 0; h < 16; ++h) {
      att[h] = 0.0f;
      if (h < NH && gl < cnt) {
        const float ex = first ? ex0[h] : expf(leaky02(d.s_src[(size_t)src * NH + h] + st[h]) - gmax);
        att[h] = ex / (denom[h] + 1e-16f);  // gat.py:230
      }
    }
    int i = 0;
    if (first) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u < cnt) {
#pragma unroll
          for (int h = 0; h < 16; ++h)
            if (h < NH) acc[h] = __fadd_rn(acc[h], __fmul_rn(xr[u], __shfl(att[h], u, G)));
        }
      }
      i = 4;
    }
    for (; i < cnt; ++i) {
      const int s = __shfl(src, i, G);
      const float xv = gl < FIN ? d.x[(size_t)s * FIN + gl] : 0.0f;
#pragma unroll
      for (int h = 0; h < 16; ++h)
        if (h < NH) acc[h] = __fadd_rn(acc[h], __fmul_rn(xv, __shfl(att[h], i, G)));
    }
  }
  if (gl < FIN) {
    float* row = aggx + (size_t)n * (NH + 1) * FIN;
#pragma unroll
    for (int h = 0; h < 16; ++h)
      if (h < NH) row[h * FIN + gl] = acc[h];
    row[NH * FIN + gl] = own;
  }
}

// Workspaces: aggx [N][(NH+1) FIN] lives in d->proj (smaller than the projection it replaces), Wcat
// [(NH+1) FIN][F] in d->skip; the caller has checked that both fit.
static int gat_forward_inputspace(const tsg_gat_desc* d, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int kw = (d->heads + 1) * d->fin;
  float* aggx = d->proj;
  float* wcat = d->skip;
  const int att_blocks = tsg_cdiv((long)d->fin * d->heads, 4);
  const int cat_blocks = tsg_cdiv((long)kw * d->fout, 256);
  hipLaunchKernelGGL(gat_prepare_kernel, dim3(att_blocks + cat_blocks), dim3(256), 0, s, *d, wcat, att_blocks);
  TSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(gat_node_scores_lds_kernel, dim3(tsg_cdiv(d->n_nodes, GAT_SCORE_ROWS)), dim3(GAT_SCORE_ROWS),
                     GAT_SCORE_ROWS * (size_t)(d->fin | 1) * 4, s, d->x, d->w_att, d->s_src, d->s_trg, d->n_nodes,
                     d->heads, d->fin);
  TSG_LAUNCH_CHECK();
  const int mblocks = tsg_cdiv((long)d->n_nodes * d->heads, 256);
  hipLaunchKernelGGL(gat_edge_max_nh_kernel, dim3(mblocks < 512 ? mblocks : 512), dim3(256), 0, s, d->in_ptr, d->in_src,
                     d->s_src, d->s_trg, d->gmax, d->n_nodes, d->heads);
  TSG_LAUNCH_CHECK();
  if (d->fin <= 32)
    hipLaunchKernelGGL(gat_aggregate_x_kernel<32>, dim3(tsg_cdiv(d->n_nodes, 8)), dim3(256), 0, s, *d, aggx);
  else
    hipLaunchKernelGGL(gat_aggregate_x_kernel<64>, dim3(tsg_cdiv(d->n_nodes, 4)), dim3(256), 0, s, *d, aggx);
  TSG_LAUNCH_CHECK();
  return tsg_gemm_f32_act(aggx, kw, wcat, d->fout, d->bias, d->out, d->fout, d->n_nodes, d->fout, kw, 1, stream);
}

static int gat_forward_impl(const tsg_gat_desc* d, void* stream) {
  if (!d) return TSG_E_INVALID;
  if (d->n_nodes <= 0 || d->n_edges < 0 || d->fin <= 0 || d->heads <= 0 || d->fout <= 0) return TSG_E_INVALID;
  if (d->heads > 16 || d->fout > 64 * GAT_MAXJ) return TSG_E_LIMIT;
  if (!d->x || !d->w_proj_t || !d->a_src || !d->a_trg || !d->in_ptr || !d->in_src || !d->proj || !d->s_src ||
      !d->s_trg || !d->gmax || !d->out)
    return TSG_E_INVALID;
  if (d->skip_identity && d->fin != d->fout) return TSG_E_INVALID;
  const size_t N = (size_t)d->n_nodes, HF = (size_t)d->heads * d->fout;
  const size_t outw = d->concat ? HF : (size_t)d->fout;
  if (d->proj_bytes < N * HF * 4 || d->s_bytes < N * d->heads * 4) return TSG_E_TOO_SMALL;
  if (d->out_bytes < N * outw * 4) return TSG_E_TOO_SMALL;
  if (!d->skip_identity) {
    if (!d->w_skip_t || !d->skip) return TSG_E_INVALID;
    if (d->skip_bytes < N * outw * 4) return TSG_E_TOO_SMALL;
    if (!d->concat && (!d->w_skip_mean || d->w_skip_mean_bytes < (size_t)d->fin * d->fout * 4)) return TSG_E_TOO_SMALL;
  }
  const size_t kw = (size_t)(d->heads + 1) * d->fin;
  if (!d->concat && !d->skip_identity && d->fin <= 64 && d->w_att &&
      d->w_att_bytes >= (size_t)d->fin * 2 * d->heads * 4 && N * kw * 4 <= d->proj_bytes &&
      kw * d->fout * 4 <= d->skip_bytes)
    return gat_forward_inputspace(d, stream);
  hipStream_t s = (hipStream_t)stream;
  int rc = tsg_gemm_f32(d->x, d->fin, d->w_proj_t, (int)HF, nullptr, d->proj, (int)HF, d->n_nodes, (int)HF, d->fin,
                        stream);
  if (rc) return rc;
  if (!d->skip_identity) {  // the skip connection as a dense product on the matrix cores, bias folded in
    if (d->concat) {
      rc = tsg_gemm_f32(d->x, d->fin, d->w_skip_t, (int)HF, d->bias, d->skip, (int)HF, d->n_nodes, (int)HF, d->fin,
                        stream);
    } else {
      hipLaunchKernelGGL(gat_skip_mean_kernel, dim3(tsg_cdiv((long)d->fin * d->fout, 256)), dim3(256), 0, s,
                         d->w_skip_t, d->w_skip_mean, d->fin, d->heads, d->fout);
      TSG_LAUNCH_CHECK();
      rc = tsg_gemm_f32(d->x, d->fin, d->w_skip_mean, d->fout, d->bias, d->skip, d->fout, d->n_nodes, d->fout,
                        d->fin, stream);
    }
    if (rc) return rc;
  }
  if (d->w_att && d->w_att_bytes >= (size_t)d->fin * 2 * d->heads * 4) {
    hipLaunchKernelGGL(gat_att_weights_kernel, dim3(tsg_cdiv((long)d->fin * d->heads, 4)), dim3(256), 0, s, d->w_proj_t,
                       d->a_src, d->a_trg, d->w_att, d->fin, d->heads, d->fout);
    TSG_LAUNCH_CHECK();
    hipLaunchKernelGGL(gat_node_scores_x_kernel, dim3(tsg_cdiv((long)N * d->heads, 256)), dim3(256), 0, s, d->x,
                       d->w_att, d->s_src, d->s_trg, d->n_nodes, d->heads, d->fin);
  } else {  // no workspace given: the scores from the materialised projection
    hipLaunchKernelGGL(gat_node_scores_kernel, dim3(tsg_cdiv((long)N * d->heads, 4)), dim3(256), 0, s, d->proj,
                       d->a_src, d->a_trg, d->s_src, d->s_trg, d->n_nodes, d->heads, d->fout);
  }
  TSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(gat_init_max_kernel, dim3(1), dim3(1), 0, s, d->gmax);
  TSG_LAUNCH_CHECK();
  const int mgrid = tsg_cdiv(d->n_nodes, 256) < 2048 ? tsg_cdiv(d->n_nodes, 256) : 2048;
  hipLaunchKernelGGL(gat_edge_max_kernel, dim3(mgrid), dim3(256), 0, s, d->in_ptr, d->in_src, d->s_src, d->s_trg,
                     d->gmax, d->n_nodes, d->heads);
  TSG_LAUNCH_CHECK();
  const dim3 agrid(tsg_cdiv(d->n_nodes, 4));
  if (d->fout % 256 == 0)
    hipLaunchKernelGGL(gat_aggregate_kernel<4>, agrid, dim3(256), 0, s, *d);
  else if (d->fout == 128)
    hipLaunchKernelGGL(gat_aggregate_kernel<2>, agrid, dim3(256), 0, s, *d);
  else
    hipLaunchKernelGGL(gat_aggregate_kernel<0>, agrid, dim3(256), 0, s, *d);
  TSG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tsg_gat_forward(const tsg_gat_desc* d, void* stream) { return gat_forward_impl(d, stream); }

// ------------------------------------------------------------------------------------ no-allocation variant
// tsg_gat_forward for callers that cannot allocate (include/tsg.h): the workspaces of the encoder (projection,
// node scores, global max, skip product, folded skip / attention weights) are carved out of one file-scope
// device array instead of coming in through the descriptor.  Every GPU holds its own copy of the array at its
// own address: the address is looked up once per device (hipGetSymbolAddress, with that device current) and
// kept; host code never passes the array's name to a kernel.
static __device__ __align__(256) unsigned char scratch[TSG_GAT_STATIC_SCRATCH_BYTES];

#define TSG_MAX_DEVICES 64
static void* g_scratch_ptr[TSG_MAX_DEVICES];
static std::mutex g_scratch_mu;

static int scratch_of_current_device(unsigned char** out) {
  int dev = -1;
  TSG_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= TSG_MAX_DEVICES) return TSG_E_LIMIT;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    TSG_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return TSG_E_INVALID;
    g_scratch_ptr[dev] = p;
  }
  *out = static_cast<unsigned char*>(g_scratch_ptr[dev]);
  return 0;
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" int tsg_gat_forward_static(const tsg_gat_desc* d, void* stream) {
  if (!d) return TSG_E_INVALID;
  if (d->n_nodes <= 0 || d->n_edges < 0 || d->fin <= 0 || d->heads <= 0 || d->fout <= 0) return TSG_E_INVALID;
  if (d->heads > 16 || d->fout > 64 * GAT_MAXJ) return TSG_E_LIMIT;
  const size_t N = (size_t)d->n_nodes, HF = (size_t)d->heads * d->fout;
  const size_t outw = d->concat ? HF : (size_t)d->fout;
  // the same workspaces as tsg_gat_desc documents, laid out back to back (256-byte aligned)
  const size_t proj_b = N * HF * 4, s_b = N * d->heads * 4;
  const size_t skip_b = d->skip_identity ? 0 : N * outw * 4;
  const size_t wmean_b = (d->skip_identity || d->concat) ? 0 : (size_t)d->fin * d->fout * 4;
  const size_t watt_b = (size_t)d->fin * 2 * d->heads * 4;
  size_t off = 0;
  const size_t o_proj = off;  off += align256(proj_b);
  const size_t o_ssrc = off;  off += align256(s_b);
  const size_t o_strg = off;  off += align256(s_b);
  const size_t o_gmax = off;  off += 256;
  const size_t o_skip = off;  off += align256(skip_b);
  const size_t o_wmean = off; off += align256(wmean_b);
  const size_t o_watt = off;  off += align256(watt_b);
  if (off > (size_t)TSG_GAT_STATIC_SCRATCH_BYTES) return TSG_E_LIMIT;  // graph too large for the static scratch
  unsigned char* base = nullptr;
  const int rc = scratch_of_current_device(&base);
  if (rc) return rc;
  tsg_gat_desc w = *d;
  w.proj = reinterpret_cast<float*>(base + o_proj);
  w.s_src = reinterpret_cast<float*>(base + o_ssrc);
  w.s_trg = reinterpret_cast<float*>(base + o_strg);
  w.gmax = reinterpret_cast<float*>(base + o_gmax);
  w.proj_bytes = proj_b;
  w.s_bytes = s_b;
  w.skip = skip_b ? reinterpret_cast<float*>(base + o_skip) : nullptr;
  w.skip_bytes = skip_b;
  w.w_skip_mean = wmean_b ? reinterpret_cast<float*>(base + o_wmean) : nullptr;
  w.w_skip_mean_bytes = wmean_b;
  w.w_att = reinterpret_cast<float*>(base + o_watt);
  w.w_att_bytes = watt_b;
  return gat_forward_impl(&w, stream);
}

__global__ __launch_bounds__(256) void gat_build_input_kernel(const float* __restrict__ nf, int n_nodes, int fraw,
                                                              const float* __restrict__ lon_embed, int n_lon,
                                                              const float* __restrict__ lat_embed, int n_lat, int gps,
                                                              float* __restrict__ x, int32_t* err_flag) {
  const int fin = 2 * gps + fraw - 2;
  const long total = (long)n_nodes * fin;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / fin), k = (int)(i % fin);
    float v;
    if (k < 2 * gps) {
      const bool is_lon = k < gps;
      int cell = (int)nf[(size_t)n * fraw + (is_lon ? fraw - 2 : fraw - 1)];  // .long() truncation
      const int lim = is_lon ? n_lon : n_lat;
      if (cell < 0 || cell >= lim) {
        *err_flag = 1;
        cell = 0;
      }
      v = is_lon ? lon_embed[(size_t)cell * gps + k] : lat_embed[(size_t)cell * gps + (k - gps)];
    } else {
      v = nf[(size_t)n * fraw + (k - 2 * gps)];
    }
    x[i] = v;
  }
}

extern "C" int tsg_gat_build_input(const float* node_features, int n_nodes, int fraw, const float* lon_embed, int n_lon,
                                   const float* lat_embed, int n_lat, int gps_dim, float* x, int32_t* err_flag,
                                   void* stream) {
  if (!node_features || !lon_embed || !lat_embed || !x || !err_flag) return TSG_E_INVALID;
  if (n_nodes <= 0 || fraw < 2 || n_lon <= 0 || n_lat <= 0 || gps_dim <= 0) return TSG_E_INVALID;
  const long total = (long)n_nodes * (2 * gps_dim + fraw - 2);
  const int grid = tsg_cdiv(total, 256) < 2048 ? tsg_cdiv(total, 256) : 2048;
  hipLaunchKernelGGL(gat_build_input_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, node_features, n_nodes,
                     fraw, lon_embed, n_lon, lat_embed, n_lat, gps_dim, x, err_flag);
  TSG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tsg_version(void) { return TSG_ABI_VERSION; }
