// GAT road-graph encoder (generator/gat.py:138-200 + :84-114, eval mode) on a CSR-by-target graph:
//   1. projection  proj = x . W^T                      (MFMA GEMM, tsg_dense.hip)
//   2. node scores s_src/s_trg = <proj[n,h,:], a[h,:]> (one wave per (node, head))
//   3. global max of LeakyReLU(s_src[src] + s_trg[trg]) over all edges and heads   (gat.py:221)
//   4. per target node: segmented edge softmax (wave = head) + SpMM aggregation of the gathered
//      source rows + skip projection + head mean/concat + bias + ELU, all in one pass.
// Incoming edges are visited in their original COO order, so the per-target sums are accumulated
// in the same order as the reference's sequential scatter_add_.
#include "tsg_common.h"

extern "C" int tsg_gemm_f32(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C, int ldc,
                            int M, int N, int K, void* stream);

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
// one block per target node; wave h = head h (blockDim = 64 * heads)
__global__ void gat_aggregate_kernel(tsg_gat_desc d) {
  extern __shared__ float lds[];  // [heads][fout] head outputs, then [fin] x row
  const int n = blockIdx.x;
  const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int F = d.fout, NH = d.heads, FIN = d.fin;
  float* head_out = lds;
  float* xrow = lds + (size_t)NH * F;
  for (int k = threadIdx.x; k < FIN; k += blockDim.x) xrow[k] = d.x[(size_t)n * FIN + k];
  const int e0 = d.in_ptr[n], e1 = d.in_ptr[n + 1];
  const float gmax = *d.gmax;
  const float st = d.s_trg[(size_t)n * NH + h];
  // segmented softmax denominator over the incoming edges of n (gat.py:225,246)
  float denom = 0.0f;
  for (int e = e0; e < e1; ++e) {
    const int s = d.in_src[e];
    denom += expf(leaky02(d.s_src[(size_t)s * NH + h] + st) - gmax);
  }
  const float dinv_base = denom + 1e-16f;  // gat.py:230
  float acc[GAT_MAXJ];
#pragma unroll
  for (int j = 0; j < GAT_MAXJ; ++j) acc[j] = 0.0f;
  for (int e = e0; e < e1; ++e) {
    const int s = d.in_src[e];
    const float att = expf(leaky02(d.s_src[(size_t)s * NH + h] + st) - gmax) / dinv_base;
    const float* p = d.proj + ((size_t)s * NH + h) * F;
#pragma unroll
    for (int j = 0; j < GAT_MAXJ; ++j) {
      const int f = lane + 64 * j;
      if (f < F) acc[j] = __fadd_rn(acc[j], __fmul_rn(p[f], att));  // gat.py:188,263 (mul then add, no fma)
    }
  }
  __syncthreads();  // xrow ready
#pragma unroll
  for (int j = 0; j < GAT_MAXJ; ++j) {
    const int f = lane + 64 * j;
    if (f < F) {
      float sk;
      if (d.skip_identity) {
        sk = xrow[f];  // gat.py:94-97
      } else {
        sk = 0.0f;    // gat.py:101 skip_proj
        const float* w = d.w_skip_t + (size_t)h * F + f;
        for (int k = 0; k < FIN; ++k) sk += xrow[k] * w[(size_t)k * NH * F];
      }
      head_out[h * F + f] = acc[j] + sk;
    }
  }
  __syncthreads();
  if (d.concat) {
    for (int i = threadIdx.x; i < NH * F; i += blockDim.x) {
      float v = head_out[i] + (d.bias ? d.bias[i] : 0.0f);
      d.out[(size_t)n * NH * F + i] = v > 0.0f ? v : expm1f(v);
    }
  } else {
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
      float sum = 0.0f;
      for (int hh = 0; hh < NH; ++hh) sum += head_out[hh * F + f];
      float v = sum / (float)NH + (d.bias ? d.bias[f] : 0.0f);  // gat.py:109,112
      d.out[(size_t)n * F + f] = v > 0.0f ? v : expm1f(v);        // gat.py:114 ELU
    }
  }
}

extern "C" int tsg_gat_forward(const tsg_gat_desc* d, void* stream) {
  if (!d) return TSG_E_INVALID;
  if (d->n_nodes <= 0 || d->n_edges < 0 || d->fin <= 0 || d->heads <= 0 || d->fout <= 0) return TSG_E_INVALID;
  if (d->heads > 16 || d->fout > 64 * GAT_MAXJ) return TSG_E_LIMIT;
  if (!d->x || !d->w_proj_t || !d->a_src || !d->a_trg || !d->in_ptr || !d->in_src || !d->proj || !d->s_src ||
      !d->s_trg || !d->gmax || !d->out)
    return TSG_E_INVALID;
  if (!d->skip_identity && !d->w_skip_t) return TSG_E_INVALID;
  if (d->skip_identity && d->fin != d->fout) return TSG_E_INVALID;
  const size_t N = (size_t)d->n_nodes, HF = (size_t)d->heads * d->fout;
  if (d->proj_bytes < N * HF * 4 || d->s_bytes < N * d->heads * 4) return TSG_E_TOO_SMALL;
  if (d->out_bytes < N * (d->concat ? HF : (size_t)d->fout) * 4) return TSG_E_TOO_SMALL;
  hipStream_t s = (hipStream_t)stream;
  int rc = tsg_gemm_f32(d->x, d->fin, d->w_proj_t, (int)HF, nullptr, d->proj, (int)HF, d->n_nodes, (int)HF, d->fin,
                        stream);
  if (rc) return rc;
  hipLaunchKernelGGL(gat_node_scores_kernel, dim3(tsg_cdiv((long)N * d->heads, 4)), dim3(256), 0, s, d->proj, d->a_src,
                     d->a_trg, d->s_src, d->s_trg, d->n_nodes, d->heads, d->fout);
  TSG_LAUNCH_CHECK();
  hipLaunchKernelGGL(gat_init_max_kernel, dim3(1), dim3(1), 0, s, d->gmax);
  TSG_LAUNCH_CHECK();
  const int mgrid = tsg_cdiv(d->n_nodes, 256) < 2048 ? tsg_cdiv(d->n_nodes, 256) : 2048;
  hipLaunchKernelGGL(gat_edge_max_kernel, dim3(mgrid), dim3(256), 0, s, d->in_ptr, d->in_src, d->s_src, d->s_trg,
                     d->gmax, d->n_nodes, d->heads);
  TSG_LAUNCH_CHECK();
  const size_t lds = (HF + (size_t)d->fin) * 4;
  if (lds > 64 * 1024) return TSG_E_LIMIT;
  hipLaunchKernelGGL(gat_aggregate_kernel, dim3(d->n_nodes), dim3(64 * d->heads), lds, s, *d);
  TSG_LAUNCH_CHECK();
  return 0;
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
