// GAT layer in TRAINING mode: forward with the three dropouts of generator/gat.py:149,155,180 as caller-supplied
// masks, and the backward of the sparse part (edge softmax + neighbourhood aggregation: an SDDMM for the attention
// gradients on the CSR-by-target graph, a transposed SpMM for the projected features on the CSR-by-source graph).
// The dense products around it (projection, skip projection and their weight / input gradients) are tsg_gemm_f32
// calls made by the host wrapper (ts-trajgen_amd/generator/gat.py).  One wave per node, lanes over the F features
// of a head, edges of a node visited in COO order (the order of the reference's scatter_add_).
#include "tsg_common.h"

extern "C" int tsg_gemm_f32(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C, int ldc,
                            int M, int N, int K, void* stream);

namespace {

__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  if (v >= 0.0f)
    atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else
    atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ void gt_mask_mul_kernel(const float* __restrict__ x, const float* __restrict__ m, float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = m ? x[i] * m[i] : x[i];
}

__global__ void gt_init_kernel(float* gmax) { *gmax = -INFINITY; }

// p_d = proj * m_proj (in place), s_src / s_trg = <p_d[n,h,:], a[h,:]>            (gat.py:155,164-165)
__global__ __launch_bounds__(256) void gt_scores_kernel(tsg_gat_train d) {
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= (long)d.n_nodes * d.heads) return;
  const int h = (int)(w % d.heads);
  float* p = d.p_d + (size_t)w * d.fout;
  const float* mk = d.m_proj ? d.m_proj + (size_t)w * d.fout : nullptr;
  float ps = 0.0f, pt = 0.0f;
  for (int f = lane; f < d.fout; f += 64) {
    float v = p[f];
    if (mk) {
      v *= mk[f];
      p[f] = v;
    }
    ps += v * d.a_src[h * d.fout + f];
    pt += v * d.a_trg[h * d.fout + f];
  }
  ps = tsg_wave_sum(ps);
  pt = tsg_wave_sum(pt);
  if (lane == 0) {
    d.s_src[w] = ps;
    d.s_trg[w] = pt;
  }
}

// global max of LeakyReLU(s_src[src] + s_trg[trg]) over all edges and heads      (gat.py:174,221)
__global__ __launch_bounds__(256) void gt_edge_max_kernel(tsg_gat_train d) {
  float m = -INFINITY;
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < d.n_nodes; n += gridDim.x * blockDim.x) {
    for (int j = d.in_ptr[n]; j < d.in_ptr[n + 1]; ++j) {
      const int s = d.in_src[j];
      for (int h = 0; h < d.heads; ++h) {
        const float v = d.s_src[(size_t)s * d.heads + h] + d.s_trg[(size_t)n * d.heads + h];
        m = fmaxf(m, v > 0.0f ? v : 0.2f * v);
      }
    }
  }
  m = tsg_wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > -INFINITY) atomic_max_f32(d.gmax, m);
}

// One wave per target node: edge softmax over its incoming edges, attention dropout, aggregation, skip term,
// head mean / concat, bias, ELU                                             (gat.py:177-199, 93-114)
__global__ __launch_bounds__(256) void gt_aggregate_kernel(tsg_gat_train d) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= d.n_nodes) return;
  const int j0 = d.in_ptr[n], j1 = d.in_ptr[n + 1];
  const int nh = d.heads, F = d.fout;
  const float gmax = *d.gmax;
  const int outw = d.concat ? nh * F : F;
  for (int f0 = 0; f0 < F; f0 += 64) {
    const int f = f0 + lane;
    float mean_acc = 0.0f;
    for (int h = 0; h < nh; ++h) {
      const float st = d.s_trg[(size_t)n * nh + h];
      // denominator: scatter_add_ in COO order                                  (gat.py:246)
      float denom = 0.0f;
      for (int j = j0; j < j1; ++j) {
        const float v = d.s_src[(size_t)d.in_src[j] * nh + h] + st;
        denom += __expf((v > 0.0f ? v : 0.2f * v) - gmax);
      }
      float acc = 0.0f;
      for (int j = j0; j < j1; ++j) {
        const int s = d.in_src[j];
        const float v = d.s_src[(size_t)s * nh + h] + st;
        const float att = __expf((v > 0.0f ? v : 0.2f * v) - gmax) / (denom + 1e-16f);  // gat.py:230
        const float att_d = d.m_att ? att * d.m_att[(size_t)j * nh + h] : att;           // gat.py:180
        if (f0 == 0 && lane == 0) {
          d.att[(size_t)j * nh + h] = att;
          d.att_d[(size_t)j * nh + h] = att_d;
        }
        if (f < F) acc += d.p_d[((size_t)s * nh + h) * F + f] * att_d;                   // gat.py:188,263
      }
      if (f < F) {
        acc += d.skip_identity ? d.x_d[(size_t)n * d.fin + f] : d.skip[((size_t)n * nh + h) * F + f];  // gat.py:97,101
        if (d.concat) {
          const float z = acc + (d.bias ? d.bias[h * F + f] : 0.0f);
          d.out[(size_t)n * outw + h * F + f] = z > 0.0f ? z : expm1f(z);
        } else {
          mean_acc += acc;
        }
      }
    }
    if (!d.concat && f < F) {
      const float z = mean_acc / (float)nh + (d.bias ? d.bias[f] : 0.0f);                // gat.py:109,112
      d.out[(size_t)n * outw + f] = z > 0.0f ? z : expm1f(z);
    }
  }
}

// ---- backward ---------------------------------------------------------------------------------------------
// d_o[n,h,f] = gradient of the per-head sum (aggregation + skip) : d_out * ELU'(z), spread over the heads
// (concat: its own columns; mean: 1 / NH of the node's row)
__global__ __launch_bounds__(256) void gt_bwd_out_kernel(tsg_gat_train d) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nh = d.heads, F = d.fout;
  if (i >= (long)d.n_nodes * nh * F) return;
  const int f = (int)(i % F), h = (int)((i / F) % nh);
  const long n = i / ((long)F * nh);
  const int outw = d.concat ? nh * F : F;
  const size_t o = (size_t)n * outw + (d.concat ? h * F + f : f);
  const float y = d.out[o];
  const float dz = d.d_out[o] * (y > 0.0f ? 1.0f : y + 1.0f);  // ELU'(z) = 1 (z > 0) or e^z = y + 1
  d.d_o[i] = d.concat ? dz : dz / (float)nh;
}

// column sums: d_bias[c] = sum_n d_out[n,c] * ELU'(z[n,c])
__global__ __launch_bounds__(256) void gt_bwd_bias_kernel(tsg_gat_train d) {
  const int outw = d.concat ? d.heads * d.fout : d.fout;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= outw) return;
  float s = 0.0f;
  for (int n = 0; n < d.n_nodes; ++n) {
    const float y = d.out[(size_t)n * outw + c];
    s += d.d_out[(size_t)n * outw + c] * (y > 0.0f ? 1.0f : y + 1.0f);
  }
  d.d_bias[c] = s;
}

// One wave per TARGET node and head: d_att_d[j] = <d_o[n,h,:], p_d[src_j,h,:]> (SDDMM), attention dropout,
// softmax backward over the neighbourhood, LeakyReLU backward; writes d_pre[j,h] and ds_trg[n,h]
__global__ __launch_bounds__(256) void gt_bwd_target_kernel(tsg_gat_train d) {
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int nh = d.heads, F = d.fout;
  if (w >= (long)d.n_nodes * nh) return;
  const int n = (int)(w / nh), h = (int)(w % nh);
  const int j0 = d.in_ptr[n], j1 = d.in_ptr[n + 1];
  const float* go = d.d_o + (size_t)w * F;
  float dot = 0.0f;  // sum_j att_j * d_att_j
  for (int j = j0; j < j1; ++j) {
    const float* p = d.p_d + ((size_t)d.in_src[j] * nh + h) * F;
    float s = 0.0f;
    for (int f = lane; f < F; f += 64) s += go[f] * p[f];
    s = tsg_wave_sum(s);
    const float datt = d.m_att ? s * d.m_att[(size_t)j * nh + h] : s;
    if (lane == 0) d.d_pre[(size_t)j * nh + h] = datt;  // parked until the neighbourhood's sum is known
    dot += d.att[(size_t)j * nh + h] * datt;
  }
  const float st = d.s_trg[(size_t)w];
  float ds_t = 0.0f;
  // lane 0's stores above are read back by all lanes below: the fence orders them (release at agent scope: the
  // stores have reached the L2 before any load that follows is issued), the loads go past the vector L1
  __threadfence();
  for (int j = j0 + lane; j < j1; j += 64) {
    const float att = d.att[(size_t)j * nh + h];
    const float datt = __hip_atomic_load(&d.d_pre[(size_t)j * nh + h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float de = att * (datt - dot);                                 // softmax over the neighbourhood
    const float pre = d.s_src[(size_t)d.in_src[j] * nh + h] + st;
    const float dp = de * (pre > 0.0f ? 1.0f : 0.2f);                    // LeakyReLU(0.2)
    d.d_pre[(size_t)j * nh + h] = dp;
    ds_t += dp;
  }
  ds_t = tsg_wave_sum(ds_t);
  if (lane == 0) d.ds_trg[w] = ds_t;
}

// One wave per SOURCE node and head: ds_src[m,h] = sum of d_pre over its outgoing edges;
// d_p[m,h,:] = (sum_j att_d[j,h] d_o[trg_j,h,:] + ds_src a_src[h,:] + ds_trg[m,h] a_trg[h,:]) * m_proj   (SpMM^T)
__global__ __launch_bounds__(256) void gt_bwd_source_kernel(tsg_gat_train d) {
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int nh = d.heads, F = d.fout;
  if (w >= (long)d.n_nodes * nh) return;
  const int m = (int)(w / nh), h = (int)(w % nh);
  const int k0 = d.out_ptr[m], k1 = d.out_ptr[m + 1];
  float ds_s = 0.0f;
  for (int k = k0 + lane; k < k1; k += 64) ds_s += d.d_pre[(size_t)d.out_slot[k] * nh + h];
  ds_s = tsg_wave_sum(ds_s);
  if (lane == 0) d.ds_src[w] = ds_s;
  const float ds_t = d.ds_trg[w];
  for (int f = lane; f < F; f += 64) {
    float acc = 0.0f;
    for (int k = k0; k < k1; ++k) {
      const int j = d.out_slot[k];
      acc += d.att_d[(size_t)j * nh + h] * d.d_o[((size_t)d.slot_trg[j] * nh + h) * F + f];
    }
    acc += ds_s * d.a_src[h * F + f] + ds_t * d.a_trg[h * F + f];
    d.d_p[(size_t)w * F + f] = d.m_proj ? acc * d.m_proj[(size_t)w * F + f] : acc;
  }
}

// d_a_src[h,f] = sum_n ds_src[n,h] p_d[n,h,f], d_a_trg likewise (one thread per (h, f))
__global__ __launch_bounds__(256) void gt_bwd_scoring_kernel(tsg_gat_train d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int nh = d.heads, F = d.fout;
  if (i >= nh * F) return;
  const int h = i / F;
  float a = 0.0f, b = 0.0f;
  for (int n = 0; n < d.n_nodes; ++n) {
    const float p = d.p_d[(size_t)n * nh * F + i];
    a += d.ds_src[(size_t)n * nh + h] * p;
    b += d.ds_trg[(size_t)n * nh + h] * p;
  }
  d.d_a_src[i] = a;
  d.d_a_trg[i] = b;
}

int check_train(const tsg_gat_train* d, bool backward) {
  if (!d) return TSG_E_INVALID;
  if (d->n_nodes <= 0 || d->n_edges < 0 || d->fin <= 0 || d->heads <= 0 || d->fout <= 0) return TSG_E_INVALID;
  if (d->heads > 16 || d->fout > 512) return TSG_E_LIMIT;
  if (!d->in_ptr || !d->in_src || !d->x || !d->w_proj_t || !d->a_src || !d->a_trg) return TSG_E_INVALID;
  if (!d->x_d || !d->p_d || !d->s_src || !d->s_trg || !d->att || !d->att_d || !d->gmax || !d->out) return TSG_E_INVALID;
  if (!d->skip_identity && (!d->w_skip_t || !d->skip)) return TSG_E_INVALID;
  if (d->skip_identity && d->fin != d->fout) return TSG_E_INVALID;
  const size_t N = (size_t)d->n_nodes, NHF = (size_t)d->heads * d->fout;
  const size_t outw = d->concat ? NHF : (size_t)d->fout;
  if (d->node_feat_bytes < N * NHF * 4 || d->in_bytes < N * d->fin * 4 || d->out_bytes < N * outw * 4) return TSG_E_TOO_SMALL;
  if (d->edge_bytes < (size_t)d->n_edges * d->heads * 4 || d->score_bytes < N * d->heads * 4) return TSG_E_TOO_SMALL;
  if (backward) {
    if (!d->out_ptr || !d->out_slot || !d->slot_trg || !d->d_out) return TSG_E_INVALID;
    if (!d->d_o || !d->d_p || !d->d_pre || !d->ds_src || !d->ds_trg || !d->d_a_src || !d->d_a_trg) return TSG_E_INVALID;
    if (d->bias && !d->d_bias) return TSG_E_INVALID;
  }
  return 0;
}

}  // namespace

extern "C" int tsg_gat_train_forward(const tsg_gat_train* d, void* stream) {
  int rc = check_train(d, false);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const long nin = (long)d->n_nodes * d->fin;
  const int NHF = d->heads * d->fout;
  hipLaunchKernelGGL(gt_mask_mul_kernel, dim3(tsg_cdiv(nin, 256)), dim3(256), 0, s, d->x, d->m_in, d->x_d, nin);
  TSG_LAUNCH_CHECK();
  rc = tsg_gemm_f32(d->x_d, d->fin, d->w_proj_t, NHF, nullptr, d->p_d, NHF, d->n_nodes, NHF, d->fin, stream);
  if (rc) return rc;
  if (!d->skip_identity) {
    rc = tsg_gemm_f32(d->x_d, d->fin, d->w_skip_t, NHF, nullptr, d->skip, NHF, d->n_nodes, NHF, d->fin, stream);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(gt_scores_kernel, dim3(tsg_cdiv((long)d->n_nodes * d->heads, 4)), dim3(256), 0, s, *d);
  hipLaunchKernelGGL(gt_init_kernel, dim3(1), dim3(1), 0, s, d->gmax);
  hipLaunchKernelGGL(gt_edge_max_kernel, dim3(tsg_cdiv(d->n_nodes, 256) < 1024 ? tsg_cdiv(d->n_nodes, 256) : 1024),
                     dim3(256), 0, s, *d);
  hipLaunchKernelGGL(gt_aggregate_kernel, dim3(tsg_cdiv(d->n_nodes, 4)), dim3(256), 0, s, *d);
  TSG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tsg_gat_train_backward(const tsg_gat_train* d, void* stream) {
  int rc = check_train(d, true);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const long nhf = (long)d->n_nodes * d->heads * d->fout;
  const int outw = d->concat ? d->heads * d->fout : d->fout;
  hipLaunchKernelGGL(gt_bwd_out_kernel, dim3(tsg_cdiv(nhf, 256)), dim3(256), 0, s, *d);
  if (d->bias) hipLaunchKernelGGL(gt_bwd_bias_kernel, dim3(tsg_cdiv(outw, 256)), dim3(256), 0, s, *d);
  hipLaunchKernelGGL(gt_bwd_target_kernel, dim3(tsg_cdiv((long)d->n_nodes * d->heads, 4)), dim3(256), 0, s, *d);
  hipLaunchKernelGGL(gt_bwd_source_kernel, dim3(tsg_cdiv((long)d->n_nodes * d->heads, 4)), dim3(256), 0, s, *d);
  hipLaunchKernelGGL(gt_bwd_scoring_kernel, dim3(tsg_cdiv(d->heads * d->fout, 256)), dim3(256), 0, s, *d);
  TSG_LAUNCH_CHECK();
  return 0;
}
