/* Host-side helper of the two-stage search (libtsg_host.so, plain C, no GPU): the border-road orders of
 * DoubleLayerSearcher.astar_search (search.py:897-902),
 *     np.random.choice(transfer_rid, size=len(transfer_rid), p=freq / freq.sum(), replace=False),
 * for many OD pairs at once, each OD pair drawing from its own numpy.random.RandomState(seed) as the batched
 * generate() and the goldens define it.  The uniforms are numpy's, bit for bit: MT19937 seeded by init_genrand
 * (RandomState(int)), 53-bit doubles from two outputs (random_sample), and the order is what numpy's legacy
 * choice(replace=False, p=...) builds: draw size - found uniforms, zero the found probabilities, cumsum, divide by
 * the last, searchsorted(side='right'), keep first occurrences in the order drawn; repeat until all are found.
 * Seeding one RandomState per OD pair and these draws were the host-side bottleneck of the two-stage path in Python.
 * tests/test_host.py compares every order with RandomState.choice. */
#include <stdint.h>
#include <stdlib.h>

#define MT_N 624
#define MT_M 397

typedef struct {
  uint32_t key[MT_N];
  int pos;
} mt_state;

static void mt_seed(mt_state* s, uint32_t seed) {
  for (int pos = 0; pos < MT_N; ++pos) {
    s->key[pos] = seed;
    seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)pos + 1u;
  }
  s->pos = MT_N;
}

static void mt_gen(mt_state* s) {
  uint32_t y;
  int i;
  for (i = 0; i < MT_N - MT_M; ++i) {
    y = (s->key[i] & 0x80000000u) | (s->key[i + 1] & 0x7fffffffu);
    s->key[i] = s->key[i + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
  }
  for (; i < MT_N - 1; ++i) {
    y = (s->key[i] & 0x80000000u) | (s->key[i + 1] & 0x7fffffffu);
    s->key[i] = s->key[i + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
  }
  y = (s->key[MT_N - 1] & 0x80000000u) | (s->key[0] & 0x7fffffffu);
  s->key[MT_N - 1] = s->key[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
  s->pos = 0;
}

static uint32_t mt_next(mt_state* s) {
  if (s->pos == MT_N) mt_gen(s);
  uint32_t y = s->key[s->pos++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

static double mt_double(mt_state* s) {
  const long a = (long)(mt_next(s) >> 5), b = (long)(mt_next(s) >> 6);
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* n_od OD pairs; OD i draws, from RandomState(seed[i]), one order per leg od_ptr[i] <= l < od_ptr[i+1]; leg l uses
 * the probability vector of pair leg_pair[l]: pair_p[pair_ptr[q] .. pair_ptr[q+1]) (already normalised by the
 * caller, as the reference passes it).  The order of leg l (indices into its pair's list) goes to
 * out_idx[out_ptr[l] .. out_ptr[l] + size).  Returns 0; -1 on bad arguments / allocation failure; -2 when a pair's
 * probability vector has fewer positive entries than its list has roads, or an entry that is negative or not a number
 * (numpy raises ValueError there: 'Fewer non-zero entries in p than size' / 'probabilities are not non-negative';
 * the draw loop below would otherwise never end). */
int tsg_host_border_orders(int32_t n_od, const uint32_t* seed, const int64_t* od_ptr, const int32_t* leg_pair,
                           const int64_t* pair_ptr, const double* pair_p, const int64_t* out_ptr, int32_t* out_idx,
                           int32_t max_size) {
  if (n_od < 0 || max_size <= 0 || !seed || !od_ptr || !leg_pair || !pair_ptr || !pair_p || !out_ptr || !out_idx) return -1;
  double* p = (double*)malloc(sizeof(double) * (size_t)max_size * 3);
  unsigned char* seen = (unsigned char*)malloc((size_t)max_size);
  if (!p || !seen) {
    free(p);
    free(seen);
    return -1;
  }
  double* cdf = p + max_size;
  double* x = cdf + max_size;
  mt_state st;
  for (int32_t i = 0; i < n_od; ++i) {
    mt_seed(&st, seed[i]);
    for (int64_t l = od_ptr[i]; l < od_ptr[i + 1]; ++l) {
      const int32_t q = leg_pair[l];
      const int size = (int)(pair_ptr[q + 1] - pair_ptr[q]);
      if (size <= 0 || size > max_size) {
        free(p);
        free(seen);
        return -1;
      }
      const double* pp = pair_p + pair_ptr[q];
      int32_t* found = out_idx + out_ptr[l];
      for (int k = 0; k < size; ++k) {
        p[k] = pp[k];
        seen[k] = 0;
      }
      int n_pos = 0, bad = 0;
      for (int k = 0; k < size; ++k) {
        if (!(pp[k] >= 0.0)) bad = 1; /* negative or NaN */
        if (pp[k] > 0.0) ++n_pos;
      }
      if (bad || n_pos < size) {
        free(p);
        free(seen);
        return -2;
      }
      int n_found = 0;
      while (n_found < size) {
        const int want = size - n_found;
        for (int k = 0; k < want; ++k) x[k] = mt_double(&st);
        for (int k = 0; k < n_found; ++k) p[found[k]] = 0.0;
        double acc = 0.0;
        for (int k = 0; k < size; ++k) {
          acc += p[k];
          cdf[k] = acc;
        }
        const double last = cdf[size - 1];
        for (int k = 0; k < size; ++k) cdf[k] /= last;
        for (int k = 0; k < want; ++k) {
          int lo = 0, hi = size; /* searchsorted(side='right'): first index with cdf[idx] > x */
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= x[k])
              lo = mid + 1;
            else
              hi = mid;
          }
          if (lo < size && !seen[lo]) {
            seen[lo] = 1;
            found[n_found++] = lo;
          }
        }
      }
    }
  }
  free(p);
  free(seen);
  return 0;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Hop distances of many OD pairs (synthetic workloads: default_len = BFS hop distance + 1; the
 * reference takes default_len from the length of the real trajectory, our_model_generate.py:418-424).  One
 * breadth-first search per OD pair over the successor CSR, stopped when the destination is reached; OD pair i goes
 * to thread i % n_threads (plain pthreads, all joined before the call returns: no thread pool is left behind for
 * a later fork()).  out[i] = number of transitions from origins[i] to dests[i], -1 when unreachable.
 * Returns 0, -1 on bad arguments (an id outside [0, n)) or allocation failure. */
#include <pthread.h>
#include <string.h>

typedef struct {
  int32_t n;
  const int32_t* indptr;
  const int32_t* indices;
  int64_t n_od;
  const int32_t* origins;
  const int32_t* dests;
  int32_t* out;
  int tid, n_threads, rc;
} hop_job;

static void* hop_worker(void* arg) {
  hop_job* j = (hop_job*)arg;
  const int32_t n = j->n;
  int32_t* stamp = (int32_t*)calloc((size_t)n, sizeof(int32_t)); /* epoch of the search that last reached a road */
  int32_t* queue = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  if (!stamp || !queue) {
    free(stamp);
    free(queue);
    j->rc = -1;
    return NULL;
  }
  int32_t epoch = 0;
  for (int64_t i = j->tid; i < j->n_od; i += j->n_threads) {
    const int32_t o = j->origins[i], d = j->dests[i];
    if (++epoch == INT32_MAX) {
      memset(stamp, 0, sizeof(int32_t) * (size_t)n);
      epoch = 1;
    }
    int32_t found = o == d ? 0 : -1;
    int32_t head = 0, tail = 0, level_end = 1, depth = 0;
    queue[tail++] = o;
    stamp[o] = epoch;
    while (found < 0 && head < tail) {
      if (head == level_end) {
        ++depth;
        level_end = tail;
      }
      const int32_t r = queue[head++];
      for (int32_t k = j->indptr[r]; k < j->indptr[r + 1]; ++k) {
        const int32_t c = j->indices[k];
        if (stamp[c] == epoch) continue;
        if (c == d) {
          found = depth + 1;
          break;
        }
        stamp[c] = epoch;
        queue[tail++] = c;
      }
    }
    j->out[i] = found;
  }
  free(stamp);
  free(queue);
  return NULL;
}

int tsg_host_hop_distance(int32_t n, const int32_t* indptr, const int32_t* indices, int64_t n_od,
                          const int32_t* origins, const int32_t* dests, int32_t* out, int32_t n_threads) {
  if (n <= 0 || n_od < 0 || !indptr || !indices || !origins || !dests || !out) return -1;
  for (int64_t i = 0; i < n_od; ++i)
    if (origins[i] < 0 || origins[i] >= n || dests[i] < 0 || dests[i] >= n) return -1;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 64) n_threads = 64;
  if ((int64_t)n_threads > n_od) n_threads = n_od > 0 ? (int32_t)n_od : 1;
  hop_job jobs[64];
  pthread_t th[64];
  int started = 0, rc = 0;
  for (int t = 0; t < n_threads; ++t) {
    jobs[t] = (hop_job){n, indptr, indices, n_od, origins, dests, out, t, n_threads, 0};
    if (pthread_create(&th[t], NULL, hop_worker, &jobs[t]) != 0) break;
    ++started;
  }
  for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
  for (int t = 0; t < started; ++t)
    if (jobs[t].rc) rc = -1;
  if (started < n_threads && rc == 0) { /* threads that could not be started: their share runs here */
    for (int t = started; t < n_threads; ++t) {
      hop_worker(&jobs[t]);
      if (jobs[t].rc) rc = -1;
    }
  }
  return rc;
}
