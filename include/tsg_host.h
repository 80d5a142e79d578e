/* Host-side helper of the two-stage generation path (libtsg_host.so: plain C, no GPU, built from
 * ts-trajgen_amd/csrc/tsg_host.c by the same Makefile as libtsg.so).
 *
 * Replaces, for all legs of all OD pairs at once, the draw the reference makes per leg in
 * DoubleLayerSearcher.astar_search (search.py:900-902):
 *     np.random.choice(transfer_rid, size=len(transfer_rid), replace=False, p=transfer_freq / sum)
 * under np.random.seed(od_id) (our_model_generate.py:417-426 seeds per OD pair; the goldens G7 define the result).
 * The implementation restates numpy's legacy generator bit for bit: MT19937 seeded by init_genrand(seed),
 * random_sample() = (a >> 5, b >> 6) -> 53-bit double, and choice(replace=False, p)'s loop (cumulative sum,
 * searchsorted(side='right'), unique new indices kept in first-occurrence order, found ones zeroed, renormalise);
 * tests/test_host.py checks it against numpy.random.RandomState.choice on ragged probability vectors.
 */
#ifndef TSG_HOST_H
#define TSG_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* n_od OD pairs.  OD i owns the legs od_ptr[i] <= l < od_ptr[i+1] and draws them in that order from
 * RandomState(seed[i]) (leg k of an OD pair takes the k-th draw of its generator).  Leg l draws from the
 * probability vector of region pair q = leg_pair[l]: pair_p[pair_ptr[q] .. pair_ptr[q+1]), already normalised by
 * the caller as the reference passes it.  The drawn order of leg l (indices into its pair's list) is written to
 * out_idx[out_ptr[l] .. out_ptr[l] + size_l), size_l = pair_ptr[q+1] - pair_ptr[q] <= max_size.
 * Returns 0; -1 on bad arguments or when the scratch of 25 * max_size bytes cannot be allocated; -2 when a
 * probability vector has fewer positive entries than roads or a negative / NaN entry (numpy raises ValueError). */
int tsg_host_border_orders(int32_t n_od, const uint32_t* seed, const int64_t* od_ptr, const int32_t* leg_pair,
                           const int64_t* pair_ptr, const double* pair_p, const int64_t* out_ptr, int32_t* out_idx,
                           int32_t max_size);

/* Hop distances of many OD pairs over the successor CSR (indptr [n+1], indices): out[i] = transitions on a shortest
 * path origins[i] -> dests[i], -1 when unreachable.  Only the synthetic workloads need it (default_len
 * = hop distance + 1; the reference's driver takes default_len from the real trajectory's length,
 * our_model_generate.py:418-424), so it has no counterpart in the reference.  n_threads pthreads (1..64), all joined
 * before the call returns.  Returns 0, or -1 on an id outside [0, n) or allocation failure. */
int tsg_host_hop_distance(int32_t n, const int32_t* indptr, const int32_t* indices, int64_t n_od,
                          const int32_t* origins, const int32_t* dests, int32_t* out, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif
