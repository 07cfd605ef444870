// zk_graph.hip -- the manifold consumer of the moment matrix: ForceGraph8, reference
// manifold/force_relaxed.py:285-366.  Its stages and where they run here:
//   compute_graph (:67-86)          k nearest neighbours of every row under the correlation distance (sklearn NearestNeighbors,
//                                   brute force) -- the O(N^2 D) part -- on the device: zk_rows_knn_correlation
//   calculate_asymmetric_Pij (:17-52)  per-row bisection for the neighbour weights: zk_knn_affinities (device)
//   optimize_layout / optimize_stage (:236-282)  the force-directed optimiser.  ONE sequential loop by construction (every
//                                   pair update moves two nodes the next pair reads; the repulsion partners come from one running
//                                   tau_rand_int state): compiled host code in the reference (numba) and compiled host code here,
//                                   zk_force_layout_stage, operation for operation.
// The sparse symmetrisation between them is SciPy's, as in the reference (mtflearn_amd/manifold.py).
#include "zk_internal.h"
#include "zk_fold.h"  // ZK_CONST / zk_const

#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

// the resident matrix (zk_cluster.hip)
extern "C" const double* zk_rows_data(const zk_rows* m);
int zk_rows_shape(const zk_rows* m, int* device, int64_t* n_rows, int* n_features, void** stream);

namespace {

// z_r = (x_r - mean(x_r)) / |x_r - mean(x_r)|  (rows with no variation: zeros), written row-major (N, D) and feature-major
// (D, Np) -- the second copy is what the candidates' coordinates are read from as scalar operands, eight candidates at a time
__global__ __launch_bounds__(256) void unit_rows_kernel(const double* __restrict__ X, long long N, int D, long long Np,
                                                        double* __restrict__ Z, double* __restrict__ Zt) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= Np) return;
  if (r >= N) {
    for (int f = 0; f < D; ++f) Zt[(long long)f * Np + r] = 0.0;
    return;
  }
  const double* x = X + r * D;
  double s = 0.0;
  for (int f = 0; f < D; ++f) s += x[f];
  const double mean = s / D;
  double q = 0.0;
  for (int f = 0; f < D; ++f) {
    const double v = x[f] - mean;
    q += v * v;
  }
  const double inv = q > 0.0 ? 1.0 / sqrt(q) : 0.0;
  for (int f = 0; f < D; ++f) {
    const double v = (x[f] - mean) * inv;
    Z[r * D + f] = v;
    Zt[(long long)f * Np + r] = v;
  }
}

// One wave = 64 query rows (lane = query, its unit row in LDS at stride D); the candidates go by in blocks of eight whose
// coordinates are wave-uniform scalar operands (Zt); every lane keeps its K best (distance, index) pairs sorted in registers.
// A candidate enters the insertion code only when some lane of the wave wants it (after the first few hundred candidates
// almost none does).  Ties keep the smaller index first.
// (Tried: the one-step-ahead scalar-operand pipelining of the mixture E step with 16-candidate blocks -- 85 -> 136 ms per
//  100 000 x 45: the candidate table is a 36-MB stream, each request pays an L2 round trip and one wave cannot keep enough of
//  them in flight in its SGPRs; the compiler's four-loads-then-wait schedule of this form is better.)
// WAVES waves share the query tile and split the candidates into WAVES consecutive ranges (more waves per CU: the candidate
// table is a stream through the scalar data path and one wave per 23-KiB tile cannot hide its latency -- 100 000 x 45, k = 10:
// 85 ms with one wave per tile); afterwards the waves hand their lists to wave 0 through LDS, one at a time in range order, and
// wave 0 inserts them (ranges ascend, so ties still keep the smaller index first).
template <int K, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void knn_kernel(const double* __restrict__ Z, const double* __restrict__ Zt, long long N, int D,
                                                         long long Np, int k, long long* __restrict__ ind, double* __restrict__ dist) {
  extern __shared__ __attribute__((aligned(16))) double tile[];  // [64][D], then the hand-over area [k][64] x (double, int64)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long q0 = (long long)blockIdx.x * 64;
  const long long total = N * D;
  for (int e = threadIdx.x; e < 64 * D; e += 64 * WAVES) tile[e] = q0 * D + e < total ? Z[q0 * D + e] : 0.0;
  __syncthreads();
  const double* row = tile + lane * D;
  double bd[K];
  long long bi[K];
#pragma unroll
  for (int p = 0; p < K; ++p) bd[p] = std::numeric_limits<double>::infinity(), bi[p] = -1;
  double worst = std::numeric_limits<double>::infinity();
  auto insert = [&](double cd, long long ci) {
#pragma unroll
    for (int p = 0; p < K; ++p) {
      const bool lt = cd < bd[p];
      const double td = bd[p];
      const long long ti = bi[p];
      bd[p] = lt ? cd : td;
      bi[p] = lt ? ci : ti;
      cd = lt ? td : cd;
      ci = lt ? ti : ci;
    }
#pragma unroll
    for (int p = 0; p < K; ++p)
      if (p == k - 1) worst = bd[p];
  };
  const ZK_CONST double* zt = zk_const(Zt);
  const long long span = ((Np / 8 + WAVES - 1) / WAVES) * 8;  // candidates per wave, a multiple of eight
  const long long j_lo = wave * span, j_hi = j_lo + span < Np ? j_lo + span : Np;
  for (long long j0 = j_lo; j0 < j_hi; j0 += 8) {
    double acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.0;
#pragma unroll 4
    for (int f = 0; f < D; ++f) {
      const double x = row[f];
      const ZK_CONST double* zc = zt + (long long)f * Np + j0;
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = __builtin_fma(x, zc[c], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const long long j = j0 + c;
      const double cd = 1.0 - acc[c];
      if (j < N && __ballot(cd < worst)) insert(cd, j);
    }
  }
  if constexpr (WAVES > 1) {
    double* hd = tile + 64 * D;
    long long* hi = (long long*)(hd + K * 64);
    for (int w = 1; w < WAVES; ++w) {
      __syncthreads();  // the hand-over area is free (and, the first time, every wave has finished its scan)
      if (wave == w) {
#pragma unroll
        for (int p = 0; p < K; ++p) hd[p * 64 + lane] = bd[p], hi[p * 64 + lane] = bi[p];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int p = 0; p < K; ++p) {
          const double cd = hd[p * 64 + lane];
          if (p < k && __ballot(cd < worst)) insert(cd, hi[p * 64 + lane]);
        }
      }
    }
    if (wave != 0) return;
  }
  const long long q = q0 + lane;
  if (q < N) {
#pragma unroll
    for (int p = 0; p < K; ++p)
      if (p < k) {
        ind[q * k + p] = bi[p];
        dist[q * k + p] = bd[p];
      }
  }
}

// calculate_asymmetric_Pij (reference force_relaxed.py:17-52), one thread per row of the (N, k) neighbour distances
__global__ __launch_bounds__(256) void affinity_kernel(const double* __restrict__ dist, long long N, int k, int local_connectivity,
                                                       double target, double* __restrict__ P) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= N) return;
  const double* d = dist + r * k;
  const double rho = d[local_connectivity];
  double beta_min = 0.0, beta_max = std::numeric_limits<double>::infinity(), beta = 1.0, chosen = 1.0;
  for (int n = 0; n < 100; ++n) {
    double sum = 0.0;
    for (int j = 1; j < k; ++j) {
      double v = d[j] - rho;
      v = v < 0.0 ? 0.0 : v;
      sum += exp(-v * beta);
    }
    if (fabs(sum - target) < 1e-5) {
      chosen = beta;
      break;
    }
    if (sum - target > 0.0) {
      beta_min = beta;
      beta = beta_max == std::numeric_limits<double>::infinity() ? beta * 2.0 : (beta + beta_max) / 2.0;
    } else {
      beta_max = beta;
      beta = (beta + beta_min) / 2.0;
    }
    chosen = beta;  // n == 99: the reference keeps the value after the last update
  }
  for (int j = 0; j < k; ++j) {
    double v = d[j] - rho;
    v = v < 0.0 ? 0.0 : v;
    double p = exp(-v * chosen);
    p = p < 2.220446049250313e-16 ? 2.220446049250313e-16 : p;
    P[r * k + j] = j == 0 ? 0.0 : p;
  }
}

}  // namespace

// k nearest neighbours (self included, as scikit-learn returns them for the training set) of every row of the resident matrix
// under the correlation distance 1 - corr(x_i, x_j); ind_out / dist_out (N, k) on the host, sorted by distance.  k <= 64.
// With P_out also the neighbour weights of calculate_asymmetric_Pij (perplexity = k as ForceGraph8 passes it).
extern "C" int zk_rows_knn_correlation(zk_rows* m, int k, int local_connectivity, double perplexity, int64_t* ind_out, double* dist_out,
                                       double* P_out) {
  if (!m || !ind_out || !dist_out) return zk_fail(ZK_E_BADARG, "null pointer");
  int device = 0, D = 0;
  int64_t N = 0;
  void* stream_v = nullptr;
  int rc = zk_rows_shape(m, &device, &N, &D, &stream_v);
  if (rc) return rc;
  if (k < 1 || k > 64 || k > N) return zk_fail(ZK_E_BADARG, "need 1 <= n_neighbors <= min(64, n_samples)");
  if (P_out && (local_connectivity < 0 || local_connectivity >= k || !(perplexity > 0.0))) return zk_fail(ZK_E_BADARG, "bad affinity parameters");
  if ((size_t)64 * D * sizeof(double) + 16 * 64 * 16 > 64 * 1024) return zk_fail(ZK_E_BADARG, "too many features for the query tile");
  ZK_ON_DEVICE(device);
  hipStream_t s = (hipStream_t)stream_v;
  const long long Np = (N + 7) & ~7LL;
  double *Z = nullptr, *Zt = nullptr, *d_dist = nullptr, *d_P = nullptr;
  long long* d_ind = nullptr;
  auto cleanup = [&]() {
    for (void* p : {(void*)Z, (void*)Zt, (void*)d_dist, (void*)d_P, (void*)d_ind})
      if (p) (void)hipFree(p);
  };
  hipError_t e = hipMalloc((void**)&Z, (size_t)N * D * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&Zt, (size_t)Np * D * sizeof(double) + 256);
  if (e == hipSuccess) e = hipMalloc((void**)&d_dist, (size_t)N * k * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&d_ind, (size_t)N * k * sizeof(long long));
  if (e == hipSuccess && P_out) e = hipMalloc((void**)&d_P, (size_t)N * k * sizeof(double));
  if (e != hipSuccess) {
    cleanup();
    return zk_hip_fail(e, "hipMalloc(kNN buffers)");
  }
  hipLaunchKernelGGL(unit_rows_kernel, dim3((unsigned)((Np + 255) / 256)), dim3(256), 0, s, zk_rows_data(m), (long long)N, D, Np, Z, Zt);
  const size_t lds = (size_t)64 * D * sizeof(double);
  const unsigned grid = (unsigned)((N + 63) / 64);
  if (k <= 16)
    hipLaunchKernelGGL((knn_kernel<16, 4>), dim3(grid), dim3(256), lds + (size_t)16 * 64 * 16, s, Z, Zt, (long long)N, D, Np, k, d_ind, d_dist);
  else if (k <= 32)
    hipLaunchKernelGGL((knn_kernel<32, 1>), dim3(grid), dim3(64), lds, s, Z, Zt, (long long)N, D, Np, k, d_ind, d_dist);
  else
    hipLaunchKernelGGL((knn_kernel<64, 1>), dim3(grid), dim3(64), lds, s, Z, Zt, (long long)N, D, Np, k, d_ind, d_dist);
  if (P_out)
    hipLaunchKernelGGL(affinity_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, d_dist, (long long)N, k, local_connectivity,
                       std::log2(perplexity), d_P);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(ind_out, d_ind, (size_t)N * k * sizeof(long long), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(dist_out, d_dist, (size_t)N * k * sizeof(double), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess && P_out) e = hipMemcpyAsync(P_out, d_P, (size_t)N * k * sizeof(double), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  cleanup();
  return e == hipSuccess ? 0 : zk_hip_fail(e, "zk_rows_knn_correlation");
}

// ------------------------------------------------------------------------------------------------------------------
// the layout optimiser: host code, like the reference's (numba-compiled) optimize_stage -- see the file header
// ------------------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)  // a * b + c stays two roundings, as in the reference's (and the oracle's) arithmetic

namespace {

inline double clip4(double v) { return v > 4.0 ? 4.0 : (v < -4.0 ? -4.0 : v); }

// tau_rand_int (reference force_relaxed.py:211-233) on its int64[3] state; returns the int32 it returns
inline int32_t tau_rand_int(int64_t* st) {
  auto shl = [](int64_t v, int n) { return (int64_t)((uint64_t)v << n); };
  st[0] = ((shl(st[0] & 4294967294LL, 12)) & 0xFFFFFFFFLL) ^ (((shl(st[0], 13) & 0xFFFFFFFFLL) ^ st[0]) >> 19);
  st[1] = ((shl(st[1] & 4294967288LL, 4)) & 0xFFFFFFFFLL) ^ (((shl(st[1], 2) & 0xFFFFFFFFLL) ^ st[1]) >> 25);
  st[2] = ((shl(st[2] & 4294967280LL, 17)) & 0xFFFFFFFFLL) ^ (((shl(st[2], 3) & 0xFFFFFFFFLL) ^ st[2]) >> 11);
  return (int32_t)(uint32_t)((uint64_t)(st[0] ^ st[1] ^ st[2]) & 0xFFFFFFFFULL);
}

}  // namespace

// optimize_stage (reference force_relaxed.py:236-266): num_iterations sweeps over the pairs (node1, node2, weight), each an
// attraction step followed by num_negative_samples repulsion draws; xy (n_nodes, 2) and rng_state (3) are updated in place;
// with log_out, xy after every sweep is appended (num_iterations x n_nodes x 2).  Pure host arithmetic: no device involved.
extern "C" int zk_force_layout_stage(double* xy, int64_t n_nodes, const int64_t* node1, const int64_t* node2, const double* weight,
                                     int64_t n_pairs, const int64_t* nbrs_ind, int n_neighbors, int64_t num_iterations,
                                     const double* force_params /* N, M, alpha, beta */, int num_negative_samples,
                                     double learning_rate, int64_t* rng_state, double* log_out) {
  if (!xy || !rng_state || !force_params || n_nodes <= 0 || n_pairs < 0 || (n_pairs && (!node1 || !node2 || !weight)) || !nbrs_ind ||
      n_neighbors < 0 || num_iterations < 0 || num_negative_samples < 0)
    return zk_fail(ZK_E_BADARG, "bad arguments");
  for (int64_t p = 0; p < n_pairs; ++p)
    if (node1[p] < 0 || node1[p] >= n_nodes || node2[p] < 0 || node2[p] >= n_nodes) return zk_fail(ZK_E_BADARG, "pair index out of range");
  const double N = force_params[0], M = force_params[1], alpha = force_params[2], beta = force_params[3];
  double lr = learning_rate;
  for (int64_t n = 0; n < num_iterations; ++n) {
    for (int64_t p = 0; p < n_pairs; ++p) {
      const double w = weight[p];
      double* a = xy + 2 * node1[p];
      double* b = xy + 2 * node2[p];
      {  // apply_attraction_force (:174-184)
        const double xd = a[0] - b[0], yd = a[1] - b[1];
        const double dist = std::hypot(xd, yd);
        const double force = alpha / (std::pow(dist, N) + 1.0);
        const double fx = clip4(xd * force) * lr * w, fy = clip4(yd * force) * lr * w;
        a[0] -= fx;
        a[1] -= fy;
        b[0] += fx;
        b[1] += fy;
      }
      const int64_t* nb = nbrs_ind + node1[p] * n_neighbors;
      for (int i = 0; i < num_negative_samples; ++i) {
        int64_t r = (int64_t)tau_rand_int(rng_state) % n_nodes;
        if (r < 0) r += n_nodes;  // Python's modulo
        bool repel = true;
        for (int q = 0; q < n_neighbors; ++q)
          if (nb[q] == r) repel = false;
        if (repel) {  // apply_repulsion_force (:187-197)
          double* c = xy + 2 * r;
          const double xd = a[0] - c[0], yd = a[1] - c[1];
          const double dist = std::hypot(xd, yd);
          const double force = beta / (std::pow(dist, M) + 1.0);
          const double fx = clip4(xd * force) * lr, fy = clip4(yd * force) * lr;
          a[0] += fx;
          a[1] += fy;
          c[0] -= fx;
          c[1] -= fy;
        }
      }
    }
    lr = learning_rate * (1.0 - (double)n / (double)num_iterations);
    if (log_out) memcpy(log_out + (size_t)n * n_nodes * 2, xy, (size_t)n_nodes * 2 * sizeof(double));
  }
  return 0;
}

// numpy.random.RandomState.choice(n, p = ones(n) / n) given its one uniform draw u, without the three n-sized arrays it builds:
// cdf_i = fl(sum of i + 1 copies of fl(1 / n)) / cdf_last (numpy.cumsum adds sequentially), index = number of cdf_i <= u
// (searchsorted side='right').  Two passes of dependent additions, no memory.  Host arithmetic; scikit-learn's k-means++
// takes its first seed this way, on 4 M rows 30 ms of NumPy against 8 ms here.
extern "C" int zk_uniform_choice_index(int64_t n, double u, int64_t* index_out) {
  if (n <= 0 || !index_out) return zk_fail(ZK_E_BADARG, "bad arguments");
  const double c = 1.0 / (double)n;  // ones(n) / ones(n).sum(): n is exact in float64 below 2^53
  double last = 0.0;
  for (int64_t i = 0; i < n; ++i) last += c;
  // the answer is near u * n: walk the running sum to a little before it, then count on
  int64_t guess = (int64_t)(u * (double)n) - 4;
  if (guess < 0) guess = 0;
  double s = 0.0;
  int64_t i = 0;
  for (; i < guess; ++i) s += c;
  // (monotone: once cdf_i > u every later one is too; entries before `guess` are <= u unless rounding moved the boundary by
  //  more than four steps, which the check below catches)
  if (i > 0 && s / last > u) {  // never seen; fall back to a full scan
    s = 0.0;
    i = 0;
  }
  for (; i < n; ++i) {
    s += c;
    if (s / last > u) break;
  }
  *index_out = i;
  return 0;
}
