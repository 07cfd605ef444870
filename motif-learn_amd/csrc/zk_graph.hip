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
    // (the test uses the NEW value at every position, not the entry being pushed down: an entry displaced from a run of equal
    //  distances must keep moving, or it would be overtaken by its equals and ties would lose their index order)
    const double nd = cd;
#pragma unroll
    for (int p = 0; p < K; ++p) {
      const bool lt = nd < bd[p];
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

// ---- k <= 16, D <= 96: the N x N scalar products as a GEMM on the matrix cores (round 3) ---------------------------------------
// The search is 2 N^2 D flop of  Z Z^T  followed by a top-k per row; knn_kernel above does it with one query per lane and the
// candidates as scalar operands (0.24 of the FP64 peak: a 36-MB scalar stream).  Here, v_mfma_f64_16x16x4_f64 with
//   A = 16 candidates x 4 features,  B = 4 features x 16 queries,  D[candidate kr + 4 q][query li] in lane (li = lane & 15, kr = lane >> 4):
// a lane sees, for ITS query li, the four candidates kr + 4 q of every block of 16 -- so every query is followed by four lanes.
// Both operands come from ONE blocked copy of the unit rows,
//   Zb[block of 16 rows][step of 4 features][lane] = z_{16 block + (lane & 15)}[4 step + (lane >> 4)],
// i.e. the 512 bytes an MFMA wants are contiguous: queries are read once into registers (QB blocks of 16 per wave), candidates
// are streamed by the LDS-DMA engine in stages of 64 rows shared by the four waves (64 QB queries per workgroup), double-
// buffered, one barrier a stage.  Selection: all four lanes of a query keep the SAME sorted list; a value that beats the list's
// k-th entry (a wave-wide ballot, rare after the first few hundred candidates) is handed to the four lanes by a lane shuffle
// and inserted by all of them, block values in index order -- so the threshold is the exact k-th best so far and ties keep the
// smaller index first.
// Measured, 100 000 x 45, k = 10 (kernel time under rocprofv3): scalar-operand kernel 46 ms -> 21.7 ms = 0.56 of the FP64 peak
// counting the 48 padded features (200 000 rows: 74 ms = 0.66; 400 000: 0.70).  Where the rest goes: with the selection off
// 19.4 ms; without barriers, DMA and LDS reads as well 17.5 ms -- chains of 12 dependent MFMAs at three waves per SIMD and
// 6.1 workgroups per CU (seven on some) do not go faster; tools/micro_mfma64_occ.hip has the chip's own limits (one
// accumulator chain per wave: 57 / 68 / 70 TFLOP/s at 1 / 2 / 4 waves per SIMD; alternating chains are SLOWER).
// Versions on the way: a list per lane over its quarter of the candidates, merged at the end (thresholds = the k-th best of
// a quarter let four times as many values into the insertion code), two query blocks per wave with alternating accumulator
// chains (782 workgroups for 256 CUs: four on some, three on most) -- both 27-28 ms.
typedef double zk_v4d __attribute__((ext_vector_type(4)));
#define ZK_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define ZK_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// unit rows in the blocked layout: `steps` feature steps of four (the kernel instance that holds D, zero features beyond it), rows
// padded with zeros to `Np` (a multiple of 64)
__global__ __launch_bounds__(256) void unit_rows_blocked_kernel(const double* __restrict__ X, long long N, int D, long long Np, int steps,
                                                                double* __restrict__ Zb) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= Np) return;
  double* out = Zb + (r >> 4) * steps * 64 + (r & 15);
  double mean = 0.0, inv = 0.0;
  const double* x = X + r * D;
  if (r < N) {
    double s = 0.0;
    for (int f = 0; f < D; ++f) s += x[f];
    mean = s / D;
    double q = 0.0;
    for (int f = 0; f < D; ++f) {
      const double v = x[f] - mean;
      q += v * v;
    }
    inv = q > 0.0 ? 1.0 / sqrt(q) : 0.0;
  }
  for (int f = 0; f < 4 * steps; ++f) out[(f >> 2) * 64 + (f & 3) * 16] = r < N && f < D ? (x[f] - mean) * inv : 0.0;
}

template <int K, int NS, int QB>
__global__ __launch_bounds__(256, NS <= 12 && QB == 1 ? 3 : 2) void knn_mfma_kernel(const double* __restrict__ Zb, long long N, long long n_stages, int k,
                                                          long long* __restrict__ ind, double* __restrict__ dist, int* __restrict__ part_ind,
                                                          double* __restrict__ part_dist) {
  constexpr int steps = NS;  // feature steps of the blocked copy (zero features up to 4 NS)
  extern __shared__ __attribute__((aligned(16))) double lds[];  // two stages of 64 candidates
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kr = lane >> 4;
  constexpr int SB = NS <= 12 ? 4 : 2;  // candidate blocks per stage: 2 x 24 KiB of LDS at most, so two workgroups fit a CU
  const long long n_blocks = n_stages * SB;
  const long long qblock0 = ((long long)blockIdx.x * 4 + wave) * QB;
  const int stage_doubles = SB * steps * 64;
  const int n_int = (int)N;

  double bq[QB][NS];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int s = 0; s < NS; ++s) bq[qb][s] = qblock0 + qb < n_blocks ? Zb[((qblock0 + qb) * steps + s) * 64 + lane] : 0.0;

  double bd[QB][K], thr[QB];
  int bi[QB][K];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    thr[qb] = std::numeric_limits<double>::infinity();
#pragma unroll
    for (int p = 0; p < K; ++p) bd[qb][p] = std::numeric_limits<double>::infinity(), bi[qb][p] = -1;
  }

  auto issue = [&](long long t, int buf) {
    const char* src = (const char*)(Zb + t * stage_doubles) + lane * 16;
    char* dst = (char*)(lds + buf * stage_doubles);
    for (int c = wave; c < SB * steps / 2; c += 4)  // 1-KiB pieces of the stage
      __builtin_amdgcn_global_load_lds(ZK_GLOBAL_PTR(src + c * 1024), ZK_LDS_PTR(dst + c * 1024), 16, 0, 0);
  };

  // blockIdx.y = the part of the candidate stages this workgroup scans (gridDim.y parts: more, smaller workgroups than
  // query tiles alone give -- 100 000 rows are 782 tiles for 512 resident workgroups); the parts' lists are merged afterwards
  const long long t_lo = n_stages * blockIdx.y / gridDim.y, t_hi = n_stages * (blockIdx.y + 1) / gridDim.y;
  if (t_lo < t_hi) issue(t_lo, (int)(t_lo & 1));
  double a[NS];
  for (long long t = t_lo; t < t_hi; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of stage t have landed
    __syncthreads();                                   // ... everybody's have, and nobody reads the other buffer any more
    if (t + 1 < t_hi) issue(t + 1, (int)((t + 1) & 1));
    const double* __restrict__ sb = lds + (t & 1) * stage_doubles + lane;
    const bool tail = (t + 1) * (16 * SB) > N;  // the stage holds padding rows
    // the A operand of step s of block cb + 1 is requested as soon as the MFMAs of step s of block cb have been issued (a
    // lone wave on a SIMD would otherwise sit out an LDS round trip every four MFMAs)
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = sb[s * 64];
#pragma unroll
    for (int cb = 0; cb < SB; ++cb) {
      // one accumulator chain after the other: back-to-back MFMAs into the SAME accumulator run at 74 clocks each with two waves
      // on the SIMD, two alternating chains at 78, four at 87 (tools/micro_mfma64_occ.hip); the last chain re-requests a[s]
      zk_v4d acc[QB];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        acc[qb] = zk_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          acc[qb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], bq[qb][s], acc[qb], 0, 0, 0);
          if (qb == QB - 1 && cb + 1 < SB) {
            a[s] = sb[((cb + 1) * steps + s) * 64];
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // the MFMA ...
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // ... then the request for the next block's operand
          }
        }
      }
      const int cb0 = (int)(t * (16 * SB)) + cb * 16;  // candidate cb0 + g + 4 q is element q of lane group g (= kr of its lanes)
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        double cd[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cd[q] = 1.0 - acc[qb][q];
        if (tail) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (cb0 + kr + 4 * q >= n_int) cd[q] = std::numeric_limits<double>::infinity();
        }
        const double best = __builtin_fmin(__builtin_fmin(cd[0], cd[1]), __builtin_fmin(cd[2], cd[3]));
        if (__ballot(best < thr[qb])) {
          // a value that beats its query's threshold goes to ALL FOUR lanes of the query, in index order (q outer, lane group
          // inner), so the four lists stay equal and strict comparisons keep ties in index order.  Run-time loops: one copy
          // of the insertion chain per query block keeps the loop body inside the instruction cache.
#pragma nounroll
          for (int q = 0; q < 4; ++q) {
            const double cq = q == 0 ? cd[0] : q == 1 ? cd[1] : q == 2 ? cd[2] : cd[3];
            unsigned long long hit = __ballot(cq < thr[qb]);
            while (hit) {
              const int g = __builtin_ctzll(hit) >> 4;  // the first lane group with a hit
              hit &= ~(0xffffull << (16 * g));
              const double nd = __shfl(cq, li + 16 * g, 64);
              double vd = nd;
              int vi = cb0 + g + 4 * q;
#pragma unroll
              for (int p = 0; p < K; ++p) {
                const bool lt = nd < bd[qb][p];  // the new value, not the entry being pushed down (see knn_kernel)
                const double td = bd[qb][p];
                const int ti = bi[qb][p];
                bd[qb][p] = lt ? vd : td;
                bi[qb][p] = lt ? vi : ti;
                vd = lt ? td : vd;
                vi = lt ? ti : vi;
              }
              double w = bd[qb][K - 1];
#pragma unroll
              for (int p = 0; p < K - 1; ++p)
                if (p == k - 1) w = bd[qb][p];
              thr[qb] = w;
            }
          }
        }
      }
    }
  }

  // the four lanes of a query hold the same list; lane group 0 writes it (one part: the result; else this part's list)
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const long long qrow = (qblock0 + qb) * 16 + li;
    if (kr == 0 && qrow < N) {
      if (gridDim.y == 1) {
#pragma unroll
        for (int p = 0; p < K; ++p)
          if (p < k) {
            ind[qrow * k + p] = bi[qb][p];
            dist[qrow * k + p] = bd[qb][p];
          }
      } else {
        const long long o = ((long long)blockIdx.y * N + qrow) * k;
#pragma unroll
        for (int p = 0; p < K; ++p)
          if (p < k) {
            part_ind[o + p] = bi[qb][p];
            part_dist[o + p] = bd[qb][p];
          }
      }
    }
  }
}

// the k best of a query from the sorted lists of its `parts` candidate ranges (ascending ranges: on equal distances the lower
// part, i.e. the smaller index, goes first); one thread per query
__global__ __launch_bounds__(256) void knn_merge_kernel(const int* __restrict__ part_ind, const double* __restrict__ part_dist, long long N, int k,
                                                        int parts, long long* __restrict__ ind, double* __restrict__ dist) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= N) return;
  int head[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) head[c] = 0;
  for (int p = 0; p < k; ++p) {
    double best = std::numeric_limits<double>::infinity();
    int from = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < parts && head[c] < k) {
        const double v = part_dist[((long long)c * N + q) * k + head[c]];
        if (v < best) best = v, from = c;
      }
    int h = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c == from) h = head[c]++;
    ind[q * k + p] = part_ind[((long long)from * N + q) * k + h];
    dist[q * k + p] = best;
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
  void* part_bufs[2] = {nullptr, nullptr};
  auto cleanup = [&]() {
    for (void* p : {(void*)Z, (void*)Zt, (void*)d_dist, (void*)d_P, (void*)d_ind, part_bufs[0], part_bufs[1]})
      if (p) (void)hipFree(p);
  };
  // k <= 16 and D <= 96: the matrix-core kernel on the blocked copy (ZK_KNN_SCALAR=1 keeps the scalar-operand kernel for A/B)
  const char* force_scalar = getenv("ZK_KNN_SCALAR");
  const bool mfma = k <= 16 && D <= 96 && N < (1LL << 31) - 64 && !(force_scalar && force_scalar[0] == '1');
  // feature steps of the blocked copy: the kernel instance that holds D (zero features beyond it)
  const int steps = D <= 8 ? 2 : D <= 16 ? 4 : D <= 32 ? 8 : (D + 15) / 16 * 4;
  const long long Np64 = (N + 63) & ~63LL;
  hipError_t e = hipSuccess;
  if (mfma)
    e = hipMalloc((void**)&Z, (size_t)Np64 * steps * 4 * sizeof(double));
  else {
    e = hipMalloc((void**)&Z, (size_t)N * D * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&Zt, (size_t)Np * D * sizeof(double) + 256);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&d_dist, (size_t)N * k * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&d_ind, (size_t)N * k * sizeof(long long));
  if (e == hipSuccess && P_out) e = hipMalloc((void**)&d_P, (size_t)N * k * sizeof(double));
  if (e != hipSuccess) {
    cleanup();
    return zk_hip_fail(e, "hipMalloc(kNN buffers)");
  }
  if (mfma) {
    hipLaunchKernelGGL(unit_rows_blocked_kernel, dim3((unsigned)((Np64 + 255) / 256)), dim3(256), 0, s, zk_rows_data(m), (long long)N, D, Np64,
                       steps, Z);
    const int sb = steps <= 12 ? 4 : 2;  // candidate blocks per stage (the kernel's SB)
    const size_t stage = (size_t)sb * steps * 64 * sizeof(double);
    const size_t lds = 2 * stage;
    const unsigned tiles1 = (unsigned)((N + 63) / 64);
    const long long n_stages = Np64 / (16 * sb);
    // One query block of 16 per wave, 64 queries per workgroup (QB = 1): 100 000 rows are 1 563 workgroups, 6.1 per CU -- with
    // two blocks per wave (782 workgroups, 3.05 per CU, four on some) the busiest CUs set the time: 27 ms against 22.  Three
    // workgroups fit a CU.  Small matrices are cut further, into parts of the candidate range (each part pays the fill-up of
    // its lists again -- 100 000 x 45 in 1 / 2 / 4 / 8 / 16 parts: 28 / 27 / 29 / 32 / 38 ms -- so only up to one round of
    // resident workgroups)
    const unsigned tiles = tiles1;
    int parts = (int)(3u * 256u / tiles);
    if (const char* pe = getenv("ZK_KNN_PARTS")) parts = atoi(pe);
    if (parts > n_stages / 16) parts = (int)(n_stages / 16);
    parts = parts < 1 ? 1 : parts > 16 ? 16 : parts;
    int* d_pind = nullptr;
    double* d_pdist = nullptr;
    if (parts > 1) {
      e = hipMalloc((void**)&d_pind, (size_t)parts * N * k * sizeof(int));
      if (e == hipSuccess) e = hipMalloc((void**)&d_pdist, (size_t)parts * N * k * sizeof(double));
      if (e != hipSuccess) {
        if (d_pind) (void)hipFree(d_pind);
        cleanup();
        return zk_hip_fail(e, "hipMalloc(kNN part lists)");
      }
    }
    const dim3 grid1(tiles1, parts);
    // (lists of 10 entries for k <= 10, ForceGraph8's default: a shorter insertion chain)
#define ZK_KNN_CASE(NS)                                                                                                                  \
  case NS:                                                                                                                              \
    if (k <= 10)                                                                                                                        \
      hipLaunchKernelGGL((knn_mfma_kernel<10, NS, 1>), grid1, dim3(256), lds, s, Z, (long long)N, n_stages, k, d_ind, d_dist,           \
                         d_pind, d_pdist);                                                                                              \
    else                                                                                                                                \
      hipLaunchKernelGGL((knn_mfma_kernel<16, NS, 1>), grid1, dim3(256), lds, s, Z, (long long)N, n_stages, k, d_ind, d_dist,           \
                         d_pind, d_pdist);                                                                                              \
    break;
    switch (steps) {
      ZK_KNN_CASE(2)
      ZK_KNN_CASE(4)
      ZK_KNN_CASE(8)
      ZK_KNN_CASE(12)
      ZK_KNN_CASE(16)
      ZK_KNN_CASE(20)
      ZK_KNN_CASE(24)
    }
#undef ZK_KNN_CASE
    if (parts > 1) {
      hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, d_pind, d_pdist, (long long)N, k, parts, d_ind, d_dist);
      // (freed after the stream has drained: cleanup() below runs after hipStreamSynchronize)
      part_bufs[0] = d_pind;
      part_bufs[1] = d_pdist;
    }
  } else {
    hipLaunchKernelGGL(unit_rows_kernel, dim3((unsigned)((Np + 255) / 256)), dim3(256), 0, s, zk_rows_data(m), (long long)N, D, Np, Z, Zt);
    const size_t lds = (size_t)64 * D * sizeof(double);
    const unsigned grid = (unsigned)((N + 63) / 64);
    if (k <= 16)
      hipLaunchKernelGGL((knn_kernel<16, 4>), dim3(grid), dim3(256), lds + (size_t)16 * 64 * 16, s, Z, Zt, (long long)N, D, Np, k, d_ind, d_dist);
    else if (k <= 32)
      hipLaunchKernelGGL((knn_kernel<32, 1>), dim3(grid), dim3(64), lds, s, Z, Zt, (long long)N, D, Np, k, d_ind, d_dist);
    else
      hipLaunchKernelGGL((knn_kernel<64, 1>), dim3(grid), dim3(64), lds, s, Z, Zt, (long long)N, D, Np, k, d_ind, d_dist);
  }
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
// (searchsorted side='right').  scikit-learn's k-means++ takes its first seed this way: 30 ms of NumPy on 4 M rows.
// Round 2 replayed the two runs of dependent additions (8 ms); round 3 jumps through them: inside a binade every addition of
// the same c rounds the same way, so the running sum moves by a constant step until its bit length changes -- the sum after
// any number of additions costs a few exact integer steps per binade, and the index is a binary search over that.
namespace {

inline int zk_bitlen(unsigned __int128 v) {
  const uint64_t hi = (uint64_t)(v >> 64), lo = (uint64_t)v;
  return hi ? 128 - __builtin_clzll(hi) : lo ? 64 - __builtin_clzll(lo) : 0;
}

// fl(... fl(fl(0 + c) + c) ... + c), `steps` additions in round-to-nearest-even float64, c > 0 normal.  Integers in units of
// ulp(c): c = mc (53 bits), the running sum S (at most 53 significant bits, below 2^53 * steps).
double zk_repeated_sum(double c, int64_t steps) {
  if (steps <= 0) return 0.0;
  int ec = 0;
  const double fr = std::frexp(c, &ec);                      // c = fr 2^ec, fr in [0.5, 1)
  const unsigned __int128 mc = (unsigned __int128)(uint64_t)std::ldexp(fr, 53);
  const int qc = ec - 53;                                    // c = mc 2^qc
  auto step = [&](unsigned __int128 S) {                     // one rounded addition
    unsigned __int128 T = S + mc;
    const int b = zk_bitlen(T);
    if (b <= 53) return T;
    const int sh = b - 53;
    const unsigned __int128 ulp = (unsigned __int128)1 << sh, r = T & (ulp - 1), half = ulp >> 1;
    T -= r;
    if (r > half || (r == half && ((T >> sh) & 1))) T += ulp;
    return T;
  };
  unsigned __int128 S = 0, inc1 = 0, inc2 = 0;
  int b1 = -1, b2 = -2;                                      // bit lengths of the sum before the last two steps
  int64_t left = steps;
  while (left > 0) {
    const unsigned __int128 N = step(S);
    inc2 = inc1, inc1 = N - S;
    b2 = b1, b1 = zk_bitlen(S);
    S = N;
    --left;
    const int b = zk_bitlen(S);
    // steady: two equal steps inside one binade (a tie case settles after its first step) -- the same step repeats while
    // S + c keeps the bit length, i.e. for every further step whose exact sum stays below 2^max(b, 53)
    if (left > 0 && inc1 == inc2 && b1 == b && b2 == b && inc1 > 0) {
      const unsigned __int128 top = (unsigned __int128)1 << (b < 53 ? 53 : b);
      if (S + mc < top) {
        const unsigned __int128 room = top - 1 - mc - S;
        unsigned __int128 j = room / inc1 + 1;
        if (j > (unsigned __int128)left) j = (unsigned __int128)left;
        S += j * inc1;
        left -= (int64_t)j;
      }
    }
  }
  const int t = zk_bitlen(S) > 53 ? zk_bitlen(S) - 53 : 0;
  return std::ldexp((double)(uint64_t)(S >> t), t + qc);
}

}  // namespace

extern "C" int zk_uniform_choice_index(int64_t n, double u, int64_t* index_out) {
  if (n <= 0 || !index_out) return zk_fail(ZK_E_BADARG, "bad arguments");
  const double c = 1.0 / (double)n;  // ones(n) / ones(n).sum(): n is exact in float64 below 2^53
  const double last = zk_repeated_sum(c, n);
  // the smallest i with cdf_i = fl(sum_{i+1} / last) > u (the sums do not decrease with i); n when there is none
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (zk_repeated_sum(c, mid + 1) / last > u) hi = mid;
    else lo = mid + 1;
  }
  *index_out = lo;
  return 0;
}

// the plain replay (sequential additions), kept for the tests of the jumping version
extern "C" int zk_uniform_choice_index_sequential(int64_t n, double u, int64_t* index_out, double* last_out) {
  if (n <= 0 || !index_out) return zk_fail(ZK_E_BADARG, "bad arguments");
  const double c = 1.0 / (double)n;
  double last = 0.0;
  for (int64_t i = 0; i < n; ++i) last += c;
  double s = 0.0;
  int64_t i = 0;
  for (; i < n; ++i) {
    s += c;
    if (s / last > u) break;
  }
  *index_out = i;
  if (last_out) *last_out = last;
  return 0;
}

// the sum of `steps` sequential additions of c (tests)
extern "C" double zk_repeated_sum_f64(double c, int64_t steps) { return zk_repeated_sum(c, steps); }
