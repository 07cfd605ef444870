// zk_cluster.hip -- clustering consumers of the moment matrix on the device:
//   kmeans_lbs(X, n)   reference clustering/_clustering_functions.py:8-22  (sklearn KMeans(n, random_state).fit(X).labels_)
//   gmm_lbs(X, n)      reference clustering/_clustering_functions.py:25-33 (sklearn GaussianMixture(n, type).fit(X).predict(X))
// The (N, D) float64 matrix stays in HBM (zk_rows); every pass over it is a kernel here, the decisions between passes
// (random draws, centre updates, D x D Cholesky factors, convergence tests) are scikit-learn's own control flow restated
// in the Python wrapper (mtflearn_amd/features/consumers.py).
//
// Common shape of the row kernels: a workgroup is ONE wave that owns tiles of 64 consecutive rows.  A tile is 64*D
// contiguous doubles = 32*D granules of 16 B: the LDS-DMA engine copies it as it lies (global_load_lds_dwordx4, 1 KiB per
// instruction, no VGPRs) into one of the wave's two LDS buffers while the wave computes on the other (tile_pipe).  Then
// lane r walks row r (stride D doubles: conflict-free for odd D, i.e. for n_max 8 / 12 moment matrices) while everything
// that is not a matrix element -- centre coordinates, Cholesky factors, column means -- is wave-uniform and arrives through
// scalar loads (as in the moment kernels).  Column sums per cluster are built the other way round (lane = column, the row's label wave-uniform) with
// LDS floating-point adds into a wave-private table.  Per-workgroup partial results are written out and summed by a second
// kernel in a fixed order: results do not depend on scheduling (bit-identical from run to run on the same device).
#include "zk_internal.h"
#include "zk_fold.h"  // ZK_CONST / zk_const

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

struct zk_rows {
  int device = 0;
  int n_cu = 256;
  int64_t N = 0;
  int D = 0;
  const double* X = nullptr;
  bool own = false;
  hipStream_t stream = nullptr;
  double* d_mean = nullptr;      // [D] the centring shift (column means; zeros until zk_rows_center)
  double* d_xsq = nullptr;       // [N] squared norms of the centred rows
  void* d_tab = nullptr;         // wave-uniform operand tables of the current call
  size_t tab_bytes = 0;
  void* d_part = nullptr;        // per-workgroup partial results
  size_t part_bytes = 0;
  void* d_red = nullptr;         // reduced results
  size_t red_bytes = 0;
  void* d_seed[2] = {nullptr, nullptr};  // k-means++: (t, N) candidate distance rows, ping-pong
  size_t seed_bytes[2] = {0, 0};
  int seed_cur = 0;              // buffer that holds the current closest-distance row
  const double* d_closest = nullptr;
  int seed_t = 0;                // candidates of the last zk_kmeans_seed_step ...
  int seed_last = -1;            // ... and the buffer their distance rows are in
  int32_t* d_labels = nullptr;   // [N]
  void* d_resp = nullptr;        // (k, N) responsibilities
  size_t resp_bytes = 0;
  unsigned long long* d_count = nullptr;  // [4] integer counters
  std::vector<double> h_buf;
  void* h_pin = nullptr;         // page-locked staging of the operand tables (up) and the reduced results (down)
  size_t pin_bytes = 0;
  hipEvent_t ev_up = nullptr;    // the last table upload has left the staging buffer
  void* h_res = nullptr;         // page-locked landing area of the reduced results (+ 8 bytes for a counter)
  size_t res_bytes = 0;
  bool profile = false;          // HIP events around the main kernel of zk_kmeans_step / zk_gmm_estep / zk_gmm_moments
  hipEvent_t ev[2] = {nullptr, nullptr};
  double last_kernel_ms = 0.0;
};

namespace {

constexpr int TILE = 64;

#define ZK_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define ZK_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// The wave's stream of tiles t = blockIdx.x, + gridDim.x, ...: acquire() returns tile t in LDS (row r at [r * D]) once its
// DMA has landed and puts the DMA of the next tile in flight into the other buffer; advance() moves on.  Only tiles that lie
// wholly inside the matrix are moved by DMA (whole 16-B granules); the ragged last tile is copied with ordinary loads, rows
// past the end as zeros.
struct tile_pipe {
  const double* X;
  long long N, t, step;
  int D, lane, cur, nbuf;
  double* buf;
  // nbuf = 2: the next tile's DMA flies under this tile's arithmetic; 1: half the LDS, more waves per CU (kernels whose
  // arithmetic, not the stream, sets the pace)
  __device__ __forceinline__ tile_pipe(const double* X_, long long N_, int D_, double* lds, int lane_, int nbuf_ = 2)
      : X(X_), N(N_), t(blockIdx.x), step(gridDim.x), D(D_), lane(lane_), cur(0), nbuf(nbuf_), buf(lds) {
    if (nbuf == 2 && live()) issue(t, buf);
  }
  __device__ __forceinline__ bool live() const { return t * TILE < N; }
  __device__ __forceinline__ int rows() const { return (int)(N - t * TILE < TILE ? N - t * TILE : TILE); }
  __device__ __forceinline__ void issue(long long tt, double* dst) {
    if ((tt + 1) * TILE > N) return;
    const char* src = (const char*)(X + tt * TILE * D);
    const int n_gran = 32 * D;
    for (int q = 0; q * 64 < n_gran; ++q) {
      const int g = q * 64 + lane;
      if (g < n_gran) __builtin_amdgcn_global_load_lds(ZK_GLOBAL_PTR(src + (long long)g * 16), ZK_LDS_PTR((char*)dst + q * 1024), 16, 0, 2);
    }
  }
  __device__ __forceinline__ const double* acquire() {
    double* now = buf + cur * TILE * D;
    if (nbuf == 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the buffer is no longer read
      issue(t, now);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // tile t has landed; the other buffer is no longer read
    if ((t + 1) * TILE > N) {
      const long long base = t * TILE * D, total = N * D;
#pragma unroll 8
      for (int q = 0; q < D; ++q) {
        const long long e = base + (long long)q * TILE + lane;
        now[q * TILE + lane] = e < total ? X[e] : 0.0;
      }
      __syncthreads();
    }
    const long long nt = t + step;
    if (nbuf == 2 && nt * TILE < N) issue(nt, buf + (cur ^ 1) * TILE * D);
    return now;
  }
  __device__ __forceinline__ void advance() {
    t += step;
    if (nbuf == 2) cur ^= 1;
  }
};

// v of the lane a DPP control names (quad_perm / row mirrors: lanes of the same row of 16), for a double
template <int CTRL>
__device__ __forceinline__ double zk_dpp_f64(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- column statistics: part[block][D] = sum over the block's rows of (x - shift) or (x - shift)^2 ---------------------
__global__ __launch_bounds__(64) void colsum_kernel(const double* __restrict__ X, long long N, int D, int nbuf,
                                                    const double* __restrict__ shift, int square, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  const int j0 = lane, j1 = lane + 64;
  const double m0 = shift && j0 < D ? shift[j0] : 0.0, m1 = shift && j1 < D ? shift[j1] : 0.0;
  double a0 = 0.0, a1 = 0.0;
  for (tile_pipe pipe(X, N, D, lds, lane, nbuf); pipe.live(); pipe.advance()) {
    const double* tile = pipe.acquire();
    const int rows = pipe.rows();
    for (int r = 0; r < rows; ++r) {
      if (j0 < D) {
        const double v = tile[r * D + j0] - m0;
        a0 += square ? v * v : v;
      }
      if (j1 < D) {
        const double v = tile[r * D + j1] - m1;
        a1 += square ? v * v : v;
      }
    }
  }
  if (j0 < D) part[(long long)blockIdx.x * D + j0] = a0;
  if (j1 < D) part[(long long)blockIdx.x * D + j1] = a1;
}

// out[g][i] = sum over the blocks b of group g (b = g * group ... , ascending) of part[b][i]
__global__ __launch_bounds__(256) void reduce_kernel(const double* __restrict__ part, int n_blocks, int n, int group,
                                                     double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int b0 = blockIdx.y * group, b1 = b0 + group < n_blocks ? b0 + group : n_blocks;
  double s = 0.0;
  for (int b = b0; b < b1; ++b) s += part[(long long)b * n + i];
  out[(long long)blockIdx.y * n + i] = s;
}

// xsq[r] = sum_j (x_rj - mean_j)^2 (row_norms of the centred matrix); count[0] += rows whose norm is not finite
__global__ __launch_bounds__(64) void rownorm_kernel(const double* __restrict__ X, long long N, int D, int nbuf,
                                                     const double* __restrict__ mean, double* __restrict__ xsq,
                                                     unsigned long long* __restrict__ count) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  const ZK_CONST double* cm = zk_const(mean);
  unsigned long long bad = 0;
  for (tile_pipe pipe(X, N, D, lds, lane, nbuf); pipe.live(); pipe.advance()) {
    const double* row = pipe.acquire() + lane * D;
    double s = 0.0;
    for (int i = 0; i < D; ++i) {
      const double v = row[i] - cm[i];
      s = __builtin_fma(v, v, s);
    }
    const long long r = pipe.t * TILE + lane;
    if (r < N) {
      xsq[r] = s;
      bad += !(s <= 1.7976931348623157e308);
    }
  }
  const unsigned long long any = __ballot(bad != 0);
  if (any && bad) atomicAdd(count, bad);
}

// dots of the centred row with KC wave-uniform vectors: ct is [D][KP] (vector index fastest)
template <int KC>
__device__ __forceinline__ void dots(const double* __restrict__ row, int D, const ZK_CONST double* cm, const ZK_CONST double* ct,
                                     int KP, double (&d)[KC]) {
#pragma unroll
  for (int c = 0; c < KC; ++c) d[c] = 0.0;
  for (int i = 0; i < D; ++i) {
    const double v = row[i] - cm[i];
    const ZK_CONST double* c = ct + (long long)i * KP;
#pragma unroll
    for (int cc = 0; cc < KC; ++cc) d[cc] = __builtin_fma(v, c[cc], d[cc]);
  }
}

// ---- k-means++ seeding step (sklearn/cluster/_kmeans.py _kmeans_plusplus): squared distances of every row to TP <= 8
// candidate rows, d = max(0, (-2 x.c + |c|^2) + |x|^2) as sklearn's _euclidean_distances builds them, folded with the
// closest distance so far; out[c][r]; part[block][c] = potential of candidate c over the block's rows ----------------
template <int TP>
__global__ __launch_bounds__(64) void seed_kernel(const double* __restrict__ X, long long N, int D, int nbuf,
                                                  const double* __restrict__ mean, const double* __restrict__ Ct,
                                                  const double* __restrict__ cc, int t, const double* __restrict__ xsq,
                                                  const double* __restrict__ closest, double* __restrict__ out,
                                                  double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  const ZK_CONST double* cm = zk_const(mean);
  const ZK_CONST double* ct = zk_const(Ct);
  const ZK_CONST double* ccc = zk_const(cc);
  double pot[TP];
#pragma unroll
  for (int c = 0; c < TP; ++c) pot[c] = 0.0;
  for (tile_pipe pipe(X, N, D, lds, lane, nbuf); pipe.live(); pipe.advance()) {
    const double* row = pipe.acquire() + lane * D;
    const long long r = pipe.t * TILE + lane;
    const bool live = r < N;
    const double xs = live ? xsq[r] : 0.0;
    const double prev = closest && live ? closest[r] : std::numeric_limits<double>::infinity();
    double d[TP];
    dots<TP>(row, D, cm, ct, TP, d);
#pragma unroll
    for (int c = 0; c < TP; ++c) {
      double v = __builtin_fma(-2.0, d[c], ccc[c]) + xs;
      v = v > 0.0 ? v : 0.0;
      v = v < prev ? v : prev;
      if (live && c < t) {
        __builtin_nontemporal_store(v, out + (long long)c * N + r);
        pot[c] += v;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < TP; ++c) {
    const double s = wave_sum(pot[c]);
    if (lane == 0) part[(long long)blockIdx.x * TP + c] = s;
  }
}

// sums of consecutive blocks of 1024 elements (for searchsorted(cumsum(a), v))
__global__ __launch_bounds__(256) void blocksum_kernel(const double* __restrict__ a, long long n, double* __restrict__ bsum) {
  __shared__ double w[4];
  const long long base = (long long)blockIdx.x * 1024;
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const long long i = base + q * 256 + threadIdx.x;
    s += i < n ? a[i] : 0.0;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = (w[0] + w[1]) + (w[2] + w[3]);
}

// ---- one Lloyd iteration (sklearn _k_means_lloyd.pyx lloyd_iter_chunked_dense): label = first argmin_c (|c|^2 - 2 x.c);
// with `update`: part[block][c][0..D-1] = sum of the centred rows of cluster c, [D] = their number -----------------------
template <int KC>
__device__ __forceinline__ void lloyd_chunk(const double* __restrict__ row, int D, const ZK_CONST double* cm, const ZK_CONST double* ct,
                                            const ZK_CONST double* cs, int KP, int c0, double& best, int& bl) {
  double d[KC];
  dots<KC>(row, D, cm, ct + c0, KP, d);
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const double sc = __builtin_fma(-2.0, d[c], cs[c0 + c]);
    if (sc < best) best = sc, bl = c0 + c;
  }
}

// sum over the rows in `mask` (wave-uniform) of column j of the tile, centred by mj; j == D: their number.  Four rows per
// step: four scalar bit extractions and LDS reads, one wait, then the adds in row order.
__device__ __forceinline__ double masked_column_sum(const double* __restrict__ tile, int D, int j, double mj, unsigned long long mask) {
  const double* col = tile + (j < D ? j : D - 1);
  const int n_rows = __builtin_popcountll(mask);
  int n = n_rows;
  double acc = 0.0;
  auto next = [&]() {
    const int r = __builtin_ctzll(mask);
    mask &= mask - 1;
    return col[r * D];
  };
  for (; n >= 4; n -= 4) {
    const double v0 = next(), v1 = next(), v2 = next(), v3 = next();
    acc += v0 - mj;
    acc += v1 - mj;
    acc += v2 - mj;
    acc += v3 - mj;
  }
  for (; n > 0; --n) acc += next() - mj;
  return j == D ? (double)n_rows : acc;
}

// KMAX > 0: k <= KMAX, the per-cluster column sums live in registers (lane = column; the rows of cluster c are the set bits
// of ballot(label == c), walked by a scalar loop); KMAX = 0: any k, LDS floating-point adds into a wave-private table
// (an LDS f64 atomic costs ~280 cycles per wave: 2.4x the whole pass at k = 6, so only where the registers do not reach).
// Tried: two waves per tile, both computing the labels (no exchange) and splitting the clusters of the column sums by
// parity, one tile buffer per pair (twelve waves per CU): 0.48 vs 0.47 ms per call at (4 M x 45, k = 6), 0.75 vs 0.83 at
// (2 M x 91, k = 12) -- the redundant distance pass and the two barriers per tile eat what the occupancy gives.  Not kept.
template <int KMAX>
__global__ __launch_bounds__(64) void lloyd_kernel(const double* __restrict__ X, long long N, int D, int nbuf,
                                                   const double* __restrict__ mean, const double* __restrict__ Ct,
                                                   const double* __restrict__ csq, int KP, int k, int32_t* __restrict__ labels,
                                                   int update, double* __restrict__ part, unsigned long long* __restrict__ changed) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* sums = lds + nbuf * TILE * D;
  const int D1 = D + 1;
  const int lane = threadIdx.x;
  const ZK_CONST double* cm = zk_const(mean);
  const ZK_CONST double* ct = zk_const(Ct);
  const ZK_CONST double* cs = zk_const(csq);
  if (KMAX == 0 && update)
    for (int e = lane; e < k * D1; e += 64) sums[e] = 0.0;
  const int j0 = lane, j1 = lane + 64;
  const double m0 = j0 < D ? mean[j0] : 0.0, m1 = j1 < D ? mean[j1] : 0.0;
  constexpr int KA = KMAX > 0 ? KMAX : 1;
  double acc0[KA], acc1[KA];
#pragma unroll
  for (int c = 0; c < KA; ++c) acc0[c] = acc1[c] = 0.0;
  unsigned long long nchg = 0;
  for (tile_pipe pipe(X, N, D, lds, lane, nbuf); pipe.live(); pipe.advance()) {
    const double* tile = pipe.acquire();
    const double* row = tile + lane * D;
    double best = std::numeric_limits<double>::infinity();
    int bl = 0;
    int c0 = 0;
    for (; c0 + 8 <= KP; c0 += 8) lloyd_chunk<8>(row, D, cm, ct, cs, KP, c0, best, bl);
    if (c0 < KP) lloyd_chunk<4>(row, D, cm, ct, cs, KP, c0, best, bl);
    const long long r = pipe.t * TILE + lane;
    const bool live = r < N;
    if (live) {
      nchg += labels[r] != bl;
      labels[r] = bl;
    }
    if (update) {
      const int rows = pipe.rows();
      if constexpr (KMAX > 0) {
        const unsigned long long valid = rows == TILE ? ~0ull : (1ull << rows) - 1;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) {
          const unsigned long long mask = __ballot(bl == c) & valid;
          if (j0 <= D) acc0[c] += masked_column_sum(tile, D, j0, m0, mask);
          if (D >= 64 && j1 <= D) acc1[c] += masked_column_sum(tile, D, j1, m1, mask);
        }
      } else {
        // lane = column; the label of row rr is lane rr's `bl`.  Eight rows per step: their reads first, then the adds
        for (int r0 = 0; r0 < rows; r0 += 8) {
          double v0[8], v1[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            v0[u] = j0 < D ? tile[(r0 + u) * D + j0] : m0 + 1.0;
            v1[u] = j1 < D ? tile[(r0 + u) * D + j1] : m1 + 1.0;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (r0 + u < rows) {
              const int l = __builtin_amdgcn_readlane(bl, r0 + u);
              double* dst = sums + l * D1;
              if (j0 <= D) (void)__hip_atomic_fetch_add(dst + j0, v0[u] - m0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
              if (j1 <= D) (void)__hip_atomic_fetch_add(dst + j1, v1[u] - m1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  if (update) {
    double* out = part + (long long)blockIdx.x * k * D1;
    if constexpr (KMAX > 0) {
#pragma unroll
      for (int c = 0; c < KMAX; ++c)
        if (c < k) {
          if (j0 <= D) out[c * D1 + j0] = acc0[c];
          if (j1 <= D) out[c * D1 + j1] = acc1[c];
        }
    } else {
      for (int e = lane; e < k * D1; e += 64) out[e] = sums[e];
    }
  }
  if (__ballot(nchg != 0) && nchg) atomicAdd(changed, nchg);
}

// dist[r] = sum_j ((x_rj - mean_j) - centre[label_r][j])^2  (sklearn _relocate_empty_clusters_dense)
__global__ __launch_bounds__(64) void owndist_kernel(const double* __restrict__ X, long long N, int D, int nbuf,
                                                     const double* __restrict__ mean, const double* __restrict__ centers,
                                                     const int32_t* __restrict__ labels, double* __restrict__ dist) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  const ZK_CONST double* cm = zk_const(mean);
  for (tile_pipe pipe(X, N, D, lds, lane, nbuf); pipe.live(); pipe.advance()) {
    const double* row = pipe.acquire() + lane * D;
    const long long r = pipe.t * TILE + lane;
    if (r < N) {
      const double* c = centers + (long long)labels[r] * D;
      double s = 0.0;
      for (int i = 0; i < D; ++i) {
        const double v = (row[i] - cm[i]) - c[i];
        s += v * v;
      }
      dist[r] = s;
    }
  }
}

// ---- Gaussian mixture E step (sklearn/mixture/_gaussian_mixture.py _estimate_log_gaussian_prob + _base.py
// _estimate_log_prob_resp): per component y = x P - mu P with P the upper-triangular Cholesky factor of the precision
// ([k][D][DP] row-major, DP = D rounded up to ZK_EW, zeros below the diagonal and in the padding; b = mu P as [k][DP]),
// lp_c = (-0.5 (D log 2pi + |y|^2) + logdet_c) + logw_c; log-sum-exp over components; resp (k, N) = exp(lp - lse);
// label = first argmax; part[block] = sum of lse over the block's rows -----------------------------------------------------
#ifndef ZK_EW
#define ZK_EW 16  // columns of y per pass of the E step (8: 4.2 ms, 16: 2.9 ms per 4 M x 45 rows at k = 6)
#endif
#define ZK_LGKM_WAIT()                \
  __builtin_amdgcn_s_waitcnt(0xC07F); \
  __builtin_amdgcn_sched_barrier(0)

// (Tried: the product y = x P on the matrix cores -- v_mfma_f64_16x16x4_f64 runs at 77 TFLOP/s from vector registers,
//  tools/micro_mfma64.hip, against the 28-48 of SGPR-fed v_fma_f64 -- with wave = 16-row block, A from the LDS tile, B = the
//  factor from memory: correct on every test, but 3.4 ms against 2.1: every lane fetches its own element of the factor, 512 B
//  per MFMA, 19 TB through L1 / L2 per pass over 4 M rows, where a scalar operand is fetched once per wave and used by 64 lanes x
//  16 FMAs.  Reusing B across row blocks needs a different split of the work; not pursued.
//  Also tried: two rows per lane (128-row tiles) so that every scalar operand feeds two FMAs: 2.03 against 2.08 ms -- the
//  halved occupancy takes back what the halved scalar traffic gives.)
// ZK_EWAVES waves share a tile (every wave holds the same 64 rows, lane = row): the (component, column block) pairs are dealt
// to the waves by cost on the host (`plan`: per wave a count and its (c, j0) pairs), each wave adds the squared norms of its
// blocks into its own [k][64] LDS table, and after a barrier every wave combines the tables in wave order -- redundant but
// cheap arithmetic -- for the log-sum-exp; the stores of the responsibilities are split by component.
#ifndef ZK_EWAVES
#define ZK_EWAVES 4
#endif
__global__ __launch_bounds__(64 * ZK_EWAVES) void estep_kernel(const double* __restrict__ X, long long N, int D, int DP,
                                                               const double* __restrict__ P, const double* __restrict__ B,
                                                               const double* __restrict__ cst /* [k][2]: logdet, logw */,
                                                               const int* __restrict__ plan, int plan_stride, double dlog2pi, int k,
                                                               double* __restrict__ resp, int32_t* __restrict__ labels,
                                                               double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* tile = lds;                  // [64][D]
  double* sqw = lds + TILE * D;        // [ZK_EWAVES][k][64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* mine = sqw + (long long)wave * k * 64;
  const ZK_CONST int* my_plan = (const ZK_CONST int*)plan + wave * plan_stride;
  const int n_mine = my_plan[0];
  const int n_gran = 32 * D;
  double lse_sum = 0.0;
  for (long long t = blockIdx.x; t * TILE < N; t += gridDim.x) {
    __syncthreads();  // the previous tile and tables are no longer read
    if ((t + 1) * TILE <= N) {
      const char* src = (const char*)(X + t * TILE * D);
      for (int q = wave; q * 64 < n_gran; q += ZK_EWAVES) {
        const int g = q * 64 + lane;
        if (g < n_gran) __builtin_amdgcn_global_load_lds(ZK_GLOBAL_PTR(src + (long long)g * 16), ZK_LDS_PTR((char*)tile + q * 1024), 16, 0, 2);
      }
    } else {
      const long long base = t * TILE * D, total = N * D;
      for (int q = wave; q < D; q += ZK_EWAVES) {
        const long long e = base + (long long)q * TILE + lane;
        tile[q * TILE + lane] = e < total ? X[e] : 0.0;
      }
    }
    for (int c = 0; c < k; ++c) mine[c * 64 + lane] = 0.0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const double* row = tile + lane * D;
    for (int p = 0; p < n_mine; ++p) {
      const int c = my_plan[1 + 2 * p], j0 = my_plan[2 + 2 * p];
      const ZK_CONST double* pc = zk_const(P) + (long long)c * D * DP;
      const ZK_CONST double* bc = zk_const(B) + (long long)c * DP;
      // ZK_EW columns of y at a time; row i of the factor (ZK_EW wave-uniform doubles) and x_i are requested one step ahead
      // of their FMAs (the scalar-operand pipelining of zk_sep.h: wait for this step's operands, request the next, compute).
      // The prefetch past the last row reads the next component's first row / the table that follows: in bounds.
      double a[ZK_EW], Pn[ZK_EW];
      const ZK_CONST double* pr = pc + j0;
#pragma unroll
      for (int jj = 0; jj < ZK_EW; ++jj) a[jj] = -bc[j0 + jj], Pn[jj] = pr[jj];
      double xn = row[0];
      const int imax = j0 + ZK_EW < D ? j0 + ZK_EW : D;
#define ZK_ESTEP(NEXT)                                                                    \
  {                                                                                       \
    ZK_LGKM_WAIT();                                                                       \
    const double x = xn;                                                                  \
    double Pc[ZK_EW];                                                                     \
    _Pragma("unroll") for (int jj = 0; jj < ZK_EW; ++jj) Pc[jj] = Pn[jj];                 \
    xn = row[i + (NEXT)];                                                                 \
    _Pragma("unroll") for (int jj = 0; jj < ZK_EW; ++jj) Pn[jj] = pr[(NEXT)*DP + jj];     \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    _Pragma("unroll") for (int jj = 0; jj < ZK_EW; ++jj) a[jj] = __builtin_fma(x, Pc[jj], a[jj]); \
  }
      int i = 0;
      for (; i + 3 < imax; i += 4) {
        ZK_ESTEP(1)
        ZK_ESTEP(2)
        ZK_ESTEP(3)
        ZK_ESTEP(4)
        pr += 4 * DP;
      }
      for (; i < imax; ++i) {
        ZK_ESTEP(1)
        pr += DP;
      }
#undef ZK_ESTEP
      ZK_LGKM_WAIT();
      double sq = 0.0;
#pragma unroll
      for (int jj = 0; jj < ZK_EW; ++jj) sq = __builtin_fma(a[jj], a[jj], sq);
      mine[c * 64 + lane] += sq;
    }
    __syncthreads();
    // weighted log probabilities of the 64 rows (every wave), log-sum-exp, outputs split by component / wave
    double best = -std::numeric_limits<double>::infinity();
    int bl = 0;
    for (int c = 0; c < k; ++c) {
      double sq = 0.0;
#pragma unroll
      for (int w = 0; w < ZK_EWAVES; ++w) sq += sqw[((long long)w * k + c) * 64 + lane];
      const ZK_CONST double* cc = zk_const(cst) + 2 * c;
      const double v = (-0.5 * (dlog2pi + sq) + cc[0]) + cc[1];
      if (v > best) best = v, bl = c;
    }
    double s = 0.0;
    for (int c = 0; c < k; ++c) {
      double sq = 0.0;
#pragma unroll
      for (int w = 0; w < ZK_EWAVES; ++w) sq += sqw[((long long)w * k + c) * 64 + lane];
      const ZK_CONST double* cc = zk_const(cst) + 2 * c;
      s += exp(((-0.5 * (dlog2pi + sq) + cc[0]) + cc[1]) - best);
    }
    const double lse = log(s) + best;
    const long long r = t * TILE + lane;
    if (r < N) {
      if (wave == 0) {
        lse_sum += lse;
        if (labels) labels[r] = bl;
      }
      if (resp)
        for (int c = wave; c < k; c += ZK_EWAVES) {
          double sq = 0.0;
#pragma unroll
          for (int w = 0; w < ZK_EWAVES; ++w) sq += sqw[((long long)w * k + c) * 64 + lane];
          const ZK_CONST double* cc = zk_const(cst) + 2 * c;
          __builtin_nontemporal_store(exp(((-0.5 * (dlog2pi + sq) + cc[0]) + cc[1]) - lse), resp + (long long)c * N + r);
        }
    }
  }
  if (wave == 0) {
    const double tot = wave_sum(lse_sum);
    if (lane == 0) part[blockIdx.x] = tot;
  }
}

// ---- E step on the matrix cores (round 3): y = x P_c as  (16 rows x D) . (D x 16-column blocks of the factor) -------------------
// Round 2's attempt above lost because every lane fetched its own element of the factor from global memory.  Here the factors
// of ALL components sit in LDS for the life of the workgroup, already in the MFMA operand order --
//   tabP[c][bj][s][lane] = P_c[4 s + (lane >> 4)][16 bj + (lane & 15)],  s < 4 (bj + 1)   (upper triangle; zeros elsewhere)
// (24 steps x 512 B = 12 KiB per component at D <= 48) -- so a B operand is one conflict-free ds_read_b64, and the eight waves
// of a workgroup (two per SIMD, one workgroup per CU) stream their OWN 16-row blocks of the matrix: the block is DMA'd into a
// wave-private 16 x D buffer, its A operands A[i][k] = x_{row i}[4 s + k] are pulled into registers once (all components and
// column blocks reuse them), and the DMA of the wave's next block is issued straight away -- no barrier after the table load.
// Per (component, column block): one accumulator chain (initialised with -mu P, so the result is y), then sq += y^2; per
// component a 16-lane row reduction (DPP) and four lanes park the four row sums in a wave-private table; the epilogue runs with
// lane = (row, slot): slot s takes the exponentials of the components c = s (mod 4).
// FULL: every piece of the upper triangle ('full' / 'tied' factors); else only the four diagonal pieces of a column block
// ('diag' / 'spherical' factors: the host looks at the factors it was given).
template <int NB, bool FULL, int NW>
__global__ __launch_bounds__(64 * NW) void estep_mfma_kernel(const double* __restrict__ X, long long N, int D, const double* __restrict__ tabP,
                                                            const double* __restrict__ tabB /* [k][NB][16]: -mu P */,
                                                            const double* __restrict__ cst, double dlog2pi, int k,
                                                            double* __restrict__ resp, int32_t* __restrict__ labels,
                                                            double* __restrict__ part) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  constexpr int NS = 4 * NB;                  // feature steps of a row block
  constexpr int PSTEPS = 2 * NB * (NB + 1);   // operand pieces per component
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int li = lane & 15, kr = lane >> 4;
  double* const ptab = lds;                                   // [k][PSTEPS][64]
  double* const btab = ptab + (long long)k * PSTEPS * 64;     // [k][NB][16]
  double* const mine = btab + k * NB * 16 + wave * (16 * D + 4 + 16 * 8);  // this wave's row block [16][D] (+ pad), then sq [8 slots][16]
  double* const sqt = mine + 16 * D + 4;
  {  // the factor table: one cooperative copy
    const long long n = (long long)k * PSTEPS * 64 + k * NB * 16;
    for (long long e = threadIdx.x; e < n; e += 64 * NW) lds[e] = e < (long long)k * PSTEPS * 64 ? tabP[e] : tabB[e - (long long)k * PSTEPS * 64];
  }
  const long long n_blocks = (N + 15) / 16, stride = (long long)gridDim.x * NW;
  long long b = (long long)blockIdx.x * NW + wave;
  const int n_gran = 8 * D;  // 16-byte granules of a block
  auto issue = [&](long long bb) {
    if ((bb + 1) * 16 > N) return;  // the ragged last block is copied with ordinary loads
    const char* src = (const char*)(X + bb * 16 * D);
    for (int q = 0; q * 64 < n_gran; ++q) {
      const int g = q * 64 + lane;
      if (g < n_gran) __builtin_amdgcn_global_load_lds(ZK_GLOBAL_PTR(src + (long long)g * 16), ZK_LDS_PTR((char*)mine + q * 1024), 16, 0, 2);
    }
  };
  if (b < n_blocks) issue(b);
  __syncthreads();  // the table is in place (the only barrier)
  const ZK_CONST double* cc = zk_const(cst);
  double lse_sum = 0.0;
  for (; b < n_blocks; b += stride) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((b + 1) * 16 > N) {
      const long long base = b * 16 * D, total = N * D;
      for (int e = lane; e < 16 * D; e += 64) mine[e] = base + e < total ? X[base + e] : 0.0;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    double a[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int f = 4 * s + kr;
      a[s] = mine[li * D + (f < D ? f : 0)];
      if (f >= D) a[s] = 0.0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the block is in registers: its buffer can take the next one
    if (b + stride < n_blocks) issue(b + stride);
    for (int c = 0; c < k; ++c) {
      const double* __restrict__ pc = ptab + ((long long)c * PSTEPS) * 64 + lane;
      double sq[4] = {0.0, 0.0, 0.0, 0.0};
      v4d acc[NB];  // the NB chains back to back, their results read afterwards: one wait for the matrix pipe per component
#pragma unroll
      for (int bj = 0; bj < NB; ++bj) {
        const double nb = btab[(c * NB + bj) * 16 + li];
        acc[bj] = v4d{nb, nb, nb, nb};
        const double* __restrict__ pb = pc + (2 * bj * (bj + 1)) * 64;  // pieces of the column blocks before bj: 4 (1 + .. + bj)
#pragma unroll
        for (int s = FULL ? 0 : 4 * bj; s < 4 * (bj + 1); ++s) acc[bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], pb[s * 64], acc[bj], 0, 0, 0);
      }
#pragma unroll
      for (int bj = 0; bj < NB; ++bj)
#pragma unroll
        for (int q = 0; q < 4; ++q) sq[q] = __builtin_fma(acc[bj][q], acc[bj][q], sq[q]);
      // sum over the 16 columns a lane group holds (rows kr + 4 q), halving the number of values a lane carries at each of the
      // first two steps: lane pairs (xor 1) split q {0, 1} | {2, 3}, pairs of pairs (xor 2) split again, then two row
      // rotations (by 4 and 8 lanes) -- 5 additions instead of 16; lane li ends with the whole sum of q = 2 (li & 1) + ((li >> 1) & 1)
      {
        const bool hi1 = li & 1, hi2 = li & 2;
        const double keep0 = hi1 ? sq[2] : sq[0], keep1 = hi1 ? sq[3] : sq[1];      // what this lane goes on with
        const double give0 = hi1 ? sq[0] : sq[2], give1 = hi1 ? sq[1] : sq[3];      // what its partner goes on with
        const double a0 = keep0 + zk_dpp_f64<0xB1>(give0), a1 = keep1 + zk_dpp_f64<0xB1>(give1);
        const double keep = hi2 ? a1 : a0, give = hi2 ? a0 : a1;
        double t = keep + zk_dpp_f64<0x4E>(give);
        t += zk_dpp_f64<0x124>(t);  // row_ror:4 and :8 -- the other three quads' lanes with the same li & 3
        t += zk_dpp_f64<0x128>(t);
        if (li < 4) sqt[c * 16 + kr + 4 * (2 * (li & 1) + (li >> 1))] = t;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- epilogue: lane = (row li, slot kr) ---------------------------------------------------------------------------------
    double best = -std::numeric_limits<double>::infinity();
    int bl = 0;
    for (int c = 0; c < k; ++c) {
      const double v = (-0.5 * (dlog2pi + sqt[c * 16 + li]) + cc[2 * c]) + cc[2 * c + 1];
      if (v > best) best = v, bl = c;
    }
    double ssum = 0.0, e0 = 0.0, e1 = 0.0;  // this slot's (at most two: k <= 8) exponentials
    for (int c = kr, n = 0; c < k; c += 4, ++n) {
      const double e = exp(((-0.5 * (dlog2pi + sqt[c * 16 + li]) + cc[2 * c]) + cc[2 * c + 1]) - best);
      ssum += e;
      if (n == 0) e0 = e;
      else e1 = e;
    }
    ssum += __shfl_xor(ssum, 16, 64);
    ssum += __shfl_xor(ssum, 32, 64);
    const double lse = log(ssum) + best;
    const double inv = 1.0 / ssum;  // resp_c = exp(v_c - lse) = exp(v_c - best) / sum
    const long long r = b * 16 + li;
    if (r < N) {
      if (kr == 0) {
        lse_sum += lse;
        if (labels) labels[r] = bl;
      }
      if (resp) {
        if (kr < k) __builtin_nontemporal_store(e0 * inv, resp + (long long)kr * N + r);
        if (kr + 4 < k) __builtin_nontemporal_store(e1 * inv, resp + (long long)(kr + 4) * N + r);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // sqt is read before the next block's sums overwrite it
  }
  // one partial sum per workgroup (waves in order)
  const double tot = wave_sum(lse_sum);
  __syncthreads();
  if (lane == 0) lds[wave] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < NW; ++w) t += lds[w];
    part[blockIdx.x] = t;
  }
}

// resp[c][r] = (labels[r] == c)
__global__ __launch_bounds__(256) void onehot_kernel(const int32_t* __restrict__ labels, long long N, int k, double* __restrict__ resp) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= N) return;
  const int l = labels[r];
  for (int c = 0; c < k; ++c) resp[(long long)c * N + r] = l == c ? 1.0 : 0.0;
}

// part[block][group][c] = sum over the group's rows of w_cr [x_r - shift | 1]^T [x_r - shift | 1]  (D+1 x D+1; upper triangle
// written, mirrored) for C consecutive planes w_c of the responsibilities (unit weights when w is null).  The (4T x 4T)-padded
// matrix is cut into 4 x 4 register tiles; only the T (T + 1) / 2 tiles of the upper triangle are computed, one per thread, and
// the threads of a workgroup form G groups of that many that share the LDS-staged 64-row tile and take every G-th row of it.
// C components per pass share the staging of the tile and the LDS reads of its rows (the kernel is bound by those, not by the
// FMAs): one component per pass 0.59 ms each on 4 M x 45, three per pass: see profiles/r02_cluster_kernel_stats.csv.
template <int C>
__global__ __launch_bounds__(1024) void wgram_kernel(const double* __restrict__ X, long long N, int D, int T, int G,
                                                     const double* __restrict__ shift, const double* __restrict__ w,
                                                     double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) double tile[];  // [64][4T + 4] then [C][64] weights
  const int P = 4 * T + 4, n_ut = T * (T + 1) / 2;  // + 4: a pitch of 4T doubles puts every other row on the same LDS banks
  double* wt = tile + 64 * P;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int g = tid / n_ut, u = tid - g * n_ut;
  const bool worker = g < G;
  int ti = 0, rem = u;
  while (rem >= T - ti) rem -= T - ti, ++ti;
  const int tj = ti + rem;
  double acc[C][4][4];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[c][a][b] = 0.0;
  // columns D .. P-1 never change: the constant 1, then zero padding
  for (int e = tid; e < 64 * (P - D); e += nthreads) {
    const int r = e / (P - D), c = D + e - r * (P - D);
    tile[r * P + c] = c == D ? 1.0 : 0.0;
  }
  const long long total = N * D;
  const int adv_r = nthreads / D, adv_c = nthreads - adv_r * D;
  for (long long r0 = (long long)blockIdx.x * 64; r0 < N; r0 += (long long)gridDim.x * 64) {
    __syncthreads();
    {  // the tile's 64 * D contiguous doubles, eight coalesced loads in flight per thread
      int r = tid / D, c = tid - r * D;
      const long long base = r0 * D;
      for (int e0 = tid; e0 < 64 * D; e0 += 8 * nthreads) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int e = e0 + q * nthreads;
          v[q] = e < 64 * D && base + e < total ? __builtin_nontemporal_load(X + base + e) : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (e0 + q * nthreads < 64 * D) tile[r * P + c] = v[q] - shift[c];
          r += adv_r, c += adv_c;
          if (c >= D) c -= D, ++r;
        }
      }
    }
    for (int e = tid; e < 64 * C; e += nthreads) {
      const int c = e >> 6, r = e & 63;
      wt[e] = r0 + r < N ? (w ? w[(long long)c * N + r0 + r] : 1.0) : 0.0;
    }
    __syncthreads();
    if (worker) {
#pragma unroll 2
      for (int r = g; r < 64; r += G) {
        const double* row = tile + r * P;
        double xi[4], xj[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) xi[a] = row[4 * ti + a], xj[a] = row[4 * tj + a];
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const double wr = wt[c * 64 + r];
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const double xw = xi[a] * wr;
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[c][a][b] = __builtin_fma(xw, xj[b], acc[c][a][b]);
          }
        }
      }
    }
  }
  if (worker) {
    const int D1 = D + 1;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      double* out = part + (((long long)blockIdx.x * G + g) * C + c) * D1 * D1;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int i = 4 * ti + a, j = 4 * tj + b;
          if (i < D1 && j < D1 && i <= j) {
            out[i * D1 + j] = acc[c][a][b];
            if (i != j) out[j * D1 + i] = acc[c][a][b];
          }
        }
    }
  }
}

// The same sums on the matrix cores (D + 1 <= 48, i.e. the 45 moments of n_max 8 and below): S_c = sum_r w_cr z_r z_r^T is a
// symmetric rank-N update -- GEMM-shaped work -- and v_mfma_f64_16x16x4_f64 takes both of its operands straight from LDS: one
// double per lane per operand for 2 048 flops, where the register-tiled kernel above reads eight doubles per 32 flops and is
// bound by those reads.  All components of a call (KC <= 8) are done in ONE pass over the matrix.
//   A wave owns ONE of the six 16 x 16 blocks (bi <= bj) of the 48 x 48-padded upper triangle for the whole kernel, with KC
//   accumulators (4 doubles per lane each), and walks every n-th 64-row tile.  It fetches what IT needs of a tile itself -- the
//   16 columns of block column bi (and of bj) as two 8-KiB planes [64 rows][16 doubles], and the KC x 64 weights -- with the
//   LDS-DMA engine into a slab of its own: no workgroup barrier exists (a wave only reads LDS bytes it DMA'd itself), every wave
//   does the same work, and while one wave of a SIMD waits for its tile the other one's MFMAs run.  (A column block is fetched
//   by three or four waves: L2 traffic, HBM still sees the matrix once.)
//   k dimension = rows: step s uses rows 4s .. 4s + 3: A[i][k] = z_{4s+k}[16 bi + i], B[k][j] = w_c[4s+k] z_{4s+k}[16 bj + j]
//   (lane l: i = j = l & 15, k = l >> 4; tools/micro_mfma64.hip verified the layout).  A plane's row pitch of 128 B puts the
//   four rows of a step on disjoint halves of the LDS banks.
//   Measured, six components of 4 M x 45: register-tiled kernel 2.84 ms (two passes of three); this kernel 1.38 ms
//   (53 TFLOP/s of MFMA work).
//   What bounds it: SQ_VALU_MFMA_BUSY_CYCLES = 0.70 of the kernel's SIMD cycles at a measured 2.35 GHz -- and the same 0.70 in
//   three differently organised versions (barrier per tile pair; this one; this one with 32-row stages double-buffered per
//   wave: 1.43 / 1.38 / 1.48 ms for six components).  On this chip the float64 MFMA and the float64 vector pipe have the
//   same peak (78.6 TFLOP/s) and evidently do not overlap: the ~14 vector instructions a step needs beside its KC MFMAs (the
//   weight products, the shift, address updates) take their share of the same cycles.
template <int KC>
__global__ __launch_bounds__(256, 2) void wgram_mfma_kernel(const double* __restrict__ X, long long N, int D, int sets,
                                                            const double* __restrict__ shift, const double* __restrict__ w,
                                                            double* __restrict__ part) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  constexpr int SLAB = 2 * TILE * 16 + KC * TILE;  // doubles per wave: plane A, plane Z, weights
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long gw = (long long)blockIdx.x * 4 + wv;  // global wave index: set = gw / 6 walks tiles set, set + sets, ...
  const int blk = (int)(gw % 6);
  const long long set = gw / 6;
  if (set >= sets) return;  // (no barrier in this kernel)
  const int bi = blk < 3 ? 0 : blk < 5 ? 1 : 2, bj = blk < 3 ? blk : blk < 5 ? blk - 2 : 2;
  double* const pa = lds + wv * SLAB;
  double* const pz = bi == bj ? pa : pa + TILE * 16;
  double* const pw = pa + 2 * TILE * 16;
  const int ci = 16 * bi + (lane & 15), cj = 16 * bj + (lane & 15), kr = lane >> 4;
  const bool in_i = ci < D, in_j = cj < D;
  const double one_i = ci == D ? 1.0 : 0.0, one_j = cj == D ? 1.0 : 0.0;  // the constant column of [x | 1], then zero padding
  const double si = in_i ? shift[ci] : 0.0, sj = in_j ? shift[cj] : 0.0;
  v4d acc[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) acc[c] = v4d{0.0, 0.0, 0.0, 0.0};

  const long long n_tiles = (N + TILE - 1) / TILE;
  // DMA addressing of a plane: position p = q * 64 + lane (q = 0 .. 7) is granule g = p & 7 (two columns) of row p >> 3; a
  // granule is fetched when its first column exists (its second one may be the next row's first element: never used).  The
  // last tile goes through ordinary loads (rows past the end as zeros; and a DMA of its last row could read past the matrix).
  const int g = lane & 7, rq = lane >> 3;
  const bool ga = 16 * bi + 2 * g < D, gz = 16 * bj + 2 * g < D;
  for (long long t = set; t < n_tiles; t += sets) {
    const long long r0 = t * TILE;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slab is no longer read
    if (t + 1 < n_tiles) {
      const char* rowp = (const char*)(X + (r0 + rq) * D);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const char* src = rowp + (long long)q * 8 * D * 8;
        if (ga) __builtin_amdgcn_global_load_lds(ZK_GLOBAL_PTR(src + (16 * bi + 2 * g) * 8), ZK_LDS_PTR((char*)pa + q * 1024), 16, 0, 0);
        if (bi != bj && gz)
          __builtin_amdgcn_global_load_lds(ZK_GLOBAL_PTR(src + (16 * bj + 2 * g) * 8), ZK_LDS_PTR((char*)pz + q * 1024), 16, 0, 0);
      }
      if (w) {
#pragma unroll
        for (int q = 0; q * 64 < KC * 32; ++q) {
          const int p = q * 64 + lane;  // granule p of the [KC][64] weights: component p / 32, rows 2 (p % 32)
          if (p < KC * 32)
            __builtin_amdgcn_global_load_lds(ZK_GLOBAL_PTR((const char*)(w + (long long)(p >> 5) * N + r0) + (p & 31) * 16),
                                             ZK_LDS_PTR((char*)pw + q * 1024), 16, 0, 0);
        }
      } else {
#pragma unroll
        for (int c = 0; c < KC; ++c) pw[c * TILE + lane] = 1.0;
      }
    } else {
      for (int e = lane; e < TILE * 16; e += 64) {
        const long long r = r0 + (e >> 4);
        const int c1 = 16 * bi + (e & 15), c2 = 16 * bj + (e & 15);
        pa[e] = r < N && c1 < D ? X[r * D + c1] : 0.0;
        if (bi != bj) pz[e] = r < N && c2 < D ? X[r * D + c2] : 0.0;
      }
#pragma unroll
      for (int c = 0; c < KC; ++c) pw[c * TILE + lane] = r0 + lane < N ? (w ? w[(long long)c * N + r0 + lane] : 1.0) : 0.0;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // operands of step s + 1 are read while the MFMAs of step s run (branch-free: the constant columns are selected afterwards).
    // (Round 3 also tried one accumulator chain per component -- the fast form of tools/micro_mfma64_occ.hip -- with the tile's 16
    //  operand pairs kept in registers: 1.41 ms against 1.39 for six components; the weight products, not the chain order, are
    //  what the MFMAs wait for here.)
    const double* ta = pa + kr * 16 + (lane & 15);
    const double* tz = pz + kr * 16 + (lane & 15);
    const double* wt = pw + kr;
    double a_n = ta[0], z_n = tz[0], w_n[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) w_n[c] = wt[c * TILE];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const double a = in_i ? a_n - si : one_i;
      const double z = bi == bj ? a : (in_j ? z_n - sj : one_j);
      double wz[KC];
#pragma unroll
      for (int c = 0; c < KC; ++c) wz[c] = w_n[c] * z;
      if (s + 1 < 16) {
        a_n = ta[(s + 1) * 64];
        z_n = tz[(s + 1) * 64];
#pragma unroll
        for (int c = 0; c < KC; ++c) w_n[c] = wt[c * TILE + (s + 1) * 4];
      }
#pragma unroll
      for (int c = 0; c < KC; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wz[c], acc[c], 0, 0, 0);
    }
  }
  // this wave's partial block: part[set][blk][c][16 x 16] (row (lane >> 4) + 4 q, column lane & 15)
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    double* out = part + (((long long)set * 6 + blk) * KC + c) * 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) out[(kr + 4 * q) * 16 + (lane & 15)] = acc[c][q];
  }
}

// ---- host helpers ------------------------------------------------------------------------------------------------
// Waves per CU matter more than overlap inside a wave (4 M x 45, two buffers / 3 waves -> one buffer / 6 waves per CU: Lloyd
// pass with sums 0.67 -> 0.46 ms, labels only 0.45 -> 0.31, seeding 0.41 -> 0.36): two tile buffers (the next tile's DMA under
// this tile's arithmetic) only while eight waves still fit a CU's 160 KiB, else one
int tile_bufs(const zk_rows* m, size_t extra = 0) {
  static const int forced = getenv("ZK_TILE_NBUF") ? atoi(getenv("ZK_TILE_NBUF")) : 0;  // experiments
  if (forced == 1 || forced == 2) return forced;
  return 8 * ((size_t)2 * TILE * m->D * sizeof(double) + extra + 512) <= 160 * 1024 ? 2 : 1;
}
size_t tile_lds(const zk_rows* m, int nbuf) { return (size_t)nbuf * TILE * m->D * sizeof(double); }

// persistent single-wave workgroups: as many per CU as the LDS tile allows (at most 8), never more than tiles
int row_grid(const zk_rows* m, size_t lds_bytes) {
  // (LDS is handed out in 512-byte granules: a 64 x 45 tile of doubles is exactly 45 of them, and SEVEN fit a CU's 160 KiB)
  static const int margin = getenv("ZK_ROW_GRID_MARGIN") ? atoi(getenv("ZK_ROW_GRID_MARGIN")) : 0;
  int per_cu = (int)((160 * 1024) / (((lds_bytes + 511) & ~(size_t)511) + margin));
  per_cu = std::max(1, std::min(per_cu, 8));
  const long long tiles = (m->N + TILE - 1) / TILE;
  return (int)std::min<long long>(tiles, (long long)per_cu * m->n_cu);
}

int ensure(void** buf, size_t* have, size_t need) { return zk_ensure(buf, have, need); }

// page-locked staging area of at least `need` bytes (tables are a few KiB to ~100 KiB)
int ensure_pinned(zk_rows* m, size_t need) {
  if (m->pin_bytes >= need) return 0;
  if (m->ev_up) ZK_HIP(hipEventSynchronize(m->ev_up));
  if (m->h_pin) (void)hipHostFree(m->h_pin);
  m->h_pin = nullptr;
  m->pin_bytes = 0;
  const size_t want = need < 65536 ? 65536 : need;
  ZK_HIP(hipHostMalloc(&m->h_pin, want, hipHostMallocDefault));
  m->pin_bytes = want;
  if (!m->ev_up) ZK_HIP(hipEventCreateWithFlags(&m->ev_up, hipEventDisableTiming));
  return 0;
}

// the operand tables of a pass: staged in page-locked memory, so the copy is asynchronous and ordered before the pass's kernels on
// the stream -- no synchronisation here (round 3; a pageable source plus hipStreamSynchronize cost every pass ~25 us of its
// ~100 us of host overhead).  The staging buffer is reused once the previous upload has left it (ev_up).
int upload_tab(zk_rows* m, const std::vector<double>& h) {
  const size_t bytes = h.size() * sizeof(double);
  int rc = ensure(&m->d_tab, &m->tab_bytes, bytes);
  if (rc) return rc;
  if ((rc = ensure_pinned(m, bytes))) return rc;
  ZK_HIP(hipEventSynchronize(m->ev_up));  // (complete long ago: every pass ends with a synchronised read-back)
  memcpy(m->h_pin, h.data(), bytes);
  ZK_HIP(hipMemcpyAsync(m->d_tab, m->h_pin, bytes, hipMemcpyHostToDevice, m->stream));
  ZK_HIP(hipEventRecord(m->ev_up, m->stream));
  return 0;
}

// fixed-order sum of the per-workgroup partial results ([n_blocks][n] in d_part) -> host; two levels (groups of 32 blocks,
// then the groups) so that thousands of partials do not become one thread's serial loop
int reduce_to_host(zk_rows* m, int n_blocks, int n, double* host_out, unsigned long long* counter_out = nullptr) {
  const int group = 32, n_groups = (n_blocks + group - 1) / group;
  {  // results land in page-locked memory: the copies are asynchronous and ONE synchronisation ends the pass
    const size_t need = (size_t)n * sizeof(double) + sizeof(unsigned long long);
    if (m->res_bytes < need) {
      if (m->h_res) (void)hipHostFree(m->h_res);
      m->h_res = nullptr;
      m->res_bytes = 0;
      const size_t want = need < 16384 ? 16384 : need;
      ZK_HIP(hipHostMalloc(&m->h_res, want, hipHostMallocDefault));
      m->res_bytes = want;
    }
  }
  int rc = ensure(&m->d_red, &m->red_bytes, (size_t)(n_groups + 1) * n * sizeof(double));
  if (rc) return rc;
  double* fin = (double*)m->d_red;
  double* mid = fin + n;
  if (n_groups > 1) {
    hipLaunchKernelGGL(reduce_kernel, dim3((n + 255) / 256, n_groups), dim3(256), 0, m->stream, (const double*)m->d_part, n_blocks, n,
                       group, mid);
    hipLaunchKernelGGL(reduce_kernel, dim3((n + 255) / 256, 1), dim3(256), 0, m->stream, (const double*)mid, n_groups, n, n_groups, fin);
  } else {
    hipLaunchKernelGGL(reduce_kernel, dim3((n + 255) / 256, 1), dim3(256), 0, m->stream, (const double*)m->d_part, n_blocks, n, n_blocks,
                       fin);
  }
  ZK_HIP(hipGetLastError());
  ZK_HIP(hipMemcpyAsync(m->h_res, fin, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, m->stream));
  if (counter_out)
    ZK_HIP(hipMemcpyAsync((char*)m->h_res + (size_t)n * sizeof(double), m->d_count, sizeof(unsigned long long), hipMemcpyDeviceToHost,
                          m->stream));
  ZK_HIP(hipStreamSynchronize(m->stream));
  memcpy(host_out, m->h_res, (size_t)n * sizeof(double));
  if (counter_out) memcpy(counter_out, (char*)m->h_res + (size_t)n * sizeof(double), sizeof(unsigned long long));
  return 0;
}

int prof_begin(zk_rows* m) {
  if (m->profile) ZK_HIP(hipEventRecord(m->ev[0], m->stream));
  return 0;
}
int prof_end(zk_rows* m) {
  if (m->profile) ZK_HIP(hipEventRecord(m->ev[1], m->stream));
  return 0;
}
int prof_read(zk_rows* m) {  // after the stream has been synchronised
  if (m->profile) {
    float ms = 0.f;
    ZK_HIP(hipEventElapsedTime(&ms, m->ev[0], m->ev[1]));
    m->last_kernel_ms = ms;
  }
  return 0;
}

int check_lds(size_t bytes) {
  if (bytes > 160 * 1024) return zk_fail(ZK_E_BADARG, "too many features / clusters for one 160-KiB LDS tile");
  return 0;
}

template <typename K>
int allow_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) ZK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// the resident matrix
// ------------------------------------------------------------------------------------------------------------------
static int rows_init(zk_rows* m) {
  ZK_HIP(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  (void)hipDeviceGetAttribute(&m->n_cu, hipDeviceAttributeMultiprocessorCount, m->device);
  ZK_HIP(hipMalloc((void**)&m->d_mean, (size_t)m->D * sizeof(double)));
  ZK_HIP(hipMemsetAsync(m->d_mean, 0, (size_t)m->D * sizeof(double), m->stream));
  ZK_HIP(hipMalloc((void**)&m->d_count, 4 * sizeof(unsigned long long)));
  ZK_HIP(hipMemsetAsync(m->d_count, 0, 4 * sizeof(unsigned long long), m->stream));
  return 0;
}

extern "C" int zk_rows_destroy(zk_rows* m) {
  if (!m) return 0;
  ZK_ON_DEVICE(m->device);
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  if (m->own && m->X) (void)hipFree((void*)m->X);
  void* bufs[] = {m->d_mean, m->d_xsq, m->d_tab, m->d_part, m->d_red, m->d_seed[0], m->d_seed[1], m->d_labels, m->d_resp, m->d_count};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  for (hipEvent_t e : m->ev)
    if (e) (void)hipEventDestroy(e);
  if (m->ev_up) (void)hipEventDestroy(m->ev_up);
  if (m->h_pin) (void)hipHostFree(m->h_pin);
  if (m->h_res) (void)hipHostFree(m->h_res);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
  return 0;
}

static int rows_new(int device, const double* X_dev, int64_t N, int D, bool own, zk_rows** out) {
  zk_rows* m = new zk_rows;
  m->device = device;
  m->N = N;
  m->D = D;
  m->X = X_dev;
  m->own = own;
  const int rc = rows_init(m);
  if (rc) {
    m->own = false;
    zk_rows_destroy(m);
    return rc;
  }
  *out = m;
  return 0;
}

static int rows_args(const void* X, int64_t N, int D, zk_rows** out) {
  if (!X || !out) return zk_fail(ZK_E_BADARG, "null pointer");
  if (N <= 0 || D <= 0 || D > 127) return zk_fail(ZK_E_BADARG, "need 1 <= D <= 127 features and N > 0 rows");
  *out = nullptr;
  return 0;
}

extern "C" int zk_rows_adopt(int device, const double* X_dev, int64_t N, int D, zk_rows** out) {
  int rc = rows_args(X_dev, N, D, out);
  if (rc) return rc;
  ZK_ON_DEVICE(device);
  // every pass runs on the object's own non-blocking stream, which is not ordered against whatever stream is still
  // writing the matrix (zk_transform_patches_dev is asynchronous): wait for the producer here, once
  ZK_HIP(hipDeviceSynchronize());
  return rows_new(device, X_dev, N, D, false, out);
}

extern "C" int zk_rows_create(int device, const double* X_host, int64_t N, int D, zk_rows** out) {
  int rc = rows_args(X_host, N, D, out);
  if (rc) return rc;
  ZK_ON_DEVICE(device);
  double* d = nullptr;
  const size_t bytes = (size_t)N * D * sizeof(double);
  ZK_HIP(hipMalloc((void**)&d, bytes));
  // in pieces of 256 MiB on a stream of its own, as the host pipeline of zk_host.hip does: 1.46 GB from a pageable NumPy array
  // in 26 ms (56 GB/s, the PCIe link rate; the first call of a process adds the HIP runtime's start-up, ~0.15 s)
  hipStream_t up = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
  const size_t piece = (size_t)256 << 20;
  for (size_t off = 0; off < bytes && e == hipSuccess; off += piece)
    e = hipMemcpyAsync((char*)d + off, (const char*)X_host + off, std::min(piece, bytes - off), hipMemcpyHostToDevice, up);
  if (e == hipSuccess) e = hipStreamSynchronize(up);
  if (up) (void)hipStreamDestroy(up);
  if (e != hipSuccess) {
    (void)hipFree(d);
    return zk_hip_fail(e, "hipMemcpy(X)");
  }
  rc = rows_new(device, d, N, D, true, out);
  if (rc) (void)hipFree(d);
  return rc;
}

extern "C" const double* zk_rows_data(const zk_rows* m) { return m ? m->X : nullptr; }

// for the other consumers of a resident matrix (zk_graph.hip)
int zk_rows_shape(const zk_rows* m, int* device, int64_t* n_rows, int* n_features, void** stream) {
  if (!m) return zk_fail(ZK_E_BADARG, "null matrix");
  *device = m->device;
  *n_rows = m->N;
  *n_features = m->D;
  *stream = (void*)m->stream;
  return 0;
}

// Column sums of the matrix (first pass of numpy.mean).
extern "C" int zk_rows_colsum(zk_rows* m, double* sums_out) {
  if (!m || !sums_out) return zk_fail(ZK_E_BADARG, "null pointer");
  ZK_ON_DEVICE(m->device);
  const int nbuf = tile_bufs(m);
  const size_t lds = tile_lds(m, nbuf);
  int rc = check_lds(lds);
  if (rc || (rc = allow_lds(colsum_kernel, lds))) return rc;
  const int grid = row_grid(m, lds);
  if ((rc = ensure(&m->d_part, &m->part_bytes, (size_t)grid * m->D * sizeof(double)))) return rc;
  hipLaunchKernelGGL(colsum_kernel, dim3(grid), dim3(64), lds, m->stream, m->X, (long long)m->N, m->D, nbuf, (const double*)nullptr, 0,
                     (double*)m->d_part);
  ZK_HIP(hipGetLastError());
  return reduce_to_host(m, grid, m->D, sums_out);
}

// `mean` (D) becomes the centring shift of every later k-means call (scikit-learn subtracts the column means before
// clustering; with several ranks it is the mean over all of them); sqsum_out[j] = sum_r (x_rj - mean_j)^2 (second pass of
// numpy.var), the squared norms of the centred rows stay on the device, n_bad_out = rows with a non-finite element.
extern "C" int zk_rows_center_at(zk_rows* m, const double* mean, double* sqsum_out, int64_t* n_bad_out) {
  if (!m || !mean || !sqsum_out || !n_bad_out) return zk_fail(ZK_E_BADARG, "null pointer");
  ZK_ON_DEVICE(m->device);
  const int nbuf = tile_bufs(m);
  const size_t lds = tile_lds(m, nbuf);
  int rc = check_lds(lds);
  if (rc || (rc = allow_lds(colsum_kernel, lds)) || (rc = allow_lds(rownorm_kernel, lds))) return rc;
  const int grid = row_grid(m, lds);
  if ((rc = ensure(&m->d_part, &m->part_bytes, (size_t)grid * m->D * sizeof(double)))) return rc;
  ZK_HIP(hipMemcpyAsync(m->d_mean, mean, (size_t)m->D * sizeof(double), hipMemcpyHostToDevice, m->stream));
  ZK_HIP(hipStreamSynchronize(m->stream));
  hipLaunchKernelGGL(colsum_kernel, dim3(grid), dim3(64), lds, m->stream, m->X, (long long)m->N, m->D, nbuf, (const double*)m->d_mean, 1,
                     (double*)m->d_part);
  ZK_HIP(hipGetLastError());
  if ((rc = reduce_to_host(m, grid, m->D, sqsum_out))) return rc;
  if (!m->d_xsq) ZK_HIP(hipMalloc((void**)&m->d_xsq, (size_t)m->N * sizeof(double)));
  ZK_HIP(hipMemsetAsync(m->d_count, 0, sizeof(unsigned long long), m->stream));
  hipLaunchKernelGGL(rownorm_kernel, dim3(grid), dim3(64), lds, m->stream, m->X, (long long)m->N, m->D, nbuf, (const double*)m->d_mean,
                     m->d_xsq, m->d_count);
  ZK_HIP(hipGetLastError());
  unsigned long long bad = 0;
  ZK_HIP(hipMemcpyAsync(&bad, m->d_count, sizeof(bad), hipMemcpyDeviceToHost, m->stream));
  ZK_HIP(hipStreamSynchronize(m->stream));
  *n_bad_out = (int64_t)bad;
  return 0;
}

// Column means (mean_out) and population variances about them (var_out), two passes as numpy.mean / numpy.var, of THIS
// matrix alone (= zk_rows_colsum, division, zk_rows_center_at, division).
extern "C" int zk_rows_center(zk_rows* m, double* mean_out, double* var_out, int64_t* n_bad_out) {
  if (!m || !mean_out || !var_out || !n_bad_out) return zk_fail(ZK_E_BADARG, "null pointer");
  int rc = zk_rows_colsum(m, mean_out);
  if (rc) return rc;
  for (int j = 0; j < m->D; ++j) mean_out[j] /= (double)m->N;
  if ((rc = zk_rows_center_at(m, mean_out, var_out, n_bad_out))) return rc;
  for (int j = 0; j < m->D; ++j) var_out[j] /= (double)m->N;
  return 0;
}

// rows idx[0..n) of the matrix, minus the centring shift when `centred`, to the host
extern "C" int zk_rows_fetch(zk_rows* m, const int64_t* idx, int n, int centred, double* rows_out) {
  if (!m || !idx || !rows_out || n < 0) return zk_fail(ZK_E_BADARG, "bad arguments");
  ZK_ON_DEVICE(m->device);
  for (int i = 0; i < n; ++i) {
    if (idx[i] < 0 || idx[i] >= m->N) return zk_fail(ZK_E_BADARG, "row index out of range");
    ZK_HIP(hipMemcpyAsync(rows_out + (size_t)i * m->D, m->X + idx[i] * m->D, (size_t)m->D * sizeof(double), hipMemcpyDeviceToHost,
                          m->stream));
  }
  ZK_HIP(hipStreamSynchronize(m->stream));
  if (centred) {
    std::vector<double> mean(m->D);
    ZK_HIP(hipMemcpy(mean.data(), m->d_mean, (size_t)m->D * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < m->D; ++j) rows_out[(size_t)i * m->D + j] -= mean[j];
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// k-means
// ------------------------------------------------------------------------------------------------------------------
// operand table of wave-uniform vectors: [D][KP] (vector index fastest), then their KP squared norms (+inf beyond n)
static void pack_vectors(const double* v, int n, int D, int KP, bool pad_inf, std::vector<double>& h) {
  h.assign((size_t)D * KP + KP, 0.0);
  for (int c = 0; c < n; ++c) {
    double s = 0.0;
    for (int i = 0; i < D; ++i) {
      const double x = v[(size_t)c * D + i];
      h[(size_t)i * KP + c] = x;
      s += x * x;
    }
    h[(size_t)D * KP + c] = s;
  }
  if (pad_inf)
    for (int c = n; c < KP; ++c) h[(size_t)D * KP + c] = std::numeric_limits<double>::infinity();
}

// One seeding step: the centred candidate rows cand (t, D) with their squared norms cand_sq (t) -- the caller fetched them
// with zk_rows_fetch and took the norms the way scikit-learn does -- against every row; with use_closest the distances are
// folded with the current closest-distance row.  pot_out[c] = sum_r dist[c][r].
extern "C" int zk_kmeans_seed_step(zk_rows* m, const double* cand, const double* cand_sq, int t, int use_closest, double* pot_out) {
  if (!m || !cand || !cand_sq || !pot_out) return zk_fail(ZK_E_BADARG, "null pointer");
  if (t < 1 || t > 8) return zk_fail(ZK_E_BADARG, "1 to 8 candidates per seeding step");
  if (!m->d_xsq) return zk_fail(ZK_E_BADARG, "zk_rows_center first");
  if (use_closest && !m->d_closest) return zk_fail(ZK_E_BADARG, "no closest-distance row yet");
  ZK_ON_DEVICE(m->device);
  const int TP = t <= 4 ? 4 : 8;
  pack_vectors(cand, t, m->D, TP, false, m->h_buf);
  for (int c = 0; c < t; ++c) m->h_buf[(size_t)m->D * TP + c] = cand_sq[c];
  int rc = upload_tab(m, m->h_buf);
  if (rc) return rc;
  const int nbuf = tile_bufs(m);
  const size_t lds = tile_lds(m, nbuf);
  if ((rc = check_lds(lds))) return rc;
  const int grid = row_grid(m, lds);
  const int dst = use_closest ? 1 - m->seed_cur : m->seed_cur;
  if ((rc = ensure(&m->d_seed[dst], &m->seed_bytes[dst], (size_t)8 * m->N * sizeof(double)))) return rc;
  if ((rc = ensure(&m->d_part, &m->part_bytes, (size_t)grid * TP * sizeof(double)))) return rc;
  const double* tab = (const double*)m->d_tab;
  const double* closest = use_closest ? m->d_closest : nullptr;
  if (TP == 4) {
    if ((rc = allow_lds(seed_kernel<4>, lds))) return rc;
    hipLaunchKernelGGL(seed_kernel<4>, dim3(grid), dim3(64), lds, m->stream, m->X, (long long)m->N, m->D, nbuf, (const double*)m->d_mean,
                       tab, tab + (size_t)m->D * TP, t, (const double*)m->d_xsq, closest, (double*)m->d_seed[dst], (double*)m->d_part);
  } else {
    if ((rc = allow_lds(seed_kernel<8>, lds))) return rc;
    hipLaunchKernelGGL(seed_kernel<8>, dim3(grid), dim3(64), lds, m->stream, m->X, (long long)m->N, m->D, nbuf, (const double*)m->d_mean,
                       tab, tab + (size_t)m->D * TP, t, (const double*)m->d_xsq, closest, (double*)m->d_seed[dst], (double*)m->d_part);
  }
  ZK_HIP(hipGetLastError());
  double pot[8];
  if ((rc = reduce_to_host(m, grid, TP, pot))) return rc;
  for (int c = 0; c < t; ++c) pot_out[c] = pot[c];
  m->seed_t = t;
  m->seed_last = dst;  // seed_cur changes when zk_kmeans_seed_pick adopts one of these rows
  m->h_buf.clear();
  return 0;
}

// Adopt candidate `which` of the last step as the closest-distance row, then idx_out[i] = searchsorted(cumsum(closest),
// vals[i]) (side 'left', clipped to N - 1) for the next step's draws (n_vals may be 0).
extern "C" int zk_kmeans_seed_pick(zk_rows* m, int which, const double* vals, int n_vals, int64_t* idx_out) {
  if (!m || m->seed_last < 0 || which < 0 || which >= m->seed_t) return zk_fail(ZK_E_BADARG, "no seeding step to pick from");
  if (n_vals < 0 || (n_vals > 0 && (!vals || !idx_out))) return zk_fail(ZK_E_BADARG, "bad arguments");
  ZK_ON_DEVICE(m->device);
  m->seed_cur = m->seed_last;
  m->d_closest = (const double*)m->d_seed[m->seed_cur] + (size_t)which * m->N;
  if (n_vals == 0) return 0;
  const long long nb = (m->N + 1023) / 1024;
  int rc = ensure(&m->d_red, &m->red_bytes, (size_t)nb * sizeof(double));
  if (rc) return rc;
  hipLaunchKernelGGL(blocksum_kernel, dim3((unsigned)nb), dim3(256), 0, m->stream, m->d_closest, (long long)m->N, (double*)m->d_red);
  ZK_HIP(hipGetLastError());
  std::vector<double> bs(nb), cum(nb), blk(1024);
  ZK_HIP(hipMemcpyAsync(bs.data(), m->d_red, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, m->stream));
  ZK_HIP(hipStreamSynchronize(m->stream));
  double run = 0.0;
  for (long long b = 0; b < nb; ++b) cum[b] = (run += bs[b]);
  for (int i = 0; i < n_vals; ++i) {
    long long b = std::lower_bound(cum.begin(), cum.end(), vals[i]) - cum.begin();
    long long found = m->N;  // past the end: clipped below
    while (b < nb && found == m->N) {
      const long long first = b * 1024, cnt = std::min<long long>(1024, m->N - first);
      ZK_HIP(hipMemcpy(blk.data(), m->d_closest + first, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost));
      double s = b ? cum[b - 1] : 0.0;
      for (long long e = 0; e < cnt; ++e) {
        s += blk[e];
        if (s >= vals[i]) {
          found = first + e;
          break;
        }
      }
      ++b;
    }
    idx_out[i] = std::min<long long>(found, m->N - 1);
  }
  return 0;
}

// One Lloyd pass with the centred centres (k, D): labels are rewritten on the device; with `update` sums_out (k, D) = sum
// of the centred rows of each cluster and counts_out (k); n_changed_out = rows whose label changed.
extern "C" int zk_kmeans_step(zk_rows* m, const double* centers, int k, int update, double* sums_out, double* counts_out,
                              int64_t* n_changed_out) {
  if (!m || !centers || !n_changed_out || (update && (!sums_out || !counts_out))) return zk_fail(ZK_E_BADARG, "null pointer");
  if (k < 1 || k > 256) return zk_fail(ZK_E_BADARG, "1 to 256 clusters");
  ZK_ON_DEVICE(m->device);
  const int KP = (k + 3) & ~3, D1 = m->D + 1;
  pack_vectors(centers, k, m->D, KP, true, m->h_buf);
  int rc = upload_tab(m, m->h_buf);
  m->h_buf.clear();
  if (rc) return rc;
  const int kmax = k <= 4 ? 4 : k <= 8 ? 8 : k <= 16 ? 16 : 0;  // per-cluster sums in registers up to 16 clusters
  const size_t sums_lds = update && !kmax ? (size_t)k * D1 * sizeof(double) : 0;
  const int nbuf = tile_bufs(m, sums_lds);
  const size_t lds = tile_lds(m, nbuf) + sums_lds;
  if ((rc = check_lds(lds))) return rc;
  const int grid = row_grid(m, lds);
  if (!m->d_labels) {
    ZK_HIP(hipMalloc((void**)&m->d_labels, (size_t)m->N * sizeof(int32_t)));
    ZK_HIP(hipMemsetAsync(m->d_labels, 0xff, (size_t)m->N * sizeof(int32_t), m->stream));  // -1, as sklearn's labels_old
  }
  if (update && (rc = ensure(&m->d_part, &m->part_bytes, (size_t)grid * k * D1 * sizeof(double)))) return rc;
  ZK_HIP(hipMemsetAsync(m->d_count, 0, sizeof(unsigned long long), m->stream));
  const double* tab = (const double*)m->d_tab;
#define ZK_LLOYD(KM)                                                                                                              \
  {                                                                                                                               \
    if ((rc = allow_lds(lloyd_kernel<KM>, lds))) return rc;                                                                       \
    hipLaunchKernelGGL(lloyd_kernel<KM>, dim3(grid), dim3(64), lds, m->stream, m->X, (long long)m->N, m->D, nbuf,                 \
                       (const double*)m->d_mean, tab, tab + (size_t)m->D * KP, KP, k, m->d_labels, update, (double*)m->d_part,    \
                       m->d_count);                                                                                               \
  }
  if ((rc = prof_begin(m))) return rc;
  switch (kmax) {
    case 4: ZK_LLOYD(4) break;
    case 8: ZK_LLOYD(8) break;
    case 16: ZK_LLOYD(16) break;
    default: ZK_LLOYD(0) break;
  }
#undef ZK_LLOYD
  ZK_HIP(hipGetLastError());
  if ((rc = prof_end(m))) return rc;
  unsigned long long chg = 0;
  if (update) {
    std::vector<double> red((size_t)k * D1);
    if ((rc = reduce_to_host(m, grid, k * D1, red.data(), &chg))) return rc;  // sums and the changed-label count: one wait
    for (int c = 0; c < k; ++c) {
      for (int j = 0; j < m->D; ++j) sums_out[(size_t)c * m->D + j] = red[(size_t)c * D1 + j];
      counts_out[c] = red[(size_t)c * D1 + m->D];
    }
  } else {
    ZK_HIP(hipMemcpyAsync(&chg, m->d_count, sizeof(chg), hipMemcpyDeviceToHost, m->stream));
    ZK_HIP(hipStreamSynchronize(m->stream));
  }
  *n_changed_out = (int64_t)chg;
  return prof_read(m);
}

// HIP-event timing of the main kernel of the next zk_kmeans_step calls (bench.py's roofline of the Lloyd pass)
extern "C" int zk_rows_profile(zk_rows* m, int enable) {
  if (!m) return zk_fail(ZK_E_BADARG, "null pointer");
  ZK_ON_DEVICE(m->device);
  if (enable && !m->ev[0]) {
    ZK_HIP(hipEventCreate(&m->ev[0]));
    ZK_HIP(hipEventCreate(&m->ev[1]));
  }
  m->profile = enable != 0;
  return 0;
}
extern "C" double zk_rows_last_kernel_ms(const zk_rows* m) { return m ? m->last_kernel_ms : 0.0; }

// forget the labels (a new run starts from labels_old = -1)
extern "C" int zk_rows_reset_labels(zk_rows* m) {
  if (!m) return zk_fail(ZK_E_BADARG, "null pointer");
  ZK_ON_DEVICE(m->device);
  if (m->d_labels) ZK_HIP(hipMemsetAsync(m->d_labels, 0xff, (size_t)m->N * sizeof(int32_t), m->stream));
  return 0;
}

extern "C" int zk_rows_labels(zk_rows* m, int32_t* labels_host) {
  if (!m || !labels_host) return zk_fail(ZK_E_BADARG, "null pointer");
  if (!m->d_labels) return zk_fail(ZK_E_BADARG, "no labels yet");
  ZK_ON_DEVICE(m->device);
  ZK_HIP(hipMemcpyAsync(labels_host, m->d_labels, (size_t)m->N * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
  ZK_HIP(hipStreamSynchronize(m->stream));
  return 0;
}

extern "C" const int32_t* zk_rows_labels_dev(const zk_rows* m) { return m ? m->d_labels : nullptr; }

// squared distance of every centred row to the centre of its label (k, D) -> host (N): the empty-cluster relocation
extern "C" int zk_kmeans_own_distance(zk_rows* m, const double* centers, int k, double* dist_host) {
  if (!m || !centers || !dist_host || k < 1) return zk_fail(ZK_E_BADARG, "bad arguments");
  if (!m->d_labels) return zk_fail(ZK_E_BADARG, "no labels yet");
  ZK_ON_DEVICE(m->device);
  m->h_buf.assign(centers, centers + (size_t)k * m->D);
  int rc = upload_tab(m, m->h_buf);
  m->h_buf.clear();
  if (rc) return rc;
  const int nbuf = tile_bufs(m);
  const size_t lds = tile_lds(m, nbuf);
  if ((rc = check_lds(lds)) || (rc = allow_lds(owndist_kernel, lds))) return rc;
  const int dst = 1 - m->seed_cur;  // the seeding scratch is free by now
  if ((rc = ensure(&m->d_seed[dst], &m->seed_bytes[dst], (size_t)m->N * sizeof(double)))) return rc;
  hipLaunchKernelGGL(owndist_kernel, dim3(row_grid(m, lds)), dim3(64), lds, m->stream, m->X, (long long)m->N, m->D, nbuf,
                     (const double*)m->d_mean, (const double*)m->d_tab, (const int32_t*)m->d_labels, (double*)m->d_seed[dst]);
  ZK_HIP(hipGetLastError());
  ZK_HIP(hipMemcpyAsync(dist_host, m->d_seed[dst], (size_t)m->N * sizeof(double), hipMemcpyDeviceToHost, m->stream));
  ZK_HIP(hipStreamSynchronize(m->stream));
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Gaussian mixture
// ------------------------------------------------------------------------------------------------------------------
// E step with the upper-triangular precision Cholesky factors prec_chol (k, D, D), means (k, D), log-determinants and log
// weights (k each).  Writes the responsibilities (k, N) and the labels (first argmax) on the device;
// lse_sum_out = sum_r logsumexp_c(weighted log prob).
extern "C" int zk_gmm_estep(zk_rows* m, const double* prec_chol, const double* means, const double* log_det, const double* log_w, int k,
                            int want_resp, double* lse_sum_out) {
  if (!m || !prec_chol || !means || !log_det || !log_w || !lse_sum_out) return zk_fail(ZK_E_BADARG, "null pointer");
  if (k < 1 || k > 64) return zk_fail(ZK_E_BADARG, "1 to 64 mixture components");
  ZK_ON_DEVICE(m->device);
  const int D = m->D, DP = (D + ZK_EW - 1) / ZK_EW * ZK_EW, n_blk = DP / ZK_EW;
  // the (component, column block) pairs, dealt to the waves: longest first, each to the least loaded wave
  const int plan_stride = 1 + 2 * k * n_blk;
  std::vector<int> plan((size_t)ZK_EWAVES * plan_stride, 0);
  {
    long long load[ZK_EWAVES] = {0};
    for (int jb = n_blk - 1; jb >= 0; --jb)
      for (int c = 0; c < k; ++c) {
        int w = 0;
        for (int v = 1; v < ZK_EWAVES; ++v)
          if (load[v] < load[w]) w = v;
        int* mine = plan.data() + (size_t)w * plan_stride;
        mine[1 + 2 * mine[0]] = c;
        mine[2 + 2 * mine[0]] = jb * ZK_EW;
        ++mine[0];
        load[w] += std::min(D, (jb + 1) * ZK_EW) + 4;
      }
  }
  const size_t plan_doubles = (plan.size() * sizeof(int) + 7) / 8;
  std::vector<double>& h = m->h_buf;
  h.assign((size_t)k * D * DP + (size_t)k * DP + 2 * (size_t)k + plan_doubles, 0.0);
  double* P = h.data();
  double* B = P + (size_t)k * D * DP;
  double* C = B + (size_t)k * DP;
  memcpy(C + 2 * (size_t)k, plan.data(), plan.size() * sizeof(int));
  for (int c = 0; c < k; ++c) {
    const double* pc = prec_chol + (size_t)c * D * D;
    for (int i = 0; i < D; ++i)
      for (int j = i; j < D; ++j) P[((size_t)c * D + i) * DP + j] = pc[(size_t)i * D + j];
    for (int j = 0; j < D; ++j) {  // (mu P)_j, the order of numpy.dot(mu, prec_chol)
      double s = 0.0;
      for (int i = 0; i <= j; ++i) s += means[(size_t)c * D + i] * pc[(size_t)i * D + j];
      B[(size_t)c * DP + j] = s;
    }
    C[2 * c] = log_det[c];
    C[2 * c + 1] = log_w[c];
  }
  // matrix-core form (D <= 48, k <= 8: the factors of all components fit the LDS of a CU beside the waves' row blocks)
  static const bool no_mfma = getenv("ZK_ESTEP_VALU") != nullptr;  // A/B runs: the scalar-operand kernel
  const int NB = (D + 15) / 16;
  if (NB <= 3 && k <= 8 && D >= 2 && !no_mfma) {
    const int psteps = 2 * NB * (NB + 1);
    bool diag = true;  // only the diagonal pieces hold anything: 'diag' / 'spherical' factors
    for (int c = 0; c < k && diag; ++c)
      for (int i = 0; i < D && diag; ++i)
        for (int j = i; j < D; ++j)
          if (prec_chol[((size_t)c * D + i) * D + j] != 0.0 && i / 16 != j / 16) {
            diag = false;
            break;
          }
    h.assign((size_t)k * psteps * 64 + (size_t)k * NB * 16 + 2 * (size_t)k, 0.0);
    double* TP = h.data();
    double* TB = TP + (size_t)k * psteps * 64;
    double* TC = TB + (size_t)k * NB * 16;
    for (int c = 0; c < k; ++c) {
      const double* pc = prec_chol + (size_t)c * D * D;
      for (int bj = 0; bj < NB; ++bj)
        for (int s = 0; s < 4 * (bj + 1); ++s)
          for (int l = 0; l < 64; ++l) {
            const int i = 4 * s + (l >> 4), j = 16 * bj + (l & 15);
            TP[((size_t)c * psteps + 2 * bj * (bj + 1) + s) * 64 + l] = i < D && j < D && i <= j ? pc[(size_t)i * D + j] : 0.0;
          }
      for (int j = 0; j < D; ++j) {  // -(mu P)_j, the order of numpy.dot(mu, prec_chol)
        double sj = 0.0;
        for (int i = 0; i <= j; ++i) sj += means[(size_t)c * D + i] * pc[(size_t)i * D + j];
        TB[(size_t)c * NB * 16 + j] = -sj;
      }
      TC[2 * c] = log_det[c];
      TC[2 * c + 1] = log_w[c];
    }
    int rc = upload_tab(m, h);
    h.clear();
    if (rc) return rc;
    // twelve waves per workgroup (three per SIMD) while the factors and the waves' row blocks fit a CU's LDS, else eight: the
    // chains are 4 / 8 / 12 MFMAs long, and short dependent chains gain from a third wave (tools/micro_mfma64_chain.hip)
    static const int forced_nw = getenv("ZK_ESTEP_WAVES") ? atoi(getenv("ZK_ESTEP_WAVES")) : 0;
    const size_t lds_tab = ((size_t)k * psteps * 64 + (size_t)k * NB * 16) * sizeof(double);
    const size_t lds_wave = ((size_t)16 * D + 4 + 128) * sizeof(double);
    int nw = lds_tab + 12 * lds_wave <= 160 * 1024 ? 12 : 8;
    if (forced_nw == 8 || (forced_nw == 12 && nw == 12)) nw = forced_nw;
    const size_t lds = lds_tab + nw * lds_wave;
    if ((rc = check_lds(lds))) return rc;
    const long long n_blocks = (m->N + 15) / 16;
    const int grid = (int)std::min<long long>((n_blocks + nw - 1) / nw, (long long)m->n_cu);
    if ((rc = ensure(&m->d_part, &m->part_bytes, (size_t)grid * sizeof(double)))) return rc;
    if (want_resp && (rc = ensure(&m->d_resp, &m->resp_bytes, (size_t)k * m->N * sizeof(double)))) return rc;
    if (!m->d_labels) ZK_HIP(hipMalloc((void**)&m->d_labels, (size_t)m->N * sizeof(int32_t)));
    const double* tab = (const double*)m->d_tab;
    const double* d_tb = tab + (size_t)k * psteps * 64;
    const double* d_tc = d_tb + (size_t)k * NB * 16;
    if ((rc = prof_begin(m))) return rc;
#define ZK_ESTEP_LAUNCH_NW(NBV, FULLV, NWV)                                                                                             \
  {                                                                                                                                     \
    if ((rc = allow_lds(estep_mfma_kernel<NBV, FULLV, NWV>, lds))) return rc;                                                           \
    hipLaunchKernelGGL((estep_mfma_kernel<NBV, FULLV, NWV>), dim3(grid), dim3(64 * NWV), lds, m->stream, m->X, (long long)m->N, D, tab, \
                       d_tb, d_tc, (double)D * std::log(2.0 * M_PI), k, want_resp ? (double*)m->d_resp : nullptr, m->d_labels,          \
                       (double*)m->d_part);                                                                                             \
  }
#define ZK_ESTEP_LAUNCH(NBV, FULLV)                                      \
  {                                                                      \
    if (nw == 12) ZK_ESTEP_LAUNCH_NW(NBV, FULLV, 12) else ZK_ESTEP_LAUNCH_NW(NBV, FULLV, 8) \
  }
    if (NB == 1) {
      if (diag) ZK_ESTEP_LAUNCH(1, false) else ZK_ESTEP_LAUNCH(1, true)
    } else if (NB == 2) {
      if (diag) ZK_ESTEP_LAUNCH(2, false) else ZK_ESTEP_LAUNCH(2, true)
    } else {
      if (diag) ZK_ESTEP_LAUNCH(3, false) else ZK_ESTEP_LAUNCH(3, true)
    }
#undef ZK_ESTEP_LAUNCH
#undef ZK_ESTEP_LAUNCH_NW
    ZK_HIP(hipGetLastError());
    if ((rc = prof_end(m))) return rc;
    if ((rc = reduce_to_host(m, grid, 1, lse_sum_out))) return rc;
    return prof_read(m);
  }
  int rc = upload_tab(m, h);
  h.clear();
  if (rc) return rc;
  const size_t lds = (size_t)TILE * D * sizeof(double) + (size_t)ZK_EWAVES * k * 64 * sizeof(double);
  if ((rc = check_lds(lds)) || (rc = allow_lds(estep_kernel, lds))) return rc;
  int per_cu = (int)std::min<size_t>((160 * 1024) / (lds + 512), (size_t)(32 / ZK_EWAVES));
  per_cu = std::max(1, per_cu);
  const int grid = (int)std::min<long long>((m->N + TILE - 1) / TILE, (long long)per_cu * m->n_cu);
  if ((rc = ensure(&m->d_part, &m->part_bytes, (size_t)grid * sizeof(double)))) return rc;
  if (want_resp && (rc = ensure(&m->d_resp, &m->resp_bytes, (size_t)k * m->N * sizeof(double)))) return rc;
  if (!m->d_labels) ZK_HIP(hipMalloc((void**)&m->d_labels, (size_t)m->N * sizeof(int32_t)));
  const double* tab = (const double*)m->d_tab;
  const double* d_cst = tab + (size_t)k * D * DP + (size_t)k * DP;
  hipLaunchKernelGGL(estep_kernel, dim3(grid), dim3(64 * ZK_EWAVES), lds, m->stream, m->X, (long long)m->N, D, DP, tab,
                     tab + (size_t)k * D * DP, d_cst, (const int*)(d_cst + 2 * (size_t)k), plan_stride,
                     (double)D * std::log(2.0 * M_PI), k, want_resp ? (double*)m->d_resp : nullptr, m->d_labels, (double*)m->d_part);
  ZK_HIP(hipGetLastError());
  return reduce_to_host(m, grid, 1, lse_sum_out);
}

// responsibilities = one-hot of the current labels (the k-means initialisation of the mixture)
extern "C" int zk_gmm_resp_from_labels(zk_rows* m, int k) {
  if (!m || k < 1 || k > 64) return zk_fail(ZK_E_BADARG, "bad arguments");
  if (!m->d_labels) return zk_fail(ZK_E_BADARG, "no labels yet");
  ZK_ON_DEVICE(m->device);
  int rc = ensure(&m->d_resp, &m->resp_bytes, (size_t)k * m->N * sizeof(double));
  if (rc) return rc;
  hipLaunchKernelGGL(onehot_kernel, dim3((unsigned)((m->N + 255) / 256)), dim3(256), 0, m->stream, (const int32_t*)m->d_labels,
                     (long long)m->N, k, (double*)m->d_resp);
  ZK_HIP(hipGetLastError());
  return 0;
}

// sum_r w_cr [x_r - shift | 1]^T [x_r - shift | 1] for `count` (1 to 3) consecutive planes of the responsibilities starting at
// w_dev, or with unit weights (w_dev null, count 1); gram_out (count, D+1, D+1)
static int rows_gram(zk_rows* m, const double* w_dev, int count, const double* shift, double* gram_out) {
  const int D = m->D, D1 = D + 1, T = (D1 + 3) / 4, n_ut = T * (T + 1) / 2;
  m->h_buf.assign(shift, shift + D);
  int rc = upload_tab(m, m->h_buf);
  m->h_buf.clear();
  if (rc) return rc;
  static const bool no_mfma = getenv("ZK_WGRAM_VALU") != nullptr;  // A/B runs: the register-tiled kernel
  if (D1 <= 48 && D >= 2 && !no_mfma) {
    // matrix-core form: every component of the call in one pass; `sets` groups of six waves (one per block of the triangle),
    // each set walking its share of the tiles
    const size_t lds = (size_t)4 * (2 * TILE * 16 + count * TILE) * sizeof(double);
    const long long n_tiles = (m->N + TILE - 1) / TILE;
    const long long sets = std::max<long long>(1, std::min<long long>(n_tiles, (long long)m->n_cu * 8 / 6));
    const long long waves = sets * 6, blocks = (waves + 3) / 4;
    const int n = 6 * count * 256;
    if ((rc = ensure(&m->d_part, &m->part_bytes, (size_t)sets * n * sizeof(double)))) return rc;
    if ((rc = prof_begin(m))) return rc;
#define ZK_WGRAM_M(CC)                                                                                                      \
  {                                                                                                                         \
    if ((rc = allow_lds(wgram_mfma_kernel<CC>, lds))) return rc;                                                            \
    hipLaunchKernelGGL(wgram_mfma_kernel<CC>, dim3((unsigned)blocks), dim3(256), lds, m->stream, m->X, (long long)m->N, D,  \
                       (int)sets, (const double*)m->d_tab, w_dev, (double*)m->d_part);                                      \
  }
    switch (count) {
      case 1: ZK_WGRAM_M(1) break;
      case 2: ZK_WGRAM_M(2) break;
      case 3: ZK_WGRAM_M(3) break;
      case 4: ZK_WGRAM_M(4) break;
      case 5: ZK_WGRAM_M(5) break;
      case 6: ZK_WGRAM_M(6) break;
      case 7: ZK_WGRAM_M(7) break;
      default: ZK_WGRAM_M(8) break;
    }
#undef ZK_WGRAM_M
    ZK_HIP(hipGetLastError());
    if ((rc = prof_end(m))) return rc;
    // fixed-order sum over the sets, then the six blocks are put in place (upper triangle mirrored; inside a diagonal block
    // only i <= j is taken: its two halves differ in the last bit)
    std::vector<double> blocks_sum((size_t)n);
    if ((rc = reduce_to_host(m, (int)sets, n, blocks_sum.data()))) return rc;
    static const int BI[6] = {0, 0, 0, 1, 1, 2}, BJ[6] = {0, 1, 2, 1, 2, 2};
    for (int b = 0; b < 6; ++b)
      for (int c = 0; c < count; ++c) {
        const double* src = blocks_sum.data() + ((size_t)b * count + c) * 256;
        double* dst = gram_out + (size_t)c * D1 * D1;
        for (int i = 0; i < 16; ++i)
          for (int j = 0; j < 16; ++j) {
            const int gi = 16 * BI[b] + i, gj = 16 * BJ[b] + j;
            if (gi < D1 && gj < D1 && gi <= gj) dst[gi * D1 + gj] = dst[gj * D1 + gi] = src[i * 16 + j];
          }
      }
    return prof_read(m);
  }
  if (count > 3) return zk_fail(ZK_E_BADARG, "more than 3 components per pass need D <= 47");
  const int threads = std::max(256, (n_ut + 63) & ~63);
  const int G = std::min(threads / n_ut, 16);
  const size_t lds = ((size_t)64 * (4 * T + 4) + 64 * count) * sizeof(double);
  int per_cu = (int)std::min<size_t>((160 * 1024) / (lds + 512), (size_t)(2048 / threads));
  per_cu = std::max(1, std::min(per_cu, count > 1 ? 4 : 6));
  long long blocks = std::min<long long>((m->N + 63) / 64, (long long)per_cu * m->n_cu);
  if ((rc = ensure(&m->d_part, &m->part_bytes, (size_t)blocks * G * count * D1 * D1 * sizeof(double)))) return rc;
  if ((rc = prof_begin(m))) return rc;
#define ZK_WGRAM(CC)                                                                                                        \
  {                                                                                                                         \
    if ((rc = allow_lds(wgram_kernel<CC>, lds))) return rc;                                                                 \
    hipLaunchKernelGGL(wgram_kernel<CC>, dim3((unsigned)blocks), dim3(threads), lds, m->stream, m->X, (long long)m->N, D, T, \
                       G, (const double*)m->d_tab, w_dev, (double*)m->d_part);                                              \
  }
  switch (count) {
    case 1: ZK_WGRAM(1) break;
    case 2: ZK_WGRAM(2) break;
    default: ZK_WGRAM(3) break;
  }
#undef ZK_WGRAM
  ZK_HIP(hipGetLastError());
  if ((rc = prof_end(m))) return rc;
  if ((rc = reduce_to_host(m, (int)(blocks * G), count * D1 * D1, gram_out))) return rc;
  return prof_read(m);
}

// M-step sums about `shift` (D) of `count` (1 to 3) consecutive components starting at c: gram_out (count, D+1, D+1), each
// sum_r resp[c][r] [x_r - shift | 1]^T [x_r - shift | 1] -- second moments, first moments in the last row / column, the
// component's weight in the corner.
extern "C" int zk_gmm_moments(zk_rows* m, int c, int count, const double* shift, double* gram_out) {
  if (!m || !shift || !gram_out) return zk_fail(ZK_E_BADARG, "null pointer");
  if (count < 1 || count > 8 || (count > 3 && m && m->D > 47)) return zk_fail(ZK_E_BADARG, "1 to 8 components per pass (1 to 3 with more than 47 features)");
  if (!m->d_resp || c < 0 || (size_t)(c + count) * m->N * sizeof(double) > m->resp_bytes) return zk_fail(ZK_E_BADARG, "no such component");
  ZK_ON_DEVICE(m->device);
  return rows_gram(m, (const double*)m->d_resp + (size_t)c * m->N, count, shift, gram_out);
}

// The same sums with unit weights: everything a covariance needs (pca), in a fixed summation order.
extern "C" int zk_rows_gram(zk_rows* m, const double* shift, double* gram_out) {
  if (!m || !shift || !gram_out) return zk_fail(ZK_E_BADARG, "null pointer");
  ZK_ON_DEVICE(m->device);
  return rows_gram(m, nullptr, 1, shift, gram_out);
}
