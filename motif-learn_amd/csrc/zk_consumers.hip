// zk_consumers.hip -- the first downstream consumer of the moment matrix, on the device:
//   pca(X, n_components)  reference features/_dimension_reduction.py:3-6 = sklearn.decomposition.PCA(n).fit_transform(X)
// so that a moment matrix that is already resident in HBM (or just arrived there) does not travel to the host for its
// covariance pass and back for the projection.  Two streaming kernels; the D x D symmetric eigen-problem in between
// (45 x 45 at n_max 8) is LAPACK's on the host, as in scikit-learn's own "covariance_eigh" solver, whose arithmetic the
// Python wrapper follows (mtflearn_amd/features/consumers.py).
#include "zk_internal.h"

namespace {

// G = [X | 1]^T [X | 1]  (D+1 x D+1, row-major, accumulated with atomics): the Gram matrix with the column sums in its
// last row / column and N in the corner.  A workgroup of T x T threads owns the (4T x 4T)-padded G as 4 x 4 register
// tiles and walks row tiles of 64 rows staged in LDS (rows zero-padded to 4T features, the constant 1 appended).
__global__ __launch_bounds__(1024) void gram_kernel(const double* __restrict__ X, long long N, int D, int T, double* __restrict__ G) {
  extern __shared__ __attribute__((aligned(16))) double tile[];  // [64][4T]
  const int P = 4 * T;
  const int ti = threadIdx.x / T, tj = threadIdx.x % T;
  const int nthreads = T * T;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (long long r0 = (long long)blockIdx.x * 64; r0 < N; r0 += (long long)gridDim.x * 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * P; e += nthreads) {
      const int r = e / P, c = e - r * P;
      double v = 0.0;
      if (r0 + r < N) v = c < D ? X[(r0 + r) * D + c] : (c == D ? 1.0 : 0.0);
      tile[e] = v;
    }
    __syncthreads();
    if (ti <= tj) {  // symmetric: the upper triangle of tiles only
#pragma unroll 4
      for (int r = 0; r < 64; ++r) {
        const double* row = tile + r * P;
        double xi[4], xj[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) xi[a] = row[4 * ti + a], xj[a] = row[4 * tj + a];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_fma(xi[a], xj[b], acc[a][b]);
      }
    }
  }
  if (ti <= tj) {
    const int D1 = D + 1;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = 4 * ti + a, j = 4 * tj + b;
        if (i < D1 && j < D1 && i <= j) {
          atomicAdd(&G[i * D1 + j], acc[a][b]);
          if (i != j) atomicAdd(&G[j * D1 + i], acc[a][b]);
        }
      }
  }
}

// Y[r][c] = sum_i (X[r][i] - mean[i]) comp[c][i]: one row per lane, the k x D components and the mean wave-uniform
template <int KMAX>
__global__ __launch_bounds__(256) void project_kernel(const double* __restrict__ X, long long N, int D, const double* __restrict__ mean,
                                                      const double* __restrict__ comp, int k, double* __restrict__ Y) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= N) return;
  const double* x = X + r * D;
  double y[KMAX];
#pragma unroll
  for (int c = 0; c < KMAX; ++c) y[c] = 0.0;
  for (int i = 0; i < D; ++i) {
    const double v = x[i] - mean[i];
#pragma unroll
    for (int c = 0; c < KMAX; ++c)
      if (c < k) y[c] = __builtin_fma(v, comp[c * D + i], y[c]);
  }
  for (int c = 0; c < k; ++c) Y[r * k + c] = y[c];
}

}  // namespace

extern "C" int zk_gram_dev(int device, const double* X_dev, int64_t N, int D, double* gram_dev, void* hip_stream) {
  if (!X_dev || !gram_dev || N <= 0 || D <= 0 || D > 127) return zk_fail(ZK_E_BADARG, "need 1 <= D <= 127 features and N > 0 rows");
  ZK_ON_DEVICE(device);
  hipStream_t s = (hipStream_t)hip_stream;
  const int D1 = D + 1, T = (D1 + 3) / 4;
  ZK_HIP(hipMemsetAsync(gram_dev, 0, (size_t)D1 * D1 * sizeof(double), s));
  int n_cu = 256;
  (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device);
  long long blocks = (N + 63) / 64;
  if (blocks > 2LL * n_cu) blocks = 2LL * n_cu;
  const size_t lds = (size_t)64 * 4 * T * sizeof(double);
  hipLaunchKernelGGL(gram_kernel, dim3((unsigned)blocks), dim3(T * T), lds, s, X_dev, (long long)N, D, T, gram_dev);
  ZK_HIP(hipGetLastError());
  return 0;
}

extern "C" int zk_project_dev(int device, const double* X_dev, int64_t N, int D, const double* mean_dev, const double* comp_dev, int k,
                              double* Y_dev, void* hip_stream) {
  if (!X_dev || !mean_dev || !comp_dev || !Y_dev || N <= 0 || D <= 0 || k <= 0 || k > 16)
    return zk_fail(ZK_E_BADARG, "need N, D > 0 and 1 <= k <= 16 components");
  ZK_ON_DEVICE(device);
  hipLaunchKernelGGL(project_kernel<16>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, X_dev, (long long)N, D,
                     mean_dev, comp_dev, k, Y_dev);
  ZK_HIP(hipGetLastError());
  return 0;
}

// host-buffer forms: X goes up once and stays for both passes of a PCA (handle = the device copy)
extern "C" int zk_gram(int device, const double* X_host, int64_t N, int D, double* gram_host, void** X_dev_out) {
  if (!X_host || !gram_host || !X_dev_out) return zk_fail(ZK_E_BADARG, "null pointer");
  if (N <= 0 || D <= 0 || D > 127) return zk_fail(ZK_E_BADARG, "need 1 <= D <= 127 features and N > 0 rows");
  ZK_ON_DEVICE(device);
  *X_dev_out = nullptr;
  double *d_x = nullptr, *d_g = nullptr;
  const size_t gb = (size_t)(D + 1) * (D + 1) * sizeof(double);
  ZK_HIP(hipMalloc((void**)&d_x, (size_t)N * D * sizeof(double)));
  hipError_t e = hipMalloc((void**)&d_g, gb);
  if (e == hipSuccess) e = hipMemcpy(d_x, X_host, (size_t)N * D * sizeof(double), hipMemcpyHostToDevice);
  int rc = e == hipSuccess ? zk_gram_dev(device, d_x, N, D, d_g, nullptr) : zk_hip_fail(e, "zk_gram staging");
  if (!rc) {
    e = hipMemcpy(gram_host, d_g, gb, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = zk_hip_fail(e, "hipMemcpy(gram)");
  }
  if (d_g) (void)hipFree(d_g);
  if (rc) {
    (void)hipFree(d_x);
    return rc;
  }
  *X_dev_out = d_x;
  return 0;
}

extern "C" int zk_project(int device, const void* X_dev, int64_t N, int D, const double* mean_host, const double* comp_host, int k,
                          double* Y_host, int free_x) {
  if (!X_dev || !mean_host || !comp_host || !Y_host) return zk_fail(ZK_E_BADARG, "null pointer");
  ZK_ON_DEVICE(device);
  if (N <= 0 || D <= 0 || k <= 0 || k > 16) {
    if (free_x) (void)hipFree((void*)X_dev);  // free_x holds on every exit path once X_dev is known
    return zk_fail(ZK_E_BADARG, "need N, D > 0 and 1 <= k <= 16 components");
  }
  double *d_t = nullptr, *d_y = nullptr;
  hipError_t e = hipMalloc((void**)&d_t, (size_t)(D + (size_t)k * D) * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&d_y, (size_t)N * k * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(d_t, mean_host, (size_t)D * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_t + D, comp_host, (size_t)k * D * sizeof(double), hipMemcpyHostToDevice);
  int rc = e == hipSuccess ? zk_project_dev(device, (const double*)X_dev, N, D, d_t, d_t + D, k, d_y, nullptr)
                           : zk_hip_fail(e, "zk_project staging");
  if (!rc) {
    e = hipMemcpy(Y_host, d_y, (size_t)N * k * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = zk_hip_fail(e, "hipMemcpy(Y)");
  }
  if (d_t) (void)hipFree(d_t);
  if (d_y) (void)hipFree(d_y);
  if (free_x) (void)hipFree((void*)X_dev);
  return rc;
}
