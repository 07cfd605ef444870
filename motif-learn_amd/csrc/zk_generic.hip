// zk_generic.hip -- unfolded direct-summation kernels for any (size, n_max, dtype).
//
// One lane owns one output unit (a patch in batch mode, an output pixel in frame mode) and
// walks the disk pixels in row-major order.  The loop index is wave-uniform, so the pixel
// coordinates and the CHUNK basis values of the current pixel come in through scalar
// loads (SGPR operands of v_fma_f64) and cost no vector-memory or LDS bandwidth; the only
// per-lane traffic is the pixel itself.  In frame mode consecutive lanes read consecutive
// image columns (coalesced, L2-resident frame); in batch mode lanes are one patch apart,
// which is TCP-inefficient -- this kernel is the correctness fallback for shapes the
// folded kernels (zk_fast_*.hip) do not cover, not the measured hot path.
//
// Arithmetic: out = sum_t pixel(t) * (basis[j][t] / area), accumulated in float64 in disk
// order -- the definition both reference paths approximate (_zps.py:155, :165-178).  Dense mode of a
// point-symmetric set (zk_plan::conv_flip) pairs the table rows with the window the way the reference's
// convolution does: pixel (K-1-r, K-1-c) with row (r, c), times (-1)^n.
#include "zk_internal.h"

#ifndef ZK_GEN_PB
#define ZK_GEN_PB 8  // pixels in flight per lane
#endif
#define ZK_TAB __attribute__((address_space(4)))

namespace {

template <typename T, int MODE, int CHUNK>  // MODE 0: batch of patches, 1: dense frame; CHUNK: functions per pass
__global__ __launch_bounds__(256) void zk_generic_kernel(
    const T* __restrict__ in, double* __restrict__ out, const int2* __restrict__ pix,
    const double* __restrict__ tab, int npx, int n_poly, int n_chunks, int size, long long n_units,
    int H, int W, int row0, long long plane, const double* __restrict__ sign, int flip) {
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool live = u < n_units;
  const int ea = size - 1 - (size - 1) / 2;
  int oi = 0, ok = 0;             // frame mode: output pixel
  const T* base = in;             // batch mode: this lane's patch
  if (MODE == 1) {
    const long long il = live ? u / W : 0;
    ok = live ? (int)(u - il * W) : 0;
    oi = row0 + (int)il;
  } else {
    base = in + (live ? u : 0) * (long long)size * size;
  }

  const ZK_TAB int* cpix = (const ZK_TAB int*)pix;        // wave-uniform tables through the constant address space:
  const ZK_TAB double* ctab = (const ZK_TAB double*)tab;  // always scalar loads
  for (int c = 0; c < n_chunks; ++c) {
    double acc[CHUNK];
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) acc[j] = 0.0;
    const ZK_TAB double* __restrict__ row = ctab + (size_t)c * npx * CHUNK;
    // ZK_GEN_PB pixels are requested before the first of them is used: written one pixel at a time, every pixel exposed a
    // whole memory latency (uncoalesced in batch mode) in front of its CHUNK FMAs -- 3 % of the FP64 peak at n_max 36
    // (profiles/r03_high_orders.txt).  Same pixels in the same order: results are bit-identical.
    // (dense mode only: in batch mode the lanes' loads are one patch apart, eight of them in flight per lane thrash the
    //  L1 -- 1.32 -> 0.71 M patches/s at (28, 56); batches of these orders go through zk_direct_patches.hip)
    constexpr int PB = MODE == 1 ? ZK_GEN_PB : 1;
    for (int t0 = 0; t0 < npx; t0 += PB) {
      double f[PB];
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        const int t = t0 + q < npx ? t0 + q : npx - 1;  // (wave-uniform) past the end: the last pixel again, weighted 0 below
        const int rcx = cpix[2 * t], rcy = cpix[2 * t + 1];
        if (MODE == 1) {
          // (flip: the reference's dense path is a convolution -- table row (r, c) meets window pixel (K-1-r, K-1-c), the
          //  (-1)^n of _zps.py:173-178 is applied at the store; zk_plan::conv_flip)
          const int ii = oi - ea + (flip ? size - 1 - rcx : rcx);
          const int kk = ok - ea + (flip ? size - 1 - rcy : rcy);
          const bool inside = live && ii >= 0 && ii < H && kk >= 0 && kk < W && t0 + q < npx;
          f[q] = inside ? (double)in[(long long)ii * W + kk] : 0.0;
        } else {
          f[q] = t0 + q < npx ? (double)base[rcx * size + rcy] : 0.0;
        }
      }
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        const int t = t0 + q < npx ? t0 + q : npx - 1;
        const ZK_TAB double* __restrict__ b = row + (size_t)t * CHUNK;
#pragma unroll
        for (int j = 0; j < CHUNK; ++j) acc[j] = __builtin_fma(f[q], b[j], acc[j]);
      }
    }
    if (live) {
#pragma unroll
      for (int j = 0; j < CHUNK; ++j) {
        const int jj = c * CHUNK + j;
        if (jj < n_poly) {
          if (MODE == 1) {
            // (n_poly, n_rows, W): u already enumerates (row, col) of the band
            out[(long long)jj * plane + u] = acc[j] * ((const ZK_TAB double*)sign)[jj];
          } else {
            out[u * n_poly + jj] = acc[j];
          }
        }
      }
    }
  }
}

// Windows at key points -> (N, K, K) batch: the slice img[y-s1:y+s2, x-s1:x+s2] of the reference
// (features/_keypoint.py:60-78; s1 = K//2, s2 = K - K//2), zero outside the frame.  One thread per pixel,
// consecutive lanes along a window row.
template <typename T>
__global__ __launch_bounds__(256) void zk_gather_points_kernel(const T* __restrict__ img, const int32_t* __restrict__ pts,
                                                               T* __restrict__ out, int K, int H, int W,
                                                               long long n_points) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int kk = K * K;
  const long long p = t / kk;
  if (p >= n_points) return;
  const int q = (int)(t - p * kk), r = q / K, c = q - r * K;
  const int yy = pts[2 * p + 1] - K / 2 + r, xx = pts[2 * p] - K / 2 + c;
  out[t] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? img[(long long)yy * W + xx] : (T)0;
}

template <int MODE>
int launch(zk_plan* p, const void* in, int dtype, long long n_units, int H, int W, int row0,
           double* out, hipStream_t s, long long plane = 0) {
  if (n_units <= 0) return 0;
  const long long blocks = (n_units + 255) / 256;
  if (blocks > 0x7fffffffLL) return zk_fail(ZK_E_BADARG, "too many units for one launch");
  int rc = zk_prof_begin(p, s);
  if (rc) return rc;
#define ZK_GEN_LAUNCH(T, CH)                                                                              \
  hipLaunchKernelGGL((zk_generic_kernel<T, MODE, CH>), dim3((unsigned)blocks), dim3(256), 0, s, (const T*)in, out, \
                     p->d_pix, p->d_gen_tab, p->npx, p->n_poly, p->n_chunks, p->size, n_units, H, W, row0, plane, \
                     p->d_sign, (int)(MODE == 1 && p->conv_flip))
  if (dtype == ZK_F32) {
    if (p->gen_chunk == 64) ZK_GEN_LAUNCH(float, 64);
    else ZK_GEN_LAUNCH(float, 32);
  } else {
    if (p->gen_chunk == 64) ZK_GEN_LAUNCH(double, 64);
    else ZK_GEN_LAUNCH(double, 32);
  }
#undef ZK_GEN_LAUNCH
  ZK_HIP(hipGetLastError());
  return zk_prof_end(p, s);
}

}  // namespace

int zk_launch_gather_points(zk_plan* p, const void* img, int dtype, int64_t H, int64_t W, const int32_t* pts,
                            int64_t n_points, void* patches, hipStream_t s) {
  const long long threads = (long long)n_points * p->size * p->size;
  const long long blocks = (threads + 255) / 256;
  if (blocks > 0x7fffffffLL) return zk_fail(ZK_E_BADARG, "too many points for one launch");
  if (dtype == ZK_F32)
    hipLaunchKernelGGL(zk_gather_points_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)img, pts,
                       (float*)patches, p->size, (int)H, (int)W, (long long)n_points);
  else
    hipLaunchKernelGGL(zk_gather_points_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s, (const double*)img, pts,
                       (double*)patches, p->size, (int)H, (int)W, (long long)n_points);
  ZK_HIP(hipGetLastError());
  return 0;
}

int zk_launch_generic_patches(zk_plan* p, const void* in, int dtype, int64_t n_patches, double* out,
                              hipStream_t s) {
  return launch<0>(p, in, dtype, n_patches, 0, 0, 0, out, s);
}

int zk_launch_generic_frame(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0,
                            int64_t n_rows, double* out, hipStream_t s) {
  // (n_poly, n_rows, W) unless the plan carries an output plane stride; <= 2^31 blocks of 256 pixels per launch
  const long long plane = zk_out_plane(p, n_rows, W);
  const int64_t cap = W > 0 ? (int64_t)0x7fffffffLL * 256 / W : n_rows;
  for (int64_t b0 = 0; b0 < n_rows; b0 += cap) {
    const int64_t nb = n_rows - b0 < cap ? n_rows - b0 : cap;
    const int rc = launch<1>(p, in, dtype, nb * W, (int)H, (int)W, (int)(row0 + b0), out + b0 * W, s, plane);
    if (rc) return rc;
  }
  return 0;
}
