// zk_sep_points.hip -- Zernike moments of K x K windows at arbitrary positions of a resident frame
// (SURVEY 8f rank 2): replaces "gather patches at key points, then transform the batch"
// (reference features/_keypoint.py:60-78 followed by _zps.py:146-157) by one kernel that reads the
// windows straight from the frame -- indices in, moments out, no (N, K, K) batch in memory.
//
// One lane owns one point; the window's top-left corner is (y - K/2, x - K/2) for the point (x, y),
// exactly the slice img[y-s1:y+s2, x-s1:x+s2] the reference cuts (s1 = K//2, s2 = K - K//2).  Pixels
// outside the frame read as zero.  Arithmetic: the mirror-folded row-separable sums of zk_sep.h.
// Lanes of a wave read unrelated addresses (4 scalar-width loads per quadrant pixel); the frame is
// L2 / Infinity-Cache resident, so this is a cache-gather kernel, not an HBM stream: its algorithmic
// HBM traffic is 8 B of coordinates in and 8 N_poly B out per point.
#include "zk_sep.h"

// Build groups: the kernel instances are spread over several translation units (Makefile) so that they
// compile in parallel: group 0 = n_max kernels 4..12 (and every non-template entry point), 1 = 14 / 16,
// 2 = 20 / 24 (class-pass kernels).  Group 0's launcher forwards to the others.
#ifndef ZK_NMAX_GROUP
#define ZK_NMAX_GROUP 0
#endif
#if ZK_NMAX_GROUP == 0
#define ZK_GROUP_FN(name) name
#elif ZK_NMAX_GROUP == 1
#define ZK_GROUP_FN(name) name##_g1
#else
#define ZK_GROUP_FN(name) name##_g2
#endif

namespace {

// (two waves per SIMD at n_max 12: sorted points 2.14 -> 1.50 ms per 2^20, random points 3.98 -> 5.77: kept at one)
#ifndef ZK_POINTS_W12
#define ZK_POINTS_W12 1
#endif

template <int NMAX, typename T>
__global__ __launch_bounds__(256, (NMAX == 12 ? ZK_POINTS_W12 : 1)) void zk_points_sep_kernel(
    const T* __restrict__ img, const int32_t* __restrict__ pts, double* __restrict__ out,
    const zk_sep_row* __restrict__ rows, const double* __restrict__ xq, const double* __restrict__ tmat,
    const int32_t* __restrict__ colmap, int n_tab_rows, int K, int H, int W, long long n_points, int n_poly) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool live = p < n_points;
  const long long pc = live ? p : 0;
  const int x0 = pts[2 * pc] - K / 2, y0 = pts[2 * pc + 1] - K / 2;  // top-left corner of the window
  auto px_at = [&](int r, int c) -> double {
    const int yy = y0 + r, xx = x0 + c;
    return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (double)img[(long long)yy * W + xx] : 0.0;
  };

  zk_sep_acc<NMAX> acc;
  acc.clear_all();
  const ZK_CONST int32_t* rtab = zk_const((const int32_t*)rows);
  const ZK_CONST double* px = zk_const(xq);
  const int Q = (K + 1) / 2;
  for (int ri = 0; ri < n_tab_rows; ++ri) {
    const int r = rtab[2 * ri], cmin = rtab[2 * ri + 1];
#pragma unroll 2
    for (int c = cmin; c < Q; ++c)
      acc.pixel(px_at(r, c), px_at(r, K - 1 - c), px_at(K - 1 - r, c), px_at(K - 1 - r, K - 1 - c),
                px + c * ZK_SEP_ROW);
    acc.row_end(px + r * ZK_SEP_ROW);
  }
  const ZK_CONST int32_t* cmap = zk_const(colmap);
  double* __restrict__ dst = out + pc * n_poly;
  acc.transform(zk_const(tmat), [&](auto slot, double z) {
    const int col = cmap[slot];
    if (live && col >= 0) dst[col] = z;
  });
}

template <int NMAX, typename T>
int launch_one(zk_plan* p, const void* img, const int32_t* pts, int64_t H, int64_t W, int64_t n_points, double* out,
               hipStream_t s) {
  const zk_sep_tables* t = p->sep;
  const long long blocks = (n_points + 255) / 256;
  if (blocks > 0x7fffffffLL) return zk_fail(ZK_E_BADARG, "too many points for one launch");
  int rc = zk_prof_begin(p, s);
  if (rc) return rc;
  hipLaunchKernelGGL((zk_points_sep_kernel<NMAX, T>), dim3((unsigned)blocks), dim3(256), 0, s, (const T*)img, pts, out,
                     t->d_rows, t->d_xq, t->d_T, t->d_colmap, t->n_rows, p->size, (int)H, (int)W, (long long)n_points,
                     p->n_poly);
  ZK_HIP(hipGetLastError());
  return zk_prof_end(p, s);
}

template <typename T>
int launch_t(zk_plan* p, const void* img, const int32_t* pts, int64_t H, int64_t W, int64_t n_points, double* out,
             hipStream_t s) {
  switch (p->sep->kernel_nmax) {
#if ZK_NMAX_GROUP == 0
    case 4: return launch_one<4, T>(p, img, pts, H, W, n_points, out, s);
    case 6: return launch_one<6, T>(p, img, pts, H, W, n_points, out, s);
    case 8: return launch_one<8, T>(p, img, pts, H, W, n_points, out, s);
    case 10: return launch_one<10, T>(p, img, pts, H, W, n_points, out, s);
    case 12: return launch_one<12, T>(p, img, pts, H, W, n_points, out, s);
#endif
#if ZK_NMAX_GROUP == 1
    case 14: return launch_one<14, T>(p, img, pts, H, W, n_points, out, s);
    case 16: return launch_one<16, T>(p, img, pts, H, W, n_points, out, s);
#endif
  }
  return zk_fail(ZK_E_BADARG, "no point kernel for this n_max");
}

}  // namespace

#if ZK_NMAX_GROUP == 0
bool zk_sep_points_available(const zk_plan* p, int dtype) {
  (void)dtype;
  return p->sep && p->sep->n_rows > 0 && p->sep->kernel_nmax <= 16;
}
int zk_launch_sep_points_g1(zk_plan* p, const void* img, int dtype, int64_t H, int64_t W, const int32_t* pts,
                            int64_t n_points, double* out, hipStream_t s);
#endif

int ZK_GROUP_FN(zk_launch_sep_points)(zk_plan* p, const void* img, int dtype, int64_t H, int64_t W, const int32_t* pts,
                                      int64_t n_points, double* out, hipStream_t s) {
#if ZK_NMAX_GROUP == 0
  if (!zk_sep_points_available(p, dtype))
    return zk_fail(ZK_E_BADARG, "plan has no key-point kernel (needs the separable tables and n_max <= 16)");
  if (p->sep->kernel_nmax > 12) return zk_launch_sep_points_g1(p, img, dtype, H, W, pts, n_points, out, s);
#endif
  if (dtype == ZK_F32) return launch_t<float>(p, img, pts, H, W, n_points, out, s);
  return launch_t<double>(p, img, pts, H, W, n_points, out, s);
}
