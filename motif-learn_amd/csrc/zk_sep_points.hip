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
//
// Round 4 -- locality.  With the points in the caller's order a wave's 64 lanes read 64 unrelated cache lines per load:
// 148 M points/s for 2^20 random points against 1.19 G/s for the same points sorted by (row, column).  What the sorted
// order buys is that the lanes of a wave sit in one frame row within a few hundred pixels (8 lines per load); a counting
// sort into buckets of one row x 256 columns gives exactly that without ordering anything inside a bucket: a histogram
// of bucket keys (global atomics), an exclusive scan (one workgroup), a scatter of point indices (atomics again) -- three
// small kernels, ~30 us per 2^20 points -- and the moment kernel takes its points through that index list and writes
// every result to the point's ORIGINAL row.  A point's arithmetic does not depend on its lane, so the moments are
// bit-identical to the unbucketed kernel's (ZK_POINTS_NO_BUCKET=1 in the environment: A/B, tests).
#include <stdlib.h>

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

// Round 4 -- whole window rows as 16-byte loads (float32 frames, windows of 8 .. 64 px that lie inside the frame).  The
// scalar-width form issues four 4-byte loads per quadrant pixel (740 per 32-px window): the kernel is bound by the rate of
// vector-memory instructions, not by bytes (0.19 of the L2's gather rate, 0.14 of the FP64 peak).  A window row is K
// contiguous floats, so a row pair is 4 QV unaligned `global_load_dwordx4` (QV = ceil(Q / 4): the left halves ascending, the
// right halves from their far end so that register i of the pair (L, R) holds the mirror columns (c, K-1-c)): 16 loads per
// row pair at K = 32 instead of ~48, every register index static, the next row pair's loads in flight under the current
// pair's arithmetic.  Same operations in the same order on the same values: bit-identical to the scalar-width form, which
// still serves float64 frames, other window sizes and every wave that holds a window crossing the frame's border.
typedef float zk_f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int QV>
struct zk_pair_rows {
  zk_f4u L0[QV], R0[QV], L1[QV], R1[QV];  // rows r and K-1-r: L*[j] = columns 4j .. 4j+3, R*[j] = columns K-4-4j .. K-1-4j
};

template <int NMAX, typename T, int QV>
__global__ __launch_bounds__(256, (NMAX == 12 ? ZK_POINTS_W12 : 1)) void zk_points_sep_kernel(
    const T* __restrict__ img, const int32_t* __restrict__ pts, double* __restrict__ out,
    const zk_sep_row* __restrict__ rows, const double* __restrict__ xq, const double* __restrict__ tmat,
    const int32_t* __restrict__ colmap, int n_tab_rows, int K, int H, int W, long long n_points, int n_poly,
    const int32_t* __restrict__ perm) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool live = p < n_points;
  const long long pc = perm ? (long long)perm[live ? p : 0] : (live ? p : 0);  // bucket order in, original row out
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
  bool wide = false;
  if constexpr (QV > 0 && sizeof(T) == 4) {
    const bool inside = x0 >= 0 && y0 >= 0 && x0 + K <= W && y0 + K <= H;
    wide = __all(inside) != 0;  // wave-uniform: one window across the border sends its whole wave down the scalar-width path
  }
  if constexpr (QV > 0 && sizeof(T) == 4) {
    if (wide) {
      const float* __restrict__ base = (const float*)img + (long long)y0 * W + x0;
      auto load_pair = [&](zk_pair_rows<QV>& w, int r) __attribute__((always_inline)) {
        const float* pa = base + (long long)r * W;
        const float* pb = base + (long long)(K - 1 - r) * W;
#pragma unroll
        for (int j = 0; j < QV; ++j) {
          w.L0[j] = *(const zk_f4u*)(pa + 4 * j);
          w.R0[j] = *(const zk_f4u*)(pa + K - 4 - 4 * j);
          w.L1[j] = *(const zk_f4u*)(pb + 4 * j);
          w.R1[j] = *(const zk_f4u*)(pb + K - 4 - 4 * j);
        }
      };
      auto compute = [&](const zk_pair_rows<QV>& w, int r, int cmin) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 4 * QV; ++c)
          if (c >= cmin && c < Q)  // wave-uniform
            acc.pixel((double)w.L0[c / 4][c % 4], (double)w.R0[c / 4][3 - c % 4], (double)w.L1[c / 4][c % 4],
                      (double)w.R1[c / 4][3 - c % 4], px + c * ZK_SEP_ROW);
        acc.row_end(px + r * ZK_SEP_ROW);
      };
      zk_pair_rows<QV> wa, wb;
      if (n_tab_rows > 0) load_pair(wa, rtab[0]);
      for (int ri = 0; ri < n_tab_rows; ri += 2) {
        if (ri + 1 < n_tab_rows) load_pair(wb, rtab[2 * (ri + 1)]);
        compute(wa, rtab[2 * ri], rtab[2 * ri + 1]);
        if (ri + 1 >= n_tab_rows) break;
        if (ri + 2 < n_tab_rows) load_pair(wa, rtab[2 * (ri + 2)]);
        compute(wb, rtab[2 * (ri + 1)], rtab[2 * (ri + 1) + 1]);
      }
    }
  }
  if (!wide) {
    for (int ri = 0; ri < n_tab_rows; ++ri) {
      const int r = rtab[2 * ri], cmin = rtab[2 * ri + 1];
#pragma unroll 2
      for (int c = cmin; c < Q; ++c)
        acc.pixel(px_at(r, c), px_at(r, K - 1 - c), px_at(K - 1 - r, c), px_at(K - 1 - r, K - 1 - c),
                  px + c * ZK_SEP_ROW);
      acc.row_end(px + r * ZK_SEP_ROW);
    }
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
               const int32_t* perm, hipStream_t s) {
  const zk_sep_tables* t = p->sep;
  const long long blocks = (n_points + 255) / 256;
  if (blocks > 0x7fffffffLL) return zk_fail(ZK_E_BADARG, "too many points for one launch");
  int rc = zk_prof_begin(p, s);
  if (rc) return rc;
#define ZK_POINTS_LAUNCH(QV)                                                                                              \
  hipLaunchKernelGGL((zk_points_sep_kernel<NMAX, T, QV>), dim3((unsigned)blocks), dim3(256), 0, s, (const T*)img, pts, out, \
                     t->d_rows, t->d_xq, t->d_T, t->d_colmap, t->n_rows, p->size, (int)H, (int)W, (long long)n_points,      \
                     p->n_poly, perm)
  // whole-row 16-byte loads: float32 frames, windows of 8 .. 64 px (ZK_POINTS_NO_WIDE=1 in the environment: A/B, tests)
  const int K = p->size;
  const bool wide_ok = sizeof(T) == 4 && K >= 8 && K <= 64 && !getenv("ZK_POINTS_NO_WIDE");
  if constexpr (sizeof(T) == 4) {
    if (wide_ok && K <= 32) ZK_POINTS_LAUNCH(4);
    else if (wide_ok) ZK_POINTS_LAUNCH(8);
    else ZK_POINTS_LAUNCH(0);
  } else {
    ZK_POINTS_LAUNCH(0);
  }
#undef ZK_POINTS_LAUNCH
  ZK_HIP(hipGetLastError());
  return zk_prof_end(p, s);
}

template <typename T>
int launch_t(zk_plan* p, const void* img, const int32_t* pts, int64_t H, int64_t W, int64_t n_points, double* out,
             const int32_t* perm, hipStream_t s) {
  switch (p->sep->kernel_nmax) {
#if ZK_NMAX_GROUP == 0
    case 4: return launch_one<4, T>(p, img, pts, H, W, n_points, out, perm, s);
    case 6: return launch_one<6, T>(p, img, pts, H, W, n_points, out, perm, s);
    case 8: return launch_one<8, T>(p, img, pts, H, W, n_points, out, perm, s);
    case 10: return launch_one<10, T>(p, img, pts, H, W, n_points, out, perm, s);
    case 12: return launch_one<12, T>(p, img, pts, H, W, n_points, out, perm, s);
#endif
#if ZK_NMAX_GROUP == 1
    case 14: return launch_one<14, T>(p, img, pts, H, W, n_points, out, perm, s);
    case 16: return launch_one<16, T>(p, img, pts, H, W, n_points, out, perm, s);
#endif
  }
  return zk_fail(ZK_E_BADARG, "no point kernel for this n_max");
}

#if ZK_NMAX_GROUP == 0
// ---- counting sort of the points into buckets of `rh` frame rows x 256 columns -------------------------------------------
__device__ __forceinline__ int zk_point_bucket(const int32_t* pts, long long i, int H, int W, int rh_shift, int bx) {
  int x = pts[2 * i], y = pts[2 * i + 1];
  x = x < 0 ? 0 : (x >= W ? W - 1 : x);
  y = y < 0 ? 0 : (y >= H ? H - 1 : y);
  return (y >> rh_shift) * bx + (x >> 8);
}
// Consecutive lanes with the same bucket (points that arrive sorted: whole waves in one bucket) share ONE atomic: the first
// lane of a run adds the run's length, the others take their place behind it.  head lane / length of the run of lane `lane`.
__device__ __forceinline__ void zk_key_run(int key, bool live, int lane, int* head_lane, int* run_len) {
  const int prev = __shfl_up(key, 1);
  const bool head = live && (lane == 0 || key != prev || !__shfl_up((int)live, 1));
  const unsigned long long heads = __ballot(head) | (~__ballot(live) & ~0ull);  // dead lanes end a run as well
  const unsigned long long at_or_below = heads & (~0ull >> (63 - lane));
  const int hl = 63 - __clzll(at_or_below ? at_or_below : 1ull);
  const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1)) << (lane + 1);
  const unsigned long long above_head = hl == 63 ? 0ull : (heads >> (hl + 1)) << (hl + 1);
  (void)above;
  const int next = above_head ? __ffsll((long long)above_head) - 1 : 64;
  *head_lane = hl;
  *run_len = next - hl;
}
__global__ __launch_bounds__(256) void zk_points_hist_kernel(const int32_t* __restrict__ pts, long long n, int H, int W, int rh_shift,
                                                             int bx, int32_t* __restrict__ hist) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n;
  const int lane = threadIdx.x & 63;
  const int key = live ? zk_point_bucket(pts, i, H, W, rh_shift, bx) : -1;
  int hl, len;
  zk_key_run(key, live, lane, &hl, &len);
  if (live && hl == lane) atomicAdd(hist + key, len);
}
// exclusive scan of hist[0 .. nb) into cursor, one workgroup: a thread owns a run of consecutive buckets
__global__ __launch_bounds__(1024) void zk_points_scan_kernel(const int32_t* __restrict__ hist, int32_t* __restrict__ cursor, int nb) {
  __shared__ int part[1024];
  const int t = threadIdx.x, per = (nb + 1023) / 1024, b0 = t * per, b1 = b0 + per < nb ? b0 + per : nb;
  int sum = 0;
  for (int b = b0; b < b1; ++b) sum += hist[b];
  part[t] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele over the 1024 partial sums
    const int v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - sum;  // exclusive prefix of this thread's run
  for (int b = b0; b < b1; ++b) {
    cursor[b] = run;
    run += hist[b];
  }
}
__global__ __launch_bounds__(256) void zk_points_scatter_kernel(const int32_t* __restrict__ pts, long long n, int H, int W, int rh_shift,
                                                                int bx, int32_t* __restrict__ cursor, int32_t* __restrict__ perm) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n;
  const int lane = threadIdx.x & 63;
  const int key = live ? zk_point_bucket(pts, i, H, W, rh_shift, bx) : -1;
  int hl, len;
  zk_key_run(key, live, lane, &hl, &len);
  int base = 0;
  if (live && hl == lane) base = atomicAdd(cursor + key, len);
  base = __shfl(base, hl);
  if (live) perm[base + (lane - hl)] = (int32_t)i;
}

// fills the plan's index list with the points in bucket order; *perm_out = nullptr when the call is not worth it
int bucket_points(zk_plan* p, const int32_t* pts, int64_t H, int64_t W, int64_t n_points, const int32_t** perm_out, hipStream_t s) {
  *perm_out = nullptr;
  const bool off = getenv("ZK_POINTS_NO_BUCKET") != nullptr;
  if (off || n_points < 4096 || n_points > 0x7fffffffLL) return 0;  // (a few waves: nothing to gain; int32 indices)
  const int bx = (int)((W + 255) >> 8);
  int rh_shift = 0;
  while ((((H - 1) >> rh_shift) + 1) * (long long)bx > (1 << 20)) ++rh_shift;
  const int nb = (int)((((H - 1) >> rh_shift) + 1) * bx);
  const size_t need = ((size_t)2 * nb + (size_t)n_points) * sizeof(int32_t);
  int rc = zk_ensure(&p->d_points_tmp, &p->d_points_tmp_bytes, need);
  if (rc) return rc;
  int32_t* hist = (int32_t*)p->d_points_tmp;
  int32_t* cursor = hist + nb;
  int32_t* perm = cursor + nb;
  ZK_HIP(hipMemsetAsync(hist, 0, (size_t)nb * sizeof(int32_t), s));
  const unsigned blocks = (unsigned)((n_points + 255) / 256);
  hipLaunchKernelGGL(zk_points_hist_kernel, dim3(blocks), dim3(256), 0, s, pts, (long long)n_points, (int)H, (int)W, rh_shift, bx, hist);
  hipLaunchKernelGGL(zk_points_scan_kernel, dim3(1), dim3(1024), 0, s, hist, cursor, nb);
  hipLaunchKernelGGL(zk_points_scatter_kernel, dim3(blocks), dim3(256), 0, s, pts, (long long)n_points, (int)H, (int)W, rh_shift, bx,
                     cursor, perm);
  ZK_HIP(hipGetLastError());
  *perm_out = perm;
  return 0;
}
#endif

}  // namespace

#if ZK_NMAX_GROUP == 0
bool zk_sep_points_available(const zk_plan* p, int dtype) {
  (void)dtype;
  return p->sep && p->sep->n_rows > 0 && p->sep->kernel_nmax <= 16;
}
int zk_launch_sep_points_g1(zk_plan* p, const void* img, int dtype, int64_t H, int64_t W, const int32_t* pts,
                            int64_t n_points, double* out, const int32_t* perm, hipStream_t s);

int zk_launch_sep_points(zk_plan* p, const void* img, int dtype, int64_t H, int64_t W, const int32_t* pts, int64_t n_points,
                         double* out, hipStream_t s) {
  if (!zk_sep_points_available(p, dtype))
    return zk_fail(ZK_E_BADARG, "plan has no key-point kernel (needs the separable tables and n_max <= 16)");
  const int32_t* perm = nullptr;
  int rc = bucket_points(p, pts, H, W, n_points, &perm, s);
  if (rc) return rc;
  if (p->sep->kernel_nmax > 12) return zk_launch_sep_points_g1(p, img, dtype, H, W, pts, n_points, out, perm, s);
  if (dtype == ZK_F32) return launch_t<float>(p, img, pts, H, W, n_points, out, perm, s);
  return launch_t<double>(p, img, pts, H, W, n_points, out, perm, s);
}
#else
int ZK_GROUP_FN(zk_launch_sep_points)(zk_plan* p, const void* img, int dtype, int64_t H, int64_t W, const int32_t* pts,
                                      int64_t n_points, double* out, const int32_t* perm, hipStream_t s) {
  if (dtype == ZK_F32) return launch_t<float>(p, img, pts, H, W, n_points, out, perm, s);
  return launch_t<double>(p, img, pts, H, W, n_points, out, perm, s);
}
#endif
