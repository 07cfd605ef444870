// zk_sep_strip.hip -- dense-frame Zernike moments (reference _zps.py:159-193) for small bases (n_max <= 8):
// one lane owns TWO vertically adjacent output pixels and shares the column work between them.
//
// The dense kernel of zk_sep_frame.hip is bound by f64 instruction issue, so only fewer operations make
// it faster.  The disk rows are nested intervals: the part of window row r inside the disk is the
// quadrant columns cmin_r .. Q-1 and their mirrors.  The outputs (i, k) and (i+1, k) see frame row rho
// as window rows r and r-1, i.e. they need the x-direction sums
//     S_a(rho; c) = sum_{q = c}^{Q-1} [f(rho, q) +- f(rho, K-1-q)] P_a(x_q)          (+ for even a)
// of the SAME frame row at two different inner limits c = cmin_r, cmin_{r-1}.  Summing from the centre
// outwards yields both on the way: the accumulators pass through the narrower limit first and are used
// there (M_(a,b) += P_b(y) S_a for that output) before they run on to the wider one.  Pixel work per output
// halves; the price is the y mirror fold (rows r and K-1-r of one output come from different frame rows), so
// the row step costs N_poly FMAs per disk row instead of per row pair.  Net at (32, 8): ~4.1 k f64
// operations per output against ~6.3 k.  Measured per 2048^2 frame, float32 (ms, one-output kernel -> this one):
// (32, 8) 1.17 -> 0.95, (48, 8) 2.13 -> 1.58, (64, 8) 3.85 -> 2.36, (32, 6) 0.90 -> 0.65.  Two accumulator sets only fit two waves per SIMD up to n_max 8.
//
// 256-thread workgroup = 4 waves x 2 output rows x 64 columns; the zero-padded (K+7) x (K+63) tile is staged
// in LDS as float64, as in zk_sep_frame.hip.  Tables: the quadrant x table (d_xq), the full-width table of the
// stream kernel for the y direction (d_pfull: P_1 .. P_nmax of every row, P_0 = 1 implicit), cmin per row.
#include "zk_sep.h"

namespace {

#ifndef ZK_STRIP_GROUP
#define ZK_STRIP_GROUP 3  // column pairs per group of the sweep (2 and 3 measure alike, 4 is 4-6 % slower; one at a
                          // time -- a scalar-memory wait per column pair -- was 12 % slower).
                          // Round 2: the explicit wait / prefetch-next / compute pipeline of zk_sep_row_pair does NOT pay
                          // here: two table-row sets + the y row need more SGPRs than there are (79 spilled to VGPR
                          // lanes at one column per step, 141 at two): (32, 8) 0.95 -> 1.38 ms per 2048^2.
#endif

template <int NMAX, typename T>
__global__ __launch_bounds__(256, 2) void zk_frame_strip_kernel(
    const T* __restrict__ img, double* __restrict__ out, const int32_t* __restrict__ cmin_tab,
    const double* __restrict__ xq, const double* __restrict__ pfull, const double* __restrict__ tmat,
    const int32_t* __restrict__ colmap, int K, int H, int W, int row0, int n_rows, int tile_pitch, long long plane) {
  using S = zk_sep_set<NMAX>;
  constexpr int YROW = ZK_STREAM_ROW(NMAX);
  extern __shared__ __attribute__((aligned(16))) double tile[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ea = K - 1 - (K - 1) / 2;
  const int i0 = row0 + blockIdx.y * 8;
  const int k0 = blockIdx.x * 64;
  const int tile_elems = (K + 7) * tile_pitch;

  for (int e = tid; e < tile_elems; e += 256) {
    const int tr = e / tile_pitch;
    const int tc = e - tr * tile_pitch;
    const int ii = i0 - ea + tr;
    const int kk = k0 - ea + tc;
    double v = 0.0;
    if (ii >= 0 && ii < H && kk >= 0 && kk < W) v = (double)img[(long long)ii * W + kk];
    tile[e] = v;
  }
  __syncthreads();

  zk_sep_acc<NMAX> acc0, acc1;  // moments of output rows i0 + 2 wave and i0 + 2 wave + 1 (only M is used)
  acc0.clear_moments();
  acc1.clear_moments();
  const ZK_CONST int32_t* ctab = zk_const(cmin_tab);
  const ZK_CONST double* px = zk_const(xq);
  const ZK_CONST double* py = zk_const(pfull);
  const int Q = (K + 1) / 2;
  const double* __restrict__ mine = tile + (2 * wave) * tile_pitch + lane;  // window row 0 of output 0

  // frame row fr (relative to output 0's window) is window row fr of output 0 and fr - 1 of output 1.  The
  // inner limits shrink towards the centre row and grow again, so in the upper half of the window output 1
  // (one row higher up in its window) has the narrower row and is served first, in the lower half output 0:
  // two loops with a fixed order each, every accumulator set touched at one place per loop body.
  auto frame_row = [&](int fr, auto first_is_1) {
    constexpr bool F1 = decltype(first_is_1)::value;
    const int c0 = fr < K ? ctab[fr] : Q;  // inner limits (Q: the row has no disk pixel / does not exist)
    const int c1 = fr > 0 ? ctab[fr - 1] : Q;
    const int ca = F1 ? c1 : c0, cb = F1 ? c0 : c1;  // ca >= cb
    if (cb >= Q) return;
    double X[S::NA];
#pragma unroll
    for (int i = 0; i < S::NA; ++i) X[i] = 0.0;
    const double* __restrict__ row = mine + fr * tile_pitch;
    auto fma_pair = [&](double a, double b, const double (&xr)[S::NA]) {  // one column pair (q, K-1-q)
      const double s = a + b, d = a - b;
#pragma unroll
      for (int i = 0; i < S::NA; ++i) X[i] = __builtin_fma((i & 1) ? d : s, xr[i], X[i]);
    };
    // quadrant columns q_from down to q_to, ZK_STRIP_GROUP at a time: the table rows of a group are requested
    // together, ahead of its arithmetic (one scalar-memory wait per group), and two running pointers make the
    // LDS addresses of a group immediates
    auto sweep = [&](int q_from, int q_to) {
      int q = q_from;
      const double* pa = row + q;          // walks down
      const double* pb = row + K - 1 - q;  // walks up
      const ZK_CONST double* pr = px + q * ZK_SEP_ROW;
      constexpr int G = ZK_STRIP_GROUP;
      for (; q - (G - 1) >= q_to; q -= G) {
        double xr[G][S::NA];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int i = 0; i < S::NA; ++i) xr[g][i] = pr[-g * ZK_SEP_ROW + i];
#pragma unroll
        for (int g = 0; g < G; ++g) fma_pair(pa[-g], pb[g], xr[g]);
        pa -= G;
        pb += G;
        pr -= G * ZK_SEP_ROW;
      }
      for (; q >= q_to; --q) {
        double xr[S::NA];
#pragma unroll
        for (int i = 0; i < S::NA; ++i) xr[i] = pr[i];
        fma_pair(pa[0], pb[0], xr);
        --pa;
        ++pb;
        pr -= ZK_SEP_ROW;
      }
    };
    const int stop_a = ca < Q ? ca : cb;
    sweep(Q - 1, stop_a);
    if (ca < Q) {
      if constexpr (F1) acc1.stream_accumulate(X, py + (fr - 1) * YROW);
      else acc0.stream_accumulate(X, py + fr * YROW);
      sweep(ca - 1, cb);
    }
    if constexpr (F1) acc0.stream_accumulate(X, py + fr * YROW);
    else acc1.stream_accumulate(X, py + (fr - 1) * YROW);
  };
  const int half = Q;  // rows 0 .. Q-1: limits non-increasing; rows Q-1 .. K-1: non-decreasing
  for (int fr = 0; fr < half; ++fr) frame_row(fr, std::true_type{});
  for (int fr = half; fr <= K; ++fr) frame_row(fr, std::false_type{});

  const int ok = k0 + lane;
  const ZK_CONST int32_t* cmap = zk_const(colmap);
  const ZK_CONST double* tb = zk_const(tmat);
  {
    const int oi = i0 + 2 * wave;
    const bool live = oi < row0 + n_rows && ok < W;
    double* __restrict__ dst = out + (long long)(oi - row0) * W + ok;
    acc0.transform(tb, [&](auto slot, double z) {
      const int col = cmap[slot];
      if (live && col >= 0) dst[col * plane] = z;
    });
  }
  {
    const int oi = i0 + 2 * wave + 1;
    const bool live = oi < row0 + n_rows && ok < W;
    double* __restrict__ dst = out + (long long)(oi - row0) * W + ok;
    acc1.transform(tb, [&](auto slot, double z) {
      const int col = cmap[slot];
      if (live && col >= 0) dst[col * plane] = z;
    });
  }
}

template <int NMAX, typename T>
int launch_one(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out,
               hipStream_t s) {
  const zk_sep_tables* t = p->sep;
  const size_t lds = (size_t)(p->size + 7) * t->tile_pitch * sizeof(double);
  auto kern = zk_frame_strip_kernel<NMAX, T>;
  if (lds > 64 * 1024)
    ZK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long plane = zk_out_plane(p, n_rows, W);
  return zk_for_row_bands(row0, n_rows, W, 8, [&](int64_t r0, int64_t nr, long long off) {
    dim3 grid((unsigned)((W + 63) / 64), (unsigned)((nr + 7) / 8));
    int rc = zk_prof_begin(p, s);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const T*)in, out + off, t->d_cmin, t->d_xq, t->d_pfull, t->d_T,
                       t->d_colmap, p->size, (int)H, (int)W, (int)r0, (int)nr, t->tile_pitch, plane);
    ZK_HIP(hipGetLastError());
    return zk_prof_end(p, s);
  });
}

template <typename T>
int launch_t(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out,
             hipStream_t s) {
  switch (p->sep->kernel_nmax) {
    case 4: return launch_one<4, T>(p, in, H, W, row0, n_rows, out, s);
    case 6: return launch_one<6, T>(p, in, H, W, row0, n_rows, out, s);
    case 8: return launch_one<8, T>(p, in, H, W, row0, n_rows, out, s);
  }
  return zk_fail(ZK_E_BADARG, "no strip frame kernel for this n_max");
}

}  // namespace

bool zk_sep_strip_available(const zk_plan* p, int dtype) {
  (void)dtype;
  if (!zk_sep_frame_available(p, dtype) || p->sep->kernel_nmax > 8 || !p->sep->d_pfull || !p->sep->d_cmin) return false;
  // two workgroups per CU must fit (two waves per SIMD is what the kernel is compiled for): K <= 65.  Beyond,
  // the one-output kernel is ahead again (72 px: 4.7 vs 5.0 ms per 2048^2).
  return (size_t)(p->size + 7) * p->sep->tile_pitch * sizeof(double) <= 80 * 1024;
}

int zk_launch_sep_strip(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0, int64_t n_rows,
                        double* out, hipStream_t s) {
  if (dtype == ZK_F32) return launch_t<float>(p, in, H, W, row0, n_rows, out, s);
  return launch_t<double>(p, in, H, W, row0, n_rows, out, s);
}
