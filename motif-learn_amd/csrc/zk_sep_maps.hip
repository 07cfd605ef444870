// zk_sep_maps.hip -- fused dense pipeline: frame -> per-pixel symmetry maps, without materialising
// the (N_poly, H, W) moments in HBM (BASELINE configs[4]).
//
// What it computes, per output pixel, from the moments Z of the pixel's window (zk_sep_frame.hip):
//   abs   [(n,|m|)] = |Z_{n,m} + i Z_{n,-m}|                      np.abs(zm.to_complex().data)   _zmoments.py:300-316
//   rot   [f]       = sum_j w_{f,|m_j|} Z_j^2 / ||Z_sel||^2        zm.rot_maps(folds, p=2)        _zmoments.py:420-462
//   mirror          = max_theta sum_k Re[(Zhat^c_k)^2 e^{-i m_k theta}]   zm.mirror_map(theta)    _zmoments.py:464-493
// with "sel" = moments whose |m| is not in m_unselect (default (0,1)) and Zhat = Z_sel / ||Z_sel||_2
// (p = 2; p = None skips the normalisation).  Since ||.|| is a positive per-pixel scalar it is applied
// once at the end: rot = (sum w Z^2) / ||Z||^2,  mirror = max_theta(sum_m C_m cos m theta + S_m sin m theta) / ||Z||^2
// with C_m = sum_n (A^2 - B^2), S_m = sum_n 2AB, A = Z_{n,m}, B = Z_{n,-m}.
//
// Register discipline: the rows of the T product (zk_sep.h) are evaluated pair by pair -- Z_{n,+m} and
// Z_{n,-m} back to back -- and folded into 3 (n_max+1) running sums, so no moment array ever exists.
// Algorithmic HBM bytes: s_in + 8 (n_folds + N_c + 1) per pixel instead of s_in + 8 N_poly written and
// read again by a separate map pass.
#include <math.h>

#include <stdlib.h>

#include "zk_sep.h"

// Build groups: the kernel instances are spread over several translation units (Makefile) so that they
// compile in parallel: group 0 = n_max kernels 4..12 (and every non-template entry point), 1 = 14 / 16,
// 2 = 20 / 24 (class-pass kernels), 3 = 28 .. 40 (moments from the matrix-core plain sum, zk_direct_patches.hip).
// Group 0's launcher forwards to the others.
#ifndef ZK_NMAX_GROUP
#define ZK_NMAX_GROUP 0
#endif
#if ZK_NMAX_GROUP == 0
#define ZK_GROUP_FN(name) name
#elif ZK_NMAX_GROUP == 1
#define ZK_GROUP_FN(name) name##_g1
#elif ZK_NMAX_GROUP == 2
#define ZK_GROUP_FN(name) name##_g2
#else
#define ZK_GROUP_FN(name) name##_g3
#endif

#define ZK_MAX_FOLDS 8
#define ZK_MAPS_WROW 48  // doubles per fold-weight row of the device table (|m| 0 .. 40 used)
#define ZK_TRIG_CACHE 8  // device tables kept per plan, one per distinct (folds, m_unselect, theta) option set
// waves per SIMD the register allocator is asked to fit (launch bound)
#ifndef ZK_MAPS_2W
#define ZK_MAPS_2W 12  // (12 at two waves: 4.23 -> 3.86 ms per 2048^2 with all outputs)
#endif
#ifndef ZK_MAPS_WAVES
#define ZK_MAPS_WAVES(NMAX) ((NMAX) <= 8 ? 3 : (NMAX) <= ZK_MAPS_2W ? 2 : 1)
#endif

struct zk_maps_params {
  int n_folds;
  unsigned long long unselect_mask;   // bit am set: moments with |m| == am are dropped (bit 0 always set)
  int normalize;       // 1: p = 2, 0: p = None
  int n_theta;
  int plan_nmax;       // moments with n > plan_nmax are padding of the kernel set
  int theta_sym;       // theta is the uniform grid 2 pi k / n_theta with 1: n_theta % 4 == 0, 2: n_theta % 8 == 0 (see below)
};

namespace {

template <typename F, int... Is>
__device__ __forceinline__ void zk_for_each_int(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}

// The symmetry-map tail shared by the fused kernel (moments from the T product in registers) and the
// planes kernel (moments from a (N_poly, rows, W) matrix in memory): `get(n, |m|, A, B)` delivers the
// complex moment A + iB = Z_{n,+m} + i Z_{n,-m} of this lane's pixel.  `pix` / `plane` address the outputs.
template <int NMAX, bool BY_ORDER = false, typename GET>
__device__ __forceinline__ void zk_maps_tail(GET&& get, const zk_maps_params& prm, const double* __restrict__ trig,
                                             bool live, long long plane, long long pix, double* __restrict__ rot_out,
                                             double* __restrict__ abs_out, double* __restrict__ mirror_out,
                                             long long pix_rot = -1, long long pix_mir = -1) {
  using Z = zk_set<NMAX>;
  // (plane, pix) address the |Z| output; the two others default to the same pixel offset (planes layout) and differ in
  // the rows layout, where a pixel's outputs are rows of (N, n_folds), (N, N_c) and (N)
  if (pix_rot < 0) pix_rot = pix;
  if (pix_mir < 0) pix_mir = pix;
  // per-|m| sums over n of the complex moments A + iB:  E = A^2 + B^2,  C = A^2 - B^2,  S = 2AB
  double Em[NMAX + 1], Cm[NMAX + 1], Sm[NMAX + 1];
#pragma unroll
  for (int m = 0; m <= NMAX; ++m) Em[m] = Cm[m] = Sm[m] = 0.0;

  // one complex moment (n, am) with real part A = Z_{n,+am} and imaginary part B = Z_{n,-am}
  auto combine = [&](auto nn, auto amm, double A, double B) {
    constexpr int n = decltype(nn)::value, am = decltype(amm)::value;
    const double a2 = A * A, b2 = B * B;
    if (abs_out != nullptr && live && n <= prm.plan_nmax)
      abs_out[Z::complex_index(n, am) * plane + pix] = __builtin_sqrt(a2 + b2);
    if constexpr (am > 0) {  // m = 0 is never selected (the host insists on it in m_unselect): its sums are not kept at all
      Em[am] += a2 + b2;
      Cm[am] += a2 - b2;
      Sm[am] = __builtin_fma(2.0 * A, B, Sm[am]);
    }
  };

  if constexpr (BY_ORDER) {
    // Moments that come from memory (planes kernel, n_max 17-24).  Written out per (n, |m|) at compile time -- 231
    // combine bodies at n_max 20, each with its double-precision square-root expansion -- the kernel needed 512 registers
    // plus 640 spilled (2.2 KB of scratch per lane) and ran at 0.3 TB/s.  Here the order n is a RUN-TIME loop and only
    // |m| is unrolled: an order's 2 (NMAX + 1) loads are issued together, unconditionally (slots that do not exist for
    // this n re-read plane 0 and are ignored), then folded under wave-uniform tests.
    for (int n = 0; n <= prm.plan_nmax; ++n) {
      double A[NMAX + 1], B[NMAX + 1];
#pragma unroll
      for (int am = 0; am <= NMAX; ++am) {
        const bool valid = am <= n && ((n - am) & 1) == 0;
        get(n, am, valid, A[am], B[am]);
      }
#pragma unroll
      for (int am = 0; am <= NMAX; ++am) {
        if (am <= n && ((n - am) & 1) == 0) {  // wave-uniform
          const double a2 = A[am] * A[am], b2 = B[am] * B[am];
          if (abs_out != nullptr && live) abs_out[Z::complex_index(n, am) * plane + pix] = __builtin_sqrt(a2 + b2);
          if (am > 0) {
            Em[am] += a2 + b2;
            Cm[am] += a2 - b2;
            Sm[am] = __builtin_fma(2.0 * A[am], B[am], Sm[am]);
          }
        }
      }
    }
  } else {
    // +m and -m of one (n, |m|) are produced back to back (they live in partner parity classes of the T
    // product) and folded into the running sums at once, so no moment outlives its own combine().
    // Even |m| first, then odd |m|: Z_{n,+m} and Z_{n,-m} of an even m come from the classes EE and OO of the T product, of an
    // odd m from OE and EO -- after the first pass half of the 3 (n_max + 1)(n_max + 2) / 2 ... sums M of the pixel loop are dead
    // and the second pass has their registers (the n_max 10 instance, capped at 256 registers for two waves per SIMD, spilled
    // 404 bytes per lane with the (n, |m|) pairs in index order).  Inside a pass the orders n run DOWNWARDS: Z_n only uses the
    // sums M_(a,b) with a + b <= n, so those with a + b = n die with order n while the per-|m| sums grow.
    zk_for_each_int(
        [&](auto k) {
          constexpr int kk = Z::NC - 1 - decltype(k)::value;  // highest order first: see below
          constexpr int n = Z::complex_n(kk), am = Z::complex_m(kk);
          if constexpr ((am & 1) == 0) {
            double A, B = 0.0;
            get(std::integral_constant<int, n>{}, std::integral_constant<int, am>{}, A, B);
            combine(std::integral_constant<int, n>{}, std::integral_constant<int, am>{}, A, B);
          }
        },
        std::make_integer_sequence<int, Z::NC>{});
    zk_for_each_int(
        [&](auto k) {
          constexpr int kk = Z::NC - 1 - decltype(k)::value;
          constexpr int n = Z::complex_n(kk), am = Z::complex_m(kk);
          if constexpr ((am & 1) == 1) {
            double A, B = 0.0;
            get(std::integral_constant<int, n>{}, std::integral_constant<int, am>{}, A, B);
            combine(std::integral_constant<int, n>{}, std::integral_constant<int, am>{}, A, B);
          }
        },
        std::make_integer_sequence<int, Z::NC>{});
  }

  // drop the unselected |m| (wave-uniform mask), then the per-pixel scalar ||Z_sel||^2
  double norm2 = 0.0;
#pragma unroll
  for (int m = 0; m <= NMAX; ++m) {
    const double keep_m = ((prm.unselect_mask >> m) & 1) ? 0.0 : 1.0;
    Em[m] *= keep_m;
    Cm[m] *= keep_m;
    Sm[m] *= keep_m;
    norm2 += Em[m];
  }
  const double inv = prm.normalize ? 1.0 / norm2 : 1.0;  // 0/0 -> NaN exactly where NumPy gives NaN
  // table layout: [ZK_MAX_FOLDS][ZK_MAPS_WROW] fold weights by |m|, then per angle one compact row
  // [cos(1 t) .. cos(NMAX t) | sin(1 t) .. sin(NMAX t)] -- 2 NMAX doubles, so the quarter grid of the default
  // 360 angles (91 rows, 14.6 KB at n_max 10) stays resident in the 16-KiB scalar cache
  const ZK_CONST double* wtab = zk_const(trig);
  if (rot_out != nullptr) {
    for (int f = 0; f < prm.n_folds; ++f) {
      double r = 0.0;
#pragma unroll
      for (int m = 0; m <= NMAX; ++m) r = __builtin_fma(wtab[f * ZK_MAPS_WROW + m], Em[m], r);
      if (live) rot_out[f * plane + pix_rot] = r * inv;
    }
  }
  if (mirror_out != nullptr) {
    const ZK_CONST double* cs = wtab + ZK_MAX_FOLDS * ZK_MAPS_WROW;
    double best = -__builtin_inf();
    auto take = [&](double s) { best = s > best || s != s ? s : best; };  // NaN propagates like numpy.max
    if (prm.theta_sym == 2) {
      // Uniform full-circle grid with n_theta % 8 == 0.  theta, pi - theta, pi + theta, 2 pi - theta AND the same four
      // reflections of pi/2 - theta are all grid points, and with c_m = cos m theta, s_m = sin m theta
      //   m = 0 (mod 4): cos m(pi/2 - t) =  c_m, sin =  -s_m        m = 1: cos =  s_m, sin =  c_m
      //   m = 2        : cos              = -c_m, sin =   s_m        m = 3: cos = -s_m, sin = -c_m
      // so ONE table row (angle index 0 .. n_theta/8) yields eight scores: the even-m products are shared outright,
      // the odd m need two extra products each.  46 rows of 2 NMAX doubles instead of 91 at 360 angles: half the
      // scalar-cache footprint (7.4 KB at n_max 10 -- the 14.6-KB quarter table, next to the pixel loop's own
      // tables, missed the 16-KB cache: ~800 cycles per row and wave) and 3/4 of the FMAs.
      // The table holds only (cos theta, sin theta) per row; cos m theta / sin m theta come from the three-term
      // recurrence x_{m+1} = 2 cos(theta) x_m - x_{m-1} on the VALU (lane-uniform values, 2 FMAs per m; error
      // ~m^2 ulp, 1e-13 at m = 24).  A full [angle][2 NMAX] table kept missing the scalar cache (~500 cycles
      // per half row and wave even at 7.4 KB, next to the pixel loop's own tables); this one is 0.7 KB and one
      // scalar load feeds four rows.  The host pads it to a multiple of four rows by repeating the last row.
      // (Round 3 tried the third place a table can live: [angle][m] pairs (cos m theta, sin m theta) in LDS behind the tile, one
      //  broadcast ds_read_b128 per m instead of the two recurrence FMAs -- all outputs per 4096^2 at n_max 10: 9.17 -> 9.83 ms,
      //  n_max 12: 14.8 -> 15.9: a wave-wide 16-byte LDS read costs more than two lane-uniform FMAs.  Not kept.)
      const ZK_CONST double* tr = cs;
      double best8 = -__builtin_inf();
      const int rows4 = (prm.n_theta / 8 + 4) & ~3;
      for (int i0 = 0; i0 < rows4; i0 += 4) {
        double cs4[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) cs4[t] = tr[2 * i0 + t];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // sums over m = r (mod 4): C c (k*), S s (l*), and for odd m the cross products C s (x*), S c (y*)
          double k0 = 0.0, k1 = 0.0, k2 = 0.0, k3 = 0.0, l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0;
          double x1 = 0.0, x3 = 0.0, y1 = 0.0, y3 = 0.0;
          const double c1 = cs4[2 * j], s1 = cs4[2 * j + 1], two_c1 = c1 + c1;
          double cp_ = 1.0, sp_ = 0.0, c = c1, sn = s1;  // (cos, sin) of (m - 1) theta and m theta
#pragma unroll
          for (int m = 1; m <= NMAX; ++m) {  // m = 0 is always unselected (C_0 = S_0 = 0)
            if ((m & 3) == 0) k0 = __builtin_fma(Cm[m], c, k0), l0 = __builtin_fma(Sm[m], sn, l0);
            if ((m & 3) == 1)
              k1 = __builtin_fma(Cm[m], c, k1), l1 = __builtin_fma(Sm[m], sn, l1), x1 = __builtin_fma(Cm[m], sn, x1),
              y1 = __builtin_fma(Sm[m], c, y1);
            if ((m & 3) == 2) k2 = __builtin_fma(Cm[m], c, k2), l2 = __builtin_fma(Sm[m], sn, l2);
            if ((m & 3) == 3)
              k3 = __builtin_fma(Cm[m], c, k3), l3 = __builtin_fma(Sm[m], sn, l3), x3 = __builtin_fma(Cm[m], sn, x3),
              y3 = __builtin_fma(Sm[m], c, y3);
            if (m < NMAX) {
              const double cn_ = __builtin_fma(two_c1, c, -cp_), sn_ = __builtin_fma(two_c1, sn, -sp_);
              cp_ = c;
              sp_ = sn;
              c = cn_;
              sn = sn_;
            }
          }
          // the four reflections of an angle t: scores cp +- sp (t, 2 pi - t) and cn -+ sm (pi - t, pi + t); the larger
          // of each pair is cp + |sp| resp. cn + |sm|, bit for bit
          auto four = [&](double ce, double co, double se, double so) {
            const double cp = ce + co, cn = ce - co, sp = se + so, sm = se - so;
            best8 = __builtin_fmax(best8, cp + __builtin_fabs(sp));
            best8 = __builtin_fmax(best8, cn + __builtin_fabs(sm));
          };
          four(k0 + k2, k1 + k3, l0 + l2, l1 + l3);  // t = theta
          four(k0 - k2, x1 - x3, l2 - l0, y1 - y3);  // t = pi/2 - theta
        }
      }
      // v_max_f64 ignores NaN operands where numpy.max propagates them: every score is a combination of ALL the
      // selected C_m / S_m with non-zero coefficients, so the result is NaN exactly when one of them is (non-finite
      // moments -- inf pixels, or 0/0 from the normalisation -- give NaN here, as they do in NumPy whenever a NaN
      // score exists).
      double poison = 0.0;
#pragma unroll
      for (int m = 1; m <= NMAX; ++m) poison = __builtin_fma(Cm[m], 0.0, __builtin_fma(Sm[m], 0.0, poison));
      best = poison != poison ? poison : best8;
    } else if (prm.theta_sym) {
      // Uniform full-circle grid: theta, pi - theta, pi + theta and 2 pi - theta are all grid points and
      //   cos m(pi -+ t) = (-1)^m cos mt,  sin m(pi - t) = -(-1)^m sin mt,  sin m(pi + t) = (-1)^m sin mt,
      // so the first quarter of the grid (n_theta/4 + 1 rows) yields all n_theta scores.
#pragma unroll 2
      for (int i = 0; i <= prm.n_theta / 4; ++i) {
        double ce = 0.0, co = 0.0, se = 0.0, so = 0.0;  // C / S parts over even / odd m
#pragma unroll
        for (int m = 1; m <= NMAX; ++m) {  // m = 0 is always unselected (C_0 = S_0 = 0)
          const double c = cs[i * (2 * NMAX) + m - 1], sn = cs[i * (2 * NMAX) + NMAX + m - 1];
          if (m & 1) {
            co = __builtin_fma(Cm[m], c, co);
            so = __builtin_fma(Sm[m], sn, so);
          } else {
            ce = __builtin_fma(Cm[m], c, ce);
            se = __builtin_fma(Sm[m], sn, se);
          }
        }
        const double cp = ce + co, cn = ce - co, sp = se + so, sm = se - so;
        take(cp + sp);  // theta
        take(cp - sp);  // 2 pi - theta
        take(cn - sm);  // pi - theta
        take(cn + sm);  // pi + theta
      }
    } else {
#pragma unroll 2
      for (int i = 0; i < prm.n_theta; ++i) {
        double sc = 0.0;
#pragma unroll
        for (int m = 1; m <= NMAX; ++m) {
          sc = __builtin_fma(Cm[m], cs[i * (2 * NMAX) + m - 1], sc);
          sc = __builtin_fma(Sm[m], cs[i * (2 * NMAX) + NMAX + m - 1], sc);
        }
        take(sc);
      }
    }
    if (live) mirror_out[pix_mir] = best * inv;
  }
}

template <int NMAX, typename T>
__global__ __launch_bounds__(256, ZK_MAPS_WAVES(NMAX)) void zk_frame_maps_kernel(
    const T* __restrict__ img, const zk_sep_row* __restrict__ rows, const double* __restrict__ xq,
    const double* __restrict__ tmat, const double* __restrict__ trig, double* __restrict__ rot_out,
    double* __restrict__ abs_out, double* __restrict__ mirror_out, zk_maps_params prm, int n_tab_rows, int K, int H,
    int W, int row0, int n_rows, int tile_pitch, long long plane) {
  using Z = zk_set<NMAX>;
  extern __shared__ __attribute__((aligned(16))) double tile[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int ea = K - 1 - (K - 1) / 2;
  const int i0 = row0 + blockIdx.y * 4;
  const int k0 = blockIdx.x * 64;
  const int tile_elems = (K + 3) * tile_pitch;

  zk_stage_tile(tile, img, H, W, i0 - ea, k0 - ea, K + 3, tile_pitch);
  (void)tile_elems;
  __syncthreads();

  zk_sep_acc<NMAX> acc;
  acc.clear_all();
  const double* __restrict__ mine = tile + wave * tile_pitch + lane;
  const ZK_CONST int32_t* rtab = zk_const((const int32_t*)rows);
  const ZK_CONST double* px = zk_const(xq);
  const int Q = (K + 1) / 2;
  for (int ri = 0; ri < n_tab_rows; ++ri) {
    const int r = rtab[2 * ri], cmin = rtab[2 * ri + 1];
    const double* __restrict__ top = mine + r * tile_pitch;
    const double* __restrict__ bot = mine + (K - 1 - r) * tile_pitch;
    zk_sep_row_pair<NMAX>(acc, top, bot, cmin, Q, K, px);
    acc.row_end(px + r * ZK_SEP_ROW);
  }

  // ---- fused tail ---------------------------------------------------------------------------------
  const int oi = i0 + wave;
  const int ok = k0 + lane;
  const bool live = oi < row0 + n_rows && ok < W;
  const long long pix = (long long)(oi - row0) * W + ok;

  const ZK_CONST double* tb = zk_const(tmat);
  zk_maps_tail<NMAX>(
      [&](auto nn, auto amm, double& A, double& B) {
        // +m and -m of one (n, |m|) are produced back to back (they live in partner parity classes of the
        // T product) and folded into the running sums at once, so no moment outlives its own combine()
        constexpr int n = decltype(nn)::value, am = decltype(amm)::value;
        A = acc.template moment<Z::slot_of(n, am)>(tb);
        if constexpr (am > 0) B = acc.template moment<Z::slot_of(n, -am)>(tb);
      },
      prm, trig, live, plane, pix, rot_out, abs_out, mirror_out);
}

// Planes form (plans whose moments come out of several kernel launches, n_max > 16): one lane per pixel,
// moments read from the (N_poly, mom_rows, W) matrix the dense kernels just wrote (plane j of (n, m) is
// the reference index j = (n (n + 2) + m) / 2), outputs written at row offset `out_row0` of (.., out_rows, W).
template <int NMAX>
__global__ __launch_bounds__(256) void zk_maps_planes_kernel(const double* __restrict__ mom, const double* __restrict__ trig,
                                                             double* __restrict__ rot_out, double* __restrict__ abs_out,
                                                             double* __restrict__ mirror_out, zk_maps_params prm,
                                                             int mom_rows, int W, int out_row0, long long plane) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long mplane = (long long)mom_rows * W;
  const bool live = t < mplane;
  const long long mp = live ? t : 0;
  const long long pix = mp + (long long)out_row0 * W;
  zk_maps_tail<NMAX, true>(
      [&](int n, int am, bool valid, double& A, double& B) {
        // plane j of (n, m) is the reference index j = (n (n + 2) + m) / 2; an absent slot reads plane 0
        const long long ja = valid ? (n * (n + 2) + am) / 2 : 0, jb = valid ? (n * (n + 2) - am) / 2 : 0;
        A = mom[ja * mplane + mp];
        B = am > 0 ? mom[jb * mplane + mp] : 0.0;
      },
      prm, trig, live, plane, pix, rot_out, abs_out, mirror_out);
}

// Rows form: the tail of a BATCH of moment vectors -- zmoments.to_complex / rot_maps / mirror_map on rank-2 data
// (reference _zmoments.py:300-316, 420-493; e.g. the moments at key points), one lane per row of the (N, n_poly)
// row-major matrix, outputs as rows of (N, n_folds), (N, N_c) and (N).  4 068 289 x 45: all outputs 1.87 ms, rot only 0.74 ms
// (tools/time_rows_maps.py).  Tried: one wave per 64-row tile with both transpositions in LDS (coalesced copy in, |Z| / rot
// assembled in LDS and stored as contiguous runs) -- 38 KiB of LDS per wave leaves four waves per CU and it ran SLOWER (2.12 /
// 1.45 ms): the strided per-lane accesses here are served by the caches, and the tail's arithmetic wants the occupancy.
template <int NMAX>
__global__ __launch_bounds__(256) void zk_maps_rows_kernel(const double* __restrict__ mom, const double* __restrict__ trig,
                                                           double* __restrict__ rot_out, double* __restrict__ abs_out,
                                                           double* __restrict__ mirror_out, zk_maps_params prm, long long n_rows,
                                                           int n_poly, int n_complex) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool live = t < n_rows;
  const double* __restrict__ row = mom + (live ? t : 0) * n_poly;
  zk_maps_tail<NMAX, true>(
      [&](int n, int am, bool valid, double& A, double& B) {
        const int ja = valid ? (n * (n + 2) + am) / 2 : 0, jb = valid ? (n * (n + 2) - am) / 2 : 0;
        A = row[ja];
        B = am > 0 ? row[jb] : 0.0;
      },
      prm, trig, live, 1, t * n_complex, rot_out, abs_out, mirror_out, t * prm.n_folds, t);
}

template <int NMAX>
int launch_rows_one(zk_plan* p, const double* mom, int64_t n_rows, const zk_maps_params& prm, const double* d_trig, double* rot,
                    double* ab, double* mirror, hipStream_t s) {
  int rc = zk_prof_begin(p, s);
  if (rc) return rc;
  hipLaunchKernelGGL(zk_maps_rows_kernel<NMAX>, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, s, mom, d_trig, rot, ab, mirror,
                     prm, (long long)n_rows, p->n_poly, zk_complex_count(prm.plan_nmax));
  ZK_HIP(hipGetLastError());
  return zk_prof_end(p, s);
}

template <int NMAX, typename T>
int launch_one(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows,
               const zk_maps_params& prm, const double* d_trig, double* rot, double* ab, double* mirror,
               hipStream_t s) {
  const zk_sep_tables* t = p->sep;
  const size_t lds = (size_t)(p->size + 3) * t->tile_pitch * sizeof(double);
  auto kern = zk_frame_maps_kernel<NMAX, T>;
  if (lds > 64 * 1024)
    ZK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long plane = zk_out_plane(p, n_rows, W);
  return zk_for_row_bands(row0, n_rows, W, 4, [&](int64_t r0, int64_t nr, long long off) {
    dim3 grid((unsigned)((W + 63) / 64), (unsigned)((nr + 3) / 4));
    int rc = zk_prof_begin(p, s);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const T*)in, t->d_rows, t->d_xq, t->d_T, d_trig,
                       rot ? rot + off : nullptr, ab ? ab + off : nullptr, mirror ? mirror + off : nullptr, prm, t->n_rows,
                       p->size, (int)H, (int)W, (int)r0, (int)nr, t->tile_pitch, plane);
    ZK_HIP(hipGetLastError());
    return zk_prof_end(p, s);
  });
}

template <typename T>
int launch_t(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows,
             const zk_maps_params& prm, const double* d_trig, double* rot, double* ab, double* mirror, hipStream_t s) {
  switch (p->sep->kernel_nmax) {
#if ZK_NMAX_GROUP == 0
    case 4: return launch_one<4, T>(p, in, H, W, row0, n_rows, prm, d_trig, rot, ab, mirror, s);
    case 6: return launch_one<6, T>(p, in, H, W, row0, n_rows, prm, d_trig, rot, ab, mirror, s);
    case 8: return launch_one<8, T>(p, in, H, W, row0, n_rows, prm, d_trig, rot, ab, mirror, s);
    case 10: return launch_one<10, T>(p, in, H, W, row0, n_rows, prm, d_trig, rot, ab, mirror, s);
    case 12: return launch_one<12, T>(p, in, H, W, row0, n_rows, prm, d_trig, rot, ab, mirror, s);
#endif
#if ZK_NMAX_GROUP == 1
    case 14: return launch_one<14, T>(p, in, H, W, row0, n_rows, prm, d_trig, rot, ab, mirror, s);
    case 16: return launch_one<16, T>(p, in, H, W, row0, n_rows, prm, d_trig, rot, ab, mirror, s);
#endif
  }
  return zk_fail(ZK_E_BADARG, "no fused maps kernel for this n_max");
}

}  // namespace

// the kernel launch proper, per build group (the host-side preparation below lives in group 0)
int ZK_GROUP_FN(zk_maps_dispatch)(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0, int64_t n_rows,
                                  const zk_maps_params& prm, const double* d_trig, double* rot, double* ab,
                                  double* mirror, hipStream_t s) {
  if (dtype == ZK_F32) return launch_t<float>(p, in, H, W, row0, n_rows, prm, d_trig, rot, ab, mirror, s);
  return launch_t<double>(p, in, H, W, row0, n_rows, prm, d_trig, rot, ab, mirror, s);
}

int ZK_GROUP_FN(zk_maps_rows_dispatch)(zk_plan* p, const double* mom, int64_t n_rows, const zk_maps_params& prm, const double* d_trig,
                                       double* rot, double* ab, double* mirror, hipStream_t s) {
  switch (p->sep->kernel_nmax) {
#if ZK_NMAX_GROUP == 0
    case 4: return launch_rows_one<4>(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
    case 6: return launch_rows_one<6>(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
    case 8: return launch_rows_one<8>(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
    case 10: return launch_rows_one<10>(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
    case 12: return launch_rows_one<12>(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
#endif
#if ZK_NMAX_GROUP == 1
    case 14: return launch_rows_one<14>(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
    case 16: return launch_rows_one<16>(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
#endif
#if ZK_NMAX_GROUP == 2
    case 20: return launch_rows_one<20>(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
    case 24: return launch_rows_one<24>(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
#endif
  }
  return zk_fail(ZK_E_BADARG, "no symmetry-map rows kernel for this n_max");
}

#if ZK_NMAX_GROUP >= 2
// Rows of a band whose moments come from zk_frame_direct_kernel (a workgroup = 8 output rows x 64 columns, one workgroup per CU at
// a time): of the multiples of 8 between half the scratch limit and the limit, the one whose grid fills whole rounds of the chip's
// CUs best -- 88 rows of a 2048-wide frame are 352 workgroups = 1.4 rounds on 256 CUs (the time of two), 64 rows are one round.
static int64_t zk_direct_band_rows(const zk_plan* p, int64_t W, int64_t limit, int64_t n_rows) {
  if (limit >= n_rows) return n_rows;
  const int64_t cb = (W + 63) / 64, top = limit / 8 * 8;
  if (top <= 8) return 8;
  int64_t best = top;
  double best_fill = 0.0;
  for (int64_t r = top; r >= 8 && 2 * r > top; r -= 8) {
    const int64_t blocks = r / 8 * cb, rounds = (blocks + p->n_cu - 1) / p->n_cu;
    const double fill = (double)blocks / (double)(rounds * p->n_cu);
    if (fill > best_fill + 1e-9) best_fill = fill, best = r;
  }
  return best;
}
#endif

#if ZK_NMAX_GROUP == 2
// n_max 17-20: the moments of a row band come from the class-pass dense kernel into the plan's scratch
// matrix (<= 1 GiB at a time), the planes kernel turns them into the maps of those rows.
int zk_maps_planes_g2(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0, int64_t n_rows,
                      const zk_maps_params& prm, const double* d_trig, double* rot, double* ab, double* mirror,
                      hipStream_t s) {
  // (bands of <= 1 GiB of moments: smaller bands mean more launches of five under-filled kernels each -- 2048^2 at (48, 20):
  //  28.8 ms with 1-GiB bands, 45 ms at 256 MiB, 187 ms at 32 MiB)
  int64_t band = (int64_t)((size_t)1 << 30) / ((int64_t)p->n_poly * W * (int64_t)sizeof(double));
  band = band < 8 ? 8 : (band > n_rows ? n_rows : band);
  if (p->path != ZK_PATH_SEPARABLE && zk_plan_auto_direct(p, 1, dtype) && !getenv("ZK_NO_DIRECT")) band = zk_direct_band_rows(p, W, band, n_rows);
  const size_t need = (size_t)p->n_poly * band * W * sizeof(double);
  if (p->d_scratch_bytes < need) {
    if (p->d_scratch) ZK_HIP(hipFree(p->d_scratch));
    p->d_scratch = nullptr;
    p->d_scratch_bytes = 0;
    ZK_HIP(hipMalloc((void**)&p->d_scratch, need));
    p->d_scratch_bytes = need;
  }
  const long long plane = zk_out_plane(p, n_rows, W);
  for (int64_t b0 = 0; b0 < n_rows; b0 += band) {
    const int64_t nb = n_rows - b0 < band ? n_rows - b0 : band;
    const long long keep = p->out_plane;
    p->out_plane = 0;  // the scratch matrix is compact
    // the band's moments from the kernel family ZK_PATH_AUTO takes at this order (zk_api.hip: the plain sum on the matrix
    // cores where the polynomial kernels miss the parity criterion), unless the separable family is forced
    const bool direct = p->path != ZK_PATH_SEPARABLE && zk_plan_auto_direct(p, 1, dtype) && !getenv("ZK_NO_DIRECT");
    int rc = direct ? zk_launch_direct_frame(p, in, dtype, H, W, row0 + b0, nb, p->d_scratch, s)
                    : zk_launch_sep_frame(p, in, dtype, H, W, row0 + b0, nb, p->d_scratch, s);
    p->out_plane = keep;
    if (rc) return rc;
    if ((rc = zk_prof_begin(p, s))) return rc;
    const dim3 grid((unsigned)((nb * W + 255) / 256));
    if (p->sep->kernel_nmax == 20)
      hipLaunchKernelGGL(zk_maps_planes_kernel<20>, grid, dim3(256), 0, s, p->d_scratch, d_trig, rot, ab, mirror, prm,
                         (int)nb, (int)W, (int)b0, plane);
    else
      hipLaunchKernelGGL(zk_maps_planes_kernel<24>, grid, dim3(256), 0, s, p->d_scratch, d_trig, rot, ab, mirror, prm,
                         (int)nb, (int)W, (int)b0, plane);
    ZK_HIP(hipGetLastError());
    if ((rc = zk_prof_end(p, s))) return rc;
  }
  return 0;
}
#endif

#if ZK_NMAX_GROUP == 3
// n_max 25-40: as above with the moments of a band from the matrix-core plain sum (zk_direct_patches.hip)
template <int NMAX>
static int planes_band_g3(zk_plan* p, const zk_maps_params& prm, const double* d_trig, double* rot, double* ab, double* mirror,
                          int64_t nb, int64_t W, int64_t b0, long long plane, hipStream_t s) {
  int rc = zk_prof_begin(p, s);
  if (rc) return rc;
  hipLaunchKernelGGL(zk_maps_planes_kernel<NMAX>, dim3((unsigned)((nb * W + 255) / 256)), dim3(256), 0, s, p->d_scratch, d_trig,
                     rot, ab, mirror, prm, (int)nb, (int)W, (int)b0, plane);
  ZK_HIP(hipGetLastError());
  return zk_prof_end(p, s);
}

int zk_maps_planes_large(zk_plan* p, int knm, const void* in, int dtype, int64_t H, int64_t W, int64_t row0, int64_t n_rows,
                      const zk_maps_params& prm, const double* d_trig, double* rot, double* ab, double* mirror,
                      hipStream_t s) {
  int64_t band = (int64_t)((size_t)1 << 30) / ((int64_t)p->n_poly * W * (int64_t)sizeof(double));
  band = band < 8 ? 8 : (band > n_rows ? n_rows : band);
  band = zk_direct_band_rows(p, W, band, n_rows);
  const size_t need = (size_t)p->n_poly * band * W * sizeof(double);
  if (p->d_scratch_bytes < need) {
    if (p->d_scratch) ZK_HIP(hipFree(p->d_scratch));
    p->d_scratch = nullptr;
    p->d_scratch_bytes = 0;
    ZK_HIP(hipMalloc((void**)&p->d_scratch, need));
    p->d_scratch_bytes = need;
  }
  const long long plane = zk_out_plane(p, n_rows, W);
  for (int64_t b0 = 0; b0 < n_rows; b0 += band) {
    const int64_t nb = n_rows - b0 < band ? n_rows - b0 : band;
    const long long keep = p->out_plane;
    p->out_plane = 0;  // the scratch matrix is compact
    int rc = zk_launch_direct_frame(p, in, dtype, H, W, row0 + b0, nb, p->d_scratch, s);
    p->out_plane = keep;
    if (rc) return rc;
    switch (knm) {
      case 28: rc = planes_band_g3<28>(p, prm, d_trig, rot, ab, mirror, nb, W, b0, plane, s); break;
      case 32: rc = planes_band_g3<32>(p, prm, d_trig, rot, ab, mirror, nb, W, b0, plane, s); break;
      case 36: rc = planes_band_g3<36>(p, prm, d_trig, rot, ab, mirror, nb, W, b0, plane, s); break;
      case 40: rc = planes_band_g3<40>(p, prm, d_trig, rot, ab, mirror, nb, W, b0, plane, s); break;
      default: return zk_fail(ZK_E_BADARG, "no symmetry-map planes kernel for this n_max");
    }
    if (rc) return rc;
  }
  return 0;
}

int zk_maps_rows_large(zk_plan* p, int knm, const double* mom, int64_t n_rows, const zk_maps_params& prm,
                             const double* d_trig, double* rot, double* ab, double* mirror, hipStream_t s) {
  switch (knm) {
    case 28: return launch_rows_one<28>(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
    case 32: return launch_rows_one<32>(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
    case 36: return launch_rows_one<36>(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
    case 40: return launch_rows_one<40>(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
  }
  return zk_fail(ZK_E_BADARG, "no symmetry-map rows kernel for this n_max");
}
#endif

#if ZK_NMAX_GROUP == 0
int zk_maps_planes_g2(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0, int64_t n_rows,
                      const zk_maps_params& prm, const double* d_trig, double* rot, double* ab, double* mirror,
                      hipStream_t s);
int zk_maps_dispatch_g1(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0, int64_t n_rows,
                        const zk_maps_params& prm, const double* d_trig, double* rot, double* ab, double* mirror,
                        hipStream_t s);

// n_max 25-40 (no polynomial tables: the reference basis is not the polynomial there): the moments of a row band come
// from the matrix-core plain sum, the planes kernel of group 3 turns them into the maps
static bool maps_on_direct(const zk_plan* p) {
  const int n_max = zk_full_set_nmax(p);
  return !p->sep && p->direct && n_max > 24 && n_max <= 40;
}

// the kernel instance a plan's maps run on
static int maps_kernel_nmax(const zk_plan* p) {
  if (p->sep) return p->sep->kernel_nmax;
  return (zk_full_set_nmax(p) + 3) / 4 * 4;  // 28, 32, 36, 40
}

bool zk_sep_maps_available(const zk_plan* p, int dtype) {
  // n_max <= 16: fused in one kernel; 17-24: dense passes + planes kernel; 25-40: matrix-core sums + planes kernel
  return zk_sep_frame_available(p, dtype) || (maps_on_direct(p) && zk_direct_frame_available(p, dtype));
}

int zk_maps_planes_large(zk_plan* p, int knm, const void* in, int dtype, int64_t H, int64_t W, int64_t row0, int64_t n_rows,
                      const zk_maps_params& prm, const double* d_trig, double* rot, double* ab, double* mirror,
                      hipStream_t s);
int zk_maps_rows_large(zk_plan* p, int knm, const double* mom, int64_t n_rows, const zk_maps_params& prm,
                             const double* d_trig, double* rot, double* ab, double* mirror, hipStream_t s);

int zk_maps_rows_dispatch_g1(zk_plan* p, const double* mom, int64_t n_rows, const zk_maps_params& prm, const double* d_trig,
                             double* rot, double* ab, double* mirror, hipStream_t s);
int zk_maps_rows_dispatch_g2(zk_plan* p, const double* mom, int64_t n_rows, const zk_maps_params& prm, const double* d_trig,
                             double* rot, double* ab, double* mirror, hipStream_t s);

// options -> kernel parameters + the device table of fold weights / angles (cached per option set)
static int maps_setup(zk_plan* p, const int32_t* folds, int n_folds, const int32_t* m_unselect, int n_unselect, int p_norm,
                      const double* theta, int n_theta, double* rot, double* mirror, zk_maps_params* prm_out,
                      const double** d_trig_out) {
  if (n_folds < 0 || n_folds > ZK_MAX_FOLDS) return zk_fail(ZK_E_BADARG, "at most 8 folds per call");
  if (p_norm != 0 && p_norm != 2) return zk_fail(ZK_E_BADARG, "p must be 2 or 0 (None)");
  if (rot && (!folds || n_folds == 0)) return zk_fail(ZK_E_BADARG, "rot output requested without folds");
  if (mirror && (!theta || n_theta <= 0)) return zk_fail(ZK_E_BADARG, "mirror output requested without theta");
  const int knm = maps_kernel_nmax(p);
  zk_maps_params prm = {};
  prm.n_folds = rot ? n_folds : 0;
  prm.normalize = p_norm == 2;
  prm.n_theta = mirror ? n_theta : 0;
  prm.plan_nmax = zk_full_set_nmax(p);
  prm.unselect_mask = 0;
  for (int k = 0; k < n_unselect; ++k) {
    const int am = m_unselect[k] < 0 ? -m_unselect[k] : m_unselect[k];
    if (am < 64) prm.unselect_mask |= 1ull << am;
  }
  if (!(prm.unselect_mask & 1)) return zk_fail(ZK_E_BADARG, "m=0 must be included in m_unselect.");
  // device table: fold weights, then the trig rows (see the kernel)
  // (+ one spare row of zeros after the angles: the pipelined mirror scan prefetches one row ahead)
  std::vector<double> tab((size_t)ZK_MAX_FOLDS * ZK_MAPS_WROW + (size_t)(prm.n_theta > 0 ? n_theta + 1 : 0) * 2 * knm + 16, 0.0);
  for (int f = 0; f < prm.n_folds; ++f) {
    const int fold = folds[f];
    if (fold <= 0) return zk_fail(ZK_E_BADARG, "folds must be positive");
    for (int am = 0; am <= knm; ++am) {
      // reference construct_rot_maps_matrix (_zmoments.py:199-235): 1 where |m| % fold == 0 and |m| > 1,
      // 0 for |m| in {0, 1}, -1/(fold-1) elsewhere (0 when fold == 1)
      double w;
      if (am <= 1) w = 0.0;
      else if (am % fold == 0) w = 1.0;
      else w = fold > 1 ? -1.0 / (double)(fold - 1) : 0.0;
      if ((prm.unselect_mask >> am) & 1) w = 0.0;  // dropped before the weights apply
      tab[(size_t)f * ZK_MAPS_WROW + am] = w;
    }
  }
  if (prm.n_theta > 0) {
    prm.theta_sym = n_theta % 4 == 0;
    for (int i = 0; i < n_theta && prm.theta_sym; ++i)
      prm.theta_sym = fabs(theta[i] - 2.0 * M_PI * (double)i / (double)n_theta) <= 1e-12;
    if (prm.theta_sym && n_theta % 8 == 0) prm.theta_sym = 2;  // eighth-grid scan
    double* tr = tab.data() + (size_t)ZK_MAX_FOLDS * ZK_MAPS_WROW;
    if (prm.theta_sym == 2) {
      // (cos theta_i, sin theta_i) for i = 0 .. n_theta/8, padded to a multiple of four rows with the last row
      const int last = n_theta / 8, rows4 = (last + 4) & ~3;
      tab.resize((size_t)ZK_MAX_FOLDS * ZK_MAPS_WROW + 2 * (size_t)rows4 + 16);
      tr = tab.data() + (size_t)ZK_MAX_FOLDS * ZK_MAPS_WROW;
      for (int i = 0; i < rows4; ++i) {
        const int k = i < last ? i : last;
        tr[2 * i] = cos(theta[k]);
        tr[2 * i + 1] = sin(theta[k]);
      }
      for (size_t z = 2 * (size_t)rows4; z < 2 * (size_t)rows4 + 16; ++z) tr[z] = 0.0;
    } else
    for (int i = 0; i < n_theta; ++i)
      for (int m = 0; m <= knm; ++m) {
        if (m == 0) continue;  // m = 0 is always unselected
        tr[(size_t)i * 2 * knm + m - 1] = cos((double)m * theta[i]);
        tr[(size_t)i * 2 * knm + knm + m - 1] = sin((double)m * theta[i]);
      }
  }
  zk_plan* t = p;
  const double* d_trig = nullptr;
  for (size_t k = 0; k < t->trig_cache.size() && !d_trig; ++k)
    if (t->trig_cache[k].host == tab) {
      d_trig = t->trig_cache[k].dev;
      if (k) std::swap(t->trig_cache[k], t->trig_cache[k - 1]);  // drift towards the front: eviction takes the back
    }
  if (!d_trig) {
    // a new option set: a blocking upload into a fresh table (first call with these options only -- repeated
    // calls never synchronise).  When the cache is full the least recently used table goes, after the device
    // has drained (some launch may still read it).
    if (t->trig_cache.size() >= ZK_TRIG_CACHE) {
      ZK_HIP(hipDeviceSynchronize());
      if (t->trig_cache.back().dev) (void)hipFree(t->trig_cache.back().dev);
      t->trig_cache.pop_back();
    }
    zk_plan::trig_entry e;
    ZK_HIP(hipMalloc((void**)&e.dev, tab.size() * sizeof(double)));
    hipError_t he = hipMemcpy(e.dev, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice);
    if (he != hipSuccess) {
      (void)hipFree(e.dev);
      return zk_hip_fail(he, "hipMemcpy(trig table)");
    }
    e.host = std::move(tab);
    d_trig = e.dev;
    t->trig_cache.insert(t->trig_cache.begin(), std::move(e));
  }
  *prm_out = prm;
  *d_trig_out = d_trig;
  return 0;
}

int zk_launch_sep_maps(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0, int64_t n_rows,
                       const int32_t* folds, int n_folds, const int32_t* m_unselect, int n_unselect, int p_norm,
                       const double* theta, int n_theta, double* rot, double* ab, double* mirror, hipStream_t s) {
  if (!zk_sep_maps_available(p, dtype))
    return zk_fail(ZK_E_BADARG, "plan has no symmetry-map kernels (needs the full Zernike set, n_max <= 40)");
  zk_maps_params prm;
  const double* d_trig = nullptr;
  int rc = maps_setup(p, folds, n_folds, m_unselect, n_unselect, p_norm, theta, n_theta, rot, mirror, &prm, &d_trig);
  if (rc) return rc;
  const int knm = maps_kernel_nmax(p);
  if (knm > 24) return zk_maps_planes_large(p, knm, in, dtype, H, W, row0, n_rows, prm, d_trig, rot, ab, mirror, s);
  if (knm > 16) return zk_maps_planes_g2(p, in, dtype, H, W, row0, n_rows, prm, d_trig, rot, ab, mirror, s);
  if (knm > 12) return zk_maps_dispatch_g1(p, in, dtype, H, W, row0, n_rows, prm, d_trig, rot, ab, mirror, s);
  return zk_maps_dispatch(p, in, dtype, H, W, row0, n_rows, prm, d_trig, rot, ab, mirror, s);
}

// the same tail on a batch of moment vectors already in device memory: (N, n_poly) -> (N, n_folds), (N, N_c), (N)
int zk_launch_maps_rows(zk_plan* p, const double* mom, int64_t n_rows, const int32_t* folds, int n_folds, const int32_t* m_unselect,
                        int n_unselect, int p_norm, const double* theta, int n_theta, double* rot, double* ab, double* mirror,
                        hipStream_t s) {
  if ((!p->sep && !maps_on_direct(p)) || zk_full_set_nmax(p) < 0)
    return zk_fail(ZK_E_BADARG, "plan has no symmetry-map kernels (needs the full Zernike set, n_max <= 40)");
  zk_maps_params prm;
  const double* d_trig = nullptr;
  int rc = maps_setup(p, folds, n_folds, m_unselect, n_unselect, p_norm, theta, n_theta, rot, mirror, &prm, &d_trig);
  if (rc) return rc;
  const int knm = maps_kernel_nmax(p);
  if (knm > 24) return zk_maps_rows_large(p, knm, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
  if (knm > 16) return zk_maps_rows_dispatch_g2(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
  if (knm > 12) return zk_maps_rows_dispatch_g1(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
  return zk_maps_rows_dispatch(p, mom, n_rows, prm, d_trig, rot, ab, mirror, s);
}
#endif
