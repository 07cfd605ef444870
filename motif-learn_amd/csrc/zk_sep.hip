// zk_sep.hip -- host side of the row-separable path: builds the Legendre tables and the
// Zernike <- Legendre-product matrix T of a plan (see zk_sep.h), in extended precision.
#include <math.h>

#include <algorithm>

#include "zk_sep.h"

typedef long double ld;

static const int kSepNmax[] = {4, 6, 8, 10, 12, 14, 16, 20, 24};  // instantiated kernels (20, 24: one pass per parity class)

// x^a = sum_i L[a][i] P_i(x), from x P_i = ((i+1) P_{i+1} + i P_{i-1}) / (2i+1); all terms positive.
static std::vector<std::vector<ld>> monomial_to_legendre(int deg) {
  std::vector<std::vector<ld>> L(deg + 1, std::vector<ld>(deg + 2, 0.0L));
  L[0][0] = 1.0L;
  for (int a = 0; a < deg; ++a)
    for (int i = 0; i <= a; ++i) {
      const ld v = L[a][i];
      if (v == 0.0L) continue;
      L[a + 1][i + 1] += v * (ld)(i + 1) / (ld)(2 * i + 1);
      if (i > 0) L[a + 1][i - 1] += v * (ld)i / (ld)(2 * i + 1);
    }
  return L;
}

static ld factl(int n) {
  ld f = 1.0L;
  for (int i = 2; i <= n; ++i) f *= (ld)i;
  return f;
}

static ld binom(int n, int k) { return factl(n) / (factl(k) * factl(n - k)); }  // exact for n <= 20

// Coefficients mono[a][b] of  R_n^{|m|}(rho) * (cos(m t) | sin(|m| t))  as a polynomial in x, y:
//   rho^{k+2t} cos(k t) = (x^2+y^2)^t Re (x+iy)^k,  sin: Im  (reference _zps.py:52-64, 85-88).
static void zernike_monomials(int n, int m, int deg, std::vector<ld>& mono) {
  mono.assign((size_t)(deg + 1) * (deg + 1), 0.0L);
  const int k = m < 0 ? -m : m;
  for (int s = 0; s <= (n - k) / 2; ++s) {
    const ld c = ((s & 1) ? -1.0L : 1.0L) * factl(n - s) /
                 (factl(s) * factl((n + k) / 2 - s) * factl((n - k) / 2 - s));
    const int t = (n - 2 * s - k) / 2;
    for (int u = 0; u <= t; ++u)
      for (int q = 0; q <= k; ++q) {
        if ((m >= 0) == ((q & 1) != 0)) continue;  // cos keeps even q, sin keeps odd q
        const int h = m >= 0 ? q / 2 : (q - 1) / 2;
        const ld sg = (h & 1) ? -1.0L : 1.0L;
        const int a = 2 * (t - u) + k - q, b = 2 * u + q;
        mono[(size_t)a * (deg + 1) + b] += c * binom(t, u) * binom(k, q) * sg;
      }
  }
}

template <typename T>
static int upload(T** dst, const std::vector<T>& src) {
  if (src.empty()) return 0;
  ZK_HIP(hipMalloc((void**)dst, src.size() * sizeof(T)));
  ZK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

void zk_sep_free(zk_plan* p) {
  zk_sep_tables* t = p->sep;
  if (!t) return;
  if (t->d_xq) (void)hipFree(t->d_xq);
  if (t->d_T) (void)hipFree(t->d_T);
  if (t->d_colmap) (void)hipFree(t->d_colmap);
  if (t->d_rows) (void)hipFree(t->d_rows);
  if (t->d_cmin) (void)hipFree(t->d_cmin);
  if (t->d_strip_rows) (void)hipFree(t->d_strip_rows);
  if (t->d_psplit_alloc) (void)hipFree(t->d_psplit_alloc);
  for (auto& b : t->batch) {
    if (b.d_units) (void)hipFree(b.d_units);
    if (b.d_row_starts) (void)hipFree(b.d_row_starts);
  }
  if (t->d_pfull_alloc) (void)hipFree(t->d_pfull_alloc);
  for (auto& b : t->stream) {
    if (b.d_units) (void)hipFree(b.d_units);
    if (b.d_rows) (void)hipFree(b.d_rows);
  }
  delete t;
  p->sep = nullptr;
}

int zk_sep_build(zk_plan* p, const double* basis) {
  const int K = p->size, NP = p->n_poly;
  const int n_max = zk_full_set_nmax(p);
  if (n_max < 0 || K < 2) return 0;
  int knm = -1;
  for (int k : kSepNmax)
    if (n_max <= k) {
      knm = k;
      break;
    }
  if (knm < 0) return 0;
  const int npk = (knm + 1) * (knm + 2) / 2;
  const int D = knm + 1;

  // ---- grid and Legendre values (numpy.linspace arithmetic: start + i*step, exact endpoint) ----
  std::vector<double> x(K);
  const double step = 2.0 / (double)(K - 1);
  for (int c = 0; c < K; ++c) x[c] = (double)c * step + -1.0;
  x[K - 1] = 1.0;
  std::vector<ld> P((size_t)K * D);
  for (int c = 0; c < K; ++c) {
    P[(size_t)c * D] = 1.0L;
    if (D > 1) P[(size_t)c * D + 1] = (ld)x[c];
    for (int i = 1; i + 1 < D; ++i)
      P[(size_t)c * D + i + 1] = ((ld)(2 * i + 1) * (ld)x[c] * P[(size_t)c * D + i] - (ld)i * P[(size_t)c * D + i - 1]) / (ld)(i + 1);
  }

  // ---- T: Zernike j  <-  Legendre products (a, b), extended precision --------------------------
  const auto L = monomial_to_legendre(knm);
  std::vector<ld> Tfull((size_t)NP * D * D, 0.0L);  // [j][a][b]
  std::vector<ld> mono;
  for (int j = 0; j < NP; ++j) {
    const int n = p->n[j], m = p->m[j];
    zernike_monomials(n, m, knm, mono);
    const ld norm = sqrtl((ld)(2 * (n + 1)) / (m == 0 ? 2.0L : 1.0L));
    for (int a = 0; a <= knm; ++a)
      for (int b = 0; a + b <= knm; ++b) {
        const ld cf = mono[(size_t)a * D + b];
        if (cf == 0.0L) continue;
        for (int ia = 0; ia <= a; ++ia)
          for (int ib = 0; ib <= b; ++ib)
            Tfull[((size_t)j * D + ia) * D + ib] += cf * L[a][ia] * L[b][ib];
      }
    for (int t = 0; t < D * D; ++t) Tfull[(size_t)j * D * D + t] *= norm;
  }

  // ---- check the caller's basis: mirror-symmetric disk, contiguous row ranges, V == T P P ------
  std::vector<char> disk((size_t)K * K, 0);
  for (int r = 0; r < K; ++r)
    for (int c = 0; c < K; ++c)
      for (int j = 0; j < NP && !disk[(size_t)r * K + c]; ++j) disk[(size_t)r * K + c] = basis[((size_t)j * K + r) * K + c] != 0.0;
  for (int r = 0; r < K; ++r)
    for (int c = 0; c < K; ++c)
      if (disk[(size_t)r * K + c] != disk[(size_t)r * K + (K - 1 - c)] ||
          disk[(size_t)r * K + c] != disk[(size_t)(K - 1 - r) * K + c])
        return 0;
  const int Q = (K + 1) / 2;
  std::vector<zk_sep_row> rows;
  for (int r = 0; r < Q; ++r) {
    int cmin = Q;
    for (int c = Q - 1; c >= 0 && disk[(size_t)r * K + c]; --c) cmin = c;
    for (int c = 0; c < cmin; ++c)
      if (disk[(size_t)r * K + c]) return 0;  // not a suffix: cannot happen for a disk
    if (cmin < Q) rows.push_back({r, cmin});
  }
  if (rows.empty()) return 0;
  // The kernels use the exact polynomial; the caller's float64 basis carries the rounding of the reference's
  // factorial sums (_zps.py:52-64): ~2e-11 of max|V| at n_max 16, ~1e-9 at 20, ~3e-8 at 24 (and 2e-4 by 36).
  // Up to 24 that is two orders inside the 1e-6 parity tolerance and the substitution is accepted; the
  // tolerance below still rejects any basis that is not this polynomial set.
  const ld tol = n_max <= 16 ? 1e-9L : n_max <= 20 ? 4e-9L : 1e-7L;
  for (int j = 0; j < NP; ++j) {
    double vmax = 0.0;
    for (int t = 0; t < K * K; ++t) vmax = std::max(vmax, fabs(basis[(size_t)j * K * K + t]));
    // row by row, the y direction first: V_j(r, c) = sum_a [sum_b T_j(a, b) P_b(y_r)] P_a(x_c) -- K (D^2 / 2 + K D) operations
    // per function instead of K^2 D^2 / 2 (plan creation at (64, 12): 51 -> 12 ms, at (48, 24): 330 -> 50 ms)
    std::vector<ld> ca(D);
    for (int r = 0; r < K; ++r) {
      for (int a = 0; a <= knm; ++a) {
        ld s = 0.0L;
        for (int b = 0; a + b <= knm; ++b) s += Tfull[((size_t)j * D + a) * D + b] * P[(size_t)r * D + b];
        ca[a] = s;
      }
      for (int c = 0; c < K; ++c) {
        if (!disk[(size_t)r * K + c]) continue;
        ld v = 0.0L;
        for (int a = 0; a <= knm; ++a) v += ca[a] * P[(size_t)c * D + a];
        if (fabsl(v - (ld)basis[((size_t)j * K + r) * K + c]) > tol * (ld)vmax + 1e-300L) return 0;
      }
    }
  }

  // ---- device tables -----------------------------------------------------------------------
  zk_sep_tables* t = new zk_sep_tables();
  p->sep = t;
  t->kernel_nmax = knm;
  t->np_kernel = npk;
  t->Q = Q;
  t->tile_pitch = K + 63;
  std::vector<double> xq((size_t)(Q + 1) * ZK_SEP_ROW, 0.0);  // + one zero row: the pipelined loops prefetch one row ahead
  for (int c = 0; c < Q; ++c) {
    const double w = (c == K - 1 - c) ? 0.5 : 1.0;
    for (int a = 0; a < D; ++a) xq[(size_t)c * ZK_SEP_ROW + a] = (double)(P[(size_t)c * D + a] * (ld)w);
  }
  // class-ordered slots: Zernike (rows of T) and Legendre products (columns of T)
  std::vector<int32_t> colmap;
  std::vector<double> Tdev;
  const ld inv_area = 1.0L / (M_PIl * (ld)K * (ld)K / 4.0L);
  for (int cls = 0; cls < 4; ++cls) {
    std::vector<int> zj;  // plan column of each Zernike slot of this class (or -1)
    std::vector<int> zn;  // its radial order
    int j = 0;
    for (int n = 0; n <= knm; ++n)
      for (int m = -n; m <= n; m += 2, ++j)
        if (zk_class_of(m) == cls) {
          zj.push_back(n <= n_max ? j : -1);
          zn.push_back(n);
        }
    std::vector<std::pair<int, int>> ab;
    for (int a = 0; a <= knm; ++a)
      for (int b = 0; a + b <= knm; ++b) {
        const int c2 = (a & 1) ? ((b & 1) ? ZK_OO : ZK_OE) : ((b & 1) ? ZK_EO : ZK_EE);
        if (c2 == cls) ab.push_back({a, b});
      }
    if (zj.size() != ab.size()) return zk_fail(ZK_E_BADARG, "internal: class size mismatch");
    for (size_t jj = 0; jj < zj.size(); ++jj) {
      colmap.push_back(zj[jj]);
      for (size_t ii = 0; ii < ab.size(); ++ii) {
        const bool structural_zero = ab[ii].first + ab[ii].second > zn[jj];  // total degree of V_j is n_j
        ld v = 0.0L;
        if (zj[jj] >= 0) v = Tfull[((size_t)zj[jj] * D + ab[ii].first) * D + ab[ii].second] * inv_area;
        if (structural_zero) {
          if (v != 0.0L) return zk_fail(ZK_E_BADARG, "internal: T entry above the polynomial degree");
          continue;  // packed rows: the kernels skip these entries (zk_sep_pack)
        }
        Tdev.push_back((double)v);
      }
    }
  }
  int rc;
  if ((rc = upload(&t->d_xq, xq))) return rc;
  t->d_yq = t->d_xq;  // same grid in x and y
  if ((rc = upload(&t->d_T, Tdev))) return rc;
  if ((rc = upload(&t->d_colmap, colmap))) return rc;
  t->n_rows = (int)rows.size();
  if ((rc = upload(&t->d_rows, rows))) return rc;
  {
    std::vector<int32_t> cmin_full(K, Q);
    for (const zk_sep_row& row : rows) cmin_full[row.r] = cmin_full[K - 1 - row.r] = row.cmin;
    bool nested = true;  // the strip kernel relies on limits that shrink towards the centre row (a disk's do)
    for (int r = 0; r + 1 < Q; ++r) nested = nested && cmin_full[r] >= cmin_full[r + 1];
    if (nested && (rc = upload(&t->d_cmin, cmin_full))) return rc;
    if (nested && K % 2 == 0 && Q <= 32) {
      // strip kernel, round-3 form (zk_sep_strip.hip): frame row fr is window row fr of the upper output and fr - 1 of the
      // lower one; the sweep from the centre outwards first reaches the limit of the output that sees the frame row as its
      // narrower window row (n1 column pairs), then runs on to the other's (n2 more)
      std::vector<int32_t> rec(K + 1, 0);
      for (int fr = 0; fr <= K; ++fr) {
        const int c0 = fr < K ? cmin_full[fr] : Q, c1 = fr > 0 ? cmin_full[fr - 1] : Q;
        const bool first_is_1 = fr < Q;
        const int ca = first_is_1 ? c1 : c0, cb = first_is_1 ? c0 : c1;
        if (ca < cb) return zk_fail(ZK_E_BADARG, "internal: strip limits are not nested");
        if (cb >= Q) continue;
        const int n1 = ca < Q ? Q - ca : 0;
        rec[fr] = n1 | ((Q - cb - n1) << 8);
      }
      if ((rc = upload(&t->d_strip_rows, rec))) return rc;
    }
  }

  // ---- batch kernel unit lists ------------------------------------------------------------------
  // A unit is UP quadrant columns c0..c0+UP-1 of one row pair (UP = 16 float32 / 8 float64 pixels = 64 B),
  // fetched as 64-B runs: (r, c0..) and its column mirror (r, K-UP-c0..K-1-c0), same for row K-1-r; for
  // float32 at K == 32 the two runs of a row are the halves of one 128-B line and are fetched as one run
  // (RUN = 8).  Columns >= Q of the last unit of a row belong to the mirrored half and are masked by
  // cmax.  Any K >= UP works: a run is 64 contiguous bytes of one patch row wherever it starts.
  for (int dt = 0; dt < 2; ++dt) {
    const int es = dt == 0 ? 4 : 8, UP = 64 / es;
    if (K < UP) continue;  // (rows need not be 16-B aligned: LDS-DMA sources only need element alignment)
    zk_sep_tables::batch_tables& bt = t->batch[dt];
    bt.run = (dt == 0 && K == 32 && knm <= 16) ? 8 : 4;  // class-pass kernels (n_max > 16) only use 64-B runs
    std::vector<zk_sep_unit> units;
    const int LW = 128 / es;  // pixels per 128-B line: 32 float32, 16 float64
    if (K % (2 * LW) == 0 && knm <= 16) {
      // Wide patches (float32 K % 64 == 0, float64 K % 32 == 0): the quadrant of a row is a whole number
      // of 128-B lines, so a unit is one line of quadrant columns of ONE row -- (y, c0..c0+LW-1) -- and
      // its mirror line, and the two rows of a pair are consumed one after the other
      // (zk_sep_acc::row_pixel / pair_combine).  Every DMA instruction then moves whole lines, as for
      // float32 at K == 32.
      bt.run = 8;
      bt.wide = 1;
      for (const zk_sep_row& row : rows) {
        const size_t first = units.size();
        for (int second = 0; second < 2; ++second) {
          const int y = second ? K - 1 - row.r : row.r;
          for (int c0 = 0; c0 < Q; c0 += LW) {
            if (c0 + LW <= row.cmin) continue;
            zk_sep_unit u = {};
            u.run_off[0] = (y * K + c0) * es;
            u.run_off[1] = (y * K + K - LW - c0) * es;
            u.c0 = c0;
            u.cmin = row.cmin;
            u.r = row.r;
            u.row_end = (Q << 8) | (second << 1);
            units.push_back(u);
          }
        }
        if (units.size() > first) units.back().row_end |= 1;
      }
    } else
    for (const zk_sep_row& row : rows) {
      const int r = row.r, rm = K - 1 - r;
      const size_t first = units.size();
      for (int c0 = 0; c0 < Q; c0 += UP) {
        if (c0 + UP <= row.cmin) continue;  // unit entirely outside the disk
        zk_sep_unit u = {};
        if (bt.run == 8) {
          u.run_off[0] = r * K * es;
          u.run_off[1] = rm * K * es;
        } else {
          u.run_off[0] = (r * K + c0) * es;
          u.run_off[1] = (r * K + K - UP - c0) * es;
          u.run_off[2] = (rm * K + c0) * es;
          u.run_off[3] = (rm * K + K - UP - c0) * es;
        }
        u.c0 = c0;
        u.cmin = row.cmin;
        u.r = r;
        u.row_end = Q << 8;  // bits 8..: cmax (first column that is not a quadrant column); bit 0: row end
        units.push_back(u);
      }
      if (units.size() > first) units.back().row_end |= 1;
    }
    bt.n_units = (int)units.size();
    std::vector<int32_t> starts;
    for (size_t k = 0; k < units.size(); ++k)
      if (k == 0 || (units[k - 1].row_end & 1)) starts.push_back((int32_t)k);
    bt.n_row_starts = (int)starts.size();
    if ((rc = upload(&bt.d_units, units))) return rc;
    if ((rc = upload(&bt.d_row_starts, starts))) return rc;
  }

  // ---- stream batch kernel (zk_sep_stream.hip): full-width table, disk rows and 128-B lines -------
  {
    const int srow = ZK_STREAM_ROW(knm);
    std::vector<double> pfull((size_t)(K + 2 * ZK_STREAM_PAD) * srow, 0.0);
    for (int c = 0; c < K; ++c)
      for (int a = 1; a < D; ++a) pfull[(size_t)(c + ZK_STREAM_PAD) * srow + a - 1] = (double)P[(size_t)c * D + a];
    if ((rc = upload(&t->d_pfull_alloc, pfull))) return rc;
    t->d_pfull = t->d_pfull_alloc + ZK_STREAM_PAD * srow;
    if (knm > 8 && knm <= 2 * ZK_SPLIT_HALF) {
      // the same values split by the parity of the degree: the x-even / x-odd passes of the strip kernel (zk_sep_strip.hip)
      // each request one half of a row
      const int prow = 2 * ZK_SPLIT_HALF;
      std::vector<double> ps((size_t)(K + 2 * ZK_STREAM_PAD) * prow, 0.0);
      for (int c = 0; c < K; ++c)
        for (int a = 1; a < D; ++a)
          ps[(size_t)(c + ZK_STREAM_PAD) * prow + ((a & 1) ? ZK_SPLIT_HALF + (a - 1) / 2 : a / 2 - 1)] = (double)P[(size_t)c * D + a];
      if ((rc = upload(&t->d_psplit_alloc, ps))) return rc;
      t->d_psplit = t->d_psplit_alloc + ZK_STREAM_PAD * prow;
    }
    std::vector<zk_stream_row> srows;
    for (int r = 0; r < K; ++r) {
      int lo = -1, hi = -1;
      for (int c = 0; c < K; ++c)
        if (disk[(size_t)r * K + c]) {
          if (lo < 0) lo = c;
          hi = c;
        }
      if (lo >= 0) srows.push_back({r * K + lo, r * K + hi, r, 0});  // contiguity was checked above
    }
    const int n_srows = (int)srows.size();
    srows.push_back({0x7fffffff, 0x7fffffff, 0, 0});  // sentinel: never reached by a pixel index
    srows.push_back({0x7fffffff, 0x7fffffff, 0, 0});  // (two: the kernel prefetches one entry ahead)
    for (int dt = 0; dt < 2; ++dt) {
      const int es = dt == 0 ? 4 : 8, LW = 128 / es;
      const long long patch_bytes = (long long)K * K * es;
      if (K < 8 || K > 1024 || knm > 16) continue;
      // granules that would cross the end of a patch are clamped back into it (their content is then
      // misplaced), so every disk pixel must lie in the granules below that point
      if ((long long)(srows[n_srows - 1].te + 1) * es > patch_bytes / 16 * 16) continue;
      zk_sep_tables::stream_tables& st = t->stream[dt];
      std::vector<zk_stream_unit> units;
      int ri = 0;
      for (long long off = 0; off < patch_bytes; off += 128) {
        const int t0 = (int)(off / es);
        while (ri < n_srows && srows[ri].te < t0) ++ri;
        if (ri == n_srows) break;
        if (srows[ri].ts >= t0 + LW) continue;  // no disk pixel in this line
        units.push_back({(int32_t)off, t0, ri, off + 128 > patch_bytes ? 1 : 0});
      }
      st.n_units = (int)units.size();
      st.n_rows = n_srows;
      st.aligned = patch_bytes % 128 == 0;
      // preferred where the row-pair kernel has no whole-line units, except the sizes at which its 64-B runs
      // pair up inside lines anyway and it measures level or ahead, and n_max > 12, where both kernels are
      // bound by arithmetic and the folded one does less of it (tools/sweep_batch.py)
      st.preferred = t->batch[dt].run != 8 && K != 16 && !(dt == 0 && K % 32 == 0) && knm <= 12;
      if ((rc = upload(&st.d_units, units))) return rc;
      if ((rc = upload(&st.d_rows, srows))) return rc;
    }
  }
  return 0;
}
