// zk_fold.hip -- builds the parity-folded tables of a plan (host side; see zk_fold.h).
#include <math.h>

#include <algorithm>

#include "zk_fold.h"

// NMAX values the folded frame kernel is instantiated for (zk_fast_frame.hip);
// a plan with n_max below an entry is zero-padded up to it.
static const int kKernelNmax[] = {4, 6, 8, 10, 12};

static int pick_kernel_nmax(int n_max) {
  for (int k : kKernelNmax)
    if (n_max <= k) return k;
  return -1;
}

int zk_full_set_nmax(const zk_plan* p) {
  // the plan's (n, m) must be the complete real set 0..n_max in reference order (_zps.py:77-81)
  int n_max = -1;
  for (int k = 0; k <= 64; ++k)
    if ((k + 1) * (k + 2) / 2 == p->n_poly) n_max = k;
  if (n_max < 0) return -1;
  int j = 0;
  for (int n = 0; n <= n_max; ++n)
    for (int m = -n; m <= n; m += 2, ++j)
      if (p->n[j] != n || p->m[j] != m) return -1;
  return n_max;
}

template <typename T>
static int upload(T** dst, const std::vector<T>& src) {
  if (src.empty()) return 0;
  ZK_HIP(hipMalloc((void**)dst, src.size() * sizeof(T)));
  ZK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

void zk_fold_free(zk_plan* p) {
  zk_fold_tables* f = p->fold;
  if (!f) return;
  if (f->d_colmap) (void)hipFree(f->d_colmap);
  if (f->d_fpx_off) (void)hipFree(f->d_fpx_off);
  if (f->d_ftab) (void)hipFree(f->d_ftab);
  delete f;
  p->fold = nullptr;
}

int zk_fold_build(zk_plan* p, const double* basis) {
  const int K = p->size, NP = p->n_poly;
  const int n_max = zk_full_set_nmax(p);
  if (n_max < 0) return 0;  // not a standard set: generic kernels only
  const int knm = pick_kernel_nmax(n_max);
  if (knm < 0) return 0;
  const int npk = (knm + 1) * (knm + 2) / 2;

  // slot order of the kernel set and its map to the plan's columns
  std::vector<int> slot_col;  // [npk] plan column or -1
  std::vector<int> slot_cls;
  for (int cls = 0; cls < 4; ++cls) {
    int j = 0;
    for (int n = 0; n <= knm; ++n)
      for (int m = -n; m <= n; m += 2, ++j)
        if (zk_class_of(m) == cls) {
          slot_col.push_back(n <= n_max ? j : -1);
          slot_cls.push_back(cls);
        }
  }

  const double inv_area = 1.0 / (M_PI * (double)K * (double)K / 4.0);
  std::vector<double> vmax(NP, 0.0);
  for (int j = 0; j < NP; ++j)
    for (int t = 0; t < K * K; ++t) vmax[j] = std::max(vmax[j], fabs(basis[(size_t)j * K * K + t]));

  const int Q = (K + 1) / 2;
  auto V = [&](int j, int r, int c) { return basis[((size_t)j * K + r) * K + c]; };
  std::vector<double> qtab((size_t)Q * Q * npk, 0.0);  // folded value per quadrant pixel and slot
  std::vector<char> qact((size_t)Q * Q, 0);
  for (int r = 0; r < Q; ++r)
    for (int c = 0; c < Q; ++c) {
      const int rm = K - 1 - r, cm = K - 1 - c;
      const double w = (r == rm ? 0.5 : 1.0) * (c == cm ? 0.5 : 1.0);
      bool act = false;
      for (int s = 0; s < npk; ++s) {
        const int j = slot_col[s];
        if (j < 0) continue;
        const double sx = (slot_cls[s] == ZK_OE || slot_cls[s] == ZK_OO) ? -1.0 : 1.0;
        const double sy = (slot_cls[s] == ZK_EO || slot_cls[s] == ZK_OO) ? -1.0 : 1.0;
        const double va = V(j, r, c), vb = sx * V(j, r, cm), vc = sy * V(j, rm, c), vd = sx * sy * V(j, rm, cm);
        const double mean = 0.25 * (va + vb + vc + vd);
        const double tol = 1e-9 * vmax[j];
        if (fabs(va - mean) > tol || fabs(vb - mean) > tol || fabs(vc - mean) > tol || fabs(vd - mean) > tol)
          return 0;  // basis does not have the mirror parities (e.g. a mask edge split by rounding)
        act = act || va != 0.0 || vb != 0.0 || vc != 0.0 || vd != 0.0;
        qtab[((size_t)r * Q + c) * npk + s] = mean * w * inv_area;
      }
      qact[(size_t)r * Q + c] = act;
    }

  zk_fold_tables* f = new zk_fold_tables();
  p->fold = f;
  f->kernel_nmax = knm;
  f->np_kernel = npk;
  int rc = upload(&f->d_colmap, std::vector<int32_t>(slot_col.begin(), slot_col.end()));
  if (rc) return rc;

  // ---- frame kernel ------------------------------------------------------------------
  f->tile_pitch = K + 63;
  std::vector<int4> off;
  std::vector<double> ftab;
  for (int r = 0; r < Q; ++r)
    for (int c = 0; c < Q; ++c) {
      if (!qact[(size_t)r * Q + c]) continue;
      const int rm = K - 1 - r, cm = K - 1 - c, tp = f->tile_pitch;
      off.push_back(make_int4(r * tp + c, r * tp + cm, rm * tp + c, rm * tp + cm));
      const double* src = &qtab[((size_t)r * Q + c) * npk];
      ftab.insert(ftab.end(), src, src + npk);
    }
  f->n_fpx = (int)off.size();
  if ((rc = upload(&f->d_fpx_off, off))) return rc;
  if ((rc = upload(&f->d_ftab, ftab))) return rc;

  return 0;
}
