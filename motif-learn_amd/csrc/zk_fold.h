// zk_fold.h -- parity-folded Zernike tables and the device-side accumulation step shared by
// the direct folded frame kernel (zk_fast_frame.hip) and, for the folds, by zk_sep.h.
//
// Every real Zernike function on the symmetric grid linspace(-1,1,K)^2 (reference
// _zps.py:68-72) is even or odd under the column mirror c -> K-1-c and under the row mirror
// r -> K-1-r:  cos(m t) has x-parity (-1)^m and is y-even;  sin(|m| t) has x-parity -(-1)^m
// and is y-odd (_zps.py:85-88).  So the K*K-term inner product of a window with one basis
// function collapses onto the quadrant r,c < ceil(K/2):
//     sum = sum_{r,c in quadrant} V[r,c] * F_class(r,c),
//     F_EE = (a+b)+(c+d)  F_OE = (a-b)+(c-d)  F_EO = (a+b)-(c+d)  F_OO = (a-b)-(c-d)
// with a,b,c,d the window pixels at (r,c), (r,c'), (r',c), (r',c').  The four folded values
// cost 8 adds per quadrant pixel and are shared by all N_poly functions, which cuts the f64
// FMA count 4x.  For float32 pixels the folds are exact in float64.
#pragma once

#include "zk_internal.h"

// ---- compile-time description of the full real Zernike set 0..NMAX -----------------------
// Accumulators are kept in class order [EE | OE | EO | OO], ascending reference index j
// inside a class.
template <int NMAX>
struct zk_set {
  static constexpr int count(int cls) {
    int k = 0;
    for (int n = 0; n <= NMAX; ++n)
      for (int m = -n; m <= n; m += 2) {
        const int am = m < 0 ? -m : m;
        const int c = m >= 0 ? ((am & 1) ? ZK_OE : ZK_EE) : ((am & 1) ? ZK_EO : ZK_OO);
        k += (c == cls);
      }
    return k;
  }
  static constexpr int EE = count(ZK_EE), OE = count(ZK_OE), EO = count(ZK_EO), OO = count(ZK_OO);
  static constexpr int NP = EE + OE + EO + OO;
  static constexpr int cls_of_m(int m) {
    const int am = m < 0 ? -m : m;
    return m >= 0 ? ((am & 1) ? ZK_OE : ZK_EE) : ((am & 1) ? ZK_EO : ZK_OO);
  }
  // class-ordered slot of (n, m): classes in order EE, OE, EO, OO; ascending reference index inside
  static constexpr int slot_of(int n, int m) {
    int k = 0;
    for (int cls = 0; cls < 4; ++cls)
      for (int nn = 0; nn <= NMAX; ++nn)
        for (int mm = -nn; mm <= nn; mm += 2)
          if (cls_of_m(mm) == cls) {
            if (nn == n && mm == m) return k;
            ++k;
          }
    return -1;
  }
  static constexpr int slot_n(int slot) {
    int k = 0;
    for (int cls = 0; cls < 4; ++cls)
      for (int nn = 0; nn <= NMAX; ++nn)
        for (int mm = -nn; mm <= nn; mm += 2)
          if (cls_of_m(mm) == cls) {
            if (k == slot) return nn;
            ++k;
          }
    return -1;
  }
  static constexpr int slot_m(int slot) {
    int k = 0;
    for (int cls = 0; cls < 4; ++cls)
      for (int nn = 0; nn <= NMAX; ++nn)
        for (int mm = -nn; mm <= nn; mm += 2)
          if (cls_of_m(mm) == cls) {
            if (k == slot) return mm;
            ++k;
          }
    return -1;
  }
  // index of the complex moment (n, |m|) in the reference's to_complex() order (n, then m ascending)
  // (closed form: orders below n hold floor(n'/2) + 1 entries each.  The planes / rows kernels call this with a RUN-TIME n --
  //  the counting loop it used to be ran as scalar code in front of every |Z| store: 2.9 of 3.2 ms per 128 x 2048 band at n_max 24)
  static constexpr int complex_index(int n, int am) { return ((n + 1) >> 1) * ((n + 2) >> 1) + ((am - (n & 1)) >> 1); }
  static constexpr int NC = complex_index(NMAX, NMAX) + 1;
  static constexpr int complex_n(int k) {
    int i = 0;
    for (int nn = 0; nn <= NMAX; ++nn)
      for (int mm = nn & 1; mm <= nn; mm += 2) {
        if (i == k) return nn;
        ++i;
      }
    return -1;
  }
  static constexpr int complex_m(int k) {
    int i = 0;
    for (int nn = 0; nn <= NMAX; ++nn)
      for (int mm = nn & 1; mm <= nn; mm += 2) {
        if (i == k) return mm;
        ++i;
      }
    return -1;
  }
};

inline int zk_class_of(int m) {
  const int am = m < 0 ? -m : m;
  return m >= 0 ? ((am & 1) ? ZK_OE : ZK_EE) : ((am & 1) ? ZK_EO : ZK_OO);
}

struct zk_fold_tables {
  int kernel_nmax = -1;     // instantiated NMAX the tables are padded to (>= the plan's n_max)
  int np_kernel = 0;        // zk_set<kernel_nmax>::NP
  int32_t* d_colmap = nullptr;  // [np_kernel] class-ordered slot -> output column, -1 = padding

  // frame kernel: active quadrant pixels in row-major order
  int n_fpx = 0;
  int tile_pitch = 0;       // elements per LDS tile row the offsets were built for
  int4* d_fpx_off = nullptr;    // [n_fpx] tile element offsets of a, b, c, d
  double* d_ftab = nullptr;     // [n_fpx][np_kernel], class order, scaled by weight/area

};

#ifdef __HIPCC__
// Tables are read through the constant address space: a wave-uniform load from it is always a
// scalar load (s_load -> SGPR), whatever stores, LDS-DMA or asm barriers surround it.
#define ZK_CONST __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const ZK_CONST T* zk_const(const T* p) {
  return (const ZK_CONST T*)p;
}

// Dense kernels: stage the zero-padded tile (tile_rows x tile_pitch, frame rows i_first.., columns k_first..) into LDS as
// float64.  256 threads; wave w takes tile rows w, w + 4, ..., lane l the columns l, l + 64, ...  The loads of
// ZK_STAGE_BATCH rows are issued before the first is converted: one exposed memory latency per batch.  (Written as
// load - convert - store per element, the compiler waits for every single load: ~20 serial round trips per wave, a quarter of
// a (32, 8) workgroup's life with the SIMDs half empty -- profiles/r03_strip_notes.txt.)
#ifndef ZK_STAGE_BATCH
#define ZK_STAGE_BATCH 10
#endif
template <typename T>
__device__ __forceinline__ void zk_stage_tile(double* __restrict__ tile, const T* __restrict__ img, int H, int W, int i_first,
                                              int k_first, int tile_rows, int tile_pitch) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (int tr0 = wave; tr0 < tile_rows; tr0 += 4 * ZK_STAGE_BATCH) {
    for (int tc = lane; tc < tile_pitch; tc += 64) {
      const int kk = k_first + tc;
      const bool col_ok = kk >= 0 && kk < W;
      T v[ZK_STAGE_BATCH];
#pragma unroll
      for (int b = 0; b < ZK_STAGE_BATCH; ++b) {
        const int ii = i_first + tr0 + 4 * b;  // (wave-uniform)
        v[b] = (T)0;
        if (tr0 + 4 * b < tile_rows && ii >= 0 && ii < H && col_ok) v[b] = img[(long long)ii * W + kk];
      }
#pragma unroll
      for (int b = 0; b < ZK_STAGE_BATCH; ++b)
        if (tr0 + 4 * b < tile_rows) tile[(tr0 + 4 * b) * tile_pitch + tc] = (double)v[b];
    }
  }
}

// acc[slot] += F_class(slot) * bt[slot] for every slot of the NMAX set; bt is wave-uniform, so
// its elements arrive through scalar loads and feed v_fma_f64 as SGPR operands.
template <int NMAX>
__device__ __forceinline__ void zk_fold_fma(double (&acc)[zk_set<NMAX>::NP], double a, double b, double c,
                                            double d, const ZK_CONST double* bt) {
  using S = zk_set<NMAX>;
  const double s1 = a + b, d1 = a - b, s2 = c + d, d2 = c - d;
  const double fee = s1 + s2, foe = d1 + d2, feo = s1 - s2, foo = d1 - d2;
#pragma unroll
  for (int i = 0; i < S::EE; ++i) acc[i] = __builtin_fma(fee, bt[i], acc[i]);
#pragma unroll
  for (int i = 0; i < S::OE; ++i) acc[S::EE + i] = __builtin_fma(foe, bt[S::EE + i], acc[S::EE + i]);
#pragma unroll
  for (int i = 0; i < S::EO; ++i)
    acc[S::EE + S::OE + i] = __builtin_fma(feo, bt[S::EE + S::OE + i], acc[S::EE + S::OE + i]);
#pragma unroll
  for (int i = 0; i < S::OO; ++i)
    acc[S::EE + S::OE + S::EO + i] =
        __builtin_fma(foo, bt[S::EE + S::OE + S::EO + i], acc[S::EE + S::OE + S::EO + i]);
}
#endif
