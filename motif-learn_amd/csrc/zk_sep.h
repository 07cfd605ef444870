// zk_sep.h -- row-separable evaluation of the Zernike inner products (fast kernels).
//
// Every real Zernike function of radial order n is a bivariate polynomial of total degree n, so on
// the pixel grid x_c = y_c = linspace(-1, 1, K)[c] (reference _zps.py:68-72)
//
//     V_j(r, c) = sum_{a+b <= n_max} T[j][(a,b)] * P_a(x_c) * P_b(y_r)        inside the disk,
//
// with P the Legendre polynomials and T a constant (N_poly x N_poly) matrix that is block-diagonal in
// the four mirror-parity classes (zk_fold.h).  The moment of a window f is then
//
//     Z_j = (1/area) * sum_i T[j][i] * M_i,     M_(a,b) = sum_r P_b(y_r) * sum_{c in disk row r} f[r][c] P_a(x_c)
//
// i.e. a row-separable sum: per disk pixel only (n_max+1) FMAs (x direction), per disk row N_poly FMAs
// (y direction), and one small class-blocked matrix product at the end -- ~1.6x fewer f64 operations
// than the direct folded sum at n_max = 8, and, more importantly, scalar tables of a few KiB
// (P values per column/row + T) that stay resident in the 16-KiB scalar data cache, where the direct
// method streams N_poly doubles per pixel (76 KiB at (32, 8)) through it.
//
// Mirror folding still applies: P_a has x-parity (-1)^a, so with the folds ee/eo/oe/oo of the four
// mirror pixels (zk_fold.h) one quadrant pixel updates, for every a, the two row sums
//     a even:  SEp[a] += ee * P_a(x_c)   (pairs with even b)      SEm[a] += eo * P_a(x_c)  (odd b)
//     a odd :  SOp[a] += oe * P_a(x_c)   (even b)                 SOm[a] += oo * P_a(x_c)  (odd b)
// and a finished row pair r adds  M_(a,b) += P_b(y_r) * S(a, parity of b).
//
// Numerics: T is built in extended precision on the host (zk_sep.hip); its entries reach ~2e2 at
// n_max = 8 (~1e3 at 10, ~2e5 at 16), so the result carries ~1e-13 (1e-11 at 16) * max|Z| of rounding where the direct sum
// carries ~1e-15 -- seven orders inside the 1e-6 parity tolerance.  The plan verifies at creation
// that T reproduces the caller's basis at every disk pixel and otherwise disables this path.
#pragma once

#include <utility>

#include "zk_fold.h"

// ---- compile-time slot tables of the degree-NMAX product set, class order [EE | OE | EO | OO] --------
// class of (a, b): x-parity of P_a is (-1)^a, y-parity of P_b is (-1)^b.
template <int NMAX>
struct zk_sep_set {
  static constexpr int NA = NMAX + 1;            // degrees 0..NMAX
  static constexpr int NE = NMAX / 2 + 1;        // even degrees
  static constexpr int NO = (NMAX + 1) / 2;      // odd degrees
  static constexpr int NP = (NMAX + 1) * (NMAX + 2) / 2;
  static constexpr int cls_of(int a, int b) {
    return (a & 1) ? ((b & 1) ? ZK_OO : ZK_OE) : ((b & 1) ? ZK_EO : ZK_EE);
  }
  static constexpr int cls_count(int cls) {
    int k = 0;
    for (int a = 0; a <= NMAX; ++a)
      for (int b = 0; a + b <= NMAX; ++b) k += cls_of(a, b) == cls;
    return k;
  }
  static constexpr int cls_begin(int cls) {
    int k = 0;
    for (int c = 0; c < cls; ++c) k += cls_count(c);
    return k;
  }
  // slot -> a / b (slots enumerate classes in order, inside a class a ascending then b ascending)
  static constexpr int slot_a(int slot) {
    int k = 0;
    for (int cls = 0; cls < 4; ++cls)
      for (int a = 0; a <= NMAX; ++a)
        for (int b = 0; a + b <= NMAX; ++b)
          if (cls_of(a, b) == cls) {
            if (k == slot) return a;
            ++k;
          }
    return -1;
  }
  static constexpr int slot_b(int slot) {
    int k = 0;
    for (int cls = 0; cls < 4; ++cls)
      for (int a = 0; a <= NMAX; ++a)
        for (int b = 0; a + b <= NMAX; ++b)
          if (cls_of(a, b) == cls) {
            if (k == slot) return b;
            ++k;
          }
    return -1;
  }
};

// Sparsity of T: the Zernike function of radial order n has total degree n, so T[j][(a,b)] = 0 for
// a + b > n_j.  The device table stores, per class, the rows of T packed to their a + b <= n_j entries
// (in slot order); these compile-time tables give every kernel the same addressing (about half of the
// class-block entries are structural zeros).
template <int NMAX>
struct zk_sep_pack {
  using S = zk_sep_set<NMAX>;
  using Z = zk_set<NMAX>;
  struct tables {
    int deg[S::NP];        // a + b of each Legendre-product slot
    int zn[S::NP];         // radial order of each Zernike slot (same class order)
    int row_begin[S::NP];  // offset of the packed row of Zernike slot j in the device table
    int total;
  };
  static constexpr tables make() {
    tables t = {};
    int k = 0;  // same enumerations as zk_sep_set::slot_a / slot_b and zk_set::slot_n, done once
    for (int cls = 0; cls < 4; ++cls)
      for (int a = 0; a <= NMAX; ++a)
        for (int b = 0; a + b <= NMAX; ++b)
          if (S::cls_of(a, b) == cls) t.deg[k++] = a + b;
    k = 0;
    for (int cls = 0; cls < 4; ++cls)
      for (int n = 0; n <= NMAX; ++n)
        for (int m = -n; m <= n; m += 2)
          if (Z::cls_of_m(m) == cls) t.zn[k++] = n;
    int o = 0, b = 0;
    for (int cls = 0; cls < 4; b += S::cls_count(cls), ++cls) {
      const int n = S::cls_count(cls);
      for (int j = 0; j < n; ++j) {
        t.row_begin[b + j] = o;
        for (int i = 0; i < n; ++i) o += t.deg[b + i] <= t.zn[b + j];
      }
    }
    t.total = o;
    return t;
  }
};
template <int NMAX>
struct zk_sep_meta {
  static constexpr typename zk_sep_pack<NMAX>::tables tab = zk_sep_pack<NMAX>::make();
};

#ifndef ZK_SEP_ROW
#define ZK_SEP_ROW 32  // doubles per row of the P-value tables (degrees 0..24 used)
#endif

struct zk_sep_row {     // one quadrant row pair (r, K-1-r) with at least one disk pixel
  int32_t r;            // row index
  int32_t cmin;         // first quadrant column inside the disk (columns cmin .. Q-1 are inside)
};

struct zk_sep_unit {    // batch kernel: 16 (float32) / 8 (float64) quadrant columns of one row pair
  int32_t run_off[4];   // byte offsets of the source runs inside a patch
  int32_t c0;           // first quadrant column of the unit
  int32_t cmin;         // first quadrant column of this ROW inside the disk
  int32_t r;            // row index (selects the y table row)
  int32_t row_end;      // bit 0: the row pair is complete after this unit; bit 1 (wide units): this unit
                        // belongs to the second row (K-1-r) of the pair; bits 8..: cmax = ceil(K/2)
};

#define ZK_STREAM_ROW(NMAX) (((NMAX) + 1) & ~1)  // doubles per row of the stream kernel's table (P_1 .. P_NMAX)
#define ZK_SPLIT_HALF 6  // strip kernel, n_max 10 / 12: doubles per half of a row of the parity-split x table (d_psplit)
#define ZK_STREAM_PAD 4  // zero rows either side of the stream kernel's Legendre table (one granule of pixels)

// Stream ("flat") batch kernel, zk_sep_stream.hip: a patch is read as the contiguous pixel stream it is
// in memory, one 128-B line per unit, whatever K is.
struct zk_stream_row {  // one patch row with disk pixels, in flat pixel indices t = r * K + c
  int32_t ts, te;       // first / last (inclusive) disk pixel of the row
  int32_t r;            // row index (selects the y table row); column of pixel t is t - r * K
  int32_t pad;
};

struct zk_stream_unit { // one 128-B line of the patch stream that holds at least one disk pixel
  int32_t byte_off;     // line offset inside the patch (multiple of 128)
  int32_t t0;           // flat index of the line's first pixel
  int32_t row_i;        // first entry of the row table with te >= t0
  int32_t clamp;        // 1: the line crosses the end of the patch (source offsets are clamped)
};

struct zk_sep_tables {
  int kernel_nmax = -1;
  int np_kernel = 0;
  int Q = 0;                       // quadrant side ceil(K/2)
  double* d_xq = nullptr;          // [Q][ZK_SEP_ROW] P_a(x_c) (x 0.5 on the centre column of odd K)
  double* d_yq = nullptr;          // [Q][ZK_SEP_ROW] P_b(y_r) (x 0.5 on the centre row)
  double* d_T = nullptr;           // class blocks [cls][j], each row packed to its a + b <= n_j entries (zk_sep_pack), scaled by 1/area
  int32_t* d_colmap = nullptr;     // [np_kernel] class-ordered Zernike slot -> output column or -1
  int n_rows = 0;
  zk_sep_row* d_rows = nullptr;    // [n_rows]
  int32_t* d_cmin = nullptr;       // [K] first quadrant column inside the disk of window row r (Q: none); strip kernel
  double* d_psplit_alloc = nullptr; // strip kernel, n_max 9-12: ZK_STREAM_PAD zero rows | [K][2 ZK_SPLIT_HALF] | pad: per column
  double* d_psplit = nullptr;       // [P_2 P_4 .. P_12 | P_1 P_3 .. P_11] (zero beyond the kernel's n_max)
  int32_t* d_strip_rows = nullptr; // [K + 1] strip kernel (even K): per frame row n1 | n2 << 8, the column pairs of its two sweeps
  int tile_pitch = 0;
  // batch kernel, one unit list per element type ([0] float32: K % 4 == 0, K >= 16; [1] float64: K even, K >= 8)
  struct batch_tables {
    int run = 0;                      // granules per source run: 8 (float32, K == 32 or wide) or 4
    int wide = 0;                     // 1: row-at-a-time units of one quadrant line (float32 K % 64 == 0, float64 K % 32 == 0)
                                      // 2: two-row units (float32 K == 32): 256 contiguous bytes per patch
    int n_units = 0;
    zk_sep_unit* d_units = nullptr;
    int n_row_starts = 0;             // units that begin a row pair (the unit order may rotate to any of them)
    int32_t* d_row_starts = nullptr;  // [n_row_starts] unit indices
  };
  batch_tables batch[2];
  // stream batch kernel: full-width Legendre table + row / line lists per element type
  double* d_pfull_alloc = nullptr;    // ZK_STREAM_PAD zero rows | [K][ZK_STREAM_ROW(kernel n_max)] | ZK_STREAM_PAD zero rows
  double* d_pfull = nullptr;          // row 0 of it: P_1(x_c) .. P_nmax(x_c) for column c (P_0 = 1 is implicit)
  struct stream_tables {
    int n_units = 0;
    zk_stream_unit* d_units = nullptr;
    int n_rows = 0;                   // without the sentinel entry that closes the table
    zk_stream_row* d_rows = nullptr;
    int aligned = 0;                  // patches start on 128-B lines (every request is a whole line)
    int preferred = 0;                // ZK_PATH_AUTO picks this kernel over the row-pair kernel
  };
  stream_tables stream[2];
};

// Timing-only ablation builds of the batch kernels (make ABLATE=n -> libzernike_hip_ablate<n>.so; outputs
// are wrong by construction): 1 = no arithmetic (DMA + LDS reads + stores), 2 = no DMA (arithmetic on
// stale LDS), 3 = no output stores.  cdna_hip_programming.md section 5.4 rule 17: stubbed values are kept live.
#ifndef ZK_ABLATE
#define ZK_ABLATE 0
#endif
// Cache policy of the streamed operands: aux = 2 is "nt" (non-temporal).  Every patch byte is read
// exactly once by one CU and every moment is written once, so both streams bypass cache retention:
// interleaved A/B on one device, median of 31 rounds (profiles/r01_ablation.txt):
//   default policy 3.365 ms | nt loads 3.136 | nt loads + nt stores 3.124 | nt stores only 3.288
#ifndef ZK_DMA_AUX
#define ZK_DMA_AUX 2
#endif
#ifndef ZK_STORE_NT
#define ZK_STORE_NT 1
#endif

#ifdef __HIPCC__
// Row sums of one row pair: S(a, parity of b) of zk_sep.h's header comment.
// MASK (bit = parity class ZK_EE .. ZK_OO) restricts the object to some classes: class EE only needs the
// fold ee and the sums SEp, OE: oe / SOp, EO: eo / SEm, OO: oo / SOm, and T is block-diagonal in the
// classes -- so the moments of a class can be computed by a pass of their own, with a quarter of the
// accumulators (n_max > 16, where the full set no longer fits a lane's registers).
template <int NMAX, int MASK = 15>
struct zk_sep_rows {
  using S = zk_sep_set<NMAX>;
  static constexpr bool kEE = MASK & (1 << ZK_EE), kOE = MASK & (1 << ZK_OE), kEO = MASK & (1 << ZK_EO),
                        kOO = MASK & (1 << ZK_OO);
  double SEp[S::NE], SEm[S::NE], SOp[S::NO > 0 ? S::NO : 1], SOm[S::NO > 0 ? S::NO : 1];

  __device__ __forceinline__ void clear_row() {
#pragma unroll
    for (int i = 0; i < S::NE; ++i) SEp[i] = SEm[i] = 0.0;
#pragma unroll
    for (int i = 0; i < S::NO; ++i) SOp[i] = SOm[i] = 0.0;
  }
  // one quadrant pixel: a=(r,c) b=(r,c') c=(r',c) d=(r',c'); px = P_*(x_c) row of the x table (a pointer into the
  // constant address space, or the row's values already in registers)
  template <typename PX>
  __device__ __forceinline__ void pixel(double a, double b, double c, double d, const PX& px) {
    const double s1 = a + b, d1 = a - b, s2 = c + d, d2 = c - d;
    const double ee = s1 + s2, oe = d1 + d2, eo = s1 - s2, oo = d1 - d2;
#pragma unroll
    for (int i = 0; i < S::NE; ++i) {
      if constexpr (kEE) SEp[i] = __builtin_fma(ee, px[2 * i], SEp[i]);
      if constexpr (kEO) SEm[i] = __builtin_fma(eo, px[2 * i], SEm[i]);
    }
#pragma unroll
    for (int i = 0; i < S::NO; ++i) {
      if constexpr (kOE) SOp[i] = __builtin_fma(oe, px[2 * i + 1], SOp[i]);
      if constexpr (kOO) SOm[i] = __builtin_fma(oo, px[2 * i + 1], SOm[i]);
    }
  }
  // Row-at-a-time form (zk_sep_patches.hip): one pixel a and its column mirror b of a SINGLE row.  The
  // first row of a pair accumulates into SEp / SOp, the second into SEm / SOm, and pair_combine() turns
  // them into the sum / difference the row_end() step expects:
  //   SEp <- X1 + X2 (even b),  SEm <- X1 - X2 (odd b),  same for SO.
  // (px: a table row in the constant address space, or its values already in registers)
  template <bool FIRST, typename PX>
  __device__ __forceinline__ void row_pixel(double a, double b, const PX& px) {
    const double s = a + b, d = a - b;
    if constexpr (kEE || kEO) {  // even x degrees (a wave of a pair kernel owns one x parity, zk_sep_patches.hip)
#pragma unroll
      for (int i = 0; i < S::NE; ++i) {
        if (FIRST) SEp[i] = __builtin_fma(s, px[2 * i], SEp[i]);
        else SEm[i] = __builtin_fma(s, px[2 * i], SEm[i]);
      }
    }
    if constexpr (kOE || kOO) {
#pragma unroll
      for (int i = 0; i < S::NO; ++i) {
        if (FIRST) SOp[i] = __builtin_fma(d, px[2 * i + 1], SOp[i]);
        else SOm[i] = __builtin_fma(d, px[2 * i + 1], SOm[i]);
      }
    }
  }
  // the same with P_0 = 1 implicit: p1[a - 1] = P_a(x_c), a = 1 .. NMAX (even patch sizes: no half-weight column)
  template <bool FIRST, typename PX>
  __device__ __forceinline__ void row_pixel_p0(double a, double b, const PX& p1) {
    const double s = a + b, d = a - b;
    if constexpr (kEE || kEO) {
      if (FIRST) SEp[0] += s;
      else SEm[0] += s;
#pragma unroll
      for (int i = 1; i < S::NE; ++i) {
        if (FIRST) SEp[i] = __builtin_fma(s, p1[2 * i - 1], SEp[i]);
        else SEm[i] = __builtin_fma(s, p1[2 * i - 1], SEm[i]);
      }
    }
    if constexpr (kOE || kOO) {
#pragma unroll
      for (int i = 0; i < S::NO; ++i) {
        if (FIRST) SOp[i] = __builtin_fma(d, p1[2 * i], SOp[i]);
        else SOm[i] = __builtin_fma(d, p1[2 * i], SOm[i]);
      }
    }
  }
  __device__ __forceinline__ void pair_combine() {
    if constexpr (kEE || kEO) {
#pragma unroll
      for (int i = 0; i < S::NE; ++i) {
        const double t = SEp[i];
        SEp[i] = t + SEm[i];
        SEm[i] = t - SEm[i];
      }
    }
    if constexpr (kOE || kOO) {
#pragma unroll
      for (int i = 0; i < S::NO; ++i) {
        const double t = SOp[i];
        SOp[i] = t + SOm[i];
        SOm[i] = t - SOm[i];
      }
    }
  }
};

template <int NMAX, int MASK = 15>
struct zk_sep_acc : zk_sep_rows<NMAX, MASK> {
  using S = zk_sep_set<NMAX>;
  using R = zk_sep_rows<NMAX, MASK>;
  double M[S::NP];

  __device__ __forceinline__ void clear_all() {
#pragma unroll
    for (int i = 0; i < S::NP; ++i) M[i] = 0.0;
    this->clear_row();
  }

  // finished row pair: py = P_*(y_r) row of the y table; `rs` holds the pair's row sums (this object's
  // own, or a second set when two row pairs are in flight).  Slots are template parameters so that
  // slot -> (a, b) folds at compile time and every register index is static.
  template <int s>
  __device__ __forceinline__ void slot_fma(const R& rs, const ZK_CONST double* py) {
    constexpr int a = S::slot_a(s), b = S::slot_b(s);
    if constexpr ((MASK >> S::cls_of(a, b)) & 1) {
      const double v = (a & 1) ? ((b & 1) ? rs.SOm[a >> 1] : rs.SOp[a >> 1])
                               : ((b & 1) ? rs.SEm[a >> 1] : rs.SEp[a >> 1]);
      M[s] = __builtin_fma(py[b], v, M[s]);
    }
  }
  template <int... Is>
  __device__ __forceinline__ void row_all(const R& rs, const ZK_CONST double* py, std::integer_sequence<int, Is...>) {
    (slot_fma<Is>(rs, py), ...);
  }
  __device__ __forceinline__ void row_end_from(R& rs, const ZK_CONST double* py) {
    row_all(rs, py, std::make_integer_sequence<int, S::NP>{});
    rs.clear_row();
  }
  __device__ __forceinline__ void row_end(const ZK_CONST double* py) { row_end_from(*this, py); }

  // Unfolded form (stream batch kernel): X[a] = sum_c f[r][c] P_a(x_c) over (part of) ONE row r;
  // M_(a,b) += P_b(y_r) X[a] for every slot.  Linear in X, so a row may be flushed in pieces.
  __device__ __forceinline__ void clear_moments() {
#pragma unroll
    for (int i = 0; i < S::NP; ++i) M[i] = 0.0;
  }
  // py = P_1(y_r) .. P_nmax(y_r) (the stream kernel's table has no P_0 column: P_0 = 1)
  template <int s, typename PY>
  __device__ __forceinline__ void stream_slot(const double (&X)[S::NA], const PY& py) {
    constexpr int a = S::slot_a(s), b = S::slot_b(s);  // (constexpr: otherwise evaluated at run time for large NMAX)
    if constexpr (((MASK >> S::cls_of(a, b)) & 1) != 0) {
      if constexpr (b == 0) M[s] += X[a];
      else M[s] = __builtin_fma(py[b - 1], X[a], M[s]);
    }
  }
  template <typename PY, int... Is>
  __device__ __forceinline__ void stream_all(const double (&X)[S::NA], const PY& py, std::integer_sequence<int, Is...>) {
    (stream_slot<Is>(X, py), ...);
  }
  // the same without clearing X (strip dense kernel: the sums run on to the next, wider row)
  template <typename PY>  // py: table row pointer, or the row's values already in registers
  __device__ __forceinline__ void stream_accumulate(const double (&X)[S::NA], const PY& py) {
    stream_all(X, py, std::make_integer_sequence<int, S::NP>{});
  }
  __device__ __forceinline__ void stream_row_end(double (&X)[S::NA], const ZK_CONST double* py) {
    stream_all(X, py, std::make_integer_sequence<int, S::NP>{});
#pragma unroll
    for (int a = 0; a < S::NA; ++a) X[a] = 0.0;
  }

  // Z (class-ordered Zernike slots) = T * M, one parity class at a time.  `emit(slot, value)` receives
  // each finished moment; `slot` is a std::integral_constant, so callers can use it both as an int
  // and (decltype(slot)::value) as a compile-time constant.  The T table holds the packed rows of
  // zk_sep_pack (entries with a + b <= n_j only).
  template <int CLS, int J, typename F>
  __device__ __forceinline__ void transform_row(const ZK_CONST double* tmat, F&& emit) {
    using P = zk_sep_meta<NMAX>;
    constexpr int n = S::cls_count(CLS), off = S::cls_begin(CLS);
    constexpr int rb = P::tab.row_begin[off + J], zn = P::tab.zn[off + J];
    const ZK_CONST double* tb = tmat + rb;
    double z = 0.0;
    int k = 0;  // position in the packed row; the loop is fully unrolled, so k and the test fold away
#pragma unroll
    for (int i = 0; i < n; ++i)
      if (P::tab.deg[off + i] <= zn) z = __builtin_fma(tb[k++], M[off + i], z);
    emit(std::integral_constant<int, off + J>{}, z);
    // keep the scheduler from hoisting every row's scalar loads to the top (hundreds of SGPRs)
    __builtin_amdgcn_sched_barrier(0);
  }
  // one moment by its class-ordered Zernike slot (zk_set<NMAX> order: [EE | OE | EO | OO])
  template <int SLOT>
  __device__ __forceinline__ double moment(const ZK_CONST double* tmat) {
    constexpr int cls = SLOT < S::cls_begin(1) ? 0 : SLOT < S::cls_begin(2) ? 1 : SLOT < S::cls_begin(3) ? 2 : 3;
    double z = 0.0;
    transform_row<cls, SLOT - S::cls_begin(cls)>(tmat, [&](auto, double v) { z = v; });
    return z;
  }
  template <int CLS, typename F, int... Js>
  __device__ __forceinline__ void transform_rows(const ZK_CONST double* tmat, F&& emit,
                                                 std::integer_sequence<int, Js...>) {
    (transform_row<CLS, Js>(tmat, emit), ...);
  }
  template <int CLS, typename F>
  __device__ __forceinline__ void transform_class(const ZK_CONST double* tmat, F&& emit) {
    transform_rows<CLS>(tmat, emit, std::make_integer_sequence<int, S::cls_count(CLS)>{});
  }
  template <typename F>
  __device__ __forceinline__ void transform(const ZK_CONST double* tmat, F&& emit) {
    if constexpr (R::kEE) transform_class<ZK_EE>(tmat, emit);
    if constexpr (R::kOE) transform_class<ZK_OE>(tmat, emit);
    if constexpr (R::kEO) transform_class<ZK_EO>(tmat, emit);
    if constexpr (R::kOO) transform_class<ZK_OO>(tmat, emit);
  }
};

// Z = T M for TWO accumulator sets at once (strip dense kernel: a lane's two outputs go to the same planes, one row
// apart): every entry of T is fetched once and feeds two FMAs.  emit(slot, z_a, z_b).
template <int NMAX, int MASK, int CLS, int J, typename F>
__device__ __forceinline__ void zk_sep_transform2_row(const zk_sep_acc<NMAX, MASK>& A, const zk_sep_acc<NMAX, MASK>& B,
                                                      const ZK_CONST double* tmat, F&& emit) {
  using S = zk_sep_set<NMAX>;
  using P = zk_sep_meta<NMAX>;
  constexpr int n = S::cls_count(CLS), off = S::cls_begin(CLS);
  constexpr int rb = P::tab.row_begin[off + J], zn = P::tab.zn[off + J];
  const ZK_CONST double* tb = tmat + rb;
  double za = 0.0, zb = 0.0;
  int k = 0;
#pragma unroll
  for (int i = 0; i < n; ++i)
    if (P::tab.deg[off + i] <= zn) {
      const double t = tb[k++];
      za = __builtin_fma(t, A.M[off + i], za);
      zb = __builtin_fma(t, B.M[off + i], zb);
    }
  emit(std::integral_constant<int, off + J>{}, za, zb);
  __builtin_amdgcn_sched_barrier(0);  // (as transform_row: keep the rows' scalar loads from piling up at the top)
}
template <int NMAX, int MASK, int CLS, typename F, int... Js>
__device__ __forceinline__ void zk_sep_transform2_rows(const zk_sep_acc<NMAX, MASK>& A, const zk_sep_acc<NMAX, MASK>& B,
                                                       const ZK_CONST double* tmat, F&& emit, std::integer_sequence<int, Js...>) {
  (zk_sep_transform2_row<NMAX, MASK, CLS, Js>(A, B, tmat, emit), ...);
}
template <int NMAX, int MASK, typename F>
__device__ __forceinline__ void zk_sep_transform2(const zk_sep_acc<NMAX, MASK>& A, const zk_sep_acc<NMAX, MASK>& B,
                                                  const ZK_CONST double* tmat, F&& emit) {
  using S = zk_sep_set<NMAX>;
  if constexpr ((MASK >> ZK_EE) & 1) zk_sep_transform2_rows<NMAX, MASK, ZK_EE>(A, B, tmat, emit, std::make_integer_sequence<int, S::cls_count(ZK_EE)>{});
  if constexpr ((MASK >> ZK_OE) & 1) zk_sep_transform2_rows<NMAX, MASK, ZK_OE>(A, B, tmat, emit, std::make_integer_sequence<int, S::cls_count(ZK_OE)>{});
  if constexpr ((MASK >> ZK_EO) & 1) zk_sep_transform2_rows<NMAX, MASK, ZK_EO>(A, B, tmat, emit, std::make_integer_sequence<int, S::cls_count(ZK_EO)>{});
  if constexpr ((MASK >> ZK_OO) & 1) zk_sep_transform2_rows<NMAX, MASK, ZK_OO>(A, B, tmat, emit, std::make_integer_sequence<int, S::cls_count(ZK_OO)>{});
}

template <typename F, int... Is>
__device__ __forceinline__ void zk_for_each_slot_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void zk_for_each_slot(F&& f) {
  zk_for_each_slot_impl(f, std::make_integer_sequence<int, N>{});
}

// Epilogue of the batch kernels: the wave's 64 x N_poly moments (lane = patch, z in class-ordered
// slots) become rows of the (N, n_poly) output through a 16-KiB LDS slab and leave as contiguous
// 16-B-per-lane non-temporal stores.  ppp = patches per pass (host: largest power of two with
// ppp * n_poly <= 2048); nv = live patches of this wave.
template <int NP>
__device__ __forceinline__ void zk_batch_store_rows(const double (&z)[NP], const ZK_CONST int32_t* cmap,
                                                    double* slab, double* obase, int lane, int nv, int n_poly,
                                                    int ppp) {
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  // The output rows of a wave start 16-B aligned unless the caller's matrix starts on an odd row of an odd n_poly
  // (a rank's block inside the gathered matrix: row 4 068 289 x 45 moments).  Then the staged image is shifted by one
  // double, so that LDS pairs and global pairs stay 16-B aligned together and only the two ends are scalar stores.
  const int mis = (int)(((uintptr_t)obase >> 3) & 1);  // wave-uniform (ppp * n_poly is even: every pass alike)
#pragma unroll 1
  for (int h = 0; h * ppp < 64; ++h) {
    if (lane / ppp == h) {
      double* const row = slab + mis + (lane % ppp) * n_poly;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int col = cmap[i];
        if (col >= 0) row[col] = z[i];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int live = nv - h * ppp;
    live = live < 0 ? 0 : (live > ppp ? ppp : live);
    const int vd = live * n_poly;  // doubles to write in this pass
    double* const dst = obase + (long long)h * ppp * n_poly;
    for (int k = lane; 2 * k < vd + mis; k += 64) {
      const f64x2 v = *(const f64x2*)(slab + 2 * k);
      const int e = 2 * k - mis;  // output element of v.x
#if ZK_ABLATE == 3
      asm volatile("" ::"v"(v));
#else
      if (e >= 0 && e + 2 <= vd) {
#if ZK_STORE_NT
        __builtin_nontemporal_store(v, (f64x2*)(dst + e));
#else
        *(f64x2*)(dst + e) = v;
#endif
      } else {
        if (e >= 0 && e < vd) dst[e] = v.x;
        if (e + 1 >= 0 && e + 1 < vd) dst[e + 1] = v.y;
      }
#endif
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // slab reads done before the next pass overwrites
  }
}

// The same for the pair kernels (zk_sep_patches.hip): the two waves of a pair hold the moments of the SAME 64 patches,
// each the columns of its x parity (MASK); both stage their columns into the pair's shared slab, then share the stores.
// Workgroup barriers order the two phases (every wave of the workgroup runs the same number of passes).
template <int NMAX, int MASK, int NP>
__device__ __forceinline__ void zk_batch_store_rows_pair(const double (&z)[NP], const ZK_CONST int32_t* cmap, double* slab,
                                                         double* obase, int lane, int part, int nv, int n_poly, int ppp) {
  using S = zk_sep_set<NMAX>;
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  const int mis = (int)(((uintptr_t)obase >> 3) & 1);
#pragma unroll 1
  for (int h = 0; h * ppp < 64; ++h) {
    if (lane / ppp == h) {
      double* const row = slab + mis + (lane % ppp) * n_poly;
      zk_for_each_slot<NP>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        constexpr int cls = i < S::cls_begin(1) ? 0 : i < S::cls_begin(2) ? 1 : i < S::cls_begin(3) ? 2 : 3;
        if constexpr ((MASK >> cls) & 1) {
          const int col = cmap[i];
          if (col >= 0) row[col] = z[i];
        }
      });
    }
    __syncthreads();  // both waves' columns are in the slab
    int live = nv - h * ppp;
    live = live < 0 ? 0 : (live > ppp ? ppp : live);
    const int vd = live * n_poly;
    double* const dst = obase + (long long)h * ppp * n_poly;
    for (int k = lane + 64 * part; 2 * k < vd + mis; k += 128) {
      const f64x2 v = *(const f64x2*)(slab + 2 * k);
      const int e = 2 * k - mis;
      if (e >= 0 && e + 2 <= vd) {
#if ZK_STORE_NT
        __builtin_nontemporal_store(v, (f64x2*)(dst + e));
#else
        *(f64x2*)(dst + e) = v;
#endif
      } else {
        if (e >= 0 && e < vd) dst[e] = v.x;
        if (e + 1 >= 0 && e + 1 < vd) dst[e + 1] = v.y;
      }
    }
    __syncthreads();  // slab reads done before the next pass overwrites
  }
}

// One row pair of the dense kernels: quadrant columns cmin..Q-1 of the LDS-resident window rows `top`
// (row r) and `bot` (row K-1-r).  Four (then two, then one) pixels per iteration through four running
// pointers, so the loop spends few integer adds on addressing (they cost v_fma_f64 issue slots) and the
// later pixels' LDS reads and scalar rows overlap the first pixel's FMAs.
#ifndef ZK_ROW_PAIR_UNROLL4
#define ZK_ROW_PAIR_UNROLL4 1  // (-1 % at (32, 8), -5 % at (64, 12) per dense frame)
#endif
// Software-pipelined form (default).  Scalar loads return out of order, so the only wait the ISA offers for them
// is lgkmcnt(0) -- which also waits for whatever was issued last.  Left to the compiler the loop became
// load - wait - a few FMAs - load - wait per pixel (two exposed scalar-cache latencies per pixel, VALU busy 61-80 % at
// two waves per SIMD, profiles/r02_sq_counters.txt).  Here every step first waits for ITS operands (fetched during
// the previous step), then puts the next pixel's four LDS reads and its Legendre row in flight, then does its
// arithmetic.  The last step prefetches one pixel past the row (LDS tile and table have the room).
#ifndef ZK_ROW_PAIR_PIPE
#define ZK_ROW_PAIR_PIPE 1
#endif
#define ZK_LGKM_WAIT()                     \
  __builtin_amdgcn_s_waitcnt(0xC07F);      \
  __builtin_amdgcn_sched_barrier(0)
template <int NMAX, int MASK>
__device__ __forceinline__ void zk_sep_row_pair(zk_sep_acc<NMAX, MASK>& acc, const double* __restrict__ top,
                                                const double* __restrict__ bot, int cmin, int Q, int K,
                                                const ZK_CONST double* px) {
  const double* tf = top + cmin;
  const double* tb = top + (K - 1 - cmin);
  const double* bf = bot + cmin;
  const double* bb = bot + (K - 1 - cmin);
  const ZK_CONST double* pr = px + cmin * ZK_SEP_ROW;
  int c = cmin;
  // (the n_max 11-12 instances are register-capped at 256 for two waves per SIMD and already spill: the extra live
  //  row costs them more than the waits did -- 2048^2 frame: (32, 12) 2.03 -> 2.50 ms, (64, 12) 5.17 -> 5.79; a
  //  scalar-only form -- this pixel's LDS reads and the next pixel's row requested together, one wait per pixel, no
  //  extra VGPR, 254 registers without spills -- lost too: 2.16 / 5.92 ms: the compiler's 4-pixel schedule stays)
  if constexpr (ZK_ROW_PAIR_PIPE && !(NMAX == 12 && MASK == 15)) {
  constexpr int NA = NMAX + 1;
  double an = tf[0], bn = tb[0], cn = bf[0], dn = bb[0], Pn[NA];
#pragma unroll
  for (int t = 0; t < NA; ++t) Pn[t] = pr[t];
#define ZK_PIPE_STEP(NEXT)                                                         \
  {                                                                                \
    ZK_LGKM_WAIT();                                                                \
    const double a_ = an, b_ = bn, c_ = cn, d_ = dn;                               \
    double Pc[NA];                                                                 \
    _Pragma("unroll") for (int t = 0; t < NA; ++t) Pc[t] = Pn[t];                  \
    an = tf[NEXT];                                                                 \
    bn = tb[-(NEXT)];                                                              \
    cn = bf[NEXT];                                                                 \
    dn = bb[-(NEXT)];                                                              \
    _Pragma("unroll") for (int t = 0; t < NA; ++t) Pn[t] = pr[(NEXT)*ZK_SEP_ROW + t]; \
    __builtin_amdgcn_sched_barrier(0);                                             \
    acc.pixel(a_, b_, c_, d_, Pc);                                                 \
  }
  for (; c + 3 < Q; c += 4) {
    ZK_PIPE_STEP(1)
    ZK_PIPE_STEP(2)
    ZK_PIPE_STEP(3)
    ZK_PIPE_STEP(4)
    tf += 4;
    tb -= 4;
    bf += 4;
    bb -= 4;
    pr += 4 * ZK_SEP_ROW;
  }
  for (; c < Q; ++c) {
    ZK_PIPE_STEP(1)
    tf += 1;
    tb -= 1;
    bf += 1;
    bb -= 1;
    pr += ZK_SEP_ROW;
  }
#undef ZK_PIPE_STEP
  ZK_LGKM_WAIT();  // the prefetch past the row has landed before anything reuses those registers
  return;
  }
#if ZK_ROW_PAIR_UNROLL4
  for (; c + 3 < Q; c += 4) {  // four pixels per pointer update (the integer adds cost v_fma_f64 issue slots)
    acc.pixel(tf[0], tb[0], bf[0], bb[0], pr);
    acc.pixel(tf[1], tb[-1], bf[1], bb[-1], pr + ZK_SEP_ROW);
    acc.pixel(tf[2], tb[-2], bf[2], bb[-2], pr + 2 * ZK_SEP_ROW);
    acc.pixel(tf[3], tb[-3], bf[3], bb[-3], pr + 3 * ZK_SEP_ROW);
    tf += 4;
    tb -= 4;
    bf += 4;
    bb -= 4;
    pr += 4 * ZK_SEP_ROW;
  }
#endif
  for (; c + 1 < Q; c += 2) {
    acc.pixel(tf[0], tb[0], bf[0], bb[0], pr);
    acc.pixel(tf[1], tb[-1], bf[1], bb[-1], pr + ZK_SEP_ROW);
    tf += 2;
    tb -= 2;
    bf += 2;
    bb -= 2;
    pr += 2 * ZK_SEP_ROW;
  }
  if (c < Q) acc.pixel(tf[0], tb[0], bf[0], bb[0], pr);
}
#endif
