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

  zk_stage_tile(tile, img, H, W, i0 - ea, k0 - ea, K + 7, tile_pitch);
  (void)tile_elems;
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

// ---------------------------------------------------------------------------------------------------------------
// Round 3: the same algorithm with the scalar bookkeeping removed (even K; odd K keeps the kernel above).
// profiles/r02_sq_counters.txt had the kernel above at SALU : VALU = 0.40 and VALU-busy 0.63: per group of three column
// pairs ~13 scalar ALU instructions (three 64-bit table pointers, counters, compares), six scalar loads, and one exposed
// s_waitcnt for all of them; per frame row two table look-ups, selects and pointer products.  Here
//   * a frame row is ONE record (host table, one s_load_dword): n1 = column pairs up to the narrower output's limit,
//     n2 = the further pairs up to the wider one's;
//   * the sweep runs centre-outwards over the stream kernel's full-width table (P_1 .. P_nmax of column Q + t, rows
//     ascending with the sweep, P_0 = 1 implicit: one s_load_dwordx16 per column pair at n_max 8), so that the second
//     sweep simply continues where the first stopped;
//   * a sweep of n pairs is a jump into straight-line code (a switch whose cases fall through, aligned at the sweep's
//     END): no counters, no compares, every table / LDS address an immediate off three bases set once per sweep;
//   * the code is software-pipelined by hand: block j waits for ITS operands (requested by block j + 1), requests those
//     of block j - 1 into the other register set (static: sets alternate with j), then does its 11 f64 operations.
// ---------------------------------------------------------------------------------------------------------------
#ifndef ZK_STRIP2
#define ZK_STRIP2 1
#endif

// Operand requests the compiler cannot move: the pipeline below relies on a block's operands being REQUESTED one block
// ahead.  Written as plain loads, LLVM sinks each request into the block that uses it (it is dead on the path that leaves
// the sweep early) and the pipeline collapses into load - wait - use per column.  As volatile asm the requests stay where
// they are written, and the kernel waits for them itself before the first use.
// The compiler takes an asm's outputs for written when the asm has issued; nothing tells it that the hardware writes them
// LATER.  Two things went wrong before the present form (the second one as a memory-aperture fault on the GPU):
//   * arithmetic on the requested values has no dependence on a bare s_waitcnt and was placed ABOVE it;
//   * a register set whose request is dead on some path (the one past the end of a sweep) was reused at once -- for a
//     re-load of kernel arguments -- and the late table row then overwrote the pointers.
// So the wait is an asm that takes the requested registers as read-write operands: every use depends on it, and the
// registers stay allocated from the request to the wait.  Requests take the CURRENT block's pixel operands the same way,
// which keeps the block's arithmetic behind them (they must issue early to be of use).  tools/check_async_requests.py
// re-checks the generated code (tests/test_isa_checks.py).
typedef double zk_v8d __attribute__((ext_vector_type(8)));
typedef double zk_v4d __attribute__((ext_vector_type(4)));
typedef double zk_v2d __attribute__((ext_vector_type(2)));
template <int N>
struct zk_sgpr_row;  // N doubles of one table row in SGPRs
#ifdef ZK_EXP_HALF_TABLE  // timing experiment only (wrong moments): half the scalar bytes per column pair, every value used twice
template <>
struct zk_sgpr_row<8> {
  zk_v4d v;
  template <int OFF>
  __device__ __forceinline__ void request(const ZK_CONST double* p, double& after) {
    asm volatile("s_load_dwordx8 %0, %2, %3" : "=s"(v), "+v"(after) : "s"(p), "n"(OFF));
  }
  __device__ __forceinline__ void wait(double& a, double& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v), "+v"(a), "+v"(b)); }
  __device__ __forceinline__ double operator[](int i) const { return v[i & 3]; }
};
#else
template <>
struct zk_sgpr_row<8> {
  zk_v8d v;
  template <int OFF>
  __device__ __forceinline__ void request(const ZK_CONST double* p, double& after) {
    asm volatile("s_load_dwordx16 %0, %2, %3" : "=s"(v), "+v"(after) : "s"(p), "n"(OFF));
  }
  __device__ __forceinline__ void wait(double& a, double& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v), "+v"(a), "+v"(b)); }
  __device__ __forceinline__ double operator[](int i) const { return v[i]; }
};
#endif
template <>
struct zk_sgpr_row<6> {
  zk_v4d a;
  zk_v2d b;
  template <int OFF>
  __device__ __forceinline__ void request(const ZK_CONST double* p, double& after) {
    asm volatile("s_load_dwordx8 %0, %2, %3" : "=s"(a), "+v"(after) : "s"(p), "n"(OFF));
    asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(b) : "s"(p), "n"(OFF + 32));
  }
  __device__ __forceinline__ void wait(double& x, double& y) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+v"(x), "+v"(y));
  }
  __device__ __forceinline__ double operator[](int i) const { return i < 4 ? a[i] : b[i - 4]; }
};
template <>
struct zk_sgpr_row<4> {
  zk_v4d a;
  template <int OFF>
  __device__ __forceinline__ void request(const ZK_CONST double* p, double& after) {
    asm volatile("s_load_dwordx8 %0, %2, %3" : "=s"(a), "+v"(after) : "s"(p), "n"(OFF));
  }
  __device__ __forceinline__ void wait(double& x, double& y) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+v"(x), "+v"(y)); }
  __device__ __forceinline__ double operator[](int i) const { return a[i]; }
};
template <int OFF>
__device__ __forceinline__ double zk_lds_request(unsigned addr, double& after) {
  double v;
  asm volatile("ds_read_b64 %0, %2 offset:%3" : "=v"(v), "+v"(after) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ unsigned zk_lds_addr(const double* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const double*)p;
}

// One pass over the staged tile for the parity classes of MASK: all four (n_max <= 8), or the two classes of one x parity
// (n_max 10 / 12, where two full accumulator sets do not fit a lane's registers -- 2 x 91 doubles at n_max 12 -- but two
// half sets do: the kernel then runs the x-even classes EE | EO and the x-odd classes OE | OO one after the other on the same
// tile; a sweep of the x-even pass needs only the sums f(q) + f(K-1-q) and the even-degree table entries, the x-odd pass the
// differences and the odd degrees, so the two passes together do the column work of ONE full pass).
//   XT: table of the sweep, XROW doubles per column; XN entries used: entry e is the factor of degree XDEG(e)
template <int NMAX, int QM, int MASK>
struct zk_strip_cfg {
  using S = zk_sep_set<NMAX>;
  static constexpr bool kAll = MASK == 15;
  static constexpr bool kEven = (MASK & ((1 << ZK_EE) | (1 << ZK_EO))) != 0, kOdd = (MASK & ((1 << ZK_OE) | (1 << ZK_OO))) != 0;
  // full table: P_1 .. P_NMAX (ZK_STREAM_ROW doubles); split table (zk_sep.hip, d_psplit): [P_2 P_4 ..|pad][P_1 P_3 ..|pad]
  static constexpr int XROW = kAll ? ZK_STREAM_ROW(NMAX) : 2 * ZK_SPLIT_HALF;
  static constexpr int XN = kAll ? NMAX : ZK_SPLIT_HALF;           // doubles requested per column
  static constexpr int XOFF = (!kAll && kOdd) ? ZK_SPLIT_HALF : 0;  // first double of this pass inside a table row
  // table entry of degree a (a >= 1) inside the requested run
  static constexpr int entry(int a) { return kAll ? a - 1 : (a & 1) ? (a - 1) / 2 : a / 2 - 1; }
  static constexpr bool has(int a) { return (a & 1) ? kOdd : kEven; }
};

template <int NMAX, int QM, int MASK>
__device__ __forceinline__ void zk_strip_pass(const double* __restrict__ tile, const ZK_CONST int32_t* rtab,
                                              const ZK_CONST double* px, const ZK_CONST double* py, const ZK_CONST double* tb,
                                              const ZK_CONST int32_t* cmap, double* __restrict__ dst, bool store, bool two, int K,
                                              int W, int tile_pitch, long long plane, int wave, int lane) {
  using S = zk_sep_set<NMAX>;
  using C = zk_strip_cfg<NMAX, QM, MASK>;
  constexpr int YROW = ZK_STREAM_ROW(NMAX);
  constexpr int XROW = C::XROW;
  const int Q = K / 2;
  zk_sep_acc<NMAX, MASK> acc0, acc1;  // moments of output rows i0 + 2 wave and i0 + 2 wave + 1 (only M is used)
  acc0.clear_moments();
  acc1.clear_moments();
  const double* __restrict__ mine = tile + (2 * wave) * tile_pitch + lane;  // window row 0 of output 0

  double X[S::NA];
  // n column pairs of frame row `row` starting at sweep index t0 (t = 0: the two centre columns), outwards
  auto sweep = [&](const double* __restrict__ row, int t0, int n) __attribute__((always_inline)) {
    if (n == 0) return;
    // block i (sweep index t0 + i) uses table row pB + i XROW and the pixels at LDS bytes La + 8 (QM-1-i) (left) and
    // Ra + 8 i (right): every offset an immediate
    const ZK_CONST double* pB = px + (Q + t0) * XROW + C::XOFF;
    const unsigned La = zk_lds_addr(row + (Q - 1 - t0) - (QM - 1));
    const unsigned Ra = zk_lds_addr(row + (Q + t0));
    zk_sgpr_row<C::XN> P0, P1 = {};
    double a0, b0, a1 = 0.0, b1 = 0.0;
    P0.template request<0>(pB, a1);
    a0 = zk_lds_request<(QM - 1) * 8>(La, a1);
    b0 = zk_lds_request<0>(Ra, b1);
    auto arith = [&](double a, double b, const zk_sgpr_row<C::XN>& P) __attribute__((always_inline)) {
      [[maybe_unused]] const double s_ = b + a;
      [[maybe_unused]] const double d_ = b - a;
      if constexpr (C::kEven) X[0] += s_;
#pragma unroll
      for (int i = 1; i < S::NA; ++i)
        if (C::has(i)) X[i] = __builtin_fma((i & 1) ? d_ : s_, P[C::entry(i)], X[i]);
    };
    // block I: wait for ITS operands (requested one block earlier), request those of block I + 1 into the other register
    // set (one column past the sweep at its end: the table and the tile row have the room), then the arithmetic
#define ZK_STRIP_BLOCK(I)                                                                       \
  {                                                                                             \
    if constexpr (((I)&1) != 0) {                                                               \
      P1.wait(a1, b1);                                                                          \
      if constexpr ((I) + 1 < QM) {                                                             \
        P0.template request<((I) + 1) * XROW * 8>(pB, a1);                                      \
        a0 = zk_lds_request<(QM - 2 - (I) >= 0 ? QM - 2 - (I) : 0) * 8>(La, a1);                \
        b0 = zk_lds_request<((I) + 1) * 8>(Ra, b1);                                             \
      }                                                                                         \
      arith(a1, b1, P1);                                                                        \
    } else {                                                                                    \
      P0.wait(a0, b0);                                                                          \
      if constexpr ((I) + 1 < QM) {                                                             \
        P1.template request<((I) + 1) * XROW * 8>(pB, a0);                                      \
        a1 = zk_lds_request<(QM - 2 - (I) >= 0 ? QM - 2 - (I) : 0) * 8>(La, a0);                \
        b1 = zk_lds_request<((I) + 1) * 8>(Ra, b0);                                             \
      }                                                                                         \
      arith(a0, b0, P0);                                                                        \
    }                                                                                           \
  }
#define ZK_STRIP_STEP(I) \
  ZK_STRIP_BLOCK(I)      \
  if (n <= (I) + 1) break;
    do {
      ZK_STRIP_STEP(0) ZK_STRIP_STEP(1) ZK_STRIP_STEP(2) ZK_STRIP_STEP(3) ZK_STRIP_STEP(4) ZK_STRIP_STEP(5)
      ZK_STRIP_STEP(6) ZK_STRIP_STEP(7) ZK_STRIP_STEP(8) ZK_STRIP_STEP(9) ZK_STRIP_STEP(10) ZK_STRIP_STEP(11)
      ZK_STRIP_STEP(12) ZK_STRIP_STEP(13) ZK_STRIP_STEP(14)
      if constexpr (QM > 16) {
        ZK_STRIP_STEP(15) ZK_STRIP_STEP(16) ZK_STRIP_STEP(17) ZK_STRIP_STEP(18) ZK_STRIP_STEP(19) ZK_STRIP_STEP(20)
        ZK_STRIP_STEP(21) ZK_STRIP_STEP(22) ZK_STRIP_STEP(23) ZK_STRIP_STEP(24) ZK_STRIP_STEP(25) ZK_STRIP_STEP(26)
        ZK_STRIP_STEP(27) ZK_STRIP_STEP(28) ZK_STRIP_STEP(29) ZK_STRIP_STEP(30)
      }
      ZK_STRIP_BLOCK(QM - 1)
    } while (0);
    // the request past the sweep has landed before anything else may live in the registers it was written to
    P0.wait(a0, b0);
    P1.wait(a1, b1);
#undef ZK_STRIP_STEP
#undef ZK_STRIP_BLOCK
  };

  // frame row fr (relative to output 0's window) is window row fr of output 0 and fr - 1 of output 1; in the upper half
  // of the window output 1 has the narrower row and is served first, in the lower half output 0 (see the kernel above)
  auto frame_row = [&](int fr, auto first_is_1) __attribute__((always_inline)) {
    constexpr bool F1 = decltype(first_is_1)::value;
    const int rec = rtab[fr];
    if (rec == 0) return;  // no disk pixel of either output in this frame row
    const int n1 = rec & 0xff, n2 = rec >> 8;
#pragma unroll
    for (int i = 0; i < S::NA; ++i) X[i] = 0.0;
    const double* __restrict__ row = mine + fr * tile_pitch;
    sweep(row, 0, n1);
    // (unconditional: with n1 = 0 -- the first output has no disk pixel in this frame row -- X is still zero and the
    //  moments do not change; a branch here would make the compiler keep two versions of 45 accumulators)
    if constexpr (F1) acc1.stream_accumulate(X, py + (n1 ? fr - 1 : fr) * YROW);
    else acc0.stream_accumulate(X, py + fr * YROW);
    sweep(row, n1, n2);
    if constexpr (F1) acc0.stream_accumulate(X, py + fr * YROW);
    else acc1.stream_accumulate(X, py + (fr - 1) * YROW);
  };
  for (int fr = 0; fr < Q; ++fr) frame_row(fr, std::true_type{});
  for (int fr = Q; fr <= K; ++fr) frame_row(fr, std::false_type{});

  // Z = T M of both outputs together (every T entry fetched once), both stores of a plane off one base
  if (!store) return;
  zk_sep_transform2<NMAX, MASK>(acc0, acc1, tb, [&](auto slot, double z0, double z1) {
    const int col = cmap[slot];
    if (col >= 0) {
      double* __restrict__ d = dst + col * plane;
      d[0] = z0;
      if (two) d[W] = z1;
    }
  });
}

// Tools-only build (-DZK_STRIP_TRACE, tools/strip_trace.py): every workgroup leaves its timeline -- the 100-MHz
// constant clock and the shader clock at its start, after staging, after the arithmetic + store issue, after its stores
// have drained, and where it ran (HW_ID, XCC_ID) -- in a buffer the tool hands in.  Not part of the product library.
#ifdef ZK_STRIP_TRACE
__device__ unsigned long long* zk_strip_trace_buf = nullptr;
extern "C" int zk_debug_strip_trace(unsigned long long* dev_buf) {
  ZK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(zk_strip_trace_buf), &dev_buf, sizeof(dev_buf)));
  return 0;
}
#define ZK_TRACE_STAMP(slot)                                                  \
  unsigned long long zk_rt_##slot = __builtin_amdgcn_s_memrealtime(), zk_ck_##slot = __builtin_amdgcn_s_memtime();
#define ZK_TRACE_RECORD()                                                                                        \
  {                                                                                                              \
    ZK_TRACE_STAMP(2)                                                                                            \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                             \
    ZK_TRACE_STAMP(3)                                                                                            \
    if (zk_strip_trace_buf && (threadIdx.x & 63) == 0) {                                                         \
      unsigned hw, xcc;                                                                                          \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                           \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                         \
      unsigned long long* r = zk_strip_trace_buf + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 10; \
      r[0] = zk_rt_0; r[1] = zk_rt_1; r[2] = zk_rt_2; r[3] = zk_rt_3;                                            \
      r[4] = zk_ck_0; r[5] = zk_ck_1; r[6] = zk_ck_2; r[7] = zk_ck_3;                                            \
      r[8] = hw; r[9] = xcc;                                                                                     \
    }                                                                                                            \
  }
#else
#define ZK_TRACE_STAMP(slot)
#define ZK_TRACE_RECORD()
#endif

template <int NMAX, typename T, int QM>
__global__ __launch_bounds__(256, 2) void zk_frame_strip2_kernel(
    const T* __restrict__ img, double* __restrict__ out, const int32_t* __restrict__ row_tab,
    const double* __restrict__ xtab, const double* __restrict__ pfull, const double* __restrict__ tmat,
    const int32_t* __restrict__ colmap, int K, int H, int W, int row0, int n_rows, int tile_pitch, long long plane) {
  extern __shared__ __attribute__((aligned(16))) double tile[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ea = K - 1 - (K - 1) / 2;
  const int i0 = row0 + blockIdx.y * 8;
  const int k0 = blockIdx.x * 64;

  ZK_TRACE_STAMP(0)
  zk_stage_tile(tile, img, H, W, i0 - ea, k0 - ea, K + 7, tile_pitch);
  __syncthreads();
  ZK_TRACE_STAMP(1)

  const int ok = k0 + lane;
  const int oi = i0 + 2 * wave;
  const bool store = ok < W && oi < row0 + n_rows;
  const bool two = oi + 1 < row0 + n_rows;  // wave-uniform
  double* __restrict__ dst = out + (long long)(oi - row0) * W + ok;
  if constexpr (NMAX <= 8) {
    zk_strip_pass<NMAX, QM, 15>(tile, zk_const(row_tab), zk_const(xtab), zk_const(pfull), zk_const(tmat), zk_const(colmap), dst, store,
                                two, K, W, tile_pitch, plane, wave, lane);
  } else {
    zk_strip_pass<NMAX, QM, (1 << ZK_EE) | (1 << ZK_EO)>(tile, zk_const(row_tab), zk_const(xtab), zk_const(pfull), zk_const(tmat),
                                                         zk_const(colmap), dst, store, two, K, W, tile_pitch, plane, wave, lane);
    zk_strip_pass<NMAX, QM, (1 << ZK_OE) | (1 << ZK_OO)>(tile, zk_const(row_tab), zk_const(xtab), zk_const(pfull), zk_const(tmat),
                                                         zk_const(colmap), dst, store, two, K, W, tile_pitch, plane, wave, lane);
  }
  ZK_TRACE_RECORD()
}

// ---------------------------------------------------------------------------------------------------------------
// Round 4: the x table in VGPR LANES (even K <= 32, n_max <= 8: configs[1] dense and configs[3]).
// profiles/r04_strip_trace.txt: the wave slots of zk_frame_strip2_kernel<8> are occupied 0.96-0.98 of the kernel's
// duration -- no dispatch gap -- and its 0.73 VALU-busy is the scalar data path: a sweep block needs 8 fresh SGPR doubles
// (64 B) per 11 f64 operations, eight waves of a CU ask for 5.8 B per clock where the scalar cache delivers ~5.6,
// i.e. the 0.7 FMA per clock a CU gets from once-used scalars.  Fetching half the table row (ZK_EXP_HALF_TABLE, wrong
// results, timing only) took the wave's life from 96.1 k to 85.0 k clocks.  So the sweeps here take NO scalar operand:
//   * the whole x table of the window -- P_1 .. P_nmax of the 16 column pairs, 128 doubles -- lives in 8 VGPR pairs, lane
//     e of every row of 16 lanes holding entry e (two columns x 8 degrees per register pair), loaded once per workgroup;
//     a block multiplies with `v_fmac_f64_dpp ... row_newbcast:e` (the one DPP control the FP64 pipe accepts: lane e of
//     the lane's own row of 16, measured at the plain instruction's rate, tools/micro_dpp64.hip);
//   * with no scalar-memory request in a sweep its LDS requests complete in order, so the pixel pairs run TWO blocks ahead
//     and a block waits with lgkmcnt(2) for its own pair only (scalar loads return out of order: with them in flight only
//     lgkmcnt(0) is safe, which is why the round-3 pipeline was one block deep);
//   * a register-resident table cannot be entered at a run-time column, so a frame row is ONE sweep over the columns of
//     the wider output: the sums are copied when they pass the narrower output's limit (block n1) and both row steps
//     follow the sweep -- one start-up and one drain per frame row instead of two of each.
// MEASURED (profiles/r04_strip3.txt, same box, steady state): scalar-memory instructions -63 %, the wave's life 97.3 k ->
// 102.6 k clocks, the shader clock 2.19 -> 2.28 GHz (less power), the kernel 0.738 -> 0.745 ms per 2048^2: a draw.  The DPP
// form reads three 64-bit VGPR operands where the scalar form reads two (the VGPR-operand v_fmac_f64 runs at 0.8 of the
// SGPR-operand one, tools/micro_dpp64.hip), the sweep carries a second branch per block, and VALU-busy stays at 0.73.
// Requesting the y rows and the next record by hand at the row's start (they would ride on the sweep) needs 33 more live
// SGPRs: 584 bytes of scratch; as ordinary loads placed before the sweep the compiler keeps them there without a spill, and the
// wave's life does not move (102.5 k clocks: the SIMD's other wave already covers those waits).  On a box that does not
// throttle (both kernels at 2.38 GHz) the round-3 kernel is 4 % ahead: 0.688 against 0.715 ms.
// Also built in round 4 and dropped: FOUR outputs per lane at one wave per SIMD (a frame row swept once for four outputs: ~2.9 k
// operations per output instead of ~4.3 k).  Four accumulator sets are 360 registers, and only the 256 architectural VGPRs can be
// VALU operands on gfx950 (the other 256 are AccVGPRs): the compiler parks half of the moments in AGPRs behind
// v_accvgpr_read / _write and spills 968 bytes.  Two outputs per lane is what n_max 8 allows in ONE pass.
// THREE outputs per lane in two parity passes (the form of n_max 9-12: a pass carries the 25 / 20 moments of one x parity, so
// three sets fit; ~3.2 k operations per output; x table in VGPR lanes, LDS requests three blocks ahead) was built too, gives the
// same bits, and runs 1.4x SLOWER (205 k clocks per 3 outputs against 97 k per 2): every pass reads the pixel pair of a block
// again, a pass's block is 5-6 f64 operations, and eight waves of a CU then ask the LDS for 64 clocks of reads per 48 clocks
// of arithmetic -- the sweeps become LDS-bound.  (With scalar table rows, one block ahead, the same form ran 1.7x slower.)
// The round-3 kernel is balanced at three limits at once: per round of its eight waves 88 clocks of FP64 issue, ~91 clocks
// of scalar-operand delivery (512 B at ~5.6 B per clock) and 64 clocks of LDS reads.  Kept as an opt-in (ZK_STRIP_V3=1 in the environment, parity-tested) and as the record of
// what bounds the round-3 kernel; ZK_PATH_AUTO stays on zk_frame_strip2_kernel.
// ---------------------------------------------------------------------------------------------------------------
#ifndef ZK_STRIP3
#define ZK_STRIP3 1
#endif

template <int E>  // acc += T[lane E of my row of 16] * x
__device__ __forceinline__ void zk_fmac_bcast(double& acc, double t, double x) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t), "v"(x), "n"(E));
}
template <int N>  // all but the newest N of this wave's LDS requests have landed; a and b are what the caller goes on to use
__device__ __forceinline__ void zk_lds_wait(double& a, double& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

template <int NMAX>
__device__ __forceinline__ void zk_strip3_pass(const double* __restrict__ tile, const ZK_CONST int32_t* rtab, const double* __restrict__ xt,
                                               const ZK_CONST double* py, const ZK_CONST double* tb, const ZK_CONST int32_t* cmap,
                                               double* __restrict__ dst, bool store, bool two, int K, int W, int tile_pitch,
                                               long long plane, int wave, int lane) {
  using S = zk_sep_set<NMAX>;
  constexpr int YROW = ZK_STREAM_ROW(NMAX), XROW = YROW, QM = 16;
  const int Q = K / 2;
  zk_sep_acc<NMAX> acc0, acc1;  // moments of output rows i0 + 2 wave and i0 + 2 wave + 1 (only M is used)
  acc0.clear_moments();
  acc1.clear_moments();
  // register pair c, lane e of a row of 16: P_{(e & 7) + 1} of column pair 2 c + (e >> 3) (sweep index: 0 = the centre columns)
  double T[8];
  {
    const int e = lane & 15, deg = e & 7;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int col = Q + 2 * c + (e >> 3);
      T[c] = (deg < NMAX && col < K) ? xt[col * XROW + deg] : 0.0;
    }
  }
  const double* __restrict__ mine = tile + (2 * wave) * tile_pitch + lane;  // window row 0 of output 0

  auto frame_row = [&](int fr, auto first_is_1) __attribute__((always_inline)) {
    constexpr bool F1 = decltype(first_is_1)::value;
    const int rec = rtab[fr];
    if (rec == 0) return;  // no disk pixel of either output in this frame row
    const int n1 = rec & 0xff, ntot = n1 + (rec >> 8);
    double X[S::NA], Xs[S::NA];
#pragma unroll
    for (int i = 0; i < S::NA; ++i) X[i] = Xs[i] = 0.0;
    const double* __restrict__ row = mine + fr * tile_pitch;
    // block t uses the pixels at LDS bytes La + 8 (QM-1-t) (left of the centre) and Ra + 8 t (right): immediates
    const unsigned La = zk_lds_addr(row + (Q - 1) - (QM - 1));
    const unsigned Ra = zk_lds_addr(row + Q);
    double A[3], B[3];
    A[2] = B[2] = 0.0;
    double dep = 0.0;
    A[0] = zk_lds_request<(QM - 1) * 8>(La, dep);
    B[0] = zk_lds_request<0>(Ra, dep);
    A[1] = zk_lds_request<(QM - 2) * 8>(La, dep);
    B[1] = zk_lds_request<8>(Ra, dep);
#define ZK_STRIP3_BLOCK(I)                                                                                   \
  {                                                                                                          \
    zk_lds_wait<((I) + 1 < QM ? 2 : 0)>(A[(I) % 3], B[(I) % 3]);                                             \
    if constexpr ((I) + 2 < QM) {                                                                            \
      A[((I) + 2) % 3] = zk_lds_request<(QM - 3 - (I) >= 0 ? QM - 3 - (I) : 0) * 8>(La, A[(I) % 3]);         \
      B[((I) + 2) % 3] = zk_lds_request<((I) + 2) * 8>(Ra, B[(I) % 3]);                                      \
    }                                                                                                        \
    const double s_ = B[(I) % 3] + A[(I) % 3], d_ = B[(I) % 3] - A[(I) % 3];                                 \
    X[0] += s_;                                                                                              \
    zk_for_each_slot<NMAX>([&](auto k) {                                                                     \
      constexpr int a = decltype(k)::value + 1; /* degree */                                                 \
      zk_fmac_bcast<8 * ((I) & 1) + a - 1>(X[a], T[(I) / 2], (a & 1) ? d_ : s_);                             \
    });                                                                                                      \
    if (n1 == (I) + 1) { /* (the empty volatile asm keeps this a branch: if-converted it is 18 selects in EVERY block) */ \
      asm volatile("");                                                                                      \
      _Pragma("unroll") for (int i = 0; i < S::NA; ++i) Xs[i] = X[i];                                        \
    }                                                                                                        \
  }
#define ZK_STRIP3_STEP(I) \
  ZK_STRIP3_BLOCK(I)      \
  if (ntot <= (I) + 1) break;
    do {
      ZK_STRIP3_STEP(0) ZK_STRIP3_STEP(1) ZK_STRIP3_STEP(2) ZK_STRIP3_STEP(3) ZK_STRIP3_STEP(4) ZK_STRIP3_STEP(5)
      ZK_STRIP3_STEP(6) ZK_STRIP3_STEP(7) ZK_STRIP3_STEP(8) ZK_STRIP3_STEP(9) ZK_STRIP3_STEP(10) ZK_STRIP3_STEP(11)
      ZK_STRIP3_STEP(12) ZK_STRIP3_STEP(13) ZK_STRIP3_STEP(14)
      ZK_STRIP3_BLOCK(15)
    } while (0);
#undef ZK_STRIP3_STEP
#undef ZK_STRIP3_BLOCK
    // the requests past the sweep's end have landed before anything else may live in their registers
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0]), "+v"(B[0]), "+v"(A[1]), "+v"(B[1]), "+v"(A[2]), "+v"(B[2]));
    // (n1 = 0 -- the first output has no disk pixel in this frame row -- leaves Xs zero: its moments do not change)
    if constexpr (F1) {
      acc1.stream_accumulate(Xs, py + (n1 ? fr - 1 : fr) * YROW);
      acc0.stream_accumulate(X, py + fr * YROW);
    } else {
      acc0.stream_accumulate(Xs, py + fr * YROW);
      acc1.stream_accumulate(X, py + (fr - 1) * YROW);
    }
  };
  for (int fr = 0; fr < Q; ++fr) frame_row(fr, std::true_type{});
  for (int fr = Q; fr <= K; ++fr) frame_row(fr, std::false_type{});

  if (!store) return;
  zk_sep_transform2<NMAX, 15>(acc0, acc1, tb, [&](auto slot, double z0, double z1) {
    const int col = cmap[slot];
    if (col >= 0) {
      double* __restrict__ d = dst + col * plane;
      d[0] = z0;
      if (two) d[W] = z1;
    }
  });
}

template <int NMAX, typename T>
__global__ __launch_bounds__(256, 2) void zk_frame_strip3_kernel(
    const T* __restrict__ img, double* __restrict__ out, const int32_t* __restrict__ row_tab, const double* __restrict__ xtab,
    const double* __restrict__ pfull, const double* __restrict__ tmat, const int32_t* __restrict__ colmap, int K, int H, int W,
    int row0, int n_rows, int tile_pitch, long long plane) {
  extern __shared__ __attribute__((aligned(16))) double tile[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ea = K - 1 - (K - 1) / 2;
  const int i0 = row0 + blockIdx.y * 8;
  const int k0 = blockIdx.x * 64;
  ZK_TRACE_STAMP(0)
  zk_stage_tile(tile, img, H, W, i0 - ea, k0 - ea, K + 7, tile_pitch);
  __syncthreads();
  ZK_TRACE_STAMP(1)
  const int ok = k0 + lane;
  const int oi = i0 + 2 * wave;
  const bool store = ok < W && oi < row0 + n_rows;
  const bool two = oi + 1 < row0 + n_rows;  // wave-uniform
  double* __restrict__ dst = out + (long long)(oi - row0) * W + ok;
  zk_strip3_pass<NMAX>(tile, zk_const(row_tab), xtab, zk_const(pfull), zk_const(tmat), zk_const(colmap), dst, store, two, K, W,
                       tile_pitch, plane, wave, lane);
  ZK_TRACE_RECORD()
}

// Also tried in round 3: requesting a sweep's FIRST operands ahead of the row step that precedes it (each of the 66 sweeps of an
// output pair starts with an exposed scalar + LDS round trip).  The request then has to stay in flight across the row step's
// 45 FMAs, and the compiler, short of SGPRs there, COPIES the not-yet-written registers of the request to other registers
// (tools/check_async_requests.py: 1 516 hazards; across the loop's back edge it even wants the 16-SGPR tuple in vector
// registers: "illegal VGPR to SGPR copy").  An in-flight request is only safe over straight-line code without register
// pressure, which is what the one-block-ahead pipeline inside a sweep is.
// Tried in round 3 and not kept: a PERSISTENT form (two workgroups per CU walking the tiles, the next tile's raw rows copied by
// the LDS-DMA engine under the current tile's arithmetic, widened to float64 at the tile switch; the DMA wait placed before the
// tile's stores, bare s_barrier instead of __syncthreads() -- whose fence waits for the 90 outstanding stores per lane -- and
// the DMA issued from inline asm so that the compiler does not put vmcnt(0) in front of the LDS reads).  Per 2048^2, one tile
// per workgroup -> persistent: (32, 4) 0.35 -> 0.53 ms, (32, 8) 0.84 -> 0.95, (32, 12) 1.64 -> 1.70; the same kernel launched
// with one workgroup per tile (no persistence, DMA staging only) is level with the one-tile form (0.88).  What the hardware's
// own dispatch gives -- workgroups of a CU drifting out of phase, so that one's staging falls under the other's arithmetic,
// and tiles handed to whichever CU is free -- is worth more than the hidden round trips; two equal workgroups that start
// together stay in lock step and stage together.
template <int NMAX, typename T, int QM>
int launch_strip2(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out, hipStream_t s) {
  const zk_sep_tables* t = p->sep;
  const size_t lds = (size_t)(p->size + 7) * t->tile_pitch * sizeof(double);
  auto kern = zk_frame_strip2_kernel<NMAX, T, QM>;
  if (lds > 64 * 1024)
    ZK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long plane = zk_out_plane(p, n_rows, W);
  return zk_for_row_bands(row0, n_rows, W, 8, [&](int64_t r0, int64_t nr, long long off) {
    dim3 grid((unsigned)((W + 63) / 64), (unsigned)((nr + 7) / 8));
    int rc = zk_prof_begin(p, s);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const T*)in, out + off, t->d_strip_rows, NMAX <= 8 ? t->d_pfull : t->d_psplit,
                       t->d_pfull, t->d_T, t->d_colmap, p->size, (int)H, (int)W, (int)r0, (int)nr, t->tile_pitch, plane);
    ZK_HIP(hipGetLastError());
    return zk_prof_end(p, s);
  });
}

static long long g_strip3_launches = 0;  // (tests: the opt-in kernel really ran -- its moments equal the default kernel's bit for bit)

template <int NMAX, typename T>
int launch_strip3(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out, hipStream_t s) {
  const zk_sep_tables* t = p->sep;
  ++g_strip3_launches;
  const size_t lds = (size_t)(p->size + 7) * t->tile_pitch * sizeof(double);
  auto kern = zk_frame_strip3_kernel<NMAX, T>;
  if (lds > 64 * 1024)
    ZK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long plane = zk_out_plane(p, n_rows, W);
  return zk_for_row_bands(row0, n_rows, W, 8, [&](int64_t r0, int64_t nr, long long off) {
    dim3 grid((unsigned)((W + 63) / 64), (unsigned)((nr + 7) / 8));
    int rc = zk_prof_begin(p, s);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const T*)in, out + off, t->d_strip_rows, t->d_pfull, t->d_pfull, t->d_T,
                       t->d_colmap, p->size, (int)H, (int)W, (int)r0, (int)nr, t->tile_pitch, plane);
    ZK_HIP(hipGetLastError());
    return zk_prof_end(p, s);
  });
}

template <int NMAX, typename T>
int launch_one(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out,
               hipStream_t s) {
  const zk_sep_tables* t = p->sep;
  static const bool v1 = getenv("ZK_STRIP_V1") != nullptr;  // A/B runs: the round-2 kernel
  const bool v3 = getenv("ZK_STRIP_V3") != nullptr;  // opt-in: round 4's kernel with the x table in VGPR lanes (see above)
  if constexpr (NMAX <= 8) {
    if (ZK_STRIP3 && !v1 && v3 && t->d_strip_rows && p->size % 2 == 0 && p->size <= 32)
      return launch_strip3<NMAX, T>(p, in, H, W, row0, n_rows, out, s);
  }
  if (ZK_STRIP2 && !v1 && t->d_strip_rows && p->size % 2 == 0) {
    if (p->size <= 32) return launch_strip2<NMAX, T, 16>(p, in, H, W, row0, n_rows, out, s);
    return launch_strip2<NMAX, T, 32>(p, in, H, W, row0, n_rows, out, s);
  }
  if constexpr (NMAX > 8) {
    return zk_fail(ZK_E_BADARG, "internal: the two-pass strip kernel needs its tables");
  } else {
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
}

template <typename T>
int launch_t(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out,
             hipStream_t s) {
  switch (p->sep->kernel_nmax) {
    case 4: return launch_one<4, T>(p, in, H, W, row0, n_rows, out, s);
    case 6: return launch_one<6, T>(p, in, H, W, row0, n_rows, out, s);
    case 8: return launch_one<8, T>(p, in, H, W, row0, n_rows, out, s);
    case 10: return launch_one<10, T>(p, in, H, W, row0, n_rows, out, s);
    case 12: return launch_one<12, T>(p, in, H, W, row0, n_rows, out, s);
  }
  return zk_fail(ZK_E_BADARG, "no strip frame kernel for this n_max");
}

}  // namespace

extern "C" long long zk_debug_strip3_launches(void) { return g_strip3_launches; }

bool zk_sep_strip_available(const zk_plan* p, int dtype) {
  (void)dtype;
  if (!zk_sep_frame_available(p, dtype) || p->sep->kernel_nmax > 12 || !p->sep->d_pfull || !p->sep->d_cmin) return false;
  // n_max 9 - 12: only the two-pass form exists (even window sizes, zk_sep.hip builds its tables)
  static const bool no12 = getenv("ZK_STRIP_NO_SPLIT") != nullptr;  // A/B runs: the one-output kernel
  if (p->sep->kernel_nmax > 8 && (no12 || !p->sep->d_psplit || !p->sep->d_strip_rows || p->size % 2 != 0)) return false;
  // two workgroups per CU must fit (two waves per SIMD is what the kernel is compiled for): K <= 65.  Beyond,
  // the one-output kernel is ahead again (72 px: 4.7 vs 5.0 ms per 2048^2).
  return (size_t)(p->size + 7) * p->sep->tile_pitch * sizeof(double) <= 80 * 1024;
}

int zk_launch_sep_strip(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0, int64_t n_rows,
                        double* out, hipStream_t s) {
  if (dtype == ZK_F32) return launch_t<float>(p, in, H, W, row0, n_rows, out, s);
  return launch_t<double>(p, in, H, W, row0, n_rows, out, s);
}
