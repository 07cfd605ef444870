// zk_sep_patches.hip -- batch-of-patches Zernike moments (reference _zps.py:146-157): HBM-streaming,
// LDS-DMA transposed, row-separable arithmetic.  float32 patches of any size K >= 16 and float64 patches
// of any size K >= 8, odd sizes included (LDS-DMA sources only need element alignment); n_max <= 24
// (17-24: one launch per mirror-parity class, see MASK below).
// (Large batches of sizes that get no whole-line units here go to zk_sep_stream.hip under ZK_PATH_AUTO.)
//
// Work decomposition.  One wave owns 64 consecutive patches, one patch per lane, and keeps that
// patch's accumulators in VGPRs for the whole patch, so every multiplier that is not a pixel is
// wave-uniform and arrives as an SGPR pair from a scalar load (no LDS or VGPR traffic for tables, no
// cross-lane reduction).  HBM holds a patch contiguously while a lane needs "pixel t of my patch";
// that transposition is done by the LDS-DMA engine:
//
//   unit    = 64 B of quadrant columns (16 float32 / 8 float64) of one row pair (r, K-1-r) and their
//             column mirrors, for all 64 patches = 16 KiB.
//   stage   = 16 global_load_lds_dwordx4; each instruction moves whole 128-B rows of 8 patches (RUN=8:
//             float32, K=32) or 64-B runs of 16 patches (RUN=4), so HBM and the TCP see full-line
//             requests (measured: 6.1 TB/s for this access pattern alone, profiles/r01_micro_sfma.txt).
//             The LDS image is lane-linear per instruction; the granule a lane fetches is rotated by
//             its patch index so that the later per-lane ds_read_b128 (lane = patch) is
//             bank-conflict-free.
//   consume = the unit is pulled into VGPRs in two halves (8 ds_read_b128 each, the outer half only if
//             it holds disk pixels); once the inner half is in registers the wave re-arms its LDS slab
//             with the DMA of its next unit and computes while that DMA is in flight.
//
// Arithmetic (zk_sep.h): per quadrant pixel 4 v_cvt + 8 v_add_f64 (mirror folds, exact for float32)
// + 2(n_max+1) v_fma_f64; per row pair N_poly v_fma_f64; one class-blocked T product per patch.
// The scalar tables (Legendre values, T) total ~7 KiB at (32, 8) and stay in the scalar data cache --
// the direct folded sum streamed 76 KiB per wave through it and ran scalar-latency-bound
// (profiles/r01_patch_fold_pmc.txt).  Compute is fully hidden behind the stream
// (profiles/r01_ablation.txt): the kernel is bound by mixed read/write HBM traffic.
//
// No workgroup barrier exists in the kernel: a wave only ever reads LDS bytes it DMA'd itself (ordered
// by its own s_waitcnt vmcnt(0)); the waves of a workgroup only share the allocation.  Rows fully
// outside the unit disk (rows 0 and 31 at K=32) are never fetched.
//
// Epilogue: moments go through the same LDS slab to become rows of the (N, N_poly) output and are
// stored as contiguous 16-B-per-lane non-temporal runs.
//
// Roofline: algorithmic bytes K*K*s + 8*N_poly per patch (4 456 B at float32 (32, 8)); HBM-bound.
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

// Unit-order rotation.  Waves start together and walk the same unit list, so at any instant nearly all
// 2048 resident waves would be fetching the SAME row pair of their patches: addresses that agree in bits
// 7..11, i.e. a fraction of the HBM channels at a time (the kernel time then moves by ~9 % with the
// buffers' placement, tools/placement.py).  Starting every wave at a different unit spreads the rows in
// flight over the whole patch period.  A wave starts at the first unit of some row pair, so the units
// of a row stay consecutive.
#ifndef ZK_ROTATE
#define ZK_ROTATE 1
#endif

namespace {

#define ZK_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define ZK_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

#ifndef ZK_BATCH_2W
#define ZK_BATCH_2W 10  // largest kernel n_max compiled for two waves per SIMD
#endif
// LDS slabs (units in flight) per wave.  Kernels that fit two waves per SIMD run 8 waves per CU with one 16-KiB
// slab each (128 KiB): the other wave of the SIMD computes while this one waits for its DMA.  Above
// ZK_BATCH_2W a SIMD holds ONE wave, which with one slab alternates between waiting for a unit (HBM latency,
// ~2-3 us under load) and consuming it (~1 us at n_max 12) -- 4.8 TB/s at (64, 12).  Two slabs per wave (4 waves x
// 32 KiB = 128 KiB) keep the next unit's DMA in flight during the whole of the current unit's arithmetic.
#ifndef ZK_BATCH_DEPTH_1W
#define ZK_BATCH_DEPTH_1W 2
#endif
#ifndef ZK_BATCH_PIPE
#define ZK_BATCH_PIPE 1  // wide units: software-pipelined scalar loads (see quarter())
#endif
#define ZK_BATCH_DEPTH(NMAX, MASK) (((NMAX) <= ZK_BATCH_2W || (MASK) != 15) ? 1 : ZK_BATCH_DEPTH_1W)

// MASK: the parity classes this launch computes -- all of them (15), or one class per launch for
// n_max > 16 (zk_sep.h); a class pass writes its moments as planes of a scratch matrix [column][patch]
// (coalesced over the lanes), which zk_transpose_kernel turns into the (N, n_poly) rows afterwards.
// PAIR: two waves share one group of 64 patches and one pair of LDS slabs; each keeps the accumulators of ONE x parity
// (MASK = EE|EO or OE|OO: T is block-diagonal in the parity classes, so the halves never meet before the output row).
// The instances whose full accumulator set needs 300-460 registers (n_max 11-16: one wave per SIMD, 28-60 % of the
// HBM peak) fit two waves per SIMD this way: the DMA of a unit is issued half by each wave, every byte still crosses
// HBM once, the LDS slab is read by both.  Two workgroup barriers per unit order landing -> reading -> re-arming.
template <int NMAX, int RUN, typename TIN, bool WIDE, int MASK, bool PAIR>
__device__ __forceinline__ void zk_patch_body(
    const TIN* __restrict__ in, double* __restrict__ out, const zk_sep_unit* __restrict__ units,
    const double* __restrict__ xq, const double* __restrict__ tmat, const int32_t* __restrict__ colmap,
    int n_units, int n_poly, long long n_patches, int patch_bytes, int ppp, const int32_t* __restrict__ row_starts,
    int n_row_starts, const double* __restrict__ pfull, float* const wl, const long long group, const int part) {
  using S = zk_sep_set<NMAX>;
  constexpr int PXG = 16 / sizeof(TIN);  // pixels per 16-B granule: 4 (float32) or 2 (float64)
  typedef TIN gran_t __attribute__((ext_vector_type(PXG)));
  constexpr int NRUN = 16 / RUN;        // source runs per unit: 2 lines (RUN=8) or 4 half-lines
  constexpr int PPI = 64 / RUN;         // patches covered by one DMA instruction
  constexpr int SH = RUN == 8 ? 1 : 2;  // rotation = patch >> SH makes ds_read_b128 conflict-free
  // nt only where a run is a whole 128-B line: 64-B runs share their line with another unit of the same
  // row, which must still find it in L2 (default policy: 3.70 -> 2.14 ms at K=16, 8.9 -> 6.0 ms at K=48)
  constexpr int DMA_AUX = RUN == 8 ? ZK_DMA_AUX : 0;
  constexpr int DEPTH = PAIR ? 2 : ZK_BATCH_DEPTH(NMAX, MASK);  // slabs of 16 KiB (per wave, or per pair)
  constexpr int NDMA = PAIR ? 8 : 16;                           // DMA instructions this wave issues per unit

  const int lane = threadIdx.x & 63;
  const long long patch0 = group * 64;
  // single-wave form: a wave without patches leaves (the kernel has no barrier).  Pair form: every wave of the
  // workgroup runs the whole loop (barriers); a pair without patches reads patch 0 of the batch and stores nothing.
  if (!PAIR && patch0 >= n_patches) return;
  const long long left = n_patches - patch0;
  const int nv = left <= 0 ? 0 : (left < 64 ? (int)left : 64);  // live patches of this wave / pair

  // ---- DMA addressing: lane -> (patch-in-group a, slot b) ---------------------------------------
  const int a = lane / RUN, b = lane % RUN;
  const int g0 = (b - (a >> SH)) & (RUN - 1);  // source granule for patch group 0
  const char* const wbase = (const char*)in + (nv > 0 ? patch0 : 0) * patch_bytes;
  int poff[RUN];  // per patch group: byte offset of this lane's patch (+ its rotated granule)
#pragma unroll
  for (int pg = 0; pg < RUN; ++pg) {
    int pi = pg * PPI + a;
    pi = pi < nv ? pi : (nv > 0 ? nv - 1 : 0);  // tail wave: re-read the last live patch
    // rot(patch) = (pg*PPI + a) >> SH = 4*pg + (a >> SH) for both RUN values
    const int g = (g0 - 4 * pg) & (RUN - 1);
    poff[pg] = pi * patch_bytes + g * 16;
  }
  const ZK_CONST int32_t* utab = zk_const((const int32_t*)units);  // 8 ints per unit
  auto issue = [&](int u, float* slab) {  // one unit into one slab: 16 DMA instructions (a pair: 8 from each wave)
#pragma unroll
    for (int rho = 0; rho < NRUN; ++rho) {
      const int ro = utab[8 * u + rho];
#pragma unroll
      for (int pg = 0; pg < RUN; ++pg) {
        if (PAIR && (pg < RUN / 2) != (part == 0)) continue;  // wave-uniform: this wave's half of the patch groups
        __builtin_amdgcn_global_load_lds(ZK_GLOBAL_PTR(wbase + (poff[pg] + ro)),
                                         ZK_LDS_PTR(slab + (rho * RUN + pg) * 256), 16, 0, DMA_AUX);
      }
    }
  };

  // ---- LDS read addressing: lane = patch ----------------------------------------------------------
  int rd[RUN];  // float index of granule g of this lane's patch inside a run image
#pragma unroll
  for (int g = 0; g < RUN; ++g) rd[g] = (lane * RUN + ((g + (lane >> SH)) & (RUN - 1))) * 4;

  zk_sep_acc<NMAX, MASK> acc;
  acc.clear_all();
  const ZK_CONST double* px = zk_const(xq);
  [[maybe_unused]] const ZK_CONST double* p1 = zk_const(pfull);

  // first unit of this wave (see ZK_ROTATE): the start of some row pair; the loop visits off, off+1, ...,
  // wrapping around, so the units of a row stay consecutive
  const int off = n_row_starts > 0 ? zk_const(row_starts)[(int)((group * 7) % n_row_starts)] : 0;
  auto unit_at = [&](int k) { return k + off < n_units ? k + off : k + off - n_units; };
#if ZK_ABLATE != 2
  issue(unit_at(0), wl);
  if constexpr (DEPTH == 2)
    if (n_units > 1) issue(unit_at(1), wl + 4096);
#endif
  for (int k = 0; k < n_units; ++k) {
    const int u = unit_at(k);
    const int c0 = utab[8 * u + 4], cmin = utab[8 * u + 5], r = utab[8 * u + 6];
    const int rend = utab[8 * u + 7] & 1, cmax = utab[8 * u + 7] >> 8;
    float* const ws = DEPTH == 2 ? wl + (k & 1) * 4096 : wl;  // wave-uniform: the slab unit k landed in
    // this wave's DMA of unit u has landed (loads complete in order: with a second unit in flight, all but its
    // 16 instructions)
    if (DEPTH == 2 && k + 1 < n_units) {
      if constexpr (NDMA == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (PAIR) __syncthreads();  // the partner's half of the unit has landed too
    // The unit is consumed in two halves of 8 quadrant pixels so that only 32 staging VGPRs are live:
    // the outer half first (quadrant columns c0..c0+7, often entirely outside the disk and then not even
    // read), then the inner half; the slab is re-armed as soon as the inner half is in registers.
    auto lds_granule = [&](int rho, int g) -> gran_t {
      return *(const gran_t*)(ws + rho * 64 * RUN * 4 + rd[g]);
    };
    auto half = [&](int q0, bool rearm) {  // quadrant granules q0, q0+1 of both rows and their mirrors
      gran_t A[2], B[2], C[2], D[2];      // A: (r, q)  B: (r, mirror of q)  C, D: same for row K-1-r
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = q0 + i;
        if constexpr (RUN == 8) {  // run 0 = row r (cols 0..31), run 1 = row K-1-r
          A[i] = lds_granule(0, q);
          B[i] = lds_granule(0, 7 - q);
          C[i] = lds_granule(1, q);
          D[i] = lds_granule(1, 7 - q);
        } else {  // runs: (r, left 16), (r, mirrored right 16), (r', left), (r', right)
          A[i] = lds_granule(0, q);
          B[i] = lds_granule(1, 3 - q);
          C[i] = lds_granule(2, q);
          D[i] = lds_granule(3, 3 - q);
        }
      }
      if (rearm) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // unit is in VGPRs: the slab may be re-armed
        if constexpr (PAIR) __syncthreads();                  // ... once the partner has read it too
#if ZK_ABLATE != 2
        if (k + DEPTH < n_units) issue(unit_at(k + DEPTH), ws);
#endif
      }
#if ZK_ABLATE == 1
#pragma unroll
      for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(A[i]), "v"(B[i]), "v"(C[i]), "v"(D[i]));
      if (false)
#endif
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < PXG; ++e) {
          const int c = c0 + PXG * (q0 + i) + e;
          if (c >= cmin && c < cmax)  // wave-uniform: quadrant pixel inside the disk
            acc.pixel((double)A[i][e], (double)B[i][PXG - 1 - e], (double)C[i][e], (double)D[i][PXG - 1 - e],
                      px + c * ZK_SEP_ROW);
        }
      }
    };
    // wide units: 8 quadrant columns (2 granules of the line + 2 of its mirror line) of a single row
    auto quarter = [&](int q0, bool rearm, auto first_row) {
      gran_t A[2], B[2];
#if ZK_BATCH_PIPE
      // The Legendre row of pixel c+1 is fetched (scalar loads) while pixel c is being accumulated: with one
      // wave per SIMD nothing else hides the scalar-cache latency, and a load - wait - use sequence per pixel cost
      // 40 % of the wave's time (profiles/r02_sq_counters.txt).  The pixels of a quarter run branch-free
      // (columns left of the disk contribute zeros) so that the loads can be placed ahead of their use.
      // Table: the stream kernel's P_1 .. P_NMAX rows (P_0 = 1 needs no operand; wide units exist for even patch
      // sizes only, where no column carries a half weight) -- at n_max 12 a row is 24 dwords, two aligned loads.
      constexpr int SROW = ZK_STREAM_ROW(NMAX);
      double Pn[NMAX];
      const ZK_CONST double* prow = p1 + (c0 + PXG * q0) * SROW;
#pragma unroll
      for (int t = 0; t < NMAX; ++t) Pn[t] = prow[t];
#endif
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        A[i] = lds_granule(0, q0 + i);
        B[i] = lds_granule(NRUN - 1, 7 - (q0 + i));
      }
      if (rearm) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (PAIR) __syncthreads();
#if ZK_ABLATE != 2
        if (k + DEPTH < n_units) issue(unit_at(k + DEPTH), ws);
#endif
      }
#if ZK_ABLATE == 1
#pragma unroll
      for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(A[i]), "v"(B[i]));
      if (false)
#endif
#if ZK_BATCH_PIPE
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < PXG; ++e) {
          const int c = c0 + PXG * (q0 + i) + e;
          // scalar loads return out of order, so the only wait there is, lgkmcnt(0), also waits for whatever was
          // issued last: wait for this pixel's row FIRST, then put the next row's loads in flight
          __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
          __builtin_amdgcn_sched_barrier(0);
          double Pc[NMAX];
#pragma unroll
          for (int t = 0; t < NMAX; ++t) Pc[t] = Pn[t];
          if (i * PXG + e + 1 < 2 * PXG) {
            prow += SROW;
#pragma unroll
            for (int t = 0; t < NMAX; ++t) Pn[t] = prow[t];
          }
          __builtin_amdgcn_sched_barrier(0);  // keep the next row's loads above this pixel's arithmetic
          const bool in = c >= cmin;          // wave-uniform: quadrant pixel inside the disk
          const TIN av = in ? A[i][e] : (TIN)0, bv = in ? B[i][PXG - 1 - e] : (TIN)0;
          acc.template row_pixel_p0<decltype(first_row)::value>((double)av, (double)bv, Pc);
        }
      }
#else
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < PXG; ++e) {
          const int c = c0 + PXG * (q0 + i) + e;
          if (c >= cmin)  // wave-uniform: quadrant pixel inside the disk
            acc.template row_pixel<decltype(first_row)::value>((double)A[i][e], (double)B[i][PXG - 1 - e],
                                                               px + c * ZK_SEP_ROW);
        }
      }
#endif
    };
    if constexpr (WIDE) {
      const bool second = (utab[8 * u + 7] >> 1) & 1;
#pragma unroll
      for (int q0 = 0; q0 < 8; q0 += 2) {
        if (cmin < c0 + PXG * (q0 + 2)) {  // the disk columns of a row are a suffix: quarter 6 always runs
          if (second) quarter(q0, q0 == 6, std::false_type{});
          else quarter(q0, q0 == 6, std::true_type{});
        }
      }
#if ZK_ABLATE != 1
      if (rend) {
        acc.pair_combine();
        acc.row_end(px + r * ZK_SEP_ROW);
      }
#endif
    } else {
      const bool outer = cmin < c0 + 2 * PXG, inner = cmax > c0 + 2 * PXG;  // which halves hold disk pixels
      if (outer) half(0, !inner);
      if (inner) half(2, true);
#if ZK_ABLATE != 1
      if (rend) acc.row_end(px + r * ZK_SEP_ROW);
#endif
    }
  }

  const ZK_CONST int32_t* cmap = zk_const(colmap);
  if constexpr (PAIR) {
    // ---- Z = T M for this wave's classes; the pair assembles the output rows in its shared slab -----------------
    double z[S::NP];
    acc.transform(zk_const(tmat), [&](auto slot, double v) { z[slot] = v; });
    zk_batch_store_rows_pair<NMAX, MASK, S::NP>(z, cmap, (double*)wl, out + (nv > 0 ? patch0 : 0) * n_poly, lane, part, nv, n_poly,
                                                ppp);
  } else if constexpr (MASK == 15) {
    // ---- Z = T M, then (patch, column) rows via LDS -> 16-B stores ----------------------------------
    double z[S::NP];
    acc.transform(zk_const(tmat), [&](auto slot, double v) { z[slot] = v; });
    // ppp = patches per pass (host: largest power of two with ppp * n_poly <= 2048)
    zk_batch_store_rows<S::NP>(z, cmap, (double*)wl, out + patch0 * n_poly, lane, nv, n_poly, ppp);
  } else {
    // ---- class pass: `out` is the scratch matrix, plane stride n_patches ------------------------------
    double* const sp = out + patch0 + lane;
    acc.transform(zk_const(tmat), [&](auto slot, double v) {
      const int col = cmap[slot];
      if (col >= 0 && lane < nv) sp[(long long)col * n_patches] = v;
    });
  }
}

template <int NMAX, int RUN, typename TIN, bool WIDE, int MASK = 15>
__global__ __launch_bounds__(256, ((NMAX <= ZK_BATCH_2W || MASK != 15) ? 2 : 1)) void zk_patch_sep_kernel(
    const TIN* __restrict__ in, double* __restrict__ out, const zk_sep_unit* __restrict__ units,
    const double* __restrict__ xq, const double* __restrict__ tmat, const int32_t* __restrict__ colmap,
    int n_units, int n_poly, long long n_patches, int patch_bytes, int ppp, const int32_t* __restrict__ row_starts,
    int n_row_starts, const double* __restrict__ pfull) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // DEPTH slabs of 16 KiB per wave
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  zk_patch_body<NMAX, RUN, TIN, WIDE, MASK, false>(in, out, units, xq, tmat, colmap, n_units, n_poly, n_patches, patch_bytes, ppp,
                                                    row_starts, n_row_starts, pfull,
                                                    lds + wave * (4096 * ZK_BATCH_DEPTH(NMAX, MASK)),
                                                    (long long)blockIdx.x * 4 + wave, 0);
}

// pair form: 4 waves = 2 pairs per workgroup, 2 x 2 slabs of 16 KiB; two workgroups per CU (8 waves, 128 KiB)
template <int NMAX, int RUN, typename TIN, bool WIDE>
__global__ __launch_bounds__(256, 2) void zk_patch_pair_kernel(
    const TIN* __restrict__ in, double* __restrict__ out, const zk_sep_unit* __restrict__ units,
    const double* __restrict__ xq, const double* __restrict__ tmat, const int32_t* __restrict__ colmap,
    int n_units, int n_poly, long long n_patches, int patch_bytes, int ppp, const int32_t* __restrict__ row_starts,
    int n_row_starts, const double* __restrict__ pfull) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = wave >> 1;
  float* const wl = lds + pair * 8192;
  const long long group = (long long)blockIdx.x * 2 + pair;
  if ((wave & 1) == 0)
    zk_patch_body<NMAX, RUN, TIN, WIDE, (1 << ZK_EE) | (1 << ZK_EO), true>(in, out, units, xq, tmat, colmap, n_units, n_poly, n_patches,
                                                                            patch_bytes, ppp, row_starts, n_row_starts, pfull, wl,
                                                                            group, 0);
  else
    zk_patch_body<NMAX, RUN, TIN, WIDE, (1 << ZK_OE) | (1 << ZK_OO), true>(in, out, units, xq, tmat, colmap, n_units, n_poly, n_patches,
                                                                            patch_bytes, ppp, row_starts, n_row_starts, pfull, wl,
                                                                            group, 1);
}

// in[c * n + p] -> out[p * n_poly + c]: the class passes' scratch planes to (N, n_poly) rows
__global__ __launch_bounds__(256) void zk_transpose_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                           int n_poly, long long n) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long long p0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  for (int j = ty; j < 32; j += 8)
    if (c0 + j < n_poly && p0 + tx < n) tile[j][tx] = in[(long long)(c0 + j) * n + p0 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (p0 + j < n && c0 + tx < n_poly) out[(p0 + j) * n_poly + c0 + tx] = tile[tx][j];
}

template <int NMAX, int RUN, typename TIN, bool WIDE, int MASK = 15>
int launch_one(zk_plan* p, const void* in, int64_t n_patches, double* out, hipStream_t s) {
  const zk_sep_tables* t = p->sep;
  const zk_sep_tables::batch_tables& bt = t->batch[sizeof(TIN) == 4 ? 0 : 1];
  const long long waves = (n_patches + 63) / 64;
  const long long blocks = (waves + 3) / 4;
  if (blocks > 0x7fffffffLL) return zk_fail(ZK_E_BADARG, "too many patches for one launch");
  int ppp = 64;
  while (ppp * p->n_poly > 2048) ppp >>= 1;
  if constexpr (MASK == 15 && (NMAX > ZK_BATCH_2W)) {
    // n_max 11-16.  Row-pair units (sizes whose rows are not whole 128-B lines twice over): the pair kernel -- two waves per
    // SIMD instead of one: (32, 12) 4.70 -> 5.39 TB/s, (32, 14) 3.96 -> 4.78, (32, 16) 2.66 -> 3.43, (40, 12) 3.65 -> 3.88.
    // Wide units (float32 K % 64 == 0, float64 K % 32 == 0) stay on the one-wave kernel with its pipelined scalar loads,
    // where each wave of a pair would use half of every Legendre row it fetches: (128, 12) 6.18 vs 5.30, (64, 16) 3.53 vs
    // 2.93 (profiles/r02_batch_sweep.txt).  ZK_BATCH_PAIR=0 / 1 in the environment forces one form (A/B runs).
    static const char* const force = getenv("ZK_BATCH_PAIR");
    const bool use_pair = force ? force[0] != '0' : !WIDE;
    if (use_pair) {
      auto pk = zk_patch_pair_kernel<NMAX, RUN, TIN, WIDE>;
      const long long pblocks = (waves + 1) / 2;
      int rc = zk_prof_begin(p, s);
      if (rc) return rc;
      hipLaunchKernelGGL(pk, dim3((unsigned)pblocks), dim3(256), 65536, s, (const TIN*)in, out, bt.d_units, t->d_xq, t->d_T,
                         t->d_colmap, bt.n_units, p->n_poly, (long long)n_patches, p->size * p->size * (int)sizeof(TIN), ppp,
                         bt.d_row_starts, ZK_ROTATE ? bt.n_row_starts : 0, t->d_pfull);
      ZK_HIP(hipGetLastError());
      return zk_prof_end(p, s);
    }
  }
  auto kern = zk_patch_sep_kernel<NMAX, RUN, TIN, WIDE, MASK>;
  const size_t lds = (size_t)4 * ZK_BATCH_DEPTH(NMAX, MASK) * 16384;
  if (lds > 64 * 1024)
    ZK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int rc = zk_prof_begin(p, s);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, (const TIN*)in,
                     out, bt.d_units, t->d_xq, t->d_T, t->d_colmap, bt.n_units, p->n_poly, (long long)n_patches,
                     p->size * p->size * (int)sizeof(TIN), ppp, bt.d_row_starts, ZK_ROTATE ? bt.n_row_starts : 0,
                     t->d_pfull);
  ZK_HIP(hipGetLastError());
  return zk_prof_end(p, s);
}

#if ZK_NMAX_GROUP == 2
// n_max > 16: per chunk of patches, one launch per parity class into the plan's scratch planes, then the
// transposition into the caller's rows (64-B-run units only: these sizes are bound by arithmetic)
template <int NMAX, typename TIN>
int launch_passes(zk_plan* p, const void* in, int64_t n_patches, double* out, hipStream_t s) {
  const int64_t chunk_max = 1 << 18;
  const size_t need = (size_t)(n_patches < chunk_max ? n_patches : chunk_max) * p->n_poly * sizeof(double);
  if (p->d_scratch_bytes < need) {
    if (p->d_scratch) ZK_HIP(hipFree(p->d_scratch));
    p->d_scratch = nullptr;
    p->d_scratch_bytes = 0;
    ZK_HIP(hipMalloc((void**)&p->d_scratch, need));
    p->d_scratch_bytes = need;
  }
  const size_t patch_elems = (size_t)p->size * p->size;
  for (int64_t first = 0; first < n_patches; first += chunk_max) {
    const int64_t n = n_patches - first < chunk_max ? n_patches - first : chunk_max;
    const TIN* src = (const TIN*)in + first * patch_elems;
    int rc = launch_one<NMAX, 4, TIN, false, 1 << ZK_EE>(p, src, n, p->d_scratch, s);
    if (!rc) rc = launch_one<NMAX, 4, TIN, false, 1 << ZK_OE>(p, src, n, p->d_scratch, s);
    if (!rc) rc = launch_one<NMAX, 4, TIN, false, 1 << ZK_EO>(p, src, n, p->d_scratch, s);
    if (!rc) rc = launch_one<NMAX, 4, TIN, false, 1 << ZK_OO>(p, src, n, p->d_scratch, s);
    if (rc) return rc;
    if ((rc = zk_prof_begin(p, s))) return rc;
    hipLaunchKernelGGL(zk_transpose_kernel, dim3((unsigned)((n + 31) / 32), (unsigned)((p->n_poly + 31) / 32)), dim3(256), 0,
                       s, p->d_scratch, out + first * p->n_poly, p->n_poly, (long long)n);
    ZK_HIP(hipGetLastError());
    if ((rc = zk_prof_end(p, s))) return rc;
  }
  return 0;
}

#else
template <int RUN, typename TIN, bool WIDE = false>
int launch_run(zk_plan* p, const void* in, int64_t n_patches, double* out, hipStream_t s) {
  switch (p->sep->kernel_nmax) {
#if ZK_NMAX_GROUP == 0
    case 4: return launch_one<4, RUN, TIN, WIDE>(p, in, n_patches, out, s);
    case 6: return launch_one<6, RUN, TIN, WIDE>(p, in, n_patches, out, s);
    case 8: return launch_one<8, RUN, TIN, WIDE>(p, in, n_patches, out, s);
    case 10: return launch_one<10, RUN, TIN, WIDE>(p, in, n_patches, out, s);
    case 12: return launch_one<12, RUN, TIN, WIDE>(p, in, n_patches, out, s);
#endif
#if ZK_NMAX_GROUP == 1
    case 14: return launch_one<14, RUN, TIN, WIDE>(p, in, n_patches, out, s);
    case 16: return launch_one<16, RUN, TIN, WIDE>(p, in, n_patches, out, s);
#endif
  }
  return zk_fail(ZK_E_BADARG, "no batch kernel for this n_max");
}
#endif

}  // namespace

#if ZK_NMAX_GROUP == 0
int zk_launch_sep_patches_g1(zk_plan* p, const void* in, int dtype, int64_t n_patches, double* out, hipStream_t s);
int zk_launch_sep_patches_g2(zk_plan* p, const void* in, int dtype, int64_t n_patches, double* out, hipStream_t s);

bool zk_sep_patches_available(const zk_plan* p, int dtype) {
  const zk_sep_tables* t = p->sep;
  // K <= 1024 keeps the 32-bit byte offsets inside a 64-patch group (63 * K * K * 8 < 2^31)
  return t && t->batch[dtype == ZK_F32 ? 0 : 1].n_units > 0 && p->n_poly <= 1024 && p->size <= 1024;
}

#endif

int ZK_GROUP_FN(zk_launch_sep_patches)(zk_plan* p, const void* in, int dtype, int64_t n_patches, double* out,
                                       hipStream_t s) {
#if ZK_NMAX_GROUP == 0
  if (((uintptr_t)in & (dtype == ZK_F32 ? 3 : 7)) || ((uintptr_t)out & 7))  // element-aligned operands
    return zk_launch_generic_patches(p, in, dtype, n_patches, out, s);
  if (p->sep->kernel_nmax > 16) return zk_launch_sep_patches_g2(p, in, dtype, n_patches, out, s);
  if (p->sep->kernel_nmax > 12) return zk_launch_sep_patches_g1(p, in, dtype, n_patches, out, s);
#endif
#if ZK_NMAX_GROUP == 2
  const bool f32 = dtype == ZK_F32;
  if (p->sep->kernel_nmax == 20)
    return f32 ? launch_passes<20, float>(p, in, n_patches, out, s) : launch_passes<20, double>(p, in, n_patches, out, s);
  return f32 ? launch_passes<24, float>(p, in, n_patches, out, s) : launch_passes<24, double>(p, in, n_patches, out, s);
#else
  if (dtype == ZK_F64)
    return p->sep->batch[1].wide ? launch_run<8, double, true>(p, in, n_patches, out, s)
                                 : launch_run<4, double>(p, in, n_patches, out, s);
  if (p->sep->batch[0].wide) return launch_run<8, float, true>(p, in, n_patches, out, s);
  return p->sep->batch[0].run == 8 ? launch_run<8, float>(p, in, n_patches, out, s)
                                   : launch_run<4, float>(p, in, n_patches, out, s);
#endif
}
