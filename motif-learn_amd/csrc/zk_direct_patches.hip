// zk_direct_patches.hip -- batch-of-patches moments as the plain sum the reference computes (_zps.py:146-157: np.dot of the
// flattened patch with the flattened basis), every disk pixel times the CALLER'S number for it, for function sets too large
// for a lane's registers and beyond the reach of the polynomial kernels: n_max 25 .. 40 (351 .. 861 functions), which is what
// the reference's own estimator returns for 56 .. 72-px patches (features/_estimate_n_max.py:95,123).
//
// Why not cheaper arithmetic: above n_max ~24 the reference's float64 basis is neither the exact polynomial (2e-4 of max|V| off
// at 36) nor mirror-symmetric (8e-4 at 36), so neither the row-separable nor a mirror-folded sum restates the reference's result
// to 1e-6 (profiles/r03_high_orders.txt).  What is left is the product itself: (patches x pixels) . (pixels x functions), 1 to
// 3 MFLOP per patch -- a GEMM, compute-bound by a wide margin (arithmetic intensity in the hundreds of flop per byte; the
// 32-px / n_max 8 headline path is the opposite case and stays off the matrix cores).  zk_generic_kernel's batch mode ran it
// at 3 % of the FP64 peak (an uncoalesced 4-byte load per lane and pixel, waited for, then 64 FMAs on scalar operands that miss
// the scalar cache: 0.5 M patches/s at (72, 36)); a DMA-staged version with the table in SGPRs still waited ~40 clocks per FMA
// for scalars out of a 5-MB table.  So: v_mfma_f64_16x16x4_f64, both operands from vector memory.
//
//   wave = 64 patches (4 blocks of 16) x CH = 96 functions (6 blocks of 16): 24 accumulator blocks of 4 doubles per lane;
//   patches move as in zk_sep_patches.hip: a unit = four 64-B runs (16 float32 / 8 float64 pixels each, pieces of the disk
//   rows) of all 64 patches = 16 KiB, moved by 16 global_load_lds_dwordx4 (each: one run of 16 patches), the granule a lane
//   fetches rotated by its patch index -- the per-lane 4-byte read of "pixel s of patch p" below is then bank-conflict-free;
//   k dimension = pixels, 4 per MFMA: A[i][k] = pixel (4 step + k) of patch 16 pb + i (LDS, converted to float64),
//   B[k][j] = table row of that pixel, function 16 fb + j (global memory: [unit][pixel slot][CH] float64, the caller's values
//   / area, zero rows where a run overlaps its neighbour or leaves the disk -- every disk pixel is owned by exactly one slot,
//   checked at plan creation; steps whose four rows are all zero are skipped);
//   the functions are done CH at a time: one launch per chunk over the same patches (the patches come from L2 / Infinity Cache
//   after the first chunk); results go straight into the (N, n_poly) rows (16 consecutive functions per 128-B segment).
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "zk_fold.h"

#ifndef ZK_DIRECT_CH
#define ZK_DIRECT_CH 96  // functions per launch: 24 accumulator blocks = 192 registers, two waves per SIMD
#endif

struct zk_direct_unit {
  int32_t run_off[4];  // byte offsets of the unit's four runs inside a patch
  int32_t steps;       // bit s: step s (4 pixel slots) has a non-zero table row
  uint32_t own[2];     // bit 4 s + k: slot k of step s stands for a disk pixel (the others contribute an exact zero, whatever
                       // the pixel holds: a NaN outside the disk, or in the overlap of two runs, never reaches a moment)
  int32_t pad;
};

struct zk_direct_tables {
  int n_chunks = 0;
  struct per_type {
    int n_units = 0;
    zk_direct_unit* d_units = nullptr;
    double* d_tab = nullptr;  // [n_chunks][n_units][4 runs x UP slots][CH]
  } t[2];                     // [0] float32 (UP = 16 pixels per run), [1] float64 (UP = 8)
  // dense mode: the disk rows in pieces of 4 consecutive pixels (a piece = one MFMA step)
  int n_steps = 0;
  int tile_pitch = 0;
  bool flipped = false;           // the dense table holds (-1)^n V(K-1-r, K-1-c): what the reference's convolution multiplies
  int32_t* d_step_off = nullptr;  // [n_steps] bits 0..23: tile element offset of the piece's first pixel (window row r:
                                  // r * tile_pitch + c); bits 24..27: which of its 4 slots stand for a disk pixel
  double* d_ftab = nullptr;       // [n_chunks][n_steps][4][CH]
};

namespace {

constexpr int CH = ZK_DIRECT_CH;
constexpr int FB = CH / 16;  // function blocks

#define ZK_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define ZK_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <typename T>
int upload(T** dst, const std::vector<T>& src) {
  if (src.empty()) return 0;
  ZK_HIP(hipMalloc((void**)dst, src.size() * sizeof(T)));
  ZK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

// one chunk of CH functions (columns col0 .. col0 + n_live - 1 of the result) for all patches
template <typename TIN, bool ROLL>
__global__ __launch_bounds__(256, 2) void zk_patch_direct_kernel(const TIN* __restrict__ in, double* __restrict__ out,
                                                                 const zk_direct_unit* __restrict__ units,
                                                                 const double* __restrict__ tab, int n_units, int col0, int n_live,
                                                                 int n_poly, long long n_patches, int patch_bytes) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  constexpr int PXG = 16 / sizeof(TIN);  // pixels per 16-B granule: 4 (float32) or 2 (float64)
  constexpr int UP = 4 * PXG;            // pixels per run
  static_assert(UP == 4 * PXG, "a run = PXG steps of 4 slots");
  extern __shared__ __attribute__((aligned(16))) float lds[];  // one 16-KiB slab per wave
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const TIN* const ws = (const TIN*)(lds + wave * 4096);
  const long long patch0 = ((long long)blockIdx.x * 4 + wave) * 64;
  if (patch0 >= n_patches) return;  // (no barrier in this kernel: a wave only reads LDS bytes it DMA'd itself)
  const long long left = n_patches - patch0;
  const int nv = left < 64 ? (int)left : 64;

  // ---- DMA addressing (as zk_sep_patches.hip, 64-B runs): lane -> (patch-in-group a, slot b) ---------------------------
  const int a = lane >> 2, b = lane & 3;
  const int g0 = (b - (a >> 2)) & 3;  // source granule: rotated by the patch index
  const char* const wbase = (const char*)in + patch0 * patch_bytes;
  int poff[4];
#pragma unroll
  for (int pg = 0; pg < 4; ++pg) {
    int pi = pg * 16 + a;
    pi = pi < nv ? pi : nv - 1;  // tail wave: re-read the last live patch
    poff[pg] = pi * patch_bytes + g0 * 16;
  }
  const ZK_CONST int32_t* utab = zk_const((const int32_t*)units);  // 8 ints per unit
  auto issue = [&](int u) {
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
      const int ro = utab[8 * u + rho];
#pragma unroll
      for (int pg = 0; pg < 4; ++pg)
        __builtin_amdgcn_global_load_lds(ZK_GLOBAL_PTR(wbase + (poff[pg] + ro)), ZK_LDS_PTR(lds + wave * 4096 + (rho * 4 + pg) * 256),
                                         16, 0, 0);
    }
  };

  // ---- MFMA operand addressing: lane = (i = lane & 15, k = lane >> 4) -----------------------------------------------------
  // pixel slot s = 4 step + k of run rho = step / PXG... a run holds UP = 4 PXG slots = PXG steps; inside the run, slot
  // x = 4 (step % PXG) + k lies in granule x / PXG at element x % PXG; patch p = 16 pb + i sits at 64 B * p of the run image
  // with its granules rotated by p >> 2
  const int li = lane & 15, kr = lane >> 4;
  int pbase[4];  // element index of (patch 16 pb + i, granule 0) inside a run image
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) pbase[pb] = (16 * pb + li) * UP;
  const int prot = li >> 2;  // (16 pb + i) >> 2 = 4 pb + (i >> 2): the same rotation mod 4 for every pb
  v4d acc[4][FB];
#pragma unroll
  for (int pb = 0; pb < 4; ++pb)
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) acc[pb][fb] = v4d{0.0, 0.0, 0.0, 0.0};
  const double* __restrict__ tlane = tab + (size_t)kr * CH + li;  // this lane's column of a step's four table rows

  // Rolling re-arm (round 4): a run's 4-KiB piece of the slab is free as soon as its PXG steps have read it, so the next unit's
  // run goes into it at once and has the other three runs' arithmetic (12 steps x 24 MFMAs) to land -- the DMA latency that the
  // whole-slab form exposed once per unit is hidden.  Vector-memory operations complete in order: at the start of a run the only
  // younger ones are the four DMA instructions issued at the end of the previous run, so vmcnt(4) says "this run has landed".
  auto issue_run = [&](int u, int rho) {
    const int ro = utab[8 * u + rho];
#pragma unroll
    for (int pg = 0; pg < 4; ++pg)
      __builtin_amdgcn_global_load_lds(ZK_GLOBAL_PTR(wbase + (poff[pg] + ro)), ZK_LDS_PTR(lds + wave * 4096 + (rho * 4 + pg) * 256), 16,
                                       0, 0);
  };
  (void)issue;
#pragma unroll
  for (int rho = 0; rho < 4; ++rho) issue_run(0, rho);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (!ROLL) {  // the round-3 form: the whole slab re-armed after the unit's last step (ZK_DIRECT_NO_ROLL=1 / 0 forces either)
    for (int u = 0; u < n_units; ++u) {
      const int steps = utab[8 * u + 4];
      const unsigned own_lo = (unsigned)utab[8 * u + 5], own_hi = (unsigned)utab[8 * u + 6];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const double* __restrict__ tu = tlane + (size_t)u * (4 * UP) * CH;
#pragma unroll 2
      for (int st = 0; st < 4 * PXG; ++st) {
        if (!((steps >> st) & 1)) continue;
        const int rho = st / PXG;
        const int x = 4 * (st % PXG) + kr;
        const int gsl = ((x / PXG + prot) & 3) * PXG + x % PXG;
        double av[4], bv[FB];
        const bool mine = (((st < 8 ? own_lo : own_hi) >> (4 * (st & 7) + kr)) & 1u) != 0;
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
          const double v = (double)ws[rho * 64 * UP + pbase[pb] + gsl];
          av[pb] = mine ? v : 0.0;
        }
        const double* __restrict__ tr = tu + (size_t)st * 4 * CH;
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) bv[fb] = tr[16 * fb];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
          for (int fb = 0; fb < FB; ++fb) acc[pb][fb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[pb], bv[fb], acc[pb][fb], 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (u + 1 < n_units) issue(u + 1);
    }
  } else
  for (int u = 0; u < n_units; ++u) {
    const int steps = utab[8 * u + 4];
    const unsigned own_lo = (unsigned)utab[8 * u + 5], own_hi = (unsigned)utab[8 * u + 6];
    const double* __restrict__ tu = tlane + (size_t)u * (4 * UP) * CH;
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // run rho of unit u has landed (issued a unit ago)
#pragma unroll
      for (int sr = 0; sr < PXG; ++sr) {
        const int st = rho * PXG + sr;
        if (!((steps >> st) & 1)) continue;  // wave-uniform: four zero rows
        const int x = 4 * sr + kr;                                     // slot inside the run
        const int gsl = ((x / PXG + prot) & 3) * PXG + x % PXG;        // its element inside the patch's (rotated) 64 B
        double av[4], bv[FB];
        const bool mine = (((st < 8 ? own_lo : own_hi) >> (4 * (st & 7) + kr)) & 1u) != 0;
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
          const double v = (double)ws[rho * 64 * UP + pbase[pb] + gsl];
          av[pb] = mine ? v : 0.0;
        }
        const double* __restrict__ tr = tu + (size_t)st * 4 * CH;
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) bv[fb] = tr[16 * fb];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
          for (int fb = 0; fb < FB; ++fb) acc[pb][fb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[pb], bv[fb], acc[pb][fb], 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this run's piece of the slab is no longer read
      if (u + 1 < n_units) issue_run(u + 1, rho);
    }
  }
  // D layout: lane holds rows (patch-in-block) kr + 4 q, column (function-in-block) li
#pragma unroll
  for (int pb = 0; pb < 4; ++pb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pl = 16 * pb + kr + 4 * q;
      if (pl < nv) {
        double* __restrict__ dst = out + (patch0 + pl) * n_poly + col0 + li;
#pragma unroll
        for (int fb = 0; fb < FB; ++fb)
          if (16 * fb + li < n_live) dst[16 * fb] = acc[pb][fb][q];
      }
    }
}

template <typename TIN>
int launch_t(zk_plan* p, const void* in, int64_t n_patches, double* out, hipStream_t s) {
  const zk_direct_tables* d = p->direct;
  const zk_direct_tables::per_type& t = d->t[sizeof(TIN) == 4 ? 0 : 1];
  constexpr int UP = 64 / (int)sizeof(TIN);
  const size_t patch_elems = (size_t)p->size * p->size;
  const size_t tab_chunk = (size_t)t.n_units * 4 * UP * CH;
  // (every chunk's launch streams the patches again: 8 x 20.7 KB per patch at (72, 36) against 5.5 MFLOP -- a quarter of the
  //  arithmetic's time at the HBM rate, and it overlaps)
  const int64_t round_max = (int64_t)1 << 22;
  for (int64_t first = 0; first < n_patches; first += round_max) {
    const int64_t n = std::min<int64_t>(n_patches - first, round_max);
    const TIN* src = (const TIN*)in + first * patch_elems;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    for (int c = 0; c < d->n_chunks; ++c) {
      const int n_live = std::min(CH, p->n_poly - c * CH);
      int rc = zk_prof_begin(p, s);
      if (rc) return rc;
      // whole-slab re-arm when every SIMD holds two waves (the other wave hides the DMA wait: 0.64-0.84 of the FP64 peak,
      // 2 % ahead of the rolling form); rolling re-arm for batches that leave CUs half empty (+7 %): profiles/r04_direct_batch.txt
      const char* force = getenv("ZK_DIRECT_NO_ROLL");
      if (force ? *force == '1' : blocks >= 2u * (unsigned)p->n_cu)
        hipLaunchKernelGGL((zk_patch_direct_kernel<TIN, false>), dim3(blocks), dim3(256), 65536, s, src, out + first * p->n_poly, t.d_units,
                           t.d_tab + c * tab_chunk, t.n_units, c * CH, n_live, p->n_poly, (long long)n, (int)(patch_elems * sizeof(TIN)));
      else
        hipLaunchKernelGGL((zk_patch_direct_kernel<TIN, true>), dim3(blocks), dim3(256), 65536, s, src, out + first * p->n_poly, t.d_units,
                           t.d_tab + c * tab_chunk, t.n_units, c * CH, n_live, p->n_poly, (long long)n, (int)(patch_elems * sizeof(TIN)));
      ZK_HIP(hipGetLastError());
      if ((rc = zk_prof_end(p, s))) return rc;
    }
  }
  return 0;
}

// Dense mode of the same sum: 8 output rows x 64 columns per workgroup, the zero-padded tile staged once in LDS (in the
// image's own element type) and walked once per chunk; wave = one output row = 64 positions (4 blocks of 16) x CH functions.
// A step = 4 consecutive pixels of a disk row: A[i][k] = tile[window origin of position 16 pb + i + piece offset + k].
template <typename TIN>
__global__ __launch_bounds__(512) void zk_frame_direct_kernel(const TIN* __restrict__ img, double* __restrict__ out,
                                                               const int32_t* __restrict__ step_off, const double* __restrict__ tab,
                                                               int n_steps, int n_chunks, int n_poly, int K, int H, int W, int row0,
                                                               int n_rows, int tile_pitch, long long plane) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float lds[];
  TIN* const tile = (TIN*)lds;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ea = K - 1 - (K - 1) / 2;
  const int i0 = row0 + blockIdx.y * 8, k0 = blockIdx.x * 64;
  const int tile_rows = K + 7;
  for (int tr = wave; tr < tile_rows; tr += 8) {
    const int ii = i0 - ea + tr;
    for (int tc0 = 0; tc0 < tile_pitch; tc0 += 256) {
      TIN v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int tc = tc0 + q * 64 + lane, kk = k0 - ea + tc;
        v[q] = (TIN)0;
        if (tc < tile_pitch && ii >= 0 && ii < H && kk >= 0 && kk < W) v[q] = img[(long long)ii * W + kk];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (tc0 + q * 64 + lane < tile_pitch) tile[tr * tile_pitch + tc0 + q * 64 + lane] = v[q];
    }
  }
  __syncthreads();

  const int li = lane & 15, kr = lane >> 4;
  const TIN* __restrict__ mine = tile + wave * tile_pitch + li + kr;  // (+ 16 pb + piece offset)
  const ZK_CONST int32_t* soff = zk_const(step_off);
  const int oi = i0 + wave;
  const bool row_live = oi < row0 + n_rows;
  for (int ch = 0; ch < n_chunks; ++ch) {
    v4d acc[4][FB];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb)
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) acc[pb][fb] = v4d{0.0, 0.0, 0.0, 0.0};
    const double* __restrict__ tl = tab + ((size_t)ch * n_steps * 4 + kr) * CH + li;
    for (int st = 0; st < n_steps; ++st) {
      const int so = soff[st];
      const TIN* __restrict__ px = mine + (so & 0xffffff);
      const bool own = ((so >> (24 + kr)) & 1) != 0;
      double av[4], bv[FB];
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) {
        const double v = (double)px[16 * pb];
        av[pb] = own ? v : 0.0;
      }
      const double* __restrict__ tr = tl + (size_t)st * 4 * CH;
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) bv[fb] = tr[16 * fb];
#pragma unroll
      for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) acc[pb][fb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[pb], bv[fb], acc[pb][fb], 0, 0, 0);
    }
    if (row_live) {
      const int n_live = n_poly - ch * CH < CH ? n_poly - ch * CH : CH;
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) {
        const bool col_live = 16 * fb + li < n_live;
        double* __restrict__ dst = out + (long long)(ch * CH + 16 * fb + li) * plane + (long long)(oi - row0) * W + k0;
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ok = 16 * pb + kr + 4 * q;
            if (col_live && k0 + ok < W) dst[ok] = acc[pb][fb][q];
          }
      }
    }
  }
}

template <typename TIN>
int launch_frame_t(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out, hipStream_t s) {
  const zk_direct_tables* d = p->direct;
  const size_t lds = (size_t)(p->size + 7) * d->tile_pitch * sizeof(TIN);
  auto kern = zk_frame_direct_kernel<TIN>;
  if (lds > 64 * 1024)
    ZK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long plane = zk_out_plane(p, n_rows, W);
  return zk_for_row_bands(row0, n_rows, W, 8, [&](int64_t r0, int64_t nr, long long off) {
    dim3 grid((unsigned)((W + 63) / 64), (unsigned)((nr + 7) / 8));
    int rc = zk_prof_begin(p, s);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, (const TIN*)in, out + off, d->d_step_off, d->d_ftab, d->n_steps, d->n_chunks,
                       p->n_poly, p->size, (int)H, (int)W, (int)r0, (int)nr, d->tile_pitch, plane);
    ZK_HIP(hipGetLastError());
    return zk_prof_end(p, s);
  });
}

}  // namespace

void zk_direct_free(zk_plan* p) {
  zk_direct_tables* d = p->direct;
  if (!d) return;
  for (auto& t : d->t) {
    if (t.d_units) (void)hipFree(t.d_units);
    if (t.d_tab) (void)hipFree(t.d_tab);
  }
  if (d->d_step_off) (void)hipFree(d->d_step_off);
  if (d->d_ftab) (void)hipFree(d->d_ftab);
  delete d;
  p->direct = nullptr;
}

// Built for large sets only (the register-resident and polynomial kernels serve the others): any basis, no assumption beyond
// its zero pattern -- a table row holds the caller's values (/ area) of one pixel for the CH functions of a chunk.
int zk_direct_build(zk_plan* p, const double* basis) {
  const int K = p->size, NP = p->n_poly;
  if (NP < 92 || NP > 1024 || K < 16 || K > 512) return 0;
  const double inv_area = 1.0 / (M_PI * (double)K * (double)K / 4.0);
  std::vector<char> act((size_t)K * K, 0);
  for (int j = 0; j < NP; ++j)
    for (int t = 0; t < K * K; ++t)
      if (basis[(size_t)j * K * K + t] != 0.0) act[t] = 1;
  zk_direct_tables* d = new zk_direct_tables();
  p->direct = d;
  d->n_chunks = (NP + CH - 1) / CH;
  for (int dt = 0; dt < 2; ++dt) {
    const int es = dt == 0 ? 4 : 8, UP = 64 / es;
    // runs: the disk segment [lo, hi) of every row in pieces of UP pixels; the last piece of a row is moved back so that it
    // ends inside the row (it then overlaps its neighbour: the overlapped slots own nothing)
    std::vector<int> run_off;  // byte offset of each run
    std::vector<int> owner;    // [run][UP]: pixel index r * K + c the slot stands for, or -1
    std::vector<int> cover((size_t)K * K, 0);
    for (int r = 0; r < K; ++r) {
      int lo = K, hi = 0;
      for (int c = 0; c < K; ++c)
        if (act[(size_t)r * K + c]) {
          lo = std::min(lo, c);
          hi = c + 1;
        }
      for (int c0 = lo; c0 < hi; c0 += UP) {
        const int start = std::min(c0, K - UP);
        run_off.push_back((r * K + start) * es);
        for (int x = 0; x < UP; ++x) {
          const int c = start + x;
          const bool own = c >= c0 && c < std::min(hi, c0 + UP) && act[(size_t)r * K + c];
          owner.push_back(own ? r * K + c : -1);
          if (own) ++cover[(size_t)r * K + c];
        }
      }
    }
    for (int t = 0; t < K * K; ++t)
      if (cover[t] != (act[t] ? 1 : 0)) {  // every disk pixel exactly once
        zk_direct_free(p);
        return zk_fail(ZK_E_BADARG, "internal: direct batch runs do not tile the disk");
      }
    while (run_off.size() % 4) {  // whole units: an empty run re-reads the patch's first bytes, owns nothing
      run_off.push_back(0);
      owner.insert(owner.end(), UP, -1);
    }
    const int n_units = (int)run_off.size() / 4, slots = 4 * UP;
    std::vector<zk_direct_unit> units((size_t)n_units);
    for (int u = 0; u < n_units; ++u) {
      zk_direct_unit& un = units[u];
      un = zk_direct_unit{};
      for (int rho = 0; rho < 4; ++rho) un.run_off[rho] = run_off[4 * u + rho];
      for (int st = 0; st < slots / 4; ++st)
        for (int k = 0; k < 4; ++k)
          if (owner[(size_t)u * slots + 4 * st + k] >= 0) {
            un.steps |= 1 << st;
            un.own[st >> 3] |= 1u << (4 * (st & 7) + k);
          }
    }
    zk_direct_tables::per_type& t = d->t[dt];
    t.n_units = n_units;
    std::vector<double> tab((size_t)d->n_chunks * n_units * slots * CH, 0.0);
    for (int c = 0; c < d->n_chunks; ++c)
      for (size_t k = 0; k < owner.size(); ++k) {
        if (owner[k] < 0) continue;
        double* dst = &tab[((size_t)c * owner.size() + k) * CH];
        for (int i = 0; i < CH && c * CH + i < NP; ++i) dst[i] = basis[(size_t)(c * CH + i) * K * K + owner[k]] * inv_area;
      }
    int rc = upload(&t.d_units, units);
    if (!rc) rc = upload(&t.d_tab, tab);
    if (rc) return rc;
  }
  // ---- dense mode: disk rows in pieces of 4 consecutive pixels.  A piece may run up to 3 pixels past the window's last
  // column (its extra slots own nothing and are masked), so position 63 of a workgroup reads up to column K + 65 of the tile.
  // What the reference's dense path multiplies a window pixel (r, c) with is not V(r, c) but (-1)^n V(K-1-r, K-1-c)
  // (_zps.py:165-178: a convolution, then the sign): the dense table holds the flipped, signed values whenever the plan
  // says so (zk_plan::conv_flip).
  {
    d->tile_pitch = K + 66;
    d->flipped = p->conv_flip;
    auto value = [&](int j, int px) {
      if (!d->flipped) return basis[(size_t)j * K * K + px];
      return ((p->n[j] & 1) ? -1.0 : 1.0) * basis[(size_t)j * K * K + (K * K - 1 - px)];
    };
    std::vector<char> actd((size_t)K * K);  // window pixels the dense table has a non-zero row for
    for (int t = 0; t < K * K; ++t) actd[t] = d->flipped ? act[K * K - 1 - t] : act[t];
    std::vector<int32_t> soff;
    std::vector<int> owner;  // [step][4]
    for (int r = 0; r < K; ++r)
      for (int c = 0; c < K;) {
        if (!actd[(size_t)r * K + c]) {
          ++c;
          continue;
        }
        int packed = r * d->tile_pitch + c;
        for (int k = 0; k < 4; ++k) {
          const bool own = c + k < K && actd[(size_t)r * K + c + k];
          owner.push_back(own ? r * K + c + k : -1);
          if (own) packed |= 1 << (24 + k);
        }
        soff.push_back(packed);
        c += 4;
      }
    std::vector<int> cover((size_t)K * K, 0);
    for (int o : owner)
      if (o >= 0) ++cover[o];
    for (int t = 0; t < K * K; ++t)
      if (cover[t] != (actd[t] ? 1 : 0)) {
        zk_direct_free(p);
        return zk_fail(ZK_E_BADARG, "internal: direct dense pieces do not tile the disk");
      }
    d->n_steps = (int)soff.size();
    std::vector<double> ftab((size_t)d->n_chunks * owner.size() * CH, 0.0);
    for (int c = 0; c < d->n_chunks; ++c)
      for (size_t k = 0; k < owner.size(); ++k) {
        if (owner[k] < 0) continue;
        double* dst = &ftab[((size_t)c * owner.size() + k) * CH];
        for (int i = 0; i < CH && c * CH + i < NP; ++i) dst[i] = value(c * CH + i, owner[k]) * inv_area;
      }
    int rc = upload(&d->d_step_off, soff);
    if (!rc) rc = upload(&d->d_ftab, ftab);
    if (rc) return rc;
  }
  return 0;
}

bool zk_direct_patches_available(const zk_plan* p, int dtype) {
  return p->direct && p->direct->t[dtype == ZK_F32 ? 0 : 1].n_units > 0;
}

bool zk_direct_frame_available(const zk_plan* p, int dtype) {
  return p->direct && p->direct->n_steps > 0 &&
         (size_t)(p->size + 7) * p->direct->tile_pitch * (dtype == ZK_F32 ? 4 : 8) <= 150 * 1024;
}

int zk_launch_direct_frame(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out,
                           hipStream_t s) {
  if (dtype == ZK_F32) return launch_frame_t<float>(p, in, H, W, row0, n_rows, out, s);
  return launch_frame_t<double>(p, in, H, W, row0, n_rows, out, s);
}

int zk_launch_direct_patches(zk_plan* p, const void* in, int dtype, int64_t n_patches, double* out, hipStream_t s) {
  if (((uintptr_t)in & (dtype == ZK_F32 ? 3 : 7)) || ((uintptr_t)out & 7)) return zk_launch_generic_patches(p, in, dtype, n_patches, out, s);
  if (dtype == ZK_F32) return launch_t<float>(p, in, n_patches, out, s);
  return launch_t<double>(p, in, n_patches, out, s);
}
