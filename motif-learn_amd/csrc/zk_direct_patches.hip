// zk_direct_patches.hip -- batch-of-patches moments as the plain sum the reference computes (_zps.py:146-157: np.dot of the
// flattened patch with the flattened basis), every disk pixel times the CALLER'S number for it, for function sets too large
// for a lane's registers and beyond the reach of the polynomial kernels: what ZK_PATH_AUTO runs from n_max 17 (171 .. 861
// functions up to n_max 40 -- the orders the reference's own estimator returns for 40 .. 72-px patches,
// features/_estimate_n_max.py:95,123) and for any other set of >= 92 functions.
//
// Why not cheaper arithmetic: on structured inputs the polynomial kernels leave the reference's outputs by more than SURVEY
// 8(c)'s criterion from n_max 20-22 (profiles/r04_high_orders.txt), and above n_max ~24 the reference's float64 basis is neither
// the exact polynomial (2e-4 of max|V| off at 36) nor mirror-symmetric (8e-4 at 36), so neither the row-separable nor a
// mirror-folded sum restates the reference's result to 1e-6 (profiles/r03_high_orders.txt).  What is left is the product itself: (patches x pixels) . (pixels x functions), 1 to
// 3 MFLOP per patch -- a GEMM, compute-bound by a wide margin (arithmetic intensity in the hundreds of flop per byte; the
// 32-px / n_max 8 headline path is the opposite case and stays off the matrix cores).  zk_generic_kernel's batch mode ran it
// at 3 % of the FP64 peak (an uncoalesced 4-byte load per lane and pixel, waited for, then 64 FMAs on scalar operands that miss
// the scalar cache: 0.5 M patches/s at (72, 36)); a DMA-staged version with the table in SGPRs still waited ~40 clocks per FMA
// for scalars out of a 5-MB table.  So: v_mfma_f64_16x16x4_f64, both operands from vector memory.
//
//   wave = 64 patches (4 blocks of 16) x CH <= 96 functions (FB <= 6 blocks of 16): up to 24 accumulator blocks of 4 doubles
//   per lane.  The set's ceil(NP / 16) blocks are dealt to the fewest chunks of at most 6 as evenly as they go (325 functions =
//   21 blocks = 6 + 5 + 5 + 5: 336 columns of arithmetic where four chunks of 96 do 384), the wider chunks first;
//   patches move as in zk_sep_patches.hip: a run = 64 B (16 float32 / 8 float64 pixels, a piece of a disk row) of all 64
//   patches = 4 KiB, moved by 4 global_load_lds_dwordx4 (each: the run of 16 patches), the granule a lane fetches rotated by
//   its patch index; a wave's slab holds three runs and a run's image is re-armed, two runs ahead, as soon as its steps have
//   read it;
//   k dimension = pixels, 4 per MFMA: A[i][k] = pixel (4 step + k) of patch 16 pb + i (LDS, converted to float64),
//   B[k][j] = table row of that pixel, function 16 fb + j ([unit][pixel slot][CH] float64 in global memory, the caller's values
//   / area, zero rows where a run overlaps its neighbour or leaves the disk -- every disk pixel is owned by exactly one slot,
//   checked at plan creation; steps whose four rows are all zero are skipped).  The rows reach the waves through LDS: a
//   workgroup's waves walk the runs together and share one copy (below);
//   the functions are done CH at a time, every chunk over the same patches, all chunks in one launch (workgroup = (chunk,
//   256 patches)); results go straight into the (N, n_poly) rows (16 consecutive functions per 128-B segment).
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "zk_fold.h"

#define ZK_DIRECT_FB_MAX 6  // function blocks of 16 per chunk: 24 accumulator blocks = 192 registers, two waves per SIMD

struct zk_direct_unit {
  int32_t run_off[4];  // byte offsets of the unit's four runs inside a patch
  int32_t steps;       // bit s: step s (4 pixel slots) has a non-zero table row
  uint32_t own[2];     // bit 4 s + k: slot k of step s stands for a disk pixel (the others contribute an exact zero, whatever
                       // the pixel holds: a NaN outside the disk, or in the overlap of two runs, never reaches a moment)
  int32_t pad;
};

struct zk_direct_tables {
  int n_chunks = 0;
  int fb_hi = ZK_DIRECT_FB_MAX;  // chunks 0 .. n_hi - 1 hold fb_hi blocks of 16 functions, the other n_chunks - n_hi hold fb_hi - 1
  int n_hi = 0;
  int chunk_fb(int c) const { return c < n_hi ? fb_hi : fb_hi - 1; }
  int chunk_col0(int c) const { return 16 * (c < n_hi ? c * fb_hi : n_hi * fb_hi + (c - n_hi) * (fb_hi - 1)); }
  struct per_type {
    int n_units = 0;
    zk_direct_unit* d_units = nullptr;
    double* d_tab = nullptr;  // chunk after chunk: [n_units][4 runs x UP slots][CH of the chunk]
  } t[2];                     // [0] float32 (UP = 16 pixels per run), [1] float64 (UP = 8)
  // dense mode: the disk rows in pieces of 4 consecutive pixels (a piece = one MFMA step)
  int n_steps = 0;
  int tile_pitch = 0;
  bool flipped = false;           // the dense table holds (-1)^n V(K-1-r, K-1-c): what the reference's convolution multiplies
  int32_t* d_step_off = nullptr;  // [n_steps] bits 0..23: tile element offset of the piece's first pixel (window row r:
                                  // r * tile_pitch + c); bits 24..27: which of its 4 slots stand for a disk pixel
  double* d_ftab = nullptr;       // chunk after chunk: [n_steps][4][CH of the chunk]
};

namespace {


#define ZK_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define ZK_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <typename T>
int upload(T** dst, const std::vector<T>& src) {
  if (src.empty()) return 0;
  ZK_HIP(hipMalloc((void**)dst, src.size() * sizeof(T)));
  ZK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

// Four pixels (one per block of 16 patches: 1 KiB apart) of run image RHO (the images are 4 KiB apart; addr = the pixel's place
// in image 0), read by instructions the compiler does not know to be LDS reads: it makes every LDS read it knows of wait for
// ALL LDS DMA in flight (s_waitcnt vmcnt(0): it cannot tell the pieces of the slab apart), which would put the wait for a
// piece's re-arm DMA right behind its issue.
struct zk_px4_f32 {
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 lo, hi;
  template <int RHO>
  __device__ __forceinline__ void issue_at(unsigned addr) {
    asm volatile("ds_read2st64_b32 %0, %2 offset0:%3 offset1:%4\n\tds_read2st64_b32 %1, %2 offset0:%5 offset1:%6"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(addr), "n"(16 * RHO), "n"(16 * RHO + 4), "n"(16 * RHO + 8), "n"(16 * RHO + 12));
  }
  __device__ __forceinline__ void wait() { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi)); }
  __device__ __forceinline__ double get(int pb) const { return (double)(pb < 2 ? lo[pb & 1] : hi[pb & 1]); }
};
struct zk_px4_f64 {
  typedef double v2 __attribute__((ext_vector_type(2)));
  v2 lo, hi;
  template <int RHO>
  __device__ __forceinline__ void issue_at(unsigned addr) {
    asm volatile("ds_read2st64_b64 %0, %2 offset0:%3 offset1:%4\n\tds_read2st64_b64 %1, %2 offset0:%5 offset1:%6"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(addr), "n"(8 * RHO), "n"(8 * RHO + 2), "n"(8 * RHO + 4), "n"(8 * RHO + 6));
  }
  __device__ __forceinline__ void wait() { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi)); }
  __device__ __forceinline__ double get(int pb) const { return pb < 2 ? lo[pb & 1] : hi[pb & 1]; }
};
template <typename TIN> struct zk_px4;
template <> struct zk_px4<float> { typedef zk_px4_f32 type; };
template <> struct zk_px4<double> { typedef zk_px4_f64 type; };
__device__ __forceinline__ unsigned zk_lds_offset(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }

// ---- the batch kernel ------------------------------------------------------------------------------------------------------
// Round 3's form had every wave fetch a step's table rows from L2 for itself, right before the step's MFMAs.  Those loads count
// on vmcnt behind the slab's re-arm DMA (in order), so every re-arm's HBM latency sat in front of a row load: 0.60-0.71 of the
// FP64 peak where the arithmetic alone, re-arm and row loads compiled out, ran at 0.75-0.86 (profiles/r04_direct_batch.txt).
// Here a workgroup's four waves (one per SIMD; two workgroups per CU) walk the runs together: the rows of run r + 1 arrive by DMA
// (each wave moves a share) while run r is computed, one barrier per run hands them over, and a step's operands are LDS reads
// issued a step ahead -- vmcnt counts DMA only, so "this run has landed" is an exact count.
//   table piece of a run in LDS: [step pair][function block][step of the pair][k][16 functions] float64 = 512 B per (step,
//   block), which lane (k, j) = lane 16 k + j reads at 8 lane: one conflict-free ds_read_b64 per MFMA B operand;
//   a 1-KiB DMA instruction = one function block of a step pair; two pieces (run r, run r + 1) behind the four 12-KiB slabs
//   (three run images each: the run being read and the next two on their way): 72 KiB per workgroup at 96 functions.
template <int I, int N, typename F>
__device__ __forceinline__ void zk_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    zk_static_for<I + 1, N>(f);
  }
}

template <typename TIN, int FB>
__device__ __forceinline__ void zk_patch_direct_chunk(const TIN* __restrict__ in, double* __restrict__ out,
                                                      const zk_direct_unit* __restrict__ units, const double* __restrict__ tab, int n_units,
                                                      int col0, unsigned patch_block, int n_poly, long long n_patches, int patch_bytes) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  constexpr int CH = 16 * FB;
  constexpr int PXG = 16 / sizeof(TIN);  // steps per run: 4 (float32) or 2 (float64)
  constexpr int UP = 4 * PXG;            // pixels per run
  constexpr int SP = PXG / 2;            // step pairs per run
  constexpr int PIECE = SP * FB * 1024;  // bytes of a run's table rows
  constexpr int N_INSTR = SP * FB;       // DMA instructions that move a piece
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  constexpr int WAVES = 4;
  char* const ring = (char*)lds + WAVES * 12288;  // the waves' slabs (three 4-KiB run images each), then the two table pieces
  const int n_live = n_poly - col0 < CH ? n_poly - col0 : CH;
  const long long patch0 = ((long long)patch_block * WAVES + wave) * 64;
  const long long left = n_patches - patch0;
  // a wave past the end of the batch keeps the workgroup's barriers and its share of the table DMA: it reads patch 0, stores nothing
  const int nv = left <= 0 ? 0 : left < 64 ? (int)left : 64;

  const int a = lane >> 2, b = lane & 3;
  const int g0 = (b - (a >> 2)) & 3;  // source granule: rotated by the patch index
  const char* const wbase = (const char*)in + (nv > 0 ? patch0 : 0) * patch_bytes;
  int poff[4];
#pragma unroll
  for (int pg = 0; pg < 4; ++pg) {
    int pi = pg * 16 + a;
    pi = pi < nv ? pi : (nv > 0 ? nv - 1 : 0);
    poff[pg] = pi * patch_bytes + g0 * 16;
  }
  const ZK_CONST int32_t* utab = zk_const((const int32_t*)units);  // 8 ints per unit
  auto issue_run = [&](int u, int rho, int image) {  // run rho of unit u into run image `image` of the wave's slab
    const int ro = utab[8 * u + rho];
#pragma unroll
    for (int pg = 0; pg < 4; ++pg)
      __builtin_amdgcn_global_load_lds(ZK_GLOBAL_PTR(wbase + (poff[pg] + ro)), ZK_LDS_PTR(lds + wave * 3072 + (image * 4 + pg) * 256), 16,
                                       0, 0);
  };
  // table DMA: lane -> (step of the pair h, k, granule g of the block's 128 B)
  const int t_off = ((((lane >> 5) * 4 + ((lane >> 3) & 3)) * CH) + 2 * (lane & 7)) * 8;
  const char* const tbytes = (const char*)tab;
  auto issue_table = [&](int r) {
#pragma unroll
    for (int i0 = 0; i0 < N_INSTR; i0 += WAVES) {
      const int i = i0 + wave;
      if (i < N_INSTR) {
        const int sp = i / FB, fb = i - sp * FB;
        __builtin_amdgcn_global_load_lds(ZK_GLOBAL_PTR(tbytes + ((size_t)r * UP * CH + sp * 8 * CH + 16 * fb) * 8 + t_off),
                                         ZK_LDS_PTR(ring + (r & 1) * PIECE + i * 1024), 16, 0, 0);
      }
    }
  };

  const int li = lane & 15, kr = lane >> 4;
  const int prot = li >> 2;
  unsigned px_addr[PXG];  // LDS byte address of (patch li, this lane's slot of step sr) in run image 0
#pragma unroll
  for (int sr = 0; sr < PXG; ++sr) {
    const int x = 4 * sr + kr;
    const int gsl = ((x / PXG + prot) & 3) * PXG + x % PXG;
    px_addr[sr] = zk_lds_offset(lds + wave * 3072) + (unsigned)((li * UP + gsl) * sizeof(TIN));
  }
  const unsigned b_addr = zk_lds_offset(ring) + 8u * lane;
  v4d acc[4][FB];
#pragma unroll
  for (int pb = 0; pb < 4; ++pb)
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) acc[pb][fb] = v4d{0.0, 0.0, 0.0, 0.0};

  // as if the tops of runs -2, -1 had been passed: a run's top issues the table of the next run, then the patches' run 2 ahead into
  // the image that the run before it has just left (run r lives in image r mod 3)
  const int n_runs = 4 * n_units;
  issue_run(0, 0, 0);
  issue_table(0);
  issue_run(0, 1, 1);
  int image = 0;  // r mod 3
  for (int u = 0; u < n_units; ++u) {
    const int steps = utab[8 * u + 4];
    const unsigned own_lo = (unsigned)utab[8 * u + 5], own_hi = (unsigned)utab[8 * u + 6];
    zk_static_for<0, 4>([&](auto rho_c) {
      constexpr int RHO = decltype(rho_c)::value;
      const int r = 4 * u + RHO;
      // everything but the youngest four DMA instructions (the patches' run r + 1, if there is one) has landed: this wave's share
      // of run r's rows and its patches' run r
      if (r + 1 < n_runs)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave's share has, and every wave is done with run r - 1 (its rows' piece is free)
      if (r + 1 < n_runs) issue_table(r + 1);
      if (r + 2 < n_runs) issue_run(RHO < 2 ? u : u + 1, (RHO + 2) & 3, image == 0 ? 2 : image - 1);
      const unsigned image_at = 4096u * image;
      image = image == 2 ? 0 : image + 1;
      // a step's operands are read (into a second set of registers) before the step ahead of it multiplies
      typename zk_px4<TIN>::type px[2];
      double bv[2][FB];
      auto read_step = [&](auto sr_c) {
        constexpr int SR = decltype(sr_c)::value;
        px[SR & 1].template issue_at<0>(px_addr[SR] + image_at);
        zk_static_for<0, FB>([&](auto fb_c) {
          constexpr int F = decltype(fb_c)::value;
          double row;  // (an asm operand may not name a variable of the enclosing lambda)
          asm volatile("ds_read_b64 %0, %1 offset:%2"
                       : "=v"(row)
                       : "v"(b_addr), "n"((RHO & 1) * PIECE + ((SR / 2) * FB + F) * 1024 + (SR & 1) * 512));
          bv[SR & 1][F] = row;
        });
      };
      read_step(std::integral_constant<int, 0>{});
      zk_static_for<0, PXG>([&](auto sr_c) {
        constexpr int SR = decltype(sr_c)::value;
        constexpr int ST = RHO * PXG + SR;
        constexpr int B = SR & 1;
        px[B].wait();  // (LDS reads return in order: this step's are older than any of the next step's)
        if constexpr (FB == 6)
          asm volatile("" : "+v"(bv[B][0]), "+v"(bv[B][1]), "+v"(bv[B][2]), "+v"(bv[B][3]), "+v"(bv[B][4]), "+v"(bv[B][5]));
        else if constexpr (FB == 5)
          asm volatile("" : "+v"(bv[B][0]), "+v"(bv[B][1]), "+v"(bv[B][2]), "+v"(bv[B][3]), "+v"(bv[B][4]));
        else if constexpr (FB == 4)
          asm volatile("" : "+v"(bv[B][0]), "+v"(bv[B][1]), "+v"(bv[B][2]), "+v"(bv[B][3]));
        else
          asm volatile("" : "+v"(bv[B][0]), "+v"(bv[B][1]), "+v"(bv[B][2]));
        if constexpr (SR + 1 < PXG) read_step(std::integral_constant<int, SR + 1>{});
        if ((steps >> ST) & 1) {  // wave-uniform, and the same in every wave: a dead step has four zero rows
          const bool mine = (((ST < 8 ? own_lo : own_hi) >> (4 * (ST & 7) + kr)) & 1u) != 0;
          double av[4];
#pragma unroll
          for (int pb = 0; pb < 4; ++pb) av[pb] = mine ? px[B].get(pb) : 0.0;
#pragma unroll
          for (int pb = 0; pb < 4; ++pb)
#pragma unroll
            for (int fb = 0; fb < FB; ++fb) acc[pb][fb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[pb], bv[B][fb], acc[pb][fb], 0, 0, 0);
        }
      });
    });
  }
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

// One launch = every chunk over all patches, chunk after chunk (workgroup index = chunk * blocks_per_chunk + patch block): a small
// batch fills as many CUs as it has chunks times workgroups, and no chunk waits for the tail of the one before it.  The first n_hi
// chunks hold FB blocks of 16 functions, the others FB - 1 (their tables follow the wider ones').
template <typename TIN, int FB>
__global__ __launch_bounds__(256, 2) void zk_patch_direct_kernel(const TIN* __restrict__ in, double* __restrict__ out,
                                                                 const zk_direct_unit* __restrict__ units,
                                                                 const double* __restrict__ tab, int n_units, int n_hi,
                                                                 unsigned blocks_per_chunk, int n_poly, long long n_patches,
                                                                 int patch_bytes) {
  constexpr int UP = 64 / (int)sizeof(TIN);
  const int chunk = (int)(blockIdx.x / blocks_per_chunk);
  const unsigned patch_block = blockIdx.x - (unsigned)chunk * blocks_per_chunk;
  const size_t rows = (size_t)n_units * (4 * UP);  // table rows of a chunk
  if (chunk < n_hi)
    zk_patch_direct_chunk<TIN, FB>(in, out, units, tab + (size_t)chunk * rows * (16 * FB), n_units, chunk * 16 * FB, patch_block, n_poly,
                                   n_patches, patch_bytes);
  else
    zk_patch_direct_chunk<TIN, FB - 1>(in, out, units, tab + (size_t)n_hi * rows * (16 * FB) + (size_t)(chunk - n_hi) * rows * (16 * (FB - 1)),
                                       n_units, n_hi * 16 * FB + (chunk - n_hi) * 16 * (FB - 1), patch_block, n_poly, n_patches, patch_bytes);
}

template <typename TIN, int FB>
int launch_fb(zk_plan* p, int64_t n, const TIN* src, double* out, const zk_direct_tables::per_type& t, int patch_bytes, hipStream_t s) {
  const zk_direct_tables* d = p->direct;
  constexpr int PIECE = (int)(16 / sizeof(TIN) / 2) * FB * 1024;
  const int lds = 4 * 12288 + 2 * PIECE;  // <= 72 KiB: two workgroups per CU
  auto kern = zk_patch_direct_kernel<TIN, FB>;
  ZK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const unsigned blocks = (unsigned)((n + 255) / 256);
  int rc = zk_prof_begin(p, s);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3(blocks * (unsigned)d->n_chunks), dim3(256), lds, s, src, out, t.d_units, t.d_tab, t.n_units, d->n_hi, blocks,
                     p->n_poly, (long long)n, patch_bytes);
  ZK_HIP(hipGetLastError());
  return zk_prof_end(p, s);
}

template <typename TIN>
int launch_t(zk_plan* p, const void* in, int64_t n_patches, double* out, hipStream_t s) {
  const zk_direct_tables* d = p->direct;
  const zk_direct_tables::per_type& t = d->t[sizeof(TIN) == 4 ? 0 : 1];
  const size_t patch_elems = (size_t)p->size * p->size;
  const int pb = (int)(patch_elems * sizeof(TIN));
  // (every chunk streams the patches again: 8 x 20.7 KB per patch at (72, 36) against 5.5 MFLOP -- a quarter of the arithmetic's
  //  time at the HBM rate, and it overlaps)
  const int64_t round_max = (int64_t)1 << 20;  // (x up to 11 chunks: the grid stays far below 2^31 workgroups)
  for (int64_t first = 0; first < n_patches; first += round_max) {
    const int64_t n = std::min<int64_t>(n_patches - first, round_max);
    const TIN* src = (const TIN*)in + first * patch_elems;
    double* o = out + first * p->n_poly;
    int rc;
    switch (d->fb_hi) {
      case 6: rc = launch_fb<TIN, 6>(p, n, src, o, t, pb, s); break;
      case 5: rc = launch_fb<TIN, 5>(p, n, src, o, t, pb, s); break;
      case 4: rc = launch_fb<TIN, 4>(p, n, src, o, t, pb, s); break;
      default: return zk_fail(ZK_E_BADARG, "internal: direct chunk width");
    }
    if (rc) return rc;
  }
  return 0;
}

// Dense mode of the same sum: 8 output rows x 64 columns per workgroup, the zero-padded tile staged once in LDS (in the
// image's own element type) and walked once per chunk; wave = one output row = 64 positions (4 blocks of 16) x CH functions.
// A step = 4 consecutive pixels of a disk row: A[i][k] = tile[window origin of position 16 pb + i + piece offset + k].
template <typename TIN, int F>
__device__ __forceinline__ void zk_frame_direct_chunk(const TIN* __restrict__ mine, const ZK_CONST int32_t* soff, const double* __restrict__ tl,
                                                      int n_steps, int kr, double* __restrict__ dst0, int n_live, int li, bool row_live,
                                                      int cols_left, long long plane) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  if constexpr (F >= 1) {
    constexpr int CH = 16 * F;
    v4d acc[4][F];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb)
#pragma unroll
      for (int fb = 0; fb < F; ++fb) acc[pb][fb] = v4d{0.0, 0.0, 0.0, 0.0};
    for (int st = 0; st < n_steps; ++st) {
      const int so = soff[st];
      const TIN* __restrict__ px = mine + (so & 0xffffff);
      const bool own = ((so >> (24 + kr)) & 1) != 0;
      double av[4], bv[F];
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) {
        const double v = (double)px[16 * pb];
        av[pb] = own ? v : 0.0;
      }
      const double* __restrict__ tr = tl + (size_t)st * 4 * CH;
#pragma unroll
      for (int fb = 0; fb < F; ++fb) bv[fb] = tr[16 * fb];
#pragma unroll
      for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int fb = 0; fb < F; ++fb) acc[pb][fb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[pb], bv[fb], acc[pb][fb], 0, 0, 0);
    }
    if (row_live) {
#pragma unroll
      for (int fb = 0; fb < F; ++fb) {
        const bool col_live = 16 * fb + li < n_live;
        double* __restrict__ dst = dst0 + (long long)(16 * fb) * plane;
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ok = 16 * pb + kr + 4 * q;
            if (col_live && ok < cols_left) dst[ok] = acc[pb][fb][q];
          }
      }
    }
  }
}

// FB = blocks of 16 functions in the first n_hi chunks; the other n_chunks - n_hi hold FB - 1
template <typename TIN, int FB>
__global__ __launch_bounds__(512) void zk_frame_direct_kernel(const TIN* __restrict__ img, double* __restrict__ out,
                                                               const int32_t* __restrict__ step_off, const double* __restrict__ tab,
                                                               int n_steps, int n_chunks, int n_hi, int n_poly, int K, int H, int W,
                                                               int row0, int n_rows, int tile_pitch, long long plane) {
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
  double* __restrict__ const orow = out + (long long)(oi - row0) * W + k0;
  const double* __restrict__ tl = tab + (size_t)kr * (16 * FB) + li;
  int col0 = 0;
  for (int ch = 0; ch < n_hi; ++ch) {
    zk_frame_direct_chunk<TIN, FB>(mine, soff, tl, n_steps, kr, orow + (long long)(col0 + li) * plane, n_poly - col0, li, row_live, W - k0, plane);
    tl += (size_t)n_steps * 4 * (16 * FB);
    col0 += 16 * FB;
  }
  tl -= kr * 16;  // the narrower chunks' rows are 16 (FB - 1) wide
  for (int ch = n_hi; ch < n_chunks; ++ch) {
    zk_frame_direct_chunk<TIN, FB - 1>(mine, soff, tl, n_steps, kr, orow + (long long)(col0 + li) * plane, n_poly - col0, li, row_live, W - k0,
                                       plane);
    tl += (size_t)n_steps * 4 * (16 * (FB - 1));
    col0 += 16 * (FB - 1);
  }
}

template <typename TIN, int FB>
int launch_frame_fb(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out, hipStream_t s) {
  const zk_direct_tables* d = p->direct;
  const size_t lds = (size_t)(p->size + 7) * d->tile_pitch * sizeof(TIN);
  auto kern = zk_frame_direct_kernel<TIN, FB>;
  if (lds > 64 * 1024)
    ZK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long plane = zk_out_plane(p, n_rows, W);
  return zk_for_row_bands(row0, n_rows, W, 8, [&](int64_t r0, int64_t nr, long long off) {
    dim3 grid((unsigned)((W + 63) / 64), (unsigned)((nr + 7) / 8));
    int rc = zk_prof_begin(p, s);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, (const TIN*)in, out + off, d->d_step_off, d->d_ftab, d->n_steps, d->n_chunks,
                       d->n_hi, p->n_poly, p->size, (int)H, (int)W, (int)r0, (int)nr, d->tile_pitch, plane);
    ZK_HIP(hipGetLastError());
    return zk_prof_end(p, s);
  });
}

template <typename TIN>
int launch_frame_t(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out, hipStream_t s) {
  switch (p->direct->fb_hi) {
    case 6: return launch_frame_fb<TIN, 6>(p, in, H, W, row0, n_rows, out, s);
    case 5: return launch_frame_fb<TIN, 5>(p, in, H, W, row0, n_rows, out, s);
    case 4: return launch_frame_fb<TIN, 4>(p, in, H, W, row0, n_rows, out, s);
    default: return zk_fail(ZK_E_BADARG, "internal: direct chunk width");
  }
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
  // chunks: the set's blocks of 16 functions dealt to the fewest chunks of at most 6, as evenly as they go (ZK_DIRECT_CH96=1:
  // round 3's chunks of 96 whatever the set, for comparison)
  {
    const int blocks = (NP + 15) / 16;
    d->n_chunks = (blocks + ZK_DIRECT_FB_MAX - 1) / ZK_DIRECT_FB_MAX;
    d->fb_hi = (blocks + d->n_chunks - 1) / d->n_chunks;
    d->n_hi = blocks % d->n_chunks ? blocks % d->n_chunks : d->n_chunks;
    if (getenv("ZK_DIRECT_CH96")) d->fb_hi = ZK_DIRECT_FB_MAX, d->n_hi = d->n_chunks;
  }
  size_t cols = 0;  // columns of all chunks together
  for (int c = 0; c < d->n_chunks; ++c) cols += 16 * d->chunk_fb(c);
  for (int dt = 0; dt < 2; ++dt) {
    const int es = dt == 0 ? 4 : 8, UP = 64 / es;
    // runs: the disk segment [lo, hi) of every row in pieces of UP pixels; a piece at the patch's very end is moved back so that it
    // ends inside the patch (it then overlaps its neighbour: the overlapped slots own nothing)
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
        // (a piece starts where the one before it ended, running on into the next row's bytes if it must -- its steps then stay
        //  aligned with the row's first disk pixel: 306 instead of 318 steps at 40 px; only a piece that would leave the PATCH is
        //  moved back to end with it)
        const int start = r * K + c0 + UP <= K * K ? c0 : K - UP;
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
    std::vector<double> tab(owner.size() * cols + 4 * 16 * ZK_DIRECT_FB_MAX, 0.0);  // (+ one step: the kernels fetch rows one step ahead)
    for (int c = 0; c < d->n_chunks; ++c) {
      const int CH = 16 * d->chunk_fb(c), col0 = d->chunk_col0(c);
      double* const chunk = &tab[owner.size() * col0];  // (the chunks before this one hold col0 columns of every slot)
      for (size_t k = 0; k < owner.size(); ++k) {
        if (owner[k] < 0) continue;
        double* dst = chunk + k * CH;
        for (int i = 0; i < CH && col0 + i < NP; ++i) dst[i] = basis[(size_t)(col0 + i) * K * K + owner[k]] * inv_area;
      }
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
    std::vector<double> ftab(owner.size() * cols + 4 * 16 * ZK_DIRECT_FB_MAX, 0.0);
    for (int c = 0; c < d->n_chunks; ++c) {
      const int CH = 16 * d->chunk_fb(c), col0 = d->chunk_col0(c);
      double* const chunk = &ftab[owner.size() * col0];
      for (size_t k = 0; k < owner.size(); ++k) {
        if (owner[k] < 0) continue;
        double* dst = chunk + k * CH;
        for (int i = 0; i < CH && col0 + i < NP; ++i) dst[i] = value(col0 + i, owner[k]) * inv_area;
      }
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
