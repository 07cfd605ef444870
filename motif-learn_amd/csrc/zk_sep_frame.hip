// zk_sep_frame.hip -- dense-frame Zernike moments (reference _zps.py:159-193), row-separable form.
//
// Same decomposition as zk_fast_frame.hip: a 256-thread workgroup owns 4 rows x 64 columns of output
// pixels, one output pixel per lane, the zero-padded (K+3) x (K+63) image tile staged once in LDS as
// float64.  The arithmetic is the row-separable sum of zk_sep.h: per quadrant pixel 4 conflict-free
// ds_read_b64, 8 v_add_f64 for the mirror folds and 2(n_max+1) v_fma_f64 whose Legendre operand is an
// SGPR pair; per disk row N_poly v_fma_f64; one class-blocked T product at the end.  All scalar tables
// together are ~7 KiB at (32, 8) and stay in the scalar data cache.
//
// Roofline: FP64-VALU-bound; algorithmic HBM bytes s_in + 8 N_poly per pixel.
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

// n_max 11-12 needs 264 registers; capped at 256 (8 spilled outside the pixel loop) two waves share a SIMD:
// 2.65 -> 1.91 ms per 2048^2 at (32, 12), 7.32 -> 5.43 ms at (64, 12)
#ifndef ZK_FRAME_W12
#define ZK_FRAME_W12 2
#endif
#define ZK_FRAME_WAVES(NMAX) ((NMAX) == 12 ? ZK_FRAME_W12 : 1)

namespace {

template <int NMAX, typename T, int MASK>
__global__ __launch_bounds__(256, ZK_FRAME_WAVES(NMAX)) void zk_frame_sep_kernel(
    const T* __restrict__ img, double* __restrict__ out, const zk_sep_row* __restrict__ rows,
    const double* __restrict__ xq, const double* __restrict__ tmat, const int32_t* __restrict__ colmap,
    int n_tab_rows, int K, int H, int W, int row0, int n_rows, int tile_pitch, long long plane) {
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

  zk_sep_acc<NMAX, MASK> acc;  // MASK: the parity classes this launch computes (all, or one per pass)
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

  const int oi = i0 + wave;
  const int ok = k0 + lane;
  const bool live = oi < row0 + n_rows && ok < W;
  double* __restrict__ dst = out + (long long)(oi - row0) * W + ok;
  const ZK_CONST int32_t* cmap = zk_const(colmap);
  acc.transform(zk_const(tmat), [&](auto slot, double z) {
    const int col = cmap[slot];
    if (live && col >= 0) dst[col * plane] = z;
  });
}

template <int NMAX, typename T, int MASK = 15>
int launch_one(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out,
               hipStream_t s) {
  const zk_sep_tables* t = p->sep;
  const size_t lds = (size_t)(p->size + 3) * t->tile_pitch * sizeof(double);
  auto kern = zk_frame_sep_kernel<NMAX, T, MASK>;
  if (lds > 64 * 1024)
    ZK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long plane = zk_out_plane(p, n_rows, W);
  return zk_for_row_bands(row0, n_rows, W, 4, [&](int64_t r0, int64_t nr, long long off) {
    dim3 grid((unsigned)((W + 63) / 64), (unsigned)((nr + 3) / 4));
    int rc = zk_prof_begin(p, s);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const T*)in, out + off, t->d_rows, t->d_xq, t->d_T, t->d_colmap,
                       t->n_rows, p->size, (int)H, (int)W, (int)r0, (int)nr, t->tile_pitch, plane);
    ZK_HIP(hipGetLastError());
    return zk_prof_end(p, s);
  });
}

// n_max > 16: one launch per parity class (a quarter of the accumulators each; every launch writes the
// output planes of its own class)
template <int NMAX, typename T>
int launch_passes(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out,
                  hipStream_t s) {
  int rc = launch_one<NMAX, T, 1 << ZK_EE>(p, in, H, W, row0, n_rows, out, s);
  if (!rc) rc = launch_one<NMAX, T, 1 << ZK_OE>(p, in, H, W, row0, n_rows, out, s);
  if (!rc) rc = launch_one<NMAX, T, 1 << ZK_EO>(p, in, H, W, row0, n_rows, out, s);
  if (!rc) rc = launch_one<NMAX, T, 1 << ZK_OO>(p, in, H, W, row0, n_rows, out, s);
  return rc;
}

template <typename T>
int launch_t(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out,
             hipStream_t s) {
  switch (p->sep->kernel_nmax) {
#if ZK_NMAX_GROUP == 0
    case 4: return launch_one<4, T>(p, in, H, W, row0, n_rows, out, s);
    case 6: return launch_one<6, T>(p, in, H, W, row0, n_rows, out, s);
    case 8: return launch_one<8, T>(p, in, H, W, row0, n_rows, out, s);
    case 10: return launch_one<10, T>(p, in, H, W, row0, n_rows, out, s);
    case 12: return launch_one<12, T>(p, in, H, W, row0, n_rows, out, s);
#endif
#if ZK_NMAX_GROUP == 1
    // 330-400 registers in one pass (one wave per SIMD) against 4 class passes at two waves per SIMD that redo
    // the folds and the tile staging: the passes win while the T product and the row ends weigh enough, i.e. for
    // small windows (1024^2 frame, ms: (32,14) 1.07 -> 0.78, (48,14) 1.69 -> 1.51, (72,14) 2.92 -> 3.51;
    // (32,16) 1.54 -> 0.93, (48,16) 2.12 -> 1.68, (72,16) 3.45 -> 3.74)
    case 14: return p->size <= 56 ? launch_passes<14, T>(p, in, H, W, row0, n_rows, out, s) : launch_one<14, T>(p, in, H, W, row0, n_rows, out, s);
    case 16: return p->size <= 56 ? launch_passes<16, T>(p, in, H, W, row0, n_rows, out, s) : launch_one<16, T>(p, in, H, W, row0, n_rows, out, s);
#endif
#if ZK_NMAX_GROUP == 2
    case 20: return launch_passes<20, T>(p, in, H, W, row0, n_rows, out, s);
    case 24: return launch_passes<24, T>(p, in, H, W, row0, n_rows, out, s);
#endif
  }
  return zk_fail(ZK_E_BADARG, "no separable frame kernel for this n_max");
}

}  // namespace

#if ZK_NMAX_GROUP == 0
bool zk_sep_frame_available(const zk_plan* p, int dtype) {
  (void)dtype;
  if (!p->sep || p->sep->n_rows == 0) return false;
  return (size_t)(p->size + 3) * p->sep->tile_pitch * sizeof(double) <= 160 * 1024;
}
int zk_launch_sep_frame_g1(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0, int64_t n_rows,
                           double* out, hipStream_t s);
int zk_launch_sep_frame_g2(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0, int64_t n_rows,
                           double* out, hipStream_t s);
#endif

int ZK_GROUP_FN(zk_launch_sep_frame)(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0,
                                     int64_t n_rows, double* out, hipStream_t s) {
#if ZK_NMAX_GROUP == 0
  if (p->sep->kernel_nmax > 16) return zk_launch_sep_frame_g2(p, in, dtype, H, W, row0, n_rows, out, s);
  if (p->sep->kernel_nmax > 12) return zk_launch_sep_frame_g1(p, in, dtype, H, W, row0, n_rows, out, s);
#endif
  if (dtype == ZK_F32) return launch_t<float>(p, in, H, W, row0, n_rows, out, s);
  return launch_t<double>(p, in, H, W, row0, n_rows, out, s);
}
