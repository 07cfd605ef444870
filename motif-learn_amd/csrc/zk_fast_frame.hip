// zk_fast_frame.hip -- dense-frame Zernike moments (reference _zps.py:159-193), parity-folded.
//
// Work decomposition.  A 256-thread workgroup owns a 4-row x 64-column block of output pixels;
// each wave owns one output row segment and each lane one output pixel, so every lane keeps
// the full set of N_poly float64 accumulators in VGPRs and no cross-lane reduction is needed.
// The workgroup stages the (K+3) x (K+63) zero-padded image tile it needs into LDS once,
// converted to float64 (so the inner loop carries no v_cvt), then walks the quadrant pixels
// inside the disk.  Per quadrant pixel a lane reads its four mirror pixels from LDS
// (consecutive lanes -> consecutive addresses, conflict-free), forms the four parity folds
// (8 v_add_f64) and issues N_poly v_fma_f64 whose basis operand is an SGPR pair filled by
// s_load from the wave-uniform table -- the basis never touches LDS or VGPRs.
//
// Roofline.  Algorithmic HBM bytes are s_in + 8*N_poly per output pixel (frame read once,
// every moment written once).  The kernel is FP64-VALU-bound, not HBM-bound:
// (N_poly + 8) v_*_f64 per quadrant pixel and ~K^2*pi/16 quadrant pixels.  DESIGN.md, section 5.
#include "zk_fold.h"

namespace {

template <int NMAX, typename T>
__global__ __launch_bounds__(256) void zk_frame_fold_kernel(
    const T* __restrict__ img, double* __restrict__ out, const int4* __restrict__ fpx_off,
    const double* __restrict__ ftab, const int32_t* __restrict__ colmap, int n_fpx, int K, int H, int W,
    int row0, int n_rows, int tile_pitch, long long plane) {
  using S = zk_set<NMAX>;
  extern __shared__ __attribute__((aligned(16))) double tile[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int ea = K - 1 - (K - 1) / 2;
  const int i0 = row0 + blockIdx.y * 4;   // first output row of the block
  const int k0 = blockIdx.x * 64;         // first output column
  const int tile_rows = K + 3;
  const int tile_elems = tile_rows * tile_pitch;

  zk_stage_tile(tile, img, H, W, i0 - ea, k0 - ea, tile_rows, tile_pitch);
  (void)tile_elems;
  __syncthreads();

  // ---- accumulate -----------------------------------------------------------------------
  double acc[S::NP];
#pragma unroll
  for (int i = 0; i < S::NP; ++i) acc[i] = 0.0;
  const double* __restrict__ mine = tile + wave * tile_pitch + lane;
  const ZK_CONST int32_t* offs = zk_const((const int32_t*)fpx_off);
  const ZK_CONST double* bt = zk_const(ftab);
  for (int t = 0; t < n_fpx; ++t) {
    const double a = mine[offs[4 * t]], b = mine[offs[4 * t + 1]], c = mine[offs[4 * t + 2]],
                 d = mine[offs[4 * t + 3]];
    zk_fold_fma<NMAX>(acc, a, b, c, d, bt);
    bt += S::NP;
  }

  // ---- store: (n_poly, n_rows, W), one coalesced row segment per moment -----------------
  const int oi = i0 + wave;
  const int ok = k0 + lane;
  if (oi < row0 + n_rows && ok < W) {
    double* __restrict__ dst = out + (long long)(oi - row0) * W + ok;
    const ZK_CONST int32_t* cmap = zk_const(colmap);
#pragma unroll
    for (int i = 0; i < S::NP; ++i) {
      const int col = cmap[i];
      if (col >= 0) dst[col * plane] = acc[i];
    }
  }
}

template <int NMAX, typename T>
int launch_one(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out,
               hipStream_t s) {
  const zk_fold_tables* f = p->fold;
  const size_t lds = (size_t)(p->size + 3) * f->tile_pitch * sizeof(double);
  auto kern = zk_frame_fold_kernel<NMAX, T>;
  if (lds > 64 * 1024)
    ZK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long plane = zk_out_plane(p, n_rows, W);
  return zk_for_row_bands(row0, n_rows, W, 4, [&](int64_t r0, int64_t nr, long long off) {
    dim3 grid((unsigned)((W + 63) / 64), (unsigned)((nr + 3) / 4));
    int rc = zk_prof_begin(p, s);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const T*)in, out + off, f->d_fpx_off, f->d_ftab, f->d_colmap,
                       f->n_fpx, p->size, (int)H, (int)W, (int)r0, (int)nr, f->tile_pitch, plane);
    ZK_HIP(hipGetLastError());
    return zk_prof_end(p, s);
  });
}

template <typename T>
int launch_t(zk_plan* p, const void* in, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out,
             hipStream_t s) {
  switch (p->fold->kernel_nmax) {
    case 4: return launch_one<4, T>(p, in, H, W, row0, n_rows, out, s);
    case 6: return launch_one<6, T>(p, in, H, W, row0, n_rows, out, s);
    case 8: return launch_one<8, T>(p, in, H, W, row0, n_rows, out, s);
    case 10: return launch_one<10, T>(p, in, H, W, row0, n_rows, out, s);
    case 12: return launch_one<12, T>(p, in, H, W, row0, n_rows, out, s);
  }
  return zk_fail(ZK_E_BADARG, "no frame kernel for this n_max");
}

}  // namespace

bool zk_fast_frame_available(const zk_plan* p, int dtype) {
  (void)dtype;
  if (!p->fold || p->fold->n_fpx == 0) return false;
  const size_t lds = (size_t)(p->size + 3) * p->fold->tile_pitch * sizeof(double);
  return lds <= 160 * 1024;
}

int zk_launch_fast_frame(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0,
                         int64_t n_rows, double* out, hipStream_t s) {
  if (dtype == ZK_F32) return launch_t<float>(p, in, H, W, row0, n_rows, out, s);
  return launch_t<double>(p, in, H, W, row0, n_rows, out, s);
}
