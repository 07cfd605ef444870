// zk_sep_stream.hip -- batch-of-patches Zernike moments (reference _zps.py:146-157) for patch sizes whose
// rows are not a whole number of 128-B lines: the patch is read as the contiguous pixel stream it is in
// memory, one 128-B line at a time, whatever K is (float32 / float64, any K >= 8, n_max <= 16).
//
// Same work decomposition as zk_sep_patches.hip -- one wave owns 64 consecutive patches, one patch per
// lane, accumulators in VGPRs, wave-uniform multipliers through scalar loads, LDS-DMA transposition with
// the granule rotation that makes the per-lane ds_read_b128 conflict-free -- but the unit is
//
//   unit    = bytes [128 u, 128 u + 128) of each of the 64 patches = 8 KiB = 8 global_load_lds_dwordx4,
//             each moving one whole line of 8 patches.  Every request is a full line when K*K*s is a
//             multiple of 128 (float32 K % 8 == 0, float64 K % 4 == 0), where the row-pair kernel issues
//             64-B half-line runs; otherwise a request straddles two lines, both of which the neighbouring
//             units use.  Lines without a disk pixel are skipped.
//   slabs   = two 8-KiB slabs per wave, ping-pong: unit k+1 is in flight while unit k is consumed straight
//             from LDS (a granule ahead), unit k+2 is issued into the slab unit k leaves.
//
// Arithmetic: the row-separable sum of zk_sep.h WITHOUT mirror folding (the mirror pixels of a stream
// position live in other lines): per disk pixel n_max v_fma_f64 + one v_add_f64 (P_0 = 1) into the row sums
// X_a, per disk row N_poly v_fma_f64 (M_(a,b) += P_b(y_r) X_a), one class-blocked T product per patch.
// ~1.4x the f64 work of the folded kernel at (32, 8): hidden behind the stream for float64 patches and for
// float32 up to n_max 8, issue-bound above (profiles/r01_stream_sweep.txt).  The position in the
// stream -- row, column, inside the disk or not -- is wave-uniform and lives in SGPRs; a row may be flushed
// in pieces (the sums are linear), so a wave can start at any line of the patch (channel spreading, see
// ZK_ROTATE in zk_sep_patches.hip).
//
// Roofline: algorithmic bytes K*K*s + 8*N_poly per patch; HBM-bound.
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

namespace {

#define ZK_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define ZK_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

#ifndef ZK_STREAM_WG
#define ZK_STREAM_WG 256  // threads per workgroup (128, i.e. 10 waves per CU by LDS, measured slower: 1.28 -> 1.50 ms at (48, 8))
#endif
#define ZK_STREAM_WPB (ZK_STREAM_WG / 64)

template <int NMAX, typename TIN>
__global__ __launch_bounds__(ZK_STREAM_WG, (NMAX <= 12 ? 2 : 1)) void zk_patch_stream_kernel(
    const TIN* __restrict__ in, double* __restrict__ out, const zk_stream_unit* __restrict__ units,
    const zk_stream_row* __restrict__ rows, const double* __restrict__ pfull, const double* __restrict__ tmat,
    const int32_t* __restrict__ colmap, int n_units, int n_poly, long long n_patches, int patch_bytes, int ppp,
    int K, int aligned) {
  using S = zk_sep_set<NMAX>;
  // table rows are packed (P_1 .. P_NMAX, NMAX even): 64 B per column at n_max 8 (a 256-B pitch measured the same)
  constexpr int SROW = ZK_STREAM_ROW(NMAX);
  constexpr int PXG = 16 / sizeof(TIN);  // pixels per 16-B granule: 4 (float32) or 2 (float64)
  typedef TIN gran_t __attribute__((ext_vector_type(PXG)));
  __shared__ __attribute__((aligned(16))) float lds[ZK_STREAM_WPB * 4096];  // 2 x 8 KiB per wave

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* const wl = lds + wave * 4096;
  const long long wave_id = (long long)blockIdx.x * ZK_STREAM_WPB + wave;
  const long long patch0 = wave_id * 64;
  if (patch0 >= n_patches) return;  // wave-uniform; the kernel has no workgroup barrier
  const long long left = n_patches - patch0;
  const int nv = left < 64 ? (int)left : 64;  // live patches of this wave

  // ---- DMA addressing: instruction pg moves one line of patches 8 pg .. 8 pg + 7; lane -> (patch a, slot b)
  const int a = lane >> 3, b = lane & 7;
  const int g0 = (b - (a >> 1)) & 7;  // source granule for patch group 0; rot(patch) = patch >> 1
  const char* const wbase = (const char*)in + patch0 * patch_bytes;
  int poff[8];  // per patch group: byte offset of this lane's patch + its rotated granule
#pragma unroll
  for (int pg = 0; pg < 8; ++pg) {
    int pi = pg * 8 + a;
    pi = pi < nv ? pi : nv - 1;  // tail wave: re-read the last live patch
    poff[pg] = pi * patch_bytes + ((g0 - 4 * pg) & 7) * 16;
  }
  const ZK_CONST int32_t* utab = zk_const((const int32_t*)units);  // 4 ints per unit
  auto issue = [&](int u, int slab) {
    const int bo = utab[4 * u], clamp = utab[4 * u + 3];
    float* const dst = wl + slab * 2048;
    if (clamp) {  // the line crosses the end of the patch: keep every request inside it (those granules
                  // hold no disk pixel, zk_sep.hip checks)
#pragma unroll
      for (int pg = 0; pg < 8; ++pg) {
        const int go = ((g0 - 4 * pg) & 7) * 16;
        int o = go + bo;
        o = o + 16 > patch_bytes ? patch_bytes - 16 : o;
        __builtin_amdgcn_global_load_lds(ZK_GLOBAL_PTR(wbase + (poff[pg] - go + o)), ZK_LDS_PTR(dst + pg * 256), 16, 0, 0);
      }
    } else if (aligned) {  // whole lines read once: non-temporal
#pragma unroll
      for (int pg = 0; pg < 8; ++pg)
        __builtin_amdgcn_global_load_lds(ZK_GLOBAL_PTR(wbase + (poff[pg] + bo)), ZK_LDS_PTR(dst + pg * 256), 16, 0,
                                         ZK_DMA_AUX);
    } else {  // straddled lines are shared with the neighbouring units: keep them in L2
#pragma unroll
      for (int pg = 0; pg < 8; ++pg)
        __builtin_amdgcn_global_load_lds(ZK_GLOBAL_PTR(wbase + (poff[pg] + bo)), ZK_LDS_PTR(dst + pg * 256), 16, 0, 0);
    }
  };

  zk_sep_acc<NMAX> acc;
  acc.clear_moments();
  double X[S::NA];
#pragma unroll
  for (int i = 0; i < S::NA; ++i) X[i] = 0.0;
  const ZK_CONST double* pf = zk_const(pfull);
  const ZK_CONST int32_t* rtab = zk_const((const int32_t*)rows);  // 4 ints per row: ts, te, r, -

  // stream position: the current disk row (flat pixels ts..te, row rr) and the start of the one after it.
  // xcur + t * SROW is the Legendre row of pixel t of the current row (its column is t - rr * K).
  int ri = -1, ts = 0, te = -1, rr = 0, nts = 0, nte = -1, nrr = 0;
  const ZK_CONST double* xcur = pf;
  const ZK_CONST double* const zrow = pf + K * SROW;  // all zeros: pixels outside the disk
  auto load_rows = [&](int i) {
    ri = i;
    ts = rtab[4 * i], te = rtab[4 * i + 1], rr = rtab[4 * i + 2];
    nts = rtab[4 * i + 4], nte = rtab[4 * i + 5], nrr = rtab[4 * i + 6];
    xcur = pf - rr * K * SROW;
  };
  auto next_row = [&]() {
    ++ri;
    ts = nts, te = nte, rr = nrr;
    nts = rtab[4 * ri + 4], nte = rtab[4 * ri + 5], nrr = rtab[4 * ri + 6];
    xcur = pf - rr * K * SROW;
  };

  const int off = (int)((wave_id * 7) % n_units);  // first line of this wave (channel spreading)
  auto unit_at = [&](int k) { return k + off < n_units ? k + off : k + off - n_units; };
  const int lbase = lane * 32, rot = lane >> 1;  // float index of this lane's line image; granule rotation

#if ZK_ABLATE != 2
  issue(unit_at(0), 0);
  if (n_units > 1) issue(unit_at(1), 1);
#endif
  for (int k = 0; k < n_units; ++k) {
    const int u = unit_at(k);
    const int t0 = utab[4 * u + 1], urow = utab[4 * u + 2];
    if (urow != ri) {  // first unit, or the wrap-around of a rotated start: flush the row in progress
      if (ri >= 0) acc.stream_row_end(X, pf + rr * SROW);
      load_rows(urow);
    }
    if (k + 1 < n_units) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // unit k landed, k + 1 in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float* const sl = wl + (k & 1) * 2048 + lbase;
    auto granule = [&](int g) -> gran_t { return *(const gran_t*)(sl + (((g + rot) & 7) << 2)); };
    gran_t nv4 = granule(0);
#pragma unroll 1
    for (int g = 0; g < 8; ++g) {
      const gran_t v = nv4;
      if (g < 7) nv4 = granule(g + 1);
      const int tg = t0 + PXG * g;
      if (tg + PXG - 1 < ts) continue;  // the current row has not begun (te >= tg always holds)
#if ZK_ABLATE == 1
      asm volatile("" ::"v"(v));
      if (false)
#endif
      {
        // The loop body is straight-line code plus ONE conditional block, the row end (with any more
        // control flow around the moment updates the compiler keeps two copies of M and moves them every
        // granule), and scalar instructions are kept to a minimum: a wave issues them at the same rate
        // as its v_fma_f64.  The table rows of the granule's four columns are requested together (one
        // scalar-memory wait per granule) from xcur + tg * ROW onwards, whether or not the pixels are
        // inside the disk -- the table has ZK_STREAM_PAD spare rows either side -- and the pixels that
        // are not (or that lie behind the end of the current row) are zeroed instead, which only happens
        // in the granules where a row begins or ends.
        gran_t vm = v;
        if (tg < ts || tg + PXG - 1 > te) {
#pragma unroll
          for (int e = 0; e < PXG; ++e) vm[e] = (tg + e >= ts && tg + e <= te) ? v[e] : (TIN)0;
        }
        const ZK_CONST double* base = xcur + tg * SROW;
        double xv[PXG][S::NA - 1];
#pragma unroll
        for (int e = 0; e < PXG; ++e)
#pragma unroll
          for (int i = 0; i < S::NA - 1; ++i) xv[e][i] = base[e * SROW + i];
#pragma unroll
        for (int e = 0; e < PXG; ++e) {
          const double f = (double)vm[e];
          X[0] += f;  // P_0 = 1
#pragma unroll
          for (int i = 1; i < S::NA; ++i) X[i] = __builtin_fma(f, xv[e][i - 1], X[i]);
        }
        if (te < tg + PXG) {  // the current row ends in this granule; pixels behind its end may already
                              // belong to the next row (a granule touches at most two rows: a row is longer)
          acc.stream_row_end(X, pf + rr * SROW);
          next_row();
#pragma unroll
          for (int e = 1; e < PXG; ++e) {
            const int t = tg + e;
            const bool in = t >= ts && t <= te;
            const ZK_CONST double* xn = in ? xcur + t * SROW : zrow;
            const double f = (double)(in ? v[e] : (TIN)0);
            X[0] += f;
#pragma unroll
            for (int i = 1; i < S::NA; ++i) X[i] = __builtin_fma(f, xn[i - 1], X[i]);
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // slab consumed: re-arm it with unit k + 2
#if ZK_ABLATE != 2
    if (k + 2 < n_units) issue(unit_at(k + 2), k & 1);
#endif
  }
  acc.stream_row_end(X, pf + rr * SROW);

  // ---- Z = T M, then (patch, column) rows via LDS -> 16-B stores ------------------------------------
  double z[S::NP];
  acc.transform(zk_const(tmat), [&](auto slot, double v) { z[slot] = v; });
  zk_batch_store_rows<S::NP>(z, zk_const(colmap), (double*)wl, out + patch0 * n_poly, lane, nv, n_poly, ppp);
}

template <int NMAX, typename TIN>
int launch_one(zk_plan* p, const void* in, int64_t n_patches, double* out, hipStream_t s) {
  const zk_sep_tables* t = p->sep;
  const zk_sep_tables::stream_tables& st = t->stream[sizeof(TIN) == 4 ? 0 : 1];
  const long long waves = (n_patches + 63) / 64;
  const long long blocks = (waves + ZK_STREAM_WPB - 1) / ZK_STREAM_WPB;
  if (blocks > 0x7fffffffLL) return zk_fail(ZK_E_BADARG, "too many patches for one launch");
  int ppp = 64;
  while (ppp * p->n_poly > 2048) ppp >>= 1;
  int rc = zk_prof_begin(p, s);
  if (rc) return rc;
  hipLaunchKernelGGL((zk_patch_stream_kernel<NMAX, TIN>), dim3((unsigned)blocks), dim3(ZK_STREAM_WG), 0, s, (const TIN*)in, out,
                     st.d_units, st.d_rows, t->d_pfull, t->d_T, t->d_colmap, st.n_units, p->n_poly,
                     (long long)n_patches, p->size * p->size * (int)sizeof(TIN), ppp, p->size,
                     (int)(st.aligned && ((uintptr_t)in & 127) == 0));
  ZK_HIP(hipGetLastError());
  return zk_prof_end(p, s);
}

template <typename TIN>
int launch_t(zk_plan* p, const void* in, int64_t n_patches, double* out, hipStream_t s) {
  switch (p->sep->kernel_nmax) {
#if ZK_NMAX_GROUP == 0
    case 4: return launch_one<4, TIN>(p, in, n_patches, out, s);
    case 6: return launch_one<6, TIN>(p, in, n_patches, out, s);
    case 8: return launch_one<8, TIN>(p, in, n_patches, out, s);
    case 10: return launch_one<10, TIN>(p, in, n_patches, out, s);
    case 12: return launch_one<12, TIN>(p, in, n_patches, out, s);
#endif
#if ZK_NMAX_GROUP == 1
    case 14: return launch_one<14, TIN>(p, in, n_patches, out, s);
    case 16: return launch_one<16, TIN>(p, in, n_patches, out, s);
#endif
  }
  return zk_fail(ZK_E_BADARG, "no stream batch kernel for this n_max");
}

}  // namespace

#if ZK_NMAX_GROUP == 0
int zk_launch_sep_stream_g1(zk_plan* p, const void* in, int dtype, int64_t n_patches, double* out, hipStream_t s);

bool zk_sep_stream_available(const zk_plan* p, int dtype) {
  const zk_sep_tables* t = p->sep;
  // K <= 1024 keeps the 32-bit byte offsets inside a 64-patch group (63 * K * K * 8 < 2^31)
  return t && t->stream[dtype == ZK_F32 ? 0 : 1].n_units > 0 && p->n_poly <= 1024 && p->size <= 1024;
}

// ZK_PATH_AUTO takes this kernel where the row-pair kernel has no whole-line units for the patch size and
// the batch fills the chip: a wave keeps 8-16 KiB in flight here against 16 KiB there, so with fewer waves
// than wave slots (2048) the row-pair kernel is ahead (tools/sweep_batch.py, profiles/r01_stream_sweep.txt).
bool zk_sep_stream_preferred(const zk_plan* p, int dtype, int64_t n_patches) {
  return p->sep && p->sep->stream[dtype == ZK_F32 ? 0 : 1].preferred && n_patches >= 98304;
}

#endif

int ZK_GROUP_FN(zk_launch_sep_stream)(zk_plan* p, const void* in, int dtype, int64_t n_patches, double* out,
                                      hipStream_t s) {
#if ZK_NMAX_GROUP == 0
  if (p->sep->kernel_nmax > 12) return zk_launch_sep_stream_g1(p, in, dtype, n_patches, out, s);
#endif
  if (((uintptr_t)in & (dtype == ZK_F32 ? 3 : 7)) || ((uintptr_t)out & 7))  // element-aligned operands
    return zk_launch_generic_patches(p, in, dtype, n_patches, out, s);
  if (dtype == ZK_F64) return launch_t<double>(p, in, n_patches, out, s);
  return launch_t<float>(p, in, n_patches, out, s);
}
