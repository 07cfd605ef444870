// zk_api.hip -- host side of libzernike_hip.so: plans, tables, dispatch, C ABI (include/zernike_hip.h).
#include <math.h>
#include <string.h>

#include <new>

#include <stdlib.h>

#include "zk_internal.h"

// ------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------
static thread_local std::string g_last_error = "";

int zk_fail(int code, const std::string& what) {
  g_last_error = what;
  return code;
}

int zk_hip_fail(hipError_t e, const char* what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  (void)hipGetLastError();  // clear the sticky error
  return -(int)e;
}

extern "C" const char* zk_last_error_string(void) { return g_last_error.c_str(); }
extern "C" int zk_abi_version(void) { return ZK_ABI_VERSION; }

extern "C" int zk_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e == hipErrorNoDevice) {
    (void)hipGetLastError();
    return 0;
  }
  if (e != hipSuccess) return zk_hip_fail(e, "hipGetDeviceCount");
  return n;
}

// ------------------------------------------------------------------------------------
// profiling: one event pair per launch, resolved lazily
// ------------------------------------------------------------------------------------
int zk_prof_begin(zk_plan* p, hipStream_t s) {
  if (!p->profile) return 0;
  if (p->ev_used + 2 > p->ev_pool.size()) {
    for (int k = 0; k < 2; ++k) {
      hipEvent_t e;
      ZK_HIP(hipEventCreate(&e));
      p->ev_pool.push_back(e);
    }
  }
  ZK_HIP(hipEventRecord(p->ev_pool[p->ev_used], s));
  return 0;
}

int zk_prof_end(zk_plan* p, hipStream_t s) {
  if (!p->profile) return 0;
  ZK_HIP(hipEventRecord(p->ev_pool[p->ev_used + 1], s));
  p->ev_used += 2;
  return 0;
}

extern "C" int zk_plan_profile(zk_plan* p, int enable) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  p->profile = enable != 0;
  return 0;
}

// resolves the recorded event pairs; the first `cap` per-launch times go to ms_out (may be NULL); *n_recorded = how many
// launches were recorded (may exceed cap: the caller sees that its list is incomplete)
static int prof_drain(zk_plan* p, double* ms_out, int64_t cap, int64_t* n_recorded) {
  int64_t w = 0, total = 0;
  for (size_t k = 0; k + 1 < p->ev_used; k += 2) {
    ZK_HIP(hipEventSynchronize(p->ev_pool[k + 1]));
    float ms = 0.f;
    ZK_HIP(hipEventElapsedTime(&ms, p->ev_pool[k], p->ev_pool[k + 1]));
    p->prof_ms += ms;
    p->prof_launches += 1;
    if (ms_out && w < cap) ms_out[w++] = (double)ms;
    ++total;
  }
  p->ev_used = 0;
  if (n_recorded) *n_recorded = total;
  return 0;
}

extern "C" int zk_plan_profile_read(zk_plan* p, int64_t* launches, double* total_ms) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  ZK_ON_PLAN_DEVICE(p);
  const int rc = prof_drain(p, nullptr, 0, nullptr);
  if (rc) return rc;
  if (launches) *launches = p->prof_launches;
  if (total_ms) *total_ms = p->prof_ms;
  p->prof_launches = 0;
  p->prof_ms = 0.0;
  return 0;
}

extern "C" int zk_plan_profile_read_launches(zk_plan* p, double* ms_out, int64_t cap, int64_t* n_out) {
  if (!p || !n_out || cap < 0 || (cap > 0 && !ms_out)) return zk_fail(ZK_E_BADARG, "bad arguments");
  ZK_ON_PLAN_DEVICE(p);
  const int rc = prof_drain(p, ms_out, cap, n_out);
  p->prof_launches = 0;
  p->prof_ms = 0.0;
  return rc;
}

// ------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------
static int build_generic_tables(zk_plan* p, const double* basis) {
  const int K = p->size, NP = p->n_poly;
  const double inv_area = 1.0 / (M_PI * (double)K * (double)K / 4.0);
  std::vector<int2> pix;
  for (int r = 0; r < K; ++r)
    for (int c = 0; c < K; ++c) {
      bool any = false;
      for (int j = 0; j < NP && !any; ++j) any = basis[((size_t)j * K + r) * K + c] != 0.0;
      if (any) pix.push_back(make_int2(r, c));
    }
  p->npx = (int)pix.size();
  p->gen_chunk = NP <= 64 ? 64 : 32;
  const int CH = p->gen_chunk;
  p->n_chunks = (NP + CH - 1) / CH;
  std::vector<double> tab((size_t)p->n_chunks * p->npx * CH, 0.0);
  for (int j = 0; j < NP; ++j) {
    const int c = j / CH, l = j % CH;
    for (int t = 0; t < p->npx; ++t)
      tab[((size_t)c * p->npx + t) * CH + l] =
          basis[((size_t)j * K + pix[t].x) * K + pix[t].y] * inv_area;
  }
  // dense mode of the kernels that sum the caller's own numbers follows the reference's convolution (zk_plan::conv_flip)
  p->conv_flip = !getenv("ZK_NO_CONV_FLIP");
  std::vector<double> sign((size_t)p->n_chunks * CH, 1.0);
  for (int j = 0; j < NP; ++j) sign[j] = (p->conv_flip && (p->n[j] & 1)) ? -1.0 : 1.0;
  ZK_HIP(hipMalloc((void**)&p->d_sign, sign.size() * sizeof(double)));
  ZK_HIP(hipMemcpy(p->d_sign, sign.data(), sign.size() * sizeof(double), hipMemcpyHostToDevice));
  if (p->npx == 0) return 0;
  ZK_HIP(hipMalloc((void**)&p->d_pix, pix.size() * sizeof(int2)));
  ZK_HIP(hipMemcpy(p->d_pix, pix.data(), pix.size() * sizeof(int2), hipMemcpyHostToDevice));
  ZK_HIP(hipMalloc((void**)&p->d_gen_tab, tab.size() * sizeof(double)));
  ZK_HIP(hipMemcpy(p->d_gen_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

extern "C" void zk_plan_destroy(zk_plan* p) {
  if (!p) return;
  zk_device_scope scope(p->device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  zk_fold_free(p);
  zk_sep_free(p);
  zk_direct_free(p);
  for (auto& e : p->trig_cache)
    if (e.dev) (void)hipFree(e.dev);
  for (hipEvent_t e : p->ev_pool) (void)hipEventDestroy(e);
  if (p->d_pix) (void)hipFree(p->d_pix);
  if (p->d_gen_tab) (void)hipFree(p->d_gen_tab);
  if (p->d_sign) (void)hipFree(p->d_sign);
  zk_host_release(p);
  if (p->d_scratch) (void)hipFree(p->d_scratch);
  if (p->d_gather) (void)hipFree(p->d_gather);
  if (p->d_points_tmp) (void)hipFree(p->d_points_tmp);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
}

extern "C" int zk_plan_create(int size, int n_poly, const int32_t* n, const int32_t* m,
                              const double* basis, int device, zk_plan** out) {
  if (!out) return zk_fail(ZK_E_BADARG, "out is null");
  *out = nullptr;
  if (size <= 0 || n_poly <= 0 || !basis || !n || !m)
    return zk_fail(ZK_E_BADARG, "size and n_poly must be positive and basis/n/m non-null");
  if (size > 4096) return zk_fail(ZK_E_BADARG, "size too large");
  const int ndev = zk_device_count();
  if (ndev < 0) return ndev;
  if (ndev == 0) return zk_fail(ZK_E_NODEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return zk_fail(ZK_E_BADARG, "device index out of range");
  zk_plan* p = new (std::nothrow) zk_plan();
  if (!p) return zk_fail(ZK_E_NOMEM, "out of host memory");
  p->size = size;
  p->n_poly = n_poly;
  p->device = device;
  p->n.assign(n, n + n_poly);
  p->m.assign(m, m + n_poly);
  int rc = 0;
  zk_device_scope scope(device);
  hipError_t e = scope.err;
  if (e != hipSuccess) rc = zk_hip_fail(e, "hipSetDevice");
  if (!rc) {
    e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e != hipSuccess) rc = zk_hip_fail(e, "hipStreamCreate");
  }
  if (!rc) {
    e = hipDeviceGetAttribute(&p->n_cu, hipDeviceAttributeMultiprocessorCount, device);
    if (e != hipSuccess || p->n_cu <= 0) rc = zk_hip_fail(e, "hipDeviceGetAttribute(MultiprocessorCount)");
  }
  if (!rc) rc = build_generic_tables(p, basis);
  if (!rc) rc = zk_fold_build(p, basis);
  if (!rc) rc = zk_sep_build(p, basis);
  if (!rc) rc = zk_direct_build(p, basis);  // the plain sum on the matrix cores: sets of >= 92 functions (n_max >= 13)
  if (!rc) {
    // From which order ZK_PATH_AUTO leaves the polynomial kernels for the plain sum (see resolve_path); ZK_AUTO_DIRECT_NMAX
    // in the environment moves the switch for A/B measurements.
    const char* env = getenv("ZK_AUTO_DIRECT_NMAX");
    p->auto_direct_from = env && *env ? atoi(env) : ZK_AUTO_DIRECT_NMAX_DEFAULT;
  }
  if (rc) {
    std::string keep = g_last_error;
    zk_plan_destroy(p);
    g_last_error = keep;
    return rc;
  }
  *out = p;
  return 0;
}

static bool path_available(const zk_plan* p, int mode, int dtype, int path) {
  switch (path) {
    case ZK_PATH_GENERIC: return true;
    case ZK_PATH_FOLDED: return mode == 1 && zk_fast_frame_available(p, dtype);
    case ZK_PATH_SEPARABLE: return mode == 0 ? zk_sep_patches_available(p, dtype) : zk_sep_frame_available(p, dtype);
    case ZK_PATH_STREAM: return mode == 0 && zk_sep_stream_available(p, dtype);
    case ZK_PATH_DIRECT: return mode == 0 ? zk_direct_patches_available(p, dtype) : zk_direct_frame_available(p, dtype);
  }
  return false;
}

// Does ZK_PATH_AUTO take the plain sum over the caller's own numbers (matrix cores) for this plan?  Yes for every large
// set without polynomial tables (n_max > 24, other sets), and for the full Zernike sets from order `auto_direct_from`:
// the polynomial kernels substitute the exact polynomial for the caller's values and recombine Legendre sums whose
// coefficients grow like (1 + sqrt 2)^n, so their distance from the reference's np.dot grows with the order -- measured
// against reference outputs on structured inputs (tests/golden: st_*), profiles/r04_high_orders.txt -- while the plain
// sum is the reference's own arithmetic at every order.
bool zk_plan_auto_direct(const zk_plan* p, int mode, int dtype) {
  if (!path_available(p, mode, dtype, ZK_PATH_DIRECT)) return false;
  if (!p->sep) return true;
  const int n_max = zk_full_set_nmax(p);
  return n_max >= p->auto_direct_from;
}

// ZK_PATH_AUTO: separable (batches: its stream form where the plan prefers it), else folded (frame), else generic.
// `n_units` = patches of the call (batch mode).
static int resolve_path(const zk_plan* p, int mode, int dtype, int64_t n_units = 0) {
  if (p->path != ZK_PATH_AUTO) return path_available(p, mode, dtype, p->path) ? p->path : -1;
  if (!getenv("ZK_NO_DIRECT")) {
    if (zk_plan_auto_direct(p, mode, dtype))
      return mode == 1 || n_units >= 64 ? ZK_PATH_DIRECT : ZK_PATH_GENERIC;  // (less than a wave of patches: the per-lane sum)
    // a full Zernike set of an order that AUTO serves with the plain sum, but outside the matrix-core kernels' shapes (dense windows
    // whose staged tile exceeds the LDS: float64 above ~100 px, float32 above ~150): the per-lane plain sum -- slow, and exact
    if (p->sep && zk_full_set_nmax(p) >= p->auto_direct_from) return ZK_PATH_GENERIC;
  }
  if (mode == 0 && path_available(p, mode, dtype, ZK_PATH_STREAM) &&
      (zk_sep_stream_preferred(p, dtype, n_units) || !path_available(p, mode, dtype, ZK_PATH_SEPARABLE)))
    return ZK_PATH_STREAM;
  if (path_available(p, mode, dtype, ZK_PATH_SEPARABLE)) return ZK_PATH_SEPARABLE;
  if (path_available(p, mode, dtype, ZK_PATH_FOLDED)) return ZK_PATH_FOLDED;
  return ZK_PATH_GENERIC;
}

extern "C" int zk_plan_has_path(const zk_plan* p, int mode, int dtype, int path) {
  if (!p || (dtype != ZK_F32 && dtype != ZK_F64)) return 0;
  return (int)path_available(p, mode, dtype, path);
}

extern "C" int zk_plan_resolved_path(const zk_plan* p, int mode, int dtype, int64_t n_units) {
  if (!p || (dtype != ZK_F32 && dtype != ZK_F64) || (mode != 0 && mode != 1)) return -1;
  return resolve_path(p, mode, dtype, n_units);
}

extern "C" int zk_plan_supports(const zk_plan* p, int op, int dtype) {
  if (!p || (dtype != ZK_F32 && dtype != ZK_F64)) return 0;
  switch (op) {
    case ZK_OP_POINTS: return (int)zk_sep_points_available(p, dtype);
    case ZK_OP_MAPS: return (int)zk_sep_maps_available(p, dtype);
  }
  return 0;
}

extern "C" int zk_plan_disk_pixels(const zk_plan* p) { return p ? p->npx : 0; }

extern "C" int zk_plan_set_path(zk_plan* p, int path) {
  if (!p || path < ZK_PATH_AUTO || path > ZK_PATH_DIRECT) return zk_fail(ZK_E_BADARG, "bad path");
  p->path = path;
  return 0;
}

// ------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------
static int check_dtype(int dtype) {
  if (dtype != ZK_F32 && dtype != ZK_F64) return zk_fail(ZK_E_BADARG, "dtype must be ZK_F32 or ZK_F64");
  return 0;
}

static size_t elem_size(int dtype) { return dtype == ZK_F32 ? 4 : 8; }

extern "C" int zk_transform_patches_dev(zk_plan* p, const void* patches, int dtype, int64_t n_patches,
                                        double* out, void* hip_stream) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  int rc = check_dtype(dtype);
  if (rc) return rc;
  if (n_patches < 0) return zk_fail(ZK_E_BADARG, "negative patch count");
  if (n_patches == 0) return 0;
  if (!patches || !out) return zk_fail(ZK_E_BADARG, "null device pointer");
  ZK_ON_PLAN_DEVICE(p);
  hipStream_t s = (hipStream_t)hip_stream;  // exactly the caller's stream; NULL is HIP's default stream
  const int path = resolve_path(p, 0, dtype, n_patches > p->job_units ? n_patches : p->job_units);
  if (path < 0) return zk_fail(ZK_E_BADARG, "the forced kernel path is not available for this plan / dtype");
  if (path == ZK_PATH_SEPARABLE) return zk_launch_sep_patches(p, patches, dtype, n_patches, out, s);
  if (path == ZK_PATH_STREAM) return zk_launch_sep_stream(p, patches, dtype, n_patches, out, s);
  // the plain sum over the caller's numbers on the matrix cores (ZK_PATH_AUTO: resolve_path; ZK_NO_DIRECT in the environment
  // keeps AUTO off it); a forced ZK_PATH_GENERIC is always the per-lane kernel
  if (path == ZK_PATH_DIRECT) return zk_launch_direct_patches(p, patches, dtype, n_patches, out, s);
  return zk_launch_generic_patches(p, patches, dtype, n_patches, out, s);
}

extern "C" int zk_transform_frame_dev(zk_plan* p, const void* image, int dtype, int64_t H, int64_t W,
                                      int64_t row0, int64_t n_rows, double* out, void* hip_stream) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  int rc = check_dtype(dtype);
  if (rc) return rc;
  if (H <= 0 || W <= 0 || H > 0x3fffffff || W > 0x3fffffff) return zk_fail(ZK_E_BADARG, "bad frame shape");
  if (row0 < 0 || n_rows < 0 || row0 + n_rows > H) return zk_fail(ZK_E_BADARG, "row band outside the frame");
  if (n_rows == 0) return 0;
  if (!image || !out) return zk_fail(ZK_E_BADARG, "null device pointer");
  ZK_ON_PLAN_DEVICE(p);
  hipStream_t s = (hipStream_t)hip_stream;  // exactly the caller's stream; NULL is HIP's default stream
  const int path = resolve_path(p, 1, dtype);
  if (path < 0) return zk_fail(ZK_E_BADARG, "the forced kernel path is not available for this plan / dtype");
  // n_max <= 8 and windows up to 65 px: the strip form of the dense kernel (two outputs per lane, zk_sep_strip.hip);
  // ZK_NO_STRIP in the environment keeps the one-output kernel (A/B measurements, tests)
  if (path == ZK_PATH_SEPARABLE && zk_sep_strip_available(p, dtype) && !getenv("ZK_NO_STRIP"))
    return zk_launch_sep_strip(p, image, dtype, H, W, row0, n_rows, out, s);
  if (path == ZK_PATH_SEPARABLE) return zk_launch_sep_frame(p, image, dtype, H, W, row0, n_rows, out, s);
  if (path == ZK_PATH_FOLDED) return zk_launch_fast_frame(p, image, dtype, H, W, row0, n_rows, out, s);
  if (path == ZK_PATH_DIRECT) return zk_launch_direct_frame(p, image, dtype, H, W, row0, n_rows, out, s);
  return zk_launch_generic_frame(p, image, dtype, H, W, row0, n_rows, out, s);
}

extern "C" int zk_transform_frame_dev_strided(zk_plan* p, const void* image, int dtype, int64_t H, int64_t W,
                                              int64_t row0, int64_t n_rows, double* out, int64_t plane_stride,
                                              void* hip_stream) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  if (plane_stride < n_rows * W) return zk_fail(ZK_E_BADARG, "plane_stride smaller than the band");
  p->out_plane = plane_stride;
  const int rc = zk_transform_frame_dev(p, image, dtype, H, W, row0, n_rows, out, hip_stream);
  p->out_plane = 0;
  return rc;
}

// ------------------------------------------------------------------------------------
// fused symmetry maps
// ------------------------------------------------------------------------------------
int zk_complex_count(int n_max) {
  int k = 0;
  for (int n = 0; n <= n_max; ++n) k += n / 2 + 1;
  return k;
}

extern "C" int zk_frame_maps_dev(zk_plan* p, const void* image, int dtype, int64_t H, int64_t W, int64_t row0,
                                 int64_t n_rows, const int32_t* folds, int n_folds, const int32_t* m_unselect,
                                 int n_unselect, int p_norm, const double* theta, int n_theta, double* rot,
                                 double* ab, double* mirror, void* hip_stream) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  int rc = check_dtype(dtype);
  if (rc) return rc;
  if (H <= 0 || W <= 0 || H > 0x3fffffff || W > 0x3fffffff) return zk_fail(ZK_E_BADARG, "bad frame shape");
  if (row0 < 0 || n_rows < 0 || row0 + n_rows > H) return zk_fail(ZK_E_BADARG, "row band outside the frame");
  if (n_rows == 0) return 0;
  if (!image) return zk_fail(ZK_E_BADARG, "null device pointer");
  if (!m_unselect || n_unselect <= 0) return zk_fail(ZK_E_BADARG, "m=0 must be included in m_unselect.");
  ZK_ON_PLAN_DEVICE(p);
  return zk_launch_sep_maps(p, image, dtype, H, W, row0, n_rows, folds, n_folds, m_unselect, n_unselect, p_norm,
                            theta, n_theta, rot, ab, mirror, (hipStream_t)hip_stream);
}

extern "C" int zk_frame_maps_dev_strided(zk_plan* p, const void* image, int dtype, int64_t H, int64_t W, int64_t row0,
                                         int64_t n_rows, const int32_t* folds, int n_folds,
                                         const int32_t* m_unselect, int n_unselect, int p_norm, const double* theta,
                                         int n_theta, double* rot, double* ab, double* mirror, int64_t plane_stride,
                                         void* hip_stream) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  if (plane_stride < n_rows * W) return zk_fail(ZK_E_BADARG, "plane_stride smaller than the band");
  p->out_plane = plane_stride;
  const int rc = zk_frame_maps_dev(p, image, dtype, H, W, row0, n_rows, folds, n_folds, m_unselect, n_unselect, p_norm,
                                   theta, n_theta, rot, ab, mirror, hip_stream);
  p->out_plane = 0;
  return rc;
}

extern "C" int zk_moment_maps_dev(zk_plan* p, const double* moments, int64_t n_rows, const int32_t* folds, int n_folds,
                                  const int32_t* m_unselect, int n_unselect, int p_norm, const double* theta, int n_theta,
                                  double* rot, double* ab, double* mirror, void* hip_stream) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  if (n_rows < 0) return zk_fail(ZK_E_BADARG, "negative row count");
  if (n_rows == 0) return 0;
  if (!moments) return zk_fail(ZK_E_BADARG, "null device pointer");
  if (!m_unselect || n_unselect <= 0) return zk_fail(ZK_E_BADARG, "m=0 must be included in m_unselect.");
  ZK_ON_PLAN_DEVICE(p);
  return zk_launch_maps_rows(p, moments, n_rows, folds, n_folds, m_unselect, n_unselect, p_norm, theta, n_theta, rot, ab, mirror,
                             (hipStream_t)hip_stream);
}

// ------------------------------------------------------------------------------------
// moments at points
// ------------------------------------------------------------------------------------
extern "C" int zk_transform_points_dev(zk_plan* p, const void* image, int dtype, int64_t H, int64_t W,
                                       const int32_t* points, int64_t n_points, double* out, void* hip_stream) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  int rc = check_dtype(dtype);
  if (rc) return rc;
  if (H <= 0 || W <= 0 || H > 0x3fffffff || W > 0x3fffffff) return zk_fail(ZK_E_BADARG, "bad frame shape");
  if (n_points < 0) return zk_fail(ZK_E_BADARG, "negative point count");
  if (n_points == 0) return 0;
  if (!image || !points || !out) return zk_fail(ZK_E_BADARG, "null device pointer");
  ZK_ON_PLAN_DEVICE(p);
  hipStream_t s = (hipStream_t)hip_stream;
  if (zk_sep_points_available(p, dtype)) return zk_launch_sep_points(p, image, dtype, H, W, points, n_points, out, s);
  // Plans without the key-point kernel (n_max > 16, sets or bases off the separable path): cut the windows
  // on the device into a plan-owned batch, chunk by chunk, and run the batch path on it -- what the
  // reference does on the host (features/_keypoint.py:60-78 + _zps.py:146-157).
  const size_t patch_bytes = (size_t)p->size * p->size * elem_size(dtype);
  int64_t chunk = (int64_t)((size_t)1 << 30) / (int64_t)patch_bytes;  // <= 1 GiB of windows at a time
  if (chunk < 64) chunk = 64;
  if (chunk > n_points) chunk = n_points;
  if ((rc = zk_ensure(&p->d_gather, &p->d_gather_bytes, (size_t)chunk * patch_bytes))) return rc;
  for (int64_t first = 0; first < n_points; first += chunk) {
    const int64_t n = n_points - first < chunk ? n_points - first : chunk;
    if ((rc = zk_launch_gather_points(p, image, dtype, H, W, points + 2 * first, n, p->d_gather, s))) return rc;
    if ((rc = zk_transform_patches_dev(p, p->d_gather, dtype, n, out + first * p->n_poly, hip_stream))) return rc;
  }
  return 0;
}

