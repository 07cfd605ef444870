// zk_internal.h -- shared declarations of libzernike_hip (not part of the public ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "zernike_hip.h"

// basis functions per accumulation pass of the generic kernel: 64 when the whole set fits one pass, else 32
// (batch (56, n_max 25): 25.9 ms at 16 -> 13.6 at 32 -> 12.8 at 64; batch (32, 8): 14.5 -> 9.8 -> 2.6;
//  dense (56, 25): 37.0 -> 26.7 -> 43.1)
#define ZK_GEN_CHUNK_MAX 64

// Parity class of a real Zernike function under the mirrors x -> -x and y -> -y
// (cos(m t): x-parity (-1)^m, y-even; sin(|m| t): x-parity -(-1)^m, y-odd).
enum zk_class { ZK_EE = 0, ZK_OE = 1, ZK_EO = 2, ZK_OO = 3 };  // <x parity><y parity>, E even / O odd

struct zk_fold_tables;  // zk_fold.h
struct zk_sep_tables;   // zk_sep.h

struct zk_plan {
  int size = 0;
  int n_poly = 0;
  int device = 0;
  int n_cu = 256;                // compute units of the device (persistent grids)
  int path = ZK_PATH_AUTO;
  std::vector<int32_t> n, m;

  // ---- generic (unfolded) tables -------------------------------------------------
  int npx = 0;                   // pixels with a non-zero basis value (the rho<=1 disk)
  int gen_chunk = 32;            // 32 or 64 (see ZK_GEN_CHUNK_MAX)
  int n_chunks = 0;              // ceil(n_poly / gen_chunk)
  int2* d_pix = nullptr;         // [npx] (row, col) of each disk pixel, row-major order
  double* d_gen_tab = nullptr;   // [n_chunks][npx][gen_chunk], basis/area, zero padded
  // Dense mode restates the reference's CONVOLUTION + sign fix (_zps.py:165-178): window pixel (r, c) meets
  // (-1)^n V(K-1-r, K-1-c) -- equal to V(r, c) only for an exactly point-symmetric basis, and the reference's float64 basis
  // is that to 1e-13 (n_max 10) .. 1e-8 (24) .. 8e-4 (36) of max|V|.  The kernels that sum the caller's own numbers
  // (generic, direct) therefore take them flipped and signed in dense mode (ZK_NO_CONV_FLIP in the environment: the plain
  // inner product, for A/B); the polynomial kernels evaluate the exactly symmetric polynomial either way.
  bool conv_flip = false;
  double* d_sign = nullptr;      // [n_chunks * gen_chunk] (-1)^n per function (1 without conv_flip and past n_poly)

  // ---- parity-folded tables (fast kernels) ----------------------------------------
  zk_fold_tables* fold = nullptr;  // nullptr when the basis lacks the mirror parities

  // ---- row-separable tables (fastest kernels) ----------------------------------------
  zk_sep_tables* sep = nullptr;    // nullptr when the basis is not the standard polynomial set

  // ---- large sets, batch mode: DMA-staged direct sums, CH functions per launch (zk_direct_patches.hip) ----
  struct zk_direct_tables* direct = nullptr;  // nullptr below 92 functions
  int auto_direct_from = 1 << 30;             // full Zernike sets: ZK_PATH_AUTO takes the plain sum from this n_max on

  // ---- symmetry maps: fold weights + cos / sin(m theta) (zk_sep_maps.hip).  One device table per distinct
  // (folds, m_unselect, theta) option set, kept for the life of the plan (a few KiB each, at most
  // ZK_TRIG_CACHE of them): a repeated call uploads nothing and never synchronises, and a launch in flight
  // never sees its table overwritten.
  struct trig_entry {
    std::vector<double> host;
    double* dev = nullptr;
  };
  std::vector<trig_entry> trig_cache;

  // ---- execution state -------------------------------------------------------------
  hipStream_t stream = nullptr;  // owned; host-variant calls and default for *_dev
  struct zk_host_ring* ring = nullptr;  // staging of the host-buffer entry points (zk_host.hip)
  size_t host_chunk = 0;                // bytes of input + output per chunk; 0 = default
  void* d_gather = nullptr;      // key points without the key-point kernel: windows cut on the device
  size_t d_gather_bytes = 0;
  void* d_points_tmp = nullptr;  // key points: bucket histogram, cursors and the index list in bucket order (zk_sep_points.hip)
  size_t d_points_tmp_bytes = 0;
  double* d_scratch = nullptr;   // class-pass batch kernels (n_max > 16): [n_poly][chunk] planes
  size_t d_scratch_bytes = 0;
  // distance in doubles between consecutive output planes of the dense / maps kernels; 0 = compact
  // (n_rows * W).  Set for the duration of one *_strided call (zk_api.hip) so that a row band lands inside
  // the full (planes, H, W) array -- the layout the multi-GPU gather reassembles in place.
  long long out_plane = 0;
  // patches of the whole job while a host-buffer call feeds the batch kernels chunk by chunk: ZK_PATH_AUTO picks
  // the kernel the job as a whole would get (same numbers whatever the chunking)
  int64_t job_units = 0;

  bool profile = false;
  std::vector<hipEvent_t> ev_pool;  // pairs: [2k] start, [2k+1] stop
  size_t ev_used = 0;               // events handed out since the last read
  int64_t prof_launches = 0;
  double prof_ms = 0.0;
};

// error plumbing (zk_api.hip)
int zk_fail(int code, const std::string& what);
int zk_hip_fail(hipError_t e, const char* what);
#define ZK_HIP(call)                                   \
  do {                                                 \
    hipError_t zk_e_ = (call);                         \
    if (zk_e_ != hipSuccess) return zk_hip_fail(zk_e_, #call); \
  } while (0)

// every entry point runs on its plan's / communicator's device and leaves the caller's current device as it was
struct zk_device_scope {
  int prev = -1, dev;
  hipError_t err = hipSuccess;
  explicit zk_device_scope(int d) : dev(d) {
    if (hipGetDevice(&prev) != hipSuccess) {
      (void)hipGetLastError();
      prev = -1;
    }
    if (prev != dev) err = hipSetDevice(dev);
  }
  ~zk_device_scope() {
    if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
  }
};
#define ZK_ON_DEVICE(d)           \
  zk_device_scope zk_scope_(d);   \
  if (zk_scope_.err != hipSuccess) return zk_hip_fail(zk_scope_.err, "hipSetDevice")
#define ZK_ON_PLAN_DEVICE(p) ZK_ON_DEVICE((p)->device)

static inline long long zk_out_plane(const zk_plan* p, int64_t n_rows, int64_t W) {
  return p->out_plane ? p->out_plane : (long long)n_rows * W;
}

// A dense launch covers at most 65535 blocks of `rows_per_block` output rows (grid.y); taller bands are cut
// into consecutive launches.  f(row0, n_rows, out_offset) with out_offset = doubles to add to the output base.
template <class F>
static inline int zk_for_row_bands(int64_t row0, int64_t n_rows, int64_t W, int rows_per_block, F f) {
  const int64_t cap = (int64_t)65535 * rows_per_block;
  for (int64_t b0 = 0; b0 < n_rows; b0 += cap) {
    const int64_t nb = n_rows - b0 < cap ? n_rows - b0 : cap;
    const int rc = f(row0 + b0, nb, (long long)b0 * W);
    if (rc) return rc;
  }
  return 0;
}

// grow-only device buffer
static inline int zk_ensure(void** buf, size_t* have, size_t need) {
  if (*have >= need) return 0;
  if (*buf) {
    ZK_HIP(hipFree(*buf));
    *buf = nullptr;
    *have = 0;
  }
  ZK_HIP(hipMalloc(buf, need));
  *have = need;
  return 0;
}

// host-buffer entry points (zk_host.hip)
void zk_host_release(zk_plan* p);  // frees the staging ring
int zk_complex_count(int n_max);   // number of (n, |m|) pairs (zk_api.hip)

// profiling brackets (zk_api.hip)
int zk_prof_begin(zk_plan* p, hipStream_t s);
int zk_prof_end(zk_plan* p, hipStream_t s);

// kernel launchers; each returns 0 or a negative code.
int zk_launch_generic_patches(zk_plan* p, const void* in, int dtype, int64_t n_patches, double* out,
                              hipStream_t s);
int zk_launch_gather_points(zk_plan* p, const void* img, int dtype, int64_t H, int64_t W, const int32_t* pts,
                            int64_t n_points, void* patches, hipStream_t s);  // zk_generic.hip
int zk_launch_generic_frame(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0,
                            int64_t n_rows, double* out, hipStream_t s);

// direct parity-folded path (zk_fold.hip / zk_fast_frame.hip)
int zk_full_set_nmax(const zk_plan* p);              // n_max if (n, m) is the full reference set, else -1
int zk_fold_build(zk_plan* p, const double* basis);  // fills p->fold or leaves it null
void zk_fold_free(zk_plan* p);
bool zk_fast_frame_available(const zk_plan* p, int dtype);
int zk_launch_fast_frame(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0,
                         int64_t n_rows, double* out, hipStream_t s);

// row-separable path (zk_sep.hip / zk_sep_frame.hip / zk_sep_patches.hip)
int zk_sep_build(zk_plan* p, const double* basis);   // fills p->sep or leaves it null
void zk_sep_free(zk_plan* p);
bool zk_sep_frame_available(const zk_plan* p, int dtype);
bool zk_sep_patches_available(const zk_plan* p, int dtype);
int zk_direct_build(zk_plan* p, const double* basis);  // zk_direct_patches.hip
void zk_direct_free(zk_plan* p);
bool zk_direct_patches_available(const zk_plan* p, int dtype);
bool zk_plan_auto_direct(const zk_plan* p, int mode, int dtype);  // zk_api.hip: does ZK_PATH_AUTO take the plain sum?
#ifndef ZK_AUTO_DIRECT_NMAX_DEFAULT
#define ZK_AUTO_DIRECT_NMAX_DEFAULT 17
#endif
int zk_launch_direct_patches(zk_plan* p, const void* in, int dtype, int64_t n_patches, double* out, hipStream_t s);
bool zk_direct_frame_available(const zk_plan* p, int dtype);
int zk_launch_direct_frame(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0, int64_t n_rows, double* out,
                           hipStream_t s);
bool zk_sep_strip_available(const zk_plan* p, int dtype);   // zk_sep_strip.hip: dense, n_max <= 8, two outputs per lane
int zk_launch_sep_strip(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0, int64_t n_rows,
                        double* out, hipStream_t s);
bool zk_sep_points_available(const zk_plan* p, int dtype);  // single-pass kernels only (n_max <= 16)
bool zk_sep_maps_available(const zk_plan* p, int dtype);
int zk_launch_sep_frame(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0,
                        int64_t n_rows, double* out, hipStream_t s);
int zk_launch_sep_patches(zk_plan* p, const void* in, int dtype, int64_t n_patches, double* out,
                          hipStream_t s);
// stream form of the batch kernel (zk_sep_stream.hip)
bool zk_sep_stream_available(const zk_plan* p, int dtype);
bool zk_sep_stream_preferred(const zk_plan* p, int dtype, int64_t n_patches);
int zk_launch_sep_stream(zk_plan* p, const void* in, int dtype, int64_t n_patches, double* out, hipStream_t s);
int zk_launch_sep_maps(zk_plan* p, const void* in, int dtype, int64_t H, int64_t W, int64_t row0, int64_t n_rows,
                       const int32_t* folds, int n_folds, const int32_t* m_unselect, int n_unselect, int p_norm,
                       const double* theta, int n_theta, double* rot, double* ab, double* mirror, hipStream_t s);
int zk_launch_maps_rows(zk_plan* p, const double* mom, int64_t n_rows, const int32_t* folds, int n_folds, const int32_t* m_unselect,
                        int n_unselect, int p_norm, const double* theta, int n_theta, double* rot, double* ab, double* mirror,
                        hipStream_t s);  // zk_sep_maps.hip: the symmetry-map tail on an (N, n_poly) matrix of moments
int zk_launch_sep_points(zk_plan* p, const void* img, int dtype, int64_t H, int64_t W, const int32_t* pts,
                         int64_t n_points, double* out, hipStream_t s);
