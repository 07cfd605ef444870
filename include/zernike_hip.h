/*
 * zernike_hip.h -- C ABI of libzernike_hip.so, the MI355X (gfx950) implementation of
 * motif-learn's per-patch Zernike-moment hot path.
 *
 * The reference (jiadongdan/motif-learn, pure Python) has no native interface; its boundary is
 * the Python class `mtflearn.features.ZPs`.  Each entry point below replaces the arithmetic of
 * one reference method and is what a maintainer would bind with ctypes from inside that method
 * (binding stub: INTEGRATION.md):
 *
 *   zk_plan_create            <- ZPs.__init__ keeps `self.polynomials`            (_zps.py:48-50)
 *   zk_transform_patches      <- ZPs._transform_dot_product                       (_zps.py:146-157)
 *   zk_transform_frame        <- ZPs._transform_fft_convolve                      (_zps.py:159-193)
 *   zk_*_dev                  <- same, operands already resident in HBM (bench / multi-GPU)
 *   zk_transform_points       <- KeyPoints.extract_patches + ZPs.transform        (_keypoint.py:60-78)
 *   zk_frame_maps             <- zmoments.to_complex / rot_maps / mirror_map      (_zmoments.py:300-316, 420-493)
 *   zk_autocorr_mean, zk_polar_profile        <- estimate_patch_size, radial_profile  (_patch_size.py:48-100, 221-302)
 *   zk_power_spectra, zk_denoise_fft          <- _get_cumulative_energy, denoise_fft   (_estimate_n_max.py:8-86,
 *                                                                                      denoise/_denoise_fft.py:4-47)
 *   zk_comm_*, zk_allgather_rows              (no counterpart: the multi-GPU exchange of north_star)
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative code
 *     (-hipError_t for runtime failures, ZK_E_* for argument errors) and never throws;
 *     zk_last_error_string() describes the most recent failure on the calling thread.
 *   - the caller owns every host buffer; inputs are read-only, outputs are C-contiguous
 *     float64; no pointer is retained after the call returns.  Device memory owned by the
 *     library lives inside the plan and is released by zk_plan_destroy().
 *   - a plan is bound to one device and is not thread-safe (one stream per plan).  Every entry point
 *     leaves the calling thread's current HIP device as it found it.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails.
 */
#ifndef ZERNIKE_HIP_H
#define ZERNIKE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZK_ABI_VERSION 2

/* element type of the image / patch operand */
#define ZK_F32 0
#define ZK_F64 1
/* Narrow detector formats, accepted by the HOST-buffer entry points only (zk_transform_patches / _frame / _points,
 * zk_frame_maps): the bytes cross PCIe as they are (1-2 B per pixel instead of 4) and are widened to float32 on the
 * device, which is exact -- the moments equal those of NumPy's float64 promotion of the same integers
 * (reference _zps.py:151-155 upcasts whatever dtype it is given). */
#define ZK_U8  2
#define ZK_U16 3
#define ZK_I16 4

/* argument-error codes (runtime failures are -hipError_t, i.e. -1 .. -1999) */
#define ZK_E_BADARG   (-10001)
#define ZK_E_NODEVICE (-10002)
#define ZK_E_NOMEM    (-10003)
#define ZK_E_COMM     (-10004)  /* RCCL unavailable, rendezvous failed or a collective returned an error */
#define ZK_E_FFT      (-10005)  /* hipFFT unavailable or a transform failed */

/* kernel selection, for tests and A/B measurements.  Default ZK_PATH_AUTO = the fastest family that restates the reference
 * to SURVEY 8c's criterion (elementwise rtol 1e-6, floor 1e-12 max|Z| up to n_max 12, 1e-11 above) on the reference's own
 * outputs: the polynomial kernels (SEPARABLE / STREAM) for full Zernike sets up to n_max 16, the plain sum on the matrix
 * cores (DIRECT) from n_max 17 and for every other set of >= 92 functions, else FOLDED / GENERIC. */
#define ZK_PATH_AUTO      0
#define ZK_PATH_GENERIC   1  /* any size / n_max / dtype: unfolded direct summation, one output per lane, ~1e-16 of the definition */
#define ZK_PATH_FOLDED    2  /* frame only: mirror-folded direct summation (4x fewer FMAs), ~1e-15            */
#define ZK_PATH_SEPARABLE 3  /* mirror-folded row-separable sums (Legendre products), ~1e-13; fastest.  Full
                                Zernike sets up to n_max 24 (17-24: one pass per mirror-parity class; ~1e-9 at 20, ~3e-8 at 24,
                                which is the rounding of the reference's own float64 basis there) */
#define ZK_PATH_STREAM    4  /* patches only: row-separable sums over the contiguous pixel stream of a patch,
                                whole 128-B lines whatever the patch size; AUTO prefers it where the row-pair
                                kernel of ZK_PATH_SEPARABLE would issue half-line requests */
#define ZK_PATH_DIRECT    5  /* sets of >= 92 functions (n_max >= 13), windows of 16 .. 512 px: the plain sum over the CALLER'S
                                basis values as a float64 GEMM on the matrix cores (v_mfma_f64_16x16x4_f64) -- the reference's
                                own arithmetic (np.dot, _zps.py:155), ~1e-15, at any order.  Dense mode multiplies window pixel
                                (r, c) with (-1)^n V(K-1-r, K-1-c), which is what the reference's convolution + sign fix does
                                (_zps.py:165-178; equal to V(r, c) for an exactly point-symmetric basis), whenever the set is
                                point-symmetric to 1e-6.  SEPARABLE stays available as an explicit opt-in above n_max 16:
                                2-6x faster, ~1e-10 (n_max 20) .. 1e-9 (n_max 24) of max|Z| from the reference's result */

typedef struct zk_plan zk_plan;

/* Library / device discovery. */
int         zk_abi_version(void);
int         zk_device_count(void);            /* >= 0, or a negative code               */
const char* zk_last_error_string(void);

/*
 * Create a plan for one (size, n_poly) basis on `device`.
 *   basis : host, (n_poly, size, size) float64, C order -- ZPs.polynomials      (_zps.py:90)
 *   n, m  : host, (n_poly) int32 radial order / azimuthal frequency per basis function, in
 *           the order of `basis`                                                  (_zps.py:74-81)
 * The plan uploads the disk-masked basis, pre-divided by the reference's normalising area
 * pi*size^2/4 (_zps.py:154,177).  When (n, m, basis) is the reference's full real Zernike set
 * it also builds the mirror-folded table and the row-separable (Legendre) tables of the fast
 * kernels, after verifying that they reproduce `basis` at every pixel of the disk.
 */
int  zk_plan_create(int size, int n_poly, const int32_t* n, const int32_t* m,
                    const double* basis, int device, zk_plan** out);
void zk_plan_destroy(zk_plan* plan);

/* Introspection: 1 if `path` (ZK_PATH_*) is available for `mode` (0 patches, 1 frame) and `dtype`. */
int zk_plan_has_path(const zk_plan* plan, int mode, int dtype, int path);
/* 1 if the plan has the single-kernel form of zk_transform_points (ZK_OP_POINTS: full Zernike set, n_max <= 16) /
 * the kernels behind zk_frame_maps (ZK_OP_MAPS: full Zernike set, n_max <= 40; fused in one kernel up to 16, moments into a
 * scratch matrix + a planes kernel above)
 * for `dtype`.  (No reference counterpart: the reference composes these from ZPs.transform.) */
#define ZK_OP_POINTS 1
#define ZK_OP_MAPS   2
int zk_plan_supports(const zk_plan* plan, int op, int dtype);
/* Number of pixels inside the unit disk (rho <= 1 as evaluated by the caller's basis). */
int zk_plan_disk_pixels(const zk_plan* plan);
/* The family (ZK_PATH_*, never AUTO) a transform of `n_units` patches (mode 0) / of a frame (mode 1) would run on with the
 * plan's current setting: what ZK_PATH_AUTO resolves to, or the forced path; -1 if a forced path is unavailable. */
int zk_plan_resolved_path(const zk_plan* plan, int mode, int dtype, int64_t n_units);
/* Force a kernel family (ZK_PATH_*); a forced path that is unavailable makes transforms fail. */
int zk_plan_set_path(zk_plan* plan, int path);

/*
 * Batch of patches (reference _zps.py:146-157):
 *   out[p, j] = sum_{r,c} patches[p, r, c] * basis[j, r, c] / (pi size^2 / 4)
 *   patches : (N, size, size) of `dtype`, C order;  out : (N, n_poly) float64.
 * Host variant copies in/out through staging buffers owned by the plan, on the plan's stream,
 * and returns when the result is in out_host.  The *_dev variants enqueue the kernel on exactly
 * `hip_stream` (a hipStream_t; NULL = HIP's default stream) and return without synchronising.
 */
int zk_transform_patches(zk_plan* plan, const void* patches_host, int dtype, int64_t n_patches,
                         double* out_host);
int zk_transform_patches_dev(zk_plan* plan, const void* patches_dev, int dtype,
                             int64_t n_patches, double* out_dev, void* hip_stream);

/*
 * Dense frame (reference _zps.py:159-193, i.e. fftconvolve(mode='same') * (-1)^n / area):
 *   out[j, i, k] = (-1)^n[j] * sum_{r,c} pad(image)[i - ea + r, k - ea + c] * basis[j, size-1-r, size-1-c] / (pi size^2/4)
 *   with eb = (size-1)/2, ea = size-1-eb and zero padding outside the image -- the convolution and the sign fix
 *   of the reference written out.  For a point-symmetric basis (V(-x, -y) = (-1)^n V(x, y): every Zernike set in exact
 *   arithmetic) this is the inner product of the window with basis[j]; the reference's float64 basis is point-symmetric
 *   only up to rounding (1e-13 of max|V| at n_max 10, 1e-8 at 24, 8e-4 at 36), and the kernels that sum the caller's own
 *   numbers (ZK_PATH_GENERIC, ZK_PATH_DIRECT) follow the formula above to the letter; the polynomial kernels
 *   (ZK_PATH_SEPARABLE / FOLDED, verified Zernike sets only) evaluate the exactly symmetric polynomial.
 *   image : (H, W) of `dtype`;  out : (n_poly, H, W) float64 (moment-major, as the reference).
 * The *_dev variant computes output rows [row0, row0+n_rows) only and writes them to
 * out_dev laid out as (n_poly, n_rows, W) -- the row-band shard of one GPU; the image operand
 * is always the whole frame.
 */
int zk_transform_frame(zk_plan* plan, const void* image_host, int dtype, int64_t height,
                       int64_t width, double* out_host);
int zk_transform_frame_dev(zk_plan* plan, const void* image_dev, int dtype, int64_t height,
                           int64_t width, int64_t row0, int64_t n_rows, double* out_dev,
                           void* hip_stream);

/*
 * Row band written IN PLACE into a larger array: as zk_transform_frame_dev, but plane j of the band starts
 * at out_dev + j * plane_stride (doubles).  With out_dev = full + row0 * W and plane_stride = H * W the
 * band lands inside the full (n_poly, H, W) array -- the layout zk_allgather_rows reassembles without a copy.
 */
int zk_transform_frame_dev_strided(zk_plan* plan, const void* image_dev, int dtype, int64_t height,
                                   int64_t width, int64_t row0, int64_t n_rows, double* out_dev,
                                   int64_t plane_stride, void* hip_stream);

/*
 * Moments of the size x size windows at given positions of a frame: the device form of "cut patches at
 * key points, then transform the batch" (reference features/_keypoint.py:60-78 + _zps.py:146-157)
 * without materialising the (N, size, size) batch.
 *   points : (n_points, 2) int32 (x, y) = (column, row); the window is
 *            image[y - size/2 : y - size/2 + size, x - size/2 : x - size/2 + size]  (pixels outside the
 *            frame read as zero; the reference's KeyPoints drops such points beforehand)
 *   out    : (n_points, n_poly) float64
 * Plans with the key-point kernel (zk_plan_supports(plan, ZK_OP_POINTS, dtype): full Zernike set, n_max <= 16)
 * read the windows straight from the frame; the others cut them on the device into a plan-owned batch
 * (<= 1 GiB at a time) and run the batch path on it.
 */
int zk_transform_points(zk_plan* plan, const void* image_host, int dtype, int64_t height, int64_t width,
                        const int32_t* points_host, int64_t n_points, double* out_host);
int zk_transform_points_dev(zk_plan* plan, const void* image_dev, int dtype, int64_t height, int64_t width,
                            const int32_t* points_dev, int64_t n_points, double* out_dev, void* hip_stream);

/*
 * Fused dense pipeline: frame -> per-pixel symmetry maps, without writing the (n_poly, H, W) moments
 * to memory (reference notebook-3 tail: zmoments.to_complex / rot_maps / mirror_map,
 * _zmoments.py:300-316, 420-493).  Any of the three outputs may be NULL.
 *   abs_out    : (N_c, n_rows, W)      |Z_{n,m} + i Z_{n,-m}| of the raw moments, (n, m>=0) in the
 *                                      order of to_complex(), N_c = number of (n, |m|) pairs
 *   rot_out    : (n_folds, n_rows, W)  zmoments.rot_maps(folds, p, m_unselect); n_folds <= 8
 *   mirror_out : (n_rows, W)           zmoments.mirror_map(theta, p, m_unselect)
 *   m_unselect : |m| values to drop (must contain 0; reference default (0, 1));  p_norm: 2, or 0 for None
 *   theta      : host, n_theta angles in radians (reference default linspace(0, 2 pi, 360, endpoint=False))
 * Needs the row-separable tables (zk_plan_has_path(plan, 1, dtype, ZK_PATH_SEPARABLE)); otherwise fails
 * and the caller composes zk_transform_frame with the host-side container methods.
 * The *_dev variant returns without synchronising, except on the FIRST call with a new (folds, m_unselect, theta)
 * option set: that call uploads a small table with a blocking copy (the plan keeps the tables of the 8 most
 * recently used option sets, so a launch in flight never sees its table overwritten).
 */
int zk_frame_maps(zk_plan* plan, const void* image_host, int dtype, int64_t height, int64_t width,
                  const int32_t* folds, int n_folds, const int32_t* m_unselect, int n_unselect, int p_norm,
                  const double* theta, int n_theta, double* rot_host, double* abs_host, double* mirror_host);
int zk_frame_maps_dev(zk_plan* plan, const void* image_dev, int dtype, int64_t height, int64_t width,
                      int64_t row0, int64_t n_rows, const int32_t* folds, int n_folds,
                      const int32_t* m_unselect, int n_unselect, int p_norm, const double* theta, int n_theta,
                      double* rot_dev, double* abs_dev, double* mirror_dev, void* hip_stream);

/*
 * The same tail on a BATCH of moment vectors -- zmoments.to_complex / rot_maps / mirror_map on rank-2 data (reference
 * _zmoments.py:300-316, 420-493), e.g. the moments at key points:
 *   moments (N, n_poly) row-major ->  rot (N, n_folds), abs (N, N_c), mirror (N)   (any output may be NULL)
 * zk_points_maps = zk_transform_points followed by that tail with the (N, n_poly) matrix never leaving the device
 * (the reference's notebook flow KeyPoints.extract_patches -> ZPs.transform -> rot_maps).  Full Zernike sets, n_max <= 40.
 */
int zk_moment_maps(zk_plan* plan, const double* moments_host, int64_t n_rows, const int32_t* folds, int n_folds,
                   const int32_t* m_unselect, int n_unselect, int p_norm, const double* theta, int n_theta,
                   double* rot_host, double* abs_host, double* mirror_host);
int zk_moment_maps_dev(zk_plan* plan, const double* moments_dev, int64_t n_rows, const int32_t* folds, int n_folds,
                       const int32_t* m_unselect, int n_unselect, int p_norm, const double* theta, int n_theta,
                       double* rot_dev, double* abs_dev, double* mirror_dev, void* hip_stream);
int zk_points_maps(zk_plan* plan, const void* image_host, int dtype, int64_t height, int64_t width,
                   const int32_t* points_host, int64_t n_points, const int32_t* folds, int n_folds,
                   const int32_t* m_unselect, int n_unselect, int p_norm, const double* theta, int n_theta,
                   double* rot_host, double* abs_host, double* mirror_host);

/* As zk_frame_maps_dev with every output plane `plane_stride` doubles apart (see zk_transform_frame_dev_strided);
 * rot_dev / abs_dev / mirror_dev point at the first row of the band inside their full arrays. */
int zk_frame_maps_dev_strided(zk_plan* plan, const void* image_dev, int dtype, int64_t height, int64_t width,
                              int64_t row0, int64_t n_rows, const int32_t* folds, int n_folds,
                              const int32_t* m_unselect, int n_unselect, int p_norm, const double* theta,
                              int n_theta, double* rot_dev, double* abs_dev, double* mirror_dev,
                              int64_t plane_stride, void* hip_stream);

/*
 * Kernel timing with HIP events on the stream the kernels are launched on.
 * zk_plan_profile(plan, 1) brackets every subsequent kernel launch with an event pair;
 * zk_plan_profile_read() synchronises, returns the launch count and summed kernel
 * milliseconds since the last read, and resets the counters.
 */
int zk_plan_profile(zk_plan* plan, int enable);
int zk_plan_profile_read(zk_plan* plan, int64_t* launches, double* total_ms);
/* Same, launch by launch: the first `cap` kernel times (ms, in launch order) since the last read go to ms_out; *n_out =
 * the number of launches RECORDED since the last read -- larger than `cap` when the list handed back is incomplete (the
 * rest is dropped; bench.py: minimum / median / maximum of the timed steps). */
int zk_plan_profile_read_launches(zk_plan* plan, double* ms_out, int64_t cap, int64_t* n_out);

/*
 * Host staging of the host-buffer entry points (zk_transform_patches / _frame / _points, zk_frame_maps): the
 * job is cut into chunks of at most `chunk_bytes` of input + output (default 256 MiB, ZK_HOST_CHUNK_MB in the
 * environment), each chunk goes host -> pinned ring -> device -> kernel -> pinned ring -> host with the
 * three stages of consecutive chunks overlapping on three streams; the device footprint is bounded by the
 * ring, whatever the job size.  zk_plan_release_staging frees the ring and every grown staging buffer
 * (they are re-created on demand).
 */
int zk_plan_set_host_chunk(zk_plan* plan, int64_t chunk_bytes);
int zk_plan_release_staging(zk_plan* plan);
/* Page-locked host memory (hipHostMalloc, portable): arrays in it move over PCIe by DMA at link speed and
 * without blocking the calling thread; free with zk_host_free. */
int zk_host_alloc(int64_t bytes, void** out_host);
int zk_host_free(void* host);

/* ------------------------------------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, units sharded with no data-path collective, ONE exchange step that
 * reassembles the result on every rank (north_star "a single RCCL all-gather over xGMI").
 * The reference has no counterpart (single-process NumPy).  RCCL is loaded on first use (dlopen of the
 * librccl.so.1 already in the process, else the one next to the HIP runtime in use, else the system's), so
 * the library itself has no link-time dependency on it.
 *
 * A communicator owns one RCCL communicator and one HIP stream of its own.  Collectives are ORDERED AFTER
 * everything enqueued so far on the `hip_stream` argument (the producer's stream) but RUN on the
 * communicator's stream, so kernels enqueued later on `hip_stream` overlap with the transfer;
 * zk_comm_join(comm, hip_stream) makes `hip_stream` wait for every collective issued so far.  Nothing here
 * synchronises the host except zk_comm_allgather_host and the init / destroy calls.
 * ------------------------------------------------------------------------------------------------------ */
typedef struct zk_comm zk_comm;
#define ZK_COMM_ID_BYTES 128

/* Rendezvous.  Any of the three: (a) the caller moves the 128-byte id from rank 0 to the others itself;
 * (b) ranks of one node meet through a file: rank 0 writes the id to `path` (atomically), the others poll it,
 * rank 0 removes it once everyone has joined; (c) rank 0 listens on host:port and hands the id to each peer. */
int zk_comm_unique_id(void* id_out /* ZK_COMM_ID_BYTES */);
int zk_comm_init_rank(int device, int rank, int world, const void* id, zk_comm** out);
int zk_comm_init_file(int device, int rank, int world, const char* path, double timeout_s, zk_comm** out);
int zk_comm_init_tcp(int device, int rank, int world, const char* host, int port, double timeout_s, zk_comm** out);
int zk_comm_destroy(zk_comm* comm);
int zk_comm_rank(const zk_comm* comm);
int zk_comm_world(const zk_comm* comm);

/*
 * In-place all-gather of row blocks.  `full_dev` is the same-shaped (n_planes, H, W) float64 array on every
 * rank; rank r owns rows [r * rows_per_rank, min((r + 1) * rows_per_rank, H)) of every plane (the blocks of
 * shard_bounds: equal, the tail ranks possibly short or empty).  On entry each rank holds valid data in rows
 * [row_off, row_off + n_rows) of ITS block (clipped to the block); on completion those rows of EVERY block are
 * valid on every rank.  row_off = 0, n_rows = rows_per_rank gathers whole blocks; smaller windows are the
 * chunks of a pipelined gather (kernel of chunk c+1 overlapping the transfer of chunk c).
 *   moment matrix of a patch batch (N, n_poly):      n_planes = 1, H = N, W = n_poly
 *   dense moments / symmetry maps (planes, H, W):    rows of the frame
 *   a batch of frames (F, n_poly, H, W):             n_planes = 1, H = F, W = n_poly * H * W
 * Whole equal blocks of a single plane go through ncclAllGather (in place); everything else is one grouped
 * ncclSend / ncclRecv per peer and plane -- on the fully connected xGMI mesh that is the direct exchange
 * (every link carries one block, no ring forwarding).  ZK_COMM_ALGO=p2p|allgather|bcast in the environment
 * forces one form (bcast: one grouped ncclBroadcast per owner).
 */
int zk_allgather_rows(zk_comm* comm, double* full_dev, int64_t n_planes, int64_t height, int64_t width,
                      int64_t rows_per_rank, int64_t row_off, int64_t n_rows, void* hip_stream);

/*
 * The schedule of zk_allgather_rows as data.  zk_allgather_rows is two parts: this PURE planner (no GPU, no
 * RCCL, no communicator: callable on any machine) and an executor that hands the list, in order, to RCCL.
 * One zk_xfer is one RCCL call on the (n_planes, H, W) array seen as a flat run of doubles:
 *   ZK_XFER_SEND       ncclSend(full + offset, count, peer)
 *   ZK_XFER_RECV       ncclRecv(full + offset, count, peer)
 *   ZK_XFER_ALLGATHER  ncclAllGather(send = full + offset, recv = full + offset - rank * count, count per rank); peer -1
 *   ZK_XFER_BCAST      ncclBroadcast(full + offset, in place, count, root = peer)
 * Entries with the same `group` are issued between one ncclGroupStart / ncclGroupEnd (groups are numbered from 0,
 * ascending along the list).  Contract of a plan set (what tests/test_comm_plan.py checks for every rank of a world):
 * within a group the sends of rank a to rank b and the receives of b from a are equally many and pairwise of
 * equal count IN ORDER (RCCL matches point-to-point calls between two ranks in issue order); a rank's receives
 * are exactly the other ranks' windows, once each, disjoint from one another and from its own window; collective
 * entries (ALLGATHER / BCAST) appear on every rank in the same order with the same root and count.
 * `algo`: ZK_COMM_AUTO (whole equal blocks of one plane -> ALLGATHER, else P2P), ZK_COMM_P2P, ZK_COMM_ALLGATHER
 * (falls back to P2P when the blocks are not whole and equal), ZK_COMM_BCAST.
 * Writes at most `cap` entries to `out` (may be NULL with cap 0) and the number the plan HAS to *n_out.
 */
typedef struct zk_xfer {
  int32_t op;     /* ZK_XFER_* */
  int32_t peer;   /* destination (SEND), source (RECV), root (BCAST), -1 (ALLGATHER) */
  int32_t group;  /* ncclGroupStart / End bracket this entry belongs to */
  int32_t plane;  /* plane index the run lies in (information only; offset already includes it) */
  int64_t offset; /* first element, in doubles from full_dev */
  int64_t count;  /* elements */
} zk_xfer;
enum { ZK_XFER_SEND = 1, ZK_XFER_RECV = 2, ZK_XFER_ALLGATHER = 3, ZK_XFER_BCAST = 4 };
enum { ZK_COMM_AUTO = 0, ZK_COMM_P2P = 1, ZK_COMM_ALLGATHER = 2, ZK_COMM_BCAST = 3 };
#define ZK_COMM_PLANES_PER_GROUP 16
int zk_allgather_rows_plan(int rank, int world, int64_t n_planes, int64_t height, int64_t width, int64_t rows_per_rank,
                           int64_t row_off, int64_t n_rows, int algo, zk_xfer* out, int64_t cap, int64_t* n_out);
int zk_comm_join(zk_comm* comm, void* hip_stream);
/* Blocking all-gather of the same number of host bytes from every rank (timings, checksums, the k x D sums of a sharded
 * k-means step; doubles as a barrier).  Up to 64 MiB per rank. */
int zk_comm_allgather_host(zk_comm* comm, const void* send_host, void* recv_host, int64_t bytes_per_rank);

/* ------------------------------------------------------------------------------------------------------
 * Parameter pickers: the device side of the two reference routines that choose (size, n_max) for ZPs
 * (SURVEY 8f rank 3).  Host buffers in, host buffers out, blocking; no plan involved.  The FFTs are hipFFT's
 * (bound at run time: the library loads without it); the reference uses scipy / numpy FFTs at the same places.
 *
 *   zk_autocorr_mean  <- the loop of estimate_patch_size (features/_patch_size.py:268-279): for each window
 *                        image[y:y+window, x:x+window] (origins as (y, x) pairs), standardise_image (zero mean, unit
 *                        population std; a constant window fails with the reference's message), then
 *                        scipy.signal.correlate(p, p, mode='same', method='fft'); out = mean over the windows,
 *                        (window, window) float64, zero lag at [window/2, window/2].
 *   zk_polar_profile  <- radial_profile (features/_patch_size.py:48-100): skimage.transform.warp_polar(data,
 *                        center=(h/2, w/2), scaling='linear') aggregated over its 360 angles (method 0 mean, 1 max,
 *                        2 sum) -> (batch, zk_polar_radii(h, w)) float64.  scikit-image is not available in the
 *                        build image: the resampling restates its published algorithm (see zk_pickers.hip).
 *   zk_power_spectra  <- _get_cumulative_energy up to the radial profile (features/_estimate_n_max.py:42-65):
 *                        |fftshift(fft2(patch * outer(w, w)))|^2 for every window; window_1d may be NULL.
 *   zk_denoise_fft    <- denoise_fft (denoise/_denoise_fft.py:4-47): keep the ceil(p H W) Fourier coefficients of
 *                        largest power (exact selection; among coefficients that tie at the cut-off an arbitrary
 *                        subset survives, as with numpy.argpartition), inverse transform, real part.
 * ------------------------------------------------------------------------------------------------------ */
int zk_autocorr_mean(int device, const void* image_host, int dtype, int64_t height, int64_t width, int64_t window,
                     const int32_t* origins_yx, int n_windows, int standardize, double* out_host);
int64_t zk_polar_radii(int64_t h, int64_t w);
int zk_polar_profile(int device, const double* data_host, int64_t batch, int64_t h, int64_t w, int64_t center_row,
                     int64_t center_col /* negative: (h/2, w/2) */, int method, double* out_host);
int zk_power_spectra(int device, const void* image_host, int dtype, int64_t height, int64_t width, int64_t size,
                     const int32_t* origins_yx, int n_windows, const double* window_1d, double* out_host);
int zk_denoise_fft(int device, const void* image_host, int dtype, int64_t height, int64_t width, double p,
                   double* out_host);
/* skimage.restoration.estimate_sigma of a 2-D image, as estimate_n_max uses it (_estimate_n_max.py:109): median of the
 * non-zero |db2 diagonal detail coefficients| (PyWavelets dwtn, mode 'symmetric') / 0.6745 -- restated from scikit-image /
 * PyWavelets, which are not installed here (parity unpinned).  Coefficients and the median's order statistics on the device. */
int zk_wavelet_sigma(int device, const void* image_host, int dtype, int64_t height, int64_t width, double* sigma_out);

/* ------------------------------------------------------------------------------------------------------
 * First downstream consumer of the moment matrix: the two streaming passes of
 *   pca(X, n_components)   reference features/_dimension_reduction.py:3-6 (sklearn PCA(n).fit_transform(X))
 * on a float64 matrix X (N, D), D <= 127 (45 moments at n_max 8).
 *   zk_gram[_dev]     G (D+1, D+1) = [X | 1]^T [X | 1]: X^T X with the column sums in the last row / column and N in
 *                     the corner -- everything the covariance needs, one pass over X.
 *   zk_project[_dev]  Y (N, k) = (X - mean) components^T, k <= 16.
 * The D x D eigen-problem between them is solved on the host (LAPACK, as scikit-learn's "covariance_eigh" solver does).
 * The host-buffer zk_gram leaves X on the device and hands the copy back (X_dev_out) for zk_project, which frees it
 * when free_x is non-zero.
 * ------------------------------------------------------------------------------------------------------ */
int zk_gram(int device, const double* X_host, int64_t n_rows, int n_features, double* gram_host, void** X_dev_out);
int zk_project(int device, const void* X_dev, int64_t n_rows, int n_features, const double* mean_host,
               const double* components_host, int n_components, double* Y_host, int free_x);
int zk_gram_dev(int device, const double* X_dev, int64_t n_rows, int n_features, double* gram_dev, void* hip_stream);
int zk_project_dev(int device, const double* X_dev, int64_t n_rows, int n_features, const double* mean_dev,
                   const double* components_dev, int n_components, double* Y_dev, void* hip_stream);

/* ------------------------------------------------------------------------------------------------------
 * Clustering consumers of the moment matrix, csrc/zk_cluster.hip:
 *   kmeans_lbs(X, n)   reference clustering/_clustering_functions.py:8-22  (sklearn KMeans(n, random_state).fit(X).labels_)
 *   gmm_lbs(X, n)      reference clustering/_clustering_functions.py:25-33 (sklearn GaussianMixture(n, type).fit(X).predict(X))
 * zk_rows = a float64 matrix (N, D), D <= 127, resident on one device together with the work buffers of these passes.
 * Every pass over the matrix is one call here; what happens between passes (random draws, centre updates, D x D Cholesky
 * factors, convergence tests) is scikit-learn's control flow, restated by the caller (mtflearn_amd/clustering.py).
 * Results do not depend on scheduling: per-workgroup partial sums are reduced in a fixed order.
 * ------------------------------------------------------------------------------------------------------ */
typedef struct zk_rows zk_rows;
int zk_rows_create(int device, const double* X_host, int64_t n_rows, int n_features, zk_rows** out); /* uploads a copy */
/* borrows X_dev (the caller keeps it alive and unchanged for the object's lifetime).  The passes run on a stream of the
 * object's own; zk_rows_adopt therefore waits (hipDeviceSynchronize) for whatever is still writing X_dev -- e.g. an
 * asynchronous zk_transform_patches_dev on the caller's stream -- before it returns. */
int zk_rows_adopt(int device, const double* X_dev, int64_t n_rows, int n_features, zk_rows** out);
int zk_rows_destroy(zk_rows* rows);
const double* zk_rows_data(const zk_rows* rows);                 /* the device matrix */
/* Column means and population variances (two passes, as numpy.mean / numpy.var); the means become the centring shift of
 * the k-means calls (scikit-learn subtracts them before clustering); n_bad_out = rows with a non-finite element. */
int zk_rows_center(zk_rows* rows, double* mean_out, double* var_out, int64_t* n_bad_out);
/* The two halves of zk_rows_center for a matrix whose rows are spread over several ranks: local column sums; then, with the
 * mean over ALL ranks, the local sums of squared deviations (and the centring shift, row norms, finiteness count). */
int zk_rows_colsum(zk_rows* rows, double* sums_out);
int zk_rows_center_at(zk_rows* rows, const double* mean, double* sqsum_out, int64_t* n_bad_out);
int zk_rows_fetch(zk_rows* rows, const int64_t* idx, int n, int centred, double* rows_out);   /* (n, D) to the host */
int zk_rows_reset_labels(zk_rows* rows);                         /* labels := -1 (a new k-means run) */
int zk_rows_labels(zk_rows* rows, int32_t* labels_host);         /* (N) labels of the last k-means / E step */
const int32_t* zk_rows_labels_dev(const zk_rows* rows);
/* k-means++ seeding (sklearn _kmeans_plusplus): squared distances of every centred row to t <= 8 candidate rows
 * (cand (t, D) centred, cand_sq their squared norms), max(0, (-2 x.c + |c|^2) + |x|^2), folded with the closest distance
 * so far when use_closest; pot_out[c] = their sum over the rows.  zk_kmeans_seed_pick adopts candidate `which` as the
 * closest-distance row and returns searchsorted(cumsum(closest), vals) clipped to N - 1 (n_vals may be 0). */
int zk_kmeans_seed_step(zk_rows* rows, const double* cand, const double* cand_sq, int t, int use_closest, double* pot_out);
int zk_kmeans_seed_pick(zk_rows* rows, int which, const double* vals, int n_vals, int64_t* idx_out);
/* One Lloyd pass (sklearn lloyd_iter_chunked_dense) with the centred centres (k, D), k <= 256: label = first argmin_c
 * (|c|^2 - 2 x.c); with `update` sums_out (k, D) = sum of the centred rows per cluster, counts_out (k);
 * n_changed_out = rows whose label changed. */
int zk_kmeans_step(zk_rows* rows, const double* centers, int k, int update, double* sums_out, double* counts_out,
                   int64_t* n_changed_out);
int zk_kmeans_own_distance(zk_rows* rows, const double* centers, int k, double* dist_host);   /* |x - c[label]|^2, (N) */
/* HIP-event time of the Lloyd kernel inside the zk_kmeans_step calls that follow (measurement aid of bench.py). */
int zk_rows_profile(zk_rows* rows, int enable);
double zk_rows_last_kernel_ms(const zk_rows* rows);
/* Gaussian mixture (sklearn _estimate_log_gaussian_prob / _estimate_log_prob_resp / _estimate_gaussian_parameters):
 * E step with upper-triangular precision Cholesky factors (k, D, D), means (k, D), log-determinants and log weights (k):
 * responsibilities (k, N) and labels (first argmax) stay on the device, lse_sum_out = sum_r logsumexp_c; the M step's
 * sums of `count` (1 to 3) consecutive components about `shift` in one pass over the matrix: gram_out (count, D+1, D+1), each
 * sum_r resp[c][r] [x_r - shift | 1]^T [x_r - shift | 1]. */
int zk_gmm_estep(zk_rows* rows, const double* prec_chol, const double* means, const double* log_det, const double* log_w, int k,
                 int want_resp, double* lse_sum_out);
int zk_gmm_resp_from_labels(zk_rows* rows, int k);               /* one-hot of the current labels */
int zk_gmm_moments(zk_rows* rows, int component, int count, const double* shift, double* gram_out);  /* count 1..8 per pass (1..3 with more than 47 features) */
/* The same sums with unit weights (the Gram matrix about `shift` with the column sums and N: what a covariance needs). */
int zk_rows_gram(zk_rows* rows, const double* shift, double* gram_out);

/* ------------------------------------------------------------------------------------------------------
 * The manifold consumer: ForceGraph8, reference manifold/force_relaxed.py:285-366 (csrc/zk_graph.hip).
 *   zk_rows_knn_correlation  compute_graph's neighbour search (:67-74: sklearn NearestNeighbors(metric='correlation'), brute
 *                            force) on the resident matrix: ind_out / dist_out (N, k), self first, sorted by distance; with
 *                            P_out also calculate_asymmetric_Pij (:17-52) for the given local_connectivity / perplexity.
 *   zk_force_layout_stage    optimize_stage (:236-266), the strictly sequential force-directed sweep over the graph's pairs
 *                            with its tau_rand_int negative sampling: HOST code here as in the reference (numba), operation
 *                            for operation; xy (n_nodes, 2) and rng_state (3) are updated in place, log_out (optional) receives
 *                            xy after every sweep.  force_params = (N, M, alpha, beta).
 * ------------------------------------------------------------------------------------------------------ */
int zk_rows_knn_correlation(zk_rows* rows, int n_neighbors, int local_connectivity, double perplexity, int64_t* ind_out,
                            double* dist_out, double* P_out);
/* numpy.random.RandomState.choice(n, p = uniform) from its one uniform draw u (host arithmetic, no n-sized arrays): the first
 * seed of scikit-learn's k-means++. */
int zk_uniform_choice_index(int64_t n, double u, int64_t* index_out);
/* test hooks of the above: the same index by plain sequential additions (and the last cumulative sum), and the float64 sum of
 * `steps` sequential additions of c -- what numpy.cumsum of a constant array holds at index steps - 1 -- by the jumping method */
int zk_uniform_choice_index_sequential(int64_t n, double u, int64_t* index_out, double* last_out);
double zk_repeated_sum_f64(double c, int64_t steps);
int zk_force_layout_stage(double* xy, int64_t n_nodes, const int64_t* node1, const int64_t* node2, const double* weight,
                          int64_t n_pairs, const int64_t* nbrs_ind, int n_neighbors, int64_t num_iterations,
                          const double* force_params, int num_negative_samples, double learning_rate, int64_t* rng_state,
                          double* log_out);

/* Device memory for callers that have no allocator of their own (a NumPy / C user of the *_dev entry points). */
int zk_device_malloc(int device, int64_t bytes, void** out_dev);
int zk_device_free(int device, void* dev);
int zk_device_copy(int device, void* dst, const void* src, int64_t bytes, int kind /* 1 H2D, 2 D2H, 3 D2D */);
int zk_device_synchronize(int device);

/* What this GPU sustains on a plain stream over a caller's buffers, measured with HIP events (bench.py prints it beside the
 * batch kernel's figure, so that a slow box, an unlucky physical placement, or the cost of mixing reads and writes shows as
 * such).  `bytes` of src_dev are read in 256-KiB groups (rounded down):
 *   dst_dev == NULL                          every wave reads contiguous 16-KiB stages through the LDS-DMA engine
 *                                            (global_load_lds nt, the batch kernel's own load instruction) and discards them;
 *   dst_dev != NULL, store_per_group == 0    a 16-B-per-lane copy src -> dst (non-temporal loads and stores);
 *   dst_dev != NULL, store_per_group  > 0    the read stream above, and after each group its wave writes store_per_group
 *                                            bytes (a multiple of 16) to dst_dev + group * store_per_group as 1-KiB
 *                                            non-temporal runs: the batch kernel's traffic mix without its arithmetic
 *                                            (23040 = 64 patches x 45 moments per 256 KiB of float32 32-px patches).
 * *ms_out = average over `reps` launches after one warm-up. */
int zk_hbm_probe(int device, const void* src_dev, void* dst_dev, int64_t bytes, int64_t store_per_group, int reps, double* ms_out);

/* The shader clock while other work runs on the device (measurement aid of bench.py; no reference counterpart): start puts ONE
 * wave on a stream of its own that samples the shader-clock counter against the constant 100-MHz counter and then naps until
 * stop is called or `max_ms` (<= 2000) have passed; stop returns the mean clock in GHz over the `ms` the wave was resident.
 * The FP64-bound kernels of this library run at 1.7-2.2 GHz of the nominal 2.4 (power management), so their rates are stated
 * against the measured clock as well as the nominal one. */
long long zk_debug_strip3_launches(void);  /* launches of the opt-in dense kernel zk_frame_strip3_kernel (ZK_STRIP_V3=1) so far: tests */
typedef struct zk_clock_monitor zk_clock_monitor;
int zk_clock_monitor_start(int device, double max_ms, zk_clock_monitor** out);
int zk_clock_monitor_stop(zk_clock_monitor* monitor, double* ghz_out, double* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* ZERNIKE_HIP_H */
