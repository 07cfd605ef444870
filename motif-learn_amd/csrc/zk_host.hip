// zk_host.hip -- the host-buffer entry points of the C ABI (what ZPs.transform calls with NumPy arrays):
// zk_transform_patches / zk_transform_frame / zk_transform_points / zk_frame_maps.
//
// A job is cut into chunks of at most `host_chunk` bytes of input + output (default 256 MiB); chunk c goes
//     H2D on the ring's copy-in stream  ->  kernel on the plan's stream  ->  D2H on the ring's copy-out stream
// through one of ZK_RING_SLOTS device slots, consecutive chunks overlapping (events order the three streams;
// the host only waits when it needs a slot back).  The device footprint is the ring (<= 3 chunks), whatever
// the job size: a 4096^2 frame at n_max 10 needs 0.8 GB of staging instead of the 9 GB result.
//
// Host memory: copies straight from / into the caller's arrays.  Page-locked arrays (zk_host_alloc -- the
// Python layer hands its results out of a pool of them -- or anything hipHostRegister'ed) move by DMA at
// link speed and fully asynchronously; pageable arrays go through the runtime's own staging, which blocks
// the calling thread per copy (the kernel of the chunk before still overlaps with it).
#include <string.h>

#include "zk_internal.h"

#define ZK_RING_SLOTS 3


struct zk_host_ring {
  void* d_in[ZK_RING_SLOTS] = {};
  size_t in_cap[ZK_RING_SLOTS] = {};
  void* d_out[ZK_RING_SLOTS] = {};
  size_t out_cap[ZK_RING_SLOTS] = {};
  hipEvent_t ev_in[ZK_RING_SLOTS] = {};    // H2D of the slot's chunk done
  hipEvent_t ev_k[ZK_RING_SLOTS] = {};     // kernel of the slot's chunk done
  hipEvent_t ev_out[ZK_RING_SLOTS] = {};   // D2H of the slot's chunk done
  bool out_busy[ZK_RING_SLOTS] = {};
  hipStream_t s_in = nullptr, s_out = nullptr;
  void* d_frame = nullptr;                 // whole frame (+ points) of the dense / key-point calls
  size_t frame_cap = 0;
  void* d_raw[ZK_RING_SLOTS] = {};         // narrow inputs (ZK_U8 / U16 / I16) as they arrive, before widening
  size_t raw_cap[ZK_RING_SLOTS] = {};
};

namespace {

size_t elem_size(int dtype) { return dtype == ZK_U8 ? 1 : (dtype == ZK_U16 || dtype == ZK_I16) ? 2 : dtype == ZK_F32 ? 4 : 8; }
bool is_narrow(int dtype) { return dtype == ZK_U8 || dtype == ZK_U16 || dtype == ZK_I16; }
int kernel_dtype(int dtype) { return is_narrow(dtype) ? ZK_F32 : dtype; }  // what the kernels read

int check_dtype(int dtype) {
  if (dtype < ZK_F32 || dtype > ZK_I16) return zk_fail(ZK_E_BADARG, "dtype must be ZK_F32, ZK_F64, ZK_U8, ZK_U16 or ZK_I16");
  return 0;
}

// narrow integers -> float32 (exact), 4 elements per thread
template <typename S>
__global__ __launch_bounds__(256) void widen_kernel(const S* __restrict__ in, float* __restrict__ out, long long n) {
  const long long t = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (t + 3 < n) {
    float4 v;
    v.x = (float)in[t];
    v.y = (float)in[t + 1];
    v.z = (float)in[t + 2];
    v.w = (float)in[t + 3];
    *(float4*)(out + t) = v;
  } else {
    for (long long k = t; k < n; ++k) out[k] = (float)in[k];
  }
}

int widen(int dtype, const void* in, float* out, long long n, hipStream_t s) {
  const unsigned blocks = (unsigned)((n + 1023) / 1024);
  if (dtype == ZK_U8) hipLaunchKernelGGL(widen_kernel<uint8_t>, dim3(blocks), dim3(256), 0, s, (const uint8_t*)in, out, n);
  else if (dtype == ZK_U16) hipLaunchKernelGGL(widen_kernel<uint16_t>, dim3(blocks), dim3(256), 0, s, (const uint16_t*)in, out, n);
  else hipLaunchKernelGGL(widen_kernel<int16_t>, dim3(blocks), dim3(256), 0, s, (const int16_t*)in, out, n);
  ZK_HIP(hipGetLastError());
  return 0;
}

size_t chunk_bytes(const zk_plan* p) {
  if (p->host_chunk) return p->host_chunk;
  if (const char* e = getenv("ZK_HOST_CHUNK_MB")) {
    const long mb = atol(e);
    if (mb > 0) return (size_t)mb << 20;
  }
  return (size_t)256 << 20;
}

int ring_get(zk_plan* p, zk_host_ring** out) {
  if (!p->ring) {
    zk_host_ring* r = new (std::nothrow) zk_host_ring();
    if (!r) return zk_fail(ZK_E_NOMEM, "out of host memory");
    p->ring = r;
    ZK_HIP(hipStreamCreateWithFlags(&r->s_in, hipStreamNonBlocking));
    ZK_HIP(hipStreamCreateWithFlags(&r->s_out, hipStreamNonBlocking));
    for (int k = 0; k < ZK_RING_SLOTS; ++k) {
      ZK_HIP(hipEventCreateWithFlags(&r->ev_in[k], hipEventDisableTiming));
      ZK_HIP(hipEventCreateWithFlags(&r->ev_k[k], hipEventDisableTiming));
      ZK_HIP(hipEventCreateWithFlags(&r->ev_out[k], hipEventDisableTiming));
    }
  }
  for (int k = 0; k < ZK_RING_SLOTS; ++k) p->ring->out_busy[k] = false;
  *out = p->ring;
  return 0;
}

// wait until every copy of the job has landed, whatever happened before
int ring_drain(zk_plan* p, zk_host_ring* r, int rc) {
  const hipError_t a = hipStreamSynchronize(r->s_in), b = hipStreamSynchronize(p->stream),
                   c = hipStreamSynchronize(r->s_out);
  if (rc) return rc;
  if (a != hipSuccess) return zk_hip_fail(a, "hipStreamSynchronize(copy-in)");
  if (b != hipSuccess) return zk_hip_fail(b, "hipStreamSynchronize(kernel)");
  if (c != hipSuccess) return zk_hip_fail(c, "hipStreamSynchronize(copy-out)");
  return 0;
}

// Software pipeline over the chunks of a job: chunk c is staged in and its kernel enqueued BEFORE the copy-out
// of chunk c-1 is issued, so that a copy that blocks the calling thread (pageable memory) still runs under the
// kernel of the next chunk.  `launch(c, slot)` / `copy_out(c, slot)` return 0 or a negative code; the streams are
// drained before returning, whatever happened.
template <class L, class O>
int run_chunks(zk_plan* p, zk_host_ring* r, int n_chunks, L launch, O copy_out) {
  int rc = 0;
  for (int c = 0; c <= n_chunks && !rc; ++c) {
    if (c < n_chunks) rc = launch(c, c % ZK_RING_SLOTS);
    if (c > 0 && !rc) rc = copy_out(c - 1, (c - 1) % ZK_RING_SLOTS);
  }
  return ring_drain(p, r, rc);
}

// the slot is about to be reused: its previous D2H must have left d_out, its previous kernel d_in
int slot_acquire(zk_host_ring* r, int slot) {
  if (r->out_busy[slot]) {
    ZK_HIP(hipEventSynchronize(r->ev_out[slot]));
    r->out_busy[slot] = false;
  }
  return 0;
}

}  // namespace

void zk_host_release(zk_plan* p) {
  zk_host_ring* r = p->ring;
  if (!r) return;
  if (r->s_in) (void)hipStreamSynchronize(r->s_in);
  if (r->s_out) (void)hipStreamSynchronize(r->s_out);
  for (int k = 0; k < ZK_RING_SLOTS; ++k) {
    if (r->d_in[k]) (void)hipFree(r->d_in[k]);
    if (r->d_raw[k]) (void)hipFree(r->d_raw[k]);
    if (r->d_out[k]) (void)hipFree(r->d_out[k]);
    if (r->ev_in[k]) (void)hipEventDestroy(r->ev_in[k]);
    if (r->ev_k[k]) (void)hipEventDestroy(r->ev_k[k]);
    if (r->ev_out[k]) (void)hipEventDestroy(r->ev_out[k]);
  }
  if (r->d_frame) (void)hipFree(r->d_frame);
  if (r->s_in) (void)hipStreamDestroy(r->s_in);
  if (r->s_out) (void)hipStreamDestroy(r->s_out);
  delete r;
  p->ring = nullptr;
}

extern "C" int zk_plan_set_host_chunk(zk_plan* p, int64_t bytes) {
  if (!p || bytes < 0) return zk_fail(ZK_E_BADARG, "bad arguments");
  p->host_chunk = (size_t)bytes;
  return 0;
}

extern "C" int zk_plan_release_staging(zk_plan* p) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  ZK_ON_PLAN_DEVICE(p);
  ZK_HIP(hipStreamSynchronize(p->stream));
  zk_host_release(p);
  if (p->d_gather) (void)hipFree(p->d_gather);
  p->d_gather = nullptr;
  p->d_gather_bytes = 0;
  if (p->d_scratch) (void)hipFree(p->d_scratch);
  p->d_scratch = nullptr;
  p->d_scratch_bytes = 0;
  return 0;
}

extern "C" int zk_host_alloc(int64_t bytes, void** out) {
  if (!out || bytes < 0) return zk_fail(ZK_E_BADARG, "bad arguments");
  *out = nullptr;
  if (bytes == 0) return 0;
  ZK_HIP(hipHostMalloc(out, (size_t)bytes, hipHostMallocPortable));
  return 0;
}

extern "C" int zk_host_free(void* ptr) {
  if (!ptr) return 0;
  ZK_HIP(hipHostFree(ptr));
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// batch of patches
// ------------------------------------------------------------------------------------------------------
extern "C" int zk_transform_patches(zk_plan* p, const void* patches_host, int dtype, int64_t n_patches,
                                    double* out_host) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  int rc = check_dtype(dtype);
  if (rc) return rc;
  if (n_patches < 0) return zk_fail(ZK_E_BADARG, "negative patch count");
  if (n_patches == 0) return 0;
  if (!patches_host || !out_host) return zk_fail(ZK_E_BADARG, "null host pointer");
  ZK_ON_PLAN_DEVICE(p);
  zk_host_ring* r;
  if ((rc = ring_get(p, &r))) return rc;
  const int kdt = kernel_dtype(dtype);
  const bool narrow = is_narrow(dtype);
  const size_t px_per = (size_t)p->size * p->size;
  const size_t raw_unit = px_per * elem_size(dtype), in_unit = px_per * elem_size(kdt),
               out_unit = (size_t)p->n_poly * sizeof(double);
  int64_t chunk = (int64_t)(chunk_bytes(p) / (in_unit + out_unit)) & ~(int64_t)255;  // whole waves of 64 patches
  if (chunk < 256) chunk = 256;
  if (chunk > n_patches) chunk = n_patches;
  const int n_chunks = (int)((n_patches + chunk - 1) / chunk);
  auto count = [&](int c) { return n_patches - c * chunk < chunk ? n_patches - c * chunk : chunk; };
  p->job_units = n_patches;
  struct reset_units {
    zk_plan* p;
    ~reset_units() { p->job_units = 0; }
  } reset{p};
  return run_chunks(
      p, r, n_chunks,
      [&](int c, int slot) -> int {
        int e = slot_acquire(r, slot);
        if (!e) e = zk_ensure(&r->d_in[slot], &r->in_cap[slot], (size_t)chunk * in_unit);
        if (!e && narrow) e = zk_ensure(&r->d_raw[slot], &r->raw_cap[slot], (size_t)chunk * raw_unit);
        if (!e) e = zk_ensure(&r->d_out[slot], &r->out_cap[slot], (size_t)chunk * out_unit);
        if (e) return e;
        // the slot's previous kernels must have consumed the landing buffer before the next H2D overwrites it
        if (c >= ZK_RING_SLOTS) ZK_HIP(hipStreamWaitEvent(r->s_in, r->ev_k[slot], 0));
        ZK_HIP(hipMemcpyAsync(narrow ? r->d_raw[slot] : r->d_in[slot], (const char*)patches_host + (size_t)c * chunk * raw_unit,
                              (size_t)count(c) * raw_unit, hipMemcpyHostToDevice, r->s_in));
        ZK_HIP(hipEventRecord(r->ev_in[slot], r->s_in));
        ZK_HIP(hipStreamWaitEvent(p->stream, r->ev_in[slot], 0));
        if (narrow && (e = widen(dtype, r->d_raw[slot], (float*)r->d_in[slot], (long long)count(c) * (long long)px_per, p->stream)))
          return e;
        if ((e = zk_transform_patches_dev(p, r->d_in[slot], kdt, count(c), (double*)r->d_out[slot], p->stream))) return e;
        ZK_HIP(hipEventRecord(r->ev_k[slot], p->stream));
        return 0;
      },
      [&](int c, int slot) -> int {
        ZK_HIP(hipStreamWaitEvent(r->s_out, r->ev_k[slot], 0));
        ZK_HIP(hipMemcpyAsync((char*)out_host + (size_t)c * chunk * out_unit, r->d_out[slot], (size_t)count(c) * out_unit,
                              hipMemcpyDeviceToHost, r->s_out));
        ZK_HIP(hipEventRecord(r->ev_out[slot], r->s_out));
        r->out_busy[slot] = true;
        return 0;
      });
}

// ------------------------------------------------------------------------------------------------------
// dense frame: the frame goes up once, the result comes down a row band at a time
// ------------------------------------------------------------------------------------------------------
namespace {

// copy the (planes, nb, W) band in d_src into rows [b0, b0 + nb) of the host array (planes, H, W)
int band_to_host(double* host, const double* d_src, int64_t planes, int64_t H, int64_t W, int64_t b0, int64_t nb,
                 hipStream_t s) {
  if (nb == H) {
    ZK_HIP(hipMemcpyAsync(host, d_src, (size_t)planes * H * W * sizeof(double), hipMemcpyDeviceToHost, s));
    return 0;
  }
  ZK_HIP(hipMemcpy2DAsync(host + (size_t)b0 * W, (size_t)H * W * sizeof(double), d_src, (size_t)nb * W * sizeof(double),
                          (size_t)nb * W * sizeof(double), (size_t)planes, hipMemcpyDeviceToHost, s));
  return 0;
}

// the whole frame onto the device as the kernels' element type (narrow formats: raw bytes up, widened there);
// `bytes` = size in the kernels' type, `extra` = room behind it for the caller
int frame_up(zk_plan* p, zk_host_ring* r, const void* image_host, int dtype, long long n_px, size_t bytes, size_t extra) {
  int rc = zk_ensure(&r->d_frame, &r->frame_cap, bytes + extra);
  if (rc) return rc;
  if (!is_narrow(dtype)) {
    ZK_HIP(hipMemcpyAsync(r->d_frame, image_host, bytes, hipMemcpyHostToDevice, p->stream));
    return 0;
  }
  const size_t raw = (size_t)n_px * elem_size(dtype);
  if ((rc = zk_ensure(&r->d_raw[0], &r->raw_cap[0], raw))) return rc;
  ZK_HIP(hipMemcpyAsync(r->d_raw[0], image_host, raw, hipMemcpyHostToDevice, p->stream));
  return widen(dtype, r->d_raw[0], (float*)r->d_frame, n_px, p->stream);
}

int64_t band_rows(const zk_plan* p, size_t row_bytes, int64_t H) {
  int64_t band = (int64_t)(chunk_bytes(p) / row_bytes) & ~(int64_t)7;  // whole blocks of the dense kernels
  if (band < 8) band = 8;
  return band > H ? H : band;
}

}  // namespace

extern "C" int zk_transform_frame(zk_plan* p, const void* image_host, int dtype, int64_t H, int64_t W,
                                  double* out_host) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  int rc = check_dtype(dtype);
  if (rc) return rc;
  if (H <= 0 || W <= 0) return zk_fail(ZK_E_BADARG, "bad frame shape");
  if (!image_host || !out_host) return zk_fail(ZK_E_BADARG, "null host pointer");
  ZK_ON_PLAN_DEVICE(p);
  zk_host_ring* r;
  if ((rc = ring_get(p, &r))) return rc;
  const int kdt = kernel_dtype(dtype);
  if ((rc = frame_up(p, r, image_host, dtype, (long long)H * W, (size_t)H * W * elem_size(kdt), 0))) return ring_drain(p, r, rc);
  const size_t row_bytes = (size_t)p->n_poly * W * sizeof(double);
  const int64_t band = band_rows(p, row_bytes, H);
  const int n_chunks = (int)((H + band - 1) / band);
  auto rows = [&](int c) { return H - c * band < band ? H - c * band : band; };
  return run_chunks(
      p, r, n_chunks,
      [&](int c, int slot) -> int {
        int e = slot_acquire(r, slot);
        if (!e) e = zk_ensure(&r->d_out[slot], &r->out_cap[slot], (size_t)band * row_bytes);
        if (!e) e = zk_transform_frame_dev(p, r->d_frame, kdt, H, W, c * band, rows(c), (double*)r->d_out[slot], p->stream);
        if (e) return e;
        ZK_HIP(hipEventRecord(r->ev_k[slot], p->stream));
        return 0;
      },
      [&](int c, int slot) -> int {
        ZK_HIP(hipStreamWaitEvent(r->s_out, r->ev_k[slot], 0));
        const int e = band_to_host(out_host, (const double*)r->d_out[slot], p->n_poly, H, W, c * band, rows(c), r->s_out);
        if (e) return e;
        ZK_HIP(hipEventRecord(r->ev_out[slot], r->s_out));
        r->out_busy[slot] = true;
        return 0;
      });
}

// ------------------------------------------------------------------------------------------------------
// fused symmetry maps
// ------------------------------------------------------------------------------------------------------
extern "C" int zk_frame_maps(zk_plan* p, const void* image_host, int dtype, int64_t H, int64_t W,
                             const int32_t* folds, int n_folds, const int32_t* m_unselect, int n_unselect,
                             int p_norm, const double* theta, int n_theta, double* rot_host, double* abs_host,
                             double* mirror_host) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  int rc = check_dtype(dtype);
  if (rc) return rc;
  if (H <= 0 || W <= 0) return zk_fail(ZK_E_BADARG, "bad frame shape");
  if (!image_host) return zk_fail(ZK_E_BADARG, "null host pointer");
  ZK_ON_PLAN_DEVICE(p);
  zk_host_ring* r;
  if ((rc = ring_get(p, &r))) return rc;
  const int kdt = kernel_dtype(dtype);
  if ((rc = frame_up(p, r, image_host, dtype, (long long)H * W, (size_t)H * W * elem_size(kdt), 0))) return ring_drain(p, r, rc);
  const int64_t nc = zk_complex_count(zk_full_set_nmax(p));
  const int64_t pl_rot = rot_host ? n_folds : 0, pl_abs = abs_host ? nc : 0, pl_mir = mirror_host ? 1 : 0;
  const size_t row_bytes = (size_t)(pl_rot + pl_abs + pl_mir + 1) * W * sizeof(double);
  const int64_t band = band_rows(p, row_bytes, H);
  const int n_chunks = (int)((H + band - 1) / band);
  auto rows = [&](int c) { return H - c * band < band ? H - c * band : band; };
  auto d_rot = [&](int c, int slot) { return rot_host ? (double*)r->d_out[slot] : nullptr; };
  auto d_abs = [&](int c, int slot) { return abs_host ? (double*)r->d_out[slot] + (size_t)pl_rot * rows(c) * W : nullptr; };
  auto d_mir = [&](int c, int slot) {
    return mirror_host ? (double*)r->d_out[slot] + (size_t)(pl_rot + pl_abs) * rows(c) * W : nullptr;
  };
  return run_chunks(
      p, r, n_chunks,
      [&](int c, int slot) -> int {
        int e = slot_acquire(r, slot);
        if (!e) e = zk_ensure(&r->d_out[slot], &r->out_cap[slot], (size_t)band * row_bytes);
        if (!e)
          e = zk_frame_maps_dev(p, r->d_frame, kdt, H, W, c * band, rows(c), folds, n_folds, m_unselect, n_unselect, p_norm,
                                theta, n_theta, d_rot(c, slot), d_abs(c, slot), d_mir(c, slot), p->stream);
        if (e) return e;
        ZK_HIP(hipEventRecord(r->ev_k[slot], p->stream));
        return 0;
      },
      [&](int c, int slot) -> int {
        ZK_HIP(hipStreamWaitEvent(r->s_out, r->ev_k[slot], 0));
        int e = 0;
        if (rot_host) e = band_to_host(rot_host, d_rot(c, slot), pl_rot, H, W, c * band, rows(c), r->s_out);
        if (abs_host && !e) e = band_to_host(abs_host, d_abs(c, slot), pl_abs, H, W, c * band, rows(c), r->s_out);
        if (mirror_host && !e) e = band_to_host(mirror_host, d_mir(c, slot), 1, H, W, c * band, rows(c), r->s_out);
        if (e) return e;
        ZK_HIP(hipEventRecord(r->ev_out[slot], r->s_out));
        r->out_busy[slot] = true;
        return 0;
      });
}

// ------------------------------------------------------------------------------------------------------
// moments at key points: frame and point list go up once, the moments come down in chunks
// ------------------------------------------------------------------------------------------------------
extern "C" int zk_transform_points(zk_plan* p, const void* image_host, int dtype, int64_t H, int64_t W,
                                   const int32_t* points_host, int64_t n_points, double* out_host) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  int rc = check_dtype(dtype);
  if (rc) return rc;
  if (H <= 0 || W <= 0) return zk_fail(ZK_E_BADARG, "bad frame shape");
  if (n_points < 0) return zk_fail(ZK_E_BADARG, "negative point count");
  if (n_points == 0) return 0;
  if (!image_host || !points_host || !out_host) return zk_fail(ZK_E_BADARG, "null host pointer");
  ZK_ON_PLAN_DEVICE(p);
  zk_host_ring* r;
  if ((rc = ring_get(p, &r))) return rc;
  const int kdt = kernel_dtype(dtype);
  const size_t img_bytes = (size_t)H * W * elem_size(kdt);
  const size_t img_pad = (img_bytes + 255) & ~(size_t)255;
  const size_t pts_bytes = (size_t)n_points * 2 * sizeof(int32_t);
  if ((rc = frame_up(p, r, image_host, dtype, (long long)H * W, img_bytes, img_pad - img_bytes + pts_bytes))) return ring_drain(p, r, rc);
  int32_t* d_pts = (int32_t*)((char*)r->d_frame + img_pad);
  {
    const hipError_t e = hipMemcpyAsync(d_pts, points_host, pts_bytes, hipMemcpyHostToDevice, p->stream);
    if (e != hipSuccess) return ring_drain(p, r, zk_hip_fail(e, "hipMemcpyAsync(points)"));
  }
  const size_t out_unit = (size_t)p->n_poly * sizeof(double);
  int64_t chunk = (int64_t)(chunk_bytes(p) / out_unit) & ~(int64_t)255;
  if (chunk < 256) chunk = 256;
  if (chunk > n_points) chunk = n_points;
  const int n_chunks = (int)((n_points + chunk - 1) / chunk);
  auto count = [&](int c) { return n_points - c * chunk < chunk ? n_points - c * chunk : chunk; };
  return run_chunks(
      p, r, n_chunks,
      [&](int c, int slot) -> int {
        int e = slot_acquire(r, slot);
        if (!e) e = zk_ensure(&r->d_out[slot], &r->out_cap[slot], (size_t)chunk * out_unit);
        if (!e)
          e = zk_transform_points_dev(p, r->d_frame, kdt, H, W, d_pts + 2 * c * chunk, count(c), (double*)r->d_out[slot],
                                      p->stream);
        if (e) return e;
        ZK_HIP(hipEventRecord(r->ev_k[slot], p->stream));
        return 0;
      },
      [&](int c, int slot) -> int {
        ZK_HIP(hipStreamWaitEvent(r->s_out, r->ev_k[slot], 0));
        ZK_HIP(hipMemcpyAsync((char*)out_host + (size_t)c * chunk * out_unit, r->d_out[slot], (size_t)count(c) * out_unit,
                              hipMemcpyDeviceToHost, r->s_out));
        ZK_HIP(hipEventRecord(r->ev_out[slot], r->s_out));
        r->out_busy[slot] = true;
        return 0;
      });
}

// ------------------------------------------------------------------------------------------------------
// symmetry maps of a batch of moment vectors (rank-2 tail), and of the windows at key points (moments + tail fused:
// the (N, n_poly) matrix stays on the device)
// ------------------------------------------------------------------------------------------------------
namespace {

struct rows_out {
  double *rot, *ab, *mir;     // host destinations (may be null)
  int64_t pl_rot, pl_abs, pl_mir;
};

// D2H of one chunk of the three row-major outputs, packed on the device as [rot | abs | mirror]
int rows_to_host(const rows_out& o, const double* d, int64_t first, int64_t n, hipStream_t s) {
  if (o.rot) ZK_HIP(hipMemcpyAsync(o.rot + first * o.pl_rot, d, (size_t)n * o.pl_rot * 8, hipMemcpyDeviceToHost, s));
  if (o.ab) ZK_HIP(hipMemcpyAsync(o.ab + first * o.pl_abs, d + n * o.pl_rot, (size_t)n * o.pl_abs * 8, hipMemcpyDeviceToHost, s));
  if (o.mir) ZK_HIP(hipMemcpyAsync(o.mir + first, d + n * (o.pl_rot + o.pl_abs), (size_t)n * 8, hipMemcpyDeviceToHost, s));
  return 0;
}

}  // namespace

extern "C" int zk_moment_maps(zk_plan* p, const double* moments_host, int64_t n_rows, const int32_t* folds, int n_folds,
                              const int32_t* m_unselect, int n_unselect, int p_norm, const double* theta, int n_theta,
                              double* rot_host, double* abs_host, double* mirror_host) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  if (n_rows < 0) return zk_fail(ZK_E_BADARG, "negative row count");
  if (n_rows == 0) return 0;
  if (!moments_host) return zk_fail(ZK_E_BADARG, "null host pointer");
  ZK_ON_PLAN_DEVICE(p);
  zk_host_ring* r;
  int rc;
  if ((rc = ring_get(p, &r))) return rc;
  const int64_t nc = zk_complex_count(zk_full_set_nmax(p));
  const rows_out o = {rot_host, abs_host, mirror_host, rot_host ? n_folds : 0, abs_host ? nc : 0, mirror_host ? 1 : 0};
  const size_t in_unit = (size_t)p->n_poly * 8, out_unit = (size_t)(o.pl_rot + o.pl_abs + o.pl_mir + 1) * 8;
  int64_t chunk = (int64_t)(chunk_bytes(p) / (in_unit + out_unit)) & ~(int64_t)255;
  if (chunk < 256) chunk = 256;
  if (chunk > n_rows) chunk = n_rows;
  const int n_chunks = (int)((n_rows + chunk - 1) / chunk);
  auto count = [&](int c) { return n_rows - c * chunk < chunk ? n_rows - c * chunk : chunk; };
  return run_chunks(
      p, r, n_chunks,
      [&](int c, int slot) -> int {
        int e = slot_acquire(r, slot);
        if (!e) e = zk_ensure(&r->d_in[slot], &r->in_cap[slot], (size_t)chunk * in_unit);
        if (!e) e = zk_ensure(&r->d_out[slot], &r->out_cap[slot], (size_t)chunk * out_unit);
        if (e) return e;
        if (c >= ZK_RING_SLOTS) ZK_HIP(hipStreamWaitEvent(r->s_in, r->ev_k[slot], 0));
        ZK_HIP(hipMemcpyAsync(r->d_in[slot], moments_host + (size_t)c * chunk * p->n_poly, (size_t)count(c) * in_unit,
                              hipMemcpyHostToDevice, r->s_in));
        ZK_HIP(hipEventRecord(r->ev_in[slot], r->s_in));
        ZK_HIP(hipStreamWaitEvent(p->stream, r->ev_in[slot], 0));
        double* d = (double*)r->d_out[slot];
        const int64_t n = count(c);
        e = zk_moment_maps_dev(p, (const double*)r->d_in[slot], n, folds, n_folds, m_unselect, n_unselect, p_norm, theta, n_theta,
                               o.rot ? d : nullptr, o.ab ? d + n * o.pl_rot : nullptr,
                               o.mir ? d + n * (o.pl_rot + o.pl_abs) : nullptr, p->stream);
        if (e) return e;
        ZK_HIP(hipEventRecord(r->ev_k[slot], p->stream));
        return 0;
      },
      [&](int c, int slot) -> int {
        ZK_HIP(hipStreamWaitEvent(r->s_out, r->ev_k[slot], 0));
        const int e = rows_to_host(o, (const double*)r->d_out[slot], (int64_t)c * chunk, count(c), r->s_out);
        if (e) return e;
        ZK_HIP(hipEventRecord(r->ev_out[slot], r->s_out));
        r->out_busy[slot] = true;
        return 0;
      });
}

extern "C" int zk_points_maps(zk_plan* p, const void* image_host, int dtype, int64_t H, int64_t W, const int32_t* points_host,
                              int64_t n_points, const int32_t* folds, int n_folds, const int32_t* m_unselect, int n_unselect,
                              int p_norm, const double* theta, int n_theta, double* rot_host, double* abs_host,
                              double* mirror_host) {
  if (!p) return zk_fail(ZK_E_BADARG, "null plan");
  int rc = check_dtype(dtype);
  if (rc) return rc;
  if (H <= 0 || W <= 0) return zk_fail(ZK_E_BADARG, "bad frame shape");
  if (n_points < 0) return zk_fail(ZK_E_BADARG, "negative point count");
  if (n_points == 0) return 0;
  if (!image_host || !points_host) return zk_fail(ZK_E_BADARG, "null host pointer");
  ZK_ON_PLAN_DEVICE(p);
  zk_host_ring* r;
  if ((rc = ring_get(p, &r))) return rc;
  const int kdt = kernel_dtype(dtype);
  const size_t img_bytes = (size_t)H * W * elem_size(kdt);
  const size_t img_pad = (img_bytes + 255) & ~(size_t)255;
  const size_t pts_bytes = (size_t)n_points * 2 * sizeof(int32_t);
  if ((rc = frame_up(p, r, image_host, dtype, (long long)H * W, img_bytes, img_pad - img_bytes + pts_bytes))) return ring_drain(p, r, rc);
  int32_t* d_pts = (int32_t*)((char*)r->d_frame + img_pad);
  {
    const hipError_t e = hipMemcpyAsync(d_pts, points_host, pts_bytes, hipMemcpyHostToDevice, p->stream);
    if (e != hipSuccess) return ring_drain(p, r, zk_hip_fail(e, "hipMemcpyAsync(points)"));
  }
  const int64_t nc = zk_complex_count(zk_full_set_nmax(p));
  const rows_out o = {rot_host, abs_host, mirror_host, rot_host ? n_folds : 0, abs_host ? nc : 0, mirror_host ? 1 : 0};
  const size_t mom_unit = (size_t)p->n_poly * 8, out_unit = (size_t)(o.pl_rot + o.pl_abs + o.pl_mir + 1) * 8;
  int64_t chunk = (int64_t)(chunk_bytes(p) / (mom_unit + out_unit)) & ~(int64_t)255;
  if (chunk < 256) chunk = 256;
  if (chunk > n_points) chunk = n_points;
  const int n_chunks = (int)((n_points + chunk - 1) / chunk);
  auto count = [&](int c) { return n_points - c * chunk < chunk ? n_points - c * chunk : chunk; };
  return run_chunks(
      p, r, n_chunks,
      [&](int c, int slot) -> int {
        int e = slot_acquire(r, slot);
        if (!e) e = zk_ensure(&r->d_in[slot], &r->in_cap[slot], (size_t)chunk * mom_unit);  // the chunk's moments, device only
        if (!e) e = zk_ensure(&r->d_out[slot], &r->out_cap[slot], (size_t)chunk * out_unit);
        if (e) return e;
        const int64_t n = count(c);
        double* mom = (double*)r->d_in[slot];
        double* d = (double*)r->d_out[slot];
        if ((e = zk_transform_points_dev(p, r->d_frame, kdt, H, W, d_pts + 2 * c * chunk, n, mom, p->stream))) return e;
        e = zk_moment_maps_dev(p, mom, n, folds, n_folds, m_unselect, n_unselect, p_norm, theta, n_theta, o.rot ? d : nullptr,
                               o.ab ? d + n * o.pl_rot : nullptr, o.mir ? d + n * (o.pl_rot + o.pl_abs) : nullptr, p->stream);
        if (e) return e;
        ZK_HIP(hipEventRecord(r->ev_k[slot], p->stream));
        return 0;
      },
      [&](int c, int slot) -> int {
        ZK_HIP(hipStreamWaitEvent(r->s_out, r->ev_k[slot], 0));
        const int e = rows_to_host(o, (const double*)r->d_out[slot], (int64_t)c * chunk, count(c), r->s_out);
        if (e) return e;
        ZK_HIP(hipEventRecord(r->ev_out[slot], r->s_out));
        r->out_busy[slot] = true;
        return 0;
      });
}
