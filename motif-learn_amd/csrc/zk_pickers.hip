// zk_pickers.hip -- device side of the parameter pickers that choose (size, n_max) for ZPs:
//   estimate_patch_size  reference features/_patch_size.py:221-302  (window autocorrelations -> radial profile -> peak)
//   estimate_n_max       reference features/_estimate_n_max.py:108-125 (FFT denoising, windowed patch power spectra ->
//                        radial cumulative energy)
// The FFTs are hipFFT's (a third-party numeric substrate, as pocketfft is in the reference: scipy.signal.correlate,
// scipy.fft.fft2, numpy.fft.fft2), bound at run time like RCCL so that the library loads without it; everything
// around them -- standardisation, zero padding, power spectra, lag cropping and averaging, the polar resampling the
// reference gets from skimage.transform.warp_polar, the exact top-k coefficient selection of denoise_fft -- is
// written here.  The 1-D peak search / cumulative-energy logic stays on the host (mtflearn_amd/features/pickers.py).
//
// All transforms are complex-to-complex float64 (one-off, per-image work: simplicity over the factor 2).
#include <dlfcn.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include <hipfft/hipfft.h>  // types and prototypes only; nothing here links against libhipfft

#include "zk_internal.h"

namespace {

struct fft_api {
  void* handle = nullptr;
  decltype(&hipfftPlanMany) PlanMany = nullptr;
  decltype(&hipfftExecZ2Z) ExecZ2Z = nullptr;
  decltype(&hipfftSetStream) SetStream = nullptr;
  decltype(&hipfftDestroy) Destroy = nullptr;
};
fft_api g_fft;

int fft_load() {
  if (g_fft.handle) return 0;
  void* h = dlopen("libhipfft.so.0", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  if (!h) h = dlopen("libhipfft.so", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  if (!h) {
    Dl_info info;
    if (dladdr((const void*)&hipGetDeviceCount, &info) && info.dli_fname) {
      std::string dir(info.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        dir.resize(slash + 1);
        for (const char* name : {"libhipfft.so.0", "libhipfft.so"}) {
          h = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_GLOBAL);
          if (h) break;
        }
      }
    }
  }
  if (!h) h = dlopen("libhipfft.so.0", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/libhipfft.so.0", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return zk_fail(ZK_E_FFT, std::string("cannot load libhipfft: ") + (dlerror() ? dlerror() : "?"));
  fft_api a;
  a.handle = h;
#define ZK_SYM(field, name)                                  \
  a.field = (decltype(a.field))dlsym(h, name);               \
  if (!a.field) return zk_fail(ZK_E_FFT, std::string("libhipfft lacks the symbol ") + name)
  ZK_SYM(PlanMany, "hipfftPlanMany");
  ZK_SYM(ExecZ2Z, "hipfftExecZ2Z");
  ZK_SYM(SetStream, "hipfftSetStream");
  ZK_SYM(Destroy, "hipfftDestroy");
#undef ZK_SYM
  g_fft = a;
  return 0;
}

#define ZK_FFT(call)                                                                         \
  do {                                                                                       \
    const hipfftResult zk_f_ = (call);                                                       \
    if (zk_f_ != HIPFFT_SUCCESS) return zk_fail(ZK_E_FFT, std::string(#call) + " failed with hipfftResult " + std::to_string((int)zk_f_)); \
  } while (0)

// Batched 2-D complex plans are kept between calls (round 3): building one costs milliseconds even when rocFFT has its
// kernels already, and the pickers are called with the same shapes over and over (every window batch of one frame size, every
// patch set of one size).  A plan in use is checked out of the cache and goes back when its holder dies; the cache keeps the
// four most recently returned ones and destroys what falls out.
struct fft_cached {
  int device, ny, nx, batch;
  hipfftHandle h;
};
static std::mutex g_fft_cache_mutex;
static std::vector<fft_cached> g_fft_cache;  // most recently returned last

struct fft_plan {
  hipfftHandle h = 0;
  bool live = false;
  fft_cached key = {};
  ~fft_plan() {
    if (!live) return;
    hipfftHandle drop = 0;
    bool have_drop = false;
    {
      std::lock_guard<std::mutex> lock(g_fft_cache_mutex);
      g_fft_cache.push_back(key);
      if (g_fft_cache.size() > 4) {
        drop = g_fft_cache.front().h;
        have_drop = true;
        g_fft_cache.erase(g_fft_cache.begin());
      }
    }
    if (have_drop) (void)g_fft.Destroy(drop);
  }
  int make(int ny, int nx, int batch) {
    int device = 0;
    ZK_HIP(hipGetDevice(&device));
    {
      std::lock_guard<std::mutex> lock(g_fft_cache_mutex);
      for (size_t i = g_fft_cache.size(); i-- > 0;)
        if (g_fft_cache[i].device == device && g_fft_cache[i].ny == ny && g_fft_cache[i].nx == nx && g_fft_cache[i].batch == batch) {
          key = g_fft_cache[i];
          h = key.h;
          live = true;
          g_fft_cache.erase(g_fft_cache.begin() + (long)i);
          return 0;
        }
    }
    int n[2] = {ny, nx};
    ZK_FFT(g_fft.PlanMany(&h, 2, n, nullptr, 1, ny * nx, nullptr, 1, ny * nx, HIPFFT_Z2Z, batch));
    key = fft_cached{device, ny, nx, batch, h};
    live = true;
    return 0;
  }
};

struct dev_buf {
  void* p = nullptr;
  ~dev_buf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    ZK_HIP(hipMalloc(&p, bytes ? bytes : 16));
    return 0;
  }
  template <typename T>
  T* as() const {
    return (T*)p;
  }
};

typedef hipfftDoubleComplex cplx;

__device__ __forceinline__ double block_sum(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  return t;
}

// mean and population standard deviation of each window (reference standardize_image, _patch_size.py:9-19);
// one block per window, two passes (mean first: the windows are positive images, E[x^2] - mean^2 would cancel)
template <typename T>
__global__ __launch_bounds__(256) void window_stats_kernel(const T* __restrict__ img, int W, const int32_t* __restrict__ org,
                                                           int ws, double* __restrict__ stats) {
  __shared__ double sh[4];
  const int b = blockIdx.x;
  const T* base = img + (long long)org[2 * b] * W + org[2 * b + 1];
  const int n = ws * ws;
  double s = 0.0;
  for (int t = threadIdx.x; t < n; t += 256) s += (double)base[(long long)(t / ws) * W + t % ws];
  const double mean = block_sum(s, sh) / (double)n;
  double q = 0.0;
  for (int t = threadIdx.x; t < n; t += 256) {
    const double d = (double)base[(long long)(t / ws) * W + t % ws] - mean;
    q += d * d;
  }
  const double var = block_sum(q, sh) / (double)n;
  if (threadIdx.x == 0) {
    stats[2 * b] = mean;
    stats[2 * b + 1] = sqrt(var);
  }
}

// window (optionally standardised, optionally multiplied by an outer-product window function) into the top-left corner of
// an N x N complex field, zero elsewhere
template <typename T>
__global__ __launch_bounds__(256) void fill_kernel(const T* __restrict__ img, int W, const int32_t* __restrict__ org, int ws,
                                                   const double* __restrict__ stats, const double* __restrict__ win, int N,
                                                   cplx* __restrict__ out, long long total) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long long per = (long long)N * N;
  const int b = (int)(t / per);
  const int r = (int)((t - b * per) / N), c = (int)(t - b * per - (long long)r * N);
  double v = 0.0;
  if (r < ws && c < ws) {
    v = (double)img[(long long)(org[2 * b] + r) * W + org[2 * b + 1] + c];
    if (stats) v = (v - stats[2 * b]) / stats[2 * b + 1];
    if (win) v *= win[r] * win[c];
  }
  out[t].x = v;
  out[t].y = 0.0;
}

__global__ __launch_bounds__(256) void power_kernel(cplx* __restrict__ z, long long total) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const double re = z[t].x, im = z[t].y;
  z[t].x = re * re + im * im;
  z[t].y = 0.0;
}

// acc[i][j] += sum_b Re ac_b[(i - ws/2) mod N][(j - ws/2) mod N] * scale: the 'same' crop of the linear autocorrelation
// (scipy.signal.correlate mode='same': lags -(ws/2) .. ws - 1 - ws/2, zero lag at index ws/2)
__global__ __launch_bounds__(256) void crop_accumulate_kernel(const cplx* __restrict__ z, int B, int N, int ws, double scale,
                                                              double* __restrict__ acc) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ws * ws) return;
  const int i = t / ws, j = t - i * ws;
  const int r = (i - ws / 2 + N) % N, c = (j - ws / 2 + N) % N;
  double s = 0.0;
  for (int b = 0; b < B; ++b) s += z[((long long)b * N + r) * N + c].x;
  acc[t] += s * scale;
}

// |fftshift(F)|^2 of a batch of size x size transforms (numpy.fft.fftshift: out[i] = in[(i - n/2) mod n])
__global__ __launch_bounds__(256) void shift_power_kernel(const cplx* __restrict__ z, int n, double* __restrict__ out,
                                                          long long total) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long long per = (long long)n * n;
  const long long b = t / per;
  const int i = (int)((t - b * per) / n), j = (int)(t - b * per - (long long)i * n);
  const cplx v = z[b * per + (long long)((i - n / 2 + n) % n) * n + (j - n / 2 + n) % n];
  out[t] = v.x * v.x + v.y * v.y;
}

// min / max of each (h, w) item -> mm[2 b], mm[2 b + 1] (one block per item)
__global__ __launch_bounds__(256) void minmax_kernel(const double* __restrict__ data, long long per, double* __restrict__ mm) {
  __shared__ double lo_s[256], hi_s[256];
  const double* base = data + blockIdx.x * per;
  double lo = INFINITY, hi = -INFINITY;
  for (long long t = threadIdx.x; t < per; t += 256) {
    const double v = base[t];
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
  lo_s[threadIdx.x] = lo;
  hi_s[threadIdx.x] = hi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      lo_s[threadIdx.x] = fmin(lo_s[threadIdx.x], lo_s[threadIdx.x + o]);
      hi_s[threadIdx.x] = fmax(hi_s[threadIdx.x], hi_s[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    mm[2 * blockIdx.x] = lo_s[0];
    mm[2 * blockIdx.x + 1] = hi_s[0];
  }
}

// Radial profile through a polar resampling: what the reference's radial_profile (_patch_size.py:48-100) obtains from
// skimage.transform.warp_polar(data, center=(h//2, w//2), scaling='linear') -- scikit-image is not installed here, so
// this restates its published algorithm (skimage/transform/_warps.py: warp_polar, _linear_polar_mapping, warp with
// order=1, mode='constant', cval=0, clip=True; _warps_cy.pyx: bilinear_interpolation):
//   output (360 angles, R = ceil(sqrt((h/2)^2 + (w/2)^2)) radii); sample (a, x) at
//     row = x / k_r * sin(a / k_a) + center_row, col = x / k_r * cos(a / k_a) + center_col,  k_a = 360 / 2 pi, k_r = R / radius
//   bilinear between floor / ceil neighbours, pixels outside the array count as 0, result clipped to the input's [min, max]
//   (values equal to the fill value 0 are kept when 0 lies outside that range).
// One thread per (item, radius); the 360 angles are aggregated in the thread: method 0 mean, 1 max, 2 sum.
__global__ __launch_bounds__(128) void polar_profile_kernel(const double* __restrict__ data, int h, int w, int ci, int cj, int R,
                                                            double radius, int method, const double* __restrict__ mm,
                                                            double* __restrict__ out) {
  const int x = blockIdx.x * 128 + threadIdx.x;
  const int b = blockIdx.y;
  if (x >= R) return;
  const double* img = data + (long long)b * h * w;
  const double lo = mm[2 * b], hi = mm[2 * b + 1];
  const bool keep_fill = !(lo <= 0.0 && 0.0 <= hi);
  const double k_angle = 360.0 / (2.0 * M_PI), k_radius = (double)R / radius;
  const double rad = (double)x / k_radius;
  auto px = [&](long long r, long long c) -> double { return (r < 0 || r >= h || c < 0 || c >= w) ? 0.0 : img[r * w + c]; };
  double agg = method == 1 ? -INFINITY : 0.0;
  for (int a = 0; a < 360; ++a) {
    const double ang = (double)a / k_angle;
    const double rr = rad * sin(ang) + (double)ci, cc = rad * cos(ang) + (double)cj;
    const double fr = floor(rr), fc = floor(cc);
    const long long r0 = (long long)fr, c0 = (long long)fc, r1 = (long long)ceil(rr), c1 = (long long)ceil(cc);
    const double dr = rr - fr, dc = cc - fc;
    const double top = (1.0 - dc) * px(r0, c0) + dc * px(r0, c1);
    const double bot = (1.0 - dc) * px(r1, c0) + dc * px(r1, c1);
    double v = (1.0 - dr) * top + dr * bot;
    if (!(keep_fill && v == 0.0)) v = v < lo ? lo : (v > hi ? hi : v);
    if (method == 1) agg = (v > agg || v != v) ? v : agg;
    else agg += v;
  }
  out[(long long)b * R + x] = method == 0 ? agg / 360.0 : agg;
}

// ---- exact k-th largest of non-negative doubles: radix select on the IEEE bit patterns, 16 bits per pass ----------
__global__ __launch_bounds__(256) void radix_hist_kernel(const double* __restrict__ v, int stride, long long n,
                                                         unsigned long long prefix, int shift, unsigned int* __restrict__ hist) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const unsigned long long key = (unsigned long long)__double_as_longlong(v[t * stride]);
  if (shift < 48 && (key >> (shift + 16)) != prefix) return;
  atomicAdd(&hist[(key >> shift) & 0xffffu], 1u);
}

// spectrum *= mask: keep power > T, and the first `ties` elements (in arrival order) with power == T
__global__ __launch_bounds__(256) void mask_kernel(cplx* __restrict__ spec, const cplx* __restrict__ power, long long n,
                                                   unsigned long long tkey, unsigned int ties, unsigned int* __restrict__ counter) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const unsigned long long key = (unsigned long long)__double_as_longlong(power[t].x);
  bool keep = key > tkey;
  if (key == tkey) keep = atomicAdd(counter, 1u) < ties;
  if (!keep) {
    spec[t].x = 0.0;
    spec[t].y = 0.0;
  }
}

__global__ __launch_bounds__(256) void real_scale_kernel(const cplx* __restrict__ z, double scale, double* __restrict__ out,
                                                         long long n) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < n) out[t] = z[t].x * scale;
}

template <typename T>
__global__ __launch_bounds__(256) void to_complex_kernel(const T* __restrict__ in, cplx* __restrict__ out, long long n) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < n) {
    out[t].x = (double)in[t];
    out[t].y = 0.0;
  }
}

inline unsigned blocks_of(long long n, int per = 256) { return (unsigned)((n + per - 1) / per); }

// Bit pattern of the k-th largest (k >= 1) of n non-negative doubles v[0], v[stride], ... and the number of strictly
// larger ones: radix select, four passes of 16-bit histograms (`d_hist`: 65536 counters on the device).
int kth_largest(const double* d_v, int stride, long long n, long long k, unsigned int* d_hist, unsigned long long* key_out,
                long long* above_out) {
  unsigned long long prefix = 0;
  long long above = 0;
  std::vector<unsigned int> hist(65536);
  for (int shift = 48; shift >= 0; shift -= 16) {
    ZK_HIP(hipMemset(d_hist, 0, 65536 * 4));
    hipLaunchKernelGGL(radix_hist_kernel, dim3(blocks_of(n)), dim3(256), 0, 0, d_v, stride, n, prefix, shift, d_hist);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpy(hist.data(), d_hist, 65536 * 4, hipMemcpyDeviceToHost));
    int bin = 65535;
    for (; bin > 0; --bin) {
      if (above + (long long)hist[(size_t)bin] >= k) break;
      above += hist[(size_t)bin];
    }
    prefix = (prefix << 16) | (unsigned long long)bin;
  }
  *key_out = prefix;
  *above_out = above;
  return 0;
}

// |dd| coefficients of a single-level db2 DWT with symmetric extension (PyWavelets dwtn(image, 'db2')['dd']):
// out[i][j] = | sum_{a,b} h[a] h[b] ext(2 i + 1 + 3 - a, 2 j + 1 + 3 - b) |, ext = half-sample mirrored image
template <typename T>
__global__ __launch_bounds__(256) void db2_dd_kernel(const T* __restrict__ img, int H, int W, int oh, int ow, double* __restrict__ out) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)oh * ow) return;
  const int i = (int)(t / ow), j = (int)(t - (long long)i * ow);
  const double h[4] = {-0.48296291314469025, 0.836516303737469, -0.22414386804185735, -0.12940952255092145};
  auto mirror = [](int e, int n) {  // index into the signal of position e of the extension (3 mirrored samples either side)
    int q = e - 3;
    if (q < 0) q = -1 - q;
    if (q >= n) q = 2 * n - 1 - q;
    return q < 0 ? 0 : (q >= n ? n - 1 : q);
  };
  double s = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = mirror(2 * i + 4 - a, H);
    double rs = 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) rs += h[b] * (double)img[(long long)r * W + mirror(2 * j + 4 - b, W)];
    s += h[a] * rs;
  }
  out[t] = fabs(s);
}

__global__ __launch_bounds__(256) void count_nonzero_kernel(const double* __restrict__ v, long long n, unsigned long long* __restrict__ count) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long mask = __ballot(t < n && v[t] != 0.0);
  if ((threadIdx.x & 63) == 0 && mask) atomicAdd(count, (unsigned long long)__popcll(mask));
}

int check_image(int dtype, int64_t H, int64_t W, const void* p) {
  if (dtype != ZK_F32 && dtype != ZK_F64) return zk_fail(ZK_E_BADARG, "dtype must be ZK_F32 or ZK_F64");
  if (H <= 0 || W <= 0 || H > 0x3fffffff || W > 0x3fffffff) return zk_fail(ZK_E_BADARG, "bad image shape");
  if (!p) return zk_fail(ZK_E_BADARG, "null host pointer");
  return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
extern "C" int zk_autocorr_mean(int device, const void* image_host, int dtype, int64_t H, int64_t W, int64_t window,
                                const int32_t* origins_yx, int n_windows, int standardize, double* out_host) {
  int rc = check_image(dtype, H, W, image_host);
  if (rc) return rc;
  if (window <= 0 || window > H || window > W || window > 8192) return zk_fail(ZK_E_BADARG, "bad window size");
  if (n_windows <= 0 || !origins_yx || !out_host) return zk_fail(ZK_E_BADARG, "need windows and an output array");
  for (int b = 0; b < n_windows; ++b)
    if (origins_yx[2 * b] < 0 || origins_yx[2 * b] + window > H || origins_yx[2 * b + 1] < 0 || origins_yx[2 * b + 1] + window > W)
      return zk_fail(ZK_E_BADARG, "window outside the image");
  if ((rc = fft_load())) return rc;
  ZK_ON_DEVICE(device);
  const int ws = (int)window, N = 2 * ws;  // any N >= 2 ws - 1 turns the circular correlation into the linear one
  const size_t es = dtype == ZK_F32 ? 4 : 8;
  dev_buf d_img, d_org, d_stats, d_z, d_acc;
  if ((rc = d_img.alloc((size_t)H * W * es)) || (rc = d_org.alloc((size_t)n_windows * 8)) ||
      (rc = d_stats.alloc((size_t)n_windows * 16)) || (rc = d_acc.alloc((size_t)ws * ws * 8)))
    return rc;
  ZK_HIP(hipMemcpy(d_img.p, image_host, (size_t)H * W * es, hipMemcpyHostToDevice));
  ZK_HIP(hipMemcpy(d_org.p, origins_yx, (size_t)n_windows * 8, hipMemcpyHostToDevice));
  ZK_HIP(hipMemset(d_acc.p, 0, (size_t)ws * ws * 8));
  if (standardize) {
    if (dtype == ZK_F32)
      hipLaunchKernelGGL(window_stats_kernel<float>, dim3(n_windows), dim3(256), 0, 0, d_img.as<float>(), (int)W,
                         d_org.as<int32_t>(), ws, d_stats.as<double>());
    else
      hipLaunchKernelGGL(window_stats_kernel<double>, dim3(n_windows), dim3(256), 0, 0, d_img.as<double>(), (int)W,
                         d_org.as<int32_t>(), ws, d_stats.as<double>());
    ZK_HIP(hipGetLastError());
    std::vector<double> st((size_t)2 * n_windows);
    ZK_HIP(hipMemcpy(st.data(), d_stats.p, st.size() * 8, hipMemcpyDeviceToHost));
    for (int b = 0; b < n_windows; ++b)
      if (!(st[2 * b + 1] > 0.0)) return zk_fail(ZK_E_BADARG, "Standard deviation is zero, can't standardize the image.");
  }
  // batches of at most ~1 GiB of complex field
  int B = (int)std::max<long long>(1, ((long long)1 << 30) / ((long long)N * N * 16));
  B = std::min(B, n_windows);
  if ((rc = d_z.alloc((size_t)B * N * N * 16))) return rc;
  fft_plan plan, plan_tail;
  if ((rc = plan.make(N, N, B))) return rc;
  for (int first = 0; first < n_windows; first += B) {
    const int nb = std::min(B, n_windows - first);
    fft_plan* use = &plan;
    if (nb != B) {
      if ((rc = plan_tail.make(N, N, nb))) return rc;
      use = &plan_tail;
    }
    const long long total = (long long)nb * N * N;
    const double* st = standardize ? d_stats.as<double>() + 2 * first : nullptr;
    if (dtype == ZK_F32)
      hipLaunchKernelGGL(fill_kernel<float>, dim3(blocks_of(total)), dim3(256), 0, 0, d_img.as<float>(), (int)W,
                         d_org.as<int32_t>() + 2 * first, ws, st, (const double*)nullptr, N, d_z.as<cplx>(), total);
    else
      hipLaunchKernelGGL(fill_kernel<double>, dim3(blocks_of(total)), dim3(256), 0, 0, d_img.as<double>(), (int)W,
                         d_org.as<int32_t>() + 2 * first, ws, st, (const double*)nullptr, N, d_z.as<cplx>(), total);
    ZK_HIP(hipGetLastError());
    ZK_FFT(g_fft.ExecZ2Z(use->h, d_z.as<cplx>(), d_z.as<cplx>(), HIPFFT_FORWARD));
    hipLaunchKernelGGL(power_kernel, dim3(blocks_of(total)), dim3(256), 0, 0, d_z.as<cplx>(), total);
    ZK_FFT(g_fft.ExecZ2Z(use->h, d_z.as<cplx>(), d_z.as<cplx>(), HIPFFT_BACKWARD));
    hipLaunchKernelGGL(crop_accumulate_kernel, dim3(blocks_of((long long)ws * ws)), dim3(256), 0, 0, d_z.as<cplx>(), nb, N, ws,
                       1.0 / ((double)N * N * n_windows), d_acc.as<double>());
    ZK_HIP(hipGetLastError());
  }
  ZK_HIP(hipMemcpy(out_host, d_acc.p, (size_t)ws * ws * 8, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int64_t zk_polar_radii(int64_t h, int64_t w) {
  if (h <= 0 || w <= 0) return 0;
  return (int64_t)ceil(sqrt((double)h / 2.0 * ((double)h / 2.0) + (double)w / 2.0 * ((double)w / 2.0)));
}

extern "C" int zk_polar_profile(int device, const double* data_host, int64_t batch, int64_t h, int64_t w, int64_t center_row,
                                int64_t center_col, int method, double* out_host) {
  if (center_row < 0) center_row = h / 2;  // fftshift convention (reference _patch_size.py:72-74)
  if (center_col < 0) center_col = w / 2;
  if (!data_host || !out_host || batch <= 0 || h <= 0 || w <= 0 || h > 32768 || w > 32768 || batch > 65535)
    return zk_fail(ZK_E_BADARG, "bad arguments");
  if (method < 0 || method > 2) return zk_fail(ZK_E_BADARG, "Invalid method. Must be 'mean', 'max', or 'sum'.");
  ZK_ON_DEVICE(device);
  const int R = (int)zk_polar_radii(h, w);
  const double radius = sqrt((double)h / 2.0 * ((double)h / 2.0) + (double)w / 2.0 * ((double)w / 2.0));
  dev_buf d_in, d_mm, d_out;
  int rc;
  if ((rc = d_in.alloc((size_t)batch * h * w * 8)) || (rc = d_mm.alloc((size_t)batch * 16)) ||
      (rc = d_out.alloc((size_t)batch * R * 8)))
    return rc;
  ZK_HIP(hipMemcpy(d_in.p, data_host, (size_t)batch * h * w * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)batch), dim3(256), 0, 0, d_in.as<double>(), (long long)h * w, d_mm.as<double>());
  hipLaunchKernelGGL(polar_profile_kernel, dim3(blocks_of(R, 128), (unsigned)batch), dim3(128), 0, 0, d_in.as<double>(), (int)h,
                     (int)w, (int)center_row, (int)center_col, R, radius, method, d_mm.as<double>(), d_out.as<double>());
  ZK_HIP(hipGetLastError());
  ZK_HIP(hipMemcpy(out_host, d_out.p, (size_t)batch * R * 8, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int zk_power_spectra(int device, const void* image_host, int dtype, int64_t H, int64_t W, int64_t size,
                                const int32_t* origins_yx, int n_windows, const double* window_1d, double* out_host) {
  int rc = check_image(dtype, H, W, image_host);
  if (rc) return rc;
  if (size <= 0 || size > H || size > W || size > 4096) return zk_fail(ZK_E_BADARG, "bad patch size");
  if (n_windows <= 0 || !origins_yx || !out_host) return zk_fail(ZK_E_BADARG, "need windows and an output array");
  for (int b = 0; b < n_windows; ++b)
    if (origins_yx[2 * b] < 0 || origins_yx[2 * b] + size > H || origins_yx[2 * b + 1] < 0 || origins_yx[2 * b + 1] + size > W)
      return zk_fail(ZK_E_BADARG, "window outside the image");
  if ((rc = fft_load())) return rc;
  ZK_ON_DEVICE(device);
  const int n = (int)size;
  const size_t es = dtype == ZK_F32 ? 4 : 8;
  const long long total = (long long)n_windows * n * n;
  dev_buf d_img, d_org, d_win, d_z, d_out;
  if ((rc = d_img.alloc((size_t)H * W * es)) || (rc = d_org.alloc((size_t)n_windows * 8)) || (rc = d_win.alloc((size_t)n * 8)) ||
      (rc = d_z.alloc((size_t)total * 16)) || (rc = d_out.alloc((size_t)total * 8)))
    return rc;
  ZK_HIP(hipMemcpy(d_img.p, image_host, (size_t)H * W * es, hipMemcpyHostToDevice));
  ZK_HIP(hipMemcpy(d_org.p, origins_yx, (size_t)n_windows * 8, hipMemcpyHostToDevice));
  if (window_1d) ZK_HIP(hipMemcpy(d_win.p, window_1d, (size_t)n * 8, hipMemcpyHostToDevice));
  const double* win = window_1d ? d_win.as<double>() : nullptr;
  if (dtype == ZK_F32)
    hipLaunchKernelGGL(fill_kernel<float>, dim3(blocks_of(total)), dim3(256), 0, 0, d_img.as<float>(), (int)W, d_org.as<int32_t>(), n,
                       (const double*)nullptr, win, n, d_z.as<cplx>(), total);
  else
    hipLaunchKernelGGL(fill_kernel<double>, dim3(blocks_of(total)), dim3(256), 0, 0, d_img.as<double>(), (int)W, d_org.as<int32_t>(),
                       n, (const double*)nullptr, win, n, d_z.as<cplx>(), total);
  ZK_HIP(hipGetLastError());
  fft_plan plan;
  if ((rc = plan.make(n, n, n_windows))) return rc;
  ZK_FFT(g_fft.ExecZ2Z(plan.h, d_z.as<cplx>(), d_z.as<cplx>(), HIPFFT_FORWARD));
  hipLaunchKernelGGL(shift_power_kernel, dim3(blocks_of(total)), dim3(256), 0, 0, d_z.as<cplx>(), n, d_out.as<double>(), total);
  ZK_HIP(hipGetLastError());
  ZK_HIP(hipMemcpy(out_host, d_out.p, (size_t)total * 8, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int zk_denoise_fft(int device, const void* image_host, int dtype, int64_t H, int64_t W, double p, double* out_host) {
  int rc = check_image(dtype, H, W, image_host);
  if (rc) return rc;
  if (!(p > 0.0 && p <= 1.0)) return zk_fail(ZK_E_BADARG, "Fraction p must be between 0 and 1.");
  if (!out_host) return zk_fail(ZK_E_BADARG, "null host pointer");
  if (H > 32768 || W > 32768) return zk_fail(ZK_E_BADARG, "image too large for one transform");
  if ((rc = fft_load())) return rc;
  ZK_ON_DEVICE(device);
  const long long n = (long long)H * W;
  const size_t es = dtype == ZK_F32 ? 4 : 8;
  dev_buf d_img, d_spec, d_pow, d_hist, d_out;
  if ((rc = d_img.alloc((size_t)n * es)) || (rc = d_spec.alloc((size_t)n * 16)) || (rc = d_pow.alloc((size_t)n * 16)) ||
      (rc = d_hist.alloc(65537 * 4)) || (rc = d_out.alloc((size_t)n * 8)))
    return rc;
  ZK_HIP(hipMemcpy(d_img.p, image_host, (size_t)n * es, hipMemcpyHostToDevice));
  if (dtype == ZK_F32)
    hipLaunchKernelGGL(to_complex_kernel<float>, dim3(blocks_of(n)), dim3(256), 0, 0, d_img.as<float>(), d_spec.as<cplx>(), n);
  else
    hipLaunchKernelGGL(to_complex_kernel<double>, dim3(blocks_of(n)), dim3(256), 0, 0, d_img.as<double>(), d_spec.as<cplx>(), n);
  ZK_HIP(hipGetLastError());
  fft_plan plan;
  if ((rc = plan.make((int)H, (int)W, 1))) return rc;
  ZK_FFT(g_fft.ExecZ2Z(plan.h, d_spec.as<cplx>(), d_spec.as<cplx>(), HIPFFT_FORWARD));
  ZK_HIP(hipMemcpy(d_pow.p, d_spec.p, (size_t)n * 16, hipMemcpyDeviceToDevice));
  hipLaunchKernelGGL(power_kernel, dim3(blocks_of(n)), dim3(256), 0, 0, d_pow.as<cplx>(), n);
  // k = ceil(p n) coefficients survive (reference _denoise_fft.py:33-38): find the k-th largest power exactly
  long long k = (long long)ceil(p * (double)n);
  k = k < 1 ? 1 : (k > n ? n : k);
  unsigned long long prefix = 0;
  long long above = 0;
  if ((rc = kth_largest((const double*)d_pow.p, 2, n, k, d_hist.as<unsigned int>(), &prefix, &above))) return rc;
  // `prefix` is now the bit pattern of the k-th largest power; `above` counts the strictly larger ones
  ZK_HIP(hipMemset(d_hist.p, 0, 4));
  hipLaunchKernelGGL(mask_kernel, dim3(blocks_of(n)), dim3(256), 0, 0, d_spec.as<cplx>(), d_pow.as<cplx>(), n, prefix,
                     (unsigned int)(k - above), d_hist.as<unsigned int>());
  ZK_HIP(hipGetLastError());
  ZK_FFT(g_fft.ExecZ2Z(plan.h, d_spec.as<cplx>(), d_spec.as<cplx>(), HIPFFT_BACKWARD));
  hipLaunchKernelGGL(real_scale_kernel, dim3(blocks_of(n)), dim3(256), 0, 0, d_spec.as<cplx>(), 1.0 / (double)n, d_out.as<double>(), n);
  ZK_HIP(hipGetLastError());
  ZK_HIP(hipMemcpy(out_host, d_out.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  return 0;
}

// skimage.restoration.estimate_sigma for a 2-D image (restated, see mtflearn_amd/features/pickers.py): the median of the
// non-zero |db2 diagonal detail coefficients| / 0.6745.  The coefficients and the two order statistics of the median are
// computed on the device.
extern "C" int zk_wavelet_sigma(int device, const void* image_host, int dtype, int64_t H, int64_t W, double* sigma_out) {
  int rc = check_image(dtype, H, W, image_host);
  if (rc) return rc;
  if (!sigma_out) return zk_fail(ZK_E_BADARG, "null pointer");
  if (H < 4 || W < 4) return zk_fail(ZK_E_BADARG, "image too small for a db2 decomposition (needs at least 4 x 4)");
  ZK_ON_DEVICE(device);
  const int oh = (int)((H + 3) / 2), ow = (int)((W + 3) / 2);
  const long long n = (long long)oh * ow;
  const size_t es = dtype == ZK_F32 ? 4 : 8;
  dev_buf d_img, d_d, d_hist;
  if ((rc = d_img.alloc((size_t)H * W * es)) || (rc = d_d.alloc((size_t)n * 8)) || (rc = d_hist.alloc(65536 * 4 + 8))) return rc;
  ZK_HIP(hipMemcpy(d_img.p, image_host, (size_t)H * W * es, hipMemcpyHostToDevice));
  if (dtype == ZK_F32)
    hipLaunchKernelGGL(db2_dd_kernel<float>, dim3(blocks_of(n)), dim3(256), 0, 0, d_img.as<float>(), (int)H, (int)W, oh, ow, d_d.as<double>());
  else
    hipLaunchKernelGGL(db2_dd_kernel<double>, dim3(blocks_of(n)), dim3(256), 0, 0, d_img.as<double>(), (int)H, (int)W, oh, ow, d_d.as<double>());
  ZK_HIP(hipGetLastError());
  unsigned long long* d_count = (unsigned long long*)(d_hist.as<char>() + 65536 * 4);
  ZK_HIP(hipMemset(d_count, 0, 8));
  hipLaunchKernelGGL(count_nonzero_kernel, dim3(blocks_of(n)), dim3(256), 0, 0, d_d.as<double>(), n, d_count);
  ZK_HIP(hipGetLastError());
  unsigned long long m = 0;
  ZK_HIP(hipMemcpy(&m, d_count, 8, hipMemcpyDeviceToHost));
  if (m == 0) {  // numpy: median of an empty array is NaN
    *sigma_out = NAN;
    return 0;
  }
  // numpy.median of the m non-zero values (the zeros are the smallest keys, so ranks among the largest are unaffected):
  // the (m + 1) / 2-th largest for odd m, the mean of the m / 2-th and the (m / 2 + 1)-th largest for even m
  unsigned long long key = 0;
  long long above = 0;
  double lo, hi;
  if ((rc = kth_largest(d_d.as<double>(), 1, n, (long long)(m / 2 + 1), d_hist.as<unsigned int>(), &key, &above))) return rc;
  memcpy(&lo, &key, 8);
  hi = lo;
  if (m % 2 == 0) {
    if ((rc = kth_largest(d_d.as<double>(), 1, n, (long long)(m / 2), d_hist.as<unsigned int>(), &key, &above))) return rc;
    memcpy(&hi, &key, 8);
  }
  *sigma_out = 0.5 * (lo + hi) / 0.6744897501960817;
  return 0;
}
