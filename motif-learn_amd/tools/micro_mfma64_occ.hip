// micro_mfma64_occ.hip -- v_mfma_f64_16x16x4_f64 rate against the number of waves per SIMD and of independent accumulator chains per
// wave (measurement aid for the kNN / mixture kernels; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o bin/micro_mfma64_occ micro_mfma64_occ.hip && ./bin/micro_mfma64_occ
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int CH>
__global__ __launch_bounds__(256) void rate_kernel(double* out, int iters) {
  v4d c[CH];
  for (int i = 0; i < CH; ++i) c[i] = v4d{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8 / CH; ++r)
#pragma unroll
      for (int j = 0; j < CH; ++j) c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[j], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < CH; ++i) s += c[i][i & 3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
void run(double* d, int wgs_per_cu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000, blocks = 256 * wgs_per_cu;
  rate_kernel<CH><<<blocks, 256>>>(d, 100);
  hipEventRecord(e0);
  rate_kernel<CH><<<blocks, 256>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2.0 * 1024 * 8.0 * iters * (blocks * 4.0);
  printf("chains %d  waves/SIMD %d: %.2f ms  %.1f TFLOP/s  (%.1f clocks per MFMA and SIMD at 2.4 GHz)\n", CH, wgs_per_cu, ms, flop / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (8.0 * iters * wgs_per_cu));
}

int main() {
  double* d;
  hipMalloc(&d, 1 << 24);
  for (int w : {1, 2, 4}) {
    run<1>(d, w);
    run<2>(d, w);
    run<4>(d, w);
    run<8>(d, w);
  }
  return 0;
}
