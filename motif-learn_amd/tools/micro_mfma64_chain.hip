// micro_mfma64_chain.hip -- v_mfma_f64_16x16x4_f64 rate against the LENGTH of the dependent accumulator chain: L MFMAs into one
// accumulator (started from zero), the four results read by the vector pipe (one v_add each), next chain.  What a kernel whose
// 16 x 16 output blocks need only L k-steps can reach (kNN: L = 12; mixture E step: 4 / 8 / 12).  Measurement aid.
//   hipcc --offload-arch=gfx950 -O3 -o bin/micro_mfma64_chain micro_mfma64_chain.hip && ./bin/micro_mfma64_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int L>
__global__ __launch_bounds__(256) void chain_kernel(double* out, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4, s = 0.0;
  for (int i = 0; i < iters; ++i) {
    v4d c = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < L; ++j) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    s += c[0] + c[1] + c[2] + c[3];
    a += 1e-9;  // (keeps the chains from being hoisted)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int L>
void run(double* d, int wgs_per_cu) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 96000 / L, blocks = 256 * wgs_per_cu;
  chain_kernel<L><<<blocks, 256>>>(d, 10);
  (void)hipEventRecord(e0);
  chain_kernel<L><<<blocks, 256>>>(d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2.0 * 1024 * (double)L * iters * (blocks * 4.0);
  printf("chain %3d  waves/SIMD %d: %6.1f TFLOP/s  (%.1f clocks per MFMA and SIMD at 2.4 GHz)\n", L, wgs_per_cu, flop / ms / 1e9,
         ms * 1e-3 * 2.4e9 / ((double)L * iters * wgs_per_cu));
}

int main() {
  double* d;
  (void)hipMalloc(&d, 1 << 24);
  for (int w : {1, 2, 3, 4}) {
    run<4>(d, w);
    run<8>(d, w);
    run<12>(d, w);
    run<24>(d, w);
    run<96>(d, w);
  }
  return 0;
}
