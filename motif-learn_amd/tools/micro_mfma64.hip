// micro_mfma64.hip -- rate and layout of v_mfma_f64_16x16x4_f64 on gfx950 (measurement aid; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o bin/micro_mfma64 micro_mfma64.hip && ./bin/micro_mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void rate_kernel(double* out, int iters) {
  v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ void fma_kernel(double* out, int iters) {
  double c[16];
  for (int i = 0; i < 16; ++i) c[i] = i;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = __builtin_fma(a, b, c[j]);
  double s = 0;
  for (int i = 0; i < 16; ++i) s += c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// D = A (16 x 4) B (4 x 16): lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]
__global__ void layout_kernel(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];   // the guide's map: row = (lane >> 4) + 4 reg, col = lane & 15
}

int main() {
  double* d;
  hipMalloc(&d, 1 << 24);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000, blocks = 256 * 8, threads = 256;
  for (int pass = 0; pass < 2; ++pass) {
    hipEventRecord(e0);
    rate_kernel<<<blocks, threads>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 1024 * 4.0 * iters * (blocks * threads / 64);
    printf("mfma_f64_16x16x4: %.2f ms  %.1f TFLOP/s\n", ms, flop / ms / 1e9);
    hipEventRecord(e0);
    fma_kernel<<<blocks, threads>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double flop2 = 2.0 * 16.0 * iters * blocks * threads;
    printf("v_fma_f64:        %.2f ms  %.1f TFLOP/s\n", ms, flop2 / ms / 1e9);
  }
  std::vector<double> A(64), B(64), D(256), R(256, 0.0);
  for (int i = 0; i < 64; ++i) A[i] = 1 + i * 0.5, B[i] = 2 - i * 0.25;
  double *dA, *dB, *dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  layout_kernel<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j];
      worst = fmax(worst, fabs(s - D[i * 16 + j]));
    }
  printf("layout check: max |D - A B| = %.3e\n", worst);
  return 0;
}
