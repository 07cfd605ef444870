// micro_power.hip -- what the float64 instruction mixes of the dense kernels cost in board power (not part of the library).
// One variant per process, run back to back for a few seconds while tools/power_mix.py samples rocm-smi:
//   V  v_fma_f64, all operands in vector registers (16 independent accumulators per lane)
//   S  v_fma_f64 with one operand in a scalar register that a streaming s_load_dwordx16 keeps replacing (the strip kernels' form)
//   L  V plus one ds_read_b64 per four FMAs (the strip kernels read a pixel pair per ~5 operations)
//   M  v_mfma_f64_16x16x4_f64 from registers (8 independent accumulator blocks)
//   W  V plus a float64 store of every lane every 128 FMAs (~2.4 TB/s of writes at full rate: the dense kernels' output stream)
// build: hipcc --offload-arch=gfx950 -O3 -o bin/micro_power micro_power.hip ; run: bin/micro_power V 4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>

#define CK(x)                                                                               \
  do {                                                                                      \
    hipError_t e_ = (x);                                                                    \
    if (e_ != hipSuccess) {                                                                 \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));     \
      exit(1);                                                                              \
    }                                                                                       \
  } while (0)

constexpr int NACC = 16;

template <int MODE>  // 0 V, 1 S, 2 L, 4 W
__global__ __launch_bounds__(256, 2) void fma_kernel(const double* __restrict__ tab, int iters, double* __restrict__ out, double seed) {
  __shared__ double lds[2048];
  for (int e = threadIdx.x; e < 2048; e += 256) lds[e] = seed + e * 1e-9;
  __syncthreads();
  double acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) acc[j] = j * 1e-3;
  double x = seed + threadIdx.x * 1e-6, y = 1.0 - 1e-9;
  const __attribute__((address_space(4))) double* st = (const __attribute__((address_space(4))) double*)tab;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1) {
      // 16 scalar operands per 16 FMAs, streamed from a 32-KiB table (scalar-cache resident after the first pass)
      const __attribute__((address_space(4))) double* row = st + ((it & 255) * 16);
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_fma(row[j], x, acc[j]);
    } else {
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_fma(y, x, acc[j]);
    }
    if (MODE == 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) x += lds[(threadIdx.x + 64 * q + it) & 2047] * 1e-30;
    }
    if (MODE == 4 && (it & 7) == 7) __builtin_nontemporal_store(acc[it & 15], out + ((size_t)blockIdx.x * 256 + threadIdx.x) + (size_t)(it >> 3 & 63) * gridDim.x * 256);
  }
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < NACC; ++j) s += acc[j];
  if (s == 12345.678) out[0] = s + x;
}

__global__ __launch_bounds__(256, 2) void mfma_kernel(int iters, double* __restrict__ out, double seed) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  v4d acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = v4d{0.0, 0.0, 0.0, 0.0};
  const double a = seed + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (s == 12345.678) out[0] = s;
}

int main(int argc, char** argv) {
  const char mode = argc > 1 ? argv[1][0] : 'V';
  const double secs = argc > 2 ? atof(argv[2]) : 4.0;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 2;  // two workgroups of four waves per CU: two waves per SIMD
  double *tab, *out;
  CK(hipMalloc(&tab, 4096 * sizeof(double)));
  CK(hipMemset(tab, 0, 4096 * sizeof(double)));
  CK(hipMalloc(&out, (size_t)blocks * 256 * 64 * sizeof(double)));
  const int iters = mode == 'M' ? 20000 : 40000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double ms_sum = 0.0;
  int launches = 0;
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    CK(hipEventRecord(e0));
    for (int k = 0; k < 4; ++k) {
      switch (mode) {
        case 'V': hipLaunchKernelGGL((fma_kernel<0>), dim3(blocks), dim3(256), 0, 0, tab, iters, out, 0.5); break;
        case 'S': hipLaunchKernelGGL((fma_kernel<1>), dim3(blocks), dim3(256), 0, 0, tab, iters, out, 0.5); break;
        case 'L': hipLaunchKernelGGL((fma_kernel<2>), dim3(blocks), dim3(256), 0, 0, tab, iters, out, 0.5); break;
        case 'W': hipLaunchKernelGGL((fma_kernel<4>), dim3(blocks), dim3(256), 0, 0, tab, iters, out, 0.5); break;
        default: hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(256), 0, 0, iters, out, 0.5); break;
      }
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms_sum += ms;
    launches += 4;
  }
  const double flops = mode == 'M' ? (double)blocks * 4 * iters * 8 * 2048.0 : (double)blocks * 256 * (double)iters * NACC * 2.0;
  const double written = mode == 'W' ? (double)blocks * 256 * (iters / 8) * 8.0 : 0.0;
  printf("%c: %d launches, %.3f ms each, %.1f TFLOP/s float64", mode, launches, ms_sum / launches, flops * launches / (ms_sum * 1e-3) / 1e12);
  if (written > 0) printf(", %.2f TB/s written", written * launches / (ms_sum * 1e-3) / 1e12);
  printf("\n");
  return 0;
}
