// micro_sfma.hip -- design microbenchmarks for the Zernike kernels (not part of the library).
//
//  A. f64 FMA rate when one operand streams in through scalar loads (s_load -> SGPR), the
//     other is a VGPR and the accumulators are VGPRs: the inner loop of every zk kernel.
//     Variants: NACC accumulators per lane fed by NACC doubles per step, P lanes-worth of
//     reuse of each scalar (P independent "pixels" per lane), table size (scalar-cache fit).
//  B. the same loop with register-only operands (the VALU ceiling).
//  C. HBM read patterns for the batch kernel: LDS-DMA of 128-B rows at a 4-KiB stride (one
//     patch row per 8 lanes, the row-pair staging of zk_sep_patches.hip) against a plain
//     contiguous LDS-DMA stream of the same bytes.
//
// build: hipcc --offload-arch=gfx950 -O3 -o micro_sfma micro_sfma.hip ; run: ./micro_sfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// ---------------------------------------------------------------- A / B: scalar-fed FMA
template <int NACC, int P, bool SCALAR>
__global__ __launch_bounds__(256) void sfma_kernel(const double* __restrict__ tab, int steps, int reps,
                                                   double* __restrict__ out, double seed) {
  double acc[P][NACC];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[p][j] = 0.0;
  double f[P];
#pragma unroll
  for (int p = 0; p < P; ++p) f[p] = seed + threadIdx.x * 1e-3 + p;
  for (int r = 0; r < reps; ++r) {
    for (int t = 0; t < steps; ++t) {
      const double* __restrict__ b = tab + (size_t)t * NACC;
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        const double bj = SCALAR ? b[j] : (double)(j + 1) * seed;
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p][j] = __builtin_fma(f[p], bj, acc[p][j]);
      }
#pragma unroll
      for (int p = 0; p < P; ++p) f[p] += 1e-9;  // keep the multiplicand loop-variant
    }
  }
  double s = 0;
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[p][j];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int P, bool SCALAR>
static void run_sfma(const char* name, const double* d_tab, int steps, int blocks, double* d_out) {
  const int reps = 40;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((sfma_kernel<NACC, P, SCALAR>), dim3(blocks), dim3(256), 0, 0, d_tab, steps, 2, d_out, 1.0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((sfma_kernel<NACC, P, SCALAR>), dim3(blocks), dim3(256), 0, 0, d_tab, steps, reps, d_out, 1.0);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double fma = (double)blocks * 256 * reps * steps * NACC * P;
  printf("%-34s steps=%5d tab=%7.1f KB blocks=%5d  %8.3f ms  %7.2f TFLOP/s(f64)\n", name, steps,
         steps * NACC * 8 / 1024.0, blocks, ms, 2 * fma / ms * 1e-9);
}

// ---------------------------------------------------------------- C: LDS-DMA read patterns
// Each wave (one workgroup) owns `64 patches` = 256 KiB contiguous and pulls it through a
// 16-KiB LDS buffer in 16 stages.  ROWPAIR: stage s = rows (s, 31-s) of all 64 patches (128-B
// lines, 4-KiB stride).  CONTIG: stage s = the s-th contiguous 16 KiB.
// HALF: as ROWPAIR, but only the middle 64 B of every 128-B row are requested (lanes fetching granules
// 2..5): does a partially requested line cost half the HBM traffic?
// AUX: cache policy of the LDS-DMA loads (0 default, 2 nt -- what the batch kernel uses).  STORE_B: bytes of results every
// wave writes per 64-patch group (0: none; 23040 = 64 x 45 float64 moments, written as 1-KiB non-temporal runs after the
// group's reads, like the batch kernel's epilogue) into `res`.
template <bool ROWPAIR, bool HALF = false, int AUX = 0, int STORE_B = 0, int SPOL = 0>
__global__ __launch_bounds__(64) void dma_kernel(const float* __restrict__ in, long long n_groups,
                                                 float* __restrict__ out, double* __restrict__ res = nullptr) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int lane = threadIdx.x;
  float sum = 0.f;
  for (long long g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const char* base = (const char*)in + g * 262144;
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const char* src;
        if (ROWPAIR) {
          const int seg = i >> 3, pg = i & 7;
          const int patch = pg * 8 + (lane >> 3);
          const int gran = ((lane & 7) - (patch >> 1)) & 7;
          const int row = seg ? 31 - s : s;
          src = base + patch * 4096 + row * 128 + gran * 16;
        } else {
          src = base + s * 16384 + i * 1024 + lane * 16;
        }
        if (!HALF || (((src - base) >> 4) & 7) - 2 < 4u)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(lds + i * 256), 16, 0, AUX);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      sum += lds[lane * 7 + s];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (STORE_B > 0) {
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2* dst = (d2*)((char*)res + g * STORE_B);
      const d2 v = {(double)sum, 1.0};
      for (int k = lane; k < STORE_B / 16; k += 64) __builtin_nontemporal_store(v, dst + k);
    }
  }
  out[(size_t)blockIdx.x * 64 + lane] = sum;
}

// F (round 3): the same read stream + stores, but the stores of all waves happen TOGETHER: a wave that has read its group
// arrives at a counter and waits (bounded spin: the barrier is a hint, not a dependency) until every wave of the grid has
// read its group of this round; then all write.  Reads and writes then reach HBM in separate phases instead of mixed.
template <int STORE_B>
__global__ __launch_bounds__(64) void dma_phased_kernel(const float* __restrict__ in, long long n_groups, float* __restrict__ out,
                                                        double* __restrict__ res, unsigned* __restrict__ counter, int spin_limit) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int lane = threadIdx.x;
  float sum = 0.f;
  unsigned round = 0;
  for (long long g = blockIdx.x; g < n_groups; g += gridDim.x, ++round) {
    const char* base = (const char*)in + g * 262144;
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + s * 16384 + i * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(lds + i * 256), 16, 0, 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      sum += lds[lane * 7 + s];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = (round + 1) * gridDim.x;
    for (int spin = 0; spin < spin_limit; ++spin) {
      if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
      __builtin_amdgcn_s_sleep(8);
    }
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2* dst = (d2*)((char*)res + g * STORE_B);
    const d2 v = {(double)sum, 1.0};
    for (int k = lane; k < STORE_B / 16; k += 64) __builtin_nontemporal_store(v, dst + k);
  }
  out[(size_t)blockIdx.x * 64 + lane] = sum;
}

__global__ __launch_bounds__(256) void write_only_kernel(double* __restrict__ res, long long n16) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2 v = {1.0, 2.0};
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) __builtin_nontemporal_store(v, (d2*)res + i);
}

template <bool ROWPAIR, bool HALF = false, int AUX = 0, int STORE_B = 0, int SPOL = 0>
static void run_dma(const char* name, const float* d_in, long long n_groups, int blocks, float* d_out, double* d_res = nullptr) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((dma_kernel<ROWPAIR, HALF, AUX, STORE_B, SPOL>), dim3(blocks), dim3(64), 0, 0, d_in, n_groups, d_out, d_res);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int it = 5;
  for (int k = 0; k < it; ++k)
    hipLaunchKernelGGL((dma_kernel<ROWPAIR, HALF, AUX, STORE_B, SPOL>), dim3(blocks), dim3(64), 0, 0, d_in, n_groups, d_out, d_res);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-34s blocks=%6d  %8.3f ms/launch  %8.1f GB/s\n", name, blocks, ms / it,
         (double)n_groups * 262144 * it / ms * 1e-6);
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);

  const int max_steps = 1 << 16;
  std::vector<double> h((size_t)max_steps * 48);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (double)(i % 977);
  double* d_tab;
  CK(hipMalloc((void**)&d_tab, h.size() * 8));
  CK(hipMemcpy(d_tab, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  double* d_out;
  CK(hipMalloc((void**)&d_out, (size_t)8192 * 256 * 8));

  const int cu = prop.multiProcessorCount;
  if (argc > 1 && argv[1][0] == 'E') {
    // ---- E (round 3): what bounds the batch kernel's stream?  16 GiB of "patches" (the bench's batch is 16.7 GB: far
    // beyond the 256-MiB Infinity Cache), 8 waves per CU; GB/s count the READ bytes only.
    const long long n_groups = 65536;  // x 256 KiB = 16 GiB
    float* d_in;
    CK(hipMalloc((void**)&d_in, (size_t)n_groups * 262144));
    CK(hipMemset(d_in, 0, (size_t)n_groups * 262144));
    float* d_fo;
    CK(hipMalloc((void**)&d_fo, (size_t)65536 * 64 * 4));
    double* d_res;
    CK(hipMalloc((void**)&d_res, (size_t)n_groups * 23040));
    printf("--- E: LDS-DMA read patterns over 16 GiB, cache policy, and the moment stores (GB/s of the reads) ---\n");
    for (int wpc : {8}) {
      printf("[%d waves per CU x 16 KiB in flight]\n", wpc);
      run_dma<false, false, 0>("contiguous, default policy", d_in, n_groups, cu * wpc, d_fo);
      run_dma<false, false, 2>("contiguous, nt", d_in, n_groups, cu * wpc, d_fo);
      run_dma<true, false, 0>("row pairs @ 4 KiB, default", d_in, n_groups, cu * wpc, d_fo);
      run_dma<true, false, 2>("row pairs @ 4 KiB, nt", d_in, n_groups, cu * wpc, d_fo);
      run_dma<false, false, 2, 23040>("contiguous nt + 23 KB stores/group", d_in, n_groups, cu * wpc, d_fo, d_res);
      run_dma<true, false, 2, 23040>("row pairs nt + 23 KB stores/group", d_in, n_groups, cu * wpc, d_fo, d_res);
      run_dma<true, false, 2, 23040, 1>("  stores: default policy", d_in, n_groups, cu * wpc, d_fo, d_res);
      run_dma<true, false, 2, 23040, 2>("  stores: sc0 sc1", d_in, n_groups, cu * wpc, d_fo, d_res);
      run_dma<true, false, 2, 23040, 3>("  stores: sc1", d_in, n_groups, cu * wpc, d_fo, d_res);
      run_dma<true, false, 2, 23040, 4>("  stores: sc0 sc1 nt", d_in, n_groups, cu * wpc, d_fo, d_res);
      run_dma<true, false, 2, 23040>("  stores: nt (again)", d_in, n_groups, cu * wpc, d_fo, d_res);
    }
    if (false) {
      printf("--- F: stores in chip-wide phases (2048 persistent waves, bounded-spin barrier) ---\n");
      unsigned* d_cnt;
      CK(hipMalloc((void**)&d_cnt, 4));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      for (int limit : {0, 20000}) {
        float tot = 0.f;
        const int it = 5;
        for (int k = 0; k < it + 1; ++k) {
          CK(hipMemset(d_cnt, 0, 4));
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL((dma_phased_kernel<23040>), dim3(2048), dim3(64), 0, 0, d_in, n_groups, d_fo, d_res, d_cnt, limit);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (k) tot += ms;
        }
        printf("%-34s %8.3f ms/launch  %8.1f GB/s (reads)\n", limit ? "phased stores (barrier)" : "same kernel, barrier off", tot / it,
               (double)n_groups * 262144 * it / tot * 1e-6);
      }
      CK(hipEventRecord(e0));
      for (int k = 0; k < 5; ++k)
        hipLaunchKernelGGL(write_only_kernel, dim3(8192), dim3(256), 0, 0, d_res, (long long)n_groups * 23040 / 16);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("write-only stream (nt, 1.5 GB)     %8.3f ms/launch  %8.1f GB/s (writes)\n", ms / 5, (double)n_groups * 23040 * 5 / ms * 1e-6);
    }
    return 0;
  }
  printf("--- A/B: f64 FMA fed by scalar loads vs registers (45 accumulators) ---\n");
  for (int wpc : {1, 2, 3, 4}) {  // workgroups of 4 waves per CU -> waves per SIMD
    printf("[%d waves/SIMD]\n", wpc);
    run_sfma<45, 1, false>("register operands  NACC=45 P=1", d_tab, 185, cu * wpc, d_out);
    run_sfma<45, 1, true>("scalar operands    NACC=45 P=1", d_tab, 185, cu * wpc, d_out);     // 65 KB table
    run_sfma<45, 1, true>("scalar operands    NACC=45 P=1", d_tab, 40, cu * wpc, d_out);      // 14 KB table
    run_sfma<45, 1, true>("scalar operands    NACC=45 P=1", d_tab, 2000, cu * wpc, d_out);    // 700 KB
    if (wpc <= 2) {
      run_sfma<45, 2, true>("scalar operands    NACC=45 P=2", d_tab, 185, cu * wpc, d_out);
      run_sfma<91, 1, true>("scalar operands    NACC=91 P=1", d_tab, 185, cu * wpc, d_out);
    }
    run_sfma<16, 1, true>("scalar operands    NACC=16 P=1", d_tab, 740, cu * wpc, d_out);
  }

  printf("--- C: LDS-DMA read patterns (4 GiB of patches, 16 KiB LDS per wave) ---\n");
  const long long n_groups = 16384;  // x 256 KiB = 4 GiB
  float* d_in;
  CK(hipMalloc((void**)&d_in, (size_t)n_groups * 262144));
  CK(hipMemset(d_in, 0, (size_t)n_groups * 262144));
  float* d_fo;
  CK(hipMalloc((void**)&d_fo, (size_t)65536 * 64 * 4));
  for (int wpc : {4, 8, 10}) {
    run_dma<false>("contiguous 16 KiB stages", d_in, n_groups, cu * wpc, d_fo);
    run_dma<true>("row pairs (128 B @ 4 KiB stride)", d_in, n_groups, cu * wpc, d_fo);
    run_dma<true, true>("row pairs, middle 64 B only (GB/s as if whole)", d_in, n_groups, cu * wpc, d_fo);
  }
  return 0;
}
