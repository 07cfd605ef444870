// micro_dpp64.hip -- can the x table of the strip dense kernel live in VGPRs?  Rate and semantics of
//   v_fmac_f64_dpp vdst, vsrc0, vsrc1 row_newbcast:N      (vdst += vsrc0[lane N of my row of 16] * vsrc1)
// on gfx950 against plain v_fmac_f64 with a VGPR and with an SGPR operand; the shader clock of each loop (s_memtime against
// the 100-MHz s_memrealtime); and the round-trip latencies the hand-pipelined sweep has to cover: ds_read_b64, s_load_dwordx16
// (scalar-cache hit).  Measurement aid; not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -o bin/micro_dpp64 micro_dpp64.hip && ./bin/micro_dpp64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define FMAC_DPP(acc, t, x, N) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t), "v"(x))
#define FMAC_V(acc, t, x) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(t), "v"(x))
#define FMAC_S(acc, t, x) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "s"(t), "v"(x))

struct stamp { unsigned long long rt, ck; };
__device__ __forceinline__ stamp now() { return {__builtin_amdgcn_s_memrealtime(), __builtin_amdgcn_s_memtime()}; }

template <int MODE>  // 0: dpp row_newbcast, 1: VGPR operand, 2: SGPR operand
__global__ __launch_bounds__(256, 2) void rate_kernel(double* out, unsigned long long* clk, int iters, double seed) {
  double c[8];
  for (int i = 0; i < 8; ++i) c[i] = i;
  double t = threadIdx.x * 1e-3 + seed, x = 1.0 + threadIdx.x * 1e-4;
  double ts = __builtin_amdgcn_readfirstlane((int)seed) + 0.5;
  const stamp a = now();
#pragma unroll 4
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      FMAC_DPP(c[0], t, x, 0); FMAC_DPP(c[1], t, x, 1); FMAC_DPP(c[2], t, x, 2); FMAC_DPP(c[3], t, x, 3);
      FMAC_DPP(c[4], t, x, 4); FMAC_DPP(c[5], t, x, 5); FMAC_DPP(c[6], t, x, 6); FMAC_DPP(c[7], t, x, 7);
    } else if (MODE == 1) {
      FMAC_V(c[0], t, x); FMAC_V(c[1], t, x); FMAC_V(c[2], t, x); FMAC_V(c[3], t, x);
      FMAC_V(c[4], t, x); FMAC_V(c[5], t, x); FMAC_V(c[6], t, x); FMAC_V(c[7], t, x);
    } else {
      FMAC_S(c[0], ts, x); FMAC_S(c[1], ts, x); FMAC_S(c[2], ts, x); FMAC_S(c[3], ts, x);
      FMAC_S(c[4], ts, x); FMAC_S(c[5], ts, x); FMAC_S(c[6], ts, x); FMAC_S(c[7], ts, x);
    }
  }
  const stamp b = now();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    clk[2 * blockIdx.x] = b.rt - a.rt;
    clk[2 * blockIdx.x + 1] = b.ck - a.ck;
  }
}

// semantics: out[lane][n] = value the DPP operand delivers for row_newbcast:n
__global__ void sem_kernel(double* out) {
  const int l = threadIdx.x;
  double t = 100.0 + l, one = 1.0;
#define ONE(N) { double acc = 0.0; FMAC_DPP(acc, t, one, N); out[l * 16 + N] = acc; }
  ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7) ONE(8) ONE(9) ONE(10) ONE(11) ONE(12) ONE(13) ONE(14) ONE(15)
}

// latencies (one wave, dependent chains)
__global__ void lat_kernel(const double* __restrict__ tab, unsigned long long* res, int iters) {
  __shared__ double lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (double)((i * 8 + 8) % 8192);   // a pointer chain in bytes
  __syncthreads();
  // LDS: dependent ds_read_b64 chain
  unsigned addr = threadIdx.x * 8;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  double v = 0;
  for (int i = 0; i < iters; ++i) {
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    addr = (unsigned)v;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  // SMEM: dependent s_load_dwordx16 chain (table of 1 KiB: always a scalar-cache hit after the first pass)
  typedef unsigned v16u __attribute__((ext_vector_type(16)));
  const __attribute__((address_space(4))) double* p = (const __attribute__((address_space(4))) double*)tab;
  double accs = 0;
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  unsigned off = __builtin_amdgcn_readfirstlane(0);
  for (int i = 0; i < iters; ++i) {
    v16u r;
    asm volatile("s_load_dwordx16 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p), "s"(off));
    off = r[0] & 0x3c0;   // next offset depends on the loaded data
    accs += (double)r[1];
  }
  unsigned long long t3 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    res[0] = t1 - t0;
    res[1] = t3 - t2;
    res[2] = (unsigned long long)(accs + v);
  }
}

int main() {
  double* d;
  unsigned long long* clk;
  hipMalloc(&d, 1 << 24);
  hipMalloc(&clk, 1 << 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 40000, blocks = 256 * 2, threads = 256;   // 2 workgroups of 4 waves per CU: two waves per SIMD
  hipFuncSetAttribute((const void*)rate_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  hipFuncSetAttribute((const void*)rate_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  hipFuncSetAttribute((const void*)rate_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  const char* names[3] = {"v_fmac_f64_dpp row_newbcast", "v_fmac_f64 (VGPR operand)", "v_fmac_f64 (SGPR operand)"};
  for (int pass = 0; pass < 2; ++pass)
    for (int mode = 0; mode < 3; ++mode) {
      hipEventRecord(e0);
      // 72 KiB of (unused) dynamic LDS per workgroup: exactly two workgroups = two waves per SIMD on every CU
      if (mode == 0) rate_kernel<0><<<blocks, threads, 72 * 1024>>>(d, clk, iters, 1.0);
      if (mode == 1) rate_kernel<1><<<blocks, threads, 72 * 1024>>>(d, clk, iters, 1.0);
      if (mode == 2) rate_kernel<2><<<blocks, threads, 72 * 1024>>>(d, clk, iters, 1.0);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(2048);
      hipMemcpy(h.data(), clk, 2048 * 8, hipMemcpyDeviceToHost);
      double rt = 0, ck = 0;
      for (int b = 0; b < blocks; ++b) { rt += h[2 * b]; ck += h[2 * b + 1]; }
      const double flop = 2.0 * 64 * 8.0 * iters * (blocks * threads / 64);
      printf("%-30s %.2f ms  %.1f TFLOP/s  shader clock %.3f GHz  clocks per wave instruction (2 waves per SIMD) %.2f\n", names[mode], ms,
             flop / ms / 1e9, ck / (rt * 10.0), (ck / blocks) / (8.0 * iters) );
    }
  double* sem;
  hipMalloc(&sem, 64 * 16 * 8);
  sem_kernel<<<1, 64>>>(sem);
  std::vector<double> hs(64 * 16);
  hipMemcpy(hs.data(), sem, hs.size() * 8, hipMemcpyDeviceToHost);
  bool ok = true;
  for (int l = 0; l < 64; ++l)
    for (int n = 0; n < 16; ++n) ok &= hs[l * 16 + n] == 100.0 + (l / 16) * 16 + n;
  printf("row_newbcast:N delivers lane N of the lane's own row of 16: %s (lane 37, N 5 -> %.0f)\n", ok ? "yes" : "NO", hs[37 * 16 + 5]);
  std::vector<unsigned> tab(256);
  for (int i = 0; i < 256; ++i) tab[i] = (unsigned)(((i / 16 + 1) % 16) * 64);
  double* dt;
  hipMalloc(&dt, 1024);
  hipMemcpy(dt, tab.data(), 1024, hipMemcpyHostToDevice);
  unsigned long long* res;
  hipMalloc(&res, 64);
  for (int pass = 0; pass < 2; ++pass) {
    lat_kernel<<<1, 64>>>(dt, res, 2000);
    unsigned long long hr[3];
    hipMemcpy(hr, res, 24, hipMemcpyDeviceToHost);
    printf("dependent round trips (one wave, idle chip): ds_read_b64 %.0f clocks, s_load_dwordx16 (cache hit) %.0f clocks\n", hr[0] / 2000.0,
           hr[1] / 2000.0);
  }
  return 0;
}
