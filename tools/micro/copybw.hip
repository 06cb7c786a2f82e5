// Development probe: what shape of a plain streaming copy / fill reaches the HBM bandwidth the guide quotes (6.29 TB/s, float4 copy)?
// Sweeps workgroups per CU, pieces per lane, grid-stride vs one-shot, non-temporal loads / stores, buffer size.  The winner becomes
// bg_stream_copy_kernel / bg_stream_fill_kernel (bench.py's `peak_measured`).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n16; i += stride) {
    u32x4 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) if (i + (size_t)k * 256 < n16) v[k] = NTL ? __builtin_nontemporal_load(&src[i + (size_t)k * 256]) : src[i + (size_t)k * 256];
#pragma unroll
    for (int k = 0; k < U; k++) if (i + (size_t)k * 256 < n16) { if (NTS) __builtin_nontemporal_store(v[k], &dst[i + (size_t)k * 256]); else dst[i + (size_t)k * 256] = v[k]; }
  }
}
template <int U, bool NTS>
__global__ __launch_bounds__(256) void fill_k(u32x4* __restrict__ dst, size_t n16, unsigned seed) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n16; i += stride) {
#pragma unroll
    for (int k = 0; k < U; k++) if (i + (size_t)k * 256 < n16) {
      u32x4 v = {seed, (unsigned)i, (unsigned)k, seed ^ (unsigned)i};
      if (NTS) __builtin_nontemporal_store(v, &dst[i + (size_t)k * 256]); else dst[i + (size_t)k * 256] = v;
    }
  }
}
template <typename F> static double time_it(F launch, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); launch();
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; i++) launch();
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e-3 / iters;
}
int main() {
  for (size_t gib4 : {2, 4, 8, 16}) {   // quarter-GiB units: 0.5, 1, 2, 4 GiB per buffer
    const size_t bytes = gib4 << 28, n16 = bytes / 16;
    u32x4 *src, *dst; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes));
    CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 2, bytes));
    printf("== %.2f GiB per buffer\n", bytes / 1073741824.0);
    for (int wgpc : {4, 8, 16, 32, 0}) {
#define RUN(U, NTL, NTS) { const unsigned grid = wgpc ? 256u * wgpc : (unsigned)((n16 + 256 * U - 1) / (256 * U)); \
      double s = time_it([&] { hipLaunchKernelGGL((copy_k<U, NTL, NTS>), dim3(grid), dim3(256), 0, 0, src, dst, n16); }, 10); \
      printf("copy wg/CU %2d U %d ntl %d nts %d : %7.1f GB/s (read + written)\n", wgpc, U, NTL, NTS, 2.0 * bytes / s / 1e9); }
      RUN(1, false, false) RUN(2, false, false) RUN(4, false, false) RUN(8, false, false)
      RUN(4, true, false) RUN(4, false, true) RUN(4, true, true) RUN(2, true, true) RUN(8, true, true)
#define RUNF(U, NTS) { const unsigned grid = wgpc ? 256u * wgpc : (unsigned)((n16 + 256 * U - 1) / (256 * U)); \
      double s = time_it([&] { hipLaunchKernelGGL((fill_k<U, NTS>), dim3(grid), dim3(256), 0, 0, dst, n16, 7u); }, 10); \
      printf("fill wg/CU %2d U %d nts %d : %7.1f GB/s (written)\n", wgpc, U, NTS, (double)bytes / s / 1e9); }
      RUNF(4, false) RUNF(4, true) RUNF(1, false) RUNF(1, true)
    }
    { double s = time_it([&] { CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0)); }, 10);
      printf("hipMemcpyAsync D2D : %7.1f GB/s (read + written)\n", 2.0 * bytes / s / 1e9); }
    CK(hipFree(src)); CK(hipFree(dst));
  }
  return 0;
}
