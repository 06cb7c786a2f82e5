// Development probe: does an ALU-bound kernel on one SIMD per CU lower the shader clock the step engine runs at?
// A = engine-shaped workgroups (7 waves, 256 VGPRs, 155 KB LDS) that measure shader cycles (clock64) per 100 MHz wall-clock tick while
// doing light work; B = dependent v_mul_lo chains, 8 one-wave workgroups per CU, beside A.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(448) void a_kernel(long long wall_ticks, double* mhz, unsigned* sink, int work) {
  extern __shared__ unsigned lds[];
  lds[threadIdx.x] = threadIdx.x;
  asm volatile("v_mov_b32 v255, 0" ::: "v255");
  const long long w0 = wall_clock64(), c0 = clock64();
  unsigned x = threadIdx.x;
  while (wall_clock64() - w0 < wall_ticks) {
    if (work) { for (int i = 0; i < 64; i++) { x = lds[(x + i) & 255] + x * 3u; } }
    else __builtin_amdgcn_s_sleep(16);
  }
  const long long w1 = wall_clock64(), c1 = clock64();
  if (threadIdx.x == 0) { mhz[blockIdx.x] = 100.0 * (double)(c1 - c0) / (double)(w1 - w0); sink[blockIdx.x] = x; }
}
__global__ __launch_bounds__(64) void b_kernel(unsigned* out, int iters) {
  unsigned x = threadIdx.x + blockIdx.x, y = x ^ 12345u;
  for (int i = 0; i < iters; i++) { x = (x ^ (x >> 30)) * 1664525u + 7u; y = (y ^ (y >> 30)) * 1566083941u + x; }
  out[blockIdx.x * 64 + threadIdx.x] = x + y;
}
int main() {
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  double* mhz; unsigned *sink, *out;
  CK(hipMalloc(&mhz, 256 * 8)); CK(hipMalloc(&sink, 256 * 4)); CK(hipMalloc(&out, 65536 * 64 * 4));
  CK(hipFuncSetAttribute((const void*)a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  double h[256];
  for (int work = 0; work < 2; work++) for (int withb = 0; withb < 2; withb++) for (int rep = 0; rep < 3; rep++) {
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(a_kernel, dim3(256), dim3(448), 155 * 1024, s1, 400000LL, mhz, sink, work); // 4 ms
    usleep(200);
    hipEvent_t b0, b1; CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    CK(hipEventRecord(b0, s2));
    if (withb) hipLaunchKernelGGL(b_kernel, dim3(2048 * 8), dim3(64), 0, s2, out, 20000);
    CK(hipEventRecord(b1, s2));
    CK(hipDeviceSynchronize());
    float tb; CK(hipEventElapsedTime(&tb, b0, b1));
    CK(hipMemcpy(h, mhz, sizeof(h), hipMemcpyDeviceToHost));
    double s = 0, mn = 1e9, mx = 0; for (int i = 0; i < 256; i++) { s += h[i]; mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; }
    printf("A %s, B %s: shader clock seen by A %.0f MHz (min %.0f max %.0f), B took %.2f ms\n", work ? "LDS work" : "sleeping", withb ? "beside" : "absent", s / 256, mn, mx, tb);
  }
  return 0;
}
