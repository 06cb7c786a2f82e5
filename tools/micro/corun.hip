// Development probe: which kernels does the gfx950 dispatcher place beside a resident workgroup of the step engine's shape?
// A = one workgroup per CU (NW waves of 64, 256 VGPRs each, LDSB bytes of LDS) spinning for ~2 ms; B = small waves on a second stream,
// launched ~300 us later (host sleep).  Prints when B finished relative to A's start/end: "beside" if B ended long before A did.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NW, int VG>
__global__ __launch_bounds__(NW * 64) void spin_kernel(long long cycles, unsigned* sink, int ldsb) {
  extern __shared__ unsigned lds[];
  if (ldsb > 0) lds[threadIdx.x] = threadIdx.x;
  if (VG == 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
  if (VG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) { __builtin_amdgcn_s_sleep(32); }
  if (sink && threadIdx.x == 0 && ldsb > 0) sink[blockIdx.x] = lds[0];
}
template <int VG>
__global__ __launch_bounds__(64) void small_kernel(unsigned* out, int iters, int ldsb) {
  extern __shared__ unsigned lds[];
  if (ldsb > 0) lds[threadIdx.x] = 1;
  if (VG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (VG == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
  unsigned x = threadIdx.x + blockIdx.x;
  for (int i = 0; i < iters; i++) x = x * 1664525u + 1013904223u;
  out[blockIdx.x * 64 + threadIdx.x] = x;
}

template <int NT>
__global__ __launch_bounds__(NT) void wide_kernel(unsigned* out, int iters) {
  unsigned x = threadIdx.x + blockIdx.x;
  for (int i = 0; i < iters; i++) x = x * 1664525u + 1013904223u;
  out[blockIdx.x * NT + threadIdx.x] = x;
}
template <int NW, int VG> void launch_spin(int grid, int ldsb, hipStream_t s, long long cyc, unsigned* sink) {
  CK(hipFuncSetAttribute((const void*)spin_kernel<NW, VG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((spin_kernel<NW, VG>), dim3(grid), dim3(NW * 64), ldsb, s, cyc, sink, ldsb);
}

int main() {
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  unsigned *sink, *out;
  CK(hipMalloc(&sink, 4096 * 4)); CK(hipMalloc(&out, 4096 * 64 * 4));
  hipEvent_t a0, a1, b0, b1;
  CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
  const long long cyc = 200000; // wall_clock64 runs at 100 MHz: 2 ms
  struct Case { const char* name; int nw, vg, ldsb, bvg, bldsb, bgrid; };
  std::vector<Case> cases = {
    {"A 8w 256v 159K | B 24v", 8, 256, 159 * 1024, 0, 0, 1024},
    {"A 7w 256v 159K | B 24v", 7, 256, 159 * 1024, 0, 0, 1024},
    {"A 7w 256v 155K | B 24v", 7, 256, 155 * 1024, 0, 0, 1024},
    {"A 7w 256v 0K   | B 24v", 7, 256, 0, 0, 0, 1024},
    {"A 7w 256v 64K  | B 24v", 7, 256, 64 * 1024, 0, 0, 1024},
    {"A 6w 256v 155K | B 24v", 6, 256, 155 * 1024, 0, 0, 1024},
    {"A 4w 256v 155K | B 24v", 4, 256, 155 * 1024, 0, 0, 1024},
    {"A 8w 128v 155K | B 24v", 8, 128, 155 * 1024, 0, 0, 1024},
    {"A 7w 256v 155K | B 64v", 7, 256, 155 * 1024, 64, 0, 1024},
    {"A 7w 256v 155K | B 128v", 7, 256, 155 * 1024, 128, 0, 1024},
    {"A 7w 256v 155K | B 24v 3K lds", 7, 256, 155 * 1024, 0, 3328, 1024},
    {"A 7w 256v 155K | B memset 16B", 7, 256, 155 * 1024, -1, 0, 1024},
    {"A 7w 256v 155K | B 256-thread WGs", 7, 256, 155 * 1024, -2, 0, 256},
    {"A 7w 256v 155K | B 128-thread WGs", 7, 256, 155 * 1024, -3, 0, 512},
  };
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; rep++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a0, s1));
#define SP(NWV, VGV) if (c.nw == NWV && c.vg == VGV) launch_spin<NWV, VGV>(256, c.ldsb, s1, cyc, sink)
      SP(8, 256); SP(7, 256); SP(6, 256); SP(4, 256); SP(8, 128);
      CK(hipEventRecord(a1, s1));
      // B: 200 us later
      CK(hipStreamQuery(s1) == hipErrorNotReady ? hipSuccess : hipSuccess); usleep(300);
      CK(hipEventRecord(b0, s2));
      if (c.bvg == 0) hipLaunchKernelGGL(small_kernel<0>, dim3(c.bgrid), dim3(64), c.bldsb, s2, out, 2000, c.bldsb);
      if (c.bvg == -1) CK(hipMemsetAsync(out, 0, 16, s2));
      if (c.bvg == -2) hipLaunchKernelGGL(wide_kernel<256>, dim3(c.bgrid), dim3(256), 0, s2, out, 2000);
      if (c.bvg == -3) hipLaunchKernelGGL(wide_kernel<128>, dim3(c.bgrid), dim3(128), 0, s2, out, 2000);
      if (c.bvg == 64) hipLaunchKernelGGL(small_kernel<64>, dim3(c.bgrid), dim3(64), c.bldsb, s2, out, 2000, c.bldsb);
      if (c.bvg == 128) hipLaunchKernelGGL(small_kernel<128>, dim3(c.bgrid), dim3(64), c.bldsb, s2, out, 2000, c.bldsb);
      CK(hipEventRecord(b1, s2));
      CK(hipDeviceSynchronize());
      float ta, tb0, tb1;
      CK(hipEventElapsedTime(&ta, a0, a1)); CK(hipEventElapsedTime(&tb0, a0, b0)); CK(hipEventElapsedTime(&tb1, a0, b1));
      if (rep) printf("%-34s A %.2f ms   B start %.2f end %.2f ms  -> %s\n", c.name, ta, tb0, tb1, tb1 < ta - 0.3f ? "BESIDE" : "after");
    }
  }
  return 0;
}
