// Development probe: how many workgroups of the owner kernel's shape (256 threads, 97 KB of LDS: one per CU) are RESIDENT at once, and on
// which CUs?  Every workgroup records HW_REG_HW_ID / XCC_ID and its start time, then spins 1 ms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void k(unsigned long long* out, long long ticks) {
  extern __shared__ unsigned lds[];
  lds[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = wall_clock64();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = ((unsigned long long)xcc << 32) | hw; if (lds[1] == 77777u) out[0] = 0; }
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s, multiProcessorCount %d\n", p.name, p.multiProcessorCount);
  unsigned long long* out; CK(hipMalloc(&out, 4096 * 16));
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int grid : {224, 240, 256, 288}) {
    CK(hipMemset(out, 0, 4096 * 16));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 97 * 1024, 0, out, 100000LL);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(2 * grid); CK(hipMemcpy(h.data(), out, 16 * grid, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull; for (int i = 0; i < grid; i++) if (h[2 * i] < tmin) tmin = h[2 * i];
    int early = 0; std::set<unsigned long long> cus; int perx[8] = {0};
    for (int i = 0; i < grid; i++) {
      if (h[2 * i] - tmin < 10000) early++;
      const unsigned hw = (unsigned)h[2 * i + 1], xcc = (unsigned)(h[2 * i + 1] >> 32) & 7;
      const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      cus.insert(((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu);
      if (h[2 * i] - tmin < 10000) perx[xcc]++;
    }
    printf("grid %d: %d workgroups started within 100 us of the first; distinct (xcc, se, sh, cu) = %zu; early per xcc:", grid, early, cus.size());
    for (int x = 0; x < 8; x++) printf(" %d", perx[x]);
    printf("\n");
  }
  return 0;
}
