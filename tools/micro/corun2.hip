// Development probe for the two-kernel engine (bg_engine2.h): does the gfx950 dispatcher place one-wave workgroups of the service
// kernel's shape (VB VGPRs, LB bytes of LDS, optional scratch) on a CU that already holds a four-wave workgroup of the owner kernel's shape
// (VA VGPRs per wave, LA bytes of LDS)?  A = one workgroup per CU spinning for 2 ms; B = 1024 one-wave workgroups on a second stream, started
// 300 us later.  Reports how many B waves STARTED before A ended.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VG>
__device__ __forceinline__ void touch_vgpr() {
  if (VG == 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
  if (VG == 224) asm volatile("v_mov_b32 v223, 0" ::: "v223");
  if (VG == 192) asm volatile("v_mov_b32 v191, 0" ::: "v191");
  if (VG == 152) asm volatile("v_mov_b32 v151, 0" ::: "v151");
  if (VG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (VG == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  if (VG == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
}
template <int NW, int VG>
__global__ __launch_bounds__(NW * 64) void a_kernel(long long ticks, unsigned long long* a_end, int ldsb) {
  extern __shared__ unsigned lds[];
  if (ldsb > 0) lds[threadIdx.x] = threadIdx.x;
  touch_vgpr<VG>();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0) { a_end[blockIdx.x] = wall_clock64(); if (ldsb > 0 && lds[1] == 12345u) a_end[0] = 0; }
}
template <int VG, bool SCRATCH>
__global__ __launch_bounds__(64) void b_kernel(unsigned long long* b_start, int ldsb, int iters) {
  extern __shared__ unsigned lds[];
  if (ldsb > 0) lds[threadIdx.x] = 1;
  touch_vgpr<VG>();
  if (threadIdx.x == 0) b_start[blockIdx.x] = wall_clock64();
  unsigned x = threadIdx.x + blockIdx.x;
  if (SCRATCH) { volatile unsigned arr[16]; for (int i = 0; i < 16; i++) arr[i] = x + i; for (int i = 0; i < iters; i++) x += arr[(x >> 3) & 15]; }
  for (int i = 0; i < iters; i++) x = x * 1664525u + 1013904223u;
  if (x == 77u) b_start[0] = 0;
}

int main() {
  hipStream_t s1, s2;
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi));
  unsigned long long *a_end, *b_start;
  CK(hipMalloc(&a_end, 256 * 8)); CK(hipMalloc(&b_start, 1024 * 8));
  struct Case { const char* name; int va, la, vb, lb, scratch; };
  std::vector<Case> cases = {
    {"A 4w 152v  97K | B 256v 13.4K scratch", 152, 97 * 1024, 256, 13376, 1},
    {"A 4w 152v  97K | B 256v 13.4K", 152, 97 * 1024, 256, 13376, 0},
    {"A 4w 152v  97K | B 256v 0K", 152, 97 * 1024, 256, 0, 0},
    {"A 4w 152v  97K | B 224v 13.4K", 152, 97 * 1024, 224, 13376, 0},
    {"A 4w 152v  97K | B 192v 13.4K", 152, 97 * 1024, 192, 13376, 0},
    {"A 4w 152v  97K | B 128v 13.4K", 152, 97 * 1024, 128, 13376, 0},
    {"A 4w 152v  97K | B 64v  13.4K", 152, 97 * 1024, 64, 13376, 0},
    {"A 4w 128v  97K | B 256v 13.4K", 128, 97 * 1024, 256, 13376, 0},
    {"A 4w 96v   97K | B 256v 13.4K", 96, 97 * 1024, 256, 13376, 0},
    {"A 4w 64v   97K | B 256v 13.4K", 64, 97 * 1024, 256, 13376, 0},
    {"A 4w 152v  64K | B 256v 13.4K", 152, 64 * 1024, 256, 13376, 0},
    {"A 4w 152v   0K | B 256v 13.4K", 152, 0, 256, 13376, 0},
    {"A 4w 64v    0K | B 256v 0K", 64, 0, 256, 0, 0},
    {"A 4w 64v   97K | B 128v 13.4K", 64, 97 * 1024, 128, 13376, 0},
  };
  const long long ticks = 200000; // 2 ms
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; rep++) {
      CK(hipMemset(a_end, 0, 256 * 8)); CK(hipMemset(b_start, 0, 1024 * 8));
      CK(hipDeviceSynchronize());
#define LA(VAV) if (c.va == VAV) { CK(hipFuncSetAttribute((const void*)a_kernel<4, VAV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((a_kernel<4, VAV>), dim3(256), dim3(256), c.la, s1, ticks, a_end, c.la); }
      LA(152) LA(128) LA(96) LA(64)
      usleep(300);
#define LB(VBV) if (c.vb == VBV) { if (c.scratch) hipLaunchKernelGGL((b_kernel<VBV, true>), dim3(1024), dim3(64), c.lb, s2, b_start, c.lb, 2000); \
        else hipLaunchKernelGGL((b_kernel<VBV, false>), dim3(1024), dim3(64), c.lb, s2, b_start, c.lb, 2000); }
      LB(256) LB(224) LB(192) LB(128) LB(64)
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> ae(256), bs(1024);
      CK(hipMemcpy(ae.data(), a_end, 256 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(bs.data(), b_start, 1024 * 8, hipMemcpyDeviceToHost));
      unsigned long long amin = ~0ull; for (auto v : ae) if (v && v < amin) amin = v;
      int beside = 0; for (auto v : bs) if (v && v + 1000 < amin) beside++;   // started > 10 us before the first A workgroup ended
      if (rep) printf("%-42s B waves started beside A: %4d of 1024\n", c.name, beside);
    }
  }
  return 0;
}
