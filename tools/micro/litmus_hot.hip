// Litmus test for the step engine's cross-wave hand-over (bg_engine.h): wave A of a workgroup stores an env's state to GLOBAL memory and
// then publishes the env through an LDS word; wave B of the SAME workgroup sees the LDS word and loads that state with plain loads.
// The engine relies on B seeing A's data without a vmcnt(0) drain in A and without cache-bypassing loads in B: both waves' vector-memory
// operations go through the one texture-address / L1 pipeline of their CU in issue order, and A's store instruction has issued before
// A's LDS write executes.  The hard case is a STALE L1 LINE: B reads the line first (so it sits in the CU's vector L1), then A overwrites
// it.  This program runs exactly that, many times, with the engine's own idiom (relaxed LDS atomics + compiler barriers, 16-byte stores /
// loads, other waves hammering the memory pipeline with non-temporal stores) and counts stale reads.  Prints "stale 0 of N".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)

__device__ __forceinline__ unsigned lds_ld(unsigned* p) { unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("" ::: "memory"); return v; }
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) { asm volatile("" ::: "memory"); __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// 256 threads = 4 waves: wave 0 = producer A, wave 1 = consumer B, waves 2-3 = noise (streaming non-temporal stores)
__global__ __launch_bounds__(256) void litmus(uint4* state, uint4* noise, size_t noise_n, int rounds, unsigned long long* stale, unsigned long long* trials) {
  __shared__ unsigned flag, ack;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint4* mine = state + (size_t)blockIdx.x * 64 * 8;   // 8 chunks of 16 bytes per lane, chunk-major like d.hot: [k][lane]
  if (threadIdx.x == 0) { flag = 0; ack = 0; }
  __syncthreads();
  unsigned long long bad = 0, n = 0;
  if (wave == 0) {
    for (int r = 1; r <= rounds; r++) {
      while (lds_ld(&ack) != (unsigned)(r - 1)) __builtin_amdgcn_s_sleep(1);       // B has read round r-1 (and now holds the lines in L1)
#pragma unroll
      for (int k = 0; k < 8; k++) mine[k * 64 + lane] = make_uint4((unsigned)r, (unsigned)k, (unsigned)lane, (unsigned)r ^ 0x5a5a5a5au);
      lds_st(&flag, (unsigned)r);                                                  // "data, then flag": no vmcnt drain, as in the engine
    }
  } else if (wave == 1) {
    for (int r = 1; r <= rounds; r++) {
      // pull the OLD lines into this CU's L1 first (the stale-line case), then wait for the hand-over
      uint4 old[8];
#pragma unroll
      for (int k = 0; k < 8; k++) old[k] = mine[k * 64 + lane];
      asm volatile("" :: "v"(old[0].x), "v"(old[7].x));
      while (lds_ld(&flag) != (unsigned)r) __builtin_amdgcn_s_sleep(1);
      uint4 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = mine[k * 64 + lane];
#pragma unroll
      for (int k = 0; k < 8; k++) { n++; if (v[k].x != (unsigned)r || v[k].w != ((unsigned)r ^ 0x5a5a5a5au) || v[k].y != (unsigned)k) bad++; }
      lds_st(&ack, (unsigned)r);
    }
    for (int off = 32; off > 0; off >>= 1) { bad += __shfl_down(bad, off); n += __shfl_down(n, off); }
    if (lane == 0) { atomicAdd(stale, bad); atomicAdd(trials, n); }
  } else {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4* dst = (u32x4*)noise;
    size_t i = ((size_t)blockIdx.x * 128 + (threadIdx.x - 128)) % noise_n;
    for (int r = 0; r < rounds * 8; r++) { u32x4 v = {(unsigned)r, 1u, 2u, 3u}; __builtin_nontemporal_store(v, &dst[i]); i = (i + 40009) % noise_n; if (lds_ld(&ack) >= (unsigned)rounds) break; }
  }
}
int main(int argc, char** argv) {
  const int blocks = 256 * 4, rounds = argc > 1 ? atoi(argv[1]) : 2000;
  uint4 *state, *noise; unsigned long long *cnt;
  const size_t noise_n = (1ull << 30) / 16;
  CK(hipMalloc(&state, (size_t)blocks * 64 * 8 * 16)); CK(hipMemset(state, 0, (size_t)blocks * 64 * 8 * 16));
  CK(hipMalloc(&noise, noise_n * 16)); CK(hipMalloc(&cnt, 16)); CK(hipMemset(cnt, 0, 16));
  hipLaunchKernelGGL(litmus, dim3(blocks), dim3(256), 0, 0, state, noise, noise_n, rounds, cnt, cnt + 1);
  CK(hipDeviceSynchronize());
  unsigned long long h[2]; CK(hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost));
  printf("stale %llu of %llu\n", h[0], h[1]);
  return h[0] ? 1 : 0;
}
