// Development probe for DESIGN.md section 8 (service batches formed across the chip): what does a request / response hand-over between two
// CUs cost when it goes through global memory?  Workgroup 2k ("owner") posts a request -- a 128-byte payload, then a flag (device-scope release) --
// to workgroup 2k+1 ("server"), which polls the flag (device-scope acquire, s_sleep between polls), reads the payload, writes a 384-byte answer
// (the size of an observation record) and raises the answer flag; the owner polls that, reads the answer and starts over.  Partners are chosen
// on the same XCD (workgroup ids are dealt to the 8 XCDs round robin: ids 8 apart share one) or on different XCDs.  Optionally every third
// workgroup streams non-temporal stores meanwhile (the engine's record stream).
// Prints the round trip in shader cycles and in microseconds.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


// MODE 1: no fences -- every access of the hand-over itself carries sc0 sc1 (written through / fetched past the caches), everything else stays cached
__device__ __forceinline__ void st_sys(u32x4* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u32x4 ld_sys(const u32x4* p) { u32x4 v; asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void st_sys32(unsigned* p, unsigned v) { asm volatile("s_waitcnt vmcnt(0)\n\tglobal_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned ld_sys32(const unsigned* p) { unsigned v; asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }

struct Slot { unsigned req_flag; unsigned pad0[31]; unsigned ans_flag; unsigned pad1[31]; u32x4 req[8]; u32x4 ans[24]; };

template <int MODE>
__global__ __launch_bounds__(64) void k(Slot* slots, int pairs, int partner_stride, int rounds, unsigned* lat, unsigned char* wbuf, size_t wrows, int writers,
                                        volatile unsigned* stop) {
  const int lane = threadIdx.x, wg = blockIdx.x;
  const int npp = 2 * pairs; // workgroups that play; the rest write
  if (wg >= npp) {
    if (!writers) return;
    for (unsigned b = 0;; b++) {
      if ((b & 15) == 0 && *stop >= (unsigned)pairs) break;
#pragma unroll
      for (int r = 0; r < 12; r++) {
        const int q = r * 64 + lane, j = q / 24, pc = q - 24 * j;
        const size_t row = ((size_t)wg * 7919 + (size_t)b * 104729ull + j * 31) % wrows;
        u32x4 v = {(unsigned)q, b, (unsigned)wg, (unsigned)row};
        __builtin_nontemporal_store(v, (u32x4*)(wbuf + row * 384 + 16 * pc));
      }
      __builtin_amdgcn_s_sleep(64);
    }
    return;
  }
  // pair p: owner = workgroup id o, server = o + partner_stride (ids chosen so that every workgroup < npp is in exactly one pair)
  const int grp = wg / (2 * partner_stride), pos = wg % (2 * partner_stride);
  const bool owner = pos < partner_stride;
  const int p = grp * partner_stride + (owner ? pos : pos - partner_stride);
  Slot* s = &slots[p];
  if (owner) {
    unsigned acc = 0;
    for (int i = 1; i <= rounds; i++) {
      const unsigned long long t0 = __builtin_readcyclecounter();
      unsigned f = 0, spins = 0;
      if (MODE == 0) {
        if (lane < 8) { u32x4 v = {(unsigned)i, (unsigned)lane, acc, 7u}; s->req[lane] = v; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_store(&s->req_flag, (unsigned)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        do { if (spins++) __builtin_amdgcn_s_sleep(4); f = __hip_atomic_load(&s->ans_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); f = __builtin_amdgcn_readfirstlane(f); } while (f != (unsigned)i && spins < (1u << 22));
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (lane < 24) { u32x4 v = __builtin_nontemporal_load(&s->ans[lane]); acc += v.x; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        if (lane < 8) { u32x4 v = {(unsigned)i, (unsigned)lane, acc, 7u}; st_sys(&s->req[lane], v); }
        if (lane == 0) st_sys32(&s->req_flag, (unsigned)i);   // (waits for this wave's payload stores first)
        do { if (spins++) __builtin_amdgcn_s_sleep(4); f = __builtin_amdgcn_readfirstlane(ld_sys32(&s->ans_flag)); } while (f != (unsigned)i && spins < (1u << 22));
        if (lane < 24) { u32x4 v = ld_sys(&s->ans[lane]); acc += v.x; if (v.x != (unsigned)i) atomicAdd((unsigned*)stop + 1, 1u); } // a stale answer
      }
      const unsigned long long t1 = __builtin_readcyclecounter();
      if (lane == 0) lat[(size_t)p * rounds + i - 1] = (unsigned)(t1 - t0);
    }
    if (acc == 12345u) lat[0] = 0;
    if (lane == 0) atomicAdd((unsigned*)stop, 1u);
  } else {
    unsigned acc = 0;
    for (int i = 1; i <= rounds; i++) {
      unsigned f = 0, spins = 0;
      if (MODE == 0) {
        do { if (spins++) __builtin_amdgcn_s_sleep(4); f = __hip_atomic_load(&s->req_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); f = __builtin_amdgcn_readfirstlane(f); } while (f != (unsigned)i && spins < (1u << 22));
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (lane < 8) { u32x4 v = __builtin_nontemporal_load(&s->req[lane]); acc += v.x; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane < 24) { u32x4 v = {(unsigned)i, (unsigned)lane, acc, 9u}; s->ans[lane] = v; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_store(&s->ans_flag, (unsigned)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        do { if (spins++) __builtin_amdgcn_s_sleep(4); f = __builtin_amdgcn_readfirstlane(ld_sys32(&s->req_flag)); } while (f != (unsigned)i && spins < (1u << 22));
        if (lane < 8) { u32x4 v = ld_sys(&s->req[lane]); acc += v.x; if (v.x != (unsigned)i) atomicAdd((unsigned*)stop + 1, 1u); } // a stale request
        if (lane < 24) { u32x4 v = {(unsigned)i, (unsigned)lane, acc, 9u}; st_sys(&s->ans[lane], v); }
        if (lane == 0) st_sys32(&s->ans_flag, (unsigned)i);
      }
    }
  }
}

template <int MODE>
static void run(const char* name, int pairs, int partner_stride, int writers, unsigned char* wbuf, size_t wrows) {
  const int rounds = 300, grid = 256;
  Slot* slots; unsigned* lat; unsigned* stop;
  CK(hipMalloc(&slots, sizeof(Slot) * pairs)); CK(hipMemset(slots, 0, sizeof(Slot) * pairs));
  CK(hipMalloc(&lat, sizeof(unsigned) * pairs * rounds)); CK(hipMalloc(&stop, 8)); CK(hipMemset(stop, 0, 8));
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, slots, pairs, partner_stride, rounds, lat, wbuf, wrows, writers, stop);
  CK(hipDeviceSynchronize());
  std::vector<unsigned> h((size_t)pairs * rounds);
  CK(hipMemcpy(h.data(), lat, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost));
  std::vector<unsigned> s; for (int p = 0; p < pairs; p++) for (int i = rounds / 4; i < rounds; i++) s.push_back(h[(size_t)p * rounds + i]);
  std::sort(s.begin(), s.end());
  double mean = 0; for (unsigned v : s) mean += v; mean /= s.size();
  unsigned hs[2]; CK(hipMemcpy(hs, stop, 8, hipMemcpyDeviceToHost));
  printf("%s%-58s pairs %3d (stale reads %u): round trip mean %6.0f  p50 %6u  p95 %6u cycles  (%.2f us at 2.4 GHz)\n", MODE ? "[sc0 sc1, no fences] " : "[agent-scope fences]  ", name, pairs, hs[1], mean, s[s.size() / 2], s[s.size() * 95 / 100], mean / 2400.0);
  CK(hipFree(slots)); CK(hipFree(lat)); CK(hipFree(stop));
}

int main() {
  const size_t wbytes = 4ull << 30;
  unsigned char* wbuf; CK(hipMalloc(&wbuf, wbytes)); CK(hipMemset(wbuf, 0, wbytes));
  const size_t wrows = wbytes / 384;
  run<0>("partners 1 workgroup id apart (neighbouring XCDs), idle chip", 8, 1, 0, wbuf, wrows);
  run<0>("partners 8 ids apart (same XCD), idle chip", 8, 8, 0, wbuf, wrows);
  run<0>("partners 4 ids apart (other XCD), idle chip", 8, 4, 0, wbuf, wrows);
  run<0>("partners 8 ids apart (same XCD), 64 pairs", 64, 8, 0, wbuf, wrows);
  run<0>("partners 1 id apart (other XCD), 64 pairs", 64, 1, 0, wbuf, wrows);
  run<0>("partners 8 ids apart (same XCD), 64 pairs + 128 writers", 64, 8, 1, wbuf, wrows);
  run<0>("partners 1 id apart (other XCD), 64 pairs + 128 writers", 64, 1, 1, wbuf, wrows);
  run<1>("partners 1 workgroup id apart (neighbouring XCDs), idle chip", 8, 1, 0, wbuf, wrows);
  run<1>("partners 8 ids apart (same XCD), idle chip", 8, 8, 0, wbuf, wrows);
  run<1>("partners 4 ids apart (other XCD), idle chip", 8, 4, 0, wbuf, wrows);
  run<1>("partners 8 ids apart (same XCD), 64 pairs", 64, 8, 0, wbuf, wrows);
  run<1>("partners 1 id apart (other XCD), 64 pairs", 64, 1, 0, wbuf, wrows);
  run<1>("partners 8 ids apart (same XCD), 64 pairs + 128 writers", 64, 8, 1, wbuf, wrows);
  run<1>("partners 1 id apart (other XCD), 64 pairs + 128 writers", 64, 1, 1, wbuf, wrows);
  return 0;
}
