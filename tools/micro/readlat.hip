// Development probe: what does a SCATTERED READ cost a wave while the CU's other waves stream records to HBM the way the step engine's
// copier waves do?  The play path of the engine waits ~10 k cycles for RNG words it requested ~8 k cycles earlier;
// is that the memory system under the record stream, and does the cache policy of the stores or of the loads change it?
//
// One workgroup per CU = W writer waves + 1 reader wave.
//   writers: whole 384-byte records (24 pieces of 16 bytes, lane <-> piece) to random rows of the workgroup's moving 96 KB window of an
//            8 GiB buffer -- 12 store instructions back to back, then s_sleep(pause) to set the rate.  Policy: 0 off, 1 non-temporal, 2 plain.
//   reader : every lane loads 16 bytes from its own random 128-byte line (64 lines per instruction: a service batch's state / RNG loads),
//            waits, and times the round trip with the shader clock.  Region: 4 GiB (HBM) or 2 MiB (stays in the L2).  Policy: 0 plain,
//            1 non-temporal, 2 sc0 sc1 (system scope).
//            K such loads may be in flight per round trip.
// Prints: achieved write bandwidth, mean / p50 / p95 round trip in shader cycles.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long mix(unsigned long long x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; return x;
}

template <int WPOL, int RPOL, int K = 1>
__global__ __launch_bounds__(64 * 8) void k(unsigned char* wbuf, size_t wrows, const unsigned char* rbuf, size_t rlines, int writers, int pause,
                                             int reads, unsigned* lat, volatile unsigned* stop, unsigned long long* wbytes) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t wg = blockIdx.x;
  if (wave == 0) { // reader
    unsigned long long acc = 0;
    for (int i = 0; i < reads; i++) {
      const u32x4* p[K];
#pragma unroll
      for (int q = 0; q < K; q++) { const size_t line = mix(wg * 1315423911ull + ((size_t)i * K + q) * 64 + lane) % rlines; p[q] = (const u32x4*)(rbuf + line * 128 + 16 * (lane & 7)); }
      const unsigned long long t0 = __builtin_readcyclecounter();
      u32x4 v[K];
#pragma unroll
      for (int q = 0; q < K; q++) {
        if (RPOL == 1) v[q] = __builtin_nontemporal_load(p[q]);
        else if (RPOL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[q]) : "v"(p[q]) : "memory");
        else v[q] = *(const volatile u32x4*)p[q];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < K; q++) acc += v[q].x; // use
      const unsigned long long t1 = __builtin_readcyclecounter();
      if (lane == 0) lat[wg * reads + i] = (unsigned)(t1 - t0);
      __builtin_amdgcn_s_sleep(20); // a service wave does a few hundred cycles of work between its loads
    }
    if (acc == 0x123456789ull) lat[0] = 0;
    __threadfence();
    if (lane == 0) atomicAdd((unsigned*)stop, 1u); // this workgroup's writers may stop
  } else if (wave <= writers && WPOL != 0) {
    unsigned long long n = 0;
    const size_t nwin = wrows / 256;
    for (unsigned b = 0;; b++) {
      if ((b & 15) == 0 && *stop >= gridDim.x) break;
      const size_t win = (wg + (size_t)b * gridDim.x) % nwin;
      // 32 records x 24 pieces = 768 pieces = 12 rounds of the wave
#pragma unroll
      for (int r = 0; r < 12; r++) {
        const int q = r * 64 + lane, j = q / 24, pc = q - 24 * j;
        const size_t row = win * 256 + (mix(wg * 7919 + b * 104729ull + wave * 31 + j) & 255);
        u32x4 v = {(unsigned)q, b, (unsigned)wg, (unsigned)row};
        u32x4* dst = (u32x4*)(wbuf + row * 384 + 16 * pc);
        if (WPOL == 1) __builtin_nontemporal_store(v, dst); else *dst = v;
      }
      n += 12 * 1024;
      for (int s = 0; s < pause; s++) __builtin_amdgcn_s_sleep(32);
    }
    if (lane == 0) atomicAdd(wbytes, n);
  }
}

template <int WPOL, int RPOL, int K = 1>
static void run(const char* name, unsigned char* wbuf, size_t wrows, unsigned char* rbuf, size_t rlines, int writers, int pause) {
  const int grid = 256, reads = 400;
  unsigned* lat; unsigned* stop; unsigned long long* wbytes;
  CK(hipMalloc(&lat, sizeof(unsigned) * grid * reads)); CK(hipMalloc(&stop, 4)); CK(hipMalloc(&wbytes, 8));
  CK(hipMemset(stop, 0, 4)); CK(hipMemset(wbytes, 0, 8));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a, 0));
  hipLaunchKernelGGL((k<WPOL, RPOL, K>), dim3(grid), dim3(64 * 8), 0, 0, wbuf, wrows, rbuf, rlines, writers, pause, reads, lat, stop, wbytes);
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  std::vector<unsigned> h(grid * reads); unsigned long long wb;
  CK(hipMemcpy(h.data(), lat, sizeof(unsigned) * grid * reads, hipMemcpyDeviceToHost)); CK(hipMemcpy(&wb, wbytes, 8, hipMemcpyDeviceToHost));
  std::vector<unsigned> s; for (int g = 0; g < grid; g++) for (int i = reads / 4; i < reads; i++) s.push_back(h[g * reads + i]);
  std::sort(s.begin(), s.end());
  double mean = 0; for (unsigned v : s) mean += v; mean /= s.size();
  printf("%-44s writers %d pause %2d : write %6.0f GB/s   read round trip mean %6.0f  p50 %6u  p95 %6u  p99 %6u cycles  (%.2f ms)\n", name, writers, pause,
         wb / (ms * 1e-3) / 1e9, mean, s[s.size() / 2], s[s.size() * 95 / 100], s[s.size() * 99 / 100], ms);
  CK(hipFree(lat)); CK(hipFree(stop)); CK(hipFree(wbytes));
}

int main() {
  const size_t wbytes = 8ull << 30, rbytes = 4ull << 30;
  unsigned char *wbuf, *rbuf;
  CK(hipMalloc(&wbuf, wbytes)); CK(hipMalloc(&rbuf, rbytes)); CK(hipMemset(wbuf, 0, wbytes)); CK(hipMemset(rbuf, 1, rbytes));
  const size_t wrows = wbytes / 384, cold = rbytes / 128, hot = (2ull << 20) / 128;
  run<0, 0>("no writers, reads from HBM", wbuf, wrows, rbuf, cold, 0, 0);
  run<0, 0>("no writers, reads from the L2 (2 MiB)", wbuf, wrows, rbuf, hot, 0, 0);
  for (int pause : {0, 2, 4, 8, 16}) {
    run<1, 0>("nt stores, plain reads from HBM", wbuf, wrows, rbuf, cold, 2, pause);
  }
  for (int pause : {2, 4, 8}) {
    run<1, 0>("nt stores, plain reads from the L2", wbuf, wrows, rbuf, hot, 2, pause);
    run<2, 0>("plain stores, plain reads from HBM", wbuf, wrows, rbuf, cold, 2, pause);
    run<2, 0>("plain stores, plain reads from the L2", wbuf, wrows, rbuf, hot, 2, pause);
    run<1, 1>("nt stores, nt reads from HBM", wbuf, wrows, rbuf, cold, 2, pause);
    run<1, 2>("nt stores, sc0 sc1 reads from HBM", wbuf, wrows, rbuf, cold, 2, pause);
  }
  // K independent scattered loads in flight per round trip (a burst of touches / the loads of one service step issued together)
  run<1, 0, 2>("nt stores, 2 loads in flight, HBM", wbuf, wrows, rbuf, cold, 2, 2);
  run<1, 0, 4>("nt stores, 4 loads in flight, HBM", wbuf, wrows, rbuf, cold, 2, 2);
  run<1, 0, 8>("nt stores, 8 loads in flight, HBM", wbuf, wrows, rbuf, cold, 2, 2);
  run<1, 0, 16>("nt stores, 16 loads in flight, HBM", wbuf, wrows, rbuf, cold, 2, 2);
  run<0, 0, 8>("no writers, 8 loads in flight, HBM", wbuf, wrows, rbuf, cold, 0, 0);
  return 0;
}
