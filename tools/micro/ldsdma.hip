// What `global_load_lds_dword` / `_dwordx4` (LDS-DMA, `__builtin_amdgcn_global_load_lds`) does on gfx950 in exactly the situations the step engine's
// prefetches use it (bg_device.h RngWin, bg_step.h bg_prefetch_shop / _blood / _tmpl):
//   1. under a DIVERGENT exec mask: does an active lane's data land at base + size x LANE ID (what the engine assumes), or are the active lanes compacted?
//   2. from a source address that is only DWORD aligned (a Bloodstone word, the global stream's cursor is anywhere) -- for the 4-byte form;
//      the 16-byte form is only ever used on 16-byte aligned sources (slot tails, templates) but is checked unaligned too, for the record;
//   3. is the data there behind `s_waitcnt vmcnt(0)` of the issuing wave, with no barrier;
//   4. how long from issue to landed, L2-cold (first touch) and L2-warm;
//   5. with the destination ABOVE 64 KiB of the workgroup's LDS (the engine's windows sit ~130 KB into a 154 KB allocation): M0 carries the LDS base -- all of
//      it, or 16 bits?  (kernel probe<PAD>: PAD bytes of LDS in front of the destination.)
// Prints one line per check; exit code 1 when an assumption the engine makes does not hold.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void g_cvoid;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// out[block][0..63]: what lane l finds at win4[l] (x component) / win1[l]; out[block][64..]: cycles
template <int PAD>
struct Lds { unsigned pad[PAD / 4 + 4]; __attribute__((aligned(16))) unsigned win4[64][4]; unsigned win1[64]; };
template <int PAD>
__global__ __launch_bounds__(64) void probe(const unsigned* __restrict__ src, unsigned* out, unsigned long long mask, int misalign) {
  __shared__ Lds<PAD> L;
  auto& win4 = L.win4; auto& win1 = L.win1;
  const int lane = threadIdx.x;
  for (int k = 0; k < 4; k++) win4[lane][k] = 0xdeadbeefu;
  win1[lane] = 0xdeadbeefu;
  for (int i = lane; i < PAD / 4; i += 64) L.pad[i] = 0x0badf00du;   // (a destination that wrapped modulo 64 KiB would land in here)
  __syncthreads();
  // lane l's source: 64 words apart (another line per lane), + misalign words
  const unsigned* p = src + ((size_t)blockIdx.x * 64 + lane) * 64 + misalign;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if ((mask >> lane) & 1ull) {
    __builtin_amdgcn_global_load_lds((g_cvoid*)p, (lds_void*)&win4[0][0], 16, 0, 0);
    __builtin_amdgcn_global_load_lds((g_cvoid*)(p + 7), (lds_void*)&win1[0], 4, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  const u32x4 v = *(const __attribute__((address_space(3))) u32x4*)&win4[lane][0];
  unsigned* o = out + (size_t)blockIdx.x * 512;
  o[lane] = v.x; o[64 + lane] = v.y; o[128 + lane] = v.z; o[192 + lane] = v.w; o[256 + lane] = win1[lane];
  if (lane == 0) o[320] = (unsigned)(t1 - t0);
  unsigned dirty = 0;
  for (int i = lane; i < PAD / 4; i += 64) dirty += L.pad[i] != 0x0badf00du;
  if (dirty) atomicAdd(&o[321], dirty);
}

int main() {
  const int blocks = 256;
  const size_t n = (size_t)blocks * 64 * 64 + 64;
  std::vector<unsigned> h(n);
  for (size_t i = 0; i < n; i++) h[i] = (unsigned)(i * 2654435761u + 12345u);
  unsigned *src, *out;
  CK(hipMalloc(&src, n * 4)); CK(hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, (size_t)blocks * 512 * 4));
  std::vector<unsigned> r((size_t)blocks * 512);
  int bad_total = 0;
  const unsigned long long masks[3] = {~0ull, 0xaaaaaaaaaaaaaaaaull, 0x00f0100000000801ull};
  for (int pad = 0; pad < 2; pad++)
  for (int mis = 0; mis < 4; mis += 1) {
    for (int mi = 0; mi < 3; mi++) {
      for (int rep = 0; rep < 2; rep++) {   // rep 0: L2-cold for this src region on the first (mis, mask); later ones warm
        CK(hipMemset(out, 0, (size_t)blocks * 512 * 4));
        if (pad) hipLaunchKernelGGL(probe<135168>, dim3(blocks), dim3(64), 0, 0, src, out, masks[mi], mis);
        else hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(64), 0, 0, src, out, masks[mi], mis);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(r.data(), out, r.size() * 4, hipMemcpyDeviceToHost));
        int bad16 = 0, bad4 = 0, untouched_wrong = 0; double cyc = 0; unsigned long long dirty = 0;
        for (int b = 0; b < blocks; b++) {
          const unsigned* o = &r[(size_t)b * 512];
          cyc += o[320]; dirty += o[321];
          for (int l = 0; l < 64; l++) {
            const size_t s0 = ((size_t)b * 64 + l) * 64 + mis;
            const bool on = (masks[mi] >> l) & 1ull;
            if (on) {
              if (o[l] != h[s0] || o[64 + l] != h[s0 + 1] || o[128 + l] != h[s0 + 2] || o[192 + l] != h[s0 + 3]) bad16++;
              if (o[256 + l] != h[s0 + 7]) bad4++;
            } else if (o[l] != 0xdeadbeefu || o[256 + l] != 0xdeadbeefu) untouched_wrong++;
          }
        }
        printf("LDS destination at +%6d B  misalign %d words  mask %016llx  rep %d: dwordx4 wrong lanes %d, dword wrong lanes %d, inactive lanes overwritten %d, words landed BELOW the destination %llu, issue->landed %.0f cycles (both forms, one wave per CU)\n",
               pad ? 135168 : 0, mis, masks[mi], rep, bad16, bad4, untouched_wrong, dirty, cyc / blocks);
        bad_total += dirty ? 1 : 0;
        // what the engine relies on: everything for the 4-byte form; the 16-byte form at misalign 0
        bad_total += bad4 + untouched_wrong + (mis == 0 ? bad16 : 0);
      }
    }
  }
  printf(bad_total ? "FAIL: an assumption of the engine's LDS-DMA prefetches does not hold\n" : "ok: data lands at base + size x lane id under any exec mask, dword form from any dword-aligned source\n");
  return bad_total ? 1 : 0;
}
