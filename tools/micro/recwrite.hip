// Development probe: what does the HBM sustain for the step engine's OUTPUT PATTERN -- 352-byte records (22 pieces of 16 bytes, lane <->
// piece, non-temporal stores) whose rows are not written in address order?  One-wave workgroups (like a copy-out: one wave writes ~33
// records), each "batch" = 32 records.  Patterns for the row of record j of batch b of wave w:
//   seq      rows in address order (a streaming fill in 352-byte runs)
//   win90k   a random row inside the wave's own 256-row window (90 KB: the rows of one workgroup at one step), window advancing with b
//   win4     the same, but the records of a batch spread over 4 neighbouring windows (envs of a workgroup that drifted apart by +-2 steps)
//   random   any row of the buffer
// and record strides 352 / 384 / 512 bytes.  Prints GB/s of payload (352 bytes per record).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long mix(unsigned long long x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; return x;
}
template <int PATTERN, bool NT, int P = 22>
__global__ __launch_bounds__(64) void rec_k(unsigned char* base, size_t nrows, int stride, int batches, size_t n_windows) {
  const int lane = threadIdx.x;
  const size_t w = blockIdx.x;
  for (int b = 0; b < batches; b++) {
    // 32 records x 22 pieces = 704 pieces = 11 rounds of 64 lanes
#pragma unroll 1
    for (int q0 = lane; q0 < 32 * P; q0 += 256) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int q = q0 + 64 * k;
        if (q < 32 * P) {
          const int j = q / P, pc = q - P * j;
          size_t row;
          const size_t win = (w + (size_t)b * gridDim.x) % n_windows; // the wave's window for this batch (advances like t)
          if (PATTERN == 0) row = ((w * batches + b) * 32 + j) % nrows;
          else if (PATTERN == 1) row = win * 256 + (mix(w * 7919 + b * 104729 + j) & 255);
          else if (PATTERN == 2) row = ((win + (mix(w * 31 + b * 17 + j) >> 20 & 3) * gridDim.x) % n_windows) * 256 + (mix(w * 7919 + b * 104729 + j) & 255);
          else row = mix(w * 7919 + (size_t)b * 104729 + j) % nrows;
          u32x4 v = {(unsigned)q, (unsigned)b, (unsigned)w, (unsigned)row};
          u32x4* dst = (u32x4*)(base + row * (size_t)stride + 16 * pc);
          if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
      }
    }
  }
}
// LANE = RECORD (round 5): every active lane writes the P pieces of ITS OWN record with P back-to-back stores -- one instruction = one 16-byte piece
// of up to 64 different rows (what an owner wave that kept its env's record in REGISTERS would issue).  `active` of the 64 lanes write per batch (the
// engine finishes ~19 records per owner iteration); rows: a random row of the wave's 256-row window (win90k) or of the whole buffer.
template <int PATTERN, int STORE, int P>
__global__ __launch_bounds__(64) void rec_lane_k(unsigned char* base, size_t nrows, int stride, int batches, size_t n_windows, int active) {
  const int lane = threadIdx.x;
  const size_t w = blockIdx.x;
  for (int b = 0; b < batches; b++) {
    const size_t win = (w + (size_t)b * gridDim.x) % n_windows;
    const bool on = (int)((mix(w * 131 + b * 7 + 1) + (unsigned)lane * 0x9E3779B9u) % 64u) < active;   // a changing subset of ~active lanes
    if (on) {
      const size_t row = PATTERN == 1 ? win * 256 + (mix(w * 7919 + b * 104729 + lane) & 255) : mix(w * 7919 + (size_t)b * 104729 + lane) % nrows;
      u32x4* dst = (u32x4*)(base + row * (size_t)stride);
#pragma unroll
      for (int pc = 0; pc < P; pc++) {
        u32x4 v = {(unsigned)pc, (unsigned)b, (unsigned)w, (unsigned)row};
        if (STORE == 1) __builtin_nontemporal_store(v, dst + pc); else dst[pc] = v;
      }
    }
  }
}
template <typename F> static double time_it(F launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch();
  CK(hipEventRecord(a, 0)); launch(); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e-3 / 2;
}
int main() {
  const size_t bytes = 8ull << 30;   // 8 GiB: the size of a 372-step record buffer
  unsigned char* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
  const int waves = 256 * 8, batches = 160;  // 2048 one-wave workgroups x 160 batches x 32 records = 10.5 M records = 3.7 GB of payload
  const char* names[4] = {"seq", "win90k", "win4", "random"};
  for (int stride : {352, 384, 512}) {
    const size_t nrows = bytes / stride, n_windows = nrows / 256;
    for (int nt = 1; nt >= 0; nt--) {
      double t[4];
      t[0] = time_it([&] { if (nt) hipLaunchKernelGGL((rec_k<0, true>), dim3(waves), dim3(64), 0, 0, buf, nrows, stride, batches, n_windows); else hipLaunchKernelGGL((rec_k<0, false>), dim3(waves), dim3(64), 0, 0, buf, nrows, stride, batches, n_windows); });
      t[1] = time_it([&] { if (nt) hipLaunchKernelGGL((rec_k<1, true>), dim3(waves), dim3(64), 0, 0, buf, nrows, stride, batches, n_windows); else hipLaunchKernelGGL((rec_k<1, false>), dim3(waves), dim3(64), 0, 0, buf, nrows, stride, batches, n_windows); });
      t[2] = time_it([&] { if (nt) hipLaunchKernelGGL((rec_k<2, true>), dim3(waves), dim3(64), 0, 0, buf, nrows, stride, batches, n_windows); else hipLaunchKernelGGL((rec_k<2, false>), dim3(waves), dim3(64), 0, 0, buf, nrows, stride, batches, n_windows); });
      t[3] = time_it([&] { if (nt) hipLaunchKernelGGL((rec_k<3, true>), dim3(waves), dim3(64), 0, 0, buf, nrows, stride, batches, n_windows); else hipLaunchKernelGGL((rec_k<3, false>), dim3(waves), dim3(64), 0, 0, buf, nrows, stride, batches, n_windows); });
      const double payload = (double)waves * batches * 32 * 352;
      printf("stride %3d %s :", stride, nt ? "nt   " : "plain");
      for (int p = 0; p < 4; p++) printf("  %s %7.1f GB/s", names[p], payload / t[p] / 1e9);
      printf("\n");
    }
  }
  // FULL-LINE records: every 16-byte piece of the stride is written (24 pieces = 384 bytes = 3 lines, 32 pieces = 512 bytes = 4 lines)
  {
    const double payload = (double)waves * batches * 32 * 352;
    size_t nrows = bytes / 384, n_windows = nrows / 256;
    double a = time_it([&] { hipLaunchKernelGGL((rec_k<1, true, 24>), dim3(waves), dim3(64), 0, 0, buf, nrows, 384, batches, n_windows); });
    double b = time_it([&] { hipLaunchKernelGGL((rec_k<3, true, 24>), dim3(waves), dim3(64), 0, 0, buf, nrows, 384, batches, n_windows); });
    double c = time_it([&] { hipLaunchKernelGGL((rec_k<1, false, 24>), dim3(waves), dim3(64), 0, 0, buf, nrows, 384, batches, n_windows); });
    printf("384-byte records written WHOLE (24 pieces): nt win90k %7.1f  nt random %7.1f  plain win90k %7.1f GB/s of 352-byte payload (x 384/352 raw)\n", payload / a / 1e9, payload / b / 1e9, payload / c / 1e9);
    nrows = bytes / 512; n_windows = nrows / 256;
    a = time_it([&] { hipLaunchKernelGGL((rec_k<1, true, 32>), dim3(waves), dim3(64), 0, 0, buf, nrows, 512, batches, n_windows); });
    b = time_it([&] { hipLaunchKernelGGL((rec_k<3, true, 32>), dim3(waves), dim3(64), 0, 0, buf, nrows, 512, batches, n_windows); });
    printf("512-byte records written WHOLE (32 pieces): nt win90k %7.1f  nt random %7.1f GB/s of 352-byte payload (x 512/352 raw)\n", payload / a / 1e9, payload / b / 1e9);
    nrows = bytes / 320; n_windows = nrows / 256;
    a = time_it([&] { hipLaunchKernelGGL((rec_k<1, true, 20>), dim3(waves), dim3(64), 0, 0, buf, nrows, 320, batches, n_windows); });
    printf("320-byte records (20 pieces, 64-byte multiples, stride 320): nt win90k %7.1f GB/s raw\n", (double)waves * batches * 32 * 320 / a / 1e9);
    nrows = bytes / 256; n_windows = nrows / 256;
    a = time_it([&] { hipLaunchKernelGGL((rec_k<1, true, 16>), dim3(waves), dim3(64), 0, 0, buf, nrows, 256, batches, n_windows); });
    printf("256-byte records (16 pieces): nt win90k %7.1f GB/s raw\n", (double)waves * batches * 32 * 256 / a / 1e9);
  }
  // lane = record: P = 24 pieces (384-byte whole-line records), 64 / 32 / 19 active lanes per batch; the same number of records per wave as above
  {
    const size_t nrows = bytes / 384, n_windows = nrows / 256;
    for (int active : {64, 32, 19}) {
      const int nb = batches * 32 / active;   // batches so that every wave writes ~ the same number of records
      const double payload = (double)waves * nb * active * 352;
      double a = time_it([&] { hipLaunchKernelGGL((rec_lane_k<1, 1, 24>), dim3(waves), dim3(64), 0, 0, buf, nrows, 384, nb, n_windows, active); });
      double b = time_it([&] { hipLaunchKernelGGL((rec_lane_k<3, 1, 24>), dim3(waves), dim3(64), 0, 0, buf, nrows, 384, nb, n_windows, active); });
      double c = time_it([&] { hipLaunchKernelGGL((rec_lane_k<1, 0, 24>), dim3(waves), dim3(64), 0, 0, buf, nrows, 384, nb, n_windows, active); });
      printf("LANE = RECORD, 24 pieces per lane back to back, %2d of 64 lanes active: nt win90k %7.1f  nt random %7.1f  plain win90k %7.1f GB/s of 352-byte payload\n",
             active, payload / a / 1e9, payload / b / 1e9, payload / c / 1e9);
    }
  }
  return 0;
}
