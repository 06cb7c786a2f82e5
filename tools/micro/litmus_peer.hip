// Litmus test for the sharded gather WITHOUT a collective (bg_set_gather_peers, bg_engine3.h copy-out): rank A's kernel writes records into rank B's
// buffer through a HIP IPC mapping with NON-TEMPORAL 16-byte stores (over xGMI when A and B are two GPUs), and B reads them with its NEXT kernel.
//
// Which fence makes those stores visible to the peer's next kernel -- the chain this program exercises, with the reader holding the OLD lines in
// its caches (it reads its whole buffer before every round: the stale-line case):
//   1. the WRITER's end of kernel: the command processor's release at system scope writes back whatever the writer's L2 still holds of the peer
//      lines (`nt` is not write-through, MI355X_MICROARCH.md "stores of each flavour") -- so "the writer's stream has completed"
//      (hipStreamSynchronize / the event behind the launch) is the point from which the bytes are in the peer's memory (HBM / its memory-side cache);
//   2. a host-side barrier between the ranks behind that synchronisation (bench.py's closing barrier, the tests' dist.barrier());
//   3. the READER's next kernel start: the acquire at the head of every dispatch invalidates the CU L1s and the (not mutually coherent) XCD L2s,
//      so its loads miss down to the memory side, which incoming xGMI writes have already updated.
// A reader that polls INSIDE a running kernel gets none of this (it would need agent/system-scope acquires per poll): the design never does that.
//
// usage: litmus_peer [rounds] [records]     two processes; device r % (number of devices): with ONE GPU both ranks share it (the mapping is then a
// same-device IPC mapping, as in tests/test_sharded_one_gpu.py), with two or more it is a real peer mapping.  Prints "stale 0 of N" per rank; exit 1 on a stale read.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("rank %d: %s: %s\n", g_rank, #x, hipGetErrorString(e_)); _exit(2); } } while (0)
static int g_rank = -1;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define PIECES 22   // 16-byte pieces of a 352-byte record, as the engine writes them

struct Shared {
  std::atomic<int> arrive;
  hipIpcMemHandle_t handle[2];
  int ndev;
};
static void barrier(Shared* s, int& gen) {   // both ranks arrive gen times each
  gen++;
  s->arrive.fetch_add(1);
  while (s->arrive.load() < 2 * gen) usleep(50);
}

__device__ __forceinline__ u32x4 pattern(unsigned round, unsigned rank, unsigned rec, unsigned pc) {
  return u32x4{round * 0x9E3779B9u + rec, rank ^ (pc << 8) ^ (round << 16), rec * 2654435761u + pc, ~round ^ (rec + pc)};
}
// lane <-> piece: consecutive lanes write consecutive pieces of a record (the engine's copy-out pattern), non-temporal
__global__ __launch_bounds__(256) void writer(u32x4* peer, unsigned n_rec, unsigned round, unsigned rank) {
  const size_t total = (size_t)n_rec * PIECES;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (size_t)gridDim.x * blockDim.x) {
    const unsigned rec = (unsigned)(q / PIECES), pc = (unsigned)(q % PIECES);
    __builtin_nontemporal_store(pattern(round, rank, rec, pc), (__attribute__((address_space(1))) u32x4*)&peer[q]);
  }
}
// reads the whole buffer (every line into this GPU's caches) and counts words that differ from the expected round's pattern
__global__ __launch_bounds__(256) void reader(const u32x4* mine, unsigned n_rec, unsigned round, unsigned writer_rank, unsigned long long* bad) {
  const size_t total = (size_t)n_rec * PIECES;
  unsigned long long b = 0;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (size_t)gridDim.x * blockDim.x) {
    const unsigned rec = (unsigned)(q / PIECES), pc = (unsigned)(q % PIECES);
    const u32x4 v = mine[q], w = pattern(round, writer_rank, rec, pc);
    b += (v.x != w.x) + (v.y != w.y) + (v.z != w.z) + (v.w != w.w);
  }
  if (b) atomicAdd(bad, b);
}

static int run_rank(int rank, Shared* sh, int rounds, unsigned n_rec) {
  g_rank = rank;
  int ndev = 0;
  CK(hipGetDeviceCount(&ndev));
  if (rank == 0) sh->ndev = ndev;
  CK(hipSetDevice(rank % ndev));
  const size_t bytes = (size_t)n_rec * PIECES * 16;
  u32x4* mine = nullptr;
  unsigned long long* bad = nullptr;
  CK(hipMalloc(&mine, bytes)); CK(hipMemset(mine, 0, bytes));
  CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
  CK(hipIpcGetMemHandle(&sh->handle[rank], mine));
  int gen = 0;
  barrier(sh, gen);
  u32x4* peer = nullptr;
  CK(hipIpcOpenMemHandle((void**)&peer, sh->handle[1 - rank], hipIpcMemLazyEnablePeerAccess));
  unsigned long long stale = 0, words = 0;
  for (int r = 1; r <= rounds; r++) {
    // the stale-line case: the reader holds round r-1's lines (it has just read them all)
    hipLaunchKernelGGL(reader, dim3(512), dim3(256), 0, 0, mine, n_rec, (unsigned)(r - 1), (unsigned)(1 - rank), bad);
    CK(hipDeviceSynchronize());
    if (r == 1) CK(hipMemset(bad, 0, 8));   // (round 0 is the zero fill)
    barrier(sh, gen);                    // nobody still reads round r-1 when round r is written
    hipLaunchKernelGGL(writer, dim3(512), dim3(256), 0, 0, peer, n_rec, (unsigned)r, (unsigned)rank);
    CK(hipDeviceSynchronize());             // (1) the writer's end of kernel
    barrier(sh, gen);                    // (2) the ranks meet
    hipLaunchKernelGGL(reader, dim3(512), dim3(256), 0, 0, mine, n_rec, (unsigned)r, (unsigned)(1 - rank), bad);   // (3) the reader's next kernel
    CK(hipDeviceSynchronize());
    unsigned long long b = 0;
    CK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost));
    CK(hipMemset(bad, 0, 8));
    stale += b; words += (unsigned long long)n_rec * PIECES * 4;
  }
  barrier(sh, gen);
  CK(hipIpcCloseMemHandle(peer));
  barrier(sh, gen);   // both mappings are closed before either buffer is freed
  CK(hipFree(mine)); CK(hipFree(bad));
  printf("rank %d (device %d of %d%s): stale %llu of %llu words over %d rounds of %u records\n", rank, rank % ndev, ndev,
         ndev < 2 ? ", both ranks on ONE device: same-device IPC mapping, not xGMI" : ": peer mapping", stale, words, rounds, n_rec);
  fflush(stdout);   // (the child leaves through _exit)
  return stale ? 1 : 0;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200;
  const unsigned n_rec = argc > 2 ? (unsigned)atoi(argv[2]) : 65536u;
  Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (sh == MAP_FAILED) { perror("mmap"); return 2; }
  memset((void*)sh, 0, sizeof(Shared));
  pid_t pid[2];
  for (int r = 0; r < 2; r++) {   // fork BEFORE anything touches the GPU
    pid[r] = fork();
    if (pid[r] == 0) _exit(run_rank(r, sh, rounds, n_rec));
  }
  int rc = 0;
  for (int r = 0; r < 2; r++) { int st = 0; waitpid(pid[r], &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = 1; }
  return rc;
}
