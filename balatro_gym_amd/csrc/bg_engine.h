// bg_engine.h -- the step engine: ONE kernel behind bg_step, bg_step_many, bg_rollout and bg_rollout_rows.
//
// A workgroup owns 256 envs (one workgroup per CU).  For the length of the launch every env has, in LDS,
//   * its observation RECORD IMAGE: the 352-byte packed record of include/balatro_mi355x.h (BG_ROW_*), always current;
//   * hot chunks 3 and 4 (phase / counters / hand / selection order): all a card-select toggle needs;
//   * its deck (52 card codes), its cached 60-bit action mask and its step counter;
// the other six hot chunks stay in HBM (the service batches below load and store them; every access to an env's state comes
// from this CU, whose vector-memory pipeline is in order).  The waves are WORKERS, not owners: each pulls a batch of up to 64
// envs that need the same kind of work from one of three LDS queues and completes one step for every env of the batch:
//   run queue    envs whose next step is due: counter-hash policy (or the caller's action) on the cached mask, guards
//                (balatro_env_2.py:619-627).  A card-select toggle (84 % of a random policy's steps, :1052-1058) or a rejected
//                action is settled ON THE IMAGE: the selection list in chunk 4, one int64 of selected_cards, the two mask
//                bytes that depend on "anything selected", reward / action / terminated -- no unpack of the state, no record
//                build.  Everything else is queued (the state is untouched, so queueing costs one LDS word).
//   play queue   PLAY_HAND (bg_step_play_hand: classification, joker chain, reward shaping, blind outcome, shop generation)
//   other queue  DISCARD, blind select / skip, shop end / buy / reroll / sell, consumables, the terminal guards
//                -- full state from HBM + LDS, the step, curriculum cap, SAME_STEP auto-reset, mask, a freshly built image.
// Every batch ends with the COPY-OUT of its images to their rows: lane <-> 16-byte piece, consecutive lanes = consecutive
// pieces of a row (whole 352-byte runs per row), trip count proportional to the envs of the batch.
// Why: the kernels before this one kept lane = env and rebuilt the 88-dword record from the unpacked state on every step
// (~700 of the ~1100 instructions of a step whose game logic is a dozen), at 35 of 64 lanes; a SIMD's VALU was 50 % busy doing
// that.  Per-env step counters drift apart as before (row = env + t * N; envs are independent).
// No workgroup barrier inside the loop; queues and counters are LDS words (one wave's LDS operations execute in program
// order, so "data, then flag" needs compiler barriers only); every spin is bounded (sticky device error instead of a hang).
#pragma once

#define BG_Q_RUN 0
#define BG_Q_PLAY 1
#define BG_Q_OTHER 2
#define BG_ITEM_VALID 0x80000000u
#define BG_SPIN_LIMIT (1u << 24)
#define BG_DEVERR_SPIN 16u
#define BG_ENG_LNE 8     // log2 of the envs per workgroup.  Only 8 is supported: 7 (128 envs: twice the workgroups for a small job) passes the parity tests and
                         // gives +9 % at 4 096 envs, 6 gives +15 % but trips a bounded wait (profiles/r03_small_workgroups_ab.txt)
static_assert(BG_ENG_LNE == 8, "only 256 envs per workgroup are supported: items carry the env lane in 8 bits, copy-queue generations in 7, and the epilogue assumes NE = 256");
#define BG_ENG_NE (1 << BG_ENG_LNE)   // envs per workgroup: 256 (an item holds the env's lane in 8 bits; rings have NE entries, an item's generation = position >> LNE)
// Waves per workgroup.  SEVEN, not eight: a wave of this kernel needs 256 VGPRs, so eight fill the register file of all four SIMDs and
// nothing can be placed beside the workgroup -- the RNG refill of the previous launch (~1 ms of one-wave workgroups) then waits for
// the engine to retire and the next launch waits for the refill.  With seven, one SIMD per CU keeps 256 free registers (and the
// workgroup leaves ~5 KB of LDS), the dispatcher places the refill's one-wave workgroups there (multi-wave workgroups do NOT fit:
// tools/micro/corun.hip), and the refill runs beside the engine: +8.5 % env-steps/s measured on one box although the kernel itself
// is slower with seven waves and a busy neighbour (3.86 -> 4.30 ms per 372-step launch; -DBG_ENG_NW=8 is the old shape).
#define BG_ENG_NW 7
#define BG_ENG_OCC 2     // waves per SIMD the register budget is set for (2 = 256 VGPRs; 3 = 168: measured, spills)
#define BG_ENG_IMG_PIECES 22 // 16-byte pieces from one env's LDS record image to the next (22 = dense; 23: fewer bank conflicts)
#define BG_ENG_NSV 4    // of them, how many own an RNG window and may run service batches
#define BG_ENG_SMASK_DEFAULT (((1u << BG_ENG_NSV) - 1u) << (BG_ENG_NW - BG_ENG_NSV)) // which: the last NSV waves (BG_ENG_SMASK)

struct EngineArgs {
  int T;                       // steps per env in this launch
  int policy;                  // BG_POLICY_* (ignored when actions_in is set)
  uint64_t policy_seed, env_index0, t0;
  ObsPtrs obs;
  int obs_stride_steps;        // != 0: row = env + t * N ([T, N] buffers); 0: row = env (overwritten every step)
  double* reward; uint8_t* term; uint8_t* trunc; int32_t* actions_out;
  const int32_t* actions_in;   // [T, N] actions (row t * N + env) or null = counter-hash policy on device
  InfoPtrs info;               // bg_step's info arrays (null pointers are skipped)
  bg_rollout_stats* stats;
  uint32_t th_run, th_play, th_other; // a queue is served once it holds this many items ...
  uint32_t th_part;            // ... or, while other waves are busy (their envs will come back soon), this many; anything when no wave is busy
  uint32_t serve_mask;         // bit w: wave w owns an RNG window and may run service batches (<= BG_ENG_NSV bits)
  uint32_t n_waves;            // worker waves that stay (BG_ENG_NSV .. BG_ENG_NW): the others retire before the worker loop
  uint32_t th_more;            // further cheap steps an env may take inside the batch that has it
  uint32_t autoreset;          // SAME_STEP auto-reset of terminated envs
  uint32_t copier;             // packed records: wave `n_waves` is the COPIER (it writes the records and hands the envs back), the workers never copy
  uint32_t epw;                // bg_engine3.h, the 64-env workgroup shape only: envs per workgroup that are LIVE (8 .. 64; 0 = all 64) -- a small job spreads over more CUs
  // bg_engine3.h, sharded jobs (bg_set_gather_peers): the record of the launch's LAST step is also written into every rank's gather buffer
  // ([world][N][352] bytes, rank `grank`'s shard), by the owner waves' copy-out, while the launch runs.  gworld = 0: no gather.
  // (gpeer: a DEVICE array of the `gworld` buffer pointers, read with scalar loads where a last-step record is written -- once per env and launch; as
  //  eight by-value kernel arguments they were sixteen scalar registers live through the whole owner loop)
  uint8_t* const* gpeer;
  uint32_t gworld, grank;
};

__device__ __forceinline__ uint32_t bg_lds_ld(uint32_t* p) {
  uint32_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  return v;
}
__device__ __forceinline__ void bg_lds_st(uint32_t* p, uint32_t v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t bg_lds_ld(uint16_t* p) {
  uint16_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  return (uint32_t)v;
}
__device__ __forceinline__ void bg_lds_st(uint16_t* p, uint32_t v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(p, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void bg_vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#ifdef BG_ENG_TL   // development: the timeline of a workgroup in 10 ns ticks, summed over workgroups into d.dbg (tools/eng_timeline.py)
constexpr bool kEngTl = true;
#else
constexpr bool kEngTl = false;
#endif
// (sums in LDS, flushed by the workgroup's last instructions: 256 workgroups x a dozen global atomics on one line per event would be the timeline)
#define BG_TL(k) do { if constexpr (kEngTl) __hip_atomic_fetch_add(&s_tl[k], wall_clock64() - tl_k0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)


// INFO: the launch serves bg_step / bg_step_many (per-step info arrays); false for the rollouts
template <bool HASH, bool CARDS, bool INFO>
__global__ __launch_bounds__(BG_ENG_NW * BG_BLOCK, BG_ENG_OCC) void bg_engine_kernel(BgDev d, EngineArgs a) {
  constexpr int NE = BG_ENG_NE, NW = BG_ENG_NW, NSV = BG_ENG_NSV;
  // record images: 22 pieces of 16 bytes per env, IMGP pieces apart.  A row stride of 22 pieces = 88 dwords puts lane l's word k in bank
  // (24 l + k) mod 32: four banks for the whole wave, a 16-way conflict on every per-lane field access of a cheap step; 23 pieces (92 dwords)
  // spread the lanes over eight bank groups and make the image build's 16-byte stores conflict-free.
  __shared__ bg_u32x4 s_img[NE][BG_ENG_IMG_PIECES];
  __shared__ uint4 s_c34[2][NE];              // hot chunks 3 and 4
  __shared__ uint32_t s_deck[16][NE];
  __shared__ unsigned long long s_mask[NE];
  __shared__ uint32_t s_t[NE], s_prod[NE];
  __shared__ uint32_t s_q[3][NE];             // rings of items: env lane | generation of the ring position << 8 | action << 16 | VALID
  // queue control words, 32 bytes: [0..2] items ever queued per queue, [3] envs that have finished their T steps,
  // [4..6] items ever claimed per queue, [7] waves inside a batch -- a wave reads all eight with two 16-byte LDS loads
  __shared__ __attribute__((aligned(16))) uint32_t s_ctl[8];
  uint32_t* const s_tail = &s_ctl[0];
  uint32_t* const s_head = &s_ctl[4];
#define s_done s_ctl[3]
#define s_busy s_ctl[7]
  __shared__ uint32_t s_win[NSV][BG_WIN][BG_BLOCK]; // RNG windows of the service-capable waves
  // copy queue (packed records): one entry per finished step -- .x = record row of this launch, .y = env lane | done << 8 | generation of the
  // ring position << 16 | VALID -- written by the workers, read by the copier wave alone (its head is a register)
  __shared__ uint2 s_cq[NE];
  __shared__ uint32_t s_cqt;                  // entries ever queued
  __shared__ bg_u32x4 s_zero;                 // the two padding pieces of a 384-byte record are read from here
  __shared__ unsigned long long s_stats[6];   // the workgroup's share of bg_rollout_stats (bg_step.h bg_stats_wave / bg_stats_flush)
  __shared__ JTables jt;
  __shared__ unsigned long long s_tl[kEngTl ? 32 : 1];
  const unsigned long long tl_k0 = kEngTl ? wall_clock64() : 0ull;
  if constexpr (kEngTl) { if (threadIdx.x < 32) s_tl[threadIdx.x] = 0ull; __syncthreads(); }
  __builtin_amdgcn_s_setprio(3);
  BG_PROBE_INIT();
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int env0 = blockIdx.x * NE;
  const int n_live = d.N - env0 < NE ? d.N - env0 : NE;
  const bool can_serve = ((a.serve_mask >> wave) & 1u) != 0;
  const int serve_idx = __popc(a.serve_mask & ((1u << wave) - 1u)); // which RNG window
  using DeckT = DeckLdsS<NE, CARDS>;
  const size_t N = (size_t)d.N;
  // One array per key, overwritten in place (bg_step; a rollout without per-step buffers): only the LAST observation of the launch is ever read, and every
  // env's image holds it when the launch ends -- the arrays are written ONCE, behind the loop, thread = env (consecutive lanes = consecutive rows: whole
  // lines per store), instead of 31 scattered partial stores per lane inside every batch (a one-step launch: its play batch 20.7 -> 17 us, its run batches
  // 6.6 -> 2.8 us; the per-step values -- reward, terminated, action, info -- are still written by the step that produces them)
  const bool keys_at_end = !a.obs.rows && a.obs_stride_steps == 0;
  // ---------------------------------------------------------------- prologue: HBM -> LDS, first actions classified, images built; thread = env
  // Round 6 (tools/eng_timeline.py, profiles/r06/eng_timeline_*.txt): a ONE-step launch -- bg_step, bg_step_rows -- is all latency, and the prologue was
  // four round trips in series: tables (global -> LDS, barrier), state, image builds (barrier), run queue -> first play batch at 11 us of a 30 us workgroup.
  //  (a) every global load of the prologue is ISSUED before the first barrier: the joker tables (all threads) and, thread = env, the state, the deck, the
  //      producer counter and the launch's first action -- one round trip; the queue words are zeroed in front of the same barrier;
  //  (b) behind it the action mask -- and, when the caller supplies the actions (bg_step / bg_step_rows / bg_step_many), the env's FIRST action is
  //      classified right here: one that needs a service batch (PLAY_HAND, DISCARD, blind select, shop buy / reroll / sell, a consumable, a terminal guard)
  //      goes straight into its service queue WITHOUT a record image (the service step builds the env's image itself), in front of the second barrier --
  //      so the waves that own no env thread find the launch's service batches complete when they enter their loops;
  //  (c) behind the second barrier the envs whose first step is a cheap one build their image (a cheap step patches it) at LOW priority -- an image build
  //      is ~900 back-to-back VALU instructions, and the service wave that shares its SIMD is the launch's critical path -- and join the run queue.
  // (Classification only decides WHERE an action is settled: a service batch settles any action, so a cheap one sent there is still right.)
  if (tid < 3) { s_tail[tid] = 0u; s_head[tid] = 0; }
  if (tid == 0) { s_done = 0; s_busy = 0; s_cqt = 0; s_zero = bg_u32x4{0u, 0u, 0u, 0u}; }
  if (tid < 6) s_stats[tid] = 0ull;
  const bool mine = tid < NE && tid < n_live;
  const int l0 = tid, envp = env0 + tid;
  if (tid < NE) { s_q[BG_Q_RUN][l0] = 0; s_q[BG_Q_PLAY][l0] = 0; s_q[BG_Q_OTHER][l0] = 0; s_cq[l0] = make_uint2(0u, 0u); s_t[l0] = 0; }
  constexpr int JW = (int)(sizeof(JTables) / 4), JPT = (JW + NW * BG_BLOCK - 1) / (NW * BG_BLOCK);
  uint32_t jv[JPT];
#pragma unroll
  for (int k = 0; k < JPT; k++) { const int i = tid + k * NW * BG_BLOCK; jv[k] = i < JW ? d.jtab[i] : 0u; }
  bool svc = false;   // this env's first step is a service queue's
  {
    uint4 c[BG_NHOT];
    uint4 dw[BG_NDECK];
    uint32_t prod = 0u;
    int act0 = 0;
    if (mine) {
#pragma unroll
      for (int k = 0; k < BG_NHOT; k++) c[k] = d.hot[(size_t)k * N + envp];
#pragma unroll
      for (int k = 0; k < BG_NDECK; k++) dw[k] = d.deck[(size_t)k * N + envp];
      prod = d.prod_view ? d.prod_view[envp] : 0u;
      if (a.actions_in) act0 = a.actions_in[envp];   // (step 0 of the launch)
    }
    __syncthreads();   // the queue words are zero and every slot of the three rings reads "not written" (0) before the first append
    if (tid == 0) BG_TL(1);
#pragma unroll
    for (int k = 0; k < JPT; k++) { const int i = tid + k * NW * BG_BLOCK; if (i < JW) ((uint32_t*)&jt)[i] = jv[k]; }
    if (mine) {
      s_c34[0][l0] = c[3]; s_c34[1][l0] = c[4];
      DeckT dk; dk.col = (lds_u32*)&s_deck[0][l0];
#pragma unroll
      for (int k = 0; k < BG_NDECK; k++) bg_deck_set(dk, k, dw[k]);
      s_prod[l0] = prod;
      Env pe;
      ShopRegs psr; psr.valid = false;
      bg_unpack(c, pe);
      bg_derive_ready(pe, prod);
      const uint64_t pmask = bg_action_mask(d, envp, pe, psr);
      s_mask[l0] = pmask;
      if (a.actions_in) {
        const bool valid = act0 >= 0 && act0 < 60 && ((pmask >> (act0 & 63)) & 1ull);
        const bool terminal = pe.ante > 100 || pe.chips_scored > 1000000000ll;
        const bool cheap = !terminal && (!valid || (pe.phase == 0 && act0 >= 2 && act0 < 10) || (pe.phase == 1 && act0 == 31 && pe.hand_size <= pe.nhand));
        svc = !cheap;
        if (svc) {
          const int q = (!terminal && pe.phase == 0 && act0 == 0) ? BG_Q_PLAY : BG_Q_OTHER;
          const uint32_t slot = __hip_atomic_fetch_add(&s_tail[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (< NE: generation 0)
          bg_lds_st(&s_q[q][slot & (NE - 1)], (uint32_t)l0 | (((slot >> BG_ENG_LNE) & 0xffu) << 8) | (((uint32_t)act0 & 0x7fffu) << 16) | BG_ITEM_VALID);
        }
      }
    }
  }
  if (tid == 0) BG_TL(2);
  __syncthreads();   // tables, chunks 3 / 4, decks, masks in LDS; the launch's first service batches are queued
  if (tid == 0) BG_TL(3);
  // (c): image first, queue word second (one lane's LDS operations execute in program order).  No barrier behind it: the other waves are already in their
  // loops.  (The state is unpacked a second time, from lines this CU has just read: kept in registers across the barrier it was 30 spilled registers.)
  {
    if (mine && !svc) {
      __builtin_amdgcn_s_setprio(1);
      uint4 c[BG_NHOT];
#pragma unroll
      for (int k = 0; k < BG_NHOT; k++) if (k != 3 && k != 4) c[k] = d.hot[(size_t)k * N + envp];
      c[3] = s_c34[0][l0]; c[4] = s_c34[1][l0];
      Env pe;
      ShopRegs psr; psr.valid = false;
      bg_unpack(c, pe);
      bg_derive_ready(pe, s_prod[l0]);
      const uint64_t pmask = s_mask[l0];
      DeckT dk; dk.col = (lds_u32*)&s_deck[0][l0];
      const ObsPtrs none{};
      bg_write_obs_impl<false, 3>(d, envp, 0, pe, dk, none, pmask, psr, RowExtra{0.0, 0, 0u}, RowStage{(lds_u4*)&s_img[l0][0], nullptr});
      const uint32_t slot = __hip_atomic_fetch_add(&s_tail[BG_Q_RUN], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      bg_lds_st(&s_q[BG_Q_RUN][slot & (NE - 1)], (uint32_t)l0 | (((slot >> BG_ENG_LNE) & 0xffu) << 8) | BG_ITEM_VALID);
    }
  }
  if (tid == 0) BG_TL(4);
  // A SHORT launch runs on fewer worker waves (bg_engine_waves): with all envs starting in step, fewer and fuller batches get the
  // stragglers -- the envs with many service steps, which decide when a 20-step launch ends -- through sooner (20 steps: 381 us on
  // seven waves, 330 on four; 200 steps: the other way round).  The surplus waves end here; the workgroup's later barriers count the
  // waves that are left.
  if (wave >= (int)(a.n_waves + a.copier)) return;
  // ---------------------------------------------------------------- the copier wave (packed records)
  // The record copy-out used to close every batch: ~2 700 cycles of list / image reads and a dozen store instructions on the wave that
  // the envs of the batch -- and every env queued behind them -- were waiting for, a sixth of all wave time; without it the same launch
  // ran a third shorter.  Now a worker only QUEUES the finished env (its image is complete and nobody touches it until the env's next
  // step); one wave does nothing but copy images to their rows, lane <-> 16-byte piece, and hands each env back to the run queue (or
  // counts it as through) once its image has been read.  One consumer: no claim, no list, the loop is all loads and stores.
  if (a.copier && wave >= (int)a.n_waves) {
    __builtin_amdgcn_s_setprio(2);
    // SEVERAL copier waves (a.copier = 2 or 3) split the queue into blocks of 32 positions, dealt round robin (no claim)
    const uint32_t cid = (uint32_t)wave - a.n_waves, two = a.copier > 1u ? 1u : 0u;
    uint32_t head = two ? 32u * cid : 0u, cpolls = 0;
    const bool whole = a.obs.row_stride == 384u;   // WHOLE LINES, see below
    const uint32_t PR = 22u, rcp = 2979u;          // dense layout: pieces per record; q / 22 = (q * 2979) >> 16 (exact below 8 000)
    for (;;) {
      const uint32_t tail = __builtin_amdgcn_readfirstlane(bg_lds_ld(&s_cqt));
      uint32_t navail = (int32_t)(tail - head) > 0 ? tail - head : 0u;
      if (two) { const uint32_t left = 32u - (head & 31u); navail = navail > left ? left : navail; }   // the rest of this wave's block
      if (navail == 0u) {
        if (__builtin_amdgcn_readfirstlane(bg_lds_ld(&s_done)) >= (uint32_t)n_live) break;
        __builtin_amdgcn_s_sleep(2);
        if (++cpolls > BG_SPIN_LIMIT) { if (lane == 0) atomicOr(d.err, BG_DEVERR_SPIN); break; }
        continue;
      }
      cpolls = 0;
      const uint32_t nb = navail > BG_BLOCK ? BG_BLOCK : navail;
      // this lane's entry (a producer bumps the tail first and writes the entry next: poll until the generation matches)
      uint2 ent = make_uint2(0u, 0u);
      if ((uint32_t)lane < nb) {
        const uint32_t pos = head + (uint32_t)lane, want = BG_ITEM_VALID | (((pos >> BG_ENG_LNE) & 0x7fu) << 16);
        lds_u32* ep = (lds_u32*)&s_cq[pos & (NE - 1)];
        uint32_t spin = 0, y = ep[1];
        while ((y & (BG_ITEM_VALID | 0x7f0000u)) != want && ++spin < BG_SPIN_LIMIT) { __builtin_amdgcn_s_sleep(1); asm volatile("" ::: "memory"); y = ep[1]; }
        asm volatile("" ::: "memory");
        if ((y & (BG_ITEM_VALID | 0x7f0000u)) != want) atomicOr(d.err, BG_DEVERR_SPIN);   // (the launch is lost: sticky error; this lane's entry stays 0 = through-less, row 0)
        else ent = make_uint2(ep[0], y);
      }
      BG_WAVE_SYNC();
      // WHOLE LINES.  The HBM takes scattered 352-byte records (22 pieces: two partial 128-byte lines each) at 2.6 TB/s, and scattered
      // 384-byte records whose three lines are written completely at 4.5 TB/s of the same payload (tools/micro/recwrite.hip,
      // profiles/r03_recwrite.txt): a record stride of 384 makes the copier write 24 pieces per record -- the image's 22 and two of
      // zeros.  Any other stride: the 22 pieces of the dense layout.
      // (Entries of records beyond nb are read but never used: the ring always holds 256 readable entries.)
      typedef __attribute__((address_space(3))) const char lds_cc;
      typedef uint32_t bg_u32x2 __attribute__((ext_vector_type(2)));
      if (whole) {
        // 24 pieces per record: EIGHT records are exactly three rounds of the wave, so which record of its group of eight and which piece
        // a lane handles in round j is a per-lane CONSTANT (no division, no selects in the loop); the padding pieces read a zero piece.
        // Four groups (32 records) per iteration: the 12 entry reads, the 12 image reads and the 12 stores are each issued back to back.
        uint32_t rsel[3], mul[3], ib[3], gl[3];
        lds_cc* const imgb = (lds_cc*)&s_img[0][0];
        lds_cc* const cqb = (lds_cc*)&s_cq[0];
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const uint32_t pidx = (uint32_t)j * BG_BLOCK + (uint32_t)lane, rs = (pidx * 2731u) >> 16, c = pidx - 24u * rs;
          rsel[j] = rs; gl[j] = 16u * c;
          mul[j] = c < 22u ? 16u * BG_ENG_IMG_PIECES : 0u;                     // image byte address = l * mul + ib
          ib[j] = c < 22u ? 16u * c : (uint32_t)((lds_cc*)&s_zero - imgb);
        }
        for (uint32_t g0 = 0; g0 < nb; g0 += 32u) {
          bg_u32x2 ce[4][3];
          bg_u32x4 v[4][3];
#pragma unroll
          for (int u = 0; u < 4; u++)
#pragma unroll
            for (int j = 0; j < 3; j++)
              ce[u][j] = *(__attribute__((address_space(3))) const bg_u32x2*)(cqb + (((head + g0 + 8u * (uint32_t)u + rsel[j]) & (NE - 1)) << 3));
#pragma unroll
          for (int u = 0; u < 4; u++)
#pragma unroll
            for (int j = 0; j < 3; j++)
              v[u][j] = *(__attribute__((address_space(3))) const bg_u32x4*)(imgb + __umul24(ce[u][j].y & 0xffu, mul[j]) + ib[j]);
#pragma unroll
          for (int u = 0; u < 4; u++)
#pragma unroll
            for (int j = 0; j < 3; j++)
              // NON-TEMPORAL stores: the records are a write-once stream nothing on the GPU reads back; written through the L2 as ordinary
              // stores they evicted the env state, the RNG rings and the shop streams every service step reads (+15 % with the `nt` bit)
              if (g0 + 8u * (uint32_t)u + rsel[j] < nb)
                __builtin_nontemporal_store(v[u][j], (__attribute__((address_space(1))) bg_u32x4*)(a.obs.rows + ((size_t)ce[u][j].x * 384u + gl[j])));
        }
      } else {
      const uint32_t total = PR * nb;
      // the dense layout (or any other stride): KU pieces per lane and iteration, the record of a piece by a division
#define BG_COPIER_KU 12
      constexpr int KU = BG_COPIER_KU;
      for (uint32_t q0 = (uint32_t)lane; q0 < total; q0 += (uint32_t)KU * BG_BLOCK) {
        uint2 ce[KU];
        uint32_t cpc[KU];
        bg_u32x4 v[KU];
#pragma unroll
        for (int k = 0; k < KU; k++) {
          const uint32_t q = q0 + (uint32_t)k * BG_BLOCK;
          const uint32_t r = (q * rcp) >> 16;
          cpc[k] = q - PR * r;
          ce[k] = s_cq[(head + (r < nb ? r : 0u)) & (NE - 1)];
        }
#pragma unroll
        for (int k = 0; k < KU; k++) v[k] = s_img[ce[k].y & 0xffu][cpc[k]];
#pragma unroll
        for (int k = 0; k < KU; k++)
          if (q0 + (uint32_t)k * BG_BLOCK < total)
            __builtin_nontemporal_store(v[k], (__attribute__((address_space(1))) bg_u32x4*)(a.obs.rows + (size_t)ce[k].x * (size_t)a.obs.row_stride + 16u * cpc[k]));
      }
      }
      BG_WAVE_SYNC();   // every image of the batch has been read (LDS operations of a wave complete in order): the envs may move on
      const bool valid = (uint32_t)lane < nb, through = valid && ((ent.y >> 8) & 1u);
      const unsigned long long rm = __ballot(valid && !through), dm = __ballot(through);
      if (rm) {
        uint32_t base = 0;
        if (lane == (int)(__ffsll((long long)rm) - 1)) base = __hip_atomic_fetch_add(&s_tail[BG_Q_RUN], (uint32_t)__popcll(rm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base = __builtin_amdgcn_readlane(base, __ffsll((long long)rm) - 1);
        if (valid && !through) {
          const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(rm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rm, 0u));
          bg_lds_st(&s_q[BG_Q_RUN][slot & (NE - 1)], (ent.y & 0xffu) | (((slot >> BG_ENG_LNE) & 0xffu) << 8) | BG_ITEM_VALID);
        }
      }
      if (dm && lane == 0) __hip_atomic_fetch_add(&s_done, (uint32_t)__popcll(dm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      head += nb;
      if (two && (head & 31u) == 0u) head += 32u * (a.copier - 1u);   // the next blocks are the other copiers'
    }
    if (lane == 0 && wave == (int)a.n_waves) BG_TL(11);
    BG_PROBE_FLUSH(d);
    __syncthreads();   // (the workers' epilogue barrier)
    return;
  }
  // ---------------------------------------------------------------- worker loop
  uint64_t n_steps = 0, n_eps = 0, n_plays = 0, rbits = 0, ohash = 0;
  int64_t ssum = 0;
  const uint32_t bmod3 = (uint32_t)(a.env_index0 % 3ull);
  uint32_t polls = 0;
  bool tl_first = kEngTl && wave == 4;   // (development: the first wave that owns no env thread -- when does it reach the loop, when does it get its first batch)
  if (tl_first && lane == 0) BG_TL(24);
  uint32_t tl_polls = 0; bool tl_seen = false;
  for (;;) {
    // -- pick a queue: the eight control words with two 16-byte loads (one LDS round trip; each word is written atomically by its
    // owner, so a torn pair of words is no worse than two separate loads)
    // The HEADS are read first, the tails after them (a wave's LDS loads execute in order): a tail read BEFORE its head could be older
    // than the head, the difference wraps to 4 billion "items", and a wave that claims them sends the head past the tail for good.
    asm volatile("" ::: "memory");
    const bg_u32x4 ch = *(volatile __attribute__((address_space(3))) bg_u32x4*)&s_ctl[4];
    asm volatile("" ::: "memory");
    const bg_u32x4 ct = *(volatile __attribute__((address_space(3))) bg_u32x4*)&s_ctl[0];
    asm volatile("" ::: "memory");
    const uint32_t hr = __builtin_amdgcn_readfirstlane(ch.x), hp = __builtin_amdgcn_readfirstlane(ch.y), ho = __builtin_amdgcn_readfirstlane(ch.z);
    auto queued = [&](uint32_t tail, uint32_t head) -> uint32_t { const uint32_t k = tail - head; return k <= (uint32_t)NE ? k : 0u; }; // (never more than the envs there are)
    const uint32_t nr = queued(__builtin_amdgcn_readfirstlane(ct.x), hr);
    const uint32_t np = can_serve ? queued(__builtin_amdgcn_readfirstlane(ct.y), hp) : 0u, no = can_serve ? queued(__builtin_amdgcn_readfirstlane(ct.z), ho) : 0u;
    int cls = -1;
    if (np >= a.th_play) cls = BG_Q_PLAY;          // a service-capable wave serves first: the long chains are the critical path
    else if (no >= a.th_other) cls = BG_Q_OTHER;
    else if (nr >= a.th_run) cls = BG_Q_RUN;
    else if (nr | np | no) {
      // nothing is full.  While other waves are inside batches their envs will be back in a moment, so a small batch now only
      // costs instructions at a low lane count; when no wave is busy nothing will ever arrive: take what there is.
      const uint32_t busy = __builtin_amdgcn_readfirstlane(ch.w);
      const uint32_t need = busy ? a.th_part : 1u; // (service queues only: a partial run batch is cheap)
      if (nr && nr >= np && nr >= no) cls = BG_Q_RUN;
      else if (np >= need && np >= no) cls = BG_Q_PLAY;
      else if (no >= need) cls = BG_Q_OTHER;
      else if (nr) cls = BG_Q_RUN;
      else if (np >= need) cls = BG_Q_PLAY;
    }
    if constexpr (kEngTl) if (tl_first && tl_polls == 0 && !tl_seen) { tl_seen = true; if (lane == 0) { BG_TL(30); s_tl[31] += np + 1000ull * nr + 1000000ull * no; } }
    if (cls < 0) {
      if (__builtin_amdgcn_readfirstlane(ct.w) >= (uint32_t)n_live) break; // every env has done its T steps
      __builtin_amdgcn_s_sleep(8);
      tl_polls++;
      if (++polls > BG_SPIN_LIMIT) { if (lane == 0) atomicOr(d.err, BG_DEVERR_SPIN); break; }
      continue;
    }
    const uint32_t head = cls == BG_Q_RUN ? hr : (cls == BG_Q_PLAY ? hp : ho), navail = cls == BG_Q_RUN ? nr : (cls == BG_Q_PLAY ? np : no);
    const uint32_t nb = navail > BG_BLOCK ? BG_BLOCK : navail;
    // the claim: one compare-and-swap on the queue head by lane 0; the slots of the batch are read BEFORE it in the same group of LDS
    // operations (a read of a slot this wave does not get is harmless), so a successful claim is one LDS round trip, not two.  An item
    // carries the GENERATION of its ring position ((position >> 8) & 255): a slot still holding the item of 256 positions ago is
    // "not written yet", never a second copy of an old env -- so slots are not cleared after reading.  (A producer cannot lap an unread
    // slot in practice: the reader's load precedes its claim; only a reader that found the slot not yet written polls it, and it would have
    // to miss the item for the time of 256 further pushes.)
    uint32_t item = 0;
    uint32_t* const slotp = &s_q[cls][(head + (uint32_t)lane) & (NE - 1)];
    {
      uint32_t got = 0;
      if ((uint32_t)lane < nb) item = bg_lds_ld(slotp);
      if (lane == 0) got = atomicCAS(&s_head[cls], head, head + nb) == head ? 1u : 0u;
      if (lane == 0 && got) __hip_atomic_fetch_add(&s_busy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (__builtin_amdgcn_readfirstlane(got) == 0u) {
        continue;
      }
    }
    polls = 0;
    const unsigned long long tl_b0 = kEngTl ? wall_clock64() : 0ull;
    if constexpr (kEngTl) if (tl_first) {
      if (lane == 0) { BG_TL(25); s_tl[26] += tl_polls; s_tl[27 + (cls == BG_Q_RUN ? 0 : (cls == BG_Q_PLAY ? 1 : 2))] += 1ull; }
      tl_first = false;
    }
    // the service chains are the critical path of every env they hold: they issue ahead of the run batches on their SIMD
    if (cls == BG_Q_RUN) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3);
    // A batch = the step the envs were queued for, then up to th_more further CHEAP steps of the same envs (a toggle is followed
    // by another toggle five times out of six): the env stays with the wave that has it instead of going through the run queue
    // -- and its wait for a free wave -- once per step.  A lane leaves the batch when its env has done its T steps or its next
    // action needs a service batch.
    bool active = false;           // this lane holds an env
    int l = 0, env = 0;
    uint32_t t = 0;
    if ((uint32_t)lane < nb) {
      uint32_t spin = 0;
      const uint32_t want = BG_ITEM_VALID | ((((head + (uint32_t)lane) >> BG_ENG_LNE) & 0xffu) << 8);
      while ((item & (BG_ITEM_VALID | 0xff00u)) != want && ++spin < BG_SPIN_LIMIT) { __builtin_amdgcn_s_sleep(1); item = bg_lds_ld(slotp); }
      if ((item & (BG_ITEM_VALID | 0xff00u)) != want) atomicOr(d.err, BG_DEVERR_SPIN);
      else { active = true; l = (int)(item & 0xffu); env = env0 + l; t = s_t[l]; }
    }
    // one CHEAP step of this lane's env on its image: a card-select toggle, shop end, or an action the guards reject; anything
    // else is queued for a service batch and the lane gives the env up.  Returns true when a step was completed.
    auto cheap_step = [&](int& action, double& reward, StepOut& o) __attribute__((always_inline)) -> bool {
      bool fin = false;
      uint64_t mask = s_mask[l];
      lds_u32* img32 = (lds_u32*)&s_img[l][0];
      lds_u8* img8 = (lds_u8*)&s_img[l][0];
      const uint4 c3 = s_c34[0][l];
      const uint32_t ante = bg_b(c3.x, 0), phase = bg_b(c3.x, 2), discards_left = bg_b(c3.y, 0), nsel0 = bg_b(c3.y, 3);
      if (a.actions_in) action = a.actions_in[(size_t)t * N + env];
      else {
        const uint64_t gi = a.env_index0 + (uint64_t)env;
        Env pe; pe.phase = (int)phase; // the policy only looks at the phase and the mask
        PolicyLane pl;
        pl.seed_env = a.policy_seed + 0x9E3779B97F4A7C15ull * (gi + 1);
        pl.blind = a.policy == 2 ? 45 + (int)((bmod3 + (uint32_t)env % 3u) % 3u) : 45;
        action = bg_policy_action_fast(pe, mask, a.policy, pl, pl.seed_env + BG_POLICY_PSI * (a.t0 + (uint64_t)t + 1), (lds_JTables*)&jt);
      }
      const int64_t chips_scored = (int64_t)(((uint64_t)img32[33] << 32) | img32[32]);
      const bool valid = action >= 0 && action < 60 && ((mask >> (action & 63)) & 1ull);
      const bool terminal = ante > 100u || chips_scored > 1000000000ll;   // :619-623, settled by a service batch (it resets the env)
      if (!terminal && valid && phase == 0u && action >= 2 && action < 10) {
        // :1052-1058 toggle position `pos` in state.selected_cards (bg_toggle_select on chunk 4)
        const int pos = action - 2;
        uint4 c4 = s_c34[1][l];
        Env te; te.sel = ((uint64_t)c4.w << 32) | c4.z; te.nsel = (int)nsel0;
        bg_toggle_select(te, pos);
        c4.z = (uint32_t)te.sel; c4.w = (uint32_t)(te.sel >> 32);
        s_c34[1][l] = c4;
        ((lds_u8*)&s_c34[0][l])[7] = (uint8_t)te.nsel;                   // chunk 3, word y, byte 3
        *(lds_u64*)&img32[2 * pos] = te.nsel > (int)nsel0 ? 1ull : 0ull;  // selected_cards[pos] (int64)
        if ((te.nsel > 0) != (nsel0 > 0u)) {                             // PLAY_HAND / DISCARD availability (:1436-1441)
          const uint32_t play = te.nsel > 0 ? 1u : 0u, disc = (te.nsel > 0 && discards_left > 0u) ? 1u : 0u;
          mask = (mask & ~3ull) | play | ((uint64_t)disc << 1);
          s_mask[l] = mask;
          *(__attribute__((address_space(3))) uint16_t*)&img8[BG_ROW_ACTION_MASK] = (uint16_t)(play | (disc << 8));
        }
        fin = true;
      } else if (!terminal && valid && phase == 1u && action == 31 && (int)(int8_t)bg_b(c3.y, 1) <= (int)bg_b(c3.y, 2)) {
        // :1247-1251 leave the shop; the hand is full (played cards never left it: `_draw_cards` would draw nothing), so all
        // that changes is the phase, the mask and the shop rows of the observation (shown in SHOP phase only, :1534-1539)
        const uint32_t nhand = bg_b(c3.y, 2), ncons = bg_b(c3.z, 1);
        ((lds_u8*)&s_c34[0][l])[2] = 0;                                   // chunk 3, word x, byte 2: phase = PLAY
        uint64_t m = (((1ull << (nhand < 8u ? nhand : 8u)) - 1ull) << 2) | (((1ull << ncons) - 1ull) << 10);
        if (nsel0 > 0u) m |= 1ull | (discards_left > 0u ? 2ull : 0ull);
        mask = m;
        s_mask[l] = mask;
#pragma unroll
        for (int wq = 0; wq < 15; wq++) img32[44 + wq] = __umul24((uint32_t)(mask >> (4 * wq)) & 0xfu, 0x204081u) & 0x01010101u; // action_mask i8[60]
#pragma unroll
        for (int wq = 64; wq < 74; wq++) img32[wq] = 0u;                  // shop_items, shop_costs
        img8[BG_ROW_PHASE] = 0;
        fin = true;
      } else if (!terminal && !valid) { reward = -1.0; fin = true; }    // :626-627 'Invalid action': nothing changes
      else {
        const int q = (!terminal && phase == 0u && action == 0) ? BG_Q_PLAY : BG_Q_OTHER;
        const uint32_t slot = __hip_atomic_fetch_add(&s_tail[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        bg_lds_st(&s_q[q][slot & (NE - 1)], (uint32_t)l | (((slot >> BG_ENG_LNE) & 0xffu) << 8) | (((uint32_t)action & 0x7fffu) << 16) | BG_ITEM_VALID);
        active = false;                                                   // the env is the service queue's now
      }
      if (fin) {
        *(__attribute__((address_space(3))) double*)&img32[34] = reward;  // BG_ROW_REWARD
        img32[43] = (uint32_t)action;                                      // BG_ROW_ACTION
        img8[BG_ROW_TERMINATED] = 0;
        if constexpr (INFO) { bg_step_init(o); o.reward = reward; o.error = valid ? 0 : 1; }
      }
      return fin;
    };
    // the outputs of a completed step other than the image, and the env's step counter
    auto finish = [&](bool cheap, size_t row, int action, double reward, bool terminated, const StepOut& o) __attribute__((always_inline)) {
      if (!a.obs.rows && !keys_at_end) bg_emit_keys_from_image((const lds_u4*)&s_img[l][0], a.obs, row); // per-key arrays, every step kept ([T, N] buffers)
      if (a.reward) a.reward[row] = reward;
      if (a.term) a.term[row] = terminated ? 1 : 0;
      if (a.actions_out) a.actions_out[row] = action;
      if constexpr (INFO) bg_emit_info(row, o, a.trunc, a.info);
      if (HASH) ohash ^= bg_hash_image((const lds_u4*)&s_img[l][0]) * (0x9E3779B97F4A7C15ull + 2 * (uint64_t)(a.t0 + t)) + (a.env_index0 + (uint64_t)env);
      n_steps++;
      rbits ^= (uint64_t)__double_as_longlong(reward) * (2 * (uint64_t)(a.t0 + t) + 1);
      t++;
      s_t[l] = t;
    };
    // copy-out: the images of the finished envs to their rows, lane <-> piece (whole 352-byte runs per row); then envs that are
    // through leave (after their images have been READ)
    auto copy_out = [&](bool fin, size_t row) __attribute__((always_inline)) {
    if (a.copier) {
      // packed records: the finished envs go to the copier wave, which writes their records and hands them back (or counts them as through)
      const unsigned long long fm = __ballot(fin);
      if (fm) {
        uint32_t base = 0;
        if (lane == (int)(__ffsll((long long)fm) - 1)) base = __hip_atomic_fetch_add(&s_cqt, (uint32_t)__popcll(fm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base = __builtin_amdgcn_readlane(base, __ffsll((long long)fm) - 1);
        if (fin) {
          const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
          lds_u32* ep = (lds_u32*)&s_cq[pos & (NE - 1)];
          ep[0] = (uint32_t)row;
          asm volatile("" ::: "memory");
          ep[1] = (uint32_t)l | (t >= (uint32_t)a.T ? 0x100u : 0u) | (((pos >> BG_ENG_LNE) & 0x7fu) << 16) | BG_ITEM_VALID;
          asm volatile("" ::: "memory");
          active = false;
        }
      }
      return;
    }
      if (active && t >= (uint32_t)a.T) { __hip_atomic_fetch_add(&s_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); active = false; }
    };
    // ---------------- the step the batch was claimed for
    {
      bool fin = false;
      size_t row = 0;
      if (active) {
        row = (size_t)env + (a.obs_stride_steps ? (size_t)t * N : 0);
        int action = 0;
        double reward = 0.0;
        bool terminated = false;
        StepOut o;
        if (cls == BG_Q_RUN) fin = cheap_step(action, reward, o);
        else {
          uint64_t mask = s_mask[l];
        action = a.actions_in ? a.actions_in[(size_t)t * N + env] : (int)((item >> 16) & 0x7fffu);
        BG_PROBE_BEGIN();
        uint4 c[BG_NHOT];
#pragma unroll
        for (int k = 0; k < BG_NHOT; k++) if (k != 3 && k != 4) c[k] = d.hot[(size_t)k * N + env];
        c[3] = s_c34[0][l]; c[4] = s_c34[1][l];
        Env e;
        bg_unpack(c, e);
        BG_PROBE(23);
        bg_derive_ready(e, s_prod[l]);
        DeckT dk; dk.col = (lds_u32*)&s_deck[0][l];
        ShopRegs sr; sr.valid = false;
        RngWin w;
        bg_win_init(w, &s_win[serve_idx][0][lane], &jt);
        bg_step_init(o);
        if constexpr (INFO) if (a.info.score_breakdown) o.bd_dst = a.info.score_breakdown + row * 8;
        if (bg_step_guards(e, mask, action, o)) bg_env_dispatch(d, env, e, w, sr, dk, action, o);
        BG_PROBE(cls == BG_Q_PLAY ? 20 : 21);
        if (e.max_ante > 0 && e.ante > e.max_ante) { o.terminated = true; o.flags |= 256; }
        if (o.terminated) n_eps++;
        if (o.terminated && a.autoreset) { bg_env_reset(d, env, e, dk); if (INFO) o.flags |= BG_INFO_AUTORESET; }
        if constexpr (CARDS) bg_vm_drain(); // card states / lazy streams in HBM are edited from any service wave: let the stores land
        BG_PROBE(24);
        mask = bg_action_mask(d, env, e, sr);
        BG_PROBE(25);
        {
          const ObsPtrs none{};   // (the image only -- per-key arrays are written from it, by finish() or behind the loop; bg_engine3.h's service step has the note on what passing a.obs costs in scalar registers)
          bg_write_obs_impl<false, 3>(d, env, row, e, dk, none, mask, sr, RowExtra{o.reward, action, o.terminated ? 1u : 0u}, RowStage{(lds_u4*)&s_img[l][0], nullptr});
        }
        BG_PROBE(26);
        bg_pack(e, c);
#pragma unroll
        for (int k = 0; k < BG_NHOT; k++) if (k != 3 && k != 4) d.hot[(size_t)k * N + env] = c[k];
        s_c34[0][l] = c[3]; s_c34[1][l] = c[4];
        s_mask[l] = mask;
        BG_PROBE(27);
        reward = o.reward; terminated = o.terminated;
        if (o.hand_type >= 0) { n_plays++; ssum += o.final_score; }
        fin = true;
        }
        if (fin) finish(cls == BG_Q_RUN, row, action, reward, terminated, o);
      }
      copy_out(fin, row);
    }
    // ---------------- further cheap steps of the envs this wave still holds
    // (run batches only: a service wave hands its envs back at once -- service capacity is what the whole workgroup waits for)
    for (uint32_t sub = 0; cls == BG_Q_RUN && !a.copier && sub < a.th_more && __ballot(active) != 0ull; sub++) {
      bool fin = false;
      size_t row = 0;
      if (active) {
        row = (size_t)env + (a.obs_stride_steps ? (size_t)t * N : 0);
        int action = 0;
        double reward = 0.0;
        StepOut o;
        fin = cheap_step(action, reward, o);
        if (fin) finish(true, row, action, reward, false, o);
      }
      copy_out(fin, row);
    }
    // ---- hand the remaining envs back to the run queue
    if (active) {
      const uint32_t slot = __hip_atomic_fetch_add(&s_tail[BG_Q_RUN], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      bg_lds_st(&s_q[BG_Q_RUN][slot & (NE - 1)], (uint32_t)l | (((slot >> BG_ENG_LNE) & 0xffu) << 8) | BG_ITEM_VALID);
    }
    if (lane == 0) __hip_atomic_fetch_sub(&s_busy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // after the pushes of this batch
    if constexpr (kEngTl) if (lane == 0) {
      const int b = cls == BG_Q_RUN ? 5 : (cls == BG_Q_PLAY ? 16 : 20);   // start (since the workgroup's), count, duration, lanes
      __hip_atomic_fetch_add(&s_tl[b], tl_b0 - tl_k0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&s_tl[b + 1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&s_tl[b + 2], wall_clock64() - tl_b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&s_tl[b + 3], (unsigned long long)nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  // ---------------------------------------------------------------- epilogue: chunks 3 / 4 -> HBM, statistics
  if (lane == 0 && wave == 0) BG_TL(9);
  if (lane == 0 && wave == (int)a.n_waves - 1) BG_TL(10);
  BG_PROBE_FLUSH(d);
  __syncthreads();
  if (tid == 0) BG_TL(12);
  if (tid < n_live) {
    d.hot[(size_t)3 * N + env0 + tid] = s_c34[0][tid];
    d.hot[(size_t)4 * N + env0 + tid] = s_c34[1][tid];
    if (keys_at_end) bg_emit_keys_from_image((const lds_u4*)&s_img[tid][0], a.obs, (size_t)(env0 + tid));
  }
  if (a.stats) {   // (bg_step.h: one set of global atomics per workgroup, not per wave)
    bg_stats_wave(s_stats, n_steps, n_eps, n_plays, ssum, rbits, ohash);
    __syncthreads();
    bg_stats_flush(a.stats, s_stats, tid);
  }
  if constexpr (kEngTl) if (d.dbg) {
    if (tid == 0) { BG_TL(13); s_tl[0] = 1ull; }
    __syncthreads();
    if (tid < 32 && tid != 14 && tid != 15 && s_tl[tid]) atomicAdd(&d.dbg[tid], s_tl[tid]);
    if (tid == 14) atomicMax(&d.dbg[14], ~tl_k0);
    if (tid == 15) atomicMax(&d.dbg[15], wall_clock64());
  }
}
