// bg_engine2.h -- the step engine as TWO cooperating kernels (packed-record rollouts): OWNER waves and a chip-wide pool of SERVICE waves.
//
// Why two kernels.  The one-kernel engine (bg_engine.h) is bound by the latency of a step per env -- 256 envs per CU, each taking its T
// steps one after the other -- and its 256 VGPRs per wave, set by the PLAY_HAND path, allow seven waves per CU: a card-select toggle (84 %
// of a random policy's steps, ~150 instructions on ~40 registers) waits for one of five 256-register workers, and a play batch can only
// draw on its own CU's 256 envs (26 of 64 lanes).  Here
//   * bg_owner_kernel: lane = env for the whole launch (a workgroup = 256 envs = four waves of <= 128 VGPRs).  Every iteration a wave
//     settles the CHEAP step of each of its ready envs on the env's record image in LDS (toggle / shop end / rejected action: exactly the
//     run batch of bg_engine.h, but with chunks 3 / 4 and the action mask in registers and no queue, no claim, no hand-over), copies the
//     finished records out (lane <-> 16-byte piece, whole 128-byte lines, non-temporal) and posts the other actions -- PLAY_HAND, DISCARD,
//     blind select, shop buy / reroll / sell, consumables, terminal guards -- as REQUESTS to global-memory queues.  Nothing in the wave's
//     loop touches another wave's LDS: there is no barrier and no LDS queue.
//   * bg_service_kernel: one-wave workgroups of 256 VGPRs placed beside the owners (one per SIMD).  A service wave claims up to 64
//     requests of one class from the queues of ITS XCD -- every CU of the XCD feeds them, so batches are full --, runs the step exactly as
//     a service batch of bg_engine.h does (same device functions), writes the state back, writes the finished 352-byte record STRAIGHT TO
//     ITS ROW (staged through LDS: whole lines) and answers.  The owner reads the record back from the row into its LDS image (the row IS
//     the mailbox), so a service step's record is written once.
// Hand-over (MI355X_MICROARCH.md, "inter-workgroup visibility"; cdna_hip_programming.md guideline 16): a CU's vector L1 is never refreshed
// by another CU's stores and the eight XCD L2s are not coherent with each other.  A request is served on the XCD that posted it -- both
// sides read HW_REG_XCC_ID, a fact about where the wave runs, not an assumption about dispatch order -- so every byte of an env's state
// is written and read through ONE L2 for the length of a launch (kernel boundaries make it coherent between launches, as for any two
// kernels).  Payload: plain stores (they stay in that L2), then `s_waitcnt vmcnt(0)` in the storing wave, then the flag / queue entry as an
// agent-scope atomic; the reader polls with agent-scope loads and reads the payload with agent-scope (sc1) loads, which bypass its L1.
// Every wait is bounded by the wall clock (sticky BG_DEVERR_SPIN instead of a hang).
// Both kernels must be RESIDENT together: the host launches them on two streams (bg_engine2_launch); neither ever waits for a workgroup
// that has not started (an owner that is not resident posts nothing; a service wave serves whoever posts).
#pragma once

#define BG_E2_NSUB 2                       // queues per (XCD, class): one returning atomic per owner wave, iteration and class; a word takes ~90 / us
#define BG_E2_NQX (2 * BG_E2_NSUB)         // queues per XCD: [class][sub]
#define BG_E2_NQ (8 * BG_E2_NQX)
#define BG_E2_TIMEOUT 200000000ull         // 2 s of the 100 MHz wall clock: a wait that long is a bug (or a missing partner kernel)
struct E2Queue { uint32_t tail; uint32_t pad0[31]; uint32_t head; uint32_t pad1[31]; }; // each word on its own 128-byte line
struct E2Ctl {
  uint32_t owners_done; uint32_t pad0[31];  // owner waves that have finished, ever (monotonic; only ever touched by atomics)
  uint32_t done_flag; uint32_t padf[31];    // = E2Args.done_target once the last owner wave of a launch has finished: a STORE, which the service waves' loads see
  uint32_t svc_seen[8]; uint32_t pad1[24];  // service waves started per XCD, ever (diagnostic)
  unsigned long long svc_stat[8]; uint32_t pad2[16]; // [0] batches [1] requests [2] first owner wave started [3] last owner wave ended [4] first service wave started [5] last service wave ended (wall clock; development)
  E2Queue q[BG_E2_NQ];
  // development: per XCD -- service waves that have left, OR of the reasons (1 done, 2 idle time-out), the largest owners_done / smallest target a leaving
  // wave saw, wall clock of the first / last leave, batches served
  uint32_t x_exits[8], x_reason[8], x_done[8], x_target[8]; unsigned long long x_t0[8], x_t1[8], x_batches[8];
  uint32_t x_migr[8];   // [0] owner waves / [1] service waves whose XCC id changed while they ran
  uint32_t owners_started; uint32_t pad3[31];
  unsigned long long tm[32];   // development (-DBG_E2_TIMING): cycle sums, see tools/e2_diag.py   // owner workgroups that have started, ever (the gate in front of the service kernel waits for it)
};
struct E2Args {
  E2Ctl* ctl;
  unsigned long long* ring;      // [BG_E2_NQ][1 << ring_log] entries: env | (action | valid << 8 | t << 9 | generation << 25 | 1 << 31) << 32
  uint32_t ring_log;
  uint32_t* ansq;                // [N] answers: the number of steps the env has completed in this launch once its request has been served (the mask rides in the record)
  uint4* img;                    // [N][24] record images between launches (22 pieces used)
  unsigned long long* imask;     // [N] action masks between launches
  uint32_t done_target;          // ctl->owners_done at which this launch's service waves retire
  uint32_t fill_wait;            // wall-clock ticks (10 ns) a service wave waits for a fuller batch once it has seen a request
  uint32_t max_batch;            // <= 64
  uint32_t start_target;         // ctl->owners_started at which the gate lets the service kernel start
  uint32_t zero;                 // 0, as a run-time value: `fetch_add(p, zero)` stays a returning atomic (the compiler turns `fetch_add(p, 0)` into a load)
};

__device__ __forceinline__ uint32_t bg_xcc_id() { uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 7u; }

#ifdef BG_E2_TIMING
#define E2T_DECL() unsigned long long e2t_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, e2t0_ = __builtin_readcyclecounter()
#define E2T(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); e2t_[k] += n_ - e2t0_; e2t0_ = n_; } while (0)
#define E2T_CNT(k, v) do { e2t_[k] += (v); } while (0)
#define E2T_FLUSH(base) do { if ((threadIdx.x & 63) == 0) for (int k_ = 0; k_ < 12; k_++) atomicAdd(&x.ctl->tm[(base) + k_], e2t_[k_]); } while (0)
#else
#define E2T_DECL() do {} while (0)
#define E2T(k) do {} while (0)
#define E2T_CNT(k, v) do {} while (0)
#define E2T_FLUSH(base) do {} while (0)
#endif
// ------------------------------------------------------------------------------------------------------------------------------
// images between launches: the packed record of every env + its action mask, rebuilt from the state whenever anything but the
// two-kernel engine has touched it (bg_reset, bg_step, bg_inject, ...)
// ------------------------------------------------------------------------------------------------------------------------------
template <bool CARDS>
__global__ __launch_bounds__(BG_BLOCK) void bg_e2_image_kernel(BgDev d, E2Args x) {
  using DeckT = typename std::conditional<CARDS, Deck0C, Deck0>::type;
  const int env = blockIdx.x * BG_BLOCK + threadIdx.x;
  if (env >= d.N) return;
  Env e;
  bg_load_env(d, env, e);
  DeckT dk; static_cast<Deck0&>(dk) = bg_load_deck0(d, env);
  ShopRegs sr; sr.valid = false;
  const uint64_t mask = bg_action_mask(d, env, e, sr);
  ObsPtrs o; memset(&o, 0, sizeof(o));
  o.rows = (uint8_t*)x.img; o.row_stride = 384u;
  bg_write_obs<false>(d, env, (size_t)env, e, dk, o, mask, sr, RowExtra{0.0, 0, 0u});
  x.imask[env] = mask;
}

// ------------------------------------------------------------------------------------------------------------------------------
// owner kernel
// ------------------------------------------------------------------------------------------------------------------------------
#ifndef BG_E2_AB
#define BG_E2_AB 4                          // answered envs a wave absorbs per iteration (their records come back lane <-> piece, one register each)
#endif
#define BG_E2_OW 4                          // owner waves per workgroup (nothing is shared between them but the tables)
#define BG_E2_IMG_BYTES (BG_E2_OW * BG_BLOCK * 22 * 16)
#define BG_E2_LIST_BYTES (BG_E2_OW * BG_BLOCK * 8)
#define BG_E2_OWNER_LDS (BG_E2_IMG_BYTES + BG_E2_LIST_BYTES + 16 + (int)sizeof(JTables))
template <bool HASH>
__global__ __launch_bounds__(BG_E2_OW * BG_BLOCK) void bg_owner_kernel(BgDev d, EngineArgs a, E2Args x) {
  constexpr int NE = BG_E2_OW * BG_BLOCK;
  // DYNAMIC LDS on purpose: with a static 97 KB the compiler knows that one workgroup fits a CU, derives "one wave per SIMD" from it and PADS the
  // kernel's register allocation to 257 VGPRs so that no second wave of this kernel could ever share a SIMD -- which also locks out the 256-register
  // service waves of the other kernel (found as "the service kernel starts when the owner kernel ends": tools/micro/corun2.hip, tools/e2_diag.py)
  extern __shared__ __attribute__((aligned(16))) unsigned char e2_smem[];
  typedef bg_u32x4 ImgRow[22];
  typedef uint2 ListRow[BG_BLOCK];
  ImgRow* const s_img = (ImgRow*)e2_smem;                                          // [NE][22] record images
  ListRow* const s_list = (ListRow*)(e2_smem + BG_E2_IMG_BYTES);                   // [BG_E2_OW][64] copy-out list of a wave: .x = record row, .y = env lane of the workgroup
  bg_u32x4& s_zero = *(bg_u32x4*)(e2_smem + BG_E2_IMG_BYTES + BG_E2_LIST_BYTES);
  JTables& jt = *(JTables*)(e2_smem + BG_E2_IMG_BYTES + BG_E2_LIST_BYTES + 16);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int env0 = blockIdx.x * NE, env = env0 + tid;
  const size_t N = (size_t)d.N;
  const bool live = env < d.N;
  const uint32_t xcc = bg_xcc_id();
  if (tid == 0) { atomicMax(&x.ctl->svc_stat[7], ~wall_clock64()); __hip_atomic_fetch_add((g_u32*)&x.ctl->owners_started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  typedef __attribute__((address_space(3))) const char lds_cc;
  // ---- prologue: images (the workgroup's envs are one contiguous run of pieces), chunks 3 / 4, mask
  if (tid == 0) s_zero = bg_u32x4{0u, 0u, 0u, 0u};
  {
    const int n_live = d.N - env0 < NE ? d.N - env0 : NE;
    const bg_u32x4* src = (const bg_u32x4*)x.img + (size_t)env0 * 24;
    for (int q = tid; q < n_live * 24; q += NE) {
      const int e = (q * 2731) >> 16, p = q - 24 * e;   // q / 24 (exact below 8 000)
      if (p < 22) s_img[e][p] = src[q];
    }
  }
  uint4 c3 = make_uint4(0, 0, 0, 0), c4 = c3;
  uint64_t mask = 0;
  if (live) {
    c3 = d.hot[3 * N + env]; c4 = d.hot[4 * N + env]; mask = x.imask[env];
    __hip_atomic_store((g_u32*)&x.ansq[env], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the same flavour of store as the answers; it lands before the first request's atomic returns)
  }
  bg_tables_load(&jt, d.jtab);   // (ends with the workgroup's only barrier)
  const uint32_t T = (uint32_t)a.T;
  const uint32_t bmod3 = (uint32_t)(a.env_index0 % 3ull);
  const uint64_t gi = a.env_index0 + (uint64_t)env;
  const uint64_t seed_env = a.policy_seed + 0x9E3779B97F4A7C15ull * (gi + 1);
  const int blind = a.policy == 2 ? 45 + (int)((bmod3 + (uint32_t)env % 3u) % 3u) : 45;
  lds_u32* const img32 = (lds_u32*)&s_img[tid][0];
  lds_u8* const img8 = (lds_u8*)&s_img[tid][0];
  lds_cc* const imgb = (lds_cc*)&s_img[0][0];
  const bool whole = a.obs.row_stride == 384u;
  // per-lane constants of the whole-line copy-out (bg_engine.h, copier): eight records = three rounds of the wave
  uint32_t rsel[3], cmul[3], cib[3], cgl[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const uint32_t pidx = (uint32_t)j * BG_BLOCK + (uint32_t)lane, rs = (pidx * 2731u) >> 16, c = pidx - 24u * rs;
    rsel[j] = rs; cgl[j] = 16u * c;
    cmul[j] = c < 22u ? 16u * 22u : 0u;
    cib[j] = c < 22u ? 16u * c : (uint32_t)((lds_cc*)&s_zero - imgb);
  }
  // ---- the loop, software-pipelined: a hand-over hop costs 1.5 - 2.5 us under load (the price sits in memory, not in instructions), and a
  // wave that waits for each hop in turn -- answer poll, then chunks 3 / 4 and the record, then the queue atomics -- spends 16 k cycles per
  // iteration of which 2.5 k are work (tools/e2_diag.py, -DBG_E2_TIMING).  So every iteration
  //   (1) ISSUES the answer polls of its waiting envs,
  //   (2) settles the cheap steps of its ready envs (posting the others: chunks 3 / 4, then one returning atomic per class),
  //   (3) copies the finished records out,
  //   (4) only then looks at what came back: queue positions -> entries; answers -> the loads of chunks 3 / 4 and of the record are ISSUED
  //       and fly over the whole next iteration; the loads issued one iteration ago -> image, registers, the step is complete.
  // An env's state: 0 ready, 1 request posted (polling), 2 answered (its loads are in flight).
  uint32_t t = 0, st = 0;
  uint64_t n_steps = 0, rbits = 0, ohash = 0;
  unsigned long long last_progress = wall_clock64();
  if (lane == 0) atomicMax(&x.ctl->svc_stat[2], ~last_progress);   // (development: stored complemented so that the zeroed word means 'never')
  __builtin_amdgcn_s_setprio(1);
  // in flight from the previous iteration: chunks 3 / 4 of this lane's env (state 2) and, lane <-> piece, the records of up to four envs of the wave
  bg_u32x4 ab_c3 = bg_u32x4{0u, 0u, 0u, 0u}, ab_c4 = ab_c3, ab_rec[BG_E2_AB];
  int ab_j[BG_E2_AB];
#pragma unroll
  for (int u = 0; u < BG_E2_AB; u++) { ab_rec[u] = ab_c3; ab_j[u] = -1; }
  // a request whose queue position (a returning atomic) is in flight: state 3; its entry is written at the top of the next iteration
  unsigned long long pq_m0 = 0, pq_m1 = 0;
  uint32_t pq_base = 0, pq_hi = 0;
  E2T_DECL();
#ifdef BG_E2_TIMING
  unsigned long long e2_tpost = 0, e2_tans = 0, e2_l0 = 0, e2_l1 = 0, e2_ln = 0;
#endif
  for (;;) {
    if (__ballot(live && (t < T || st != 0u)) == 0ull) break;
    E2T_CNT(10, 1);
    const size_t row = (size_t)env + (a.obs_stride_steps ? (size_t)t * N : 0);
    // ---- (0) the queue positions asked for in the previous iteration -> entries (8-byte agent-scope stores).  The atomic has had the whole
    // copy-out and absorb phase to come back, and the stores of chunks 3 / 4 before it have landed (vector-memory operations complete in order)
    if (pq_m0 | pq_m1) {
      const uint32_t b0 = pq_m0 ? __builtin_amdgcn_readlane(pq_base, __ffsll((long long)pq_m0) - 1) : 0u;
      const uint32_t b1 = pq_m1 ? __builtin_amdgcn_readlane(pq_base, __ffsll((long long)pq_m1) - 1) : 0u;
      if (st == 3u) {
        const bool c1 = ((pq_m1 >> lane) & 1ull) != 0ull;
        const unsigned long long mm = c1 ? pq_m1 : pq_m0;
        const uint32_t pos = (c1 ? b1 : b0) + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
        const uint32_t q = xcc * BG_E2_NQX + ((uint32_t)wave & (BG_E2_NSUB - 1u)) + (c1 ? BG_E2_NSUB : 0u);
        #ifdef BG_E2_TIMING   // (development: bits 24..31 of the env word carry the time of the store in 0.64 us units)
        bg_st8a(&x.ring[((size_t)q << x.ring_log) + (pos & ((1u << x.ring_log) - 1u))], (unsigned long long)((uint32_t)env | ((uint32_t)((wall_clock64() >> 6) & 0xffu) << 24)) | ((unsigned long long)(pq_hi | (((pos >> x.ring_log) & 0x3fu) << 25)) << 32));
#else
        bg_st8a(&x.ring[((size_t)q << x.ring_log) + (pos & ((1u << x.ring_log) - 1u))], (unsigned long long)(uint32_t)env | ((unsigned long long)(pq_hi | (((pos >> x.ring_log) & 0x3fu) << 25)) << 32));
#endif
        st = 1u;
#ifdef BG_E2_TIMING
        e2_tpost = wall_clock64();
#endif
      }
      pq_m0 = 0; pq_m1 = 0;
    }
    // ---- (1) polls
    uint32_t pollv = 0;
    if (st == 1u) pollv = bg_ld4a(&x.ansq[env]);
    E2T(0);
    // ---- (2) the cheap step of every ready env (bg_engine.h: cheap_step)
    bool fin = false, post = false;
    int action = 0, cls = 0;
    uint32_t pvalid = 0;
    double reward = 0.0;
    if (live && st == 0u && t < T) {
      const uint32_t ante = bg_b(c3.x, 0), phase = bg_b(c3.x, 2), discards_left = bg_b(c3.y, 0), nsel0 = bg_b(c3.y, 3);
      {
        Env pe; pe.phase = (int)phase; // the policy only looks at the phase and the mask
        PolicyLane pl; pl.seed_env = seed_env; pl.blind = blind;
        action = bg_policy_action_fast(pe, mask, a.policy, pl, seed_env + BG_POLICY_PSI * (a.t0 + (uint64_t)t + 1), (lds_JTables*)&jt);
      }
      const int64_t chips_scored = (int64_t)(((uint64_t)img32[33] << 32) | img32[32]);
      const bool valid = action >= 0 && action < 60 && ((mask >> (action & 63)) & 1ull);
      const bool terminal = ante > 100u || chips_scored > 1000000000ll;   // :619-623, settled by a service wave (it resets the env)
      if (!terminal && valid && phase == 0u && action >= 2 && action < 10) {
        // :1052-1058 toggle position `pos` in state.selected_cards (bg_toggle_select on chunk 4)
        const int pos = action - 2;
        Env te; te.sel = ((uint64_t)c4.w << 32) | c4.z; te.nsel = (int)nsel0;
        bg_toggle_select(te, pos);
        c4.z = (uint32_t)te.sel; c4.w = (uint32_t)(te.sel >> 32);
        c3.y = (c3.y & 0x00ffffffu) | ((uint32_t)te.nsel << 24);
        *(lds_u64*)&img32[2 * pos] = te.nsel > (int)nsel0 ? 1ull : 0ull;  // selected_cards[pos] (int64)
        if ((te.nsel > 0) != (nsel0 > 0u)) {                             // PLAY_HAND / DISCARD availability (:1436-1441)
          const uint32_t play = te.nsel > 0 ? 1u : 0u, disc = (te.nsel > 0 && discards_left > 0u) ? 1u : 0u;
          mask = (mask & ~3ull) | play | ((uint64_t)disc << 1);
          *(__attribute__((address_space(3))) uint16_t*)&img8[BG_ROW_ACTION_MASK] = (uint16_t)(play | (disc << 8));
        }
        fin = true;
      } else if (!terminal && valid && phase == 1u && action == 31 && (int)(int8_t)bg_b(c3.y, 1) <= (int)bg_b(c3.y, 2)) {
        // :1247-1251 leave the shop; the hand is full, so all that changes is the phase, the mask and the shop rows (:1534-1539)
        const uint32_t nhand = bg_b(c3.y, 2), ncons = bg_b(c3.z, 1);
        c3.x &= 0xff00ffffu;                                              // phase = PLAY
        uint64_t m = (((1ull << (nhand < 8u ? nhand : 8u)) - 1ull) << 2) | (((1ull << ncons) - 1ull) << 10);
        if (nsel0 > 0u) m |= 1ull | (discards_left > 0u ? 2ull : 0ull);
        mask = m;
#pragma unroll
        for (int wq = 0; wq < 15; wq++) img32[44 + wq] = __umul24((uint32_t)(mask >> (4 * wq)) & 0xfu, 0x204081u) & 0x01010101u; // action_mask i8[60]
#pragma unroll
        for (int wq = 64; wq < 74; wq++) img32[wq] = 0u;                  // shop_items, shop_costs
        img8[BG_ROW_PHASE] = 0;
        fin = true;
      } else if (!terminal && !valid) { reward = -1.0; fin = true; }    // :626-627 'Invalid action': nothing changes
      else { post = true; cls = (!terminal && phase == 0u && action == 0) ? 0 : 1; pvalid = valid ? 1u : 0u; }
      if (fin) {
        *(__attribute__((address_space(3))) double*)&img32[34] = reward;  // BG_ROW_REWARD
        img32[43] = (uint32_t)action;                                      // BG_ROW_ACTION
        img8[BG_ROW_TERMINATED] = 0;
      }
    }
    E2T(1);
    // requests: chunks 3 / 4 to HBM, then one returning atomic per class on this wave's queue (its result is looked at in (4): by then the
    // stores before it have landed -- vector-memory operations complete in order)
    const unsigned long long pm = __ballot(post);
    unsigned long long m0 = 0, m1 = 0;
    uint32_t base = 0;
    const uint32_t qb = xcc * BG_E2_NQX + ((uint32_t)wave & (BG_E2_NSUB - 1u));
    if (pm) {
      if (post) { d.hot[3 * N + env] = c3; d.hot[4 * N + env] = c4; }
      m0 = __ballot(post && cls == 0); m1 = __ballot(post && cls == 1);
      if (m0 && lane == (int)(__ffsll((long long)m0) - 1)) base = __hip_atomic_fetch_add((g_u32*)&x.ctl->q[qb].tail, (uint32_t)__popcll(m0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (m1 && lane == (int)(__ffsll((long long)m1) - 1)) base = __hip_atomic_fetch_add((g_u32*)&x.ctl->q[qb + BG_E2_NSUB].tail, (uint32_t)__popcll(m1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pq_m0 = m0; pq_m1 = m1; pq_base = base;
      if (post) { // (t is still the index of the step in flight: a posted env's t moves when its answer has been absorbed)
        pq_hi = ((uint32_t)action & 0xffu) | (pvalid << 8) | ((t & 0xffffu) << 9) | 0x80000000u;
        st = 3u;
      }
    }
    E2T(2);
    // ---- (3) accounting and copy-out of the cheap steps' records: lane <-> 16-byte piece, non-temporal
    if (fin) {
      const uint64_t rb = ((uint64_t)img32[35] << 32) | img32[34];
      if (HASH) ohash ^= bg_hash_image((const lds_u4*)&s_img[tid][0]) * (0x9E3779B97F4A7C15ull + 2 * (uint64_t)(a.t0 + t)) + gi;
      n_steps++;
      rbits ^= rb * (2 * (uint64_t)(a.t0 + t) + 1);
    }
    const unsigned long long fm = __ballot(fin);
    if (fm) {
      const uint32_t nb = (uint32_t)__popcll(fm);
      if (fin) s_list[wave][__builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u))] = make_uint2((uint32_t)row, (uint32_t)tid);
      BG_WAVE_SYNC();
      typedef uint32_t bg_u32x2 __attribute__((ext_vector_type(2)));
      lds_cc* const lsb = (lds_cc*)&s_list[wave][0];
      if (whole) {
        for (uint32_t g0 = 0; g0 < nb; g0 += 16u) {
          bg_u32x2 ce[2][3];
          bg_u32x4 v[2][3];
#pragma unroll
          for (int u = 0; u < 2; u++)
#pragma unroll
            for (int j = 0; j < 3; j++)
              ce[u][j] = *(__attribute__((address_space(3))) const bg_u32x2*)(lsb + (((g0 + 8u * (uint32_t)u + rsel[j]) & 63u) << 3));
#pragma unroll
          for (int u = 0; u < 2; u++)
#pragma unroll
            for (int j = 0; j < 3; j++)
              v[u][j] = *(__attribute__((address_space(3))) const bg_u32x4*)(imgb + __umul24(ce[u][j].y & 0xffu, cmul[j]) + cib[j]);
#pragma unroll
          for (int u = 0; u < 2; u++)
#pragma unroll
            for (int j = 0; j < 3; j++)
              if (g0 + 8u * (uint32_t)u + rsel[j] < nb)
                __builtin_nontemporal_store(v[u][j], (__attribute__((address_space(1))) bg_u32x4*)(a.obs.rows + ((size_t)ce[u][j].x * 384u + cgl[j])));
        }
      } else {
        const uint32_t total = 22u * nb;
        for (uint32_t q0 = (uint32_t)lane; q0 < total; q0 += 4u * BG_BLOCK) {
          bg_u32x2 ce[4]; uint32_t cpc[4]; bg_u32x4 v[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const uint32_t q = q0 + (uint32_t)k * BG_BLOCK, r = (q * 2979u) >> 16;   // q / 22 (exact below 8 000)
            cpc[k] = q - 22u * r;
            ce[k] = *(__attribute__((address_space(3))) const bg_u32x2*)(lsb + ((r < nb ? r : 0u) << 3));
          }
#pragma unroll
          for (int k = 0; k < 4; k++) v[k] = s_img[ce[k].y & 0xffu][cpc[k]];
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (q0 + (uint32_t)k * BG_BLOCK < total)
              __builtin_nontemporal_store(v[k], (__attribute__((address_space(1))) bg_u32x4*)(a.obs.rows + (size_t)ce[k].x * (size_t)a.obs.row_stride + 16u * cpc[k]));
        }
      }
      BG_WAVE_SYNC();   // the images have been read (a wave's LDS operations complete in order): the next iteration may patch them
    }
    if (fin) t++;
    E2T(3);
    // ---- (4a) the loads issued one iteration ago: the record into the image, chunks 3 / 4 and the mask into registers -- the step is complete
    const unsigned long long am = __ballot(st == 2u);
    if (am) {
#pragma unroll
      for (int u = 0; u < BG_E2_AB; u++) if (ab_j[u] >= 0 && lane < 22) s_img[wave * BG_BLOCK + ab_j[u]][lane] = ab_rec[u];
      BG_WAVE_SYNC();
      if (st == 2u) {
        c3 = make_uint4(ab_c3.x, ab_c3.y, ab_c3.z, ab_c3.w); c4 = make_uint4(ab_c4.x, ab_c4.y, ab_c4.z, ab_c4.w);
        uint64_t m = 0;   // the action mask, from the record's action_mask i8[60]
#pragma unroll
        for (int wq = 0; wq < 15; wq++) { const uint32_t v = img32[44 + wq]; m |= (uint64_t)((v & 1u) | ((v >> 7) & 2u) | ((v >> 14) & 4u) | ((v >> 21) & 8u)) << (4 * wq); }
        mask = m;
        const uint64_t rb = ((uint64_t)img32[35] << 32) | img32[34];
        if (HASH) ohash ^= bg_hash_image((const lds_u4*)&s_img[tid][0]) * (0x9E3779B97F4A7C15ull + 2 * (uint64_t)(a.t0 + t)) + gi;
        n_steps++;
        rbits ^= rb * (2 * (uint64_t)(a.t0 + t) + 1);
        t++;
        st = 0u;
#ifdef BG_E2_TIMING
        { const unsigned long long now = wall_clock64(); e2_l0 += e2_tans - e2_tpost; e2_l1 += now - e2_tans; e2_ln++; }
#endif
      }
    }
    E2T(4);
    // ---- (4c) answers -> issue the loads of chunks 3 / 4 and, lane <-> piece, of the records of up to four envs (the others poll again)
    unsigned long long gm = __ballot(st == 1u && pollv == t + 1u);
#pragma unroll
    for (int u = 0; u < BG_E2_AB; u++) ab_j[u] = -1;
    if (gm) {
      const size_t arow = (size_t)env + (a.obs_stride_steps ? (size_t)t * N : 0);
      const uint32_t rlo = (uint32_t)arow, rhi = (uint32_t)(arow >> 32);
      unsigned long long take = 0;
#pragma unroll
      for (int u = 0; u < BG_E2_AB; u++) {
        if (gm) {
          const int j = __ffsll((long long)gm) - 1;
          gm &= gm - 1; take |= 1ull << j;
          ab_j[u] = j;
          const size_t rj = (size_t)__builtin_amdgcn_readlane(rlo, j) | ((size_t)__builtin_amdgcn_readlane(rhi, j) << 32);
          if (lane < 22) ab_rec[u] = __builtin_nontemporal_load((const __attribute__((address_space(1))) bg_u32x4*)(a.obs.rows + rj * (size_t)a.obs.row_stride + 16u * (uint32_t)lane));
        }
      }
      if ((take >> lane) & 1ull) {
        ab_c3 = __builtin_nontemporal_load((const __attribute__((address_space(1))) bg_u32x4*)&d.hot[3 * N + env]);
        ab_c4 = __builtin_nontemporal_load((const __attribute__((address_space(1))) bg_u32x4*)&d.hot[4 * N + env]);
        st = 2u;
#ifdef BG_E2_TIMING
        e2_tans = wall_clock64();
#endif
      }
    }
    E2T_CNT(11, __popcll(fm) + __popcll(am));
    // ---- nothing moved: every env of the wave is with a service wave
    if ((fm | pm | am | __ballot(st >= 2u)) == 0ull) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - last_progress > BG_E2_TIMEOUT) { if (lane == 0) atomicOr(d.err, BG_DEVERR_SPIN | 0x100u); break; }
    } else last_progress = wall_clock64();
    E2T(5);
  }
  E2T_FLUSH(0);
#ifdef BG_E2_TIMING
  for (int off = 32; off > 0; off >>= 1) { e2_l0 += __shfl_down(e2_l0, off); e2_l1 += __shfl_down(e2_l1, off); e2_ln += __shfl_down(e2_ln, off); }
  if (lane == 0) { atomicAdd(&x.ctl->tm[12], e2_l0); atomicAdd(&x.ctl->tm[13], e2_l1); atomicAdd(&x.ctl->tm[14], e2_ln); }
#endif
  // ---- epilogue: chunks 3 / 4, images and masks back to HBM; statistics; this wave is done
  if (live) { d.hot[3 * N + env] = c3; d.hot[4 * N + env] = c4; x.imask[env] = mask; }
  {
    // (each wave stores the images of its own 64 envs: no barrier needed)
    const int w0 = wave * BG_BLOCK, n_w = d.N - (env0 + w0) < BG_BLOCK ? d.N - (env0 + w0) : BG_BLOCK;
    bg_u32x4* dst = (bg_u32x4*)x.img + (size_t)(env0 + w0) * 24;
    for (int q = lane; q < n_w * 24; q += BG_BLOCK) {
      const int e = (q * 2731) >> 16, p = q - 24 * e;
      if (p < 22) dst[q] = s_img[w0 + e][p];
    }
  }
  if (a.stats) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { n_steps += __shfl_down(n_steps, off); rbits ^= __shfl_down(rbits, off); ohash ^= __shfl_down(ohash, off); }
    if (lane == 0) {
      atomicAdd((unsigned long long*)&a.stats->steps, (unsigned long long)n_steps);
      atomicXor((unsigned long long*)&a.stats->reward_bits, (unsigned long long)rbits);
      atomicXor((unsigned long long*)&a.stats->obs_hash, (unsigned long long)ohash);
    }
  }
  if (lane == 0) {
    atomicMax(&x.ctl->svc_stat[3], wall_clock64());
    const uint32_t old = __hip_atomic_fetch_add((g_u32*)&x.ctl->owners_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == x.done_target) __hip_atomic_store((g_u32*)&x.ctl->done_flag, x.done_target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The GATE: one wave in front of the service kernel on its stream.  It ends when every owner workgroup that can be resident has started (or
// after 0.5 ms: a timeout costs time, never correctness), so the service waves are always placed AFTER the owner workgroups.  Why: an owner
// workgroup needs ONE CONTIGUOUS 97 KB of a CU's 160 KB of LDS; four service workgroups (13 KB each) that arrive first -- beside transient refill
// workgroups -- are put wherever the allocator likes, and the 106 KB that remain may not contain 97 KB in one piece for as long as those
// service waves live, which is until the owner they lock out has finished: found as launches that took the 2 s of a bounded wait.
__global__ __launch_bounds__(BG_BLOCK) void bg_e2_gate_kernel(E2Args x) {
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    const uint32_t v = __builtin_amdgcn_readfirstlane(__hip_atomic_fetch_add((g_u32*)&x.ctl->owners_started, x.zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if ((int32_t)(v - x.start_target) >= 0 || wall_clock64() - t0 > 50000ull) break;
    __builtin_amdgcn_s_sleep(8);
  }
}
// development probe: a trivial one-wave kernel of the service kernel's register / LDS shape, launched in its place (BG_E2_PROBE)
template <int VG, int LDSW>
__global__ __launch_bounds__(BG_BLOCK) void bg_e2_probe_kernel(E2Args x) {
  __shared__ uint32_t s[LDSW > 0 ? LDSW : 1];
  if (LDSW > 0) s[threadIdx.x] = threadIdx.x;
  if (VG == 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
  if (threadIdx.x == 0) atomicMax(&x.ctl->svc_stat[6], ~wall_clock64());
  if (LDSW > 0 && s[1] == 999u) x.ctl->svc_stat[0] = 1;
}
// ------------------------------------------------------------------------------------------------------------------------------
// service kernel: one wave per workgroup, lane = request of the batch
// ------------------------------------------------------------------------------------------------------------------------------
// REGISTER ALLOCATION IS PART OF THE PROTOCOL.  The two kernels must be resident together whichever of them the dispatcher places first.  A
// service wave allocates 256 VGPRs + 8 AGPRs = 264 of a SIMD's 512 registers, so a SIMD holds ONE service wave (never two) and always has 248
// registers left for an owner wave (<= 136): at most four service waves per CU, exactly four when all 4 x 256 are resident, and an owner workgroup
// (one wave per SIMD, 97 KB of the CU's 160 KB of LDS beside 4 x 13 KB) fits on every CU in either order.  With 256 registers per service wave the
// dispatcher packed two per SIMD when the service kernel won the race: eight per CU on half the CUs, no owner workgroup could be placed there, and
// the owners that did run waited for service waves of their own XCD that had no owner to serve -- found as "whole XCDs unserved from the start of a
// launch" in one launch out of three (tools/e2_diag.py).
template <bool CARDS>
__global__ __launch_bounds__(BG_BLOCK) __attribute__((amdgpu_num_vgpr(256))) void bg_service_kernel(BgDev d, EngineArgs a, E2Args x) {
  asm volatile("v_accvgpr_write_b32 a7, 0" ::: "a7");   // eight accumulation registers that nothing uses (see above)
  __shared__ __attribute__((aligned(16))) uint32_t s_win[32][BG_BLOCK];   // the wave's RNG window (24 words per lane); at the end of a batch: record staging (64 x 8 pieces)
  __shared__ unsigned long long s_addr[BG_BLOCK];
  __shared__ JTables jt;
  using DeckT = typename std::conditional<CARDS, Deck0C, Deck0>::type;
  __builtin_amdgcn_s_setprio(3);
  if (threadIdx.x == 0) atomicMax(&x.ctl->svc_stat[6], ~wall_clock64());   // (development: the first instruction of the first service wave)
  bg_tables_load(&jt, d.jtab);
  const int lane = threadIdx.x;
  const size_t N = (size_t)d.N;
  const uint32_t xcc = bg_xcc_id();
  const uint32_t rmask = (1u << x.ring_log) - 1u;
  if (lane == 0) __hip_atomic_fetch_add((g_u32*)&x.ctl->svc_seen[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint64_t n_eps = 0, n_plays = 0;
  int64_t ssum = 0;
  unsigned long long idle_since = wall_clock64();
  if (lane == 0) atomicMax(&x.ctl->svc_stat[4], ~idle_since);
  unsigned long long q_batches = 0, q_reqs = 0;
  uint32_t x_why = 0, x_dn = 0, empty_polls = 0;
  bool seen = false;
  unsigned long long first_seen = 0;
#ifdef BG_E2_TIMING
  unsigned long long e2_wait = 0, e2_waitn = 0;
#endif
  // QUEUE PROTOCOL.  Words that atomics modify are never LOADED: an agent-scope load of such a word can be served a copy that the atomics have not
  // refreshed for hundreds of microseconds (measured: 30 failed compare-and-swaps per claimed batch, requests waiting 150 us beside idle waves), while
  // a word written by an agent-scope STORE is seen by the next load.  So a service wave RESERVES the next RS slots of a queue with one returning
  // fetch_add on its head -- before anything is in them -- and then polls the slots themselves (entries are stores; a consumed slot is zeroed again).
  // Producers reserve theirs with fetch_add on the tail.  Heads and tails start at 0 every launch (host memset); a slot's entry carries the generation
  // of its position, so a lapped slot is a loud error, never a wrong env.  The launch's end is a store too (E2Ctl.done_flag).
  const uint32_t RS = x.max_batch < 1u ? 1u : (x.max_batch > BG_BLOCK ? BG_BLOCK : x.max_batch);
  const uint32_t qi0 = xcc * BG_E2_NQX + ((blockIdx.x >> 3) & (BG_E2_NSUB - 1u)), qi1 = qi0 + BG_E2_NSUB;   // this wave's play queue and other queue (workgroup b runs on XCD b % 8: the sub-queue comes from the bits above)
  // a wave holds one range of RS slots per class: `base` and the set of slots it has served already (entries arrive in any order: each producer
  // wave writes its own an iteration after reserving them, and waiting for the slowest one first -- serving only the valid PREFIX -- made a request
  // wait 35 us on average beside idle waves)
  uint32_t base0, base1;
  unsigned long long served0 = 0, served1 = 0;
  const unsigned long long full = RS >= 64u ? ~0ull : ((1ull << RS) - 1ull);
  {
    uint32_t b = 0;
    if (lane < 2) b = __hip_atomic_fetch_add((g_u32*)&x.ctl->q[lane ? qi1 : qi0].head, RS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base0 = __builtin_amdgcn_readlane(b, 0); base1 = __builtin_amdgcn_readlane(b, 1);
  }
  E2T_DECL();
  for (;;) {
    // ---- the slots this wave holds, and the launch's end
    unsigned long long e0 = 0, e1 = 0;
    if ((uint32_t)lane < RS && !((served0 >> lane) & 1ull)) e0 = bg_ld8a(&x.ring[((size_t)qi0 << x.ring_log) + ((base0 + (uint32_t)lane) & rmask)]);
    if ((uint32_t)lane < RS && !((served1 >> lane) & 1ull)) e1 = bg_ld8a(&x.ring[((size_t)qi1 << x.ring_log) + ((base1 + (uint32_t)lane) & rmask)]);
    const uint32_t dflag = bg_ld4a(&x.ctl->done_flag);
    const unsigned long long v0 = __ballot((uint32_t)(e0 >> 63) != 0u), v1 = __ballot((uint32_t)(e1 >> 63) != 0u);
    const uint32_t k0 = (uint32_t)__popcll(v0), k1 = (uint32_t)__popcll(v1);
    if ((k0 | k1) == 0u) {
      if (__builtin_amdgcn_readfirstlane(dflag) == x.done_target) { x_why = 1; x_dn = x.done_target; break; }   // every owner wave of this launch is through
      __builtin_amdgcn_s_sleep(8);
      if (++empty_polls > 16u) __builtin_amdgcn_s_sleep(40);   // an idle wave backs off to ~1.5 us between polls
      if (wall_clock64() - idle_since > BG_E2_TIMEOUT) { if (lane == 0) atomicOr(d.err, BG_DEVERR_SPIN | 0x200u); x_why = 2; break; }
      continue;
    }
    empty_polls = 0;
    // a range that is filling is given fill_wait ticks to fill up: the rest of it would otherwise wait for this wave's batch to end
    if ((served0 | v0) != full && (served1 | v1) != full && x.fill_wait) {
      const unsigned long long now = wall_clock64();
      if (!seen) { seen = true; first_seen = now; }
      if (now - first_seen < (unsigned long long)x.fill_wait) { __builtin_amdgcn_s_sleep(2); continue; }
    }
    seen = false;
    const bool cls1 = k1 > k0;   // the class with more requests waiting (plays on a tie: the longer chains)
    const unsigned long long vm = cls1 ? v1 : v0;
    const uint32_t nb = cls1 ? k1 : k0, head = cls1 ? base1 : base0, qi = cls1 ? qi1 : qi0;
    const unsigned long long ent = cls1 ? e1 : e0;
    E2T(0);
    E2T_CNT(10, 1); E2T_CNT(11, nb);
    // when this batch exhausts the range the next reservation is requested now and looked at after the batch
    const bool renew = ((cls1 ? served1 : served0) | vm) == full;
    uint32_t nbase = 0;
    if (renew && lane == 0) nbase = __hip_atomic_fetch_add((g_u32*)&x.ctl->q[qi].head, RS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool active = false;
    int env = 0, action = 0;
    uint32_t t = 0, pvalid = 0;
    if ((vm >> lane) & 1ull) {
      const uint32_t pos = head + (uint32_t)lane, hi = (uint32_t)(ent >> 32);
      if (((hi >> 25) & 0x3fu) != ((pos >> x.ring_log) & 0x3fu)) atomicOr(d.err, BG_DEVERR_SPIN | 0x400u);   // the ring was lapped
      else { active = true; env = (int)((uint32_t)ent & 0xffffffu); action = (int)(hi & 0xffu); pvalid = (hi >> 8) & 1u; t = (hi >> 9) & 0xffffu; }
#ifdef BG_E2_TIMING
      e2_wait += (uint32_t)(((wall_clock64() >> 6) - ((uint32_t)ent >> 24)) & 0xffu); e2_waitn++;
#endif
      bg_st8a(&x.ring[((size_t)qi << x.ring_log) + (pos & rmask)], 0ull);   // the slot is free again
    }
    E2T(1);
    if (active) {
      BG_PROBE_BEGIN();
      const size_t row = (size_t)env + (a.obs_stride_steps ? (size_t)t * N : 0);
      // state: 16-byte NON-TEMPORAL loads -- they bypass the CU's L1 (which may hold the line from a neighbour env's earlier service step here,
      // with this env's chunk as it was then) and are served by the XCD's L2, where the last writer's plain stores are
      uint4 c[BG_NHOT];
#pragma unroll
      for (int k = 0; k < BG_NHOT; k++) { const bg_u32x4 v = __builtin_nontemporal_load((const __attribute__((address_space(1))) bg_u32x4*)&d.hot[(size_t)k * N + env]); c[k] = make_uint4(v.x, v.y, v.z, v.w); }
      DeckT dk;
      { const bg_u32x4 dc = __builtin_nontemporal_load((const __attribute__((address_space(1))) bg_u32x4*)&d.deck[env]); dk.lo = ((uint64_t)dc.y << 32) | dc.x; dk.hi = ((uint64_t)dc.w << 32) | dc.z; }
      Env e;
      bg_unpack(c, e);
      bg_derive_ready(e, d.prod_view ? d.prod_view[env] : 0u);
      E2T(2);
      ShopRegs sr; sr.valid = false;
      RngWin w;
      bg_win_init(w, &s_win[0][lane], &jt);
      StepOut o;
      bg_step_init(o);
      o.bd_dst = nullptr;
      // the owner has checked the action against the env's mask (pvalid); the terminal guards come first, as in the reference (:619-627)
      if (bg_step_guards(e, pvalid ? ~0ull : 0ull, action, o)) bg_env_dispatch(d, env, e, w, sr, dk, action, o);
      E2T(3);
      if (e.max_ante > 0 && e.ante > e.max_ante) { o.terminated = true; o.flags |= 256; }
      if (o.terminated) n_eps++;
      if (o.terminated && a.autoreset) bg_env_reset(d, env, e, dk);
      E2T(4);
      const uint64_t mask = bg_action_mask(d, env, e, sr);
      if (o.hand_type >= 0) { n_plays++; ssum += o.final_score; }
      bg_pack(e, c);
#pragma unroll
      for (int k = 0; k < BG_NHOT; k++) d.hot[(size_t)k * N + env] = c[k];
      E2T(5);
      // the record, straight to its row: three slices of 8 / 7 / 7 pieces staged through LDS (lane <-> piece: runs of whole records)
      ObsPtrs op = a.obs;
#ifndef BG_E2_DIRECT_RECORD
      bg_write_obs_impl<false, 2>(d, env, row, e, dk, op, mask, sr, RowExtra{o.reward, action, o.terminated ? 1u : 0u}, RowStage{(lds_u4*)&s_win[0][0], (lds_u64*)&s_addr[0]});
#else
      // every lane stores its own 22 pieces: plain stores, the XCD's L2 puts the lines together (the owner reads them back from there)
      bg_write_obs_impl<false, 0>(d, env, row, e, dk, op, mask, sr, RowExtra{o.reward, action, o.terminated ? 1u : 0u}, RowStage{nullptr, nullptr});
#endif
      if (a.obs.row_stride == 384u) { // the two padding pieces of a whole-line record are zeros (bg_engine.h, copier)
        __attribute__((address_space(1))) bg_u32x4* pad = (__attribute__((address_space(1))) bg_u32x4*)(a.obs.rows + row * 384u + 352u);
        pad[0] = bg_u32x4{0u, 0u, 0u, 0u}; pad[1] = bg_u32x4{0u, 0u, 0u, 0u};
      }
      E2T(6);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // state and record are in this XCD's L2 before the answer is
      E2T(7);
      __hip_atomic_store((g_u32*)&x.ansq[env], t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    idle_since = wall_clock64();
    q_batches++; q_reqs += nb;
    if (cls1) { served1 |= vm; if (renew) { base1 = __builtin_amdgcn_readfirstlane(nbase); served1 = 0; } }
    else { served0 |= vm; if (renew) { base0 = __builtin_amdgcn_readfirstlane(nbase); served0 = 0; } }
    E2T(8);
  }
  E2T_FLUSH(16);
#ifdef BG_E2_TIMING
  for (int off = 32; off > 0; off >>= 1) { e2_wait += __shfl_down(e2_wait, off); e2_waitn += __shfl_down(e2_waitn, off); }
  if (lane == 0) { atomicAdd(&x.ctl->tm[28], e2_wait); atomicAdd(&x.ctl->tm[29], e2_waitn); }
#endif
  if (lane == 0) {
    const unsigned long long now = wall_clock64();
    atomicMax(&x.ctl->svc_stat[5], now); atomicAdd(&x.ctl->svc_stat[0], q_batches); atomicAdd(&x.ctl->svc_stat[1], q_reqs);
    atomicAdd(&x.ctl->x_exits[xcc], 1u); atomicOr(&x.ctl->x_reason[xcc], x_why); atomicMax(&x.ctl->x_done[xcc], x_dn); atomicMax(&x.ctl->x_target[xcc], ~x.done_target);
    atomicMax(&x.ctl->x_t0[xcc], ~now); atomicMax(&x.ctl->x_t1[xcc], now); atomicAdd(&x.ctl->x_batches[xcc], q_batches);
  }
  if (a.stats) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { n_eps += __shfl_down(n_eps, off); n_plays += __shfl_down(n_plays, off); ssum += __shfl_down(ssum, off); }
    if (lane == 0 && (n_eps | n_plays)) {
      atomicAdd((unsigned long long*)&a.stats->episodes, (unsigned long long)n_eps);
      atomicAdd((unsigned long long*)&a.stats->plays, (unsigned long long)n_plays);
      atomicAdd((unsigned long long*)&a.stats->score_sum, (unsigned long long)ssum);
    }
  }
}
