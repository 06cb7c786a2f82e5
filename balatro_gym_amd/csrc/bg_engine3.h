// bg_engine3.h -- the step engine with OWNER waves inside the workgroup (packed-record rollouts).
//
// What rounds 2-4 measured, put together:
//   * bg_engine.h (round 2/3): every wave is a worker that pulls batches from LDS queues.  A cheap step (card-select toggle: 84 % of a random
//     policy's steps) costs an env ~9 k cycles, of which ~4 k are the batch and the rest is waiting for one of five workers and for a copier.
//   * bg_engine2.h (round 4): cheap steps by OWNER waves (lane = env for the whole launch, no queue, no claim) and service steps by a chip-wide
//     pool of service waves in a second kernel.  The owner loop is what it should be -- ~2 k cycles of work per iteration for ~13 steps -- but every
//     hand-over hop between two CUs goes through the L2 / memory at 1.5 - 2.5 us under load, a service step needs eight of them in series
//     (queue position, entry, state, record, answer, chunks 3 / 4 ...), and the pipeline that hides them adds four owner iterations: 4.5 G
//     env-steps/s against 6.8 G (profiles/r04_two_kernel_*.txt).
// This kernel keeps the owner waves and puts the service waves back INTO the workgroup: four owner waves (lane = env: cheap step on the env's LDS
// image, copy-out of every finished record as whole lines) + three service waves that pull PLAY_HAND / other requests from two LDS queues exactly
// as bg_engine.h's workers do (same device functions, same queue protocol) and answer through an LDS word per env.  Hand-over = LDS (~100 cycles a
// hop); the state of an env in HBM is only ever touched from this CU (the memory model of bg_engine.h).
#pragma once

#ifdef BG_E3_TIMING   // development: cycle sums into d.dbg (tools/e3_timing.py)
#define E3T_DECL() unsigned long long e3t_[8] = {0,0,0,0,0,0,0,0}, e3t0_ = __builtin_readcyclecounter()
#define E3T(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); e3t_[k] += n_ - e3t0_; e3t0_ = n_; } while (0)
#define E3T_CNT(k, v) do { e3t_[k] += (v); } while (0)
#define E3T_FLUSH(base) do { if (lane == 0 && d.dbg) for (int k_ = 0; k_ < 8; k_++) atomicAdd(&d.dbg[(base) + k_], e3t_[k_]); } while (0)
constexpr bool kE3Timing = true, kE3Tl = true;
#else
#define E3T_DECL() const unsigned long long e3t_[8] = {0,0,0,0,0,0,0,0}; (void)e3t_
#define E3T(k) do {} while (0)
#define E3T_CNT(k, v) do {} while (0)
#define E3T_FLUSH(base) do {} while (0)
constexpr bool kE3Timing = false;
#ifdef BG_E3_TL   // development: only the workgroup timeline (tools/e3_timeline.py)
constexpr bool kE3Tl = true;
#else
constexpr bool kE3Tl = false;
#endif
#endif

// the record stores of the whole-line copy-out: non-temporal (`sc1` / `sc0 sc1` / plain stores: -12 %, `sc1 nt`: the same; profiles/r04_engine3/record_stream_sensitivity.txt)
#define BG_E3_REC_STORE(v, p) __builtin_nontemporal_store(v, p)
#define BG_E3_SPARSE 8u   // records finished in an owner iteration up to which the copy-out takes its one-group path
#define BG_E3_SPARSE2 16u // ... its two-group path (a three-group tier: inside the noise, profiles/r05/sparse_copy_out.txt)
template <int NOW, int KS, int NSV>
struct E3Lds {
  static constexpr int NE = NOW * KS * BG_BLOCK;
  bg_u32x4 s_img[NE][22];
  uint4 s_c34[2][NE];              // hot chunks 3 and 4
  unsigned long long s_mask[NE];
  uint32_t s_prod[NE];
  uint16_t s_ans[NE];              // service steps completed for this env in this launch, mod 2**16 (the owner counts its requests; a launch fuses < 2**16 steps)
  uint32_t s_q[2][NE];             // request rings: env lane | generation of the ring position << 8 | action << 16 | VALID
  __attribute__((aligned(16))) uint32_t s_ctl[8];   // [0..1] requests ever queued per class, [4..5] ever claimed
  uint32_t s_win[NSV][BG_WIN][BG_BLOCK];
  uint2 s_list[NOW][KS * BG_BLOCK]; // copy-out list of an owner wave: .x = record row, .y = env lane
  // The NEXT pre-shuffled deck of every env (the ring slot a reset will consume) and its state: 0 = not here (the reset reads the ring itself), 1 = here,
  // 2 | slot << 8 = consumed, the owner is to fetch ring slot `slot`.  A reset's copy ring -> deck is a dependent HBM round trip in the middle of a
  // service batch that nearly always holds a lane whose episode ends (5.9 k of the 56 k service cycles per workgroup-step, profiles/r04_engine3/probes.txt);
  // the owner waves have the time to fetch the deck ahead (their lanes are mostly waiting).
  uint4 s_nd[BG_NDECK][NE];
  uint32_t s_ndst[NE];
  bg_u32x4 s_zero;
  unsigned long long s_stats[6];   // the workgroup's share of bg_rollout_stats (bg_stats_wave / bg_stats_flush)
  uint32_t s_owners_left;
  JTables jt;
};
// A workgroup = NE = 64 * NOW * KS envs: NOW owner waves (each owns KS slices of 64 envs, lane = env of a slice) + NSV service waves.  256 envs per
// workgroup fill the chip at 65 536 envs; a small job takes 64 or 128 per workgroup and spreads over four or two times as many CUs.
template <bool HASH, bool CARDS, int NOW, int KS, int NSV>
// (Round 5 measured reading the arguments THROUGH the kernarg segment pointer instead of as by-value parameters -- whose 16-register blocks the compiler
//  spills to VGPR lanes and reloads whole, 12 % of the kernel's instructions being v_readlane / v_writelane / s_nop: SGPR spills 237 -> 61, 14 454 -> 13 114
//  instructions, and 3.4 % SLOWER at both launch lengths: a scalar load per use waits longer than sixteen lane reads.  profiles/r05/play_path_ab.txt.)
__global__ __launch_bounds__((NOW + NSV) * BG_BLOCK, 2) void bg_engine3_kernel(BgDev d, EngineArgs a) {
  constexpr int NE = NOW * KS * BG_BLOCK, LNE = NE == 256 ? 8 : (NE == 128 ? 7 : 6);
  static_assert(NE == 64 || NE == 128 || NE == 256, "envs per workgroup (a request carries the env's lane in 8 bits)");
  // (static LDS: the compiler then pads the register allocation to the 256 VGPRs that two waves per SIMD leave each -- which this kernel needs
  //  anyway: with dynamic LDS and the same cap it compiles to 256 VGPRs plus a spilled one)
  __shared__ E3Lds<NOW, KS, NSV> L;
  auto& s_img = L.s_img; auto& s_c34 = L.s_c34; auto& s_mask = L.s_mask; auto& s_prod = L.s_prod; auto& s_ans = L.s_ans; auto& s_q = L.s_q;
  auto& s_ctl = L.s_ctl; auto& s_win = L.s_win; auto& s_list = L.s_list; auto& s_zero = L.s_zero; auto& s_owners_left = L.s_owners_left; auto& jt = L.jt;
  auto& s_nd = L.s_nd; auto& s_ndst = L.s_ndst;
  __builtin_amdgcn_s_setprio(2);
  const unsigned long long e3_k0 = kE3Tl ? wall_clock64() : 0ull;
  BG_PROBE_INIT();
  bg_tables_load(&jt, d.jtab);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = tid >> 6;
  // (64-env shape: a small job may ask for fewer LIVE envs per workgroup -- a.epw of the 64 lanes of the one owner wave -- and so for more workgroups:
  //  4 096 envs as 512 workgroups of 8 instead of 64 of 64 fill the chip, and an env's service step no longer queues behind 63 neighbours)
  const int epw = (NE == 64 && a.epw >= 1u && a.epw < 64u) ? (int)a.epw : NE;
  const int env0 = blockIdx.x * epw;
  const int n_live = d.N - env0 < epw ? d.N - env0 : epw;
  // (decks stay in HBM: the first 16 cards -- every index a hand normally holds -- ride in two registers of the service lane, bg_device.h Deck0)
  using DeckT = typename std::conditional<CARDS, Deck0C, Deck0>::type;
  const size_t N = (size_t)d.N;
  typedef __attribute__((address_space(3))) const char lds_cc;
  // ---------------------------------------------------------------- prologue: HBM -> LDS, images built, lane = env (bg_engine.h)
  if (tid < 8) s_ctl[tid] = 0;
  if (tid == 0) { s_zero = bg_u32x4{0u, 0u, 0u, 0u}; s_owners_left = NOW; }
  if (tid < 6) L.s_stats[tid] = 0ull;
  for (int l = tid; l < NE; l += (NOW + NSV) * BG_BLOCK) {
    const int env = env0 + l;
    s_q[0][l] = 0; s_q[1][l] = 0; s_ans[l] = 0; s_ndst[l] = 0;
    if (l < n_live) {
      uint4 c[BG_NHOT];
#pragma unroll
      for (int k = 0; k < BG_NHOT; k++) c[k] = d.hot[(size_t)k * N + env];
      s_c34[0][l] = c[3]; s_c34[1][l] = c[4];
      DeckT dk; static_cast<Deck0&>(dk) = bg_load_deck0(d, env);
      const uint32_t prod = d.prod_view ? d.prod_view[env] : 0u;
      s_prod[l] = prod;
      Env e;
      bg_unpack(c, e);
      bg_derive_ready(e, prod);
      s_ndst[l] = e.d_ready > 0 ? 1u : 0u;
      if (e.d_ready > 0) {
#pragma unroll
        for (int k = 0; k < BG_NDECK; k++) s_nd[k][l] = d.ndeck[((size_t)e.d_head * BG_NDECK + k) * N + env];
      }
      ShopRegs sr; sr.valid = false;
      const uint64_t mask = bg_action_mask(d, env, e, sr);
      s_mask[l] = mask;
      const ObsPtrs none{};
      bg_write_obs_impl<false, 3>(d, env, 0, e, dk, none, mask, sr, RowExtra{0.0, 0, 0u}, RowStage{(lds_u4*)&s_img[l][0], nullptr});
    }
  }
  __syncthreads();
  if constexpr (kE3Timing) if (lane == 0) { uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); atomicOr(&s_ctl[2], ((hw >> 4) & 3u) << (2 * wave)); }   // (word 2: nobody's)
  const unsigned long long e3_w0 = kE3Tl ? wall_clock64() : 0ull;
  if constexpr (kE3Tl) {
    if (tid == 0 && d.dbg) atomicAdd(&d.dbg[28], e3_w0 - e3_k0);   // prologue (100 MHz ticks, summed over workgroups)
    __syncthreads();
  }
  const uint32_t T = (uint32_t)a.T;
  if (wave < NOW) {
    // ============================================================== OWNER wave: lane = env of each of its KS slices, for the whole launch
    const uint32_t bmod3 = (uint32_t)(a.env_index0 % 3ull);
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
    uint32_t t[KS], nreq[KS];   // per slice: steps done; requests posted (the service wave counts them in s_ans)
    bool waiting[KS], terminal[KS];
    // chunks 3 / 4 and the action mask of the lane's env live in REGISTERS between two service steps (a toggle is then LDS writes only -- the image --,
    // no LDS read: the step phase of an iteration was five dependent LDS round trips); they go to LDS with a request and come back with its answer
    uint4 rc3[KS], rc4[KS];
    uint64_t rmask[KS];
    // the policy's counter-hash input x0 = seed_env + PSI * (t0 + t + 1) and the record's row env + t * N advance by an ADD per finished step (they
    // were two 64-bit multiplies and a 32 x 32 one in every owner iteration: ~10 quarter-rate multiplies of a ~600-instruction step phase); the blind a
    // scripted policy picks is a constant of the env
    uint64_t px[KS];
    size_t prow[KS];
    int pblind[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) {
      t[s] = 0; nreq[s] = 0; waiting[s] = false;
      const int l = (wave * KS + s) * BG_BLOCK + lane;
      {
        const int env = env0 + l;
        const uint64_t gi0 = a.env_index0 + (uint64_t)env;
        px[s] = a.policy_seed + 0x9E3779B97F4A7C15ull * (gi0 + 1) + BG_POLICY_PSI * (a.t0 + 1);
        prow[s] = (size_t)env;
        pblind[s] = a.policy == 2 ? 45 + (int)((bmod3 + (uint32_t)env % 3u) % 3u) : 45;
      }
      rc3[s] = s_c34[0][l]; rc4[s] = s_c34[1][l]; rmask[s] = s_mask[l];
      const lds_u32* im = (const lds_u32*)&s_img[l][0];
      terminal[s] = bg_b(rc3[s].x, 0) > 100u || (int64_t)(((uint64_t)im[33] << 32) | im[32]) > 1000000000ll;   // :619-623
    }
    uint64_t n_steps = 0, rbits = 0, ohash = 0;
    uint32_t idle = 0;
    // a ring deck on its way into LDS (requested when the answer of a step that reset the env is seen, stored at the end of the iteration,
    // behind the copy-out: the round trip runs beside it): per slice
    E3T_DECL();
    for (;;) {
      bg_u32x4 ndv[KS][BG_NDECK];
      bool ndp[KS];
#pragma unroll
      for (int s = 0; s < KS; s++) ndp[s] = false;
      bool busy = false;
#pragma unroll
      for (int s = 0; s < KS; s++) busy = busy || ((wave * KS + s) * BG_BLOCK + lane < n_live && (t[s] < T || waiting[s]));
      if (__ballot(busy) == 0ull) break;
      E3T_CNT(4, 1);
      uint32_t nb = 0;   // records finished in this iteration (all slices): the copy-out list
      bool glast = false; // ... one of them is a last-step record of a launch whose current records are gathered (a.gworld)
#pragma unroll
      for (int s = 0; s < KS; s++) {
        const int l = (wave * KS + s) * BG_BLOCK + lane, env = env0 + l;
        const bool live = l < n_live;
        lds_u32* const img32 = (lds_u32*)&s_img[l][0];
        lds_u8* const img8 = (lds_u8*)&s_img[l][0];
        const size_t row = prow[s];   // env + t * N (a.obs_stride_steps), or env
        // ---- answers: the service wave has left the finished image, chunks 3 / 4 and the mask in LDS
        bool fin = false;
        if (waiting[s] && bg_lds_ld(&s_ans[l]) == (nreq[s] & 0xffffu)) {
          waiting[s] = false; fin = true;
          const uint32_t nds = bg_lds_ld(&s_ndst[l]);
          if ((nds & 0xffu) == 2u) {   // the step reset the env and consumed its LDS deck: fetch the next ring slot
#pragma unroll
            for (int k = 0; k < BG_NDECK; k++) ndv[s][k] = *(const __attribute__((address_space(1))) bg_u32x4*)&d.ndeck[((size_t)(nds >> 8) * BG_NDECK + k) * N + env];
            bg_lds_st(&s_ndst[l], 0u);   // (not here until it has landed: a reset in between reads the ring itself)
            ndp[s] = true;
          }
          rc3[s] = s_c34[0][l]; rc4[s] = s_c34[1][l]; rmask[s] = s_mask[l];
          terminal[s] = bg_b(rc3[s].x, 0) > 100u || (int64_t)(((uint64_t)img32[33] << 32) | img32[32]) > 1000000000ll;
        }
        // ---- the cheap step of every ready env (bg_engine.h: cheap_step)
        else if (live && !waiting[s] && t[s] < T) {
          uint64_t mask = rmask[s];
          const uint4 c3 = rc3[s];
          const uint32_t phase = bg_b(c3.x, 2), discards_left = bg_b(c3.y, 0), nsel0 = bg_b(c3.y, 3);
          int action;
          {
            Env pe; pe.phase = (int)phase; // the policy only looks at the phase and the mask
            PolicyLane pl; pl.seed_env = 0; pl.blind = pblind[s];   // (bg_policy_action_fast reads the blind and x0 only)
            action = bg_policy_action_fast(pe, mask, a.policy, pl, px[s], (lds_JTables*)&jt);
          }
          double reward = 0.0;
          const bool valid = action >= 0 && action < 60 && ((mask >> (action & 63)) & 1ull);
          const bool term = terminal[s];   // :619-623, settled by a service wave (it resets the env)
          if (!term && valid && phase == 0u && action >= 2 && action < 10) {
            // :1052-1058 toggle position `pos` in state.selected_cards (bg_toggle_select on chunk 4)
            const int pos = action - 2;
            Env te; te.sel = ((uint64_t)rc4[s].w << 32) | rc4[s].z; te.nsel = (int)nsel0;
            bg_toggle_select(te, pos);
            rc4[s].z = (uint32_t)te.sel; rc4[s].w = (uint32_t)(te.sel >> 32);
            rc3[s].y = (rc3[s].y & 0x00ffffffu) | ((uint32_t)te.nsel << 24);   // chunk 3, word y, byte 3
            *(lds_u64*)&img32[2 * pos] = te.nsel > (int)nsel0 ? 1ull : 0ull;  // selected_cards[pos] (int64)
            if ((te.nsel > 0) != (nsel0 > 0u)) {                             // PLAY_HAND / DISCARD availability (:1436-1441)
              const uint32_t play = te.nsel > 0 ? 1u : 0u, disc = (te.nsel > 0 && discards_left > 0u) ? 1u : 0u;
              mask = (mask & ~3ull) | play | ((uint64_t)disc << 1);
              rmask[s] = mask;
              *(__attribute__((address_space(3))) uint16_t*)&img8[BG_ROW_ACTION_MASK] = (uint16_t)(play | (disc << 8));
            }
            fin = true;
          } else if (!term && valid && phase == 1u && action == 31 && (int)(int8_t)bg_b(c3.y, 1) <= (int)bg_b(c3.y, 2)) {
            // :1247-1251 leave the shop; the hand is full, so all that changes is the phase, the mask and the shop rows (:1534-1539)
            const uint32_t nhand = bg_b(c3.y, 2), ncons = bg_b(c3.z, 1);
            rc3[s].x &= 0xff00ffffu;                                          // chunk 3, word x, byte 2: phase = PLAY
            uint64_t m = (((1ull << (nhand < 8u ? nhand : 8u)) - 1ull) << 2) | (((1ull << ncons) - 1ull) << 10);
            if (nsel0 > 0u) m |= 1ull | (discards_left > 0u ? 2ull : 0ull);
            mask = m;
            rmask[s] = mask;
#pragma unroll
            for (int wq = 0; wq < 15; wq++) img32[44 + wq] = __umul24((uint32_t)(mask >> (4 * wq)) & 0xfu, 0x204081u) & 0x01010101u; // action_mask i8[60]
#pragma unroll
            for (int wq = 64; wq < 74; wq++) img32[wq] = 0u;                  // shop_items, shop_costs
            img8[BG_ROW_PHASE] = 0;
            fin = true;
          } else if (!term && !valid) { reward = -1.0; fin = true; }    // :626-627 'Invalid action': nothing changes
          else {
            // a request: chunks 3 / 4 and the mask to LDS, then one queue word (the service waves poll the tails)
            s_c34[0][l] = rc3[s]; s_c34[1][l] = rc4[s]; s_mask[l] = mask;
            const int q = (!term && phase == 0u && action == 0) ? 0 : 1;
            const uint32_t slot = __hip_atomic_fetch_add(&s_ctl[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            bg_lds_st(&s_q[q][slot & (NE - 1)], (uint32_t)l | (((slot >> LNE) & 0xffu) << 8) | (((uint32_t)action & 0x7fffu) << 16) | BG_ITEM_VALID);
            nreq[s]++;
            waiting[s] = true;
          }
          if (fin) {
            *(__attribute__((address_space(3))) double*)&img32[34] = reward;  // BG_ROW_REWARD
            img32[43] = (uint32_t)action;                                      // BG_ROW_ACTION
            img8[BG_ROW_TERMINATED] = 0;
          }
        }
        // ---- accounting of every record finished in this iteration (cheap or served), and its place in the copy-out list
        if (fin) {
          const uint64_t rb = ((uint64_t)img32[35] << 32) | img32[34];
          if (HASH) ohash ^= bg_hash_image((const lds_u4*)&s_img[l][0]) * (0x9E3779B97F4A7C15ull + 2 * (uint64_t)(a.t0 + t[s])) + (a.env_index0 + (uint64_t)env);
          n_steps++;
          rbits ^= rb * (2 * (uint64_t)(a.t0 + t[s]) + 1);
          t[s]++;
          px[s] += BG_POLICY_PSI;
          if (a.obs_stride_steps) prow[s] += N;
        }
        const unsigned long long fms = __ballot(fin);
        // (.y bit 8: the record of the launch's last step -- the one a sharded job gathers; the copy-out reads the lane out of the low byte)
        const bool lastrec = fin && a.gworld != 0u && t[s] == T;
        if (fin) s_list[wave][nb + __builtin_amdgcn_mbcnt_hi((uint32_t)(fms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fms, 0u))] = make_uint2((uint32_t)row, (uint32_t)l | (lastrec ? 0x100u : 0u));
        nb += (uint32_t)__popcll(fms);
        glast = glast || lastrec;
      }
      E3T(0);
      // ---- copy-out: lane <-> 16-byte piece, non-temporal (32 records per round of loads: the LDS round trips overlap)
      if (nb) {
        idle = 0;
        BG_WAVE_SYNC();
        typedef uint32_t bg_u32x2 __attribute__((ext_vector_type(2)));
        lds_cc* const lsb = (lds_cc*)&s_list[wave][0];
        // NU groups of eight records from list position G0: 3 NU list reads, 3 NU image reads (the LDS round trips of a group overlap), 3 NU stores.
        // (Round 5: unconditional stores with the empty lanes aimed at a dump line -- no exec-mask save and branch per store -- cost 7-8 % at both launch
        //  lengths: profiles/r05/play_path_ab.txt.  Round 4's sensitivity builds -- no copy-out / no stores / every record into 24 KB -- are in the history
        //  at cdcab7a: profiles/r04_engine3/record_stream_sensitivity.txt)
#define BG_E3_LIVE(idx) ((idx) < nb)
#define BG_E3_ROW(x) (x)
#define BG_E3_COPY_GROUPS(NU, G0) do { \
          bg_u32x2 ce[NU][3]; \
          bg_u32x4 v[NU][3]; \
          _Pragma("unroll") for (int u = 0; u < NU; u++) _Pragma("unroll") for (int j = 0; j < 3; j++) \
            ce[u][j] = *(__attribute__((address_space(3))) const bg_u32x2*)(lsb + ((((G0) + 8u * (uint32_t)u + rsel[j]) & (KS * BG_BLOCK - 1u)) << 3)); \
          _Pragma("unroll") for (int u = 0; u < NU; u++) _Pragma("unroll") for (int j = 0; j < 3; j++) \
            v[u][j] = *(__attribute__((address_space(3))) const bg_u32x4*)(imgb + __umul24(ce[u][j].y & 0xffu, cmul[j]) + cib[j]); \
          _Pragma("unroll") for (int u = 0; u < NU; u++) _Pragma("unroll") for (int j = 0; j < 3; j++) \
            if (BG_E3_LIVE((G0) + 8u * (uint32_t)u + rsel[j])) \
              BG_E3_REC_STORE(v[u][j], (__attribute__((address_space(1))) bg_u32x4*)(a.obs.rows + ((size_t)BG_E3_ROW(ce[u][j].x) * 384u + cgl[j]))); \
        } while (0)
        const uint32_t nbs = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);   // (wave-uniform: counted from ballots)
        if (whole && nbs <= BG_E3_SPARSE) {
          // A SPARSE iteration (a small job's 16 live envs per workgroup; the first and last iterations of a short launch): at most eight records = ONE
          // group -- three list reads, three image reads, three stores instead of twelve of each (nine of which had no lane to serve)
          BG_E3_COPY_GROUPS(1, 0u);
        } else if (whole && nbs <= BG_E3_SPARSE2) {
          BG_E3_COPY_GROUPS(2, 0u);
        } else if (whole) {
          for (uint32_t g0 = 0; g0 < nb; g0 += 32u) BG_E3_COPY_GROUPS(4, g0);
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
        // ---- sharded jobs: the CURRENT record of every env (the launch's last step) also goes into every rank's gather buffer, from here -- peer-mapped
        // stores over xGMI (own buffer: local HBM), 22 pieces per record and rank -- so that the gather is part of the launch instead of a collective
        // that can only start when the launch has retired (RCCL's multi-wave workgroups do not fit beside a resident engine workgroup).  Only iterations
        // that finished such a record come here (once per env and launch).
        if (a.gworld != 0u && __ballot(glast) != 0ull) {
          const uint32_t total = 22u * nb;
          for (uint32_t q = (uint32_t)lane; q < total; q += BG_BLOCK) {
            const uint32_t r = (q * 2979u) >> 16, pc = q - 22u * r;   // q / 22 (exact below 8 000)
            const bg_u32x2 ce = *(__attribute__((address_space(3))) const bg_u32x2*)(lsb + (r << 3));
            if (ce.y & 0x100u) {
              const uint32_t gl = ce.y & 0xffu;
              const bg_u32x4 v = s_img[gl][pc];
              const size_t off = ((size_t)a.grank * N + (size_t)(env0 + (int)gl)) * 352u + 16u * pc;
              for (uint32_t g = 0; g < a.gworld; g++) __builtin_nontemporal_store(v, (__attribute__((address_space(1))) bg_u32x4*)(a.gpeer[g] + off));   // (a.gpeer[g]: a scalar load)
            }
          }
        }
        BG_WAVE_SYNC();   // the images have been read (a wave's LDS operations complete in order): the next iteration may patch them
#pragma unroll
        for (int s = 0; s < KS; s++) {
          if (ndp[s]) {
            const int l = (wave * KS + s) * BG_BLOCK + lane;
#pragma unroll
            for (int k = 0; k < BG_NDECK; k++) *(__attribute__((address_space(3))) bg_u32x4*)&s_nd[k][l] = ndv[s][k];
            bg_lds_st(&s_ndst[l], 1u);
            ndp[s] = false;
          }
        }
        E3T(1); E3T_CNT(5, nb);
      } else {
        __builtin_amdgcn_s_sleep(2);   // every env of the wave is with a service wave
        if (++idle > BG_SPIN_LIMIT) { if (lane == 0) atomicOr(d.err, BG_DEVERR_SPIN); break; }
        E3T(2);
      }
    }
    E3T_FLUSH(0);
    // (does the owner wave that shares its SIMD with the refill's waves instead of a service wave lag?)  by the number of engine waves on this wave's SIMD:
    // dbg[16 + 4 k ..] = sum of end times (100 MHz ticks since the workgroup started), iterations, waves, steps done
    if constexpr (kE3Timing) if (lane == 0 && d.dbg) {
      uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      const uint32_t my = (hw >> 4) & 3u;
      int same = 0;
      const uint32_t simds = bg_lds_ld(&s_ctl[2]);   // (posted behind the prologue's barrier)
      for (int k = 0; k < NOW + NSV; k++) same += (((simds >> (2 * k)) & 3u) == my) ? 1 : 0;
      const int key = same >= 3 ? 3 : same;
      atomicAdd(&d.dbg[16 + 4 * (key - 1) + 0], wall_clock64() - e3_w0);
      atomicAdd(&d.dbg[16 + 4 * (key - 1) + 1], e3t_[4]);
      atomicAdd(&d.dbg[16 + 4 * (key - 1) + 2], 1ull);
    }
    if constexpr (kE3Tl) if (lane == 0 && d.dbg) atomicAdd(&d.dbg[29], wall_clock64() - e3_k0);   // owner loop end, summed over owner waves
#pragma unroll
    for (int s = 0; s < KS; s++) { const int l = (wave * KS + s) * BG_BLOCK + lane; s_c34[0][l] = rc3[s]; s_c34[1][l] = rc4[s]; }   // (the epilogue stores them)
    if (lane == 0) __hip_atomic_fetch_sub(&s_owners_left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (a.stats) bg_stats_wave(L.s_stats, n_steps, 0, 0, 0, rbits, ohash);
  } else {
    // ============================================================== SERVICE wave: batches from the two request queues (bg_engine.h's service batch)
    __builtin_amdgcn_s_setprio(3);
    const int sidx = wave - NOW;
    uint64_t n_eps = 0, n_plays = 0;
    int64_t ssum = 0;
    uint32_t polls = 0;
    bool seen = false;
    unsigned long long t_seen = 0;
    E3T_DECL();
    for (;;) {
      // heads first, then tails (bg_engine.h: a tail older than its head reads as four billion items)
      asm volatile("" ::: "memory");
      const bg_u32x4 ch = *(volatile __attribute__((address_space(3))) bg_u32x4*)&s_ctl[4];
      asm volatile("" ::: "memory");
      const bg_u32x4 ct = *(volatile __attribute__((address_space(3))) bg_u32x4*)&s_ctl[0];
      asm volatile("" ::: "memory");
      const uint32_t hp = __builtin_amdgcn_readfirstlane(ch.x), ho = __builtin_amdgcn_readfirstlane(ch.y);
      auto queued = [&](uint32_t tail, uint32_t head) -> uint32_t { const uint32_t k = tail - head; return k <= (uint32_t)NE ? k : 0u; };
      const uint32_t np = queued(__builtin_amdgcn_readfirstlane(ct.x), hp), no = queued(__builtin_amdgcn_readfirstlane(ct.y), ho);
      if ((np | no) == 0u) {
        if (__builtin_amdgcn_readfirstlane(bg_lds_ld(&s_owners_left)) == 0u) break;   // every owner wave is through: nothing can arrive any more
        __builtin_amdgcn_s_sleep(2);
        if (++polls > BG_SPIN_LIMIT) { if (lane == 0) atomicOr(d.err, BG_DEVERR_SPIN); break; }
        continue;
      }
      // A batch costs its instruction stream whatever its lane count, and with the cheap steps on the owner waves the service waves are what a long
      // launch is bound by: a queue is served once it holds th_play / th_other requests, or once the wave has looked at a shorter one for th_more ticks
      int cls;
      if (np >= a.th_play) cls = 0;
      else if (no >= a.th_other) cls = 1;
      else {
        const unsigned long long now = wall_clock64();
        if (!seen) { seen = true; t_seen = now; }
        if (now - t_seen < (unsigned long long)a.th_more) { __builtin_amdgcn_s_sleep(1); continue; }
        cls = np >= no ? 0 : 1;   // the fuller queue (plays on a tie: the longer chains)
      }
      seen = false;
      const uint32_t head = cls ? ho : hp, navail = cls ? no : np;
      const uint32_t nb = navail > BG_BLOCK ? BG_BLOCK : navail;
      uint32_t item = 0;
      uint32_t* const slotp = &s_q[cls][(head + (uint32_t)lane) & (NE - 1)];
      {
        uint32_t got = 0;
        if ((uint32_t)lane < nb) item = bg_lds_ld(slotp);
        if (lane == 0) got = atomicCAS(&s_ctl[4 + cls], head, head + nb) == head ? 1u : 0u;
        if (__builtin_amdgcn_readfirstlane(got) == 0u) continue;
      }
      polls = 0;
      E3T(0);
      if ((uint32_t)lane < nb) {
        uint32_t spin = 0;
        const uint32_t want = BG_ITEM_VALID | ((((head + (uint32_t)lane) >> LNE) & 0xffu) << 8);
        while ((item & (BG_ITEM_VALID | 0xff00u)) != want && ++spin < BG_SPIN_LIMIT) { __builtin_amdgcn_s_sleep(1); item = bg_lds_ld(slotp); }
        if ((item & (BG_ITEM_VALID | 0xff00u)) != want) atomicOr(d.err, BG_DEVERR_SPIN);
        else {
          const int l = (int)(item & 0xffu), env = env0 + l;
          const int action = (int)((item >> 16) & 0x7fffu);
          uint64_t mask = s_mask[l];
          BG_PROBE_BEGIN();
          uint4 c[BG_NHOT];
#pragma unroll
          for (int k = 0; k < BG_NHOT; k++) if (k != 3 && k != 4) c[k] = d.hot[(size_t)k * N + env];
          c[3] = s_c34[0][l]; c[4] = s_c34[1][l];
          DeckT dk; static_cast<Deck0&>(dk) = bg_load_deck0(d, env);
          RngWin w;
          bg_win_init(w, &s_win[sidx][0][lane], &jt);
          Env e;
          bg_unpack(c, e);
          BG_PROBE(23);
          bg_derive_ready(e, s_prod[l]);
          ShopRegs sr; sr.valid = false;
          StepOut o;
          bg_step_init(o);
          o.bd_dst = nullptr;
          if (bg_step_guards(e, mask, action, o)) bg_env_dispatch(d, env, e, w, sr, dk, action, o);
          BG_PROBE(cls == 0 ? 20 : 21);
          if (e.max_ante > 0 && e.ante > e.max_ante) { o.terminated = true; o.flags |= 256; }
          if (o.terminated) n_eps++;
          if (o.terminated && a.autoreset) {
            const bool have = bg_lds_ld(&s_ndst[l]) == 1u;
            bg_env_reset(d, env, e, dk, nullptr, have ? (lds_cu4*)&s_nd[0][l] : (lds_cu4*)nullptr, NE);
            // the slot after it, if the ring (as this launch may see it) holds one: the env's owner fetches it
            bg_lds_st(&s_ndst[l], e.d_ready > 0 ? (2u | ((uint32_t)e.d_head << 8)) : 0u);
          }
          if constexpr (CARDS) bg_vm_drain(); // card states / lazy streams in HBM are edited from any service wave: let the stores land
          BG_PROBE(24);
          mask = bg_action_mask(d, env, e, sr);
          BG_PROBE(25);
          // (the record goes to the env's IMAGE and nowhere else: an all-null ObsPtrs, as in the prologue.  Handed the launch's own -- whose `rows` the
          //  compiler cannot know to be set -- the per-key writer's 31 pointers stayed live in scalar registers through the whole service step.)
          const ObsPtrs none{};
          bg_write_obs_impl<false, 3>(d, env, 0, e, dk, none, mask, sr, RowExtra{o.reward, action, o.terminated ? 1u : 0u}, RowStage{(lds_u4*)&s_img[l][0], nullptr});
          BG_PROBE(26);
          bg_pack(e, c);
#pragma unroll
          for (int k = 0; k < BG_NHOT; k++) if (k != 3 && k != 4) d.hot[(size_t)k * N + env] = c[k];
          s_c34[0][l] = c[3]; s_c34[1][l] = c[4];
          s_mask[l] = mask;
          if (o.hand_type >= 0) { n_plays++; ssum += o.final_score; }
          // the answer, behind everything above (one wave's LDS operations execute in program order)
          bg_lds_st(&s_ans[l], bg_lds_ld(&s_ans[l]) + 1u);
          BG_PROBE(27);
        }
      }
      E3T(1 + cls); E3T_CNT(4 + cls, 1); E3T_CNT(6 + cls, nb);
    }
    E3T_FLUSH(8);
    if constexpr (kE3Tl) if (lane == 0 && d.dbg) atomicAdd(&d.dbg[30], wall_clock64() - e3_k0);   // service loop end, summed over service waves
    if (a.stats) bg_stats_wave(L.s_stats, 0, n_eps, n_plays, ssum, 0, 0);
  }
  // ---------------------------------------------------------------- epilogue: chunks 3 / 4 -> HBM
  BG_PROBE_FLUSH(d);
  __syncthreads();
  for (int l = tid; l < n_live; l += (NOW + NSV) * BG_BLOCK) {
    d.hot[(size_t)3 * N + env0 + l] = s_c34[0][l];
    d.hot[(size_t)4 * N + env0 + l] = s_c34[1][l];
  }
  if (a.stats) bg_stats_flush(a.stats, L.s_stats, tid);
  if constexpr (kE3Tl) if (tid == 0 && d.dbg) atomicAdd(&d.dbg[31], wall_clock64() - e3_k0);   // workgroup end
}
