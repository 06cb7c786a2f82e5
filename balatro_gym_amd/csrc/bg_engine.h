// bg_engine.h -- the step engine: ONE kernel behind bg_step, bg_step_many, bg_rollout and bg_rollout_rows.
//
// A workgroup owns NE envs (256: one workgroup per CU; 128 for small jobs) whose whole state lives in LDS for the launch
// (hot chunks, deck, shop inventory, the carried observation values, the step counter).  Its waves are WORKERS, not owners:
// each pulls a BATCH of up to 64 envs that need the same kind of work from one of three LDS queues and completes one step for
// every env of the batch -- action, state change, SAME_STEP auto-reset, action mask, observation record, reward / terminated /
// info, statistics -- then hands the envs on:
//   run queue    envs whose next step is due: counter-hash policy (or the caller's action), guards (balatro_env_2.py:619-627);
//                card-select toggles and shop-end are applied on the spot; PLAY_HAND goes to the play queue and every other
//                action to the other queue (the state is untouched, so queueing costs one LDS word)
//   play queue   PLAY_HAND (bg_step_play_hand: classification, joker chain, reward shaping, blind outcome, shop generation)
//   other queue  DISCARD, blind select / skip, shop buy / reroll / sell, consumables
// Why: with lane = env for the whole launch (the service-wave kernel before this one) a lane whose action is queued idles
// until it is served, so an env-wave iteration ran with ~35 of 64 lanes and 1.9 iterations per step, and the service waves
// ran batches of ~21 / ~6 items; every instruction was paid at a third to a half of the lanes.  Here a blocked env blocks
// nobody: batches are as full as the queues allow (a wave prefers a full batch of any kind over a partial one), and the wave
// that steps an env also finishes the step, so there is no hand-back, no polling and no second pass over the env.
// The per-env step counters drift apart exactly as before (row = env + t * N; envs are independent).
// No workgroup barrier inside the loop; queues and counters are LDS words (one wave's LDS operations execute in program
// order, so "data, then flag" needs compiler barriers only); every spin is bounded (sticky device error instead of a hang).
#pragma once

#define BG_Q_RUN 0
#define BG_Q_PLAY 1
#define BG_Q_OTHER 2
#define BG_ITEM_VALID 0x80000000u
#define BG_SPIN_LIMIT (1u << 24)
#define BG_DEVERR_SPIN 16u

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
  uint32_t autoreset;          // SAME_STEP auto-reset of terminated envs
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
__device__ __forceinline__ void bg_vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// INFO: the launch serves bg_step / bg_step_many (per-step info arrays, actions from the caller); false for the rollouts
template <bool HASH, bool CARDS, bool INFO, int NE>
__global__ __launch_bounds__(NE * 2, 2) void bg_engine_kernel(BgDev d, EngineArgs a) {
  static_assert(NE == 128 || NE == 256, "128 or 256 envs per workgroup");
  constexpr int NW = NE / 32;                 // waves per workgroup (two per SIMD at NE = 256)
  __shared__ uint4 s_state[BG_NHOT][NE];
  __shared__ uint4 s_shop[4][NE];
  __shared__ uint32_t s_deck[16][NE];
  __shared__ unsigned long long s_handb[NE], s_mask[NE];
  __shared__ float s_prf[NE];
  __shared__ uint32_t s_selm[NE], s_t[NE], s_prod[NE];
  __shared__ uint32_t s_q[3][NE];             // rings of env lanes (| action << 16 | VALID)
  __shared__ uint32_t s_tail[3], s_head[3];   // items ever queued / ever claimed per queue
  __shared__ uint32_t s_done;                 // envs that have finished their T steps
  __shared__ uint32_t s_busy;                 // waves inside a batch
  __shared__ bg_u32x4 s_scratch[NW][BG_BLOCK * 8]; // per wave: record staging (8 KB), and the RNG window of a service batch (6 KB) before it
  __shared__ unsigned long long s_rowaddr[NW][BG_BLOCK];
  __shared__ JTables jt;
  __builtin_amdgcn_s_setprio(3);
  BG_PROBE_INIT();
  bg_tables_load(&jt, d.jtab);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int env0 = blockIdx.x * NE;
  const int n_live = d.N - env0 < NE ? d.N - env0 : NE;
  using DeckT = DeckLdsS<NE, CARDS>;
  // ---------------------------------------------------------------- prologue: HBM -> LDS, lane = env
  if (tid < 3) { s_tail[tid] = tid == BG_Q_RUN ? (uint32_t)n_live : 0u; s_head[tid] = 0; }
  if (tid == 0) { s_done = 0; s_busy = 0; }
  if (tid < NE) {
    const int l = tid, env = env0 + l;
    s_q[BG_Q_PLAY][l] = 0; s_q[BG_Q_OTHER][l] = 0;
    s_q[BG_Q_RUN][l] = l < n_live ? ((uint32_t)l | BG_ITEM_VALID) : 0u;
    s_t[l] = 0;
    if (l < n_live) {
      uint4 c[BG_NHOT];
#pragma unroll
      for (int k = 0; k < BG_NHOT; k++) { c[k] = d.hot[(size_t)k * d.N + env]; s_state[k][l] = c[k]; }
      DeckT dk; dk.col = (lds_u32*)&s_deck[0][l];
#pragma unroll
      for (int k = 0; k < BG_NDECK; k++) bg_deck_set(dk, k, d.deck[(size_t)k * d.N + env]);
      const uint32_t prod = d.prod_view ? d.prod_view[env] : 0u;
      s_prod[l] = prod;
      Env e;
      bg_unpack(c, e);
      bg_derive_ready(e, prod);
      ShopRegs sr; sr.valid = false;
      const uint64_t mask = bg_action_mask(d, env, e, sr);
      if (sr.valid) { s_shop[0][l] = sr.c3; s_shop[1][l] = sr.c4; s_shop[2][l] = sr.c5; s_shop[3][l] = sr.c6; }
      s_mask[l] = mask; s_handb[l] = bg_obs_handb(d, env, e, dk); s_prf[l] = bg_obs_prf(e); s_selm[l] = bg_obs_selm(e);
    }
  }
  __syncthreads();
  // ---------------------------------------------------------------- worker loop
  uint64_t n_steps = 0, n_eps = 0, n_plays = 0, rbits = 0, ohash = 0;
  int64_t ssum = 0;
  const uint32_t bmod3 = (uint32_t)(a.env_index0 % 3ull);
  uint32_t polls = 0;
#ifdef BG_TIMING4
  unsigned long long q_batches[3] = {0, 0, 0}, q_items[3] = {0, 0, 0}, q_busy[3] = {0, 0, 0}, q_idle = 0, q_fail = 0, q_sec[3][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
  const unsigned long long q_t0 = __builtin_readcyclecounter();
#define BG_Q(k) do { const unsigned long long _n = __builtin_readcyclecounter(); q_sec[cls][k] += _n - q_mark; q_mark = _n; } while (0)
#else
#define BG_Q(k) do {} while (0)
#endif
  for (;;) {
#ifdef BG_TIMING4
    const unsigned long long q_l0 = __builtin_readcyclecounter();
#endif
    // -- pick a queue: a full batch of runnable envs first (it feeds the others), then a service queue at its threshold, then
    //    whatever there is (work conserving)
    const uint32_t hr = __builtin_amdgcn_readfirstlane(bg_lds_ld(&s_head[BG_Q_RUN])), hp = __builtin_amdgcn_readfirstlane(bg_lds_ld(&s_head[BG_Q_PLAY])),
                   ho = __builtin_amdgcn_readfirstlane(bg_lds_ld(&s_head[BG_Q_OTHER]));
    const uint32_t nr = __builtin_amdgcn_readfirstlane(bg_lds_ld(&s_tail[BG_Q_RUN])) - hr, np = __builtin_amdgcn_readfirstlane(bg_lds_ld(&s_tail[BG_Q_PLAY])) - hp,
                   no = __builtin_amdgcn_readfirstlane(bg_lds_ld(&s_tail[BG_Q_OTHER])) - ho;
    int cls = -1;
    if (nr >= a.th_run) cls = BG_Q_RUN;
    else if (np >= a.th_play) cls = BG_Q_PLAY;
    else if (no >= a.th_other) cls = BG_Q_OTHER;
    else if (nr | np | no) {
      // nothing is full.  While other waves are inside batches their envs will be back in a moment, so a small batch now only
      // costs instructions at a low lane count; when no wave is busy nothing will ever arrive: take what there is.
      const uint32_t busy = __builtin_amdgcn_readfirstlane(bg_lds_ld(&s_busy));
      const uint32_t need = busy ? a.th_part : 1u;
      if (nr >= need && nr >= np && nr >= no) cls = BG_Q_RUN;
      else if (np >= need && np >= no) cls = BG_Q_PLAY;
      else if (no >= need) cls = BG_Q_OTHER;
      else if (nr >= need) cls = BG_Q_RUN;
      else if (np >= need) cls = BG_Q_PLAY;
    }
    if (cls < 0) {
      if (__builtin_amdgcn_readfirstlane(bg_lds_ld(&s_done)) >= (uint32_t)n_live) break; // every env has done its T steps
      __builtin_amdgcn_s_sleep(8);
      if (++polls > BG_SPIN_LIMIT) { if (lane == 0) atomicOr(d.err, BG_DEVERR_SPIN); break; }
#ifdef BG_TIMING4
      q_idle += __builtin_readcyclecounter() - q_l0;
#endif
      continue;
    }
    const uint32_t head = cls == BG_Q_RUN ? hr : (cls == BG_Q_PLAY ? hp : ho), navail = cls == BG_Q_RUN ? nr : (cls == BG_Q_PLAY ? np : no);
    const uint32_t nb = navail > BG_BLOCK ? BG_BLOCK : navail;
    {
      uint32_t got = 0;
      if (lane == 0) { got = atomicCAS(&s_head[cls], head, head + nb) == head ? 1u : 0u; if (got) __hip_atomic_fetch_add(&s_busy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      if (__builtin_amdgcn_readfirstlane(got) == 0u) {
#ifdef BG_TIMING4
        q_fail++;
#endif
        continue;
      }
    }
    polls = 0;
#ifdef BG_TIMING4
    unsigned long long q_mark = __builtin_readcyclecounter();
    const unsigned long long q_b0 = q_mark;
#endif
    if ((uint32_t)lane < nb) {
      uint32_t* slotp = &s_q[cls][(head + (uint32_t)lane) & (NE - 1)];
      uint32_t item = bg_lds_ld(slotp);
      uint32_t spin = 0;
      while (!(item & BG_ITEM_VALID) && ++spin < BG_SPIN_LIMIT) { __builtin_amdgcn_s_sleep(1); item = bg_lds_ld(slotp); }
      bg_lds_st(slotp, 0u);
      if (!(item & BG_ITEM_VALID)) atomicOr(d.err, BG_DEVERR_SPIN);
      else {
        const int l = (int)(item & 0xffffu), env = env0 + l;
        uint4 c[BG_NHOT];
#pragma unroll
        for (int k = 0; k < BG_NHOT; k++) c[k] = s_state[k][l];
        Env e;
        bg_unpack(c, e);
        bg_derive_ready(e, s_prod[l]);
        DeckT dk; dk.col = (lds_u32*)&s_deck[0][l];
        const uint32_t t = s_t[l];
        uint64_t mask = s_mask[l];
        BG_Q(0); // claim + state load / unpack
        StepOut o;
        bg_step_init(o);
        int action;
        bool fin = true;                 // the step completes in this batch (false: queued for a service batch)
        bool heavy = false;              // handb / prf / selm must be recomputed
        ShopRegs sr; sr.valid = false;
        if (cls == BG_Q_RUN) {
          if (a.actions_in) action = a.actions_in[(size_t)t * (size_t)d.N + env];
          else {
            PolicyLane pl;
            const uint64_t gi = a.env_index0 + (uint64_t)env;
            pl.seed_env = a.policy_seed + 0x9E3779B97F4A7C15ull * (gi + 1);
            pl.blind = a.policy == 2 ? 45 + (int)((bmod3 + (uint32_t)env % 3u) % 3u) : 45;
            action = bg_policy_action_fast(e, mask, a.policy, pl, pl.seed_env + BG_POLICY_PSI * (a.t0 + (uint64_t)t + 1), (lds_JTables*)&jt);
          }
          if (bg_step_guards(e, mask, action, o)) {
            if (e.phase == 0 && action >= 2 && action < 10) { bg_toggle_select(e, action - 2); s_selm[l] ^= 1u << (action - 2); }
            else if (e.phase == 1 && action == 31) {                                            // shop end :1247-1251
              const int nh0 = e.nhand;
              e.phase = 0; bg_draw_cards(e);
              if (e.nhand != nh0) s_handb[l] = bg_obs_handb(d, env, e, dk); // rare: the played cards never left the hand
            } else {
              const int q = (e.phase == 0 && action == 0) ? BG_Q_PLAY : BG_Q_OTHER;
              const uint32_t slot = __hip_atomic_fetch_add(&s_tail[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              bg_lds_st(&s_q[q][slot & (NE - 1)], (uint32_t)l | ((uint32_t)action << 16) | BG_ITEM_VALID);
              fin = false;
            }
          }
        } else {
          action = (int)((item >> 16) & 0x7fffu);
          if (e.phase == 1 && (e.bflags & BG_BF_SHOP_EXISTS)) { sr.c3 = s_shop[0][l]; sr.c4 = s_shop[1][l]; sr.c5 = s_shop[2][l]; sr.c6 = s_shop[3][l]; sr.valid = true; }
          RngWin w;
          bg_win_init(w, (uint32_t*)&s_scratch[wave][0] + lane, &jt);
          { BG_PROBE_BEGIN(); bg_env_dispatch(d, env, e, w, sr, dk, action, o); BG_PROBE(cls == BG_Q_PLAY ? 20 : 21); }
          if (sr.valid) { s_shop[0][l] = sr.c3; s_shop[1][l] = sr.c4; s_shop[2][l] = sr.c5; s_shop[3][l] = sr.c6; }
          heavy = true;
        }
        BG_Q(1); // policy + guards + cheap action / dispatch
        if (fin) {
          // ---- finish the step: curriculum cap, SAME_STEP auto-reset, mask, observation, outputs, statistics
          if (e.max_ante > 0 && e.ante > e.max_ante) { o.terminated = true; o.flags |= 256; }
          bool did_reset = false;
          if (o.terminated && a.autoreset) { bg_env_reset(d, env, e, dk); n_eps++; did_reset = true; if (INFO) o.flags |= BG_INFO_AUTORESET; }
          else if (o.terminated) n_eps++;
          // a reset zeroes the env's play counts (and re-applies its card states) in HBM, which another wave's play touches a
          // few steps later; card-state builds also edit card states / lazy streams in HBM from the service batches: let
          // those stores land before the env is handed on.  What else is in flight is this wave's previous record write-out.
          if (CARDS ? (heavy || __ballot(did_reset) != 0ull) : (__ballot(did_reset) != 0ull)) bg_vm_drain();
          uint64_t handb; float prf; uint32_t selm;
          if (did_reset) { handb = ~0ull; prf = 0.0f; selm = 0u; }
          else if (heavy) { handb = bg_obs_handb(d, env, e, dk); prf = bg_obs_prf(e); selm = bg_obs_selm(e); }
          else { handb = s_handb[l]; prf = s_prf[l]; selm = s_selm[l]; }
          if (!sr.valid && e.phase == 1 && (e.bflags & BG_BF_SHOP_EXISTS)) { sr.c3 = s_shop[0][l]; sr.c4 = s_shop[1][l]; sr.c5 = s_shop[2][l]; sr.c6 = s_shop[3][l]; sr.valid = true; }
          mask = bg_action_mask(d, env, e, sr);
          BG_Q(2); // cap, reset, carried values, mask
          const size_t row = (size_t)env + (a.obs_stride_steps ? (size_t)t * (size_t)d.N : 0);
          const uint64_t h = bg_write_obs_impl<HASH, 2>(d, env, row, e, dk, a.obs, mask, sr, RowExtra{o.reward, action, o.terminated ? 1u : 0u, true, prf, handb, selm},
                                                       RowStage{(lds_u4*)&s_scratch[wave][0], (lds_u64*)&s_rowaddr[wave][0]});
          BG_Q(3); // record
          if (HASH) ohash ^= h * (0x9E3779B97F4A7C15ull + 2 * (uint64_t)(a.t0 + t)) + (a.env_index0 + (uint64_t)env);
          if (a.reward) a.reward[row] = o.reward;
          if (a.term) a.term[row] = o.terminated ? 1 : 0;
          if (a.actions_out) a.actions_out[row] = action;
          if constexpr (INFO) bg_emit_info(row, o, a.trunc, a.info);
          n_steps++;
          rbits ^= (uint64_t)__double_as_longlong(o.reward) * (2 * (uint64_t)(a.t0 + t) + 1);
          if (o.hand_type >= 0) { n_plays++; ssum += o.final_score; }
          // ---- state back to LDS, env back to the run queue (or done)
          bg_pack(e, c);
#pragma unroll
          for (int k = 0; k < BG_NHOT; k++) s_state[k][l] = c[k];
          s_mask[l] = mask; s_handb[l] = handb; s_prf[l] = prf; s_selm[l] = selm; s_t[l] = t + 1;
          if (t + 1 < (uint32_t)a.T) {
            const uint32_t slot = __hip_atomic_fetch_add(&s_tail[BG_Q_RUN], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            bg_lds_st(&s_q[BG_Q_RUN][slot & (NE - 1)], (uint32_t)l | BG_ITEM_VALID);
          } else __hip_atomic_fetch_add(&s_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          BG_Q(4); // outputs, statistics, state store, requeue
        }
      }
    }
    if (lane == 0) __hip_atomic_fetch_sub(&s_busy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // after the pushes of this batch
#ifdef BG_TIMING4
    q_batches[cls]++; q_items[cls] += nb; q_busy[cls] += __builtin_readcyclecounter() - q_b0;
#endif
  }
#ifdef BG_TIMING4
  if (lane == 0 && d.dbg) {
    atomicAdd(&d.dbg[0], __builtin_readcyclecounter() - q_t0); atomicAdd(&d.dbg[1], 1ull);
    for (int c = 0; c < 3; c++) { atomicAdd(&d.dbg[2 + 3 * c], q_batches[c]); atomicAdd(&d.dbg[3 + 3 * c], q_items[c]); atomicAdd(&d.dbg[4 + 3 * c], q_busy[c]); }
    atomicAdd(&d.dbg[11], q_idle); atomicAdd(&d.dbg[12], q_fail);
    for (int c = 0; c < 3; c++) for (int k = 0; k < 5; k++) atomicAdd(&d.dbg[16 + 5 * c + k], q_sec[c][k]);
  }
#endif
  // ---------------------------------------------------------------- epilogue: LDS -> HBM, statistics
  BG_PROBE_FLUSH(d);
  __syncthreads();
  if (tid < n_live) {
#pragma unroll
    for (int k = 0; k < BG_NHOT; k++) d.hot[(size_t)k * d.N + env0 + tid] = s_state[k][tid];
  }
  if (a.stats) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      n_steps += __shfl_down(n_steps, off); n_eps += __shfl_down(n_eps, off); n_plays += __shfl_down(n_plays, off);
      ssum += __shfl_down(ssum, off); rbits ^= __shfl_down(rbits, off); ohash ^= __shfl_down(ohash, off);
    }
    if (lane == 0) {
      atomicAdd((unsigned long long*)&a.stats->steps, (unsigned long long)n_steps);
      atomicAdd((unsigned long long*)&a.stats->episodes, (unsigned long long)n_eps);
      atomicAdd((unsigned long long*)&a.stats->plays, (unsigned long long)n_plays);
      atomicAdd((unsigned long long*)&a.stats->score_sum, (unsigned long long)ssum);
      atomicXor((unsigned long long*)&a.stats->reward_bits, (unsigned long long)rbits);
      atomicXor((unsigned long long*)&a.stats->obs_hash, (unsigned long long)ohash);
    }
  }
}
