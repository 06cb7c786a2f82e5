// bg_lib.hip -- kernels + C ABI of libbalatro_mi355x.so (gfx950 / CDNA4 only; see include/balatro_mi355x.h).
//
// Kernels (one lane = one env, env index fastest in every array => every wave access is one coalesced request):
//   bg_engine_kernel   (bg_engine.h) THE step path: bg_step / bg_step_many / bg_rollout / bg_rollout_rows
//   bg_reset_kernel    masked reset() + observation
//   bg_observe_kernel  observation only
//   bg_seed_kernel     DeterministicRNG(seed): CPython init_by_array for streams 0, 2 and the per-env global stream
//   bg_refill_*_kernel RNG look-ahead: pre-shuffled decks, pre-seeded shop streams, next global-stream blocks (beside the engine)
//   bg_inject_kernel   harness injection into live state
// There is no CPU path: every entry point needs the HIP device bg_create() opened.
#include "../../include/balatro_mi355x.h"
#include "bg_step.h"
#include <hip/hip_ext.h>   // hipExtLaunchKernelGGL: a launch whose own dispatch packet carries the start / stop timestamps of the profile
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <string>
#include <vector>


struct InfoPtrs {
  int64_t* final_score; int32_t* error; int32_t* flags; int32_t* aux; int8_t* hand_type; int8_t* cards_played;
  double* reward_terms;
  double* score_breakdown;
};

static_assert(offsetof(ObsPtrs, rows) == sizeof(bg_obs_ptrs), "the first 31 members of ObsPtrs must mirror bg_obs_ptrs");
static_assert(sizeof(InfoPtrs) == sizeof(bg_info_ptrs), "InfoPtrs must mirror bg_info_ptrs");

// ---------------------------------------------------------------------------------------------------------
// step / rollout / reset / observe
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bg_emit_info(size_t row, const StepOut& o, uint8_t* trunc, const InfoPtrs& info) {
  if (trunc) trunc[row] = 0; // the reference never truncates (balatro_env_2.py:1064)
  if (info.final_score) info.final_score[row] = o.final_score;
  if (info.error) info.error[row] = o.error;
  if (info.flags) info.flags[row] = o.flags;
  if (info.aux) info.aux[row] = o.aux;
  if (info.hand_type) info.hand_type[row] = (int8_t)o.hand_type;
  if (info.cards_played) info.cards_played[row] = (int8_t)o.cards_played;
  if (info.reward_terms) {
    double2* q = (double2*)(info.reward_terms + row * 8);
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = make_double2(o.terms[2 * i], o.terms[2 * i + 1]);
  }
  if (info.score_breakdown && o.hand_type < 0) { // an accepted play stored its row itself (bg_step_play_hand); every other step: zeros
    double2* q = (double2*)(info.score_breakdown + row * 8);
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = make_double2(0.0, 0.0);
  }
}

#include "bg_engine.h" // the step engine: one kernel behind bg_step / bg_step_many / bg_rollout / bg_rollout_rows
#include "bg_engine3.h" // owner waves + service waves in ONE workgroup (packed-record rollouts)

template <bool CARDS>
__global__ __launch_bounds__(BG_BLOCK) void bg_reset_kernel(BgDev d, const uint8_t* __restrict__ mask_in, ObsPtrs obs) {
  using DeckT = typename std::conditional<CARDS, Deck0C, Deck0>::type;
  int env = blockIdx.x * BG_BLOCK + threadIdx.x;
  if (env >= d.N) return;
  Env e;
  bg_load_env(d, env, e);
  DeckT dk; static_cast<Deck0&>(dk) = bg_load_deck0(d, env);
  if (!mask_in || mask_in[env]) { bg_env_reset(d, env, e, dk); bg_store_env(d, env, e); }
  ShopRegs sr; sr.valid = false;
  uint64_t mask = bg_action_mask(d, env, e, sr);
  bg_write_obs<false>(d, env, (size_t)env, e, dk, obs, mask, sr, RowExtra{0.0, 0, 0u});
}

__global__ __launch_bounds__(BG_BLOCK) void bg_observe_kernel(BgDev d, ObsPtrs obs) {
  int env = blockIdx.x * BG_BLOCK + threadIdx.x;
  if (env >= d.N) return;
  Env e;
  bg_load_env(d, env, e);
  Deck0 dk = bg_load_deck0(d, env);
  ShopRegs sr; sr.valid = false;
  uint64_t mask = bg_action_mask(d, env, e, sr);
  bg_write_obs<false>(d, env, (size_t)env, e, dk, obs, mask, sr, RowExtra{0.0, 0, 0u});
}

// ---------------------------------------------------------------------------------------------------------
// MT19937 on a lane-private contiguous state block (word i at p[i]; see bg_device.h for why this is not SoA)
// ---------------------------------------------------------------------------------------------------------
// CPython random_seed()/init_by_array() for a ONE-word key (every stream / shop / global seed is < 2**32).
// The key-dependent passes are 1247 DEPENDENT steps per stream.  Memory stays out of the chains completely: pass 1
// (i = 1..623 plus the wrapped step at i = 1) runs in registers and keeps only its last value; pass 2 needs pass 1's
// mt[i] again, and gets it from a SECOND run of the pass-1 recurrence in lockstep (two independent chains = ILP) --
// recomputing 623 steps is far cheaper than writing 2.5 KB per stream and reading it back.  The only memory traffic
// is the final state: 156 16-byte stores.
// init_genrand(19650218), the key-independent state init_by_array starts from: a compile-time table read with SCALAR loads (the
// index is the loop counter, uniform over the wave) instead of a third multiply chain per step -- sixteen entries per load
// (`s_load_dwordx16`), the next block requested before the current one is used: with one dword per step each of the 1 870 steps
// waited ~200 cycles for its own scalar load and the seeding of a shop stream was latency bound on the constant cache (0.58 ms
// per refill for 1.1 M streams against a VALU bound of ~0.15).
struct alignas(64) BgG16 { uint32_t v[16]; };
struct BgGenrandTab {
  BgG16 blk[BG_MT_N / 16];
  constexpr BgGenrandTab() : blk{} {
    uint32_t x = 19650218u;
    blk[0].v[0] = x;
    for (int i = 1; i < BG_MT_N; i++) { x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i; blk[i / 16].v[i % 16] = x; }
  }
};
static_assert(BG_MT_N % 16 == 0, "624 = 39 blocks of 16");
static constexpr BgGenrandTab BG_GENRAND{};

// SLOT = false: the whole seeded state (156 16-byte stores).  SLOT = true: a shop-stream ring slot -- the first BG_SW_T OUTPUT words of
// the stream, tempered, then the TOP BYTES of output words 0..23 packed into six words (BG_SW_PK: all a fresh inventory looks at, bg_shop_inventory --
// two 16-byte pieces of ONE line instead of six of two), then the seed (bg_device.h): output word k = temper(S[k+397] ^ twist(S[k], S[k+1])).  The pass produces S in index
// order, so the near words S[2..63] are parked RAW in the slot as they appear (blocks 0..3) and, when S[k+397] appears (blocks 24..28),
// read back (they are this lane's own stores of ~20 000 cycles ago), combined, tempered and written over S[k] -- which nothing needs any
// more.  S[1] is the last word the seeding produces, so outputs 0 and 1 (and the group they share with 2 and 3) are written at the end.
template <bool SLOT>
__device__ __forceinline__ void bg_mt_seed_impl(uint32_t* p, uint32_t key) {
  constexpr int NB = BG_MT_N / 16;
  uint4* p4 = (uint4*)p;
  uint32_t a = BG_GENRAND.blk[0].v[0], a1 = 0; // a: pass-1 recurrence
  {
    BgG16 cur = BG_GENRAND.blk[0];
#pragma unroll 1
    for (int b = 0; b < NB; b++) {
      const BgG16 nxt = BG_GENRAND.blk[b + 1 < NB ? b + 1 : NB - 1];
#pragma unroll
      for (int c = 0; c < 16; c++) {
        if (b > 0 || c >= 1) {
          a = (cur.v[c] ^ ((a ^ (a >> 30)) * 1664525u)) + key;
          if (b == 0 && c == 1) a1 = a;
        }
      }
      cur = nxt;
    }
  }
  // wrap: mt[0] = mt[623]; 624th iteration of pass 1 at i = 1
  const uint32_t a1w = (a1 ^ ((a ^ (a >> 30)) * 1664525u)) + key;
  // pass 2: i = 2..623 (then the wrapped step at i = 1), beside a rerun of pass 1 that supplies mt[i]
  a = BG_GENRAND.blk[0].v[0];
  a = (BG_GENRAND.blk[0].v[1] ^ ((a ^ (a >> 30)) * 1664525u)) + key; // pass-1 mt[1] (before the wrap)
  uint32_t bprev = a1w, w2 = 0, w3 = 0;
  uint32_t far0 = 0, far1 = 0, out2 = 0, out3 = 0;   // SLOT: S[397], S[398]; output words 2 and 3 (group 0 is written last)
  uint32_t acc = 0;   // SLOT: the packed word under construction -- top byte of output word k -> byte k & 3 of slot word BG_SW_PK + (k >> 2), k < 24
  {
    BgG16 cur = BG_GENRAND.blk[0];
#pragma unroll 1
    for (int b = 0; b < NB; b++) {
      const BgG16 nxt = BG_GENRAND.blk[b + 1 < NB ? b + 1 : NB - 1];
      uint32_t v[16];
#pragma unroll
      for (int c = 0; c < 16; c++) {
        v[c] = 0;
        if (b > 0 || c >= 2) {
          a = (cur.v[c] ^ ((a ^ (a >> 30)) * 1664525u)) + key;
          bprev = (a ^ ((bprev ^ (bprev >> 30)) * 1566083941u)) - (uint32_t)(16 * b + c);
          v[c] = bprev;
        }
      }
      if (b == 0) { w2 = v[2]; w3 = v[3]; }
      if constexpr (!SLOT) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int q = 4 * b + k; // 16-byte group of the state
          if (q == 0) continue;
          p4[q] = make_uint4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
        }
      } else {
        if (b < 4) { // park S[4..63] raw in the slot (group 0 holds S[0..3]: S[2], S[3] stay in registers)
#pragma unroll
          for (int k = 0; k < 4; k++) if (4 * b + k > 0) p4[4 * b + k] = make_uint4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
        } else if (b >= 24 && b <= 28) {
          // words 16 b .. 16 b + 15 = S[k + 397] for k = 16 b - 397 + c
          const int kb = 16 * b - BG_MT_M;                 // k of c = 0: -13, 3, 19, 35, 51
          if (b == 24) { far0 = v[13]; far1 = v[14]; out2 = bg_temper(bg_twist(w2, w3, v[15])); }
          else {
            // near words S[kb .. kb + 16] back from the slot: five 16-byte groups starting at the group that holds word kb
            const int g0 = kb >> 2;                        // 0, 4, 8, 12 (kb = 3, 19, 35, 51 -> the word is the group's fourth)
            uint32_t nr[20];
#pragma unroll
            for (int g = 0; g < 5; g++) {
              uint4 t = make_uint4(0u, 0u, 0u, 0u);
              if (g0 + g > 0 && g0 + g < BG_SLOT_WORDS / 4) t = p4[g0 + g];   // (not restrict-qualified reads of this lane's own earlier stores)
              nr[4 * g] = t.x; nr[4 * g + 1] = t.y; nr[4 * g + 2] = t.z; nr[4 * g + 3] = t.w;
            }
            if (b == 25) nr[3] = w3;                       // (group 0 is not in memory: S[3])
            // output words kb .. kb + 15 (its near words are nr[3 + c], nr[4 + c]); S[k] is dead: its place takes the output word.  Blocks 25 and 26 also
            // collect the top bytes of words 4..23 into the packed words 1..5, each stored as soon as it is whole (slot words 57..61: the raw S[57..61]
            // parked there are needed by nobody -- block 28 only looks at S[56] of that group); word 0 waits for outputs 0..2 at the end.  b is uniform, so
            // each block is its own copy of the loop with COMPILE-TIME k.  (Six packed words kept to the end cost the kernel its fourth wave beside the
            // step engine -- 69 registers --, and indexed by a run-time k they went to scratch memory: the kernel ran 40 % longer.)
            auto outputs = [&](auto kbc) {
              constexpr int KB = decltype(kbc)::value;
#pragma unroll
              for (int c = 0; c < 16; c++) {
                const int k = KB + c;
                if (k < BG_SW_T) {
                  const uint32_t o = bg_temper(bg_twist(nr[3 + c], nr[4 + c], v[c]));
                  if (k == 3) out3 = o; else p[k] = o;
                  if (k >= 4 && k < 24) {
                    acc |= (o >> 24) << (8 * (k & 3));
                    if ((k & 3) == 3) { p[BG_SW_PK + (k >> 2)] = acc; acc = 0u; }
                  }
                }
              }
            };
            if (b == 25) outputs(std::integral_constant<int, 16 * 25 - BG_MT_M>{});
            else if (b == 26) outputs(std::integral_constant<int, 16 * 26 - BG_MT_M>{});
            else if (b == 27) outputs(std::integral_constant<int, 16 * 27 - BG_MT_M>{});
            else outputs(std::integral_constant<int, 16 * 28 - BG_MT_M>{});
          }
        }
      }
      cur = nxt;
    }
  }
  const uint32_t w1 = (a1w ^ ((bprev ^ (bprev >> 30)) * 1566083941u)) - 1u; // mt[0] = mt[623], wrapped step at i = 1
  if constexpr (!SLOT) p4[0] = make_uint4(0x80000000u, w1, w2, w3);
  else {
    const uint32_t out0 = bg_temper(bg_twist(0x80000000u, w1, far0)), out1 = bg_temper(bg_twist(w1, w2, far1));
    p4[0] = make_uint4(out0, out1, out2, out3);
    // the slot's tail: packed word 0 (the raw S[56] parked in its place has been read back by block 28), seed, padding
    static_assert(BG_SW_PK == 56 && BG_SW_SEED == 62 && BG_SLOT_WORDS == 64, "slot tail layout");
    p[BG_SW_PK] = (out0 >> 24) | ((out1 >> 24) << 8) | ((out2 >> 24) << 16) | (out3 & 0xff000000u);
    *(uint2*)(p + BG_SW_SEED) = make_uint2(key, 0u);
  }
}
__device__ void bg_mt_seed(uint32_t* __restrict__ p, uint32_t key) { bg_mt_seed_impl<false>(p, key); }
__device__ void bg_mt_seed_slot(uint32_t* __restrict__ p, uint32_t key) { bg_mt_seed_impl<true>(p, key); }

// ---------------------------------------------------------------------------------------------------------
// Lazy MT19937 for the streams only the refill kernels read (deck shuffles, shop seeds).  genrand_uint32() regenerates
// all 624 words when the block is exhausted; computing word k of the next block just before it is read gives the same
// sequence (new[k] = old[k+397 mod 624 (already new for k >= 227)] ^ twist(old[k], old[k+1]); k = 623 pairs with the
// new word 0) and never needs the 624-word pass.  State: [new words 0..c) | old words c..624), cursor c in word 624.
// The stream is read BG_LAZY_U 16-byte groups (16 words) per iteration, every lane on its own stream: the groups of the iteration,
// the one behind them (word k+1 of the last word) and the groups holding words k+397 live in registers, the groups of the NEXT
// iteration are requested before the current ones are used (beside the step engine a load takes microseconds), and a group is
// written back as soon as the lane has consumed its words (consumed words new, the others as loaded).  Nothing goes through LDS,
// so the refill kernels fit beside the step engine (which leaves one SIMD's registers and ~5 KB of LDS per CU free).
// No operand is produced less than 227 words before it is read, the look-ahead is 36 words.
// take(y): consumes one untempered word, returns true while the lane wants another.  Returns the new cursor.  Every lane of
// the wave must call this (lanes with active == false touch no memory); the loop runs until the last lane has had enough.
__device__ __forceinline__ uint32_t bg_q156(uint32_t q) { return q >= 156u ? q - 156u : q; }
#define BG_LAZY_U 4 // 16-byte groups per iteration of bg_lazy_stream
#define BG_LAZY_U_SEEDRING 2 // the seed ring draws ~34 words per env and refill: shorter iterations, 70 registers -> three waves beside the engine
#define BG_WL_COUNTERS 32 // refill work-list counters: [0..3] list lengths, [4] the deck kernel's cursor, [8 + p] the cursor of its part p (p < 16)
#define BG_REFILL_WAVE_PRIO 0 // s_setprio of the deck / seed-ring / block waves (the shop seeding stays at 0, the step engine runs at 3)
template <int U = BG_LAZY_U, class F>
__device__ __forceinline__ uint32_t bg_lazy_stream(uint32_t* S, uint32_t c, bool active, F&& take) {
  uint4* S4 = (uint4*)S;
  uint32_t q = c >> 2, off = c & 3u, fq = bg_q156(q + 99u); // fq: group of word (4*q + 396) mod 624
  uint4 a[U + 1], f[U + 1]; // groups q .. q+U and fq .. fq+U
#pragma unroll
  for (int g = 0; g <= U; g++) { a[g] = make_uint4(0, 0, 0, 0); f[g] = a[g]; }
  if (active) {
#pragma unroll
    for (int g = 0; g <= U; g++) { a[g] = S4[bg_q156(q + (uint32_t)g)]; f[g] = S4[bg_q156(fq + (uint32_t)g)]; }
  }
#pragma unroll 1
  while (__ballot(active) != 0ull) {
    uint4 an[U], fn[U]; // the U groups behind them: requested now, used by the next iteration
#pragma unroll
    for (int g = 0; g < U; g++) { an[g] = a[0]; fn[g] = a[0]; }
    if (active) {
#pragma unroll
      for (int g = 0; g < U; g++) { an[g] = S4[bg_q156(q + (uint32_t)(U + 1 + g))]; fn[g] = S4[bg_q156(fq + (uint32_t)(U + 1 + g))]; }
    }
    bool more = active;
#pragma unroll
    for (int g = 0; g < U; g++) {
      const uint4 a0 = a[g], f0 = f[g];
      const uint32_t n0 = bg_twist(a0.x, a0.y, f0.y), n1 = bg_twist(a0.y, a0.z, f0.z), n2 = bg_twist(a0.z, a0.w, f0.w), n3 = bg_twist(a0.w, a[g + 1].x, f[g + 1].x);
      const uint32_t goff = g == 0 ? off : 0u;
      uint4 o = a0;
      uint32_t used = 0;
      if (more && goff == 0u) { o.x = n0; used++; more = take(n0); }
      if (more && goff <= 1u) { o.y = n1; used++; more = take(n1); }
      if (more && goff <= 2u) { o.z = n2; used++; more = take(n2); }
      if (more) { o.w = n3; used++; more = take(n3); }
      if (used) {
        const uint32_t qg = bg_q156(q + (uint32_t)g);
        S4[qg] = o;
        if (!more) { c = 4u * qg + goff + used; if (c >= (uint32_t)BG_MT_N) c -= (uint32_t)BG_MT_N; active = false; }
      }
    }
    q = bg_q156(q + (uint32_t)U); fq = bg_q156(fq + (uint32_t)U); off = 0u;
    a[0] = a[U]; f[0] = f[U];
#pragma unroll
    for (int g = 0; g < U; g++) { a[g + 1] = an[g]; f[g + 1] = fn[g]; }
  }
  return c;
}

#define BG_MTB 16 // words per batch of the block twist
#define BG_SSEED 128 // pre-drawn shop seeds per env (power of two, < 256: the count lives in a byte); >= 2 x the shops a launch can use
#define BG_LAZY_SEEDED 0x80000000u // flag in the cursor word of a lazy stream (set by bg_seed)
// genrand_uint32()'s block regeneration: dst[kk] = twist(src[kk], src[kk+1], kk < 227 ? src[kk+397] : dst[kk-227]),
// last element uses the NEW dst[0].  dst == src gives CPython's in-place update (operands are loaded per batch before
// any element of the batch is stored; BG_MTB < 227 keeps the kk-227 operands already written).
__device__ void bg_mt_twist(const uint32_t* src, uint32_t* dst) {
  uint32_t cur = src[0], first_new = 0;
  const uint4* s4 = (const uint4*)src;
  uint4* d4 = (uint4*)dst;
#pragma unroll 1
  for (int base = 0; base < BG_MT_N; base += BG_MTB) {
    uint32_t nx[BG_MTB + 4], fr[BG_MTB + 4];
    // words base .. base+19 of src: nxt[j] = src[base + j + 1]  (src[624] is the unused index word of the 640-word block)
#pragma unroll
    for (int g = 0; g < BG_MTB / 4 + 1; g++) { uint4 v = s4[base / 4 + g]; nx[4 * g] = v.x; nx[4 * g + 1] = v.y; nx[4 * g + 2] = v.z; nx[4 * g + 3] = v.w; }
    if (base == 224) { // the batch that straddles kk = 227: old src[kk+397] below, new dst[kk-227] from there on
#pragma unroll
      for (int j = 0; j < BG_MTB; j++) { int kk = base + j; fr[j + 1] = (kk < BG_MT_N - BG_MT_M) ? src[kk + BG_MT_M] : dst[kk + BG_MT_M - BG_MT_N]; }
    } else {
      const uint4* f4 = base < 224 ? s4 + (base + BG_MT_M - 1) / 4 : (const uint4*)dst + (base + BG_MT_M - BG_MT_N - 1) / 4;
#pragma unroll
      for (int g = 0; g < BG_MTB / 4 + 1; g++) { uint4 v = f4[g]; fr[4 * g] = v.x; fr[4 * g + 1] = v.y; fr[4 * g + 2] = v.z; fr[4 * g + 3] = v.w; }
    }
    uint32_t v[BG_MTB];
#pragma unroll
    for (int j = 0; j < BG_MTB; j++) {
      uint32_t nxt = (base + j == BG_MT_N - 1) ? first_new : nx[j + 1]; // kk == 623 pairs with the NEW dst[0]
      v[j] = bg_twist(cur, nxt, fr[j + 1]);
      if (base + j == 0) first_new = v[j];
      cur = nxt;
    }
#pragma unroll
    for (int j = 0; j < BG_MTB / 4; j++) d4[base / 4 + j] = make_uint4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
  }
}

#include "bg_ops.h" // operator-level batch kernels (classify / score_hand): need bg_mt_seed, bg_mt_twist
#include "bg_sim.h" // balatro_sim.py evaluator / scorer (operator-level)

// DeterministicRNG(seed) (balatro_env_2.py:84-106) for streams 0 ('deck_shuffle') and 2 ('shop_generation'), plus the
// per-env global stream seeded G(seed).  Streams are seeded `(master + 1000 * i) % 2**32` (:105).
__global__ __launch_bounds__(BG_BLOCK) void bg_seed_kernel(BgDev d, const int64_t* __restrict__ seeds,
                                                          const uint8_t* __restrict__ mask_in, int reseed_global) {
  int env = blockIdx.x * BG_BLOCK + threadIdx.x;
  if (env >= d.N) return;
  if (mask_in && !mask_in[env]) return;
  RngWin w;
  bg_win_init(w, nullptr);
  Env e;
  bg_load_env(d, env, e);
  int64_t seed = seeds[env];
  if (reseed_global) {
    uint32_t gs = (uint32_t)((uint64_t)seed + 16000ull);
    uint32_t* g = bg_gblock(d, env, 0);
    bg_mt_seed(g, gs);
    bg_mt_twist(g, g); // first block = what the first 624 getrandbits(32) read
    e.g_cur = 0; e.g_idx = 0; e.g_cons = 0; e.g_valid = 1;
  }
  if (seed == 0) { // `master_seed or random.randint(0, 2**32 - 1)` (:88): _randbelow(2**32), k = 33 bits
    uint64_t r;
    int guard = 0;
    do {
      uint64_t lo = bg_gdraw(d, env, e, w);
      uint64_t hi = bg_gdraw(d, env, e, w) >> 31;
      r = lo | (hi << 32);
    } while (r >= 4294967296ull && ++guard < 4096);
    seed = (int64_t)(r & 0xffffffffull);
  }
  uint32_t base = (uint32_t)(uint64_t)seed;
  bg_mt_seed(bg_deckmt(d, env), base);
  bg_deckmt(d, env)[BG_MT_N] = BG_LAZY_SEEDED; // lazy streams: cursor 0 (bit 31 marks "seeded" for the refill scan)
  bg_mt_seed(bg_shopgenmt(d, env), base + 2000u);
  bg_shopgenmt(d, env)[BG_MT_N] = BG_LAZY_SEEDED;
  if (d.cardmt) { // 'card_enhancement' is stream 11 of DeterministicRNG (:96-105)
    bg_mt_seed(d.cardmt + (size_t)env * BG_MTS, base + 11000u);
    d.cardmt[(size_t)env * BG_MTS + BG_MT_N] = BG_LAZY_SEEDED;
    bg_mt_seed(d.sealmt + (size_t)env * BG_MTS, base + 13000u); // 'seal_applications' is stream 13
    d.sealmt[(size_t)env * BG_MTS + BG_MT_N] = BG_LAZY_SEEDED;
  }
  // look-ahead rings are functions of the streams: invalidate (producer counters restart at the consumer counters)
  e.d_head = 0; e.d_cons = 0; e.d_ready = 0;
  e.s_cons = 0; e.s_ready = 0;
  d.smeta[env] = 0;
  {
    uint32_t gp = reseed_global ? 1u : (uint32_t)((e.g_cons + e.g_valid) & 0xff);
    uint32_t pv = 0u | (0u << 8) | (gp << 16) | 0x80000000u; // bit 31: this env's streams are seeded (refill scan)
    d.prod_out[env] = pv;
    ((uint32_t*)d.prod_in)[env] = pv;
  }
  bg_store_env(d, env, e);
}

// ---------------------------------------------------------------------------------------------------------
// RNG look-ahead (refill).  Only ~2% of the envs need a new deck / shop stream / global block after a step, and each
// such item is a long SERIAL computation, so the work is compacted first: a scan kernel (lane = env, three 16-byte
// loads) appends the needy envs to per-kind work lists with wave-aggregated atomics; dense kernels (lane = work item)
// then run the serial MT19937 code with every lane busy.
//   * decks : `rng.shuffle('deck_shuffle', deck)` (balatro_env_2.py:525) on stream 0
//   * shops : `shop_seed = rng.get_int('shop_generation', 0, 2**31 - 1)` (:1389), `random.Random(shop_seed)` (shop.py:96)
//   * gblk  : next 624-word block(s) of the per-env global stream
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BG_BLOCK) void bg_refill_zero_kernel(BgDev d) { if (threadIdx.x < BG_WL_COUNTERS) d.wl_count[threadIdx.x] = 0; } // [4]: the deck kernel's list cursor, [8 + p]: the cursor of its part p

// A refill in PIECES (short launches, bg_refill_pieces): a dense kernel is launched once per part and works on part p of q equal parts of its work list
struct BgPart { uint32_t p, q; };
__device__ __forceinline__ uint32_t bg_part_lo(uint32_t count, BgPart pt) { return (uint32_t)(((uint64_t)count * pt.p) / pt.q); }
__device__ __forceinline__ uint32_t bg_part_hi(uint32_t count, BgPart pt) { return (uint32_t)(((uint64_t)count * (pt.p + 1u)) / pt.q); }

__global__ __launch_bounds__(BG_BLOCK) void bg_refill_scan_kernel(BgDev d) {
  int env = blockIdx.x * BG_BLOCK + threadIdx.x;
  if (env >= d.N) return;
  size_t N = d.N;
  // consumer counters (written by step / rollout kernels; each group sits in one 32-bit word, so a concurrent writer can
  // only make this kernel see a NEWER consumer state, which just frees more slots)
  uint32_t w5 = ((const uint32_t*)&d.hot[(size_t)5 * N + env])[3];
  uint32_t w6 = ((const uint32_t*)&d.hot[(size_t)6 * N + env])[3];
  uint2 w7 = *((const uint2*)&d.hot[(size_t)7 * N + env]);
  uint32_t prod = d.prod_in[env];
  int d_cons = bg_b(w5, 3), g_cons = bg_b(w6, 3), s_cur = bg_b(w7.y, 0), s_cons = bg_b(w7.y, 1);
  int d_ready = (int)((prod - (uint32_t)d_cons) & 0xffu);
  int s_ready = (int)(((prod >> 8) - (uint32_t)s_cons) & 0xffu);
  int g_valid = (int)(((prod >> 16) - (uint32_t)g_cons) & 0xffu);
  const bool seeded = (prod >> 31) != 0; // set by bg_seed (a dense word: no trip to the env's 2.5 KB stream block)
  if (!seeded) { d.prod_out[env] = prod; return; }
  if (s_ready < d.KS - 1) {
    // one work item per missing slot (balanced: every lane of the dense kernel seeds exactly one stream); the shop
    // seeds were drawn ahead from stream 2 into a small per-env ring
    uint32_t sm = d.smeta[env];
    int head = (int)(sm & 0xffu), cnt = (int)((sm >> 8) & 0xffu);
    int emitted = d.KS - 1 - s_ready;
    if (emitted > cnt) emitted = cnt;
    if (emitted > 0) { // ONE atomic per env (an atomic per slot was ~0.6 M serialised updates of one counter: 0.4 ms)
      const uint32_t base = atomicAdd(&d.wl_count[3], (uint32_t)emitted);
#pragma unroll 1
      for (int k = 0; k < emitted; k++) {
        int slot = s_cur + 1 + s_ready + k; while (slot >= d.KS) slot -= d.KS;
        d.wl_shop[2 * (size_t)(base + k)] = (uint32_t)env | ((uint32_t)slot << 24);
        d.wl_shop[2 * (size_t)(base + k) + 1] = d.sseed[(size_t)env * BG_SSEED + ((head + k) & (BG_SSEED - 1))];
      }
      head = (head + emitted) & (BG_SSEED - 1); cnt -= emitted;
    } else emitted = 0;
    if (emitted) {
      d.smeta[env] = (uint32_t)head | ((uint32_t)cnt << 8);
      prod = (prod & 0xffff00ffu) | ((((prod >> 8) + (uint32_t)emitted) & 0xffu) << 8);
    }
    if (cnt < BG_SSEED - 8) { uint32_t i = atomicAdd(&d.wl_count[1], 1u); d.wl[N + i] = (uint32_t)env; }
  }
  d.prod_out[env] = prod; // the deck / block kernels bump their byte of prod_out when their data is written
  if (d_ready < d.KD) { uint32_t i = atomicAdd(&d.wl_count[0], 1u); d.wl[i] = (uint32_t)env; }
  if (g_valid > 0 && g_valid < d.KG) { uint32_t i = atomicAdd(&d.wl_count[2], 1u); d.wl[2 * N + i] = (uint32_t)env; }
}

// Lane = an env that is short of pre-shuffled decks; rounds of one shuffle each.  An env needs a deck per blind it started (~18 per
// 372-step launch under a uniform policy: 1.2 M shuffles per refill at 65 536 envs) and its shuffles are serial (one stream), so a wave
// that kept its 64 envs until the last of them was full ran max-over-lanes rounds: a lane whose env is full takes the NEXT env of the
// work list instead (one atomic per wave and round), and the kernel runs ~ (decks / 64) wave-rounds of ~17 us (instruction-bound: ~70
// draws x ~45 instructions).  The deck under construction is the kernel's only LDS (52 bytes per lane: random.shuffle indexes
// it with a per-lane j): 3.3 KB per wave, 128 registers: two waves fit beside a step-engine workgroup (it leaves 256 registers of one SIMD
// and 10 KB of LDS per CU).
__global__ __launch_bounds__(BG_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void bg_refill_deck_kernel(BgDev d, BgPart pt, uint32_t cursor_idx, uint32_t max_made) {
  __builtin_amdgcn_s_setprio(BG_REFILL_WAVE_PRIO);
  __shared__ uint8_t sdeck[52][BG_BLOCK];
  const size_t N = d.N;
  const int tid = threadIdx.x;
  const uint32_t count0 = d.wl_count[0], lo = bg_part_lo(count0, pt), count = bg_part_hi(count0, pt);   // this launch's part of the list: [lo, count)
  uint32_t* const cursor = &d.wl_count[cursor_idx];   // [4], or [8 + k] for piece k of a refill in pieces
  const int cap = max_made ? (int)max_made : 1 << 30;   // decks per env and launch (a piece: an env's shuffles are serial -- ~17 per 372 steps -- so a piece is cut in DEPTH too)
  int env = 0, d_head = 0, d_ready = 0, made = 0;
  uint32_t prod = 0, cur = 0;
  uint32_t* mt = bg_deckmt(d, 0);
  bool have = false, need = false, dry = false; // dry (wave-uniform): the list has no more envs
#pragma unroll 1
  while (true) {
    if (!dry) {
      const unsigned long long freem = __ballot(!have);
      if (freem != 0ull) {
        const int leader = __ffsll(freem) - 1, nfree = __popcll(freem);
        uint32_t base = 0;
        if (tid == leader) base = atomicAdd(cursor, (uint32_t)nfree);
        base = lo + (uint32_t)__shfl((int)base, leader);
        dry = base + (uint32_t)nfree >= count;
        const uint32_t my = base + (uint32_t)__popcll(freem & ((1ull << tid) - 1ull));
        if (!have && my < count) {
          env = (int)d.wl[my];
          mt = bg_deckmt(d, env);
          const uint32_t w5 = ((const uint32_t*)&d.hot[(size_t)5 * N + env])[3];
          prod = d.prod_out[env];
          d_head = bg_b(w5, 2);
          d_ready = (int)((prod - (uint32_t)bg_b(w5, 3)) & 0xffu);
          cur = mt[BG_MT_N] & 0x3ffu;
          made = 0;
          need = d_ready < d.KD;
          have = need;
        }
      }
    }
    if (__ballot(need) == 0ull) { if (dry) break; continue; }
    if (need) { int p = 0; for (int s = 0; s < 4; s++) for (int r = 0; r < 13; r++) sdeck[p++][tid] = (uint8_t)(r * 4 + s); } // :519-522
    int i = 51; // random.shuffle: for i = 51..1: j = _randbelow(i + 1), swap
    cur = bg_lazy_stream(mt, cur, need, [&](uint32_t y) {
      const int k = 32 - __clz((uint32_t)(i + 1));
      const uint32_t j = bg_temper(y) >> (32 - k);
      if (j <= (uint32_t)i) {
        const uint8_t a = sdeck[i][tid], b = sdeck[j][tid];
        sdeck[i][tid] = b; sdeck[j][tid] = a;
        i--;
      }
      return i >= 1;
    });
    if (need) {
      const int slot = (d_head + d_ready) % d.KD;
#pragma unroll
      for (int k = 0; k < BG_NDECK; k++) {
        uint32_t wv[4] = {0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < 16; b++) { int i2 = k * 16 + b; if (i2 < 52) wv[b >> 2] |= (uint32_t)sdeck[i2][tid] << (8 * (b & 3)); }
        d.ndeck[((size_t)slot * BG_NDECK + k) * N + env] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
      }
      d_ready++; made++;
      need = d_ready < d.KD && made < cap;
      if (!need) { // this env's ring is full (or has its share of this piece): cursor and producer byte back, the lane is free for the next env
        mt[BG_MT_N] = cur | BG_LAZY_SEEDED;
        ((uint8_t*)&d.prod_out[env])[0] = (uint8_t)((prod + (uint32_t)made) & 0xffu); // byte store: the block kernel may run concurrently
        have = false;
      }
    }
  }
}

// top up the per-env ring of pre-drawn shop seeds: `rng.get_int('shop_generation', 0, 2**31 - 1)` (:1389) in stream order
// (_randbelow(2**31): k = 32 bits, accept r < 2**31)
__global__ __launch_bounds__(BG_BLOCK) void bg_refill_seedring_kernel(BgDev d, BgPart pt) {
  __builtin_amdgcn_s_setprio(BG_REFILL_WAVE_PRIO);
  size_t N = d.N;
  const uint32_t count0 = d.wl_count[1], lo = bg_part_lo(count0, pt), count = bg_part_hi(count0, pt);
  for (uint32_t base = lo + blockIdx.x * BG_BLOCK; base < count; base += gridDim.x * BG_BLOCK) {
    const bool valid = base + threadIdx.x < count;
    const int env = valid ? (int)d.wl[N + base + threadIdx.x] : 0;
    uint32_t* mt = bg_shopgenmt(d, env);
    int head = 0, cnt = BG_SSEED;
    uint32_t cur = 0;
    if (valid) {
      const uint32_t sm = d.smeta[env];
      head = (int)(sm & 0xffu); cnt = (int)((sm >> 8) & 0xffu);
      cur = mt[BG_MT_N] & 0x3ffu;
    }
    const bool need = valid && cnt < BG_SSEED;
    cur = bg_lazy_stream<BG_LAZY_U_SEEDRING>(mt, cur, need, [&](uint32_t y) {
      const uint32_t r = bg_temper(y);
      if (r < 2147483648u) { d.sseed[(size_t)env * BG_SSEED + ((head + cnt) & (BG_SSEED - 1))] = r; cnt++; }
      return cnt < BG_SSEED;
    });
    if (need) {
      mt[BG_MT_N] = cur | BG_LAZY_SEEDED;
      d.smeta[env] = (uint32_t)head | ((uint32_t)cnt << 8);
    }
  }
}

// `random.Random(shop_seed)` (shop.py:96): one stream per lane, pure ALU + 156 stores.  The slot holds the SEEDED state;
// the consumer regenerates the few words a shop visit reads (bg_sprefetch), so no block twist is ever run for a shop.
// (eight waves per SIMD = at most 64 registers: FOUR of these waves then fit on the SIMD the step engine leaves free -- with 66 it was three)
__global__ __launch_bounds__(BG_BLOCK, 8) void bg_refill_shop_kernel(BgDev d, BgPart pt) {
  const uint32_t count0 = d.wl_count[3], lo = bg_part_lo(count0, pt), count = bg_part_hi(count0, pt);
  for (uint32_t item = lo + blockIdx.x * BG_BLOCK + threadIdx.x; item < count; item += gridDim.x * BG_BLOCK) {
    uint32_t es = d.wl_shop[2 * (size_t)item], seed = d.wl_shop[2 * (size_t)item + 1];
    bg_mt_seed_slot(bg_sblock(d, (int)(es & 0xffffffu), (int)(es >> 24)), seed);
  }
}
// Next 624-word block(s) of the per-env global stream, ONE WAVE PER ENV: lane l holds words l, l + 64, ... of the block (ten
// registers), so the block is read and written as ten 256-byte rows (a lane per env read 16 bytes of 64 different blocks per
// instruction), word k+1 and word k+397 / k-227 come from other lanes (`__shfl`), and a block that is twisted again stays in
// registers.  Row r only needs NEW rows r-4 / r-3, so the rows go in order; word 623 pairs with the new word 0.
__global__ __launch_bounds__(BG_BLOCK) void bg_refill_gblk_kernel(BgDev d, BgPart pt) {
  __builtin_amdgcn_s_setprio(BG_REFILL_WAVE_PRIO);
  const size_t N = d.N;
  const uint32_t count0 = d.wl_count[2], lo = bg_part_lo(count0, pt), count = bg_part_hi(count0, pt);
  const int l = threadIdx.x;
  constexpr int NR = (BG_MT_N + 63) / 64; // 10 rows, the last one 48 words
  for (uint32_t item = lo + blockIdx.x; item < count; item += gridDim.x) {
    const int env = (int)d.wl[2 * N + item];
    const uint32_t w6 = ((const uint32_t*)&d.hot[(size_t)6 * N + env])[3];
    const uint32_t prod = d.prod_out[env];
    const int g_cur = bg_b(w6, 2), g_cons = bg_b(w6, 3);
    int g_valid = (int)(((prod >> 16) - (uint32_t)g_cons) & 0xffu);
    if (g_valid <= 0 || g_valid >= d.KG) continue;
    int last = (g_cur + g_valid - 1) % d.KG, made = 0;
    uint32_t o[NR + 1];
    {
      const uint32_t* src = bg_gblock(d, env, last);
#pragma unroll
      for (int r = 0; r < NR; r++) o[r] = (64 * r + l < BG_MT_N) ? __builtin_nontemporal_load(&src[64 * r + l]) : 0u; // read once
      o[NR] = 0u;
    }
#pragma unroll 1
    while (g_valid < d.KG) {
      const int nxt = last + 1 == d.KG ? 0 : last + 1;
      uint32_t* dst = bg_gblock(d, env, nxt);
      uint32_t n[NR];
#pragma unroll
      for (int r = 0; r < NR; r++) {
        const int k = 64 * r + l;
        // word k+1: lane l+1 of this row, lane 0 of the next one for l = 63
        const uint32_t up_same = __shfl(o[r], (l + 1) & 63), up_next = __shfl(o[r + 1], 0);
        uint32_t up = l == 63 ? up_next : up_same;
        if (r == NR - 1) { const uint32_t new0 = __shfl(n[0], 0); if (k == BG_MT_N - 1) up = new0; } // kk == 623 pairs with the NEW word 0
        // word k+397 (old, k < 227): 64 (r+6) + l + 13;  word k-227 (new, k >= 227): 64 (r-4) + l + 29
        uint32_t far = 0u;
        if (64 * r < BG_MT_N - BG_MT_M) { // some lane of this row has k < 227
          const uint32_t a = r + 6 < NR ? __shfl(o[r + 6], (l + 13) & 63) : 0u, b2 = r + 7 < NR ? __shfl(o[r + 7], (l + 13) & 63) : 0u;
          far = l + 13 < 64 ? a : b2;
        }
        if (64 * r + 63 >= BG_MT_N - BG_MT_M) { // some lane has k >= 227
          const uint32_t a = r >= 4 ? __shfl(n[r >= 4 ? r - 4 : 0], (l + 29) & 63) : 0u, b2 = r >= 3 ? __shfl(n[r >= 3 ? r - 3 : 0], (l + 29) & 63) : 0u;
          const uint32_t fn = l + 29 < 64 ? a : b2;
          if (k >= BG_MT_N - BG_MT_M) far = fn;
        }
        n[r] = bg_twist(o[r], up, far);
        // non-temporal: whole 256-byte rows (full lines, nothing for the L2 to combine) that are read a launch later -- kept out of the
        // L2 the step engine beside this kernel is working from
        if (k < BG_MT_N) __builtin_nontemporal_store(n[r], &dst[k]);
      }
      // the 16 spare words behind a block mirror the head of its successor: a 16-word read never has to change blocks
      if (l < 16) bg_gblock(d, env, last)[BG_MT_N + l] = n[0];
#pragma unroll
      for (int r = 0; r < NR; r++) o[r] = n[r];
      last = nxt; g_valid++; made++;
    }
    if (made && l == 0) {
      const uint32_t g = ((prod >> 16) + (uint32_t)made) & 0xffu;
      // only this wave touches byte 2 of this env's word during a refill; bytes 0 / 1 belong to the scan / deck kernels
      ((uint8_t*)&d.prod_out[env])[2] = (uint8_t)g;
    }
  }
}

// apply the reset template to the live state (bg_inject apply_now)
__global__ __launch_bounds__(BG_BLOCK) void bg_inject_kernel(BgDev d, const uint8_t* __restrict__ mask_in) {
  int env = blockIdx.x * BG_BLOCK + threadIdx.x;
  if (env >= d.N) return;
  if (mask_in && !mask_in[env]) return;
  Env e;
  bg_load_env(d, env, e);
  uint4 t0 = d.tmpl[env], t1 = d.tmpl[(size_t)d.N + env];
  if (t0.y & 0x80000000u) { e.njokers = (int)bg_b(t0.y, 1); e.jokers = (uint64_t)t0.x | ((uint64_t)(t0.y & 0xffu) << 32); }
  if (t0.y & 0x40000000u) e.money = (int32_t)t0.z;
  if (t0.y & 0x20000000u) e.ante = (int)bg_b(t0.y, 2);
  if (t0.y & 0x10000000u) { e.levels = (uint64_t)t1.x | ((uint64_t)(t1.y & 0xffffu) << 32); e.excess = 0; }
  if (t1.z & 0x80000000u) { e.ncons = (int)bg_b(t1.z, 0); e.cons0 = bg_b(t1.z, 1); e.cons1 = bg_b(t1.z, 2); }
  bg_store_env(d, env, e);
}

__global__ __launch_bounds__(256) void bg_tables_build_kernel(uint32_t* out) {
  __shared__ JTables jt;
  { uint32_t* w = (uint32_t*)&jt; for (int i = threadIdx.x; i < (int)(sizeof(JTables) / 4); i += blockDim.x) w[i] = 0u; }
  __syncthreads();
  bg_tables_init(&jt);
  const uint32_t* w = (const uint32_t*)&jt;
  for (int i = threadIdx.x; i < (int)(sizeof(JTables) / 4); i += blockDim.x) out[i] = w[i];
}

// harness injection of the live deck order (bg_inject_deck) and of the per-env curriculum cap (bg_set_max_ante)
__global__ __launch_bounds__(BG_BLOCK) void bg_inject_deck_kernel(BgDev d, const uint4* __restrict__ decks, const uint8_t* __restrict__ mask_in) {
  int env = blockIdx.x * BG_BLOCK + threadIdx.x;
  if (env >= d.N) return;
  if (mask_in && !mask_in[env]) return;
#pragma unroll
  for (int k = 0; k < BG_NDECK; k++) d.deck[(size_t)k * d.N + env] = decks[(size_t)env * BG_NDECK + k];
}
__global__ __launch_bounds__(BG_BLOCK) void bg_set_cap_kernel(BgDev d, const int32_t* __restrict__ caps, int scalar, const uint8_t* __restrict__ mask_in) {
  int env = blockIdx.x * BG_BLOCK + threadIdx.x;
  if (env >= d.N) return;
  if (mask_in && !mask_in[env]) return;
  uint32_t* w = ((uint32_t*)&d.hot[(size_t)7 * d.N + env]) + 3; // chunk 7, word 3: excess (low half) | cap << 16
  const int cap = caps ? caps[env] : scalar;
  *w = (*w & 0xffffu) | ((uint32_t)(cap & 0xff) << 16);
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
struct bg_handle {
  BgDev dev;
  int device_id;
  bool seeded;
  int64_t* d_seeds;
  uint8_t* d_mask;
  uint32_t* d_jtab;
  int steps_since_refill; // env steps launched (bg_step, bg_step_many, rollouts) since the last refill was LAUNCHED
  std::vector<uint4> h_tmpl;
  std::vector<hipEvent_t> ev_pool;   // events of the per-launch profile, re-used (bg_set_profiling fills it)
  std::vector<uint8_t> h_cap; // host mirror of every env's curriculum cap (0 = none): what template antes are validated against
  uint64_t bytes;
  std::string err;
  // optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg)
  bool profiling;
  // refill pipeline: double-buffered producer counters, a side stream and per-parity completion events
  uint32_t* d_prod[2];
  long refill_seq;       // refills launched so far; refill #i writes d_prod[i & 1]
  long refill_done;      // highest refill index KNOWN to have completed (hipEventQuery): launches that read its view need no stream wait any more
  long view_min;         // index of the last SYNCHRONOUS refill: no launch may read producer counters older than its
  bool async_refill;     // BG_ASYNC_REFILL (default on): bg_rollout overlaps refill #i with rollout chunk i+1
  hipStream_t side, side2, side3; // side: overlapped refills; side2/3: the deck and block kernels of one refill run beside the shop kernel
  hipEvent_t ev_scan, ev_deck, ev_gblk;
  hipEvent_t ev_refill[2];
  hipEvent_t ev_rollout;
  std::vector<hipEvent_t> ev_rollout_t, ev_refill_t, ev_step_t; // start/stop pairs
  std::vector<int> rollout_steps;                         // fused steps of each timed rollout launch
  // tunables read ONCE per handle in bg_create (environment variables)
  int refill_blocks, refill_blocks_shop, dev_skip_refill, refill_order, refill_min;
  // a refill in PIECES (bg_refill_pieces): the kernels of refill #(refill_seq - 1) that are still to be issued, one (or a few) beside every short launch
  struct RefillPiece { int kind; uint32_t part, nparts; int grid; uint32_t cursor, max_made; };   // kind: 0 deck, 1 seed ring, 2 global blocks, 3 shop streams
  std::vector<RefillPiece> pieces;
  size_t piece_next;
  BgDev piece_dev;
  int refill_sliced, piece_parts[4], piece_grid[4];
  int deck_passes, deck_rounds, refill_interleave, refill_sliced_div;
  uint32_t eng_run, eng_play, eng_other, eng_part, eng_more, eng_smask; int eng_waves, eng_copiers; // queue thresholds of the step engine (BG_ENG_RUN / _PLAY / _OTHER)
  int engine;            // BG_ENGINE: 3 = bg_engine3.h (owner + service waves in one workgroup) for packed-record rollouts (default), 1 = bg_engine.h everywhere
  // bg_engine3.h: workgroup shape (BG_E3_CFG = 100 * owner waves + 10 * slices + service waves; 0 = by env count) and the service waves' batch
  // thresholds (BG_E3_TH requests, or after BG_E3_WAIT ticks of 10 ns)
  int e3_cfg, e3_epw; uint32_t e3_th, e3_wait;
  // sharded jobs (bg_set_gather_peers): every rank's gather buffer as mapped into THIS process, the size of the job and this handle's rank
  uint8_t* gpeer[8]; uint8_t** d_gpeer; int gworld, grank;   // d_gpeer: the same eight pointers in device memory (what the kernel reads)
};

static std::string g_create_err;

// Every entry point works on the handle's device, whatever the caller's current device is, and leaves the caller's current
// device as it found it (a process may hold handles on several GPUs; torch's current device must not change under it).
struct BgDeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit BgDeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~BgDeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
#define BG_GUARD(h) BgDeviceGuard _guard((h)->device_id)

#define BG_HIP(call)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (call);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      h->err = std::string(#call) + ": " + hipGetErrorString(_e);                                      \
      return BG_E_HIP;                                                                                 \
    }                                                                                                  \
  } while (0)

template <typename T>
static hipError_t bg_alloc(bg_handle* h, T** p, size_t count) {
  size_t bytes = count * sizeof(T);
  hipError_t e = hipMalloc((void**)p, bytes);
  if (e != hipSuccess) return e;
  h->bytes += bytes;
  return hipMemset(*p, 0, bytes);
}

static int bg_grid(const bg_handle* h) { return (h->dev.N + BG_BLOCK - 1) / BG_BLOCK; }

// HIP events of the per-launch profile come from a pool filled when profiling is switched on: creating two events per launch inside a
// caller's timed region cost a 20-step launch a few microseconds of its 330
static hipEvent_t bg_ev_take(bg_handle* h) {
  if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
static void bg_ev_begin(bg_handle* h, std::vector<hipEvent_t>& v, hipStream_t s) {
  if (!h->profiling) return;
  hipEvent_t a = bg_ev_take(h), b = bg_ev_take(h);
  if (!a || !b) { if (a) h->ev_pool.push_back(a); if (b) h->ev_pool.push_back(b); return; }
  v.push_back(a); v.push_back(b);
  (void)hipEventRecord(a, s);
}
// a start / stop pair for a launch that records them itself (hipExtLaunchKernelGGL); both null when profiling is off
static void bg_ev_pair(bg_handle* h, std::vector<hipEvent_t>& v, hipEvent_t& a, hipEvent_t& b) {
  a = b = nullptr;
  if (!h->profiling) return;
  a = bg_ev_take(h); b = bg_ev_take(h);
  if (!a || !b) { if (a) h->ev_pool.push_back(a); if (b) h->ev_pool.push_back(b); a = b = nullptr; return; }
  v.push_back(a); v.push_back(b);
}
static void bg_ev_end(bg_handle* h, std::vector<hipEvent_t>& v, hipStream_t s) {
  if (!h->profiling || v.empty()) return;
  (void)hipEventRecord(v.back(), s);
}
static double bg_ev_sum(bg_handle* h, std::vector<hipEvent_t>& v) {
  double ms = 0;
  for (size_t i = 0; i + 1 < v.size(); i += 2) {
    float f = 0;
    if (hipEventElapsedTime(&f, v[i], v[i + 1]) == hipSuccess) ms += f;
    h->ev_pool.push_back(v[i]); h->ev_pool.push_back(v[i + 1]);
  }
  v.clear();
  return ms;
}

#ifndef BG_BUILD_SIGNATURE
#define BG_BUILD_SIGNATURE "unsigned"
#endif

extern "C" {

// sha256 prefix of the sources + flags + compiler this library was built from (balatro_gym_amd/build.py source_signature()); "unsigned" for ad-hoc builds
// (stored behind a marker so that build.py can read the signature OF THE LIBRARY out of the file without loading it: build.library_signature)
static const char bg_sig_marker[] = "bgsig:" BG_BUILD_SIGNATURE;
const char* bg_build_signature(void) { return bg_sig_marker + 6; }

// development hook: copy (and clear) the 16 phase counters written by -DBG_TIMING builds
int bg_debug_counters(bg_handle* h, unsigned long long* out16) {
  if (!h || !out16) return BG_E_ARG;
  BG_GUARD(h);
  BG_HIP(hipDeviceSynchronize());
  BG_HIP(hipMemcpy(out16, h->dev.dbg, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  BG_HIP(hipMemset(h->dev.dbg, 0, 32 * sizeof(unsigned long long)));
  return 0;
}

// Words of the per-env GLOBAL stream a run of `steps` consecutive steps can draw (the bound the block ring is sized by).
//   * an ACCEPTED play of c scoring cards (c <= 5) with j <= 5 jokers: one random() per (card, joker) pair = 2 c j words, a second one
//     per (8, 8 Ball) pair (four 8s in a deck: <= 2 * min(c, 4) * j), one randint(0, 23) per joker (>= 1 word, 1.33 expected), then the
//     next hand's on_hand_drawn (The Wheel: 8 x random() = 16 words; The Hook: two _randbelow): <= 20 c + 21 + rejections
//     (complete_joker_effects.py:35-184, boss_blinds.py:343-378);
//   * the play clears the selection and a play of c cards needs c card toggles first, which draw nothing (a REJECTED play --
//     boss_blinds.py:380-445, before the scorer -- draws nothing either): (20 c + 21) / (c + 1) <= 20.5 words per step;
//   * every other step draws less in one step than that (blind select: boss choice + on_hand_drawn <= 18; discard: on_hand_drawn <= 16
//     and needs a toggle too; a tarot / spectral <= 10).
// So the pathological policy (five 8 Balls, The Wheel on every blind, an 8 in every play) draws <= 20.5 T + 111 words in T steps plus
// ~1.7 expected rejection words per play; an ordinary random policy draws 2-4 per step.  The ring is priced at 32 T + 128 (24 T without
// the scorer-level chain: <= 18 per step).  Rounds 1-2 priced EVERY step as a five-card play (110 words): 137 blocks instead of 41.
// Underflow is loud (BG_DEVERR_GSTREAM -> BG_E_INTERNAL), never a wrong word.
static long bg_gwords(int flags, int steps) { return (flags & BG_FLAG_SCORER_JOKERS) ? 32l * steps + 128 : 24l * steps; }
static int bg_gsteps(int flags, int blocks) { // inverse: steps that `blocks` full blocks ahead of the cursor cover
  const long w = (long)blocks * BG_MT_N;
  const long t = (flags & BG_FLAG_SCORER_JOKERS) ? (w - 128) / 32 : w / 24;
  return t < 0 ? 0 : (int)t;
}

// development aid: choose which refill kernels run (bit set = skipped; tools/refill_alone.py)
int bg_debug_set_skip(bg_handle* h, int skip) { if (!h) return BG_E_ARG; h->dev_skip_refill = skip; return 0; }

// development aid: the four work-list lengths of the most recent refill (decks, seed rings, global blocks, shop streams)
int bg_debug_worklists(bg_handle* h, unsigned int* out4) {
  if (!h || !out4) return BG_E_ARG;
  BG_GUARD(h);
  BG_HIP(hipDeviceSynchronize());
  BG_HIP(hipMemcpy(out4, h->dev.wl_count, 4 * sizeof(unsigned int), hipMemcpyDeviceToHost));
  return 0;
}

int bg_set_profiling(bg_handle* h, int enable) {
  if (!h) return BG_E_ARG;
  h->profiling = enable != 0;
  if (h->profiling) { BG_GUARD(h); while (h->ev_pool.size() < 128) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) break; h->ev_pool.push_back(e); } }
  return 0;
}

// out[0..5] = rollout kernel ms, launches, fused env-steps per env summed over launches; refill ms, launches; step ms
int bg_get_profile(bg_handle* h, double* out8) {
  if (!h || !out8) return BG_E_ARG;
  BG_GUARD(h);
  BG_HIP(hipDeviceSynchronize());
  double nr = (double)(h->ev_rollout_t.size() / 2), nf = (double)(h->ev_refill_t.size() / 2), ns = (double)(h->ev_step_t.size() / 2);
  double steps = 0;
  for (int t : h->rollout_steps) steps += t;
  h->rollout_steps.clear();
  out8[0] = bg_ev_sum(h, h->ev_rollout_t); out8[1] = nr; out8[2] = steps;
  out8[3] = bg_ev_sum(h, h->ev_refill_t); out8[4] = nf;
  out8[5] = bg_ev_sum(h, h->ev_step_t); out8[6] = ns; out8[7] = 0;
  return 0;
}

int bg_create(int n_envs, int device_id, uint32_t flags, int max_ante, bg_handle** out) { return bg_create_ex(n_envs, device_id, flags, max_ante, 0, out); }

int bg_create_ex(int n_envs, int device_id, uint32_t flags, int max_ante, int fused_steps_hint, bg_handle** out) {
  if (!out || n_envs <= 0) { g_create_err = "bg_create: bad arguments"; return BG_E_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_create_err = "bg_create: no HIP device visible (this library has no CPU fallback)";
    return BG_E_NODEVICE;
  }
  if (device_id < 0 || device_id >= ndev) { g_create_err = "bg_create: device_id out of range"; return BG_E_ARG; }
  bg_handle* h = new bg_handle();
  h->device_id = device_id; h->seeded = false; h->bytes = 0; h->profiling = false;
  { const char* av = getenv("BG_ASYNC_REFILL"); h->async_refill = av ? atoi(av) != 0 : true; }
  { // every tunable is read here, once per handle (a process may A/B two handles with different settings)
    auto geti = [](const char* k, int dflt) { const char* v = getenv(k); return v ? atoi(v) : dflt; };
    h->refill_blocks = geti("BG_REFILL_BLOCKS", 4096); h->refill_blocks_shop = geti("BG_REFILL_BLOCKS_SHOP", 0);
    h->dev_skip_refill = geti("BG_DEV_SKIP_REFILL", 0);
    h->refill_order = geti("BG_REFILL_ORDER", 2);
    // BG_REFILL_MIN = m > 0: a rollout launch of >= m steps (since the last refill) takes a refill beside it.  Default 0 = only when the rings demand one:
    // 20 steps' worth of refill is ~320 us of five small latency-bound kernels against 129 us per 20 steps in bulk (profiles/r05/refill_policy_ab.txt:
    // a refill beside every 20-step launch costs 21 % of the sustained rate)
    h->refill_min = geti("BG_REFILL_MIN", 0);
    // BG_REFILL_SLICED = 1: the refill a run of SHORT launches demands (every 18th launch at 20 steps) is issued in pieces -- its scan beside the launch that
    // demanded it, then one dense kernel over a PART of a work list beside each of the next launches (BG_REFILL_PARTS = deck,seedring,blocks,shop parts;
    // BG_REFILL_GRIDS = their grids: no more one-wave workgroups than fit beside a resident engine, so that none is left to be placed in the gap between
    // two launches, where it would take the registers the next engine workgroup needs)
    h->refill_sliced = geti("BG_REFILL_SLICED", 1);
    { const int dp[4] = {2, 1, 2, 8}, dg[4] = {512, 768, 1024, 1024};
      for (int k = 0; k < 4; k++) { h->piece_parts[k] = dp[k]; h->piece_grid[k] = dg[k]; }
      auto get4 = [](const char* k, int* out, int lo, int hi) { const char* v = getenv(k); if (!v) return; int a[4]; if (sscanf(v, "%d,%d,%d,%d", &a[0], &a[1], &a[2], &a[3]) == 4) for (int i = 0; i < 4; i++) out[i] = a[i] < lo ? lo : a[i] > hi ? hi : a[i]; };
      get4("BG_REFILL_PARTS", h->piece_parts, 1, 16); get4("BG_REFILL_GRIDS", h->piece_grid, 64, 65536); }
    h->piece_next = 0;
    h->refill_interleave = geti("BG_REFILL_INTERLEAVE", 1);
    h->refill_sliced_div = geti("BG_REFILL_SLICED_DIV", 2); if (h->refill_sliced_div < 1) h->refill_sliced_div = 1;   // launches of at most max_chunk / div steps get the refill in pieces
    h->deck_passes = geti("BG_REFILL_DECK_PASSES", 3); h->deck_rounds = geti("BG_REFILL_DECK_ROUNDS", 6);
    if (h->deck_passes < 1) h->deck_passes = 1; if (h->deck_rounds < 1) h->deck_rounds = 1;
    while (h->deck_passes * h->piece_parts[0] > BG_WL_COUNTERS - 8) { if (h->deck_passes > 1) h->deck_passes--; else h->piece_parts[0]--; }
    h->eng_run = (uint32_t)geti("BG_ENG_RUN", 64); h->eng_play = (uint32_t)geti("BG_ENG_PLAY", 64); h->eng_other = (uint32_t)geti("BG_ENG_OTHER", 64);
    h->eng_part = (uint32_t)geti("BG_ENG_PART", 1); h->eng_more = (uint32_t)geti("BG_ENG_MORE", 0); h->eng_smask = (uint32_t)geti("BG_ENG_SMASK", BG_ENG_SMASK_DEFAULT); h->eng_waves = geti("BG_ENG_WAVES", 0); h->eng_copiers = geti("BG_ENG_COPIERS", 0); if (h->eng_copiers < 0 || h->eng_copiers > 3) h->eng_copiers = 0;   // 0 = by launch length (bg_engine_launch)
    if (h->eng_smask == 0 || h->eng_smask >= (1u << BG_ENG_NW) || __builtin_popcount(h->eng_smask) > BG_ENG_NSV) h->eng_smask = BG_ENG_SMASK_DEFAULT;
    h->engine = geti("BG_ENGINE", 3);
    if (h->engine != 1 && h->engine != 3) { delete h; g_create_err = "bg_create: BG_ENGINE must be 3 (bg_engine3.h, the default) or 1 (bg_engine.h); the two-kernel engine 2 was retired in round 5"; return BG_E_ARG; }
    h->e3_cfg = geti("BG_E3_CFG", 0);
    if (h->e3_cfg != 0 && h->e3_cfg != 113 && h->e3_cfg != 213 && h->e3_cfg != 413 && h->e3_cfg != 414) { delete h; g_create_err = "bg_create: BG_E3_CFG must be 113, 213, 413 or 414 (100 x owner waves + 10 x slices + service waves)"; return BG_E_ARG; }
    h->e3_th = (uint32_t)geti("BG_E3_TH", 0x7fffffff); h->e3_wait = (uint32_t)geti("BG_E3_WAIT", 0);
    for (int g = 0; g < 8; g++) h->gpeer[g] = nullptr;
    h->d_gpeer = nullptr;
    h->gworld = 0; h->grank = 0;
    h->e3_epw = geti("BG_E3_EPW", 0);   // live envs per 64-env workgroup (0 = by env count)
    if (h->e3_epw != 0 && (h->e3_epw < 1 || h->e3_epw > 64)) { delete h; g_create_err = "bg_create: BG_E3_EPW must be in [1, 64]"; return BG_E_ARG; }
  }
  h->d_prod[0] = h->d_prod[1] = nullptr; h->refill_seq = 0; h->view_min = 0; h->refill_done = -1; h->side = h->side2 = h->side3 = nullptr; h->ev_scan = h->ev_deck = h->ev_gblk = nullptr;
  h->ev_refill[0] = h->ev_refill[1] = nullptr; h->ev_rollout = nullptr;
  h->d_seeds = nullptr; h->d_mask = nullptr; h->d_jtab = nullptr; h->steps_since_refill = 0;
  memset(&h->dev, 0, sizeof(h->dev));
  BgDev& d = h->dev;
  d.N = n_envs; d.flags = flags; d.max_ante = max_ante;
  const char* kg = getenv("BG_KG"); const char* ks = getenv("BG_KS"); const char* kd = getenv("BG_KD");
  // Look-ahead depth = how many steps one bg_rollout launch may fuse (bg_max_fused_steps).  Deeper rings amortise the
  // refill over more steps and the end-of-launch tail (lanes that finished early wait for the slowest env of their
  // workgroup) over more work: 128 -> 256 -> 372 fused steps measured +6 % / +9 % at 65 536 envs.  Per env they cost 2.5 KB
  // per global block, 256 B per shop slot and 64 B per deck: 0.2 MB at full depth (372 fused steps: 12.8 GB of the 288 at 65 536 envs;
  // 0.43 MB in round 3's first half, 0.51 in round 2, ~1 MB in round 1).  Ring positions are bytes, so 250 is the deepest ring.
  int dsd = n_envs <= 131072 ? 248 : (n_envs <= 262144 ? 124 : (n_envs <= 1048576 ? 24 : 12));
  // global-stream blocks: two launches' worth of what 3 * dsd / 2 fused steps can draw (bg_gwords) + the block under the cursor:
  // 41 / 21 / 7 / 5 with the scorer-level joker chain (137 / 73 / 13 / 8 while every step was priced as a five-card play), 31 / 17 / 5 / 5 without
  int dg = 2 * (int)((bg_gwords(flags, 3 * (dsd / 2)) + BG_MT_N - 1) / BG_MT_N) + 1; if (dg < 5) dg = 5;
  if (fused_steps_hint > 0) {
    // the caller says how many steps it will ever fuse into one launch (bg_step users: a handful): rings only as deep as that
    // needs -- two launches' worth (the refill is overlapped), a deck / shop stream per 3 steps, 110 global words per step
    const int hint = fused_steps_hint > 372 ? 372 : fused_steps_hint;
    int ring = 2 * ((hint + 2) / 3); if (ring < 4) ring = 4;
    int gb = 2 * (int)((bg_gwords(flags, hint) + BG_MT_N - 1) / BG_MT_N) + 1; if (gb < 5) gb = 5;
    if (ring < dsd) dsd = ring;
    if (gb < dg) dg = gb;
  }
  d.KG = kg ? atoi(kg) : dg; d.KS = ks ? atoi(ks) : dsd + 1; d.KD = kd ? atoi(kd) : dsd;
  if (d.KG < 2 || d.KS < 2 || d.KD < 1 || d.KG > 250 || d.KS > 250 || d.KD > 250) { delete h; g_create_err = "bg_create: bad ring depths"; return BG_E_ARG; }
  size_t N = (size_t)n_envs;
  BG_GUARD(h); // the caller's current device is restored on return
  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = bg_alloc(h, &d.hot, BG_NHOT * N);
  if (e == hipSuccess) e = bg_alloc(h, &d.deck, BG_NDECK * N);
  if (e == hipSuccess) e = bg_alloc(h, &d.cold, BG_NCOLD * N);
  if (e == hipSuccess) e = bg_alloc(h, &d.tmpl, BG_NTMPL * N);
  if (e == hipSuccess) e = bg_alloc(h, &d.ndeck, (size_t)d.KD * BG_NDECK * N);
  if (e == hipSuccess) e = bg_alloc(h, &d.gblk, (size_t)d.KG * BG_MTS * N);
  if (e == hipSuccess) e = bg_alloc(h, &d.sblk, (size_t)d.KS * BG_SLOT_WORDS * N);
  if (e == hipSuccess) e = bg_alloc(h, &d.sovf, (size_t)BG_MTS * N);
  if (e == hipSuccess && (flags & BG_FLAG_CARD_STATES)) {
    e = bg_alloc(h, &d.cstate, (size_t)BG_NCST * N);
    if (e == hipSuccess) e = bg_alloc(h, &d.ctmpl, (size_t)BG_NCST * N);
    if (e == hipSuccess) e = bg_alloc(h, &d.cardmt, (size_t)BG_MTS * N);
    if (e == hipSuccess) e = bg_alloc(h, &d.sealmt, (size_t)BG_MTS * N);
  }
  if (e == hipSuccess) e = bg_alloc(h, &d.deckmt, (size_t)BG_MTS * N);
  if (e == hipSuccess) e = bg_alloc(h, &d.shopgenmt, (size_t)BG_MTS * N);
  if (e == hipSuccess) e = bg_alloc(h, &d.wl_count, BG_WL_COUNTERS);
  if (e == hipSuccess) e = bg_alloc(h, &d.wl, 3 * N);
  if (e == hipSuccess) e = bg_alloc(h, &d.wl_shop, 2 * N * (size_t)(d.KS - 1));
  if (e == hipSuccess) e = bg_alloc(h, &d.sseed, (size_t)BG_SSEED * N);
  if (e == hipSuccess) e = bg_alloc(h, &d.smeta, N);
  if (e == hipSuccess) e = bg_alloc(h, &h->d_prod[0], N);
  if (e == hipSuccess) e = bg_alloc(h, &h->d_prod[1], N);
  // The refill runs BESIDE the next step-engine launch, in the registers of the one SIMD per CU and the ~5 KB of LDS the engine
  // leaves free.  The shop seeding (1.1 M ALU-bound waves' worth per launch) goes on a stream of the LOWEST priority: its waves
  // would otherwise take every free register before the few, latency-bound deck / seed-ring / block waves are placed
  int prio_least = 0, prio_greatest = 0;
  if (e == hipSuccess) e = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  { const char* pv = getenv("BG_REFILL_PRIO"); if (pv && atoi(pv) == 0) prio_least = 0; } // development: 0 = default priority
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, prio_least);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->side2, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->side3, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_scan, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_deck, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_gblk, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_refill[0], hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_refill[1], hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_rollout, hipEventDisableTiming);
  if (e == hipSuccess) e = bg_alloc(h, &h->d_jtab, (sizeof(JTables) + 3) / 4);
  if (e == hipSuccess) e = bg_alloc(h, &d.dbg, 32);
  if (e == hipSuccess) e = bg_alloc(h, &d.err, 4);
  if (e == hipSuccess) e = bg_alloc(h, &h->d_seeds, N);
  if (e == hipSuccess) e = bg_alloc(h, &h->d_mask, N);
  if (e != hipSuccess) {
    g_create_err = std::string("bg_create: ") + hipGetErrorString(e);
    bg_destroy(h);
    return BG_E_HIP;
  }
  d.prod_view = h->d_prod[0]; d.prod_in = h->d_prod[0]; d.prod_out = h->d_prod[1];
  d.jtab = h->d_jtab;
  hipLaunchKernelGGL(bg_tables_build_kernel, dim3(1), dim3(256), 0, 0, h->d_jtab);
  if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { g_create_err = "bg_create: table kernel failed"; bg_destroy(h); return BG_E_HIP; }
  h->h_tmpl.assign(BG_NTMPL * N, make_uint4(0, 0, 0, 0));
  h->h_cap.assign(N, (uint8_t)(max_ante > 0 && max_ante < 256 ? max_ante : 0));
  if (max_ante < 0 || max_ante > 255) { g_create_err = "bg_create: max_ante must be in [0, 255]"; bg_destroy(h); return BG_E_ARG; }
  if (max_ante > 0) { // the cap lives in each env's state (bg_set_max_ante changes it later: a rising curriculum)
    hipLaunchKernelGGL(bg_set_cap_kernel, dim3(bg_grid(h)), dim3(BG_BLOCK), 0, 0, d, (const int32_t*)nullptr, max_ante, (const uint8_t*)nullptr);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { g_create_err = "bg_create: cap kernel failed"; bg_destroy(h); return BG_E_HIP; }
  }
  *out = h;
  return 0;
}

int bg_destroy(bg_handle* h) {
  if (!h) return 0;
  { BG_GUARD(h);
  BgDev& d = h->dev;
  (void)hipDeviceSynchronize();
  if (h->side) (void)hipStreamDestroy(h->side);
  if (h->side2) (void)hipStreamDestroy(h->side2);
  if (h->side3) (void)hipStreamDestroy(h->side3);
  if (h->ev_scan) (void)hipEventDestroy(h->ev_scan);
  if (h->ev_deck) (void)hipEventDestroy(h->ev_deck);
  if (h->ev_gblk) (void)hipEventDestroy(h->ev_gblk);
  for (int i = 0; i < 2; i++) if (h->ev_refill[i]) (void)hipEventDestroy(h->ev_refill[i]);
  if (h->ev_rollout) (void)hipEventDestroy(h->ev_rollout);
  for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
  for (auto* v : {&h->ev_rollout_t, &h->ev_refill_t, &h->ev_step_t}) for (hipEvent_t e : *v) (void)hipEventDestroy(e);
  hipFree(h->d_prod[0]); hipFree(h->d_prod[1]);
  hipFree(d.hot); hipFree(d.deck); hipFree(d.cold); hipFree(d.tmpl); hipFree(d.ndeck); hipFree(d.gblk); hipFree(d.sblk); hipFree(d.sovf);
  hipFree(d.deckmt); hipFree(d.shopgenmt); hipFree(d.err); hipFree(h->d_seeds); hipFree(h->d_mask); hipFree(d.wl_count); hipFree(d.wl); hipFree(d.wl_shop); hipFree(d.sseed); hipFree(d.smeta); hipFree(d.dbg); hipFree(h->d_jtab); hipFree(h->d_gpeer); hipFree(d.cstate); hipFree(d.ctmpl); hipFree(d.cardmt); hipFree(d.sealmt);
  }
  delete h;
  return 0;
}

const char* bg_last_error(const bg_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }
int bg_num_envs(const bg_handle* h) { return h ? h->dev.N : 0; }

// The look-ahead rings bound how many steps may be fused between two refills.  An env consumes at most one
// pre-shuffled deck and one pre-seeded shop stream per 3 steps (reset -> 45/47 -> select -> failing play;
// shop generated -> 31 -> select -> winning play), so R ready entries cover 3*R steps.  The global stream has
// (KG-1) full blocks of 624 words ahead; a step draws < 24 words without and < 110 with the scorer-level joker
// chain (8 cards x 5 jokers x 2 + 5 x randint, accepted plays are >= 2 steps apart).
// With the refill overlapped on the side stream a chunk only sees what the refill BEFORE the previous chunk produced,
// so the rings must hold two chunks' worth.
static int bg_chunk_limit(const bg_handle* h, bool* async_out) {
  int ring = h->dev.KD < h->dev.KS - 1 ? h->dev.KD : h->dev.KS - 1;
  int gblocks = h->dev.KG - 1;
  const bool async = h->async_refill && ring >= 2 && gblocks >= 2;
  if (async) { ring /= 2; gblocks /= 2; }
  int max_chunk = 3 * ring;
  int gchunk = bg_gsteps(h->dev.flags, gblocks);
  if (gchunk < max_chunk) max_chunk = gchunk;
  if (max_chunk < 1) max_chunk = 1;
  if (async_out) *async_out = async;
  return max_chunk;
}
int bg_max_fused_steps(const bg_handle* h) { return h ? bg_chunk_limit(h, nullptr) : 0; }
uint64_t bg_state_bytes(const bg_handle* h) { return h ? h->bytes : 0; }

// ---- refill pipeline ----------------------------------------------------------------------------------------
// refill #i reads the consumer counters (hot chunks) + d_prod[(i-1)&1] and writes rings + d_prod[i&1]
static uint32_t* bg_prod_latest(bg_handle* h) { return h->d_prod[(h->refill_seq + 1) & 1]; } // output of refill seq-1

static BgDev bg_dev_view(bg_handle* h, const uint32_t* view) {
  BgDev d = h->dev;
  d.prod_view = view;
  return d;
}

// make `s` wait until refill #(seq-1-back) is complete (events are recorded per parity)
static int bg_refill_pieces(bg_handle* h, int n);
static int bg_wait_refill(bg_handle* h, hipStream_t s, int back) {
  long i = h->refill_seq - 1 - back;
  if (i < 0) return 0;
  if (back == 0 && h->piece_next < h->pieces.size()) { int rc = bg_refill_pieces(h, 1 << 30); if (rc) return rc; }   // the latest refill is not even issued in full yet
  BG_HIP(hipStreamWaitEvent(s, h->ev_refill[i & 1], 0));
  return 0;
}

// Issue up to n of the pending pieces of the latest refill on the side stream (each bracketed by its own profiling events); behind the last one the
// refill's completion event.  The pieces only depend on the scan in front of them on that stream and on consumer counters, which may be newer than the
// scan's (more free slots, never fewer); the engine launches of the meantime read the view of the refill BEFORE this one.
static int bg_refill_pieces(bg_handle* h, int n) {
  const BgDev& d = h->piece_dev;
  hipStream_t s = h->side;
  while (n-- > 0 && h->piece_next < h->pieces.size()) {
    const bg_handle::RefillPiece& pc = h->pieces[h->piece_next++];
    const BgPart pt{pc.part, pc.nparts};
    hipEvent_t ea = nullptr, eb = nullptr;   // (profiling: the piece's own dispatch packet carries its timestamps -- no event packets around it)
    bg_ev_pair(h, h->ev_refill_t, ea, eb);
    const dim3 g(pc.grid), b(BG_BLOCK);
    if (ea) {
      if (pc.kind == 0) hipExtLaunchKernelGGL(bg_refill_deck_kernel, g, b, 0, s, ea, eb, 0, d, pt, pc.cursor, pc.max_made);
      else if (pc.kind == 1) hipExtLaunchKernelGGL(bg_refill_seedring_kernel, g, b, 0, s, ea, eb, 0, d, pt);
      else if (pc.kind == 2) hipExtLaunchKernelGGL(bg_refill_gblk_kernel, g, b, 0, s, ea, eb, 0, d, pt);
      else hipExtLaunchKernelGGL(bg_refill_shop_kernel, g, b, 0, s, ea, eb, 0, d, pt);
    } else {
      if (pc.kind == 0) hipLaunchKernelGGL(bg_refill_deck_kernel, g, b, 0, s, d, pt, pc.cursor, pc.max_made);
      else if (pc.kind == 1) hipLaunchKernelGGL(bg_refill_seedring_kernel, g, b, 0, s, d, pt);
      else if (pc.kind == 2) hipLaunchKernelGGL(bg_refill_gblk_kernel, g, b, 0, s, d, pt);
      else hipLaunchKernelGGL(bg_refill_shop_kernel, g, b, 0, s, d, pt);
    }
    BG_HIP(hipGetLastError());
    if (h->piece_next == h->pieces.size()) BG_HIP(hipEventRecord(h->ev_refill[(h->refill_seq - 1) & 1], s));   // refill #(refill_seq - 1) is complete behind this
  }
  return 0;
}

// steps_hint: env steps launched since the previous refill plus those of the chunk this one runs beside (< 0: unknown, size the grids for a full-depth refill)
// sliced: only zero + scan now, the dense kernels as pieces beside the next launches (bg_refill_pieces; anything that waits for this refill issues the rest)
static int bg_refill_on(bg_handle* h, hipStream_t s, int steps_hint = -1, bool sliced = false) {
  BgDev d = h->dev;
  d.prod_in = bg_prod_latest(h);
  d.prod_out = h->d_prod[h->refill_seq & 1];
  d.prod_view = nullptr;
  bg_ev_begin(h, h->ev_refill_t, s);
  // a one-wave kernel, not hipMemsetAsync: the runtime's fill kernel has multi-wave workgroups, which the dispatcher cannot place
  // beside a resident step-engine workgroup (only one SIMD per CU has free registers) -- the whole refill would wait for the engine
  hipLaunchKernelGGL(bg_refill_zero_kernel, dim3(1), dim3(BG_BLOCK), 0, s, d);
  hipLaunchKernelGGL(bg_refill_scan_kernel, dim3(bg_grid(h)), dim3(BG_BLOCK), 0, s, d);
  if (sliced && s == h->side && h->dev_skip_refill == 0) {
    bg_ev_end(h, h->ev_refill_t, s);
    BG_HIP(hipGetLastError());
    h->pieces.clear(); h->piece_next = 0; h->piece_dev = d;
    // the order of the serial refill: decks, seed rings, blocks, shop streams (the shop items of THIS refill were listed by the scan from seeds drawn earlier)
    // decks: an env's shuffles are one serial chain (~17 per 372 steps, up to the ring), so the deck work is cut in DEPTH as well -- `deck_passes` passes
    // over the list, all but the last one giving every env at most `deck_rounds` decks
    std::vector<bg_handle::RefillPiece> q[4];
    for (int pass = 0; pass < h->deck_passes; pass++)
      for (int p = 0; p < h->piece_parts[0]; p++)
        q[0].push_back({0, (uint32_t)p, (uint32_t)h->piece_parts[0], h->piece_grid[0], (uint32_t)(8 + pass * h->piece_parts[0] + p), pass + 1 < h->deck_passes ? (uint32_t)h->deck_rounds : 0u});
    for (int kind = 1; kind < 4; kind++)
      for (int p = 0; p < h->piece_parts[kind]; p++) q[kind].push_back({kind, (uint32_t)p, (uint32_t)h->piece_parts[kind], h->piece_grid[kind], 0u, 0u});
    // The four kinds are independent once the lists exist (the shop items were listed from seeds drawn by EARLIER refills): they are interleaved -- always
    // the kind with the largest share of its pieces still to go, the shop streams first -- so that the memory-bound pieces (an env's MT state is 2.5 KB
    // of its own: the deck and seed-ring kernels read 64 different lines per instruction, and a launch beside one takes ~40 us longer) do not sit beside
    // consecutive launches, with the ALU-bound shop pieces (~7 us) between them.  BG_REFILL_INTERLEAVE=0: kind after kind, decks first.
    static const int order[4] = {3, 0, 2, 1};
    size_t taken[4] = {0, 0, 0, 0};
    for (;;) {
      int best = -1; double share = 0.0;
      for (int o = 0; o < 4; o++) {
        const int k = h->refill_interleave ? order[o] : o;
        if (taken[k] >= q[k].size()) continue;
        const double sh = h->refill_interleave ? (double)(q[k].size() - taken[k]) / (double)q[k].size() : 1.0;
        if (best < 0 || sh > share + 1e-9) { best = k; share = sh; }
      }
      if (best < 0) break;
      h->pieces.push_back(q[best][taken[best]++]);
    }
    h->refill_seq++;
    h->steps_since_refill = 0;
    return 0;
  }
  const BgPart whole{0u, 1u};
  // Grids of ONE-WAVE workgroups, grid-stride over the compacted work lists: they are placed beside the resident step-engine
  // workgroups (bg_engine.h: one SIMD per CU and ~5 KB of LDS are left to them) or, with nothing else running, several per SIMD.
  // (a refill behind a short launch finds short work lists: a grid sized for 372 steps' worth would be thousands of one-wave workgroups that are
  //  placed beside the engine only to find nothing to do)
  int dense = h->refill_blocks;
  if (steps_hint >= 0 && steps_hint < 372) { dense = (int)((long)dense * (steps_hint + 24) / 372); if (dense < 512) dense = 512; if (dense > h->refill_blocks) dense = h->refill_blocks; }
  // the three kinds of work are independent once the lists exist: side by side on three streams, joined before the completion event
  if (h->refill_order != 2) {
    BG_HIP(hipEventRecord(h->ev_scan, s));
    BG_HIP(hipStreamWaitEvent(h->side2, h->ev_scan, 0));
    BG_HIP(hipStreamWaitEvent(h->side3, h->ev_scan, 0));
  }
  const int skip = h->dev_skip_refill; // development: contention experiments only (breaks the rings)
  const int dense_shop = h->refill_blocks_shop > 0 ? h->refill_blocks_shop : dense;
  // refill_order 0: the shop seeding (ALU-bound: 17 k waves of 64 registers, four fill the SIMD the engine leaves free) runs beside the
  // latency-bound deck / seed-ring / block waves, which then wait for registers behind it; 1: it runs AFTER them; 2: all five kernels
  // one after the other on `s` (a kernel's duration is then its own work, not its wait for a neighbour's registers)
  const bool shop_last = h->refill_order >= 1, serial = h->refill_order == 2;
  hipStream_t s_deck = serial ? s : h->side2, s_blk = serial ? s : h->side3;
  if (!shop_last && !(skip & 1)) hipLaunchKernelGGL(bg_refill_shop_kernel, dim3(dense_shop), dim3(BG_BLOCK), 0, s, d, whole); // lowest-priority stream
  if (!(skip & 2)) hipLaunchKernelGGL(bg_refill_deck_kernel, dim3(dense), dim3(BG_BLOCK), 0, s_deck, d, whole, 4u, 0u);
  if (!(skip & 4)) hipLaunchKernelGGL(bg_refill_seedring_kernel, dim3(dense), dim3(BG_BLOCK), 0, s_blk, d, whole);
  if (!(skip & 8)) hipLaunchKernelGGL(bg_refill_gblk_kernel, dim3(dense), dim3(BG_BLOCK), 0, s_blk, d, whole);
  if (!serial) {
    BG_HIP(hipEventRecord(h->ev_deck, h->side2));
    BG_HIP(hipEventRecord(h->ev_gblk, h->side3));
    BG_HIP(hipStreamWaitEvent(s, h->ev_deck, 0));
    BG_HIP(hipStreamWaitEvent(s, h->ev_gblk, 0));
  }
  if (shop_last && !(skip & 1)) hipLaunchKernelGGL(bg_refill_shop_kernel, dim3(dense_shop), dim3(BG_BLOCK), 0, s, d, whole);
  bg_ev_end(h, h->ev_refill_t, s);
  BG_HIP(hipGetLastError());
  BG_HIP(hipEventRecord(h->ev_refill[h->refill_seq & 1], s));
  h->refill_seq++;
  h->steps_since_refill = 0;
  return 0;
}

// synchronous flavour: ordered after everything on `stream`, and everything later on `stream` is ordered after it
int bg_refill(bg_handle* h, void* stream) {
  if (!h) return BG_E_ARG;
  BG_GUARD(h);
  hipStream_t s = (hipStream_t)stream;
  int rc = bg_wait_refill(h, s, 0); // the previous refill may still be running on the side stream
  if (rc) return rc;
  rc = bg_refill_on(h, s);
  h->view_min = h->refill_seq - 1;
  return rc;
}

int bg_check(bg_handle* h, void* stream) {
  if (!h) return BG_E_ARG;
  BG_GUARD(h);
  uint32_t w = 0;
  BG_HIP(hipMemcpyAsync(&w, h->dev.err, sizeof(w), hipMemcpyDeviceToHost, (hipStream_t)stream));
  BG_HIP(hipStreamSynchronize((hipStream_t)stream));
  if (w) {
    char buf[256];
    snprintf(buf, sizeof(buf), "device invariant violated: error word 0x%x (1 global-stream underflow, 2 shop block, 4 deck ring, 8 shop ring, 16 a bounded wait inside the step engine expired)", w);
    h->err = buf;
    return BG_E_INTERNAL;
  }
  return 0;
}

int bg_seed(bg_handle* h, const int64_t* seeds_host, const uint8_t* mask_host, int reseed_global, void* stream) {
  if (!h || !seeds_host) return BG_E_ARG;
  BG_GUARD(h);
  hipStream_t s = (hipStream_t)stream;
  size_t N = h->dev.N;
  BG_HIP(hipMemcpyAsync(h->d_seeds, seeds_host, N * sizeof(int64_t), hipMemcpyHostToDevice, s));
  if (mask_host) BG_HIP(hipMemcpyAsync(h->d_mask, mask_host, N, hipMemcpyHostToDevice, s));
  { int rcw = bg_wait_refill(h, s, 0); if (rcw) return rcw; }
  BgDev dseed = bg_dev_view(h, bg_prod_latest(h));
  dseed.prod_in = h->d_prod[0]; dseed.prod_out = h->d_prod[1]; // the seed kernel resets BOTH producer-counter buffers
  hipLaunchKernelGGL(bg_seed_kernel, dim3(bg_grid(h)), dim3(BG_BLOCK), 0, s, dseed, (const int64_t*)h->d_seeds,
                     mask_host ? (const uint8_t*)h->d_mask : (const uint8_t*)nullptr, reseed_global);
  BG_HIP(hipGetLastError());
  // the copies above read pageable host memory: make sure they are done before the caller reuses the buffers
  BG_HIP(hipStreamSynchronize(s));
  h->seeded = true;
  int rc = bg_refill(h, stream); // fills the deck ring and the shop-seed ring ...
  if (rc) return rc;
  return bg_refill(h, stream);   // ... then the shop streams from those seeds
}

static int bg_require_seeded(bg_handle* h) {
  if (!h) return BG_E_ARG;
  if (!h->seeded) { h->err = "bg_seed must be called before reset/step (streams are unseeded)"; return BG_E_ARG; }
  return 0;
}

static ObsPtrs bg_obs(const bg_obs_ptrs* o) {
  ObsPtrs p;
  memset(&p, 0, sizeof(p));
  if (o) memcpy(&p, o, sizeof(*o)); // the 31 per-key pointers; rows / row_stride stay 0
  return p;
}
static InfoPtrs bg_info(const bg_info_ptrs* o) {
  InfoPtrs p;
  if (o) memcpy(&p, o, sizeof(p)); else memset(&p, 0, sizeof(p));
  return p;
}

static void bg_obs_advance(ObsPtrs& o, size_t off) { // advance every non-null per-key pointer by `off` rows
  if (o.hand) o.hand += off * 8; if (o.hand_size) o.hand_size += off; if (o.deck_size) o.deck_size += off;
  if (o.selected_cards) o.selected_cards += off * 8; if (o.chips_scored) o.chips_scored += off;
  if (o.round_chips_scored) o.round_chips_scored += off; if (o.progress_ratio) o.progress_ratio += off;
  if (o.mult) o.mult += off; if (o.chips_needed) o.chips_needed += off; if (o.money) o.money += off;
  if (o.ante) o.ante += off; if (o.round) o.round += off; if (o.hands_left) o.hands_left += off;
  if (o.discards_left) o.discards_left += off; if (o.joker_count) o.joker_count += off;
  if (o.joker_ids) o.joker_ids += off * 10; if (o.joker_slots) o.joker_slots += off;
  if (o.consumable_count) o.consumable_count += off; if (o.consumables) o.consumables += off * 5;
  if (o.consumable_slots) o.consumable_slots += off; if (o.shop_items) o.shop_items += off * 10;
  if (o.shop_costs) o.shop_costs += off * 10; if (o.shop_rerolls) o.shop_rerolls += off;
  if (o.hand_levels) o.hand_levels += off * 12; if (o.phase) o.phase += off; if (o.action_mask) o.action_mask += off * 60;
  if (o.hands_played) o.hands_played += off; if (o.best_hand_this_ante) o.best_hand_this_ante += off;
  if (o.boss_blind_active) o.boss_blind_active += off; if (o.boss_blind_type) o.boss_blind_type += off;
  if (o.face_down_cards) o.face_down_cards += off * 8;
}
static void bg_info_advance(InfoPtrs& p, size_t off) {
  if (p.final_score) p.final_score += off; if (p.error) p.error += off; if (p.flags) p.flags += off; if (p.aux) p.aux += off;
  if (p.hand_type) p.hand_type += off; if (p.cards_played) p.cards_played += off; if (p.reward_terms) p.reward_terms += off * 8;
  if (p.score_breakdown) p.score_breakdown += off * 8;
}

int bg_reset(bg_handle* h, const uint8_t* mask_dev, const bg_obs_ptrs* obs, void* stream) {
  int rc = bg_require_seeded(h);
  if (rc) return rc;
  BG_GUARD(h);
  rc = bg_wait_refill(h, (hipStream_t)stream, 0);
  if (rc) return rc;
  if (h->dev.cstate) hipLaunchKernelGGL(bg_reset_kernel<true>, dim3(bg_grid(h)), dim3(BG_BLOCK), 0, (hipStream_t)stream, bg_dev_view(h, bg_prod_latest(h)), mask_dev, bg_obs(obs));
  else hipLaunchKernelGGL(bg_reset_kernel<false>, dim3(bg_grid(h)), dim3(BG_BLOCK), 0, (hipStream_t)stream, bg_dev_view(h, bg_prod_latest(h)), mask_dev, bg_obs(obs));
  BG_HIP(hipGetLastError());
  return bg_refill(h, stream);
}

// one launch of the step engine (bg_engine.h): T steps of every env of the handle
// worker waves for a launch of T fused steps (BG_ENG_WAVES overrides)
static int bg_engine_waves(const bg_handle* h, int T) {
  if (h->eng_waves >= 4 && h->eng_waves <= BG_ENG_NW) return h->eng_waves;
  // measured (tools/ab_waves.sh with BG_ENGINE=1, us per launch at 4 / 5 / 6 / 7 waves): T = 4: 80 / 79 / 80 / 78; 10: 150 / 149 / 145 / 146; 20: 260 / 254 /
  // 249 / 249; 40: 507 / 490 / 489 / 481; 80: 925 / 920 / 928 / 926; 160: 1 847 / 1 770 / 1 831 / 1 835; 372: 4 010 / 3 805 / 3 674 / 3 632.
  // (Rounds 2-3 ran short launches on FOUR waves: their measurements -- T = 20: 312 / 326 / 347 / 380 -- had six statistics atomics per WAVE in them,
  //  an end-of-launch tail that grew with the wave count; bg_step.h bg_stats_wave.)
  if (T <= 192) return BG_ENG_NW - 1;
  return BG_ENG_NW;
}
// ev_a / ev_b (both or neither): the launch's own start / stop events (hipExtLaunchKernelGGL: the timestamps ride on the kernel's dispatch packet; two
// hipEventRecord calls around the launch are two more barrier packets on the stream, ~2-3 us each in front of and behind a 220 us launch)
static void bg_engine_launch(bg_handle* h, const BgDev& dv, const EngineArgs& a0, bool hash, bool info, hipStream_t st, hipEvent_t ev_a = nullptr, hipEvent_t ev_b = nullptr) {
  const bool cards = h->dev.cstate != nullptr;
  EngineArgs a = a0;
  if (h->engine == 3 && a.obs.rows && !info && !a.actions_in && !a.reward && !a.term && !a.actions_out) { // packed-record rollouts: owner waves + service waves (bg_engine3.h)
    { // batch thresholds of the service waves: BG_E3_TH requests, or after BG_E3_WAIT ticks of 10 ns.  Default: never by threshold and no waiting --
      // a free service wave takes the FULLER of the two queues at once (measured at 372 steps: thresholds 32 / 48 / 64 with waits of 3 - 20 us all lose
      // 1 - 8 %, and serving plays as soon as one is queued -- batches of a few lanes -- loses a third: profiles/r04_engine3/thresholds_ab.txt)
      a.th_play = a.th_other = h->e3_th; a.th_more = h->e3_wait;   // (read once per handle in bg_create_ex, like every tunable)
    }
    // Shape of a workgroup: owner waves x slices of 64 envs x service waves.  BG_E3_CFG = 100 * owners + 10 * slices + service waves overrides.
    // 65 536 envs: 4 x 1 x 3 = 256 envs on seven waves per CU (the refill keeps its SIMD).  A small job spreads over more CUs: 64 envs per
    // workgroup up to 16 384 envs (256 workgroups = one per CU: 2.89 G env-steps/s against 2.59 G at 128 per workgroup), 128 up to 32 768
    // (profiles/r04_engine3/small_jobs.txt).  Unknown shapes are refused by bg_create_ex.
    const int cfg = h->e3_cfg ? h->e3_cfg : (h->dev.N <= 16384 ? 113 : (h->dev.N <= 32768 ? 213 : 413));
    // 64-env shape: live envs per workgroup (BG_E3_EPW, or by env count)
    int epw = 64;
    if (cfg == 113) {
      epw = h->e3_epw ? h->e3_epw : 64;
      // by itself: at least 256 workgroups (one per CU) of at least 16 live envs -- measured at 372 steps (profiles/r05/small_jobs.txt): 4 096 envs
      // 0.72 G env-steps/s at 64 per workgroup, 0.80 at 32, 0.87 at 16, 0.79 at 8; 8 192 envs 1.47 / 1.55 / 1.42 / 0.89; 16 384 envs 2.87 / 2.55 / 1.55
      if (!h->e3_epw) { while (epw > 16 && (h->dev.N + epw - 1) / epw < 256) epw >>= 1; }
      a.epw = (uint32_t)epw;
    }
#define BG_E3K(HV, CV, NOWV, KSV, NSVV) do { \
      const dim3 g_((NOWV * KSV == 1) ? (h->dev.N + epw - 1) / epw : (h->dev.N + NOWV * KSV * 64 - 1) / (NOWV * KSV * 64)), b_((NOWV + NSVV) * BG_BLOCK); \
      if (ev_a) hipExtLaunchKernelGGL((bg_engine3_kernel<HV, CV, NOWV, KSV, NSVV>), g_, b_, 0, st, ev_a, ev_b, 0, dv, a); \
      else hipLaunchKernelGGL((bg_engine3_kernel<HV, CV, NOWV, KSV, NSVV>), g_, b_, 0, st, dv, a); } while (0)
#define BG_E3(NOWV, KSV, NSVV) do { \
      if (hash && cards) BG_E3K(true, true, NOWV, KSV, NSVV); else if (hash) BG_E3K(true, false, NOWV, KSV, NSVV); \
      else if (cards) BG_E3K(false, true, NOWV, KSV, NSVV); else BG_E3K(false, false, NOWV, KSV, NSVV); } while (0)
    switch (cfg) {   // (other shapes were measured and dropped: profiles/r04_engine3/wave_split_ab.txt, small_jobs.txt, profiles/r05/scheduling_ab.txt)
      case 113: BG_E3(1, 1, 3); break;
      case 213: BG_E3(2, 1, 3); break;
      case 414: BG_E3(4, 1, 4); break;   // eight waves: no room for the refill beside the launch (it then runs when the workgroups retire)
      default: BG_E3(4, 1, 3); break;
    }
#undef BG_E3
#undef BG_E3K
    return;
  }
  a.n_waves = (uint32_t)bg_engine_waves(h, (int)a.T);
  // a ONE-step launch with the caller's actions (bg_step / bg_step_rows): the prologue has queued every service request of the launch before the first wave
  // enters its loop -- nothing more will arrive, so a service-capable wave takes a service queue as it finds it instead of looking at the run queue first
  if (a.T == 1 && a.actions_in) a.th_play = a.th_other = 1u;
  // packed records: one more wave, the COPIER (bg_engine.h), takes the record copy-out off the workers; with the seven-wave shape that
  // leaves room for the refill beside the launch it is one of the seven
  // ONE copier unless BG_ENG_COPIERS asks for more: a single copier drains the copy queue in order, so "one outstanding entry per env" bounds
  // tail - head by the ring size; with two or three dealing the ring out in blocks, a worker can lap a block a lagging copier has not read yet
  // (nothing publishes the copiers' heads).  Packed-record ROLLOUTS no longer come this way (bg_engine3.h); what does is bg_step_rows, a
  // one-step launch, where a second copier buys nothing.
  // ... except in a ONE-STEP launch, where every env is queued exactly once (<= 256 entries in a 256-entry ring: nothing can be lapped) and
  // the second copier is worth 10 % (bg_step_rows 3.3 -> 3.65 G env-steps/s)
  a.copier = a.obs.rows ? (uint32_t)(h->eng_copiers > 0 ? h->eng_copiers : (a.T == 1 ? 2 : 1)) : 0u;
  if (a.copier && a.n_waves > BG_ENG_NW - a.copier) a.n_waves = BG_ENG_NW - a.copier;
  if (a.n_waves < BG_ENG_NW && a.serve_mask == BG_ENG_SMASK_DEFAULT) // the last NSV of the workers (all of them when there are no more)
    a.serve_mask = a.n_waves <= BG_ENG_NSV ? (1u << a.n_waves) - 1u : ((1u << BG_ENG_NSV) - 1u) << (a.n_waves - BG_ENG_NSV);
  // a BG_ENG_SMASK that names no WORKER wave of this launch (fewer workers than the mask assumed: a short launch, copier waves) would leave the
  // service queues without a taker: fall back to the last workers
  if ((a.serve_mask & ((1u << a.n_waves) - 1u)) == 0u)
    a.serve_mask = a.n_waves <= BG_ENG_NSV ? (1u << a.n_waves) - 1u : ((1u << BG_ENG_NSV) - 1u) << (a.n_waves - BG_ENG_NSV);
  const dim3 g((h->dev.N + BG_ENG_NE - 1) / BG_ENG_NE), b(BG_ENG_NW * BG_BLOCK);
#define BG_ENG(HASHV, CARDSV, INFOV) do { \
    if (ev_a) hipExtLaunchKernelGGL((bg_engine_kernel<HASHV, CARDSV, INFOV>), g, b, 0, st, ev_a, ev_b, 0, dv, a); \
    else hipLaunchKernelGGL((bg_engine_kernel<HASHV, CARDSV, INFOV>), g, b, 0, st, dv, a); } while (0)
  if (info) { if (cards) BG_ENG(false, true, true); else BG_ENG(false, false, true); }
  else if (hash && cards) BG_ENG(true, true, false);
  else if (hash) BG_ENG(true, false, false);
  else if (cards) BG_ENG(false, true, false);
  else BG_ENG(false, false, false);
#undef BG_ENG
}

// Steps that may run between two refills when the refill is NOT overlapped (bg_step / bg_step_many): see bg_chunk_limit
static int bg_step_budget(const bg_handle* h) {
  const int ring = h->dev.KD < h->dev.KS - 1 ? h->dev.KD : h->dev.KS - 1;
  int lim = 3 * ring;
  const int g = bg_gsteps(h->dev.flags, h->dev.KG - 1);
  if (g < lim) lim = g;
  return lim < 1 ? 1 : lim;
}

// K consecutive step() calls per env, actions [K, N]: ONE launch of the step engine per refill budget.  The look-ahead rings are
// topped up only when the steps since the last refill could exhaust them (a refill per step was 5 kernel launches per step).
static int bg_step_impl(bg_handle* h, int K, const int32_t* actions_dev, const bg_obs_ptrs* obs, int obs_stride_steps, double* reward_dev,
                        uint8_t* terminated_dev, uint8_t* truncated_dev, const bg_info_ptrs* info, void* stream,
                        uint8_t* rows_dev = nullptr, uint64_t row_stride = 0) {
  int rc = bg_require_seeded(h);
  if (rc) return rc;
  BG_GUARD(h);
  if (!actions_dev || K <= 0) return BG_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int budget = bg_step_budget(h);
  const size_t N = (size_t)h->dev.N;
  int done = 0;
  while (done < K) {
    if (h->steps_since_refill >= budget) { rc = bg_refill(h, stream); if (rc) return rc; }
    int chunk = K - done;
    if (chunk > budget - h->steps_since_refill) chunk = budget - h->steps_since_refill;
    rc = bg_wait_refill(h, st, 0);
    if (rc) return rc;
    const BgDev dv = bg_dev_view(h, bg_prod_latest(h));
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    bg_ev_pair(h, h->ev_step_t, ev_a, ev_b);
    {
      EngineArgs ea;
      memset(&ea, 0, sizeof(ea));
      ea.T = chunk; ea.actions_in = actions_dev + (size_t)done * N;
      ea.obs = bg_obs(obs); ea.obs_stride_steps = obs_stride_steps;
      const size_t off = obs_stride_steps ? (size_t)done * N : 0;
      if (off) bg_obs_advance(ea.obs, off);
      if (rows_dev) { ea.obs.rows = rows_dev; ea.obs.row_stride = (uint32_t)row_stride; } // packed records: the copier waves write them
      ea.reward = reward_dev ? reward_dev + off : nullptr; ea.term = terminated_dev ? terminated_dev + off : nullptr;
      ea.trunc = truncated_dev ? truncated_dev + off : nullptr;
      ea.info = bg_info(info);
      if (off) bg_info_advance(ea.info, off);
      ea.th_run = h->eng_run; ea.th_play = h->eng_play; ea.th_other = h->eng_other; ea.th_part = h->eng_part; ea.th_more = h->eng_more; ea.serve_mask = h->eng_smask;
      ea.autoreset = (h->dev.flags & BG_FLAG_AUTORESET) ? 1u : 0u;
      bg_engine_launch(h, dv, ea, false, true, st, ev_a, ev_b);
    }
    BG_HIP(hipGetLastError());
    h->steps_since_refill += chunk;
    done += chunk;
  }
  return 0;
}

int bg_step(bg_handle* h, const int32_t* actions_dev, const bg_obs_ptrs* obs, double* reward_dev, uint8_t* terminated_dev,
            uint8_t* truncated_dev, const bg_info_ptrs* info, void* stream) {
  return bg_step_impl(h, 1, actions_dev, obs, 0, reward_dev, terminated_dev, truncated_dev, info, stream);
}

int bg_step_many(bg_handle* h, int K, const int32_t* actions_dev, const bg_obs_ptrs* obs, int obs_stride_steps, double* reward_dev,
                 uint8_t* terminated_dev, uint8_t* truncated_dev, const bg_info_ptrs* info, void* stream) {
  return bg_step_impl(h, K, actions_dev, obs, obs_stride_steps, reward_dev, terminated_dev, truncated_dev, info, stream);
}

static const char* bg_rows_args(const void* rows_dev, uint64_t row_stride_bytes) {
  if (!rows_dev || row_stride_bytes < BG_ROW_BYTES || (row_stride_bytes & 15) || ((uintptr_t)rows_dev & 15) || row_stride_bytes > 0xffffffffull)
    return "rows_dev must be 16-byte aligned and row_stride_bytes a multiple of 16, >= BG_ROW_BYTES";
  return nullptr;
}

int bg_step_rows(bg_handle* h, const int32_t* actions_dev, uint8_t* rows_dev, uint64_t row_stride_bytes, double* reward_dev,
                 uint8_t* terminated_dev, uint8_t* truncated_dev, const bg_info_ptrs* info, void* stream) {
  if (!h) return BG_E_ARG;
  if (const char* m = bg_rows_args(rows_dev, row_stride_bytes)) { h->err = std::string("bg_step_rows: ") + m; return BG_E_ARG; }
  return bg_step_impl(h, 1, actions_dev, nullptr, 0, reward_dev, terminated_dev, truncated_dev, info, stream, rows_dev, row_stride_bytes);
}

int bg_observe_rows(bg_handle* h, uint8_t* rows_dev, uint64_t row_stride_bytes, void* stream) {
  if (!h) return BG_E_ARG;
  if (const char* m = bg_rows_args(rows_dev, row_stride_bytes)) { h->err = std::string("bg_observe_rows: ") + m; return BG_E_ARG; }
  BG_GUARD(h);
  ObsPtrs o = bg_obs(nullptr);
  o.rows = rows_dev; o.row_stride = (uint32_t)row_stride_bytes;
  { const int rcw = bg_wait_refill(h, (hipStream_t)stream, 0); if (rcw) return rcw; }   // the view read below is the latest refill's: all of it must have run
  hipLaunchKernelGGL(bg_observe_kernel, dim3(bg_grid(h)), dim3(BG_BLOCK), 0, (hipStream_t)stream, bg_dev_view(h, bg_prod_latest(h)), o);
  BG_HIP(hipGetLastError());
  return 0;
}

int bg_observe(bg_handle* h, const bg_obs_ptrs* obs, void* stream) {
  if (!h) return BG_E_ARG;
  BG_GUARD(h);
  { const int rcw = bg_wait_refill(h, (hipStream_t)stream, 0); if (rcw) return rcw; }
  hipLaunchKernelGGL(bg_observe_kernel, dim3(bg_grid(h)), dim3(BG_BLOCK), 0, (hipStream_t)stream, bg_dev_view(h, bg_prod_latest(h)), bg_obs(obs));
  BG_HIP(hipGetLastError());
  return 0;
}

static int bg_rollout_impl(bg_handle* h, int T, int policy, uint64_t policy_seed, uint64_t env_index0, uint64_t t0,
                           const bg_obs_ptrs* obs, uint8_t* rows_dev, size_t row_stride, int obs_stride_steps,
                           double* reward_dev, uint8_t* terminated_dev, int32_t* actions_out_dev,
                           bg_rollout_stats* stats_dev, void* stream) {
  int rc = bg_require_seeded(h);
  if (rc) return rc;
  BG_GUARD(h);
  if (T <= 0) return BG_E_ARG;
  bool async = false;
  const int max_chunk = bg_chunk_limit(h, &async);
  // Refills are LAZY: the rings are topped up when the steps launched since the last refill (h->steps_since_refill = u) plus the
  // chunk about to run would exceed what one refill guarantees -- B = max_chunk steps' worth of every ring (half the rings when the
  // refill is overlapped, all of them otherwise) -- not after every launch: a caller that fuses 20 steps per launch pays for a
  // refill every 18 launches instead of one per launch (it was 160 us exposed behind every 380 us launch).
  //   overlapped: the refill R is queued on the side stream when the chunk STARTS, ordered after everything the stream has done
  //     so far, and runs beside the chunk; the chunk reads the producer counters of the refill BEFORE R, against which it has
  //     consumed u + chunk <= 2 B steps (u <= B is the invariant: after a chunk u is either u + chunk <= B or chunk).
  //   synchronous: R runs on the stream before the chunk.
  // bg_step / bg_step_many budget against the whole ring, so they may leave u > B: then one synchronous refill first.
  if (async && h->steps_since_refill > max_chunk) { rc = bg_refill(h, stream); if (rc) return rc; }
  int done = 0;
  while (done < T) {
    int chunk = T - done < max_chunk ? T - done : max_chunk;
    ObsPtrs o = bg_obs(obs);
    size_t off = obs_stride_steps ? (size_t)done * (size_t)h->dev.N : 0;
    if (rows_dev) { o.rows = rows_dev + off * row_stride; o.row_stride = (uint32_t)row_stride; }
    if (off) bg_obs_advance(o, off);
    if (h->profiling) h->rollout_steps.push_back(chunk);
    // (BG_REFILL_MIN = m > 0 makes them eager: a launch of >= m steps takes its own small refill beside it -- measured, not the default: bg_create_ex.)
    const bool must = h->steps_since_refill + chunk > max_chunk;
    const bool need = must || (async && h->refill_min > 0 && h->steps_since_refill + chunk >= h->refill_min);
    const long s0 = h->refill_seq;
    // Overlapped and the chunk reads the view of the refill BEFORE this one (the usual case): the engine is launched FIRST, the refill's six small
    // kernels are queued on the side stream behind it -- their host-side issue time (~30 us) is then not in front of a 220 us launch.
    const bool refill_after = need && async && s0 >= 1 && s0 - 1 >= h->view_min;
    // Pieces of an earlier refill still to be issued (BG_REFILL_SLICED): this launch takes its share beside it, queued BEHIND the launch and ordered after
    // the launches before it -- a piece then starts while this launch's workgroups are resident and fills what they leave free
    const bool pieces_after = !need && async && h->piece_next < h->pieces.size();
    if (pieces_after) BG_HIP(hipEventRecord(h->ev_rollout, (hipStream_t)stream));
    if (need && async) {
      BG_HIP(hipEventRecord(h->ev_rollout, (hipStream_t)stream));   // everything the stream has done so far (the previous launches)
      // Pieces of the latest refill still to be issued (launch lengths that changed inside a period): this launch reads THAT refill's view, so they go now,
      // in front of it -- the completion event behind the last piece is what the launch waits for below (a view whose pieces were still to come would
      // be read beside its writers, and its event would still be the one of two refills ago)
      if (h->piece_next < h->pieces.size()) {
        BG_HIP(hipStreamWaitEvent(h->side, h->ev_rollout, 0));
        rc = bg_refill_pieces(h, 1 << 30);
        if (rc) return rc;
      }
      if (!refill_after) { // R beside this chunk, queued before it (the chunk reads R's own view: right behind a synchronous refill)
        BG_HIP(hipStreamWaitEvent(h->side, h->ev_rollout, 0));
        rc = bg_wait_refill(h, h->side, 0);
        if (rc) return rc;
        rc = bg_refill_on(h, h->side, h->steps_since_refill + chunk); // (sets steps_since_refill = 0)
        if (rc) return rc;
      }
    } else if (need) {
      rc = bg_refill(h, stream);
      if (rc) return rc;
    }
    // Which producer counters the chunk reads.  Overlapped: those of the refill BEFORE the latest one -- the latest may still be
    // running (beside this chunk, or beside an earlier short one), and against the one before it this chunk has consumed at most
    // (steps between the two refills <= B) + u + chunk <= 2 B.  Never one older than the last SYNCHRONOUS refill (bg_reset, bg_step,
    // ...: the steps before those are not bounded by B).  Synchronous: the latest.
    long vi = h->refill_seq - 1;
    if (refill_after) vi = s0 - 1;   // (R, queued below, will be refill s0: the one before it)
    else if (async && vi - 1 >= h->view_min) vi--;
    // (a refill is long complete for all but the first launch or two that read its view: once hipEventQuery has said so, no wait is queued -- a
    //  cross-stream wait is a barrier packet in front of every launch, a few microseconds of a 230 us one.  vi >= refill_seq - 2, so the event of
    //  this parity still belongs to refill vi.)
    if (vi >= 0 && vi > h->refill_done) {
      if (hipEventQuery(h->ev_refill[vi & 1]) == hipSuccess) h->refill_done = vi;
      else BG_HIP(hipStreamWaitEvent((hipStream_t)stream, h->ev_refill[vi & 1], 0));
    }
    const uint32_t* view = vi >= 0 ? h->d_prod[vi & 1] : bg_prod_latest(h);
    hipEvent_t ev_a = nullptr, ev_b = nullptr;   // the events bracket the kernel, not the stream's wait for the refill
    bg_ev_pair(h, h->ev_rollout_t, ev_a, ev_b);
    BgDev dv = bg_dev_view(h, view);
    {
      const bool hash = (policy & BG_POLICY_HASH_OBS) != 0;
      const int pol = policy & 0xff;
      double* rw = reward_dev ? reward_dev + off : nullptr;
      uint8_t* tm = terminated_dev ? terminated_dev + off : nullptr;
      int32_t* ac = actions_out_dev ? actions_out_dev + off : nullptr;
      hipStream_t st = (hipStream_t)stream;
      uint64_t tt = t0 + (uint64_t)done;
      EngineArgs ea;
      memset(&ea, 0, sizeof(ea));
      ea.T = chunk; ea.policy = pol; ea.policy_seed = policy_seed; ea.env_index0 = env_index0; ea.t0 = tt;
      ea.obs = o; ea.obs_stride_steps = obs_stride_steps; ea.reward = rw; ea.term = tm; ea.actions_out = ac; ea.stats = stats_dev;
      ea.th_run = h->eng_run; ea.th_play = h->eng_play; ea.th_other = h->eng_other; ea.th_part = h->eng_part; ea.th_more = h->eng_more; ea.serve_mask = h->eng_smask; ea.autoreset = 1;
      if (h->gworld > 0 && rows_dev && h->engine == 3 && done + chunk == T) { // the call's LAST launch: its last step is every env's current record
        ea.gpeer = h->d_gpeer;
        ea.gworld = (uint32_t)h->gworld; ea.grank = (uint32_t)h->grank;
      }
      bg_engine_launch(h, dv, ea, hash, false, st, ev_a, ev_b);
    }
    BG_HIP(hipGetLastError());
    if (refill_after) { // R beside this chunk: after everything the stream had done BEFORE the chunk (ev_rollout) and after the latest refill
      BG_HIP(hipStreamWaitEvent(h->side, h->ev_rollout, 0));
      rc = bg_wait_refill(h, h->side, 0);
      if (rc) return rc;
      // A launch of at most HALF a refill period (the refill's kernels would run beside the next launches too, and whatever of them is still to be placed
      // when a launch ends takes the whole machine in the gap before the next one): the scan now, the dense kernels in pieces beside this launch and the ones
      // to come.  A launch that is a period of its own keeps the whole refill beside it (in pieces the refill -- held to what is resident beside the engine --
      // takes longer than the launch, and the next launch needs it: 372 steps 8.1 -> 7.4 G, profiles/r05/refill_pieces.txt).
      const bool sliced = h->refill_sliced != 0 && h->refill_sliced_div * chunk <= max_chunk;
      rc = bg_refill_on(h, h->side, h->steps_since_refill + chunk, sliced); // (sets steps_since_refill = 0: the chunk beside it counts below)
      if (rc) return rc;
    }
    if ((refill_after || pieces_after) && h->piece_next < h->pieces.size()) {
      // This launch's share: the pieces still to go over the launches still to come before the next refill is due (this one included, launches of this
      // length assumed), the scan counted as a piece of the launch it runs beside -- one piece per launch at 20 steps, eight or nine at 180, and none
      // left for the launch that asks for the next refill (which waits for this one)
      const int u = h->steps_since_refill;   // (0 behind a refill that was just queued)
      int m = 1 + (max_chunk - (u + chunk)) / (chunk > 0 ? chunk : 1);
      if (m < 1) m = 1;
      const int pending = (int)(h->pieces.size() - h->piece_next), extra = refill_after ? 1 : 0;
      const int np = (pending + extra + m - 1) / m - extra;
      if (np > 0) {
        if (!refill_after) BG_HIP(hipStreamWaitEvent(h->side, h->ev_rollout, 0));   // (the scan in front of the pieces already waits for it)
        rc = bg_refill_pieces(h, np);
        if (rc) return rc;
      }
    }
    h->steps_since_refill += chunk;
    done += chunk;
  }
  return 0;
}

int bg_set_gather_peers(bg_handle* h, void* const* bufs, int world, int rank) {
  if (!h) return BG_E_ARG;
  if (world == 0) { h->gworld = 0; return 0; }
  if (!bufs || world < 1 || world > 8 || rank < 0 || rank >= world) { h->err = "bg_set_gather_peers: world must be 1..8, rank inside it, bufs the world gather buffers"; return BG_E_ARG; }
  if (h->engine != 3) { h->err = "bg_set_gather_peers: only bg_engine3.h's packed-record rollouts write gather buffers (BG_ENGINE=3)"; return BG_E_ARG; }
  BG_GUARD(h);
  for (int g = 0; g < world; g++) {
    if (!bufs[g] || ((uintptr_t)bufs[g] & 15)) { h->err = "bg_set_gather_peers: every gather buffer must be a 16-byte aligned device pointer"; return BG_E_ARG; }
    // A buffer that lives on ANOTHER device (a peer's, opened from its IPC handle) is written by this device's kernels: that needs peer access, and a
    // store without it is a GPU memory fault, not an error code -- so it is established here, or the call fails and the caller keeps its collective
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, bufs[g]) != hipSuccess) { (void)hipGetLastError(); h->err = "bg_set_gather_peers: a gather buffer is not a device allocation this process knows"; return BG_E_ARG; }
    if (at.device != h->device_id) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, h->device_id, at.device) != hipSuccess || !can) { (void)hipGetLastError(); h->err = "bg_set_gather_peers: this device cannot access a peer's gather buffer (no peer access between the two devices)"; return BG_E_ARG; }
      const hipError_t pe = hipDeviceEnablePeerAccess(at.device, 0);
      if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); h->err = std::string("bg_set_gather_peers: hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe); return BG_E_HIP; }
      (void)hipGetLastError();
    }
  }
  for (int g = 0; g < 8; g++) h->gpeer[g] = g < world ? (uint8_t*)bufs[g] : nullptr;
  if (!h->d_gpeer) BG_HIP(hipMalloc((void**)&h->d_gpeer, sizeof(h->gpeer)));
  BG_HIP(hipDeviceSynchronize());   // no launch of this handle may still be reading the old pointers
  BG_HIP(hipMemcpy(h->d_gpeer, h->gpeer, sizeof(h->gpeer), hipMemcpyHostToDevice));
  h->gworld = world; h->grank = rank;
  return 0;
}

int bg_rollout(bg_handle* h, int T, int policy, uint64_t policy_seed, uint64_t env_index0, uint64_t t0,
               const bg_obs_ptrs* obs, int obs_stride_steps, double* reward_dev, uint8_t* terminated_dev,
               int32_t* actions_out_dev, bg_rollout_stats* stats_dev, void* stream) {
  return bg_rollout_impl(h, T, policy, policy_seed, env_index0, t0, obs, nullptr, 0, obs_stride_steps, reward_dev,
                         terminated_dev, actions_out_dev, stats_dev, stream);
}

int bg_rollout_rows(bg_handle* h, int T, int policy, uint64_t policy_seed, uint64_t env_index0, uint64_t t0,
                    uint8_t* rows_dev, uint64_t row_stride_bytes, int rows_stride_steps, bg_rollout_stats* stats_dev,
                    void* stream) {
  if (!h) return BG_E_ARG;
  if ((uint64_t)h->dev.N * (uint64_t)bg_max_fused_steps(h) > 0xffffffffull) { h->err = "bg_rollout_rows: more than 2**32 records per launch (envs x fused steps)"; return BG_E_ARG; }
  if (!rows_dev || row_stride_bytes < BG_ROW_BYTES || (row_stride_bytes & 15) || ((uintptr_t)rows_dev & 15) || row_stride_bytes > 0xffffffffull) {
    h->err = "bg_rollout_rows: rows_dev must be 16-byte aligned and row_stride_bytes a multiple of 16, >= BG_ROW_BYTES";
    return BG_E_ARG;
  }
  return bg_rollout_impl(h, T, policy, policy_seed, env_index0, t0, nullptr, rows_dev, (size_t)row_stride_bytes,
                         rows_stride_steps, nullptr, nullptr, nullptr, stats_dev, stream);
}

__global__ __launch_bounds__(BG_BLOCK) void bg_inject_cards_kernel(BgDev d, const uint16_t* __restrict__ cs, const uint8_t* __restrict__ mask_in, int apply_now) {
  int env = blockIdx.x * BG_BLOCK + threadIdx.x;
  if (env >= d.N) return;
  if (mask_in && !mask_in[env]) return;
  const uint4* src = (const uint4*)(cs + (size_t)env * 56); // 52 states padded to 56 u16 = 7 x 16 bytes
#pragma unroll
  for (int k = 0; k < BG_NCST; k++) {
    uint4 v = src[k];
    d.ctmpl[(size_t)k * d.N + env] = v;
    if (apply_now) d.cstate[(size_t)k * d.N + env] = v;
  }
}

int bg_inject_cards(bg_handle* h, const uint8_t* enh_host, const uint8_t* edition_host, const uint8_t* seal_host,
                    const uint8_t* mask_host, int apply_now, void* stream) {
  if (!h) return BG_E_ARG;
  if (!h->dev.cstate) { h->err = "bg_inject_cards: the handle was created without BG_FLAG_CARD_STATES"; return BG_E_ARG; }
  BG_GUARD(h);
  hipStream_t s = (hipStream_t)stream;
  const size_t N = h->dev.N;
  std::vector<uint16_t> packed(N * 56, 0);
  for (size_t i = 0; i < N; i++)
    for (int c = 0; c < 52; c++) {
      const uint32_t e = enh_host ? enh_host[i * 52 + c] : 0, d = edition_host ? edition_host[i * 52 + c] : 0, sl = seal_host ? seal_host[i * 52 + c] : 0;
      if (e > 15 || d > 15 || sl > 15) { h->err = "bg_inject_cards: state codes are 0..15"; return BG_E_ARG; }
      packed[i * 56 + c] = (uint16_t)(e | (d << 4) | (sl << 8));
    }
  uint16_t* dcs = nullptr;
  BG_HIP(hipMalloc(&dcs, packed.size() * sizeof(uint16_t)));
  BG_HIP(hipMemcpyAsync(dcs, packed.data(), packed.size() * sizeof(uint16_t), hipMemcpyHostToDevice, s));
  if (mask_host) BG_HIP(hipMemcpyAsync(h->d_mask, mask_host, N, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(bg_inject_cards_kernel, dim3(bg_grid(h)), dim3(BG_BLOCK), 0, s, h->dev, (const uint16_t*)dcs,
                     mask_host ? (const uint8_t*)h->d_mask : (const uint8_t*)nullptr, apply_now);
  BG_HIP(hipGetLastError());
  BG_HIP(hipStreamSynchronize(s));
  BG_HIP(hipFree(dcs));
  return 0;
}

int bg_inject(bg_handle* h, const int32_t* jokers_host, const int32_t* njokers_host, const int64_t* money_host,
              const int32_t* ante_host, const uint8_t* levels_host, const uint8_t* mask_host, int apply_now, void* stream) {
  if (!h) return BG_E_ARG;
  BG_GUARD(h);
  hipStream_t s = (hipStream_t)stream;
  size_t N = h->dev.N;
  for (size_t i = 0; i < N; i++) {
    if (mask_host && !mask_host[i]) continue;
    uint4& t0 = h->h_tmpl[i];
    uint4& t1 = h->h_tmpl[N + i];
    uint32_t fl = t0.y >> 24;
    uint32_t j4 = t0.y & 0xffu, nj = (t0.y >> 8) & 0xffu, an = (t0.y >> 16) & 0xffu;
    if (jokers_host && njokers_host) {
      int n = njokers_host[i];
      if (n < 0 || n > 5) { h->err = "bg_inject: njokers must be in [0,5]"; return BG_E_ARG; }
      uint32_t lo = 0; j4 = 0;
      for (int k = 0; k < n; k++) {
        int id = jokers_host[i * 5 + k];
        if (id < 1 || id > 150) { h->err = "bg_inject: joker id out of range"; return BG_E_ARG; }
        if (k < 4) lo |= (uint32_t)id << (8 * k); else j4 = (uint32_t)id;
      }
      t0.x = lo; nj = (uint32_t)n; fl |= 0x80u;
    }
    if (money_host) { if (money_host[i] >= 0) { t0.z = (uint32_t)(int32_t)money_host[i]; fl |= 0x40u; } else fl &= ~0x40u; }
    if (ante_host) {
      // an env that resets into ante > cap ends every episode on its first step: one pre-shuffled deck per STEP, three times
      // what the look-ahead rings are sized for (bg_chunk_limit)
      if (ante_host[i] > 255 || (h->h_cap[i] > 0 && ante_host[i] > (int)h->h_cap[i])) { h->err = "bg_inject: template ante above the env's curriculum cap (max_ante)"; return BG_E_ARG; }
      if (ante_host[i] > 0) { an = (uint32_t)ante_host[i] & 0xffu; fl |= 0x20u; } else fl &= ~0x20u;
    }
    if (levels_host) {
      uint64_t lv = 0; bool any = false;
      for (int k = 0; k < 12; k++) { int l = levels_host[i * 12 + k]; if (l) any = true; if (l < 1) l = 1; if (l > 15) l = 15; lv |= (uint64_t)l << (4 * k); }
      if (any) { t1.x = (uint32_t)lv; t1.y = (uint32_t)(lv >> 32); fl |= 0x10u; } else fl &= ~0x10u;
    }
    t0.y = j4 | (nj << 8) | (an << 16) | (fl << 24);
  }
  BG_HIP(hipMemcpyAsync(h->dev.tmpl, h->h_tmpl.data(), BG_NTMPL * N * sizeof(uint4), hipMemcpyHostToDevice, s));
  if (apply_now) {
    if (mask_host) BG_HIP(hipMemcpyAsync(h->d_mask, mask_host, N, hipMemcpyHostToDevice, s));
    { const int rcw = bg_wait_refill(h, s, 0); if (rcw) return rcw; }
    hipLaunchKernelGGL(bg_inject_kernel, dim3(bg_grid(h)), dim3(BG_BLOCK), 0, s, bg_dev_view(h, bg_prod_latest(h)),
                       mask_host ? (const uint8_t*)h->d_mask : (const uint8_t*)nullptr);
    BG_HIP(hipGetLastError());
  }
  BG_HIP(hipStreamSynchronize(s));
  return 0;
}

int bg_inject_consumables(bg_handle* h, const int32_t* ids_host, const int32_t* n_host, const uint8_t* mask_host, int apply_now,
                          void* stream) {
  if (!h || !ids_host || !n_host) return BG_E_ARG;
  BG_GUARD(h);
  hipStream_t s = (hipStream_t)stream;
  size_t N = h->dev.N;
  for (size_t i = 0; i < N; i++) {
    if (mask_host && !mask_host[i]) continue;
    int n = n_host[i];
    if (n < 0) { h->h_tmpl[N + i].z = 0; continue; } // back to the reset default (no consumables)
    if (n > 2) { h->err = "bg_inject_consumables: an env holds at most consumable_slots = 2"; return BG_E_ARG; }
    uint32_t v = 0x80000000u | (uint32_t)n;
    for (int k = 0; k < n; k++) {
      int id = ids_host[i * 2 + k];
      bool planet = id >= 30 && id <= 41, other = (id >= 1 && id <= 22) || (id >= 50 && id <= 67);
      if (!planet && !other) { h->err = "bg_inject_consumables: ids are 1-22 (tarots), 30-41 (planets), 50-67 (spectrals)"; return BG_E_ARG; }
      if (other && !h->dev.cstate) { h->err = "bg_inject_consumables: tarot / spectral cards need BG_FLAG_CARD_STATES"; return BG_E_ARG; }
      v |= (uint32_t)id << (8 * (k + 1));
    }
    h->h_tmpl[N + i].z = v;
  }
  BG_HIP(hipMemcpyAsync(h->dev.tmpl, h->h_tmpl.data(), BG_NTMPL * N * sizeof(uint4), hipMemcpyHostToDevice, s));
  if (apply_now) {
    if (mask_host) BG_HIP(hipMemcpyAsync(h->d_mask, mask_host, N, hipMemcpyHostToDevice, s));
    { const int rcw = bg_wait_refill(h, s, 0); if (rcw) return rcw; }
    hipLaunchKernelGGL(bg_inject_kernel, dim3(bg_grid(h)), dim3(BG_BLOCK), 0, s, bg_dev_view(h, bg_prod_latest(h)),
                       mask_host ? (const uint8_t*)h->d_mask : (const uint8_t*)nullptr);
    BG_HIP(hipGetLastError());
  }
  BG_HIP(hipStreamSynchronize(s));
  return 0;
}

// ---- save_state / load_state: raw per-env slices of every array, in a fixed order ----
struct BgSlice { void* base; size_t rows; size_t elem; };
static void bg_slices(bg_handle* h, std::vector<BgSlice>& v) {
  BgDev& d = h->dev;
  v.push_back({d.hot, BG_NHOT, 16}); v.push_back({d.deck, BG_NDECK, 16}); v.push_back({d.cold, BG_NCOLD, 16});
  v.push_back({d.tmpl, BG_NTMPL, 16}); v.push_back({d.ndeck, (size_t)d.KD * BG_NDECK, 16});
  // per-env contiguous MT blocks: one "row" of KG*2560 / KS*2560 / 2560 bytes at base + env * elem
  v.push_back({d.gblk, 1, (size_t)d.KG * BG_MTS * 4}); v.push_back({d.sblk, 1, (size_t)d.KS * BG_SLOT_WORDS * 4}); v.push_back({d.sovf, 1, (size_t)BG_MTS * 4});
  v.push_back({d.deckmt, 1, (size_t)BG_MTS * 4}); v.push_back({d.shopgenmt, 1, (size_t)BG_MTS * 4});
  v.push_back({d.sseed, 1, BG_SSEED * 4}); v.push_back({d.smeta, 1, 4});
  v.push_back({bg_prod_latest(h), 1, 4});
  if (d.cstate) { v.push_back({d.cstate, BG_NCST, 16}); v.push_back({d.ctmpl, BG_NCST, 16}); v.push_back({d.cardmt, 1, (size_t)BG_MTS * 4}); v.push_back({d.sealmt, 1, (size_t)BG_MTS * 4}); }
}
uint64_t bg_state_blob_bytes(const bg_handle* h) {
  if (!h) return 0;
  std::vector<BgSlice> v;
  bg_slices(const_cast<bg_handle*>(h), v);
  uint64_t b = 16; // header: magic, version, KG, KS/KD
  for (auto& s : v) b += s.rows * s.elem;
  return b;
}
#define BG_BLOB_MAGIC 0x42474d58u
#define BG_BLOB_VERSION 7u // 7: a shop-stream slot = 56 output words + the packed top bytes of the first 24 + the seed; 6: shop-stream slots hold finished output words (256 bytes); 5: 576-byte shop-stream slots; 4: the curriculum cap in the hot state, card-state flag in the header; 3: compact shop-stream slots + overflow block; 2: stream 13 ('seal_applications') joins the card-state slices, consumables in the reset template
// what is wrong with a blob handed to bg_set_state (or with the buffer handed to bg_get_state), as text
static int bg_blob_args(bg_handle* h, const char* fn, int env_index, const void* blob, uint64_t blob_bytes) {
  if (!h) return BG_E_ARG;
  char buf[200];
  if (!blob) { h->err = std::string(fn) + ": null blob"; return BG_E_ARG; }
  if (env_index < 0 || env_index >= h->dev.N) { snprintf(buf, sizeof(buf), "%s: env_index %d out of range [0, %d)", fn, env_index, h->dev.N); h->err = buf; return BG_E_ARG; }
  const uint64_t want = bg_state_blob_bytes(h);
  if (blob_bytes < want) {
    snprintf(buf, sizeof(buf), "%s: blob of %llu bytes, this handle's blobs are %llu bytes (ring depths and card-state flag decide the size)", fn,
             (unsigned long long)blob_bytes, (unsigned long long)want);
    h->err = buf; return BG_E_ARG;
  }
  return 0;
}
int bg_get_state(bg_handle* h, int env_index, void* blob_host, uint64_t blob_bytes) {
  int rc = bg_blob_args(h, "bg_get_state", env_index, blob_host, blob_bytes);
  if (rc) return rc;
  BG_GUARD(h);
  // a refill issued in pieces (bg_refill_pieces) has already advanced the producer counters the blob copies (the scan did): the dense kernels that fill
  // those ring slots must have RUN before the state is read -- issue what is pending, then wait for the device
  rc = bg_refill_pieces(h, 1 << 30);
  if (rc) return rc;
  BG_HIP(hipDeviceSynchronize());
  uint8_t* out = (uint8_t*)blob_host;
  uint32_t hdr[4] = {BG_BLOB_MAGIC, BG_BLOB_VERSION, (uint32_t)h->dev.KG | (h->dev.cstate ? 0x10000u : 0u), (uint32_t)h->dev.KS | ((uint32_t)h->dev.KD << 16)};
  memcpy(out, hdr, 16); out += 16;
  std::vector<BgSlice> v;
  bg_slices(h, v);
  size_t N = h->dev.N;
  for (auto& s : v) {
    BG_HIP(hipMemcpy2D(out, s.elem, (uint8_t*)s.base + (size_t)env_index * s.elem, N * s.elem, s.elem, s.rows, hipMemcpyDeviceToHost));
    out += s.rows * s.elem;
  }
  return 0;
}
int bg_set_state(bg_handle* h, int env_index, const void* blob_host, uint64_t blob_bytes) {
  int rc = bg_blob_args(h, "bg_set_state", env_index, blob_host, blob_bytes);
  if (rc) return rc;
  BG_GUARD(h);
  const uint8_t* in = (const uint8_t*)blob_host;
  uint32_t hdr[4];
  memcpy(hdr, in, 16); in += 16;
  char buf[200];
  if (hdr[0] != BG_BLOB_MAGIC) { h->err = "bg_set_state: not a state blob (bad magic)"; return BG_E_ARG; }
  if (hdr[1] != BG_BLOB_VERSION) { snprintf(buf, sizeof(buf), "bg_set_state: blob version %u, this library reads version %u", hdr[1], BG_BLOB_VERSION); h->err = buf; return BG_E_ARG; }
  if (hdr[2] != ((uint32_t)h->dev.KG | (h->dev.cstate ? 0x10000u : 0u)) || hdr[3] != ((uint32_t)h->dev.KS | ((uint32_t)h->dev.KD << 16))) {
    snprintf(buf, sizeof(buf), "bg_set_state: the blob was saved with ring depths %u/%u/%u%s, this handle has %d/%d/%d%s", hdr[2] & 0xffffu, hdr[3] & 0xffffu, hdr[3] >> 16,
             (hdr[2] & 0x10000u) ? " + card states" : "", h->dev.KG, h->dev.KS, h->dev.KD, h->dev.cstate ? " + card states" : "");
    h->err = buf; return BG_E_ARG;
  }
  // pending refill pieces hold work items of the state that is about to be replaced (shop seeds listed by the scan): they must write their ring slots
  // BEFORE the blob lands, not on top of it
  rc = bg_refill_pieces(h, 1 << 30);
  if (rc) return rc;
  BG_HIP(hipDeviceSynchronize());
  std::vector<BgSlice> v;
  bg_slices(h, v);
  size_t N = h->dev.N;
  for (auto& s : v) {
    BG_HIP(hipMemcpy2D((uint8_t*)s.base + (size_t)env_index * s.elem, N * s.elem, in, s.elem, s.elem, s.rows, hipMemcpyHostToDevice));
    if (s.base == (void*)h->dev.tmpl) // the host mirror of the reset template follows (later bg_inject calls upload the whole mirror)
      for (size_t r = 0; r < s.rows; r++) memcpy(&h->h_tmpl[r * N + (size_t)env_index], in + r * s.elem, sizeof(uint4));
    if (s.base == (void*)h->dev.hot) { uint4 c7; memcpy(&c7, in + 7 * s.elem, sizeof(uint4)); h->h_cap[env_index] = (uint8_t)((c7.w >> 16) & 0xffu); } // the cap travels in the blob
    in += s.rows * s.elem;
  }
  { // both producer-counter buffers must agree for this env (an overlapped rollout reads the older one)
    uint32_t w = 0;
    BG_HIP(hipMemcpy(&w, bg_prod_latest(h) + env_index, 4, hipMemcpyDeviceToHost));
    BG_HIP(hipMemcpy(h->d_prod[0] + env_index, &w, 4, hipMemcpyHostToDevice));
    BG_HIP(hipMemcpy(h->d_prod[1] + env_index, &w, 4, hipMemcpyHostToDevice));
  }
  h->seeded = true;
  h->steps_since_refill = 1 << 30; // the restored env's look-ahead is as deep as it was when the blob was taken: top up before the next step
  return 0;
}

// ---- harness injection of the LIVE deck order and of the curriculum cap ----
int bg_inject_deck(bg_handle* h, const uint8_t* decks_host, const uint8_t* mask_host, void* stream) {
  if (!h || !decks_host) return BG_E_ARG;
  BG_GUARD(h);
  hipStream_t s = (hipStream_t)stream;
  const size_t N = h->dev.N;
  std::vector<uint8_t> packed(N * 64, 0);
  for (size_t i = 0; i < N; i++) {
    if (mask_host && !mask_host[i]) continue;
    uint64_t seen = 0;
    for (int c = 0; c < 52; c++) {
      const uint8_t v = decks_host[i * 52 + c];
      if (v >= 52 || ((seen >> v) & 1ull)) { h->err = "bg_inject_deck: every deck must be a permutation of the 52 card codes (rank-2)*4+suit"; return BG_E_ARG; }
      seen |= 1ull << v;
      packed[i * 64 + c] = v;
    }
  }
  uint4* dd = nullptr;
  BG_HIP(hipMalloc((void**)&dd, packed.size()));
  BG_HIP(hipMemcpyAsync(dd, packed.data(), packed.size(), hipMemcpyHostToDevice, s));
  if (mask_host) BG_HIP(hipMemcpyAsync(h->d_mask, mask_host, N, hipMemcpyHostToDevice, s));
  { int rcw = bg_wait_refill(h, s, 0); if (rcw) { (void)hipFree(dd); return rcw; } }
  hipLaunchKernelGGL(bg_inject_deck_kernel, dim3(bg_grid(h)), dim3(BG_BLOCK), 0, s, h->dev, (const uint4*)dd,
                     mask_host ? (const uint8_t*)h->d_mask : (const uint8_t*)nullptr);
  BG_HIP(hipGetLastError());
  BG_HIP(hipStreamSynchronize(s));
  BG_HIP(hipFree(dd));
  return 0;
}

int bg_set_max_ante(bg_handle* h, int max_ante, const int32_t* per_env_host, const uint8_t* mask_host, void* stream) {
  if (!h) return BG_E_ARG;
  BG_GUARD(h);
  hipStream_t s = (hipStream_t)stream;
  const size_t N = h->dev.N;
  if (!per_env_host && (max_ante < 0 || max_ante > 255)) { h->err = "bg_set_max_ante: the cap must be in [0, 255] (0 = none)"; return BG_E_ARG; }
  int32_t* dc = nullptr;
  for (size_t i = 0; i < N; i++) {
    if (mask_host && !mask_host[i]) continue;
    const int c = per_env_host ? per_env_host[i] : max_ante;
    if (c < 0 || c > 255) { h->err = "bg_set_max_ante: every cap must be in [0, 255] (0 = none)"; return BG_E_ARG; }
    // an env whose reset template puts it ABOVE its cap would end every episode on its first step: one pre-shuffled deck per step,
    // three times what the look-ahead rings are budgeted for (bg_chunk_limit) -- refused here as in bg_inject
    const uint32_t ty = h->h_tmpl[i].y;
    if (c > 0 && ((ty >> 24) & 0x20u) && (int)((ty >> 16) & 0xffu) > c) { h->err = "bg_set_max_ante: a cap below the env's template ante (bg_inject)"; return BG_E_ARG; }
  }
  for (size_t i = 0; i < N; i++)
    if (!mask_host || mask_host[i]) h->h_cap[i] = (uint8_t)(per_env_host ? per_env_host[i] : max_ante);
  if (per_env_host) {
    BG_HIP(hipMalloc((void**)&dc, N * sizeof(int32_t)));
    BG_HIP(hipMemcpyAsync(dc, per_env_host, N * sizeof(int32_t), hipMemcpyHostToDevice, s));
  }
  if (mask_host) BG_HIP(hipMemcpyAsync(h->d_mask, mask_host, N, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(bg_set_cap_kernel, dim3(bg_grid(h)), dim3(BG_BLOCK), 0, s, h->dev, (const int32_t*)dc, max_ante,
                     mask_host ? (const uint8_t*)h->d_mask : (const uint8_t*)nullptr);
  BG_HIP(hipGetLastError());
  BG_HIP(hipStreamSynchronize(s));
  if (dc) BG_HIP(hipFree(dc));
  if (!per_env_host && !mask_host) h->dev.max_ante = max_ante; // the handle-wide default (template antes are validated per env: h_cap)
  return 0;
}

// ---- operator-level entry points (no handle: they run on the CURRENT device, on caller-owned device buffers) ----
#define BG_HIP0(call)                                                                                  \
  do {                                                                                                 \
    hipError_t _e = (call);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      g_create_err = std::string(#call) + ": " + hipGetErrorString(_e);                                \
      return BG_E_HIP;                                                                                 \
    }                                                                                                  \
  } while (0)

// ---- measurement hook: streaming copy (bench.py's `peak_measured`: what this GPU's HBM sustains for a plain 16-byte-per-lane
// copy, the yardstick SURVEY 8(d) asks the roofline fraction to be quoted against beside the 8 TB/s nominal) ----
// Shape from a sweep on the MI355X (tools/micro/copybw.hip, profiles/r03_copybw.txt): ONE pass per workgroup (no grid-stride loop), four
// 16-byte pieces per lane, non-temporal loads and stores: 6.5 TB/s read + written at 1 GiB (the guide's float4 copy: 6.29; the
// grid-stride kernel of rounds 1-2: 5.1-5.6).  A plain one-pass fill writes 6.8 TB/s.
typedef uint32_t bg_cp_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void bg_stream_copy_kernel(const bg_cp_u32x4* __restrict__ src, bg_cp_u32x4* __restrict__ dst, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x;
  bg_cp_u32x4 v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) if (i + (size_t)k * 256 < n16) v[k] = __builtin_nontemporal_load(&src[i + (size_t)k * 256]);
#pragma unroll
  for (int k = 0; k < 4; k++) if (i + (size_t)k * 256 < n16) __builtin_nontemporal_store(v[k], &dst[i + (size_t)k * 256]);
}
int bg_bench_copy(const void* src_dev, void* dst_dev, uint64_t bytes, int iters, double* gbps_out, void* stream) {
  if (!src_dev || !dst_dev || !gbps_out || bytes < 16 || iters < 1 || ((uintptr_t)src_dev & 15) || ((uintptr_t)dst_dev & 15)) { g_create_err = "bg_bench_copy: bad arguments"; return BG_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const size_t n16 = bytes / 16;
  const unsigned grid = (unsigned)((n16 + 1023) / 1024);
  hipEvent_t a, b;
  BG_HIP0(hipEventCreate(&a)); BG_HIP0(hipEventCreate(&b));
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL(bg_stream_copy_kernel, dim3(grid), dim3(256), 0, s, (const bg_cp_u32x4*)src_dev, (bg_cp_u32x4*)dst_dev, n16); // warm-up
  BG_HIP0(hipEventRecord(a, s));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(bg_stream_copy_kernel, dim3(grid), dim3(256), 0, s, (const bg_cp_u32x4*)src_dev, (bg_cp_u32x4*)dst_dev, n16);
  BG_HIP0(hipEventRecord(b, s));
  BG_HIP0(hipGetLastError());
  BG_HIP0(hipEventSynchronize(b));
  float ms = 0;
  BG_HIP0(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  *gbps_out = ms > 0 ? 2.0 * (double)(n16 * 16) * iters / (ms * 1e-3) / 1e9 : 0.0; // bytes read + bytes written
  return 0;
}

// write-only twin: the step engine's traffic is ~80 % stores (the record of every step), so the store bandwidth is its real ceiling
__global__ __launch_bounds__(256) void bg_stream_fill_kernel(uint4* __restrict__ dst, size_t n16, uint32_t seed) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = make_uint4(seed, (uint32_t)i, 0u, seed ^ (uint32_t)i);
}
int bg_bench_fill(void* dst_dev, uint64_t bytes, int iters, double* gbps_out, void* stream) {
  if (!dst_dev || !gbps_out || bytes < 16 || iters < 1 || ((uintptr_t)dst_dev & 15)) { g_create_err = "bg_bench_fill: bad arguments"; return BG_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const size_t n16 = bytes / 16;
  const unsigned grid = (unsigned)((n16 + 255) / 256);
  hipEvent_t a, b;
  BG_HIP0(hipEventCreate(&a)); BG_HIP0(hipEventCreate(&b));
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL(bg_stream_fill_kernel, dim3(grid), dim3(256), 0, s, (uint4*)dst_dev, n16, 0u); // warm-up
  BG_HIP0(hipEventRecord(a, s));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(bg_stream_fill_kernel, dim3(grid), dim3(256), 0, s, (uint4*)dst_dev, n16, (uint32_t)i + 1u);
  BG_HIP0(hipEventRecord(b, s));
  BG_HIP0(hipGetLastError());
  BG_HIP0(hipEventSynchronize(b));
  float ms = 0;
  BG_HIP0(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  *gbps_out = ms > 0 ? (double)(n16 * 16) * iters / (ms * 1e-3) / 1e9 : 0.0; // bytes written
  return 0;
}

// kernel-only time of a batch op (the A/B of the two lane mappings): events on the caller's stream around the launch
struct BgOpTimer {
  hipEvent_t a = nullptr, b = nullptr;
  bool on = false;
  int begin(float* out, hipStream_t s) { on = out != nullptr; if (!on) return 0; BG_HIP0(hipEventCreate(&a)); BG_HIP0(hipEventCreate(&b)); BG_HIP0(hipEventRecord(a, s)); return 0; }
  void mark(hipStream_t s) { if (on) (void)hipEventRecord(b, s); }
  int end(float* out) { if (!on) return 0; BG_HIP0(hipEventSynchronize(b)); BG_HIP0(hipEventElapsedTime(out, a, b)); (void)hipEventDestroy(a); (void)hipEventDestroy(b); return 0; }
};

int bg_classify_batch_ex(const uint8_t* cards_dev, const uint8_t* n_dev, uint8_t* hand_type_dev, int64_t m, int lanes_per_case,
                         float* kernel_ms_out, void* stream) {
  if (!cards_dev || !n_dev || !hand_type_dev || m < 0 || ((uintptr_t)cards_dev & 7) || (lanes_per_case != 1 && lanes_per_case != 8)) {
    g_create_err = "bg_classify_batch: bad arguments (cards_dev must be 8-byte aligned, lanes_per_case 1 or 8)"; return BG_E_ARG;
  }
  if (kernel_ms_out) *kernel_ms_out = 0.f;
  if (m == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  BgOpTimer tm;
  int rc = tm.begin(kernel_ms_out, s);
  if (rc) return rc;
  if (lanes_per_case == 1)
    hipLaunchKernelGGL(bg_classify_batch_kernel, dim3((unsigned)((m + BG_BLOCK - 1) / BG_BLOCK)), dim3(BG_BLOCK), 0, s, cards_dev, n_dev, hand_type_dev, (long long)m);
  else
    hipLaunchKernelGGL(bg_classify_batch_l8_kernel, dim3((unsigned)((m * 8 + BG_BLOCK - 1) / BG_BLOCK)), dim3(BG_BLOCK), 0, s, cards_dev, n_dev, hand_type_dev, (long long)m);
  tm.mark(s);
  BG_HIP0(hipGetLastError());
  return tm.end(kernel_ms_out);
}
int bg_classify_batch(const uint8_t* cards_dev, const uint8_t* n_dev, uint8_t* hand_type_dev, int64_t m, void* stream) {
  return bg_classify_batch_ex(cards_dev, n_dev, hand_type_dev, m, 1, nullptr, stream);
}

static int bg_batch_dev(BgDev& d, int m, uint32_t** scratch_out);

int bg_score_hand_batch_ex(const int32_t* cases_dev, int64_t* out_dev, int m, int lanes_per_case, float* kernel_ms_out, void* stream) {
  if (!cases_dev || !out_dev || m < 0 || (lanes_per_case != 1 && lanes_per_case != 8)) { g_create_err = "bg_score_hand_batch: bad arguments (lanes_per_case 1 or 8)"; return BG_E_ARG; }
  if (kernel_ms_out) *kernel_ms_out = 0.f;
  if (m == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  BgDev d;
  uint32_t* scratch = nullptr; // two MT19937 blocks per case + the error word
  int rc = bg_batch_dev(d, m, &scratch);
  if (rc) return rc;
  BgOpTimer tm;
  hipError_t e = hipMemsetAsync(d.err, 0, 4 * sizeof(uint32_t), s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(bg_score_seed_kernel, dim3((m + BG_BLOCK - 1) / BG_BLOCK), dim3(BG_BLOCK), 0, s, d, cases_dev); // random.seed(gseed) per case
    rc = tm.begin(kernel_ms_out, s);
    if (lanes_per_case == 1)
      hipLaunchKernelGGL(bg_score_hand_batch_kernel, dim3((m + BG_BLOCK - 1) / BG_BLOCK), dim3(BG_BLOCK), 0, s, d, cases_dev, out_dev);
    else
      hipLaunchKernelGGL(bg_score_hand_batch_l8_kernel, dim3((m + BG_BLOCK / 8 - 1) / (BG_BLOCK / 8)), dim3(BG_BLOCK), 0, s, d, cases_dev, out_dev);
    tm.mark(s);
    e = hipGetLastError();
  }
  uint32_t errw = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&errw, d.err, sizeof(errw), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e == hipSuccess && rc == 0) rc = tm.end(kernel_ms_out);
  (void)hipFree(scratch);
  if (e != hipSuccess) { g_create_err = std::string("bg_score_hand_batch: ") + hipGetErrorString(e); return BG_E_HIP; }
  if (rc) return rc;
  if (errw) { g_create_err = "bg_score_hand_batch: a case drew more than two blocks of the global stream"; return BG_E_INTERNAL; }
  return 0;
}
int bg_score_hand_batch(const int32_t* cases_dev, int64_t* out_dev, int m, void* stream) {
  return bg_score_hand_batch_ex(cases_dev, out_dev, m, 1, nullptr, stream);
}

// scratch of a batch op that draws from a per-case global stream: two MT19937 blocks per case + the device error word
static int bg_batch_dev(BgDev& d, int m, uint32_t** scratch_out) {
  memset(&d, 0, sizeof(d));
  d.N = m; d.flags = BG_FLAG_SCORER_JOKERS; d.KG = 2; d.KS = 2; d.KD = 1;
  uint32_t* scratch = nullptr;
  BG_HIP0(hipMalloc((void**)&scratch, ((size_t)m * 2 * BG_MTS + 4) * sizeof(uint32_t)));
  d.gblk = scratch; d.err = scratch + (size_t)m * 2 * BG_MTS;
  *scratch_out = scratch;
  return 0;
}

int bg_sim_evaluate_batch(const int32_t* hands_dev, const int32_t* n_dev, const int32_t* flags_dev, int8_t* out_dev, int m, void* stream) {
  if (!hands_dev || !n_dev || !flags_dev || !out_dev || m < 0) { g_create_err = "bg_sim_evaluate_batch: bad arguments"; return BG_E_ARG; }
  if (m == 0) return 0;
  hipLaunchKernelGGL(bg_sim_evaluate_batch_kernel, dim3((m + BG_BLOCK - 1) / BG_BLOCK), dim3(BG_BLOCK), 0, (hipStream_t)stream, hands_dev, n_dev, flags_dev, out_dev, m);
  BG_HIP0(hipGetLastError());
  return 0;
}

int bg_sim_score_batch(const int32_t* cases_dev, int64_t* out_dev, int m, void* stream) {
  if (!cases_dev || !out_dev || m < 0) { g_create_err = "bg_sim_score_batch: bad arguments"; return BG_E_ARG; }
  if (m == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  BgDev d;
  uint32_t* scratch = nullptr;
  int rc = bg_batch_dev(d, m, &scratch);
  if (rc) return rc;
  hipError_t e = hipMemsetAsync(d.err, 0, 4 * sizeof(uint32_t), s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(bg_sim_score_batch_kernel, dim3((m + BG_BLOCK - 1) / BG_BLOCK), dim3(BG_BLOCK), 0, s, d, cases_dev, out_dev);
    e = hipGetLastError();
  }
  uint32_t errw = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&errw, d.err, sizeof(errw), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(scratch);
  if (e != hipSuccess) { g_create_err = std::string("bg_sim_score_batch: ") + hipGetErrorString(e); return BG_E_HIP; }
  if (errw) { g_create_err = "bg_sim_score_batch: a case drew more than two blocks of the global stream"; return BG_E_INTERNAL; }
  return 0;
}

} // extern "C"
