// bg_device.h -- HBM data layout + per-env device functions of the MI355X Balatro step path (gfx950 only).
//
// Execution model: ONE LANE = ONE ENV (a wave64 steps 64 independent games in lockstep).  Every per-env array is
// structure-of-arrays with the env index fastest, so a wave's access to any field is one coalesced request:
//   hot   uint4[BG_NHOT ][N]   packed game state, 16 B "chunks" (global_load_dwordx4 per lane = 1 KiB per wave)
//   deck  uint4[4       ][N]   52 card codes ((rank-2)*4+suit, cards.py:103-104), 12 B pad
//   cold  uint4[BG_NCOLD][N]   hand_play_counts + shop inventory (touched only by the lanes that need them)
//   tmpl  uint4[2       ][N]   reset template (harness injection: jokers / money / ante / hand levels)
//   ndeck uint4[KD][4][N]      ring of pre-shuffled decks (DeterministicRNG 'deck_shuffle' look-ahead)
// MT19937 state is the exception: a stream is produced AND consumed by one lane walking consecutive words, so each env
// owns contiguous 2560-byte blocks (array-of-structures; sparse lanes then write whole lines instead of 4 bytes/line):
//   gblk  u32[N][KG][640]      per-env "global random" stream: ring of consecutive raw MT19937 blocks (624 words used)
//   sblk  u32[N][KS][640]      ring of pre-seeded shop streams (first block of random.Random(shop_seed))
//   deckmt / shopgenmt u32[N][640]   authoritative MT state of streams 0 and 2 (word 624 = index)
// The serial MT19937 work (seeding = 1247 dependent steps, block twist, 51-swap shuffle) is never on the step path:
// it runs in the refill kernel, which looks AHEAD on streams whose consumption order does not depend on play
// (stream 0 is only consumed by reset shuffles, stream 2 only by one get_int per shop visit).
//
// Reference semantics (file:line relative to the reference's balatro_gym/) are cited per function.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bg_tables.h"

#define BG_NHOT 8
#define BG_NDECK 4
#define BG_NCOLD 7
#define BG_NTMPL 2
#define BG_MT_N 624
#define BG_MT_M 397
#define BG_MTS 640 // words per stored MT block (624 + index word, padded to 20 x 128 B)
#define BG_BLOCK 64 // threads per block = one wave64

// device error word bits (sticky; checked by the host, which then fails loudly)
#define BG_DEVERR_GSTREAM 1u  // global-stream ring underflow
#define BG_DEVERR_SHOPBLK 2u  // a shop visit consumed more than one MT block
#define BG_DEVERR_DECKRING 4u // pre-shuffled deck ring empty at reset
#define BG_DEVERR_SHOPRING 8u // pre-seeded shop ring empty at shop generation

// Agent-scope (sc1) accesses: they bypass the CU's vector L1, which another CU's stores never refresh (MI355X_MICROARCH.md, "inter-workgroup
// visibility").  The two-kernel engine (bg_engine2.h) serves an env's steps on whichever CU of its XCD has a free service wave, so every
// load of MUTABLE per-env state on the service path goes through these; stores are plain (they reach the XCD's L2, the one point all
// accesses to that env share during a launch).
typedef __attribute__((address_space(1))) unsigned long long g_u64;
typedef __attribute__((address_space(1))) uint32_t g_u32;
__device__ __forceinline__ unsigned long long bg_ld8a(const void* p) { return __hip_atomic_load((g_u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t bg_ld4a(const void* p) { return __hip_atomic_load((g_u32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void bg_st8a(void* p, unsigned long long v) { __hip_atomic_store((g_u64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint4 bg_ld16a(const void* p) { // two 8-byte agent-scope loads (global_load_dwordx2 sc1)
  const unsigned long long a = bg_ld8a(p), b = bg_ld8a((const char*)p + 8);
  return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}

struct BgDev {
  int N;
  uint32_t flags;
  int max_ante;
  int KG, KS, KD;
  uint4* hot;
  uint4* deck;
  uint4* cold;
  uint4* tmpl;
  uint4* ndeck;
  uint32_t* gblk;
  uint32_t* sblk;     // [N][KS][BG_SLOT_WORDS] compact seeded shop streams (BG_SW_*)
  uint32_t* sovf;     // [N][640] full seeded state of a shop stream that was read beyond its slot (rare)
  uint32_t* deckmt;
  uint32_t* shopgenmt;
  uint32_t* err;
  uint32_t* wl_count; // [4] refill work-list lengths: decks, seed rings, global blocks, shop items
  uint32_t* wl;       // [3][N] env indexes needing a refill of each kind (0 decks, 1 shop-seed rings, 2 global blocks)
  uint32_t* wl_shop;  // [N*(KS-1)][2] shop work items: env | slot << 24, shop seed
  uint32_t* sseed;    // [N][32] pre-drawn shop seeds (stream 2 is consumed by nothing else, balatro_env_2.py:1389)
  uint32_t* smeta;    // [N] seed ring: head | count << 8
  // Ring PRODUCER counters (decks | shop streams << 8 | global blocks << 16, each mod 256) live apart from the consumer
  // counters (hot chunks) and are double-buffered, so a refill can run on a side stream while the next rollout runs:
  // the rollout reads `prod_view` (written by a refill that has completed), a refill reads prod_in and writes prod_out.
  const uint32_t* prod_view;
  const uint32_t* prod_in;
  uint32_t* prod_out;
  unsigned long long* dbg; // [16] phase cycle counters (development builds, -DBG_TIMING)
  // card states (BG_FLAG_CARD_STATES; null otherwise): per deck INDEX enhancement | edition << 4 | seal << 8 as u16,
  // eight cards per 16-byte chunk, plus the copy every reset re-applies and the 'card_enhancement' stream (index 11)
  uint4* cstate;     // uint4[7][N]
  uint4* ctmpl;      // uint4[7][N]
  uint32_t* cardmt;  // u32[N][640], lazy MT19937 (cursor in word 624)
  uint32_t* sealmt;  // u32[N][640], lazy MT19937 of stream 13 'seal_applications' (purple seals)
  const uint32_t* jtab; // the JTables of bg_step.h, built once per handle (bg_tables_build_kernel): the engine kernel copies them to LDS
};

// ---------------------------------------------------------------------------------------------------------
// Unpacked per-env state (lives in registers for the duration of a step)
// ---------------------------------------------------------------------------------------------------------
struct Env {
  // chunk 0
  int64_t chips_scored, round_chips;
  // chunk 1
  int64_t best_hand;
  int32_t chips_needed, money;
  // chunk 2
  int32_t hp_total, hp_ante, jokers_sold, shop_reroll_state, shop_reroll_base;
  // chunk 3 (16 x u8)
  int32_t ante, round, phase, hands_left, discards_left, hand_size, nhand, nsel, njokers, ncons, n_magic, n_minim,
      boss_type, boss_req, bflags, shop_n;
  // chunk 4
  uint64_t hand, sel; // 8 deck indexes / 8 selected positions (selection order), one byte each
  // chunk 5
  uint32_t highlighted, face_down, boss_hp, boss_types, cons0, cons1;
  uint64_t jokers; // up to 5 ids, one byte each
  int32_t shop_ante, d_head, d_cons, d_ready; // d_ready (like s_ready, g_valid) is derived: producer - consumer counter
  // chunk 6
  uint64_t boss_cards; // played-deck-index mask (stands in for id(card), boss_blinds.py:472)
  uint64_t levels;     // 12 x 4-bit engine hand levels (scoring_engine.py:66)
  int32_t g_cur, g_cons, g_valid;
  // chunk 7
  int32_t g_idx, s_idx, s_cur, s_cons, s_ready;
  int32_t ndrop, nfo; // cards.Card objects Immolate removed from the deck (52 - ndrop are left), copies Cryptid appended behind them
  uint64_t excess; // 12 x 4-bit (state.hand_levels - engine level): planets used at the level-15 cap
  int32_t max_ante; // CurriculumBalatroEnv.current_max_ante of this env (train_balatro_agent.py:126-152), 0 = no cap; survives reset()
};

#define BG_BF_FIRST_HAND 1
#define BG_BF_SHOP_EXISTS 2

__device__ __forceinline__ uint32_t bg_b(uint32_t w, int i) { return (w >> (8 * i)) & 0xffu; }

__device__ __forceinline__ void bg_unpack(const uint4 c[BG_NHOT], Env& e) {
  e.chips_scored = (int64_t)(((uint64_t)c[0].y << 32) | c[0].x);
  e.round_chips = (int64_t)(((uint64_t)c[0].w << 32) | c[0].z);
  e.best_hand = (int64_t)(((uint64_t)c[1].y << 32) | c[1].x);
  e.chips_needed = (int32_t)c[1].z;
  e.money = (int32_t)c[1].w;
  e.hp_total = (int32_t)c[2].x;
  e.hp_ante = (int32_t)(c[2].y & 0xffffu);
  e.jokers_sold = (int32_t)(c[2].y >> 16);
  e.shop_reroll_state = (int32_t)c[2].z;
  e.shop_reroll_base = (int32_t)c[2].w;
  e.ante = bg_b(c[3].x, 0); e.round = bg_b(c[3].x, 1); e.phase = bg_b(c[3].x, 2); e.hands_left = bg_b(c[3].x, 3);
  e.discards_left = bg_b(c[3].y, 0); e.hand_size = (int)(int8_t)bg_b(c[3].y, 1); /* signed: Manacle / Wraith / Ectoplasm only ever lower it */ e.nhand = bg_b(c[3].y, 2); e.nsel = bg_b(c[3].y, 3);
  e.njokers = bg_b(c[3].z, 0); e.ncons = bg_b(c[3].z, 1); e.n_magic = bg_b(c[3].z, 2); e.n_minim = bg_b(c[3].z, 3);
  e.boss_type = bg_b(c[3].w, 0); e.boss_req = bg_b(c[3].w, 1); e.bflags = bg_b(c[3].w, 2); e.shop_n = bg_b(c[3].w, 3);
  e.hand = ((uint64_t)c[4].y << 32) | c[4].x;
  e.sel = ((uint64_t)c[4].w << 32) | c[4].z;
  e.highlighted = c[5].x & 0xffffu; e.face_down = bg_b(c[5].x, 2); e.boss_hp = bg_b(c[5].x, 3);
  e.boss_types = c[5].y & 0xffffu; e.cons0 = bg_b(c[5].y, 2); e.cons1 = bg_b(c[5].y, 3);
  e.jokers = (uint64_t)c[5].z | ((uint64_t)(c[5].w & 0xffu) << 32);
  e.shop_ante = bg_b(c[5].w, 1); e.d_head = bg_b(c[5].w, 2); e.d_cons = bg_b(c[5].w, 3); e.d_ready = 0;
  e.boss_cards = ((uint64_t)c[6].y << 32) | c[6].x;
  e.levels = (uint64_t)c[6].z | ((uint64_t)(c[6].w & 0xffffu) << 32);
  e.g_cur = bg_b(c[6].w, 2); e.g_cons = bg_b(c[6].w, 3); e.g_valid = 0;
  e.g_idx = (int32_t)(c[7].x & 0xffffu); e.s_idx = (int32_t)(c[7].x >> 16);
  e.s_cur = bg_b(c[7].y, 0); e.s_cons = bg_b(c[7].y, 1); e.s_ready = 0; e.ndrop = bg_b(c[7].y, 2); e.nfo = bg_b(c[7].y, 3);
  e.excess = (uint64_t)c[7].z | ((uint64_t)(c[7].w & 0xffffu) << 32);
  e.max_ante = (int32_t)(c[7].w >> 16) & 0xff;
}

__device__ __forceinline__ uint32_t bg_p4(int a, int b, int c, int d) {
  return (uint32_t)(a & 0xff) | ((uint32_t)(b & 0xff) << 8) | ((uint32_t)(c & 0xff) << 16) | ((uint32_t)(d & 0xff) << 24);
}

__device__ __forceinline__ void bg_pack(const Env& e, uint4 c[BG_NHOT]) {
  c[0] = make_uint4((uint32_t)e.chips_scored, (uint32_t)((uint64_t)e.chips_scored >> 32), (uint32_t)e.round_chips,
                    (uint32_t)((uint64_t)e.round_chips >> 32));
  c[1] = make_uint4((uint32_t)e.best_hand, (uint32_t)((uint64_t)e.best_hand >> 32), (uint32_t)e.chips_needed, (uint32_t)e.money);
  c[2] = make_uint4((uint32_t)e.hp_total, ((uint32_t)e.hp_ante & 0xffffu) | ((uint32_t)e.jokers_sold << 16),
                    (uint32_t)e.shop_reroll_state, (uint32_t)e.shop_reroll_base);
  c[3] = make_uint4(bg_p4(e.ante, e.round, e.phase, e.hands_left), bg_p4(e.discards_left, e.hand_size, e.nhand, e.nsel),
                    bg_p4(e.njokers, e.ncons, e.n_magic, e.n_minim), bg_p4(e.boss_type, e.boss_req, e.bflags, e.shop_n));
  c[4] = make_uint4((uint32_t)e.hand, (uint32_t)(e.hand >> 32), (uint32_t)e.sel, (uint32_t)(e.sel >> 32));
  c[5] = make_uint4((e.highlighted & 0xffffu) | ((e.face_down & 0xffu) << 16) | ((e.boss_hp & 0xffu) << 24),
                    (e.boss_types & 0xffffu) | ((e.cons0 & 0xffu) << 16) | ((e.cons1 & 0xffu) << 24), (uint32_t)e.jokers,
                    bg_p4((int)(e.jokers >> 32), e.shop_ante, e.d_head, e.d_cons));
  c[6] = make_uint4((uint32_t)e.boss_cards, (uint32_t)(e.boss_cards >> 32), (uint32_t)e.levels,
                    ((uint32_t)(e.levels >> 32) & 0xffffu) | ((uint32_t)(e.g_cur & 0xff) << 16) | ((uint32_t)(e.g_cons & 0xff) << 24));
  c[7] = make_uint4(((uint32_t)e.g_idx & 0xffffu) | ((uint32_t)e.s_idx << 16), bg_p4(e.s_cur, e.s_cons, e.ndrop, e.nfo),
                    (uint32_t)e.excess, ((uint32_t)(e.excess >> 32) & 0xffffu) | ((uint32_t)(e.max_ante & 0xff) << 16));
}

// ring fill levels as this kernel may see them: producer counters of a COMPLETED refill minus own consumer counters
__device__ __forceinline__ void bg_derive_ready(Env& e, uint32_t prod) {
  e.d_ready = (int)((prod - (uint32_t)e.d_cons) & 0xffu);
  e.s_ready = (int)(((prod >> 8) - (uint32_t)e.s_cons) & 0xffu);
  e.g_valid = (int)(((prod >> 16) - (uint32_t)e.g_cons) & 0xffu);
}
__device__ __forceinline__ void bg_load_env(const BgDev& d, int env, Env& e) {
  uint4 c[BG_NHOT];
#pragma unroll
  for (int k = 0; k < BG_NHOT; k++) c[k] = d.hot[(size_t)k * d.N + env];
  bg_unpack(c, e);
  bg_derive_ready(e, d.prod_view ? d.prod_view[env] : 0u);
}
__device__ __forceinline__ void bg_store_env(const BgDev& d, int env, const Env& e) {
  uint4 c[BG_NHOT];
  bg_pack(e, c);
#pragma unroll
  for (int k = 0; k < BG_NHOT; k++) d.hot[(size_t)k * d.N + env] = c[k];
}

// byte i of a packed 8-byte list
__device__ __forceinline__ int bg_get8(uint64_t v, int i) { return (int)((v >> (8 * i)) & 0xffull); }
__device__ __forceinline__ uint64_t bg_set8(uint64_t v, int i, int x) {
  return (v & ~(0xffull << (8 * i))) | ((uint64_t)(x & 0xff) << (8 * i));
}
// remove byte i, shifting the tail down
__device__ __forceinline__ uint64_t bg_del8(uint64_t v, int i) {
  uint64_t lowmask = i ? ((1ull << (8 * i)) - 1) : 0ull;
  uint64_t lo = v & lowmask;
  uint64_t hi = (i < 7) ? (v >> (8 * (i + 1))) << (8 * i) : 0ull;
  return lo | hi;
}
__device__ __forceinline__ int bg_level(const Env& e, int ht) { return (int)((e.levels >> (4 * ht)) & 0xf); }

// ---------------------------------------------------------------------------------------------------------
// Card lookups.  deck chunk 0 (deck indexes 0..15) is held in registers; in the live reference the hand is always a
// subset of deck[0..hand_size), so the global-memory path below is the rare general case.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t* bg_gblock(const BgDev& d, int env, int slot) { return d.gblk + ((size_t)env * d.KG + slot) * BG_MTS; }
// A shop-stream ring slot holds the first BG_SW_T OUTPUT words of random.Random(shop_seed) -- regenerated and tempered by the refill
// kernel that seeds the stream (word k of the first output block is S[k+397] ^ twist(S[k], S[k+1]) for k < 227: the seeding pass has all
// three in hand) --, the TOP BYTES of output words 0..23 packed into six words, and the seed: 256 bytes = two lines.  A FRESH inventory (every
// generate_shop) looks at nothing but those top bytes (getrandbits(2 / 6 / 8) of consecutive words: bg_shop_inventory), so it reads the slot's
// last two 16-byte pieces -- ONE line (round 6: it was six pieces of the first 24 words, two lines) --; a rerolled inventory reads ~11 full words from where the last one stopped, so 56 words are a
// visit with three rerolls; a visit that reads further re-seeds the FULL state into the env's overflow block once (bg_shop_overflow) and
// carries on there.  Rounds 1-2 kept the seeded state itself (two 68-word windows, 576 bytes) and the consumer regenerated what it read: 14
// scattered 16-byte loads, 24 twists and 31 temperings on the winning play's critical path (~13 k cycles of a play batch).
#define BG_SW_T 56                 // output words per slot
#define BG_SW_PK 56                // slot words 56..61: top byte of output word k in byte k & 3 of word 56 + (k >> 2), k < 24
#define BG_SW_SEED 62              // slot word holding the seed
#define BG_SLOT_WORDS 64           // 56 words + 6 packed + the seed + one of padding = 256 bytes
#define BG_S_FASTMAX (BG_SW_T - 1) // largest k the slot holds
static_assert(BG_SW_T >= 28 && BG_SW_PK == BG_SW_T && BG_SW_PK % 4 == 0 && BG_SW_PK + 6 == BG_SW_SEED && BG_SW_SEED + 2 == BG_SLOT_WORDS, "shop slot");
#define BG_BF_SHOP_OVF 4           // bflags: the current shop stream lives in the overflow block (full state)
__device__ __forceinline__ uint32_t* bg_sblock(const BgDev& d, int env, int slot) { return d.sblk + ((size_t)env * d.KS + slot) * BG_SLOT_WORDS; }
__device__ __forceinline__ uint32_t* bg_sovf(const BgDev& d, int env) { return d.sovf + (size_t)env * BG_MTS; }
// random.Random(key) (init_by_array([key])) as two small rolled loops through memory: ~1 ms, for the overflow path only (a
// call to the unrolled register version would cost every kernel that contains it its stack frame and registers)
__device__ __forceinline__ void bg_mt_seed_rolled(uint32_t* p, uint32_t key) {
  uint32_t g = 19650218u, a = g; // init_genrand(19650218) on the fly; a = mt[i-1] of pass 1
#pragma unroll 1
  for (uint32_t i = 1; i < 624u; i++) {
    g = 1812433253u * (g ^ (g >> 30)) + i;
    a = (g ^ ((a ^ (a >> 30)) * 1664525u)) + key;
    p[i] = a;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const uint32_t m1 = (p[1] ^ ((a ^ (a >> 30)) * 1664525u)) + key; // 624th step of pass 1: mt[0] = mt[623], i = 1
  uint32_t b = m1;
#pragma unroll 1
  for (uint32_t i = 2; i < 624u; i++) { // pass 2
    b = (p[i] ^ ((b ^ (b >> 30)) * 1566083941u)) - i;
    p[i] = b;
  }
  p[1] = (m1 ^ ((b ^ (b >> 30)) * 1566083941u)) - 1u; // wrapped last step: mt[0] = mt[623], i = 1
  p[0] = 0x80000000u;
}
__device__ __forceinline__ uint32_t* bg_deckmt(const BgDev& d, int env) { return d.deckmt + (size_t)env * BG_MTS; }
__device__ __forceinline__ uint32_t* bg_shopgenmt(const BgDev& d, int env) { return d.shopgenmt + (size_t)env * BG_MTS; }

// development cycle probes (-DBG_TIMING): the first active lane of the wave adds the cycles since the previous probe to
// a per-workgroup LDS accumulator; kernels flush it to d.dbg once at their end (a global atomic per probe would serialise
// ~1000 waves on one L2 word and distort what is being measured)
#ifdef BG_TIMING
__shared__ unsigned long long bg_probe_lds[32];
#define BG_PROBE_BEGIN() unsigned long long _pt = __builtin_readcyclecounter()
#define BG_PROBE(k) do { unsigned long long _n = __builtin_readcyclecounter(); \
    if ((int)threadIdx.x == __ffsll((long long)__ballot(1)) - 1 + (int)(threadIdx.x & ~63u)) atomicAdd(&bg_probe_lds[k], _n - _pt); \
    _pt = __builtin_readcyclecounter(); } while (0)
#define BG_PROBE_INIT() do { if (threadIdx.x < 32) bg_probe_lds[threadIdx.x] = 0; __syncthreads(); } while (0)
#define BG_PROBE_FLUSH(d) do { __syncthreads(); if (threadIdx.x < 32 && (d).dbg && bg_probe_lds[threadIdx.x]) atomicAdd(&(d).dbg[threadIdx.x], bg_probe_lds[threadIdx.x]); } while (0)
#else
#define BG_PROBE_BEGIN() do {} while (0)
#define BG_PROBE(k) do {} while (0)
#define BG_PROBE_INIT() do {} while (0)
#define BG_PROBE_FLUSH(d) do {} while (0)
#endif

// The 52 card codes of an env's deck, as seen by the step code.  lane-per-env kernels keep the first 16 in registers
// and read the rest from HBM; the block-compacted rollout kernel keeps every deck of its workgroup in LDS
// ([dword 0..15][lane of the workgroup], bank = lane) so that no card lookup of a step -- phase B's gather, the boss
// checks, the observation's hand -- is an HBM round trip.
#define BG_RB 128 // envs per workgroup of bg_rollout2_kernel: two workgroups per CU, whose phases overlap each other
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
struct Deck0 { uint64_t lo, hi; static constexpr bool kCards = false; };   // first 16 cards in registers, the rest read from HBM
struct DeckLds { lds_u32* col; static constexpr bool kCards = false; };     // &s_deck[0][lane of the workgroup]
// the same decks in kernels built for card states (cards.py CardState: enhancement / edition / seal per deck index)
struct Deck0C : Deck0 { static constexpr bool kCards = true; };
struct DeckLdsC : DeckLds { static constexpr bool kCards = true; };
#define BG_NCST 7 // 16-byte chunks of card state per env
__device__ __forceinline__ uint32_t bg_cstate(const BgDev& d, int env, int ci) { // enh | edition << 4 | seal << 8
  return (uint32_t)((const uint16_t*)&d.cstate[(size_t)(ci >> 3) * d.N + env])[ci & 7];
}
__device__ __forceinline__ Deck0 bg_load_deck0(const BgDev& d, int env) {
  uint4 c = d.deck[env];
  Deck0 r;
  r.lo = ((uint64_t)c.y << 32) | c.x;
  r.hi = ((uint64_t)c.w << 32) | c.z;
  return r;
}
// chunk k (16 cards) of a freshly consumed deck
__device__ __forceinline__ void bg_deck_set(Deck0& dk, int k, uint4 c) {
  if (k == 0) { dk.lo = ((uint64_t)c.y << 32) | c.x; dk.hi = ((uint64_t)c.w << 32) | c.z; }
}
__device__ __forceinline__ void bg_deck_set(DeckLds& dk, int k, uint4 c) {
  dk.col[(4 * k) * BG_RB] = c.x; dk.col[(4 * k + 1) * BG_RB] = c.y; dk.col[(4 * k + 2) * BG_RB] = c.z; dk.col[(4 * k + 3) * BG_RB] = c.w;
}
__device__ __forceinline__ int bg_card(const BgDev& d, int env, const Deck0& k, int idx) {
  if (idx < 16) return (int)(((idx < 8 ? k.lo : k.hi) >> (8 * (idx & 7))) & 0xffull);
  const uint32_t* p = (const uint32_t*)&d.deck[(size_t)(idx >> 4) * d.N + env];   // (rare: past the L1, see bg_ld4a)
  return (int)((bg_ld4a(p + ((idx & 15) >> 2)) >> (8 * (idx & 3))) & 0xffu);
}
// workgroup decks with the row stride as a parameter (bg_engine_kernel: 256 envs per workgroup)
template <int S, bool C> struct DeckLdsS { lds_u32* col; static constexpr bool kCards = C; };
template <int S, bool C>
__device__ __forceinline__ void bg_deck_set(DeckLdsS<S, C>& dk, int k, uint4 c) {
  dk.col[(4 * k) * S] = c.x; dk.col[(4 * k + 1) * S] = c.y; dk.col[(4 * k + 2) * S] = c.z; dk.col[(4 * k + 3) * S] = c.w;
}
template <int S, bool C>
__device__ __forceinline__ int bg_card(const BgDev& d, int env, const DeckLdsS<S, C>& k, int idx) {
  return (int)((const lds_u8*)k.col)[(idx >> 2) * (S * 4) + (idx & 3)];
}
__device__ __forceinline__ int bg_card(const BgDev& d, int env, const DeckLds& k, int idx) {
  return (int)((const lds_u8*)k.col)[(idx >> 2) * (BG_RB * 4) + (idx & 3)];
}

// The first 16 cards of the deck in two registers: every deck index a hand normally holds (the hand is a
// permutation of deck[0..hand_size)).  Four independent LDS reads up front instead of one dependent read per card looked at.
struct DeckHead { uint64_t lo, hi; };
__device__ __forceinline__ DeckHead bg_deck_head(const BgDev& d, int env, const Deck0& k) { return DeckHead{k.lo, k.hi}; }
template <int S, bool C>
__device__ __forceinline__ DeckHead bg_deck_head(const BgDev& d, int env, const DeckLdsS<S, C>& k) {
  const uint32_t a = k.col[0], b = k.col[S], c = k.col[2 * S], e = k.col[3 * S];
  return DeckHead{((uint64_t)b << 32) | a, ((uint64_t)e << 32) | c};
}
template <class DK>
__device__ __forceinline__ int bg_card_h(const BgDev& d, int env, const DK& k, const DeckHead& h, int idx) {
  if (idx < 16) return (int)(((idx < 8 ? h.lo : h.hi) >> (8 * (idx & 7))) & 0xffull);
  return bg_card(d, env, k, idx);
}

// ---------------------------------------------------------------------------------------------------------
// MT19937 pieces (CPython Modules/_randommodule.c)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bg_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
__device__ __forceinline__ uint32_t bg_twist(uint32_t a, uint32_t b, uint32_t far) {
  uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// One word / one random() of a lazy MT19937 stream (see bg_lib.hip: word c of the next block is made when it is read).
// Serial loads and a store per word: only for rare draws (GLASS / LUCKY cards on the 'card_enhancement' stream).
__device__ __forceinline__ uint32_t bg_lazy_next(uint32_t* S) {
  uint32_t c = S[BG_MT_N] & 0x3ffu;
  const uint32_t c1 = c + 1 == BG_MT_N ? 0u : c + 1, cf = c + BG_MT_M >= BG_MT_N ? c + BG_MT_M - BG_MT_N : c + BG_MT_M;
  const uint32_t y = bg_twist(S[c], S[c1], S[cf]);
  S[c] = y;
  S[BG_MT_N] = c1 | 0x80000000u;
  return bg_temper(y);
}
__device__ __forceinline__ double bg_lazy_random(uint32_t* S) {
  const uint32_t a = bg_lazy_next(S) >> 5, b = bg_lazy_next(S) >> 6;
  return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}

// Per-lane LDS window over the next raw words of an RNG block.  The draws of a play (joker chain: ~2 words per
// (card, joker) pair) or of a shop generation are consecutive words of ONE block, so they are fetched with
// independent loads up front and then consumed from LDS: the serial chain of dependent HBM round trips (one per
// draw, ~1 us each at one wave per SIMD) becomes one batch.  Layout [word][lane] (bank = lane: conflict-free).
#define BG_WIN 16 // words per lane: The Wheel's 16 words, Immolate's 16-word deck; a rerolled shop reads its words 16 at a time (a FRESH one reads the slot's packed tail, no window)
// Per-workgroup lookup tables in LDS (filled once per launch by bg_tables_init): per-lane-different joker ids make
// `switch` statements fully divergent (a wave walks every case some lane takes) and constant-memory tables cost an HBM
// round trip per lookup at one wave per SIMD; an LDS read is ~100 cycles and never diverges.
struct JTables {
  uint64_t jd[152];    // individual-phase descriptor per joker id (bg_jdesc)
  uint32_t jm[152];    // main-phase descriptor per joker id (bg_jmain_desc)
  uint64_t jr[152];    // rank set of jd[] expanded to a 4-bit-per-rank mask (for the played-rank histogram)
  uint8_t jf[152];     // name-group flags used by reward shaping / discard hooks
  uint8_t cost[152];   // JokerInfo.base_cost (jokers.py)
  double pow115[101];  // 1.15 ** k  (shop.py:105)
  double pow15[16];    // 1.5 ** k   (Baron)
  double pow08[16];    // 0.8 ** k   (boss_blinds.py:436)
  uint32_t inv[64];    // floor(2**32 / d): h % d without a division (bg_policy_action_fast)
};
// LDS pointers keep their address space in the type: a generic pointer stored in a struct compiles to FLAT loads (the
// vector-memory path, ~1-2k cycles when nothing hides it) instead of ds_read (~100 cycles).
typedef __attribute__((address_space(3))) const JTables lds_JTables;
struct RngWin {
  lds_JTables* jt;
  lds_u32* lds;   // &win[0][lane]
  int g_blk, g_start, g_len; // window over the global stream: block, first index, words
  int s_start, s_len;        // window over the current shop stream
  bool defer_adv;            // a winning play leaves _advance_round (and the shop it generates) to a second work item
  bool need_inv;             // a shop inventory is due: generated ONCE at the end of the dispatch, whichever action asked for it
};
// (Round 6 fetched a step's late global reads -- the next shop slot's tail, the Bloodstone words, the reset template -- into this window ahead of time by
//  LDS-DMA, `global_load_lds_dword[x4]`, so that no register held them: parity-green, and inside +-1.5 % of the code without it at both launch shapes --
//  the product build already overlaps those waits; profiles/r06/lds_dma_prefetch.txt, tools/micro/ldsdma.hip.  Not kept.)
__device__ __forceinline__ void bg_win_init(RngWin& w, uint32_t* lds_lane, const JTables* jt = nullptr) {
  w.jt = (lds_JTables*)jt; w.lds = (lds_u32*)lds_lane; w.g_blk = -1; w.g_start = 0; w.g_len = 0; w.s_start = 0; w.s_len = 0; w.defer_adv = false; w.need_inv = false;
}
__device__ __noinline__ void bg_win_fill(lds_u32* lds, const uint32_t* src, int len) {
#pragma unroll 1
  for (int base = 0; base < len; base += BG_WIN) { // BG_WIN independent loads in flight, then one wait
    uint32_t v[BG_WIN];
#pragma unroll
    for (int j = 0; j < BG_WIN; j++) v[j] = (base + j < len) ? src[base + j] : 0u;
#pragma unroll
    for (int j = 0; j < BG_WIN; j++) if (base + j < len) lds[(base + j) * BG_BLOCK] = v[j];
  }
}

// next raw word of the per-env "global random" stream (ring of blocks, tempered on read)
// block switch; g_idx may have been advanced past the end by bg_gskip (words consumed without being read)
__device__ __forceinline__ void bg_gnorm(const BgDev& d, Env& e) {
  if (e.g_idx >= BG_MT_N) { e.g_cur = (e.g_cur + 1 == d.KG) ? 0 : e.g_cur + 1; e.g_idx -= BG_MT_N; e.g_valid--; e.g_cons = (e.g_cons + 1) & 0xff; }
}
// consume `count` (< 624) words whose values nobody looks at (complete_joker_effects.py draws them eagerly)
__device__ __forceinline__ void bg_gskip(const BgDev& d, Env& e, int count) {
  bg_gnorm(d, e);
  e.g_idx += count;
}
__device__ __forceinline__ uint32_t bg_gdraw(const BgDev& d, int env, Env& e, RngWin& w) {
  bg_gnorm(d, e);
  if (e.g_valid <= 0) { atomicOr(d.err, BG_DEVERR_GSTREAM); e.g_valid = 0; return 0u; }
  uint32_t off = (uint32_t)(e.g_idx - w.g_start);
  uint32_t y;
  if (w.g_blk == e.g_cur && off < (uint32_t)w.g_len) y = w.lds[off * BG_BLOCK];
  else y = bg_gblock(d, env, e.g_cur)[e.g_idx];
  e.g_idx++;
  return bg_temper(y);
}
// raw word `off` (< 624) positions ahead of the next draw, without consuming anything
__device__ __forceinline__ uint32_t bg_gpeek(const BgDev& d, int env, const Env& e, int off) {
  int idx = e.g_idx + off, blk = e.g_cur, need = 1;
  if (idx >= BG_MT_N) { idx -= BG_MT_N; blk = (blk + 1 == d.KG) ? 0 : blk + 1; need = 2; }
  if (idx >= BG_MT_N) { idx -= BG_MT_N; blk = (blk + 1 == d.KG) ? 0 : blk + 1; need = 3; }
  if (e.g_valid < need) { atomicOr(d.err, BG_DEVERR_GSTREAM); return 0u; }
  return bg_temper(bg_gblock(d, env, blk)[idx]);
}
// ADDRESS of the raw word `off` positions ahead (ok = the ring holds it; otherwise a harmless address): for loads that are issued side by side and
// looked at together -- bg_gpeek's tempering, inside a divergent branch, makes every call wait for its own HBM round trip
__device__ __forceinline__ const uint32_t* bg_gpeek_addr(const BgDev& d, int env, const Env& e, int off, bool& ok) {
  int idx = e.g_idx + off, blk = e.g_cur, need = 1;
  if (idx >= BG_MT_N) { idx -= BG_MT_N; blk = (blk + 1 == d.KG) ? 0 : blk + 1; need = 2; }
  if (idx >= BG_MT_N) { idx -= BG_MT_N; blk = (blk + 1 == d.KG) ? 0 : blk + 1; need = 3; }
  ok = e.g_valid >= need;
  return bg_gblock(d, env, ok ? blk : e.g_cur) + (ok ? idx : 0);
}
// one raw word `off` positions ahead, for its LINE (nobody looks at the value; 0 without a load when the ring does not hold it yet)
__device__ __forceinline__ uint32_t bg_gtouch(const BgDev& d, int env, const Env& e, int off) {
  int idx = e.g_idx + off, blk = e.g_cur, need = 1;
  if (idx >= BG_MT_N) { idx -= BG_MT_N; blk = (blk + 1 == d.KG) ? 0 : blk + 1; need = 2; }
  if (idx >= BG_MT_N) { idx -= BG_MT_N; blk = (blk + 1 == d.KG) ? 0 : blk + 1; need = 3; }
  return e.g_valid >= need ? bg_gblock(d, env, blk)[idx] : 0u;
}
// the 12 tempered words `skip` positions ahead of the cursor, in three (unaligned) 16-byte loads: the spare words behind a
// ring block mirror the head of the next one (bg_refill_gblk_kernel).  avail = which of them the ring already holds.
struct __attribute__((packed, aligned(4))) BgU4 { uint32_t x, y, z, w; };
__device__ __forceinline__ void bg_gpeek12(const BgDev& d, int env, const Env& e, int skip, uint32_t (&out)[12], uint32_t& avail) {
  int idx = e.g_idx + skip, blk = e.g_cur, need = 1;
  if (idx >= BG_MT_N) { idx -= BG_MT_N; blk = (blk + 1 == d.KG) ? 0 : blk + 1; need = 2; }
  if (idx >= BG_MT_N) { idx -= BG_MT_N; blk = (blk + 1 == d.KG) ? 0 : blk + 1; need = 3; }
  avail = 0;
  if (e.g_valid >= need) {
    const int in_blk = BG_MT_N - idx; // words before the mirror
    avail = (e.g_valid > need || in_blk >= 12) ? 0xfffu : ((1u << in_blk) - 1u);
    const BgU4* p = (const BgU4*)(bg_gblock(d, env, blk) + idx);
#pragma unroll
    for (int g = 0; g < 3; g++) {
      BgU4 v = p[g];
      out[4 * g] = bg_temper(v.x); out[4 * g + 1] = bg_temper(v.y); out[4 * g + 2] = bg_temper(v.z); out[4 * g + 3] = bg_temper(v.w);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 12; i++) out[i] = 0u;
  }
}
// the same 12 words UNTEMPERED: nothing here waits for the loads, so they can be issued long before the values are needed
__device__ __forceinline__ void bg_gpeek12_raw(const BgDev& d, int env, const Env& e, int skip, uint32_t (&out)[12], uint32_t& avail) {
  int idx = e.g_idx + skip, blk = e.g_cur, need = 1;
  if (idx >= BG_MT_N) { idx -= BG_MT_N; blk = (blk + 1 == d.KG) ? 0 : blk + 1; need = 2; }
  if (idx >= BG_MT_N) { idx -= BG_MT_N; blk = (blk + 1 == d.KG) ? 0 : blk + 1; need = 3; }
  const bool have = e.g_valid >= need;
  const int in_blk = BG_MT_N - idx; // words before the mirror
  avail = have ? ((e.g_valid > need || in_blk >= 12) ? 0xfffu : ((1u << in_blk) - 1u)) : 0u;
  const BgU4* p = (const BgU4*)(bg_gblock(d, env, have ? blk : e.g_cur) + (have ? idx : 0));
#pragma unroll
  for (int g = 0; g < 3; g++) {
    const BgU4 v = p[g];
    out[4 * g] = v.x; out[4 * g + 1] = v.y; out[4 * g + 2] = v.z; out[4 * g + 3] = v.w;
  }
}
// fetch the next `count` words of the global stream into the window (stops at the block end)
__device__ __forceinline__ void bg_gprefetch(const BgDev& d, int env, Env& e, RngWin& w, int count) {
  bg_gnorm(d, e);
  if (e.g_valid <= 0) return;
  int len = BG_MT_N - e.g_idx;
  if (len > count) len = count;
  if (len > BG_WIN) len = BG_WIN;
  bg_win_fill(w.lds, bg_gblock(d, env, e.g_cur) + e.g_idx, len);
  w.g_blk = e.g_cur; w.g_start = e.g_idx; w.g_len = len;
  w.s_len = 0; // the window storage is shared with the shop stream
}
// The current shop's random.Random(shop_seed) (shop.py:96).  The ring slot holds the SEEDED MT19937 state S (the refill
// kernel never twists it); word k of the first output block is B(k) = far ^ twist(S[k], nxt) with
//   far = k < 227 ? S[k + 397] : B(k - 227),   nxt = k < 623 ? S[k + 1] : B(0)
// i.e. at most three levels deep.  A shop visit reads ~13 words per inventory, so the window below (k + len <= 227: every
// operand is a seeded word) covers everything but many-reroll visits, which take the slow exact path.
// word k of the first output block from the FULL seeded state S (the overflow block), untempered
__device__ __noinline__ uint32_t bg_sword_slow(const uint32_t* S, int k) {
  uint32_t far;
  if (k < BG_MT_N - BG_MT_M) far = S[k + BG_MT_M];
  else {
    const int k1 = k - (BG_MT_N - BG_MT_M); // 0..396
    uint32_t far1;
    if (k1 < BG_MT_N - BG_MT_M) far1 = S[k1 + BG_MT_M];
    else { const int k2 = k1 - (BG_MT_N - BG_MT_M); far1 = bg_twist(S[k2], S[k2 + 1], S[k2 + BG_MT_M]); } // k2 < 170
    far = bg_twist(S[k1], S[k1 + 1], far1);
  }
  const uint32_t nxt = k < BG_MT_N - 1 ? S[k + 1] : bg_twist(S[0], S[1], S[BG_MT_M]);
  return bg_twist(S[k], nxt, far);
}
// the visit reads past what the slot holds: random.Random(seed) once more, whole state, into the env's overflow block
__device__ __forceinline__ void bg_shop_overflow(const BgDev& d, int env, Env& e) {
  bg_mt_seed_rolled(bg_sovf(d, env), bg_sblock(d, env, e.s_cur)[BG_SW_SEED]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // the lane reads these words back (and a later step may run in another wave)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  e.bflags |= BG_BF_SHOP_OVF;
}
__device__ __forceinline__ const uint32_t* bg_sbase(const BgDev& d, int env, const Env& e, bool& full) {
  full = (e.bflags & BG_BF_SHOP_OVF) != 0;
  // the overflow block may have been (re)written by another CU since this CU last cached its lines (two-kernel engine): rare path, one L1 invalidate
  if (full) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return full ? bg_sovf(d, env) : bg_sblock(d, env, e.s_cur);
}
// next output word of the shop stream; the window (w.lds) holds TEMPERED words
__device__ __forceinline__ uint32_t bg_sdraw(const BgDev& d, int env, Env& e, RngWin& w) {
  if (e.s_idx >= BG_MT_N) { atomicOr(d.err, BG_DEVERR_SHOPBLK); return 0u; }
  uint32_t off = (uint32_t)(e.s_idx - w.s_start);
  uint32_t y;
  if (off < (uint32_t)w.s_len) y = w.lds[off * BG_BLOCK];
  else {
    if (e.s_idx > BG_S_FASTMAX && !(e.bflags & BG_BF_SHOP_OVF)) bg_shop_overflow(d, env, e);
    bool full;
    const uint32_t* S = bg_sbase(d, env, e, full);
    y = full ? bg_temper(bg_sword_slow(S, e.s_idx)) : S[e.s_idx];
  }
  e.s_idx++;
  return y;
}
// the next `len` (<= BG_WIN) words of the shop stream into the window: finished words out of the slot; the overflow block (full seeded state)
// regenerates and tempers them
__device__ __forceinline__ void bg_swin_fill(lds_u32* lds, const uint32_t* S, bool full, int k0, int len) {
  if (!full) { // finished words
    uint32_t A[BG_WIN];
#pragma unroll
    for (int j = 0; j < BG_WIN; j++) A[j] = (j < len) ? S[k0 + j] : 0u;
#pragma unroll
    for (int j = 0; j < BG_WIN; j++) if (j < len) lds[j * BG_BLOCK] = A[j];
    return;
  }
  uint32_t A[BG_WIN + 1], F[BG_WIN];
  const uint32_t* SF = S + k0 + BG_MT_M; // &S[k0 + 397]
#pragma unroll
  for (int j = 0; j < BG_WIN + 1; j++) A[j] = (j <= len) ? S[k0 + j] : 0u;
#pragma unroll
  for (int j = 0; j < BG_WIN; j++) F[j] = (j < len) ? SF[j] : 0u;
#pragma unroll
  for (int j = 0; j < BG_WIN; j++) if (j < len) lds[j * BG_BLOCK] = bg_temper(bg_twist(A[j], A[j + 1], F[j]));
}
__device__ __forceinline__ void bg_sprefetch(const BgDev& d, int env, Env& e, RngWin& w, int count) {
  bool full;
  const uint32_t* S = bg_sbase(d, env, e, full);
  // stay where the words are at hand: k <= BG_S_FASTMAX in the slot, k < 227 (every operand a seeded word) in the full state
  int len = (full ? (BG_MT_N - BG_MT_M) : (BG_S_FASTMAX + 1)) - e.s_idx;
  if (len > count) len = count;
  if (len > BG_WIN) len = BG_WIN;
  if (len < 0) len = 0;
  if (len > 0) bg_swin_fill(w.lds, S, full, e.s_idx, len);
  w.s_start = e.s_idx; w.s_len = len;
  w.g_len = 0; w.g_blk = -1;
}
// Lib/random.py _randbelow_with_getrandbits (n >= 1): k = n.bit_length(); r = getrandbits(k) until r < n
template <bool SHOP>
__device__ __forceinline__ uint32_t bg_randbelow(const BgDev& d, int env, Env& e, RngWin& w, uint32_t n) {
  int k = 32 - __clz(n);
  uint32_t r;
  int guard = 0;
  do {
    r = (SHOP ? bg_sdraw(d, env, e, w) : bg_gdraw(d, env, e, w)) >> (32 - k);
  } while (r >= n && ++guard < 4096);
  return r < n ? r : 0u;
}
// random(): (a*67108864.0+b)*(1.0/9007199254740992.0) with a = u32>>5, b = u32>>6
__device__ __forceinline__ double bg_grandom(const BgDev& d, int env, Env& e, RngWin& w) {
  uint32_t a = bg_gdraw(d, env, e, w) >> 5;
  uint32_t b = bg_gdraw(d, env, e, w) >> 6;
  return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}

// ---------------------------------------------------------------------------------------------------------
// Pure game functions
// ---------------------------------------------------------------------------------------------------------
// cards.py:52-60 Rank.base_chips, on a card code
__device__ __forceinline__ int bg_card_chips(int code) {
  int r = (code >> 2) + 2;
  return r <= 10 ? r : (r == 14 ? 11 : 10);
}
// scoring_engine.py:27-40,87-101 get_hand_chips_mult
__device__ __forceinline__ void bg_hand_base(int ht, int level, int& chips, int& mult) {
  int bc, bm;
  switch (ht) {
    case 0: bc = 5; bm = 1; break;
    case 1: bc = 10; bm = 2; break;
    case 2: bc = 20; bm = 2; break;
    case 3: bc = 30; bm = 3; break;
    case 4: bc = 30; bm = 4; break;
    case 5: bc = 35; bm = 4; break;
    case 6: bc = 40; bm = 4; break;
    case 7: bc = 60; bm = 7; break;
    case 8: bc = 100; bm = 8; break;
    case 9: bc = 120; bm = 12; break;
    case 10: bc = 140; bm = 14; break;
    default: bc = 160; bm = 16; break;
  }
  chips = bc + (level - 1) * 10;
  mult = bm + (level - 1);
}

// balatro_game.py:40-93 _classify_hand, bit-parallel: `cards` = up to 8 card codes, n of them valid.
// rank presence mask (13 bits) for the straight test, 4-bit-per-rank histogram for the count tests.
__device__ __forceinline__ int bg_classify(uint64_t cards, int n) {
  if (n <= 0) return 0;
  uint64_t hist = 0; // 13 nibbles
  uint32_t suits = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (i < n) {
      int c = (int)((cards >> (8 * i)) & 0xff);
      hist += 1ull << (4 * (c >> 2));
      suits |= 1u << (c & 3);
    }
  }
  // counts, sorted: counts[0] == 4 / counts[0] == 3 and counts[1] == 2 / ... (:77-91) from "how many ranks hold >= k cards", all 13 nibbles at once
  // (round 5: it was a 13-step compare-select chain tracking the two largest counts).  A nibble holds 0..8 (eight cards of one rank at most).
  const uint64_t M = 0x1111111111111ull;
  const uint64_t b0 = hist & M, b1 = (hist >> 1) & M, b2 = (hist >> 2) & M, b3 = (hist >> 3) & M;
  const uint64_t ge1 = b0 | b1 | b2 | b3, ge2 = b1 | b2 | b3, ge3 = b3 | b2 | (b1 & b0), ge4 = b3 | b2, ge5 = b3 | (b2 & (b1 | b0));
  const int n2 = __popcll(ge2), n3 = __popcll(ge3), n4 = __popcll(ge4), n5 = __popcll(ge5);
  bool flush = (__popc(suits) == 1) && n >= 5;                                    // :60
  // five consecutive ranks present, on the nibble-spaced presence bits themselves; A,2,3,4,5 = ranks 12, 0, 1, 2, 3 (:66-73)
  const uint64_t run = ge1 & (ge1 >> 4) & (ge1 >> 8) & (ge1 >> 12) & (ge1 >> 16);
  const uint64_t wheel = 0x1000000001111ull;
  bool straight = run != 0ull || (ge1 & wheel) == wheel;
  // (>= 5 distinct ranks is implied by either pattern)
  if (straight && flush) return 8;
  if (n4 >= 1 && n5 == 0) return 7;              // counts[0] == 4
  if (n3 == 1 && n4 == 0 && n2 >= 2) return 6;   // counts[0] == 3 and counts[1] == 2
  if (flush) return 5;
  if (straight && n >= 5) return 4;
  if (n3 >= 1 && n4 == 0) return 3;              // counts[0] == 3
  if (n2 >= 2 && n3 == 0) return 2;              // counts[0] == 2 and counts[1] == 2
  if (n2 >= 1 && n3 == 0) return 1;              // counts[0] == 2
  return 0;
}

// balatro_game.py:95-109 _draw_cards: append the lowest deck indexes not in hand until len == hand_size
__device__ __forceinline__ void bg_draw_cards(Env& e) {
  if (e.nhand >= e.hand_size || e.nhand >= 8) return; // nothing to draw (the usual case at the end of a shop: played cards never left the hand)
  if (e.nhand == 0) { // an EMPTY hand (every blind select, every reset: most "other" service steps): the lowest free indexes are 0, 1, 2, ... -- no loop
    int k = e.hand_size < 8 ? e.hand_size : 8;
    const int have = 52 - e.ndrop;
    k = k < have ? k : have;
    const uint64_t km = k >= 8 ? ~0ull : ((1ull << (8 * k)) - 1ull);
    e.hand = (e.hand & ~km) | (0x0706050403020100ull & km);   // (bytes beyond the hand keep what they held, as the loop below leaves them)
    e.nhand = k;
    return;
  }
  uint64_t inhand = 0;
#pragma unroll 1
  for (int i = 0; i < e.nhand; i++) inhand |= 1ull << bg_get8(e.hand, i);
#pragma unroll 1
  while (e.nhand < e.hand_size && e.nhand < 8) {
    uint64_t freeset = ~inhand & ((1ull << (52 - e.ndrop)) - 1); // range(len(deck)); Cryptid's copies sit behind every real card and are never reached
    if (!freeset) break;
    int dnext = __ffsll((long long)freeset) - 1;
    e.hand = bg_set8(e.hand, e.nhand, dnext);
    e.nhand++;
    inhand |= 1ull << dnext;
  }
}

// counter-hash policy: splitmix64 finaliser, high 32 bits
__device__ __forceinline__ uint32_t bg_policy_hash(uint64_t policy_seed, uint64_t env_index, uint64_t t) {
  uint64_t x = policy_seed + 0x9E3779B97F4A7C15ull * (env_index + 1) + 0xD1B54A32D192ED03ull * (t + 1);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}
