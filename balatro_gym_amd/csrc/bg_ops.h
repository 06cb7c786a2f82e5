// bg_ops.h -- operator-level entry points: the units the reference exposes as callables of their own
// (BalatroGame._classify_hand, UnifiedScorer.score_hand, BalatroSimulator.evaluate_hand / calculate_score), batched, lane = case.
// They run the SAME device functions as the step path (bg_classify, bg_joker_chain, bg_hand_base, bg_card_chips): the golden
// vectors the reference's units pin (tests/golden/classify.npz, score_hand.json, sim_eval.json) therefore reach the HIP code
// itself, hand types a random env almost never produces included.  Included by bg_lib.hip (one translation unit) after the
// MT19937 helpers.
#pragma once

// ---------------------------------------------------------------------------------------------------------
// BalatroGame._classify_hand (balatro_game.py:40-93)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BG_BLOCK) void bg_classify_batch_kernel(const uint8_t* __restrict__ cards, const uint8_t* __restrict__ n,
                                                                    uint8_t* __restrict__ out, long long m) {
  const long long i = (long long)blockIdx.x * BG_BLOCK + threadIdx.x;
  if (i >= m) return;
  const uint2 c = ((const uint2*)cards)[i]; // 8 card codes
  int k = n[i];
  out[i] = (uint8_t)bg_classify(((uint64_t)c.y << 32) | c.x, k > 8 ? 8 : k);
}

// The other mapping SURVEY 7.6 asks to benchmark: EIGHT lanes per hand, lane = card.  Same-rank counts come from seven
// `__shfl_xor` exchanges inside the 8-lane group, the rank-presence / suit sets and the two largest group sizes from three-step
// butterflies; every lane of the group ends with the hand type, lane 0 stores it.  A wave classifies 8 hands instead of 64 in
// ~70 instead of ~150 instructions: shorter dependent chain, 3.7x more issue slots per hand (A/B: tools/ab_lanes.py).
template <class T> __device__ __forceinline__ T bg_g8_sum(T v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); return v; }
__device__ __forceinline__ uint32_t bg_g8_or(uint32_t v) { v |= __shfl_xor(v, 1); v |= __shfl_xor(v, 2); v |= __shfl_xor(v, 4); return v; }
__device__ __forceinline__ int bg_g8_max(int v) { v = max(v, __shfl_xor(v, 1)); v = max(v, __shfl_xor(v, 2)); v = max(v, __shfl_xor(v, 4)); return v; }

__device__ __forceinline__ int bg_classify_l8(int code, bool valid, int n) {
  const int r = code >> 2;
  const uint32_t me = valid ? (0x80u | (uint32_t)code) : 0u;
  int cnt = valid ? 1 : 0;
#pragma unroll
  for (int k = 1; k < 8; k++) {
    const uint32_t o = __shfl_xor(me, k);
    cnt += (valid && (o & 0x80u) && (int)((o & 0x7fu) >> 2) == r) ? 1 : 0;
  }
  // largest group: key = size << 4 | rank; second largest = the biggest size among the OTHER ranks
  const int top = bg_g8_max(valid ? (cnt << 4) | r : 0);
  const int c0 = top >> 4, r0 = top & 15;
  const int c1 = bg_g8_max((valid && r != r0) ? cnt : 0);
  const uint32_t sets = bg_g8_or(valid ? (1u << r) | (1u << (16 + (code & 3))) : 0u);
  const uint32_t present = sets & 0x1fffu, suits = sets >> 16;
  if (n <= 0) return 0;
  const bool flush = (__popc(suits) == 1) && n >= 5;                                          // balatro_game.py:60
  const uint32_t run = present & (present >> 1) & (present >> 2) & (present >> 3) & (present >> 4);
  const bool straight = run != 0 || ((present & 0x100fu) == 0x100fu);                         // :66-73
  if (straight && flush) return 8;
  if (c0 == 4) return 7;
  if (c0 == 3 && c1 == 2) return 6;
  if (flush) return 5;
  if (straight && n >= 5) return 4;
  if (c0 == 3) return 3;
  if (c0 == 2 && c1 == 2) return 2;
  if (c0 == 2) return 1;
  return 0;
}
__global__ __launch_bounds__(BG_BLOCK) void bg_classify_batch_l8_kernel(const uint8_t* __restrict__ cards, const uint8_t* __restrict__ n,
                                                                       uint8_t* __restrict__ out, long long m) {
  const long long t = (long long)blockIdx.x * BG_BLOCK + threadIdx.x;
  const long long i = t >> 3;
  const int g = (int)(t & 7);
  const bool live = i < m; // whole 8-lane groups
  int k = live ? n[i] : 0;
  k = k > 8 ? 8 : k;
  const int code = live ? cards[t] : 0; // 64 consecutive bytes per wave
  const int ht = bg_classify_l8(code, g < k, k);
  if (live && g == 0) out[i] = (uint8_t)ht;
}

// ---------------------------------------------------------------------------------------------------------
// UnifiedScorer.score_hand (unified_scoring.py:111-299) with game_state['jokers'] = joker NAMES (as unified_scoring.py:313-351
// calls it), the process-global `random` seeded random.seed(gseed) per case.
// case record (int32[BG_SCORE_CASE_WORDS]) and result record (int64[BG_SCORE_OUT_WORDS]): include/balatro_mi355x.h
// ---------------------------------------------------------------------------------------------------------
#define BG_SC_CARDS 0      // 8 x (rank 2..14 or 0 for a STONE card, suit 0..3 = C D H S or 4 = 'Stone', chip value)
#define BG_SC_NCARDS 24    // len(context.cards)
#define BG_SC_NSCORING 25  // the first nscoring cards are context.scoring_cards
#define BG_SC_HAND_TYPE 26
#define BG_SC_STYLE 27     // 0 = the env's names ('One Pair', 'Three Kind', 'Four Kind'), 1 = balatro_sim's ('Pair', ...)
#define BG_SC_LEVEL 28
#define BG_SC_NJOKERS 29
#define BG_SC_JOKERS 30    // 5 ids (jokers.py)
#define BG_SC_HANDS_LEFT 35
#define BG_SC_DISCARDS_LEFT 36
#define BG_SC_DECK_LEN 37
#define BG_SC_GSEED 38     // random.seed(gseed), gseed < 2**32
#define BG_SC_WORDS 40

// random.seed(gseed) per case: init_by_array([gseed]); two blocks of output cover the longest chain (8 x 5 x 2 + 16 + rejections)
__global__ __launch_bounds__(BG_BLOCK) void bg_score_seed_kernel(BgDev d, const int32_t* __restrict__ cases) {
  const int i = blockIdx.x * BG_BLOCK + threadIdx.x;
  if (i >= d.N) return;
  uint32_t* g0 = bg_gblock(d, i, 0);
  uint32_t* g1 = bg_gblock(d, i, 1);
  bg_mt_seed(g0, (uint32_t)cases[(size_t)i * BG_SC_WORDS + BG_SC_GSEED]);
  bg_mt_twist(g0, g0);
  bg_mt_twist(g0, g1);
  for (int k = 0; k < 16; k++) g0[BG_MT_N + k] = g1[k]; // the spare words behind a block mirror the head of the next one
}

__device__ __forceinline__ void bg_score_env(Env& e, const int32_t* c, int nj) {
  uint4 z[BG_NHOT];
#pragma unroll
  for (int k = 0; k < BG_NHOT; k++) z[k] = make_uint4(0, 0, 0, 0);
  bg_unpack(z, e);
  e.g_cur = 0; e.g_idx = 0; e.g_cons = 0; e.g_valid = 2;
  e.hands_left = c[BG_SC_HANDS_LEFT]; e.discards_left = c[BG_SC_DISCARDS_LEFT];
  e.njokers = nj; e.jokers = 0;
  for (int j = 0; j < nj; j++) e.jokers |= (uint64_t)(c[BG_SC_JOKERS + j] & 0xff) << (8 * j);
}
__device__ __forceinline__ void bg_score_store(const BgDev& d, int i, Env& e, int64_t* __restrict__ out, int64_t chips, int64_t mult, double x_mult, int money) {
  bg_gnorm(d, e);
  int64_t* o = out + (size_t)i * 8;
  o[0] = (int64_t)((double)(chips * mult) * x_mult);                                    // unified_scoring.py:286
  o[1] = chips; o[2] = mult; o[3] = __double_as_longlong(x_mult); o[4] = money;
  o[5] = (int64_t)e.g_cons * BG_MT_N + e.g_idx;                                         // words of the global stream consumed
  o[6] = (int64_t)bg_gpeek(d, i, e, 0);                                                 // the word the next getrandbits(32) returns
  o[7] = 0;
}

// lane = case
__global__ __launch_bounds__(BG_BLOCK) void bg_score_hand_batch_kernel(BgDev d, const int32_t* __restrict__ cases, int64_t* __restrict__ out) {
  __shared__ uint32_t win[BG_WIN][BG_BLOCK];
  __shared__ JTables jt;
  bg_tables_init(&jt);
  const int i = blockIdx.x * BG_BLOCK + threadIdx.x;
  if (i >= d.N) return;
  const int32_t* c = cases + (size_t)i * BG_SC_WORDS;
  int nj = c[BG_SC_NJOKERS]; nj = nj < 0 ? 0 : (nj > 5 ? 5 : nj);
  Env e;
  bg_score_env(e, c, nj);
  RngWin w;
  bg_win_init(w, &win[0][threadIdx.x], &jt);
  int ncards = c[BG_SC_NCARDS], nsc = c[BG_SC_NSCORING];
  ncards = ncards < 0 ? 0 : (ncards > 8 ? 8 : ncards); nsc = nsc < 0 ? 0 : (nsc > ncards ? ncards : nsc);
  ChainIn in;
  in.phist = 0; in.pcodes = 0; in.scnt = 0; in.stone = 0; in.n = nsc; in.ht = c[BG_SC_HAND_TYPE];
  in.kings = 0; in.queens = 0; in.all_black = true; in.deck_len = c[BG_SC_DECK_LEN]; in.style = c[BG_SC_STYLE];
  int chip_sum = 0;
  for (int k = 0; k < ncards; k++) {
    const int rank = c[BG_SC_CARDS + 3 * k], suit = c[BG_SC_CARDS + 3 * k + 1], chipv = c[BG_SC_CARDS + 3 * k + 2];
    in.kings += rank == 13; in.queens += rank == 12;
    if (!(suit == 3 || suit == 0)) in.all_black = false;
    if (k < nsc) {
      chip_sum += chipv;                                                                // unified_scoring.py:141-155
      if (suit == 4) { in.phist += 1ull; in.scnt += 1u << 16; in.stone |= 1u << k; }   // rank 0, suit 'Stone' (balatro_env_2.py:305-309)
      else {
        in.phist += 1ull << (4 * rank); in.scnt += 1u << (4 * suit);
        in.pcodes |= (uint64_t)(((rank - 2) << 2) | suit) << (8 * k);
      }
    }
  }
  int bchips, bmult;
  bg_hand_base(in.ht, c[BG_SC_LEVEL], bchips, bmult);                                   // :120
  int64_t chips = bchips + chip_sum, mult = bmult;
  double x_mult = 1.0;
  int money = 0;
  if (nj > 0) bg_joker_chain<true, Deck0>(d, i, e, w, in, chips, mult, x_mult, money);
  bg_score_store(d, i, e, out, chips, mult, x_mult, money);
}

// EIGHT lanes per case (SURVEY 7.6's other mapping): lane = card while the hand is gathered (the histograms are three-step
// `__shfl_xor` sums over the 8-lane group), lane = joker for the individual phase (bg_chain_joker per lane, totals by shuffle),
// lane = card again for Bloodstone's RNG words; the main phase (joker ORDER matters: x factors) runs redundantly on all eight
// lanes through the same bg_chain_main as the step path.  Lane 0 stores.
__global__ __launch_bounds__(BG_BLOCK) void bg_score_hand_batch_l8_kernel(BgDev d, const int32_t* __restrict__ cases, int64_t* __restrict__ out) {
  __shared__ uint32_t win[BG_WIN][BG_BLOCK];
  __shared__ JTables jt;
  bg_tables_init(&jt);
  const int i = blockIdx.x * (BG_BLOCK / 8) + (threadIdx.x >> 3), g = threadIdx.x & 7;
  const bool live = i < d.N; // whole 8-lane groups; dead groups keep running (shuffles) on case 0 and store nothing
  const int ci = live ? i : 0;
  const int32_t* c = cases + (size_t)ci * BG_SC_WORDS;
  int nj = c[BG_SC_NJOKERS]; nj = nj < 0 ? 0 : (nj > 5 ? 5 : nj);
  Env e;
  bg_score_env(e, c, nj);
  RngWin w;
  bg_win_init(w, &win[0][threadIdx.x], &jt);
  int ncards = c[BG_SC_NCARDS], nsc = c[BG_SC_NSCORING];
  ncards = ncards < 0 ? 0 : (ncards > 8 ? 8 : ncards); nsc = nsc < 0 ? 0 : (nsc > ncards ? ncards : nsc);
  // ---- lane = card
  const bool has = g < ncards, sc = g < nsc;
  const int rank = has ? c[BG_SC_CARDS + 3 * g] : 0, suit = has ? c[BG_SC_CARDS + 3 * g + 1] : 0, chipv = has ? c[BG_SC_CARDS + 3 * g + 2] : 0;
  const bool st = sc && suit == 4;
  const int code = (sc && !st) ? (((rank - 2) << 2) | suit) : 0;
  ChainIn in;
  in.n = nsc; in.ht = c[BG_SC_HAND_TYPE]; in.deck_len = c[BG_SC_DECK_LEN]; in.style = c[BG_SC_STYLE];
  in.phist = bg_g8_sum<unsigned long long>(sc ? (st ? 1ull : 1ull << (4 * rank)) : 0ull);
  in.pcodes = bg_g8_sum<unsigned long long>((unsigned long long)code << (8 * g)); // one byte per lane: the sum is the union
  in.scnt = bg_g8_sum<uint32_t>(sc ? (st ? 1u << 16 : 1u << (4 * suit)) : 0u);
  const int chip_sum = bg_g8_sum<int>(sc ? chipv : 0);                                // unified_scoring.py:141-155
  const uint32_t misc = bg_g8_sum<uint32_t>((has && rank == 13 ? 1u : 0u) | (has && rank == 12 ? 1u << 4 : 0u) |
                                            ((has && !(suit == 3 || suit == 0)) ? 1u << 8 : 0u) | (st ? 1u << (16 + g) : 0u));
  in.kings = (int)(misc & 15u); in.queens = (int)((misc >> 4) & 15u); in.all_black = ((misc >> 8) & 15u) == 0; in.stone = misc >> 16;
  int bchips, bmult;
  bg_hand_base(in.ht, c[BG_SC_LEVEL], bchips, bmult);                                   // :120
  int64_t chips = bchips + chip_sum, mult = bmult;
  double x_mult = 1.0;
  int money = 0;
  if (nj > 0) { // uniform over the group
    // ---- lane = joker: individual phase (unified_scoring.py:174-209)
    const int id = g < nj ? (int)((e.jokers >> (8 * g)) & 0xff) : 0;
    const ChainJ cj = bg_chain_joker(w.jt->jd[id], w.jt->jr[id], in.phist, in.scnt);
    const int ic = bg_g8_sum<int>(cj.ic), im = bg_g8_sum<int>(cj.im);
    int xexp = bg_g8_sum<int>(cj.xexp);
    const uint32_t m8 = bg_g8_or(cj.sp == 1u ? 1u << g : 0u), mb = bg_g8_or(cj.sp == 2u ? 1u << g : 0u); // slots with an 8 Ball / a Bloodstone
    money += bg_g8_sum<int>(id == 116 ? (int)((in.scnt >> 4) & 0xfu) : 0);             // Rough Gem: $1 per Diamond (:160)
    uint32_t dms[5];
#pragma unroll
    for (int j = 0; j < 5; j++) dms[j] = w.jt->jm[j < nj ? (int)((e.jokers >> (8 * j)) & 0xff) : 0];
    bg_gnorm(d, e);
    const int nb8 = __popc(m8);
    const int n8 = nb8 ? (int)((in.phist >> 32) & 0xf) : 0;                             // 8 Ball: one extra random() per played 8 and 8 Ball (:167)
    const int consumed = 2 * nsc * nj + 2 * n8 * nb8;
    // ---- lane = card: the word of this card for every Bloodstone owned (uniform over the group: the shuffles stay converged)
    const uint32_t eightmask = bg_g8_or((sc && !st && rank == 8 && nb8) ? 1u << g : 0u);
    uint32_t mm = ((in.scnt >> 8) & 0xfu) ? mb : 0u;
    while (mm) {
      const int jb = __ffs((int)mm) - 1;
      mm &= mm - 1u;
      const int boff = bg_chain_blood_off(g, code, st, nsc, nj, jb, m8, nb8, true, __popc(eightmask & ((1u << g) - 1u)));
      const uint32_t ra = boff >= 0 ? bg_gpeek(d, ci, e, boff) : 0x80000000u;
      xexp += bg_g8_sum<int>((int)((ra >> 31) ^ 1u));
    }
    uint32_t mw[12], avail = 0;
    bg_gpeek12(d, ci, e, consumed, mw, avail); // the same three 16-byte loads on all eight lanes (one request)
    bg_gskip(d, e, consumed);
    chips += ic; mult += im;
    x_mult *= (double)(1ull << xexp);
    bg_chain_main<true, Deck0>(d, ci, e, w, in, dms, mw, avail, chips, mult, x_mult);
  }
  if (live && g == 0) bg_score_store(d, i, e, out, chips, mult, x_mult, money);
}
