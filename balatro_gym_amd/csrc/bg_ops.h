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

__global__ __launch_bounds__(BG_BLOCK) void bg_score_hand_batch_kernel(BgDev d, const int32_t* __restrict__ cases, int64_t* __restrict__ out) {
  __shared__ uint32_t win[BG_WIN][BG_BLOCK];
  __shared__ JTables jt;
  bg_tables_init(&jt);
  const int i = blockIdx.x * BG_BLOCK + threadIdx.x;
  if (i >= d.N) return;
  const int32_t* c = cases + (size_t)i * BG_SC_WORDS;
  // random.seed(gseed): init_by_array([gseed]); two blocks of output cover the longest chain (8 x 5 x 2 + 16 + rejections)
  uint32_t* g0 = bg_gblock(d, i, 0);
  uint32_t* g1 = bg_gblock(d, i, 1);
  bg_mt_seed(g0, (uint32_t)c[BG_SC_GSEED]);
  bg_mt_twist(g0, g0);
  bg_mt_twist(g0, g1);
  for (int k = 0; k < 16; k++) g0[BG_MT_N + k] = g1[k]; // the spare words behind a block mirror the head of the next one
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  Env e;
  {
    uint4 z[BG_NHOT];
#pragma unroll
    for (int k = 0; k < BG_NHOT; k++) z[k] = make_uint4(0, 0, 0, 0);
    bg_unpack(z, e);
  }
  e.g_cur = 0; e.g_idx = 0; e.g_cons = 0; e.g_valid = 2;
  e.hands_left = c[BG_SC_HANDS_LEFT]; e.discards_left = c[BG_SC_DISCARDS_LEFT];
  int nj = c[BG_SC_NJOKERS]; nj = nj < 0 ? 0 : (nj > 5 ? 5 : nj);
  e.njokers = nj; e.jokers = 0;
  for (int j = 0; j < nj; j++) e.jokers |= (uint64_t)(c[BG_SC_JOKERS + j] & 0xff) << (8 * j);
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
  bg_gnorm(d, e);
  int64_t* o = out + (size_t)i * 8;
  o[0] = (int64_t)((double)(chips * mult) * x_mult);                                    // :286
  o[1] = chips; o[2] = mult; o[3] = __double_as_longlong(x_mult); o[4] = money;
  o[5] = (int64_t)e.g_cons * BG_MT_N + e.g_idx;                                         // words of the global stream consumed
  o[6] = (int64_t)bg_gpeek(d, i, e, 0);                                                 // the word the next getrandbits(32) returns
  o[7] = 0;
}
