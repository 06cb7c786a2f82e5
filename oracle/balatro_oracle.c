/* balatro_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See balatro_oracle.h.
 *
 * Scalar C restatement of the reference hot path.  Citations `file:line` are relative to
 * /root/reference/balatro_gym/.  All floating point is IEEE double evaluated left-to-right exactly as the
 * Python expressions are (build with -ffp-contract=off); libm/numpy results the reference depends on
 * (np.log10, 0.8**n, 1.15**k, 1.5**k) come from bo_tables.h, generated on the reference platform.
 */
#include "balatro_oracle.h"
#include "bo_tables.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * CPython random.Random (Modules/_randommodule.c, Lib/random.py 3.10) -- SURVEY.md Appendix B
 * ---------------------------------------------------------------------------------------------- */
#define MT_N 624
#define MT_M 397

static void mt_init_genrand(bo_mt* m, uint32_t s) {
  m->mt[0] = s;
  for (int i = 1; i < MT_N; i++) m->mt[i] = 1812433253u * (m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) + (uint32_t)i;
  m->mti = MT_N;
}

/* random_seed(): key = little-endian 32-bit words of abs(seed), at least one word; init_by_array(key). */
void bo_mt_seed(bo_mt* m, uint64_t seed) {
  uint32_t key[2];
  int klen = 1;
  key[0] = (uint32_t)seed;
  key[1] = (uint32_t)(seed >> 32);
  if (key[1]) klen = 2;
  mt_init_genrand(m, 19650218u);
  uint32_t* mt = m->mt;
  int i = 1, j = 0;
  int k = MT_N > klen ? MT_N : klen;
  for (; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
    i++; j++;
    if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
    if (j >= klen) j = 0;
  }
  for (k = MT_N - 1; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
    i++;
    if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
  }
  mt[0] = 0x80000000u;
  m->mti = MT_N;
}

uint32_t bo_mt_u32(bo_mt* m) {
  uint32_t* mt = m->mt;
  if (m->mti >= MT_N) {
    int kk;
    uint32_t y;
    for (kk = 0; kk < MT_N - MT_M; kk++) {
      y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; kk < MT_N - 1; kk++) {
      y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    m->mti = 0;
  }
  uint32_t y = mt[m->mti++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

/* getrandbits(k), 0 < k <= 32: genrand_uint32() >> (32 - k) */
uint32_t bo_mt_getrandbits(bo_mt* m, int k) { return bo_mt_u32(m) >> (32 - k); }

static int bit_length64(uint64_t n) { int k = 0; while (n) { k++; n >>= 1; } return k; }

/* Lib/random.py _randbelow_with_getrandbits */
uint32_t bo_mt_randbelow(bo_mt* m, uint32_t n) {
  if (!n) return 0;
  int k = bit_length64(n);
  uint32_t r = bo_mt_getrandbits(m, k);
  while (r >= n) r = bo_mt_getrandbits(m, k);
  return r;
}

/* same for n up to 2**32 (k = 33 -> two words, low word first) */
uint64_t bo_mt_randbelow64(bo_mt* m, uint64_t n) {
  if (n <= 0xffffffffull) return bo_mt_randbelow(m, (uint32_t)n);
  int k = bit_length64(n); /* 33 */
  for (;;) {
    uint64_t lo = bo_mt_u32(m);
    uint64_t hi = bo_mt_u32(m) >> (32 - (k - 32));
    uint64_t r = lo | (hi << 32);
    if (r < n) return r;
  }
}

/* random(): (a*67108864.0+b)*(1.0/9007199254740992.0), a = u32>>5, b = u32>>6 */
double bo_mt_random(bo_mt* m) {
  uint32_t a = bo_mt_u32(m) >> 5, b = bo_mt_u32(m) >> 6;
  return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}

/* shuffle(x): for i in reversed(range(1, len(x))): j = _randbelow(i+1); x[i], x[j] = x[j], x[i] */
void bo_mt_shuffle_u8(bo_mt* m, uint8_t* x, int n) {
  for (int i = n - 1; i >= 1; i--) {
    uint32_t j = bo_mt_randbelow(m, (uint32_t)(i + 1));
    uint8_t t = x[i]; x[i] = x[j]; x[j] = t;
  }
}

/* ------------------------------------------------------------------------------------------------
 * Pure game functions
 * ---------------------------------------------------------------------------------------------- */

/* cards.py:52-60 Rank.base_chips */
int bo_rank_base_chips(int rank) {
  if (rank <= 10) return rank;
  if (rank == 14) return 11;
  return 10;
}

/* scoring_engine.py:27-40, 87-101 */
static const int BASE_CHIPS[12] = {5, 10, 20, 30, 30, 35, 40, 60, 100, 120, 140, 160};
static const int BASE_MULT[12] = {1, 2, 2, 3, 4, 4, 4, 7, 8, 12, 14, 16};
void bo_hand_chips_mult(int hand_type, int level, int64_t* chips, int64_t* mult) {
  int lb = level - 1;
  *chips = BASE_CHIPS[hand_type] + lb * 10;
  *mult = BASE_MULT[hand_type] + lb;
}

/* balatro_env_2.py:55-74 */
int64_t bo_blind_chips(int ante, int blind) {
  static const int T[8][3] = {{300, 450, 600},    {450, 675, 900},    {600, 900, 1200},   {900, 1350, 1800},
                              {1350, 2025, 2700}, {2100, 3150, 4200}, {3300, 4950, 6600}, {5250, 7875, 10500}};
  if (ante <= 8) return T[ante - 1][blind];
  int k = ante - 8;
  if (k > 92) k = 92;
  return (int64_t)((double)T[7][blind] * BO_POW15[k]);
}

/* balatro_game.py:40-93 _classify_hand on card codes (rank = code/4 + 2, suit = code%4) */
int bo_classify(const uint8_t* cards, int n) {
  if (n <= 0) return 0; /* :42-43 HIGH_CARD */
  int rank_counts[15] = {0}, suit_counts[4] = {0};
  for (int i = 0; i < n; i++) {
    rank_counts[(cards[i] >> 2) + 2]++;
    suit_counts[cards[i] & 3]++;
  }
  int nsuits = 0;
  for (int s = 0; s < 4; s++) nsuits += suit_counts[s] > 0;
  /* counts = sorted(rank_counts.values(), reverse=True): only the two largest matter (:59) */
  int c0 = 0, c1 = 0, ndistinct = 0;
  for (int r = 2; r <= 14; r++) {
    int c = rank_counts[r];
    if (!c) continue;
    ndistinct++;
    if (c > c0) { c1 = c0; c0 = c; } else if (c > c1) c1 = c;
  }
  int is_flush = (nsuits == 1) && (n >= 5); /* :60 */
  int is_straight = 0;                      /* :63-73 */
  if (ndistinct >= 5) {
    int sorted[13], k = 0;
    for (int r = 2; r <= 14; r++) if (rank_counts[r]) sorted[k++] = r;
    for (int i = 0; i + 4 < k; i++) if (sorted[i + 4] - sorted[i] == 4) { is_straight = 1; break; }
    if (!is_straight && rank_counts[14] && rank_counts[2] && rank_counts[3] && rank_counts[4] && rank_counts[5]) is_straight = 1;
  }
  if (is_straight && is_flush && n >= 5) return 8;
  if (c0 == 4) return 7;
  if (ndistinct >= 2 && c0 == 3 && c1 == 2) return 6;
  if (is_flush && n >= 5) return 5;
  if (is_straight && n >= 5) return 4;
  if (c0 == 3) return 3;
  if (ndistinct >= 2 && c0 == 2 && c1 == 2) return 2;
  if (c0 == 2) return 1;
  return 0;
}

/* Harness convention G(seed): the per-env stand-in for the process-global `random` is seeded as if it were a
 * 17th DeterministicRNG stream (balatro_env_2.py:105 pattern with i = 16). */
uint64_t bo_global_seed(int64_t seed) {
  int64_t m = (seed + 16000) % 4294967296ll;
  if (m < 0) m += 4294967296ll;
  return (uint64_t)m;
}

/* Counter-based policy hash (harness convention; splitmix64 finaliser), returns the high 32 bits. */
uint32_t bo_policy_hash(uint64_t policy_seed, uint64_t env_index, uint64_t t) {
  uint64_t x = policy_seed + 0x9E3779B97F4A7C15ull * (env_index + 1) + 0xD1B54A32D192ED03ull * (t + 1);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}

/* ------------------------------------------------------------------------------------------------
 * Joker chain: unified_scoring.py:111-299 + complete_joker_effects.py:35-184, jokers identified by the
 * jokers.py id of their name.
 * ---------------------------------------------------------------------------------------------- */
enum { SUIT_C = 0, SUIT_D = 1, SUIT_H = 2, SUIT_S = 3, SUIT_STONE = 4 };

typedef struct { int64_t chips, mult; double x_mult; int64_t money; int has_x; } jeff;

/* hand-type name match (complete_joker_effects.py:64-80 vs balatro_env_2.py:674) */
enum { HN_PAIR, HN_THREE, HN_TWO_PAIR, HN_STRAIGHT, HN_FLUSH, HN_FOUR };
static int name_matches(int want, int hand_type, int style) {
  switch (want) {
    case HN_TWO_PAIR: return hand_type == 2;
    case HN_STRAIGHT: return hand_type == 4;
    case HN_FLUSH: return hand_type == 5;
    case HN_PAIR: return style == BO_NAMES_SIM && hand_type == 1;
    case HN_THREE: return style == BO_NAMES_SIM && hand_type == 3;
    case HN_FOUR: return style == BO_NAMES_SIM && hand_type == 7;
  }
  return 0;
}

static int has_suit(const bo_scard* sc, int n, int suit) { /* complete_joker_effects.py:245-250 */
  for (int i = 0; i < n; i++) if (sc[i].suit == suit) return 1;
  return 0;
}

/* complete_joker_effects.py:35-129 _scoring_effects.  The dict literal at :39-53 is built on EVERY call, so
 * random.randint(0, 23) is drawn for every joker. */
static void joker_main(int id, const bo_scard* cards, int ncards, const bo_scard* sc, int nsc, int hand_type,
                       int style, int njokers, int hands_left, int discards_left, int deck_len, bo_mt* g,
                       int* draws, jeff* e) {
  e->chips = 0; e->mult = 0; e->x_mult = 1.0; e->money = 0;
  uint32_t before = 0;
  (void)before;
  /* randint(0,23) -> randrange(0,24) -> _randbelow(24) */
  {
    int k = 5;
    uint32_t r;
    do { r = bo_mt_getrandbits(g, k); (*draws)++; } while (r >= 24);
    if (id == 27) { e->mult = r; return; } /* Misprint */
  }
  switch (id) {
    case 1: e->mult = 4; return;                                   /* Joker */
    case 136: e->chips = 250; return;                              /* Stuntman */
    case 38: e->mult = 15; return;                                 /* Gros Michel */
    case 61: e->x_mult = 3.0; return;                              /* Cavendish */
    case 16: if (nsc <= 3) e->mult = 20; return;                   /* Half Joker */
    case 34: e->mult = 3 * njokers; return;                        /* Abstract Joker */
    case 108: if (hands_left == 1) e->x_mult = 3.0; return;        /* Acrobat */
    case 23: if (discards_left == 0) e->mult = 15; return;         /* Mystic Summit */
    case 22: e->chips = 30 * discards_left; return;                /* Banner */
    case 53: e->chips = 2 * deck_len; return;                      /* Blue Joker */
    case 97: e->mult = 20; return;                                 /* Popcorn */
    case 50: e->chips = 100; return;                               /* Ice Cream */
    case 2: if (has_suit(sc, nsc, SUIT_D)) e->mult = 3; return;    /* Greedy */
    case 3: if (has_suit(sc, nsc, SUIT_H)) e->mult = 3; return;    /* Lusty */
    case 4: if (has_suit(sc, nsc, SUIT_S)) e->mult = 3; return;    /* Wrathful */
    case 5: if (has_suit(sc, nsc, SUIT_C)) e->mult = 3; return;    /* Gluttonous */
    case 6: if (name_matches(HN_PAIR, hand_type, style)) e->mult = 8; return;
    case 7: if (name_matches(HN_THREE, hand_type, style)) e->mult = 12; return;
    case 8: if (name_matches(HN_TWO_PAIR, hand_type, style)) e->mult = 10; return;
    case 9: if (name_matches(HN_STRAIGHT, hand_type, style)) e->mult = 12; return;
    case 10: if (name_matches(HN_FLUSH, hand_type, style)) e->mult = 10; return;
    case 11: if (name_matches(HN_PAIR, hand_type, style)) e->chips = 50; return;
    case 12: if (name_matches(HN_THREE, hand_type, style)) e->chips = 100; return;
    case 13: if (name_matches(HN_TWO_PAIR, hand_type, style)) e->chips = 80; return;
    case 14: if (name_matches(HN_STRAIGHT, hand_type, style)) e->chips = 100; return;
    case 15: if (name_matches(HN_FLUSH, hand_type, style)) e->chips = 80; return;
    case 131: if (name_matches(HN_PAIR, hand_type, style)) e->x_mult = 2.0; return;
    case 132: if (name_matches(HN_THREE, hand_type, style)) e->x_mult = 3.0; return;
    case 133: if (name_matches(HN_FOUR, hand_type, style)) e->x_mult = 4.0; return;
    case 134: if (name_matches(HN_STRAIGHT, hand_type, style)) e->x_mult = 3.0; return;
    case 135: if (name_matches(HN_FLUSH, hand_type, style)) e->x_mult = 2.0; return;
    case 48: { /* Blackboard :98-102: all `cards` are Spades or Clubs */
      int all = 1;
      for (int i = 0; i < ncards; i++) if (!(cards[i].suit == SUIT_S || cards[i].suit == SUIT_C)) all = 0;
      if (all) e->x_mult = 3.0;
      return;
    }
    case 128: { /* Seeing Double :104-108 */
      int seen = 0;
      for (int i = 0; i < nsc; i++) seen |= 1 << sc[i].suit;
      int n = __builtin_popcount((unsigned)seen);
      if ((seen & (1 << SUIT_C)) && n > 1) e->x_mult = 2.0;
      return;
    }
    case 122: { /* Flower Pot :110-114 ('Stone' counts as a suit string) */
      int seen = 0;
      for (int i = 0; i < nsc; i++) seen |= 1 << sc[i].suit;
      if (__builtin_popcount((unsigned)seen) == 4) e->x_mult = 3.0;
      return;
    }
    case 72: { /* Baron :116-120: 1.5 ** kings over `cards` */
      int kings = 0;
      for (int i = 0; i < ncards; i++) kings += cards[i].rank == 13;
      if (kings > 0) e->x_mult = BO_POW15[kings];
      return;
    }
    case 140: { /* Shoot the Moon :122-126 */
      int q = 0;
      for (int i = 0; i < ncards; i++) q += cards[i].rank == 12;
      if (q > 0) e->mult = 13 * q;
      return;
    }
    default: return;
  }
}

/* complete_joker_effects.py:131-184 _individual_scoring_effects.  suit_effects at :157-162 is built on every
 * call, so one random.random() (Bloodstone) is drawn per (card, joker) pair. */
static void joker_individual(int id, const bo_scard* card, bo_mt* g, int* draws, jeff* e) {
  e->chips = 0; e->mult = 0; e->x_mult = 1.0; e->money = 0;
  double blood = bo_mt_random(g);
  (*draws) += 2;
  int r = card->rank;
  switch (id) {
    case 31: if (r == 2 || r == 3 || r == 5 || r == 8 || r == 14) e->mult = 8; return;         /* Fibonacci */
    case 39: if (r == 2 || r == 4 || r == 6 || r == 8 || r == 10) e->mult = 4; return;         /* Even Steven */
    case 40: if (r == 3 || r == 5 || r == 7 || r == 9 || r == 14) e->chips = 31; return;       /* Odd Todd */
    case 41: if (r == 14) { e->chips = 20; e->mult = 4; } return;                                /* Scholar */
    case 101: if (r == 4 || r == 10) { e->chips = 10; e->mult = 4; } return;                     /* Walkie Talkie */
    case 124: if (r == 2) e->chips = 8; return;                                                  /* Wee Joker */
    case 26: if (r == 8) { (void)bo_mt_random(g); (*draws) += 2; } return;                       /* 8 Ball :167 */
    case 33: if (r >= 11 && r <= 13) e->chips = 30; return;                                      /* Scary Face */
    case 104: if (r >= 11 && r <= 13) e->mult = 5; return;                                       /* Smiley Face */
    case 147: if (r == 12 || r == 13) e->x_mult = 2.0; return;                                   /* Triboulet */
    case 118: if (card->suit == SUIT_S) e->chips = 50; return;                                   /* Arrowhead */
    case 119: if (card->suit == SUIT_C) e->mult = 7; return;                                     /* Onyx Agate */
    case 116: if (card->suit == SUIT_D) e->money = 1; return;                                    /* Rough Gem */
    case 117: if (card->suit == SUIT_H && blood < 0.5) e->x_mult = 2.0; return;                  /* Bloodstone */
    default: return;
  }
}

void bo_score_hand(const bo_scard* cards, int ncards, const bo_scard* sc, int nsc, int hand_type, int style,
                   int level, const int32_t* jokers, int njokers, int hands_left, int discards_left,
                   int deck_len, bo_mt* g, bo_score_out* out) {
  int64_t chips, mult;
  bo_hand_chips_mult(hand_type, level, &chips, &mult); /* unified_scoring.py:120 */
  double x_mult = 1.0;
  int64_t money = 0;
  int draws = 0;
  for (int i = 0; i < nsc; i++) chips += sc[i].chips; /* :141-155 */
  /* :158-172 before-scoring phase only mutates joker_states (Green Joker / Ride the Bus): no score effect */
  /* :174-204 individual phase: card-major, joker-minor */
  int64_t ic = 0, im = 0;
  double ix = 1.0;
  for (int c = 0; c < nsc; c++)
    for (int j = 0; j < njokers; j++) {
      jeff e;
      joker_individual(jokers[j], &sc[c], g, &draws, &e);
      ic += e.chips; im += e.mult; ix *= e.x_mult; money += e.money;
    }
  chips += ic; mult += im; x_mult *= ix; /* :207-209 */
  /* :216-244 main phase */
  for (int j = 0; j < njokers; j++) {
    jeff e;
    joker_main(jokers[j], cards, ncards, sc, nsc, hand_type, style, njokers, hands_left, discards_left, deck_len,
               g, &draws, &e);
    chips += e.chips; mult += e.mult;
    chips = (int64_t)((double)chips * 1.0); /* :231 int(chips * effect.chips_mult) */
    mult = (int64_t)((double)mult * 1.0);   /* :232 */
    x_mult *= e.x_mult;
    money += e.money;
  }
  /* :246-282 the enhancement/edition block compares IntEnums with strings: no-op */
  out->score = (int64_t)((double)(chips * mult) * x_mult); /* :286 int(chips * mult * x_mult) */
  out->chips = chips; out->mult = mult; out->x_mult = x_mult; out->money = money; out->draws = draws;
}

/* ------------------------------------------------------------------------------------------------
 * Env
 * ---------------------------------------------------------------------------------------------- */
static bo_mt* stream(bo_env* e, int i) { /* balatro_env_2.py:93-106, seeded lazily (same values) */
  if (!e->stream_ready[i]) {
    int64_t s = (e->master_seed + (int64_t)i * 1000) % 4294967296ll;
    if (s < 0) s += 4294967296ll;
    bo_mt_seed(&e->stream[i], (uint64_t)s);
    e->stream_ready[i] = 1;
  }
  return &e->stream[i];
}
enum { ST_DECK = 0, ST_CARD_DRAW = 1, ST_SHOP_GEN = 2, ST_CARD_ENH = 11, ST_SEAL = 13 };

bo_env* bo_create(uint32_t flags, int32_t max_ante) {
  bo_env* e = (bo_env*)calloc(1, sizeof(bo_env));
  e->flags = flags;
  e->max_ante = max_ante;
  return e;
}
void bo_destroy(bo_env* e) { free(e); }

static void new_rng(bo_env* e, int64_t seed) { /* DeterministicRNG.__init__ :87-91 */
  if (seed == 0) seed = (int64_t)bo_mt_randbelow64(&e->grand, 4294967296ull); /* `seed or random.randint(0, 2**32-1)` */
  e->master_seed = seed;
  memset(e->stream_ready, 0, sizeof(e->stream_ready));
}

/* balatro_game.py:95-109 */
static void draw_cards(bo_env* e) {
  uint64_t inhand = 0;
  for (int i = 0; i < e->nhand; i++) inhand |= 1ull << e->hand[i];
  for (int d = 0; d < e->ndeck && e->nhand < e->hand_size && e->nhand < BO_MAX_HAND; d++) /* range(len(deck)): the foreign tail is out of reach */
    if (!(inhand & (1ull << d))) e->hand[e->nhand++] = d;
}

void bo_reset(bo_env* e, int has_seed, int64_t seed) { /* balatro_env_2.py:505-558 */
  if (has_seed) new_rng(e, seed);
  /* UnifiedGameState() :165-212 */
  e->ante = 1; e->round = 1; e->phase = BO_PHASE_BLIND_SELECT;
  e->chips_needed = 300; e->chips_scored = 0; e->round_chips_scored = 0; e->money = 4;
  e->nhand = 0; e->nsel = 0; e->hands_left = 4; e->discards_left = 3; e->hand_size = 8;
  e->njokers = 0; e->nconsumables = 0; e->n_magic_trick = 0; e->n_minimalist = 0;
  e->joker_slots = 5; e->consumable_slots = 2; e->shop_reroll_cost_state = 5;
  e->hands_played_total = 0; e->hands_played_ante = 0; e->best_hand_this_ante = 0; e->jokers_sold = 0;
  memset(e->enh, 0, 52); memset(e->edi, 0, 52); memset(e->seal, 0, 52);
  e->boss_type = 0; e->boss_played_types = 0; e->boss_played_cards = 0; e->boss_first_hand = 1;
  e->boss_hands_played = 0; e->boss_cards_required = 5; e->face_down = 0;
  /* ScoreEngine() scoring_engine.py:64-72 */
  for (int i = 0; i < 12; i++) { e->hand_levels[i] = 1; e->obs_levels[i] = 1; e->play_counts[i] = 0; }
  /* deck :519-525: for suit in Suit: for rank in Rank */
  int p = 0;
  for (int s = 0; s < 4; s++) for (int r = 2; r <= 14; r++) e->deck[p++] = (uint8_t)((r - 2) * 4 + s);
  bo_mt_shuffle_u8(stream(e, ST_DECK), e->deck, 52);
  e->ndeck = 52; e->nforeign = 0;
  /* BalatroGame() balatro_game.py:16-28 */
  e->highlighted = 0;
  /* self.shop is NOT cleared by reset; it is unobservable outside SHOP phase. */
}

void bo_construct(bo_env* e, int64_t seed) { /* BalatroEnv.__init__ :359-384 under the per-env global stream */
  bo_mt_seed(&e->grand, bo_global_seed(seed));
  e->shop_exists = 0;
  new_rng(e, seed);
  bo_reset(e, 0, 0);
}

void bo_set_jokers(bo_env* e, const int32_t* ids, int n) {
  e->njokers = n > BO_MAX_JOKERS ? BO_MAX_JOKERS : n;
  for (int i = 0; i < e->njokers; i++) e->jokers[i] = ids[i];
}
void bo_set_card_state(bo_env* e, int idx, int enh, int edi, int seal) { e->enh[idx] = (uint8_t)enh; e->edi[idx] = (uint8_t)edi; e->seal[idx] = (uint8_t)seal; }
/* harness convention: jokers re-injected after every reset of a batched rollout (bench config 3) */
void bo_set_template_jokers(bo_env* e, const int32_t* ids, int n) {
  e->tmpl_njokers = n > 5 ? 5 : n;
  for (int i = 0; i < e->tmpl_njokers; i++) e->tmpl_jokers[i] = ids[i];
  bo_set_jokers(e, ids, e->tmpl_njokers);
}
/* harness injection: state.consumables = [names of ids]; kept as the template bo_rollout re-applies after every reset */
void bo_set_consumables(bo_env* e, const int32_t* ids, int n) {
  e->tmpl_ncons = n > 2 ? 2 : (n < 0 ? 0 : n);
  for (int i = 0; i < e->tmpl_ncons; i++) e->tmpl_cons[i] = ids[i];
  e->nconsumables = e->tmpl_ncons;
  for (int i = 0; i < e->tmpl_ncons; i++) e->consumables[i] = ids[i];
}
/* harness injection: state.deck / game.deck (one aliased list, balatro_env_2.py:528-531) given a new order; the hand keeps its deck INDEXES */
void bo_set_deck(bo_env* e, const uint8_t* codes52) { memcpy(e->deck, codes52, 52); }
void bo_set_max_ante(bo_env* e, int max_ante) { e->max_ante = max_ante; } /* CurriculumBalatroEnv.current_max_ante (train_balatro_agent.py:129-166) */
void bo_set_money(bo_env* e, int64_t money) { e->money = money; }
void bo_set_ante(bo_env* e, int ante) { e->ante = ante; }
void bo_set_hand_level(bo_env* e, int ht, int level) { e->hand_levels[ht] = (uint8_t)(level < 1 ? 1 : level > 15 ? 15 : level); e->obs_levels[ht] = e->hand_levels[ht]; }

void bo_action_mask(const bo_env* e, int8_t* mask) { /* balatro_env_2.py:1426-1471 */
  memset(mask, 0, BO_NACT);
  if (e->phase == BO_PHASE_PLAY) {
    int n = e->nhand < 8 ? e->nhand : 8;
    for (int i = 0; i < n; i++) mask[2 + i] = 1;
    if (e->nsel > 0) mask[0] = 1;
    if (e->nsel > 0 && e->discards_left > 0) mask[1] = 1;
    for (int i = 0; i < e->nconsumables; i++) mask[10 + i] = 1;
  } else if (e->phase == BO_PHASE_SHOP) {
    if (e->shop_exists) {
      for (int i = 0; i < e->shop_n; i++) if (e->money >= e->shop_items[i].cost) mask[20 + i] = 1;
      if (e->money >= e->shop_reroll_cost_state) mask[30] = 1;
    }
    mask[31] = 1;
    for (int i = 0; i < e->njokers; i++) mask[32 + i] = 1;
  } else if (e->phase == BO_PHASE_BLIND_SELECT) {
    mask[45] = mask[46] = mask[47] = mask[48] = 1;
  }
}

void bo_get_obs(const bo_env* e, bo_obs* o) { /* balatro_env_2.py:1473-1541 */
  memset(o, 0, sizeof(*o));
  for (int i = 0; i < 8; i++) o->hand[i] = -1;
  for (int i = 0; i < e->nhand && i < 8; i++) if (e->hand[i] < e->ndeck) o->hand[i] = (int8_t)e->deck[e->hand[i]];
  o->hand_size = (int8_t)e->nhand;
  o->deck_size = (int8_t)(e->ndeck + e->nforeign); /* sum(1 for _ in state.deck) :1491 */
  for (int i = 0; i < e->nsel; i++) if (e->sel[i] < 8) o->selected_cards[e->sel[i]] = 1;
  o->chips_scored = e->chips_scored;
  o->round_chips_scored = (int32_t)e->round_chips_scored;
  int64_t need = e->chips_needed > 1 ? e->chips_needed : 1;
  double pr = (double)e->round_chips_scored / (double)need;
  o->progress_ratio = (float)(pr < 2.0 ? pr : 2.0);
  o->mult = 1;
  o->chips_needed = (int32_t)e->chips_needed;
  o->money = (int32_t)e->money;
  o->ante = (int16_t)e->ante;
  o->round = (int8_t)e->round;
  o->hands_left = (int8_t)e->hands_left;
  o->discards_left = (int8_t)e->discards_left;
  o->joker_count = (int8_t)e->njokers;
  for (int i = 0; i < e->njokers && i < 10; i++) o->joker_ids[i] = (int16_t)e->jokers[i];
  o->joker_slots = (int8_t)e->joker_slots;
  o->consumable_count = (int8_t)e->nconsumables;
  for (int i = 0; i < e->nconsumables && i < 5; i++) o->consumables[i] = (int16_t)((e->consumables[i] & 0x80) ? 0 : e->consumables[i]); /* enum-form names map to 0 (:1570) */
  o->consumable_slots = (int8_t)e->consumable_slots;
  o->shop_rerolls = (int16_t)e->shop_reroll_cost_state;
  for (int i = 0; i < 12; i++) o->hand_levels[i] = (int8_t)e->obs_levels[i];
  o->phase = (int8_t)e->phase;
  bo_action_mask(e, o->action_mask);
  o->hands_played = (int32_t)e->hands_played_total;
  o->best_hand_this_ante = (int32_t)e->best_hand_this_ante;
  o->boss_blind_active = e->boss_type ? 1 : 0;
  o->boss_blind_type = (int8_t)e->boss_type;
  for (int i = 0; i < 8; i++) o->face_down_cards[i] = (e->face_down >> i) & 1;
  if (e->phase == BO_PHASE_SHOP && e->shop_exists) /* :1534-1539 */
    for (int i = 0; i < e->shop_n && i < 10; i++) {
      o->shop_items[i] = e->shop_items[i].type;
      o->shop_costs[i] = (int16_t)e->shop_items[i].cost;
    }
}

/* ---- shop.py ---- */
enum { IT_PACK = 1, IT_CARD = 2, IT_JOKER = 3, IT_VOUCHER = 4 };
enum { PK_STANDARD = 0, PK_JOKER = 1, PK_TAROT = 2, PK_PLANET = 3, PK_SPECTRAL = 4 };

static double shop_cost_mult(const bo_env* e) { /* shop.py:104-108 */
  int k = e->shop_ante - 1;
  if (k < 0) k = 0;
  if (k > 100) k = 100;
  double m = BO_POW115[k];
  if (e->n_magic_trick > 0) m *= 0.9;
  return m;
}

static void shop_generate_inventory(bo_env* e) { /* shop.py:111-139 */
  static const int PACK_COST[5] = {250, 500, 600, 900, 1300};
  bo_mt* r = &e->shop_rng;
  double mult = shop_cost_mult(e);
  int n = 0;
  int third = PK_TAROT + (int)bo_mt_randbelow(r, 3); /* rng.choice([...]) evaluated before the loop */
  int packs[3] = {PK_STANDARD, PK_JOKER, third};
  for (int i = 0; i < 3; i++) {
    e->shop_items[n].type = IT_PACK; e->shop_items[n].payload = (uint8_t)packs[i];
    e->shop_items[n].cost = (int32_t)((double)PACK_COST[packs[i]] * mult);
    n++;
  }
  /* candid = library order, base_cost > 0, not owned (:123) */
  int candid[150], nc = 0;
  for (int id = 1; id <= 150; id++) {
    if (BO_JOKER_COST[id] == 0) continue;
    int owned = 0;
    for (int j = 0; j < e->njokers; j++) owned |= e->jokers[j] == id;
    if (!owned) candid[nc++] = id;
  }
  int k = nc < 3 ? nc : 3;
  /* random.sample(candid, k): n > 21 -> set method; n <= 21 -> pool method (Lib/random.py) */
  int picked[3];
  if (nc <= 21) {
    int pool[21];
    for (int i = 0; i < nc; i++) pool[i] = i;
    for (int i = 0; i < k; i++) {
      int j = (int)bo_mt_randbelow(r, (uint32_t)(nc - i));
      picked[i] = pool[j];
      pool[j] = pool[nc - i - 1];
    }
  } else {
    for (int i = 0; i < k; i++) {
      int j, dup;
      do {
        j = (int)bo_mt_randbelow(r, (uint32_t)nc);
        dup = 0;
        for (int q = 0; q < i; q++) dup |= picked[q] == j;
      } while (dup);
      picked[i] = j;
    }
  }
  for (int i = 0; i < k; i++) {
    int id = candid[picked[i]];
    e->shop_items[n].type = IT_JOKER; e->shop_items[n].payload = (uint8_t)id;
    e->shop_items[n].cost = (int32_t)((double)BO_JOKER_COST[id] * mult);
    n++;
  }
  int v = (int)bo_mt_randbelow(r, 2); /* 0 Magic Trick 600, 1 Minimalist 750 */
  e->shop_items[n].type = IT_VOUCHER; e->shop_items[n].payload = (uint8_t)v;
  e->shop_items[n].cost = (int32_t)((double)(v ? 750 : 600) * mult);
  n++;
  for (int i = 0; i < 2; i++) {
    int c = (int)bo_mt_randbelow(r, 52); /* randint(0, 51) */
    e->shop_items[n].type = IT_CARD; e->shop_items[n].payload = (uint8_t)c; e->shop_items[n].cost = 40;
    n++;
  }
  e->shop_n = n;
}

static void generate_shop(bo_env* e) { /* balatro_env_2.py:1383-1392 */
  uint32_t seed = bo_mt_randbelow(stream(e, ST_SHOP_GEN), 2147483648u); /* get_int(0, 2**31-1) */
  e->shop_exists = 1;
  e->shop_ante = e->ante;
  bo_mt_seed(&e->shop_rng, seed);
  e->shop_reroll_base = 50;
  shop_generate_inventory(e);
  e->shop_reroll_cost_state = (int64_t)((double)e->shop_reroll_base * shop_cost_mult(e));
}

static void advance_round(bo_env* e) { /* balatro_env_2.py:1326-1381 */
  /* end_of_round_effects() returns [] (complete_joker_effects.py:252-259) */
  int64_t gold = 0;
  for (int i = 0; i < e->nhand; i++) if (e->enh[e->hand[i]] == 7) gold += 3; /* GOLD */
  e->money += gold;
  if (e->boss_type) { /* :1346-1352 */
    e->money += 5;
    e->boss_type = 0; e->boss_played_types = 0; e->boss_played_cards = 0;
    e->face_down = 0;
  }
  e->round_chips_scored = 0; e->best_hand_this_ante = 0; e->hands_played_ante = 0;
  if (e->round == 3) {
    e->ante += 1; e->round = 1;
    if (e->ante > 100) return; /* :1366-1367 */
  } else e->round += 1;
  e->money += 25 * e->round + (e->round == 3 ? 10 : 0);
  e->hands_left = 4; e->discards_left = 3;
  e->phase = BO_PHASE_SHOP;
  generate_shop(e);
}

/* boss_blinds.py:380-407 */
static int boss_can_play(const bo_env* e, int ncards, int hand_type) {
  switch (e->boss_type) {
    case 7: if (ncards != 5) return BO_ERR_PSYCHIC; break;
    case 12: if (e->boss_played_types & (1u << hand_type)) return BO_ERR_EYE; break;
    case 13: if (e->boss_played_types && !(e->boss_played_types & (1u << hand_type))) return BO_ERR_MOUTH; break;
    case 25: if (ncards < e->boss_cards_required) return BO_ERR_VERDANT; break;
  }
  return 0;
}

/* boss_blinds.py:409-445 + 447-478: played cards are Card dataclasses, so suit (IntEnum vs str) debuffs never
 * fire; Plant checks rank; Violet all; Pillar id(card) -> deck index. */
static void boss_modify(const bo_env* e, int64_t bc, int64_t bm, const int* deck_idx, int n, int64_t* oc, int64_t* om) {
  int64_t chips = bc, mult = bm;
  if (e->boss_type == 21) { chips = chips / 2; mult = mult / 2; }          /* Flint */
  else if (e->boss_type == 22) { chips = 0; }                              /* Oxide */
  else if (e->boss_type == 23) { chips = (int64_t)((double)chips * 0.75); mult = (int64_t)((double)mult * 0.75); } /* Arm */
  int deb = 0;
  for (int i = 0; i < n; i++) {
    int rank = (e->deck[deck_idx[i]] >> 2) + 2;
    if (e->boss_type == 14 && rank >= 11 && rank <= 13) deb++;              /* Plant */
    else if (e->boss_type == 24) deb++;                                     /* Violet */
    else if (e->boss_type == 16 && (e->boss_played_cards >> deck_idx[i]) & 1) deb++; /* Pillar */
  }
  if (deb > 0) {
    double pen = BO_POW08[deb > 8 ? 8 : deb];
    chips = (int64_t)((double)chips * pen);
    mult = (int64_t)((double)mult * pen);
  }
  *oc = chips; *om = mult;
}

/* boss_blinds.py:343-378 on_hand_drawn, applied as in balatro_env_2.py:936-948 */
static void boss_on_hand_drawn(bo_env* e) {
  uint32_t fd = 0;
  int hook[2], nhook = 0;
  int n = e->nhand;
  switch (e->boss_type) {
    case 1: /* Hook: random.sample(range(n), 2) -> pool method */
      if (n >= 2) {
        int pool[BO_MAX_HAND];
        for (int i = 0; i < n; i++) pool[i] = i;
        for (int i = 0; i < 2; i++) {
          int j = (int)bo_mt_randbelow(&e->grand, (uint32_t)(n - i));
          hook[i] = pool[j];
          pool[j] = pool[n - i - 1];
        }
        nhook = 2;
      }
      break;
    case 3: /* Wheel */
      for (int i = 0; i < n; i++) if (bo_mt_random(&e->grand) < 1.0 / 7.0) fd |= 1u << i;
      break;
    case 4: /* House */
      if (e->boss_first_hand) fd = (1u << n) - 1;
      break;
    case 5: /* Mark */
      for (int i = 0; i < n; i++) { int r = (e->deck[e->hand[i]] >> 2) + 2; if (r >= 11 && r <= 13) fd |= 1u << i; }
      break;
    case 6: /* Fish */
      if (!e->boss_first_hand) fd = (1u << n) - 1;
      break;
  }
  e->face_down = fd; /* :941-942 ('face_down_cards' is always present) */
  if (nhook == 2) { /* :945-948 pop in descending position order */
    int a = hook[0] > hook[1] ? hook[0] : hook[1], b = hook[0] > hook[1] ? hook[1] : hook[0];
    int order[2] = {a, b};
    for (int q = 0; q < 2; q++) {
      int idx = order[q];
      if (idx < e->nhand) {
        for (int i = idx; i + 1 < e->nhand; i++) e->hand[i] = e->hand[i + 1];
        e->nhand--;
      }
    }
  }
}

static int joker_owned(const bo_env* e, int id) {
  for (int i = 0; i < e->njokers; i++) if (e->jokers[i] == id) return 1;
  return 0;
}

static void step_play_hand(bo_env* e, double* reward, uint8_t* terminated, bo_info* info) {
  /* :650-660 selected cards in selection order */
  int didx[8], n = 0;
  bo_scard sc[8];
  for (int i = 0; i < e->nsel; i++) {
    int pos = e->sel[i];
    if (pos < e->nhand) {
      int ci = e->hand[pos];
      if (ci < e->ndeck) {
        int code = e->deck[ci], rank = (code >> 2) + 2, suit = code & 3;
        int chips = bo_rank_base_chips(rank);            /* CardAdapter.to_scoring_format :287-325 */
        if (e->enh[ci] == 1) chips += 30;                /* BONUS  cards.py:122-123 */
        else if (e->enh[ci] == 6) chips += 50;           /* STONE */
        if (e->edi[ci] == 1) chips += 50;                /* FOIL   cards.py:188-189 */
        if (e->enh[ci] == 6) { rank = 0; suit = SUIT_STONE; }
        didx[n] = ci;
        sc[n].rank = rank; sc[n].suit = suit; sc[n].chips = chips;
        n++;
      }
    }
  }
  /* :663-666 highlight (never cleared by a play) */
  for (int i = 0; i < e->nsel; i++) if (e->sel[i] < e->nhand) e->highlighted |= 1u << e->sel[i];
  /* :669-671 classify deck[p] for highlighted POSITIONS p */
  uint8_t hc[32];
  int nh = 0;
  for (int p = 0; p < 32; p++) if (e->highlighted & (1u << p)) hc[nh++] = e->deck[p];
  int hand_type = bo_classify(hc, nh);
  /* :677-680 */
  if (e->boss_type) {
    int err = boss_can_play(e, n, hand_type);
    if (err) {
      *reward = -1.0; info->error = err;
      /* what the reference's message names: f"Cannot play {hand_type} again" / f"Can only play {allowed}" (the set holds ONE type once
       * The Mouth has seen a play) / f"Must play at least {required} cards" */
      if (err == BO_ERR_EYE) info->aux = hand_type;
      else if (err == BO_ERR_MOUTH) { int a = 0; while (a < 12 && !(e->boss_played_types & (1u << a))) a++; info->aux = a; }
      else if (err == BO_ERR_VERDANT) info->aux = e->boss_cards_required;
      return;
    }
  }
  /* :683-692 */
  bo_score_out so;
  int nj = (e->flags & BO_FLAG_SCORER_JOKERS) ? e->njokers : 0; /* dict jokers are skipped */
  bo_score_hand(sc, n, sc, n, hand_type, BO_NAMES_ENV, e->hand_levels[hand_type], e->jokers, nj, e->hands_left,
                e->discards_left, e->ndeck + e->nforeign, &e->grand, &so);
  int64_t base_score = so.score;
  { /* info['score_breakdown'] (:909): the scorer's own numbers, before the env's steel / boss / red-seal factors */
    int64_t bc0, bm0;
    int card_chips = 0;
    for (int i = 0; i < n; i++) card_chips += sc[i].chips;
    bo_hand_chips_mult(hand_type, e->hand_levels[hand_type], &bc0, &bm0);
    info->breakdown[0] = (double)so.chips; info->breakdown[1] = (double)so.mult; info->breakdown[2] = so.x_mult;
    info->breakdown[3] = (double)card_chips; info->breakdown[4] = (double)bc0; info->breakdown[5] = (double)bm0;
    info->breakdown[6] = (double)so.money; info->breakdown[7] = 0.0;
  }
  /* :703-734 per-card enhancement / seal effects */
  int64_t extra_money = 0;
  int retriggers = 0, blue = 0;
  for (int i = 0; i < n; i++) {
    int ci = didx[i];
    if (e->enh[ci] == 4) { (void)bo_mt_random(stream(e, ST_CARD_ENH)); }        /* GLASS: roll only */
    else if (e->enh[ci] == 8) {                                                /* LUCKY */
      (void)bo_mt_random(stream(e, ST_CARD_ENH));
      double money_roll = bo_mt_random(stream(e, ST_CARD_ENH));
      if (money_roll < 0.0667) extra_money += 20;
    }
    if (e->seal[ci] == 1) extra_money += 3;                                     /* GOLD seal */
    else if (e->seal[ci] == 2) retriggers++;                                    /* RED seal */
    else if (e->seal[ci] == 3) blue++;                                          /* BLUE seal -> planet */
  }
  /* :732-734 each blue seal passes the guard against the UN-GROWN consumable list, then :765-767 appends
   * under the same guard (state.consumables holds names; ids per _get_consumable_ids :1545-1567) */
  if (blue && e->nconsumables < e->consumable_slots) {
    static const int PLANET_ID[12] = {38, 30, 31, 32, 33, 34, 35, 36, 37, 39, 40, 41};
    for (int q = 0; q < blue; q++)
      if (e->nconsumables < e->consumable_slots && e->nconsumables < 5) e->consumables[e->nconsumables++] = PLANET_ID[hand_type];
  }
  int64_t final_score = base_score;
  /* :741-742 steel held in hand (not selected) */
  double steel = 1.0;
  for (int i = 0; i < e->nhand; i++) {
    int ci = e->hand[i], selected = 0;
    for (int q = 0; q < e->nsel; q++) if (e->sel[q] < e->nhand && e->hand[e->sel[q]] == ci) selected = 1;
    if (!selected && e->enh[ci] == 5) steel *= 1.5;
  }
  final_score = (int64_t)((double)final_score * steel);
  /* :745-755 boss ratio */
  if (e->boss_type) {
    int64_t bc, bm, mc, mm;
    bo_hand_chips_mult(hand_type, e->hand_levels[hand_type], &bc, &bm);
    boss_modify(e, bc, bm, didx, n, &mc, &mm);
    if (bc > 0 && bm > 0) {
      double cr = (double)mc / (double)bc, mr = (double)mm / (double)bm;
      final_score = (int64_t)((double)final_score * cr * mr);
    }
  }
  /* :758-759 */
  double retrigger_bonus = (double)retriggers * 0.5;
  final_score = (int64_t)((double)final_score * (1.0 + retrigger_bonus));
  e->money += extra_money;
  /* :775-786 */
  int64_t old_round = e->round_chips_scored;
  int64_t need1 = e->chips_needed > 1 ? e->chips_needed : 1;
  double old_progress = (double)old_round / (double)need1;
  if (old_progress > 1.0) old_progress = 1.0;
  e->round_chips_scored += final_score;
  e->chips_scored += final_score;
  e->hands_played_total++; e->hands_played_ante++;
  if (final_score > e->best_hand_this_ante) e->best_hand_this_ante = final_score;
  e->play_counts[hand_type]++;
  /* :789-794 boss on_hand_scored (boss_blinds.py:480-507) */
  if (e->boss_type) {
    e->boss_played_types |= 1u << hand_type;
    e->boss_first_hand = 0;
    e->boss_hands_played++;
    if (e->boss_type == 16) for (int i = 0; i < n; i++) e->boss_played_cards |= 1ull << didx[i];
    if (e->boss_type == 25) e->boss_cards_required = e->boss_cards_required + 1 > 7 ? 7 : e->boss_cards_required + 1;
    /* Tooth / Serpent mutate a throw-away dict: no state effect */
  }
  e->nsel = 0; /* :797 */
  /* :799-892 reward shaping */
  double new_progress = (double)e->round_chips_scored / (double)need1;
  if (new_progress > 1.0) new_progress = 1.0;
  double progress_reward = 15.0 * new_progress;
  double milestone = 0.0;
  if (old_progress < 0.25 && 0.25 <= new_progress) milestone = 5.0;
  else if (old_progress < 0.5 && 0.5 <= new_progress) milestone = 10.0;
  else if (old_progress < 0.75 && 0.75 <= new_progress) milestone = 15.0;
  else if (old_progress < 1.0 && 1.0 <= new_progress) milestone = 25.0;
  double score_reward;
  if (e->ante <= 3) {
    score_reward = (double)final_score / 100.0;
    if (score_reward > 10.0) score_reward = 10.0;
  } else {
    int64_t s = final_score > 1 ? final_score : 1;
    score_reward = s >= BO_LOG10_N ? 10.0 : 3.0 * BO_LOG10[s];
    if (score_reward > 10.0) score_reward = 10.0;
  }
  static const double HQ[12] = {0.1, 0.5, 1.0, 2.0, 2.5, 2.5, 3.5, 5.0, 7.0, 10.0, 0.0, 0.0};
  double hq = HQ[hand_type];
  double eff = 0.0;
  if (hand_type >= 3 && n <= 3) eff = 2.0;
  else if (hand_type >= 5 && n == 5) eff = 1.0;
  else if (n <= 4 && e->hands_left <= 2) eff = 1.5;
  double syn = 0.0;
  if (hand_type == 5 && (joker_owned(e, 113) || joker_owned(e, 18) || joker_owned(e, 69))) syn += 2.0;
  if (hand_type == 1 || hand_type == 2 || hand_type == 3)
    if (joker_owned(e, 40) || joker_owned(e, 39) || joker_owned(e, 6) || joker_owned(e, 7)) syn += 1.5;
  int faces = 0;
  for (int i = 0; i < n; i++) faces += ((e->deck[didx[i]] >> 2) + 2) >= 11;
  if (faces > 0 && (joker_owned(e, 33) || joker_owned(e, 104) || joker_owned(e, 42))) syn += 0.5 * (double)faces;
  double strat = 0.0;
  if (new_progress > 0.7 && e->hands_left >= 3) strat = 2.0;
  else if (new_progress < 0.3 && hand_type >= 5) strat = 3.0;
  double ante_bonus = 0.0;
  if (e->ante >= 4) { ante_bonus = (double)(e->ante - 3) * 0.5; if (ante_bonus > 5.0) ante_bonus = 5.0; }
  double r = progress_reward + milestone;
  r = r + score_reward;
  r = r + hq * 2.0;
  r = r + eff * 1.5;
  r = r + syn * 3.0;
  r = r + strat * 2.0;
  r = r + ante_bonus;
  if (r > 100.0) r = 100.0;
  info->reward_terms[0] = progress_reward; info->reward_terms[1] = milestone; info->reward_terms[2] = score_reward;
  info->reward_terms[3] = hq; info->reward_terms[4] = eff; info->reward_terms[5] = syn; info->reward_terms[6] = strat;
  info->reward_terms[7] = ante_bonus;
  info->final_score = final_score;
  info->hand_type = (int8_t)hand_type;
  info->cards_played = (int8_t)n;
  /* :914-960 outcome */
  if (e->round_chips_scored >= e->chips_needed) {
    double bonus = 25.0 + 10.0 * (double)e->ante;
    r += bonus < 50.0 ? bonus : 50.0;
    advance_round(e);
    info->flags |= BO_INFO_BEAT_BLIND;
  } else if (e->hands_left <= 1) {
    r += -50.0 * (1.0 - new_progress);
    *terminated = 1;
    info->flags |= BO_INFO_FAILED;
  } else {
    e->hands_left -= 1;
    draw_cards(e);
    if (e->boss_type) boss_on_hand_drawn(e);
  }
  *reward = r;
}

static void step_discard(bo_env* e, double* reward, bo_info* info) { /* balatro_env_2.py:962-1050 */
  (void)info;
  int ranks[8], n = 0, purple = 0;
  for (int i = 0; i < e->nsel; i++) {
    int pos = e->sel[i];
    if (pos < e->nhand && e->hand[pos] < e->ndeck) {
      int ci = e->hand[pos];
      if (e->seal[ci] == 4) purple++;
      ranks[n++] = (e->deck[ci] >> 2) + 2;
    }
  }
  int first = e->discards_left == 3; /* == game.discards (balatro_game.py:25) */
  int64_t money = 0;
  int ndj = 0;
  for (int j = 0; j < e->njokers; j++) { /* complete_joker_effects.py:186-209 */
    int id = e->jokers[j];
    if (id == 95) { if (first && n == 1) { money += 3; e->money += 3; } }
    else if (id == 57) {
      int f = 0;
      for (int i = 0; i < n; i++) f += ranks[i] >= 11 && ranks[i] <= 13;
      if (f >= 3) { money += 5; e->money += 5; }
    }
    if (id == 57 || id == 130 || id == 82 || id == 77) ndj++;
  }
  /* :1010-1018 highlight on top of stale highlights, then balatro_game.py:111-127 */
  for (int i = 0; i < e->nsel; i++) if (e->sel[i] < e->nhand) e->highlighted |= 1u << e->sel[i];
  {
    int keep[BO_MAX_HAND], k = 0, orig = e->nhand;
    for (int p = 0; p < orig; p++) if (!(e->highlighted & (1u << p))) keep[k++] = e->hand[p];
    for (int i = 0; i < k; i++) e->hand[i] = keep[i];
    e->nhand = k;
  }
  e->highlighted = 0;
  e->discards_left -= 1;
  draw_cards(e);
  e->nsel = 0;
  /* :1021-1032 purple seal -> tarot via 'seal_applications' */
  for (int q = 0; q < purple; q++)
    if (e->nconsumables < e->consumable_slots && e->nconsumables < 5)
      e->consumables[e->nconsumables++] = 1 + (int)bo_mt_randbelow(stream(e, ST_SEAL), 22);
  /* :1035-1050 */
  double r = 0.2;
  if (ndj) r += 0.5 * (double)ndj;
  if (money > 0) r += (double)money / 5.0;
  int64_t need1 = e->chips_needed > 1 ? e->chips_needed : 1;
  double progress = (double)e->round_chips_scored / (double)need1;
  if (progress < 0.5 && e->discards_left > 1) r += 0.5;
  else if (progress > 0.8 && e->discards_left > 1) r -= 0.3;
  *reward = r;
}

static void step_shop(bo_env* e, int action, double* reward, bo_info* info) { /* balatro_env_2.py:1174-1253 */
  if (action >= 32 && action < 37) { /* sell joker :1202-1215 */
    int ji = action - 32;
    int id = e->jokers[ji];
    for (int i = ji; i + 1 < e->njokers; i++) e->jokers[i] = e->jokers[i + 1];
    e->njokers--;
    int64_t v = BO_JOKER_COST[id] / 2;
    if (v < 3) v = 3;
    e->money += v;
    e->jokers_sold++;
    *reward = (double)v / 5.0;
    info->flags |= BO_INFO_SOLD_JOKER; info->aux = id;
    return;
  }
  if (action == 31) { /* SKIP: shop.py:166-167, then :1247-1251 */
    e->phase = BO_PHASE_PLAY;
    draw_cards(e);
    *reward = 0.0;
    return;
  }
  if (action == 30) { /* REROLL shop.py:170-177 */
    int64_t cost = (int64_t)((double)e->shop_reroll_base * shop_cost_mult(e));
    if (e->money < cost) { *reward = -1.0; info->error = BO_ERR_REROLL_FUNDS; return; }
    e->money -= cost;
    e->shop_reroll_base = (int64_t)((double)e->shop_reroll_base * 1.35);
    shop_generate_inventory(e);
    *reward = 0.0;
    return;
  }
  /* buy 20..29 (mask guarantees index < n and money >= cost) shop.py:179-203 */
  int idx = action - 20;
  bo_item it = e->shop_items[idx];
  e->money -= it.cost;
  for (int i = idx; i + 1 < e->shop_n; i++) e->shop_items[i] = e->shop_items[i + 1];
  e->shop_n--;
  if (it.type == IT_PACK) {
    int count = it.payload == PK_STANDARD ? 3 : 1; /* shop.py:150-157 */
    for (int i = 0; i < count; i++) { int c = (int)bo_mt_randbelow(&e->shop_rng, 52); if (i == 0) info->aux = c; }
    *reward = 5.0; info->flags |= BO_INFO_OPENED_PACK;
  } else if (it.type == IT_CARD) {
    *reward = 3.0; info->flags |= BO_INFO_BOUGHT_CARD;
  } else if (it.type == IT_JOKER) {
    if (e->njokers >= 5) { *reward = -1.0; info->error = BO_ERR_JOKER_SLOTS; return; } /* chips already gone */
    e->jokers[e->njokers++] = it.payload;
    *reward = 15.0; info->flags |= BO_INFO_BOUGHT_JOKER; info->aux = it.payload;
  } else { /* voucher */
    if (it.payload == 0) e->n_magic_trick++; else e->n_minimalist++;
    *reward = 10.0; info->flags |= BO_INFO_BOUGHT_VOUCHER; info->aux = it.payload;
  }
}

static void step_blind_select(bo_env* e, int action, double* reward, bo_info* info) { /* :1255-1318 */
  if (action >= 45 && action < 48) {
    int b = action - 45;
    e->round = b + 1;
    e->chips_needed = bo_blind_chips(e->ante, b);
    *reward = 0.0;
    if (b == 2) {
      int boss = 1 + (int)bo_mt_randbelow(&e->grand, 28); /* boss_blinds.py:522-532 random.choice(list(BossBlindType)) */
      e->boss_type = boss; /* activate_boss_blind boss_blinds.py:308-341 */
      e->boss_played_types = 0; e->boss_played_cards = 0; e->boss_first_hand = 1; e->boss_hands_played = 0;
      e->boss_cards_required = 5;
      double chip_mult = boss == 2 ? 2.0 : 1.0;
      e->chips_needed = (int64_t)((double)e->chips_needed * chip_mult);
      if (boss == 9) e->discards_left = 0;       /* Water */
      if (boss == 11) e->hand_size += -1;        /* Manacle */
      if (boss == 17) e->hands_left = 1;         /* Needle */
      info->aux = boss;
      *reward = 10.0;
    }
    e->phase = BO_PHASE_PLAY;
    draw_cards(e);
  } else { /* 48 SKIP_BLIND :1305-1316 */
    *reward = -5.0;
    advance_round(e);
    info->flags |= BO_INFO_SKIPPED_BLIND;
  }
}


/* ---------------------------------------------------------------------------------------------------------
 * _use_consumable (balatro_env_2.py:1066-1172) over ConsumableManager.use_consumable (consumables.py:622-652),
 * TarotEffects.apply_tarot (:111-327) and SpectralEffects.apply_spectral (:354-613).
 *
 * A consumable is held as its _get_consumable_ids id (:1545-1567: tarots 1-22, planets 30-41, spectrals 50-67);
 * bit 7 marks a name in enum form ('THE_FOOL', what The Emperor creates, consumables.py:168): it is used like the
 * natural name (TarotCard[name.upper().replace(' ', '_')], :631) but the observation shows id 0.
 *
 * Facts of the reference this follows (all reproduced by the golden trace `consumables`):
 *  - target cards are CLASSES made by CardAdapter.to_consumable_format (:328-343), so rank / suit edits (Strength, Death,
 *    Star, Moon, Sun, World) never reach the deck; only enhancement / edition / seal are copied back (:1122-1138), as the
 *    INTEGER values of consumables.py's enums (its Seal enum is RED 1, BLUE 2, GOLD 3, PURPLE 4 while cards.py, which
 *    the env compares against, has GOLD 1, RED 2, BLUE 3: Talisman's "gold" seal acts as a blue one, and so on);
 *  - to_dict()['consumables'] is the live list (:220): created items are appended by the effect AND again by :1157-1160;
 *    to_dict()['jokers'] is a fresh list, so joker edits (Ankh, Hex, Wraith's append) are lost and only :1147-1155 counts;
 *  - list.remove(target) raises ValueError (Hanged Man, Familiar, Grim, Incantation with a target) and cards.Card is a
 *    frozen dataclass (Sigil, Ouija raise after their random.choice): BO_ERR_CONSUMABLE_RAISES, reward -1.0 by harness
 *    convention, the state stays as the exception leaves it (nothing popped, selection kept);
 *  - Cryptid appends two consumables.Card copies to the live deck; draws take the lowest free deck index and cards never
 *    leave the hand for good, so the copies are only ever COUNTED (deck_size, Blue Joker); Immolate removes five sampled
 *    cards from the live deck list, so every later deck index (hand indexes, card_states keys) now names another card.
 * --------------------------------------------------------------------------------------------------------- */
static const int WRAITH_JOKER[14] = {137, 138, 139, 140, 0, 142, 143, 144, 145, 146, 147, 148, 149, 150}; /* consumables.py:474-476 by JOKER_LIBRARY name; 'Drivers License' is not a library name */
static const int SOUL_JOKER[5] = {146, 147, 148, 149, 150};                                               /* :590 */

static void use_consumable(bo_env* e, int ci, double* reward, bo_info* info) {
  if (ci >= e->nconsumables) { *reward = -1.0; info->error = BO_ERR_CONSUMABLE; return; } /* :1068-1069 (masked out before) */
  const int code = e->consumables[ci], id = code & 0x7f;
  /* :1074-1083 target cards in selection order */
  int tgt[8], nt = 0;
  for (int i = 0; i < e->nsel; i++) {
    int pos = e->sel[i];
    if (pos < e->nhand && e->hand[pos] < e->ndeck) tgt[nt++] = e->hand[pos];
  }
  bo_mt* g = &e->grand;
  int success = 0, raises = 0, unsupported = 0;
  int64_t money_gained = 0;
  int planet = -1;
  int aff[8], naff = 0, set_enh = -1, set_edi = -1, set_seal = -1; /* cards_affected + the attribute the effect changed */
  int items[4], nitems = 0, jcreated[2], njc = 0, hs_change = 0, ncreated = 0, ndestroyed = 0;
  const int slots = e->consumable_slots;
#define AFFECT_FIRST(n) do { for (int i_ = 0; i_ < nt && i_ < (n); i_++) aff[naff++] = tgt[i_]; } while (0)
  switch (id) {
    case 1: /* The Fool :127-134 */
      if (e->nconsumables > 0) {
        int c = e->consumables[bo_mt_randbelow(g, (uint32_t)e->nconsumables)];
        if (e->nconsumables < 5) e->consumables[e->nconsumables++] = c; /* live list, no slot check */
        items[nitems++] = c; success = 1;
      }
      break;
    case 2: case 4: case 6: /* Magician LUCKY :136-143, Empress MULT :157-164, Hierophant BONUS :177-184 */
      if (nt > 0) { AFFECT_FIRST(2); set_enh = id == 2 ? 8 : id == 4 ? 2 : 1; success = 1; }
      break;
    case 3: /* The High Priestess :145-155: choice first, slot check second */
      for (int k = 0; k < 2; k++) {
        int p = 30 + (int)bo_mt_randbelow(g, 9);
        if (e->nconsumables < slots) { e->consumables[e->nconsumables++] = p; items[nitems++] = p; }
      }
      success = 1;
      break;
    case 5: /* The Emperor :166-175: slot check first; names in enum form */
      for (int k = 0; k < 2; k++)
        if (e->nconsumables < slots) {
          int t = (1 + (int)bo_mt_randbelow(g, 22)) | 0x80;
          e->consumables[e->nconsumables++] = t; items[nitems++] = t;
        }
      success = 1;
      break;
    case 7: case 8: case 12: case 16: case 17: /* Lovers WILD, Chariot STEEL, Justice GLASS, Devil GOLD, Tower STONE */
      if (nt >= 1) { AFFECT_FIRST(1); set_enh = id == 7 ? 3 : id == 8 ? 5 : id == 12 ? 4 : id == 16 ? 7 : 6; success = 1; }
      break;
    case 9: /* Strength :202-210: only cards below the ace are listed; the rank edit is lost */
      if (nt > 0) { for (int i = 0; i < nt && i < 2; i++) if ((e->deck[tgt[i]] >> 2) + 2 < 14) aff[naff++] = tgt[i]; success = 1; }
      break;
    case 10: /* The Hermit :212-219 */
      money_gained = e->money < 20 ? e->money : 20; success = 1;
      break;
    case 11: /* Wheel of Fortune :221-231: `target_cards and random.random() < 0.25` */
      if (nt > 0 && bo_mt_random(g) < 0.25) { set_edi = 1 + (int)bo_mt_randbelow(g, 3); aff[naff++] = tgt[0]; success = 1; }
      break;
    case 13: /* The Hanged Man :241-251 */
      if (nt > 0) raises = 1;
      break;
    case 14: /* Death :253-261 */
      if (nt >= 2) { AFFECT_FIRST(2); success = 1; }
      break;
    case 15: /* Temperance :263-273 */
      money_gained = 5 * e->njokers < 50 ? 5 * e->njokers : 50; success = 1;
      break;
    case 18: case 19: case 20: case 22: /* Star / Moon / Sun / World :291-316,329-336: suit edits are lost */
      if (nt > 0) { AFFECT_FIRST(3); success = 1; }
      break;
    case 21: { /* Judgement :318-327 */
      int p = 30 + (int)bo_mt_randbelow(g, 9);
      if (e->nconsumables < slots) { e->consumables[e->nconsumables++] = p; items[nitems++] = p; }
      success = 1;
      break;
    }
    case 50: case 51: case 52: /* Familiar / Grim / Incantation :373-457: deck.remove(target class) */
      if (nt >= 1) raises = 1;
      break;
    case 53: case 61: case 63: case 64: /* Talisman GOLD=3, Deja Vu RED=1, Trance BLUE=2, Medium PURPLE=4 (consumables.Seal values) */
      if (nt >= 1) { AFFECT_FIRST(1); set_seal = id == 53 ? 3 : id == 61 ? 1 : id == 63 ? 2 : 4; success = 1; }
      break;
    case 54: /* Aura :467-474 */
      if (nt >= 1) { set_edi = 1 + (int)bo_mt_randbelow(g, 3); aff[naff++] = tgt[0]; success = 1; }
      break;
    case 55: /* Wraith :476-488 */
      if (e->njokers < e->joker_slots) { jcreated[njc++] = WRAITH_JOKER[bo_mt_randbelow(g, 14)]; hs_change = -1; success = 1; }
      break;
    case 56: case 57: { /* Sigil :490-498, Ouija :500-509: random.choice, then assignment to a frozen dataclass */
      int n = 0;
      for (int i = 0; i < e->nhand; i++) n += e->hand[i] < e->ndeck;
      if (n > 0) { bo_mt_randbelow(g, id == 56 ? 4 : 13); raises = 1; }
      break;
    }
    case 58: /* Ectoplasm :511-517 */
      if (e->njokers > 0) { hs_change = -1; success = 1; }
      break;
    case 59: { /* Immolate :519-531: random.sample(deck, 5) by index (Lib/random.py sample(): selection set for n > 21, pool
                * below), then deck.remove(card) for each: every later index shifts down, card_states keep their keys */
      int n = e->ndeck + e->nforeign, k = n < 5 ? n : 5, picked[5];
      if (e->ndeck < 13) { unsupported = 1; break; } /* restated while the hand's indexes (always 0..7) stay valid (8 real cards or more behind
                                                      * this use); below that the reference's unguarded deck[i] reads (:577,670,937) would raise */
      if (n <= 21) {
        int pool[21];
        for (int i = 0; i < n; i++) pool[i] = i;
        for (int i = 0; i < k; i++) { int j = (int)bo_mt_randbelow(g, (uint32_t)(n - i)); picked[i] = pool[j]; pool[j] = pool[n - i - 1]; }
      } else
        for (int i = 0; i < k; i++) {
          int j, dup;
          do { j = (int)bo_mt_randbelow(g, (uint32_t)n); dup = 0; for (int q = 0; q < i; q++) dup |= picked[q] == j; } while (dup);
          picked[i] = j;
        }
      uint64_t gone = 0;
      for (int i = 0; i < k; i++) { if (picked[i] < e->ndeck) gone |= 1ull << picked[i]; else e->nforeign--; } /* copies are a suffix of equal-class objects */
      int w = 0;
      uint64_t played = 0; /* The Pillar remembers card OBJECTS (id(card), boss_blinds.py:472): its marks move with the cards */
      for (int i = 0; i < e->ndeck; i++)
        if (!((gone >> i) & 1)) { if ((e->boss_played_cards >> i) & 1) played |= 1ull << w; e->deck[w++] = e->deck[i]; }
      e->ndeck = w; e->boss_played_cards = played;
      ndestroyed = k; money_gained = 20; success = 1;
      break;
    }
    case 60: /* Ankh :533-543: the "name" is a {'name','id'} dict unless the scorer-level harness hands out names */
      if (e->njokers > 0) {
        int k = (int)bo_mt_randbelow(g, (uint32_t)e->njokers);
        jcreated[njc++] = (e->flags & BO_FLAG_SCORER_JOKERS) ? e->jokers[k] : 0;
        success = 1;
      }
      break;
    case 62: /* Hex :553-563 */
      if (e->njokers > 0) { bo_mt_randbelow(g, (uint32_t)e->njokers); success = 1; }
      break;
    case 65: /* Cryptid :581-591: two consumables.Card copies appended to the live deck */
      if (nt >= 1) { /* deck_size = np.int8(len(deck)) (balatro_env_2.py:1491): 128 cards raise OverflowError under numpy >= 2 and wrap under numpy 1 */
        if (e->ndeck + e->nforeign + 2 > 127) { unsupported = 1; break; }
        e->nforeign += 2; ncreated = 2; success = 1;
      }
      break;
    case 66: /* The Soul :593-601 */
      if (e->njokers < e->joker_slots) { jcreated[njc++] = SOUL_JOKER[bo_mt_randbelow(g, 5)]; success = 1; }
      break;
    case 67: /* Black Hole :603-610 */
      success = 1;
      break;
    default:
      if (id >= 30 && id <= 41) { planet = id - 30; success = 1; } /* :644-652 */
      break; /* unknown name :654 */
  }
#undef AFFECT_FIRST
  if (raises) { *reward = -1.0; info->error = BO_ERR_CONSUMABLE_RAISES; return; }
  if (unsupported) { *reward = -1.0; info->error = BO_ERR_CONSUMABLE_DECK; return; }
  if (success) { /* :1093-1164 */
    double r = 0.0;
    for (int i = ci; i + 1 < e->nconsumables; i++) e->consumables[i] = e->consumables[i + 1]; /* pop(consumable_idx) */
    e->nconsumables--;
    if (money_gained > 0) { e->money += money_gained; r += (double)money_gained / 10.0; }
    if (planet >= 0) {
      static const int PLANET_HT[12] = {1, 2, 3, 4, 5, 6, 7, 8, 0, 9, 10, 11}; /* Mercury..Eris :1103-1116 */
      int ht = PLANET_HT[planet];
      if (e->hand_levels[ht] < 15) e->hand_levels[ht]++;  /* engine.apply_planet scoring_engine.py:82-85 */
      e->obs_levels[ht]++;                                 /* state.hand_levels[...] += 1 (uncapped) :1119 */
      r += 10.0;
    }
    if (naff) {
      for (int i = 0; i < naff; i++) {
        if (set_enh >= 0) e->enh[aff[i]] = (uint8_t)set_enh;
        if (set_edi >= 0) e->edi[aff[i]] = (uint8_t)set_edi;
        if (set_seal >= 0) e->seal[aff[i]] = (uint8_t)set_seal;
      }
      r += (double)naff * 2.0;
    }
    if (ncreated) r += (double)ncreated * 3.0;     /* :1140-1141 */
    if (ndestroyed) r += (double)ndestroyed * 1.0; /* :1143-1144 */
    if (njc) {
      for (int i = 0; i < njc; i++)
        if (e->njokers < e->joker_slots && jcreated[i] > 0 && e->njokers < BO_MAX_JOKERS) e->jokers[e->njokers++] = jcreated[i];
      r += (double)njc * 15.0;
    }
    if (nitems) {
      for (int i = 0; i < nitems; i++)
        if (e->nconsumables < slots && e->nconsumables < 5) e->consumables[e->nconsumables++] = items[i];
      r += (double)nitems * 5.0;
    }
    if (hs_change) e->hand_size += hs_change;
    *reward = r;
  } else { *reward = -1.0; info->error = BO_ERR_CONSUMABLE; } /* :1166-1168 */
  e->nsel = 0; /* :1171 */
}

void bo_step(bo_env* e, int action, double* reward, uint8_t* terminated, bo_info* info) {
  bo_info local;
  if (!info) info = &local;
  memset(info, 0, sizeof(*info));
  info->hand_type = -1;
  *reward = 0.0;
  *terminated = 0;
  int8_t mask[BO_NACT];
  if (e->ante > 100) { *terminated = 1; info->error = BO_ERR_MAX_ANTE; }                          /* :619-620 */
  else if (e->chips_scored > 1000000000ll) { *terminated = 1; info->error = BO_ERR_MAX_SCORE; }  /* :622-623 */
  else if (bo_action_mask(e, mask), (action < 0 || action >= BO_NACT || !mask[action])) { *reward = -1.0; info->error = BO_ERR_INVALID_ACTION; }
  else if (e->phase == BO_PHASE_PLAY) {
    if (action == 0) step_play_hand(e, reward, terminated, info);
    else if (action == 1) step_discard(e, reward, info);
    else if (action >= 2 && action < 10) { /* :1052-1058 toggle, selection ORDER is kept */
      int pos = action - 2;
      if (pos < e->nhand) {
        int at = -1;
        for (int i = 0; i < e->nsel; i++) if (e->sel[i] == pos) at = i;
        if (at >= 0) { for (int i = at; i + 1 < e->nsel; i++) e->sel[i] = e->sel[i + 1]; e->nsel--; }
        else e->sel[e->nsel++] = pos;
      }
    } else use_consumable(e, action - 10, reward, info); /* 10..14 :1066-1172 */
  } else if (e->phase == BO_PHASE_SHOP) step_shop(e, action, reward, info);
  else if (e->phase == BO_PHASE_BLIND_SELECT) step_blind_select(e, action, reward, info);
  /* CurriculumBalatroEnv.step (train_balatro_agent.py:146-152) wraps EVERY step, also the guarded ones */
  if (e->max_ante > 0 && e->ante > e->max_ante) { *terminated = 1; info->flags |= BO_INFO_CURRICULUM; }
}

int bo_policy_action(const bo_env* e, int policy, uint64_t policy_seed, uint64_t env_index, uint64_t t) {
  if (policy != BO_POLICY_UNIFORM) {
    if (e->phase == BO_PHASE_BLIND_SELECT) return policy == BO_POLICY_CYCLE3 ? 45 + (int)(env_index % 3) : 45;
    if (e->phase == BO_PHASE_SHOP) return 31;
  }
  int8_t mask[BO_NACT];
  bo_action_mask(e, mask);
  int nv = 0;
  for (int a = 0; a < BO_NACT; a++) nv += mask[a];
  if (!nv) return 0;
  uint32_t k = bo_policy_hash(policy_seed, env_index, t) % (uint32_t)nv;
  for (int a = 0; a < BO_NACT; a++) if (mask[a]) { if (!k) return a; k--; }
  return 0;
}

int64_t bo_rollout(bo_env** envs, int n, int64_t env_index0, int T, int policy, uint64_t policy_seed, uint64_t t0,
                   double* reward_sum, int64_t* score_sum, int64_t* episodes) {
  int64_t steps = 0, eps = 0, sc = 0;
  double rs = 0.0;
  for (int i = 0; i < n; i++) {
    bo_env* e = envs[i];
    for (int t = 0; t < T; t++) {
      int a = bo_policy_action(e, policy, policy_seed, (uint64_t)(env_index0 + i), t0 + (uint64_t)t);
      double r; uint8_t term; bo_info info; bo_obs obs;
      bo_step(e, a, &r, &term, &info);
      if (term) {
        bo_reset(e, 0, 0);
        if (e->tmpl_njokers > 0) bo_set_jokers(e, e->tmpl_jokers, e->tmpl_njokers);
        if (e->tmpl_ncons > 0) bo_set_consumables(e, e->tmpl_cons, e->tmpl_ncons);
        eps++;
      }
      bo_get_obs(e, &obs); /* the reference builds the observation on every step (:1064) */
      rs += r; sc += info.final_score; steps++;
    }
  }
  if (reward_sum) *reward_sum = rs;
  if (score_sum) *score_sum = sc;
  if (episodes) *episodes = eps;
  return steps;
}
