/* balatro_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C, scalar, one-struct-per-env restatement of the play-phase hot path of the reference
 * `balatro_gym/balatro_env_2.py::BalatroEnv` and everything it calls (CPython `random.Random` = MT19937,
 * balatro_game.py, scoring_engine.py, unified_scoring.py, complete_joker_effects.py, boss_blinds.py,
 * shop.py, cards.py).  Every function in balatro_oracle.c cites the reference file:line it follows
 * (paths relative to /root/reference/balatro_gym/).
 *
 * PARITY PIN: checked against (1) the reference's own known answers (tests/chips_test.py:5-24 and
 * balatro_trajectories.json play_hand transitions), (2) golden vectors generated in the build container by
 * importing the Python reference (tests/golden/, generator oracle/gen_golden.py), and (3) -- only when
 * /root/reference is present -- the imported reference itself (tests/test_oracle_vs_reference.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.  The shipped
 * product (balatro_gym_amd/) never links, includes or calls it.
 */
#ifndef BALATRO_ORACLE_H
#define BALATRO_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BO_NACT 60
#define BO_MAX_HAND 16
#define BO_MAX_JOKERS 10
#define BO_MAX_ITEMS 10

enum { BO_PHASE_PLAY = 0, BO_PHASE_SHOP = 1, BO_PHASE_BLIND_SELECT = 2, BO_PHASE_PACK_OPEN = 3 };

/* env flags */
#define BO_FLAG_SCORER_JOKERS 1u /* give the scorer joker NAMES (unified_scoring.py:313-351 style) so the joker chain is live */

/* hand-type name style seen by the joker chain (complete_joker_effects.py:64-80 compares strings) */
#define BO_NAMES_ENV 0 /* 'One Pair','Three Kind','Four Kind' (balatro_env_2.py:674) */
#define BO_NAMES_SIM 1 /* 'Pair','Three of a Kind','Four of a Kind' */

/* error codes (info.error) */
enum {
  BO_OK = 0,
  BO_ERR_INVALID_ACTION = 1,     /* balatro_env_2.py:627 */
  BO_ERR_PSYCHIC = 2,            /* boss_blinds.py:388 */
  BO_ERR_EYE = 3,                /* boss_blinds.py:393 */
  BO_ERR_MOUTH = 4,              /* boss_blinds.py:399 */
  BO_ERR_VERDANT = 5,            /* boss_blinds.py:405 */
  BO_ERR_REROLL_FUNDS = 6,       /* shop.py:173 */
  BO_ERR_JOKER_SLOTS = 7,        /* shop.py:196 */
  BO_ERR_CONSUMABLE = 8,         /* balatro_env_2.py:1166-1168 result['success'] is False */
  BO_ERR_MAX_ANTE = 9,           /* balatro_env_2.py:620 (terminated, reward 0) */
  BO_ERR_MAX_SCORE = 10,         /* balatro_env_2.py:623 */
  BO_ERR_CONSUMABLE_RAISES = 11, /* the reference raises here (consumables.py:246,381,496,506): reward -1.0 by harness convention */
  BO_ERR_CONSUMABLE_DECK = 12    /* Immolate on fewer than 13 real cards (fewer than 8 would be left: the hand's indexes go stale) / a Cryptid that would make 128 cards (deck_size is an int8): state untouched */
};

/* info.flags */
#define BO_INFO_BEAT_BLIND 1
#define BO_INFO_FAILED 2
#define BO_INFO_SKIPPED_BLIND 4
#define BO_INFO_OPENED_PACK 8
#define BO_INFO_BOUGHT_CARD 16
#define BO_INFO_BOUGHT_VOUCHER 32
#define BO_INFO_BOUGHT_JOKER 64
#define BO_INFO_SOLD_JOKER 128
#define BO_INFO_CURRICULUM 256

typedef struct { uint32_t mt[624]; int32_t mti; } bo_mt;

/* Observation in the reference's dtypes (balatro_env_2.py:1488-1531), natural alignment. */
typedef struct {
  int64_t selected_cards[8];
  int64_t face_down_cards[8];
  int64_t chips_scored;
  int32_t round_chips_scored;
  float progress_ratio;
  int32_t mult;
  int32_t chips_needed;
  int32_t money;
  int32_t hands_played;
  int32_t best_hand_this_ante;
  int16_t ante;
  int16_t joker_ids[10];
  int16_t consumables[5];
  int16_t shop_items[10];
  int16_t shop_costs[10];
  int16_t shop_rerolls;
  int8_t hand[8];
  int8_t hand_size;
  int8_t deck_size;
  int8_t round;
  int8_t hands_left;
  int8_t discards_left;
  int8_t joker_count;
  int8_t joker_slots;
  int8_t consumable_count;
  int8_t consumable_slots;
  int8_t hand_levels[12];
  int8_t phase;
  int8_t action_mask[BO_NACT];
  int8_t boss_blind_active;
  int8_t boss_blind_type;
} bo_obs;

typedef struct {
  int64_t final_score;     /* info['final_score'] on an accepted play, else 0 */
  double reward_terms[8];  /* info['reward_breakdown']: progress, milestone, score, hand_quality, efficiency, synergy, strategy, ante_bonus */
  int32_t error;           /* BO_ERR_* */
  int32_t flags;           /* BO_INFO_* */
  int32_t aux;             /* boss type on 47 / sold or bought joker id / pack first card; on a boss rejection what the message names
                            * (boss_blinds.py:393,399,405): the hand type played again (Eye), the one allowed type (Mouth), the cards required (Verdant) */
  int8_t hand_type;        /* info['hand_type'] on an accepted play, else -1 */
  int8_t cards_played;     /* info['cards_played'] */
  double breakdown[8];     /* info['score_breakdown'] of an accepted play (unified_scoring.py:129-137, :293-297): final_chips, final_mult,
                            * final_x_mult, card_chips, base_chips, base_mult, money_gained, 0 */
} bo_info;

typedef struct { uint8_t type; uint8_t payload; int32_t cost; } bo_item;

typedef struct bo_env {
  uint32_t flags;
  int32_t max_ante; /* CurriculumBalatroEnv cap (train_balatro_agent.py:146-152); 0 = off */
  int64_t master_seed;
  bo_mt stream[16];
  uint8_t stream_ready[16];
  bo_mt grand; /* per-env stand-in for the process-global `random` module */
  /* UnifiedGameState (balatro_env_2.py:165-212) */
  int32_t ante, round, phase;
  int64_t chips_needed, chips_scored, round_chips_scored;
  int64_t money;
  uint8_t deck[52]; /* card code = (rank-2)*4 + suit (cards.py:103-104); the first `ndeck` entries are live */
  int32_t ndeck;    /* cards.Card objects left in state.deck (52 until Immolate removes some) */
  int32_t nforeign; /* consumables.Card copies Cryptid appended BEHIND them: never drawn (draws take the lowest free index), only counted */
  int32_t hand[BO_MAX_HAND];
  int32_t nhand;
  int32_t sel[8];
  int32_t nsel;
  int32_t hands_left, discards_left, hand_size;
  int32_t jokers[BO_MAX_JOKERS];
  int32_t njokers;
  int32_t consumables[5];
  int32_t nconsumables;
  int32_t n_magic_trick, n_minimalist;
  int32_t joker_slots, consumable_slots;
  int64_t shop_reroll_cost_state; /* state.shop_reroll_cost (stale after rerolls) */
  int64_t hands_played_total, hands_played_ante, best_hand_this_ante, jokers_sold;
  uint8_t hand_levels[12];   /* engine.hand_levels (scoring_engine.py:66), capped at 15 */
  int32_t obs_levels[12];    /* state.hand_levels (observation only; += 1 per planet, uncapped) */
  uint32_t play_counts[12];
  uint8_t enh[52], edi[52], seal[52]; /* state.card_states (injection only) */
  /* BalatroGame */
  uint32_t highlighted; /* bit p = position p highlighted */
  /* BossBlindManager */
  int32_t boss_type; /* 0 = none, else BossBlindType value 1..28 */
  uint32_t boss_played_types;
  uint64_t boss_played_cards; /* deck-index mask standing in for id(card) */
  int32_t boss_first_hand, boss_hands_played, boss_cards_required;
  uint32_t face_down; /* bit i = position i face down */
  /* Shop */
  int32_t shop_exists, shop_n, shop_ante;
  bo_item shop_items[BO_MAX_ITEMS];
  int64_t shop_reroll_base; /* Shop.reroll_cost */
  bo_mt shop_rng;
  /* harness: reset template used by bo_rollout */
  int32_t tmpl_jokers[5];
  int32_t tmpl_njokers;
  int32_t tmpl_cons[2];
  int32_t tmpl_ncons;
} bo_env;

/* ---- CPython random.Random restated (Appendix B of SURVEY.md) ---- */
void bo_mt_seed(bo_mt* m, uint64_t seed);            /* random.Random(int) / random.seed(int), seed < 2**64 */
uint32_t bo_mt_u32(bo_mt* m);                        /* genrand_uint32 */
uint32_t bo_mt_getrandbits(bo_mt* m, int k);         /* k <= 32 */
uint32_t bo_mt_randbelow(bo_mt* m, uint32_t n);      /* _randbelow_with_getrandbits, n < 2**32 */
uint64_t bo_mt_randbelow64(bo_mt* m, uint64_t n);    /* n <= 2**32 (k up to 33 bits) */
double bo_mt_random(bo_mt* m);                       /* random() */
void bo_mt_shuffle_u8(bo_mt* m, uint8_t* x, int n);  /* shuffle() */

/* ---- pure functions ---- */
int bo_classify(const uint8_t* cards, int n);        /* balatro_game.py:40-93; cards are card codes */
int bo_rank_base_chips(int rank);                    /* cards.py:52-60 */
void bo_hand_chips_mult(int hand_type, int level, int64_t* chips, int64_t* mult); /* scoring_engine.py:87-101 */
int64_t bo_blind_chips(int ante, int blind);         /* balatro_env_2.py:66-74 */
uint64_t bo_global_seed(int64_t seed);               /* G(seed): harness convention, see DESIGN.md */
uint32_t bo_policy_hash(uint64_t policy_seed, uint64_t env_index, uint64_t t);

typedef struct { int32_t rank; int32_t suit; int32_t chips; } bo_scard; /* suit 0..3 = C,D,H,S ; 4 = 'Stone' */
typedef struct { int64_t score, chips, mult; double x_mult; int64_t money; int32_t draws; } bo_score_out;
/* unified_scoring.py:111-299 with jokers given as ids of their names */
void bo_score_hand(const bo_scard* cards, int ncards, const bo_scard* scoring, int nscoring, int hand_type,
                   int name_style, int level, const int32_t* jokers, int njokers, int hands_left,
                   int discards_left, int deck_len, bo_mt* grand, bo_score_out* out);

/* ---- balatro_sim.py (secondary evaluator / scorer): oracle/bo_sim.c ---- */
typedef struct { int32_t rank, suit, base_value, enhancement, edition, seal; } bo_simcard; /* codes: see bo_sim.c */
typedef struct {
  int8_t top;         /* results['top'] as a hand-type number 0..11 */
  int8_t nlists[12];  /* len(results[type]) */
  int8_t n0[12];      /* len(results[type][0]) */
  int8_t pos[12][8];  /* results[type][0] as positions in the evaluated hand (-1 padded) */
} bo_sim_eval;
typedef struct { int64_t score, chips, add_mult; double x_mult; int64_t money; int32_t draws, top, nscoring; } bo_sim_score_out;
void bo_sim_evaluate(const bo_simcard* hand, int n, int four_fingers, int shortcut, bo_sim_eval* out); /* balatro_sim.py:220-366 */
void bo_sim_score(const bo_simcard* cards, int n, const int32_t* jokers, int njokers, int hands_left, int discards_left, int deck_len,
                  bo_mt* grand, bo_sim_score_out* out); /* balatro_sim.py:402-548; Four Fingers (18) / Shortcut (69) among the jokers act on the evaluation */

/* ---- env ---- */
bo_env* bo_create(uint32_t flags, int32_t max_ante);
void bo_destroy(bo_env* e);
void bo_construct(bo_env* e, int64_t seed);          /* BalatroEnv(seed=seed): rng + reset(); seeds grand with G(seed) */
void bo_reset(bo_env* e, int has_seed, int64_t seed);/* reset(seed=...) balatro_env_2.py:505-558 */
void bo_get_obs(const bo_env* e, bo_obs* o);         /* balatro_env_2.py:1473-1541 */
void bo_action_mask(const bo_env* e, int8_t* mask);  /* balatro_env_2.py:1426-1471 */
void bo_step(bo_env* e, int action, double* reward, uint8_t* terminated, bo_info* info); /* :616-637 */
void bo_set_jokers(bo_env* e, const int32_t* ids, int n);          /* harness injection (config 3) */
void bo_set_card_state(bo_env* e, int deck_idx, int enh, int edi, int seal);
void bo_set_hand_level(bo_env* e, int hand_type, int level);
void bo_set_template_jokers(bo_env* e, const int32_t* ids, int n); /* jokers now + after every reset in bo_rollout */
void bo_set_consumables(bo_env* e, const int32_t* ids, int n);     /* harness injection: state.consumables by id (config 4) */
void bo_set_deck(bo_env* e, const uint8_t* codes52); /* harness injection: the live deck order (state.deck) */
void bo_set_max_ante(bo_env* e, int max_ante); /* the curriculum cap of this env (0 = none) */
void bo_set_money(bo_env* e, int64_t money);   /* harness injection: state.money */
void bo_set_ante(bo_env* e, int ante);         /* harness injection: state.ante */
int bo_policy_action(const bo_env* e, int policy, uint64_t policy_seed, uint64_t env_index, uint64_t t);

/* policies */
#define BO_POLICY_UNIFORM 0    /* k-th valid action, k = hash % n_valid */
#define BO_POLICY_SMALL_ONLY 1 /* BLIND_SELECT->45, SHOP->31, else uniform (balatro_env_2.py:1841-1849) */
#define BO_POLICY_CYCLE3 2     /* BLIND_SELECT->45+env%3, SHOP->31, else uniform */

/* Batched random-policy rollout for the CPU baseline: envs[i] for i in [0,n), T steps each, SAME_STEP auto-reset.
 * Returns the number of env-steps executed; checksum accumulates rewards/scores for cross-checking. */
int64_t bo_rollout(bo_env** envs, int n, int64_t env_index0, int T, int policy, uint64_t policy_seed,
                   uint64_t t0, double* reward_sum, int64_t* score_sum, int64_t* episodes);

#ifdef __cplusplus
}
#endif
#endif
