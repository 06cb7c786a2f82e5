/* balatro_mi355x.h -- C ABI of the MI355X-native vectorised Balatro environment (libbalatro_mi355x.so).
 *
 * This is the drop-in boundary for ONE hot path of cassiusfive/balatro-gym: the step()/reset() loop of
 * `balatro_gym/balatro_env_2.py::BalatroEnv` (deck shuffle/draw -> hand classification -> chip/mult accumulation ->
 * joker chain -> planet-level / boss-blind multipliers -> blind outcome -> observation + action mask).  The reference
 * is pure Python with no FFI; the entry points below are what a ctypes binding on the reference side would call
 * (see INTEGRATION.md).  Each entry point cites the reference interface it replaces (paths relative to the
 * reference's balatro_gym/ directory).
 *
 * Conventions: plain C, no torch types; every function returns 0 on success and a negative BG_E_* code on failure
 * (bg_last_error() gives the text); all `*_dev` pointers are DEVICE pointers into buffers the CALLER owns (e.g.
 * torch tensors' data_ptr()); `stream` is a hipStream_t passed as void* (NULL = default stream); the library owns
 * the per-env game/RNG state (structure-of-arrays in HBM) and never synchronises the host inside bg_step /
 * bg_rollout.  One handle drives one GPU; a handle is not thread-safe, different handles are independent.
 * There is NO CPU fallback: without a HIP device bg_create fails.
 */
#ifndef BALATRO_MI355X_H
#define BALATRO_MI355X_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BG_NUM_ACTIONS 60 /* constants.py:117 ACTION_SPACE_SIZE */
#define BG_OBS_BYTES 330  /* bytes of one observation in the reference's dtypes */

/* bg_create flags */
#define BG_FLAG_SCORER_JOKERS 1u /* hand the scorer joker NAMES (unified_scoring.py:313-351) so the joker chain is live */
#define BG_FLAG_AUTORESET 2u     /* SAME_STEP auto-reset: a terminated env is reset() inside the same bg_step call */
#define BG_FLAG_CARD_STATES 4u  /* keep cards.py CardState (enhancement / edition / seal) per deck index: bg_inject_cards */

/* error codes */
#define BG_E_ARG (-1)
#define BG_E_HIP (-2)
#define BG_E_NODEVICE (-3)
#define BG_E_INTERNAL (-4) /* device-side invariant violated (RNG look-ahead ring underflow); sticky */

/* info.error values (device side) -- reference error strings in brackets */
#define BG_ERR_NONE 0
#define BG_ERR_INVALID_ACTION 1 /* balatro_env_2.py:627 'Invalid action' */
#define BG_ERR_PSYCHIC 2        /* boss_blinds.py:388 'Must play exactly 5 cards' */
#define BG_ERR_EYE 3            /* boss_blinds.py:393 'Cannot play <type> again' */
#define BG_ERR_MOUTH 4          /* boss_blinds.py:399 'Can only play <type>' */
#define BG_ERR_VERDANT 5        /* boss_blinds.py:405 'Must play at least <n> cards' */
#define BG_ERR_REROLL_FUNDS 6   /* shop.py:173 'Insufficient chips for reroll' */
#define BG_ERR_JOKER_SLOTS 7    /* shop.py:196 'Joker slots full' */
#define BG_ERR_CONSUMABLE 8     /* balatro_env_2.py:1166-1168 the consumable had no effect (result['success'] False), reward -1.0 */
#define BG_ERR_MAX_ANTE 9       /* balatro_env_2.py:620 terminated 'max_ante_reached' */
#define BG_ERR_MAX_SCORE 10     /* balatro_env_2.py:623 terminated 'max_score_reached' */
#define BG_ERR_CONSUMABLE_RAISES 11 /* the reference RAISES here (consumables.py:246,381,418,444 list.remove of a target class;
                                     * :496,506 assignment to a frozen dataclass): reward -1.0, state as the exception leaves it */
#define BG_ERR_CONSUMABLE_DECK 12   /* Immolate on a deck of fewer than 13 real cards (consumables.py:519-531: fewer than 8 would be left, and the
                                     * hand's deck indexes 0..7 go stale -- the reference's unguarded deck[i] reads, balatro_env_2.py:577,670,937,
                                     * raise from there on) / a Cryptid that would make the deck 128 cards (:581-591: deck_size is
                                     * np.int8(len(deck)), :1491 -- OverflowError under numpy >= 2, a wrapped value under numpy 1);
                                     * reward -1.0, state untouched */

/* info.flags bits */
#define BG_INFO_BEAT_BLIND 1     /* info['beat_blind'] */
#define BG_INFO_FAILED 2         /* info['failed'] */
#define BG_INFO_SKIPPED_BLIND 4  /* info['skipped_blind'] */
#define BG_INFO_OPENED_PACK 8    /* info['opened_pack'] */
#define BG_INFO_BOUGHT_CARD 16   /* info['bought_card'] */
#define BG_INFO_BOUGHT_VOUCHER 32
#define BG_INFO_BOUGHT_JOKER 64
#define BG_INFO_SOLD_JOKER 128   /* info['sold_joker'] (aux = joker id) */
#define BG_INFO_CURRICULUM 256   /* info['curriculum_limit_reached'] (train_balatro_agent.py:146-152) */
#define BG_INFO_AUTORESET 512    /* the env was reset inside this step (BG_FLAG_AUTORESET) */

/* policies of bg_rollout (counter-hash random policy) */
#define BG_POLICY_UNIFORM 0    /* k-th valid action, k = hash(seed, env, t) mod n_valid */
#define BG_POLICY_SMALL_ONLY 1 /* BLIND_SELECT->45, SHOP->31, else uniform (balatro_env_2.py:1841-1849) */
#define BG_POLICY_CYCLE3 2     /* BLIND_SELECT->45+env%3, SHOP->31, else uniform */
#define BG_POLICY_HASH_OBS 0x100 /* OR into `policy`: also fold every observation row into stats.obs_hash (tests) */

typedef struct bg_handle bg_handle;

/* Observation: one device pointer per key of BalatroEnv._get_observation() (balatro_env_2.py:1488-1531), each a
 * contiguous [n_envs, ...] array in the REFERENCE's dtype.  A NULL pointer skips that key. */
typedef struct bg_obs_ptrs {
  int8_t* hand;                 /* [N,8]  int8, -1 padded */
  int8_t* hand_size;            /* [N] */
  int8_t* deck_size;            /* [N] */
  int64_t* selected_cards;      /* [N,8]  int64 0/1 (the reference emits int64 here) */
  int64_t* chips_scored;        /* [N] */
  int32_t* round_chips_scored;  /* [N] */
  float* progress_ratio;        /* [N] */
  int32_t* mult;                /* [N] always 1 */
  int32_t* chips_needed;        /* [N] */
  int32_t* money;               /* [N] */
  int16_t* ante;                /* [N] */
  int8_t* round;                /* [N] */
  int8_t* hands_left;           /* [N] */
  int8_t* discards_left;        /* [N] */
  int8_t* joker_count;          /* [N] */
  int16_t* joker_ids;           /* [N,10] */
  int8_t* joker_slots;          /* [N] */
  int8_t* consumable_count;     /* [N] */
  int16_t* consumables;         /* [N,5] */
  int8_t* consumable_slots;     /* [N] */
  int16_t* shop_items;          /* [N,10] */
  int16_t* shop_costs;          /* [N,10] */
  int16_t* shop_rerolls;        /* [N] */
  int8_t* hand_levels;          /* [N,12] */
  int8_t* phase;                /* [N] */
  int8_t* action_mask;          /* [N,60] */
  int32_t* hands_played;        /* [N] */
  int32_t* best_hand_this_ante; /* [N] */
  int8_t* boss_blind_active;    /* [N] */
  int8_t* boss_blind_type;      /* [N] */
  int64_t* face_down_cards;     /* [N,8] int64 0/1 */
} bg_obs_ptrs;

/* Per-step info (the numeric content of the reference's info dict, balatro_env_2.py:895-925).  NULL skips. */
typedef struct bg_info_ptrs {
  int64_t* final_score;  /* [N] info['final_score'] on an accepted play, else 0 */
  int32_t* error;        /* [N] BG_ERR_* */
  int32_t* flags;        /* [N] BG_INFO_* */
  int32_t* aux;          /* [N] boss type on action 47 / joker id bought or sold / first pack card; on a boss rejection what the reference's
                          * message names (boss_blinds.py:393,399,405): BG_ERR_EYE the hand type played again, BG_ERR_MOUTH the one allowed
                          * hand type, BG_ERR_VERDANT the number of cards required */
  int8_t* hand_type;     /* [N] info['hand_type'] on an accepted play, else -1 */
  int8_t* cards_played;  /* [N] info['cards_played'] */
  double* reward_terms;  /* [N,8] info['reward_breakdown'] without 'total' */
  double* score_breakdown; /* [N,8] info['score_breakdown'] of an accepted play (balatro_env_2.py:909, unified_scoring.py:129-137,293-297):
                          * final_chips, final_mult, final_x_mult, card_chips, base_chips, base_mult, money_gained, 0 (integers are exact:
                          * below 2**53); joker_chips = final_chips - base_chips - card_chips, joker_mult = final_mult - base_mult,
                          * joker_x_mult = final_x_mult.  Zeros on every other step. */
} bg_info_ptrs;

/* Aggregate counters of a rollout (device, one struct per call; all ranks of a sharded job sum them on the host). */
typedef struct bg_rollout_stats {
  uint64_t steps;        /* env-steps executed */
  uint64_t episodes;     /* terminated episodes */
  uint64_t plays;        /* accepted PLAY_HAND steps */
  int64_t score_sum;     /* sum of final_score over accepted plays */
  uint64_t reward_bits;  /* XOR of the IEEE bit patterns of all rewards (order-independent checksum) */
  uint64_t obs_hash;     /* XOR over (env,t) of a 64-bit hash of each observation row (only with BG_POLICY_HASH_OBS) */
} bg_rollout_stats;

/* Identity of the device code: sha256 prefix over the sources, compiler flags and compiler version the library was built from
 * ("unsigned" for ad-hoc builds).  Reproducible across rebuilds of unchanged sources -- the bytes of the code object are not --, so
 * profiles/rNN_hbm_traffic.json measurements are keyed on it (bench.py roofline.traffic).  No reference counterpart. */
const char* bg_build_signature(void);

/* Replaces: constructing n_envs `BalatroEnv` objects (balatro_env_2.py:359-384) + SB3 SubprocVecEnv (hpc_train.py:60-65).
 * max_ante > 0 applies the CurriculumBalatroEnv cap (train_balatro_agent.py:146-152) to every env; bg_set_max_ante changes it
 * later.  Every entry point runs on the handle's device and leaves the caller's current device unchanged. */
int bg_create(int n_envs, int device_id, uint32_t flags, int max_ante, bg_handle** out);
/* bg_create with a memory hint: fused_steps_hint = the most steps the caller will ever ask ONE launch to fuse (bg_step users: 1;
 * an SB3-style collector: its n_steps; 0 = the default, 372-step launches).  The RNG look-ahead rings -- 0.2 MB per env at
 * full depth, 12.8 GB at 65 536 envs -- are sized for that: hint 16 -> 26 KB per env, hint 64 -> 46 KB (bg_state_bytes tells).
 * bg_max_fused_steps reports what the rings allow; longer bg_rollout / bg_step_many calls are split into several launches. */
int bg_create_ex(int n_envs, int device_id, uint32_t flags, int max_ante, int fused_steps_hint, bg_handle** out);
/* Replaces: `CurriculumBalatroEnv.current_max_ante` (train_balatro_agent.py:129-166: one wrapper, hence one cap, per env; the
 * cap rises by ante_increment when 80 % of the last 100 episodes reached it).  The cap is part of an env's state: it survives
 * reset() and travels in state blobs.  per_env_host == NULL sets `max_ante` for the masked envs (mask_host NULL = all),
 * otherwise per_env_host[i]; 0 = no cap; caps are in [0, 255].  The host decides WHEN to raise a cap (the statistics it needs
 * are the terminated flags and the ante of the finished episodes); see balatro_gym_amd/sb3_adapter.py CurriculumTracker. */
int bg_set_max_ante(bg_handle* h, int max_ante, const int32_t* per_env_host /*[N] or NULL*/, const uint8_t* mask_host, void* stream);
int bg_destroy(bg_handle* h);
const char* bg_last_error(const bg_handle* h); /* also valid with h == NULL after a failed bg_create */
int bg_num_envs(const bg_handle* h);
/* Largest T that bg_rollout runs as ONE kernel launch (longer rollouts are split into launches of this many steps, with
 * an RNG look-ahead refill between them).  Size [T, N] observation buffers to a multiple of it. */
int bg_max_fused_steps(const bg_handle* h);
uint64_t bg_state_bytes(const bg_handle* h); /* HBM held by the library for this handle */

/* Replaces: `DeterministicRNG(seed)` (balatro_env_2.py:84-106) for the masked envs; seeds_host[i] is env i's master seed,
 * mask_host (nullable = all) selects envs.  reseed_global != 0 also seeds the per-env stand-in for the process-global
 * `random` module with G(seed) = (seed + 16000) mod 2**32 (harness convention).  Does NOT reset the game. */
int bg_seed(bg_handle* h, const int64_t* seeds_host, const uint8_t* mask_host, int reseed_global, void* stream);

/* Replaces: `BalatroEnv.reset()` without a seed (balatro_env_2.py:505-558) for the envs with mask_dev[i] != 0
 * (NULL = all) and writes the observation of EVERY env. */
int bg_reset(bg_handle* h, const uint8_t* mask_dev, const bg_obs_ptrs* obs, void* stream);

/* Replaces: `BalatroEnv.step(action)` (balatro_env_2.py:616-637) for all envs in lockstep.  Invalid actions give
 * reward -1.0 / info.error exactly like the reference; nothing raises. */
int bg_step(bg_handle* h, const int32_t* actions_dev, const bg_obs_ptrs* obs, double* reward_dev,
            uint8_t* terminated_dev, uint8_t* truncated_dev, const bg_info_ptrs* info, void* stream);

/* K consecutive `step()` calls per env in ONE launch: actions_dev is [K, N] (row k = the actions of call k), and with
 * obs_stride_steps != 0 every output is a [K, N, ...] buffer (row k = what call k returned); with 0 the [N, ...] buffers are
 * overwritten and hold the last call's values.  Identical to K bg_step calls (same auto-reset rule, same info), without K
 * launches.  Replaces the inner loop of an SB3-style rollout collection when the actions of several steps are known up front
 * (scripted policies, replay, open-loop evaluation); hpc_train.py:60-65 / balatro_env_2.py:616-637. */
int bg_step_many(bg_handle* h, int K, const int32_t* actions_dev, const bg_obs_ptrs* obs, int obs_stride_steps, double* reward_dev,
                 uint8_t* terminated_dev, uint8_t* truncated_dev, const bg_info_ptrs* info, void* stream);

/* Replaces: `_get_observation()` / `_get_action_mask()` (balatro_env_2.py:1426-1541) without stepping. */
int bg_observe(bg_handle* h, const bg_obs_ptrs* obs, void* stream);

/* bg_step / bg_observe with the observation as ONE packed record per env (the BG_ROW_* layout of bg_rollout_rows below: every key a
 * strided view of a [N, row_stride_bytes] byte tensor, plus the step's reward / action / terminated) instead of 31 arrays: what a
 * policy network reads after `obs_as_tensor` + concatenation anyway.  A cheap step then patches the record image and the copier waves
 * write it (a step that emits 31 arrays spends most of its instructions on address arithmetic): the one-step launch is ~10 % shorter (19.5 against 21.6 us of kernel time at 65 536 envs: bench.py step_path).
 * Same step semantics, same reward / terminated / truncated / info arrays as bg_step (balatro_env_2.py:616-637, :1473-1541);
 * bg_observe_rows (bytes 352.. of a record untouched) after bg_reset(h, mask, NULL, stream) gives the records of a reset.
 * rows_dev: 16-byte aligned, row_stride_bytes a multiple of 16, >= BG_ROW_BYTES; BG_RECORD_STRIDE_LINES on a 128-byte aligned buffer is the fast layout. */
int bg_step_rows(bg_handle* h, const int32_t* actions_dev, uint8_t* rows_dev, uint64_t row_stride_bytes, double* reward_dev,
                 uint8_t* terminated_dev, uint8_t* truncated_dev, const bg_info_ptrs* info, void* stream);
int bg_observe_rows(bg_handle* h, uint8_t* rows_dev, uint64_t row_stride_bytes, void* stream);

/* Fused random-policy rollout: T steps of every env with the counter-hash policy computed on device and SAME_STEP
 * auto-reset, observations of step t written to row t of [T, N, ...] buffers when obs_stride_steps != 0 (or
 * overwritten in place when 0).  env_index0 = global index of this handle's env 0 (sharding); t0 = first step number.
 * Replaces the driver loop of balatro_env_2.py:1835-1859 / SB3 rollout collection. */
int bg_rollout(bg_handle* h, int T, int policy, uint64_t policy_seed, uint64_t env_index0, uint64_t t0,
               const bg_obs_ptrs* obs, int obs_stride_steps, double* reward_dev, uint8_t* terminated_dev,
               int32_t* actions_out_dev, bg_rollout_stats* stats_dev, void* stream);

/* bg_rollout with ONE packed record per (step, env) instead of one array per key: record (t, i) starts at
 * rows_dev + ((t * rows_stride_steps ? t : 0) * N + i) * row_stride_bytes, i.e. a [T, N, row_stride_bytes] byte tensor
 * when rows_stride_steps != 0 (a [N, row_stride_bytes] tensor overwritten every step when 0).  Every field keeps the
 * reference's dtype at a naturally aligned offset (BG_ROW_*), so each key is a strided view of the same buffer; the
 * step's reward, action and terminated flag ride in the record.  A record is written by one lane with whole 16-byte
 * stores, which keeps HBM write traffic at the record size however far the envs of a workgroup drift apart in time.
 * rows_dev must be 16-byte aligned, row_stride_bytes a multiple of 16 and >= BG_ROW_BYTES.
 * FAST LAYOUT: row_stride_bytes == BG_RECORD_STRIDE_LINES (384) with rows_dev 128-byte aligned.  The library then writes every record as
 * three WHOLE 128-byte lines (bytes 352..383 as zeros): the MI355X's HBM takes records scattered over a buffer at 4.5 TB/s when their
 * lines are written completely and at 2.6 TB/s when each record ends in partial lines, which is what bounds a fused rollout
 * (tools/micro/recwrite.hip).  With any other stride bytes 352.. of a record are left untouched. */
#define BG_ROW_BYTES 352
#define BG_RECORD_STRIDE_LINES 384
#define BG_ROW_SELECTED_CARDS 0       /* int64[8] */
#define BG_ROW_FACE_DOWN_CARDS 64     /* int64[8] */
#define BG_ROW_CHIPS_SCORED 128       /* int64 */
#define BG_ROW_REWARD 136             /* float64: reward of this step */
#define BG_ROW_ROUND_CHIPS_SCORED 144 /* int32 */
#define BG_ROW_PROGRESS_RATIO 148     /* float32 */
#define BG_ROW_MULT 152               /* int32 */
#define BG_ROW_CHIPS_NEEDED 156       /* int32 */
#define BG_ROW_MONEY 160              /* int32 */
#define BG_ROW_HANDS_PLAYED 164       /* int32 */
#define BG_ROW_BEST_HAND_THIS_ANTE 168 /* int32 */
#define BG_ROW_ACTION 172             /* int32: action taken at this step */
#define BG_ROW_ACTION_MASK 176        /* int8[60] */
#define BG_ROW_JOKER_IDS 236          /* int16[10] */
#define BG_ROW_SHOP_ITEMS 256         /* int16[10] */
#define BG_ROW_SHOP_COSTS 276         /* int16[10] */
#define BG_ROW_CONSUMABLES 296        /* int16[5] */
#define BG_ROW_ANTE 306               /* int16 */
#define BG_ROW_SHOP_REROLLS 308       /* int16 */
#define BG_ROW_HAND 310               /* int8[8] */
#define BG_ROW_HAND_LEVELS 318        /* int8[12] */
#define BG_ROW_HAND_SIZE 330          /* int8; the following scalars are int8 each */
#define BG_ROW_DECK_SIZE 331
#define BG_ROW_ROUND 332
#define BG_ROW_HANDS_LEFT 333
#define BG_ROW_DISCARDS_LEFT 334
#define BG_ROW_JOKER_COUNT 335
#define BG_ROW_JOKER_SLOTS 336
#define BG_ROW_CONSUMABLE_COUNT 337
#define BG_ROW_CONSUMABLE_SLOTS 338
#define BG_ROW_PHASE 339
#define BG_ROW_BOSS_BLIND_ACTIVE 340
#define BG_ROW_BOSS_BLIND_TYPE 341
#define BG_ROW_TERMINATED 342         /* uint8: the step ended the episode (SAME_STEP auto-reset: the record already shows the new episode) */
int bg_rollout_rows(bg_handle* h, int T, int policy, uint64_t policy_seed, uint64_t env_index0, uint64_t t0,
                    uint8_t* rows_dev, uint64_t row_stride_bytes, int rows_stride_steps, bg_rollout_stats* stats_dev,
                    void* stream);

/* Sharded jobs (one process per GPU): the exchange of the design -- every rank sees the CURRENT record of every env -- without a collective
 * behind the launch.  bufs[r] is rank r's gather buffer, uint8 [world][N][BG_ROW_BYTES], as mapped into THIS process (own buffer: an ordinary device
 * pointer; peers': hipIpcOpenMemHandle / torch's CUDA IPC over xGMI); N = this handle's env count, the same on every rank.  From then on the LAST launch
 * of every bg_rollout_rows call also writes the record of its last step into slot [rank] of all `world` buffers, from the engine's copy-out, while the
 * launch runs: when every rank's call has completed (a barrier between the ranks), every buffer holds all world x N current records.  world = 0 switches
 * it off.  Replaces: SubprocVecEnv's pipe traffic of observations to the learner process (hpc_train.py:60-65); the RCCL all_gather of
 * balatro_gym_amd/sharded.py stays as the fallback where peer mapping is not available. */
int bg_set_gather_peers(bg_handle* h, void* const* bufs, int world, int rank);

/* Harness injection (configs 3-4): per-env "reset template" applied by every reset of that env -- owned jokers (ids
 * from jokers.py), money, ante and hand levels; -1 / NULL leaves a field at its reset default.  apply_now != 0 also
 * writes them into the live state.  Replaces direct writes to env.state.* (e.g. train_balatro_agent.py:150). */
int bg_inject(bg_handle* h, const int32_t* jokers_host /*[N,5] or NULL*/, const int32_t* njokers_host /*[N]*/,
              const int64_t* money_host /*[N] or NULL*/, const int32_t* ante_host /*[N] or NULL*/,
              const uint8_t* levels_host /*[N,12] or NULL*/, const uint8_t* mask_host, int apply_now, void* stream);

/* The LIVE deck order of the masked envs: decks_host is [N, 52] card codes (rank-2)*4+suit, each row a permutation of 0..51.
 * Replaces direct writes to env.state.deck / env.game.deck (one aliased list, balatro_env_2.py:528-531), which is how a harness
 * or a test puts chosen cards under the hand's deck indexes (e.g. a straight flush at deck[0..4]: classification reads
 * deck[position]).  The next reset() reshuffles as always. */
int bg_inject_deck(bg_handle* h, const uint8_t* decks_host /*[N,52]*/, const uint8_t* mask_host, void* stream);

/* Card states (cards.py:62-139 CardState; the reference sets them through tarot cards -- not on this path -- the harness
 * injects them directly, like env.card_states[idx] = CardState(...)): per env and deck INDEX (position in the shuffled
 * deck, 0..51) the enhancement (0 none, 1 BONUS, 2 MULT, 3 WILD, 4 GLASS, 5 STEEL, 6 STONE, 7 GOLD, 8 LUCKY), edition
 * (0 none, 1 FOIL, 2 HOLOGRAPHIC, 3 POLYCHROME) and seal (0 none, 1 GOLD, 2 RED, 3 BLUE, 4 PURPLE) codes, [N, 52] u8 each
 * (NULL = all 0).  Effects on the path: balatro_env_2.py:287-325 (chips, stone), :703-767 (glass / lucky rolls on stream
 * 'card_enhancement', seals, steel, retriggers), :1334-1343 (gold).  reset() clears the states (:511); the injected set
 * is re-applied after every reset, like bg_inject's template.  Needs BG_FLAG_CARD_STATES at bg_create. */
int bg_inject_cards(bg_handle* h, const uint8_t* enh_host, const uint8_t* edition_host, const uint8_t* seal_host,
                    const uint8_t* mask_host, int apply_now, void* stream);

/* Consumables (balatro_env_2.py:1066-1172 _use_consumable over consumables.py; BASELINE config 4): per-env reset template
 * of state.consumables, ids as in _get_consumable_ids (balatro_env_2.py:1545-1567): tarots 1-22, planets 30-41,
 * spectrals 50-67; ids_host is [N, 2], n_host[i] in [0, 2] (-1 = back to the reset default).  Re-applied after every
 * reset like bg_inject's template.  Tarot / spectral cards edit card states, so they need BG_FLAG_CARD_STATES (planets do
 * not).  In-env sources of consumables: blue seals (planets), purple seals (tarots), The Fool / High Priestess /
 * Emperor / Judgement.  All 52 ids are followed (incl. the reference's quirks: INTEGRATION.md section 6); error codes
 * BG_ERR_CONSUMABLE / _RAISES / _DECK.  Replaces direct writes to env.state.consumables. */
int bg_inject_consumables(bg_handle* h, const int32_t* ids_host /*[N,2]*/, const int32_t* n_host /*[N]*/,
                          const uint8_t* mask_host, int apply_now, void* stream);

/* Replaces: save_state()/load_state() (balatro_env_2.py:1575-1615).  Blob = versioned raw copy of one env's state
 * (game + all RNG streams + look-ahead rings).  bg_state_blob_bytes gives the size. */
uint64_t bg_state_blob_bytes(const bg_handle* h);
int bg_get_state(bg_handle* h, int env_index, void* blob_host, uint64_t blob_bytes);
int bg_set_state(bg_handle* h, int env_index, const void* blob_host, uint64_t blob_bytes);

/* Top up the RNG look-ahead rings (pre-shuffled decks, pre-seeded shop streams, global-stream blocks).  bg_step /
 * bg_reset / bg_rollout call it themselves; exposed for tests and for overlapping it on a side stream. */
int bg_refill(bg_handle* h, void* stream);

/* Measurement hooks (bench.py): when enabled every kernel launch is bracketed by HIP events on its own stream.
 * bg_get_profile synchronises and returns out8 = {rollout kernel ms, rollout launches, fused steps summed over
 * launches, refill kernel ms, refill launches, step kernel ms, step launches, 0} since the last call. */
int bg_set_profiling(bg_handle* h, int enable);
int bg_get_profile(bg_handle* h, double* out8);

/* Measurement hook (bench.py `roofline.peak_measured`): a plain streaming copy of `bytes` (16 B per lane, src -> dst, both
 * 16-byte aligned device buffers) repeated `iters` times; *gbps_out = (bytes read + bytes written) / time.  SURVEY 8(d): the
 * roofline fraction is quoted against the nominal 8 TB/s AND against what a copy kernel reaches on the same GPU. */
int bg_bench_copy(const void* src_dev, void* dst_dev, uint64_t bytes, int iters, double* gbps_out, void* stream);
/* The write-only twin (bench.py `roofline.peak_measured_write`): a plain fill of `bytes` (16 B per lane), *gbps_out = bytes written /
 * time.  The step path's algorithmic traffic is ~80 % stores (every step's record), so this is the ceiling that applies to it. */
int bg_bench_fill(void* dst_dev, uint64_t bytes, int iters, double* gbps_out, void* stream);

/* Check the sticky device error word (synchronises the stream). */
int bg_check(bg_handle* h, void* stream);

/* ---- Operator-level entry points: the units the reference exposes as callables of their own, batched (lane = case).  They
 * need no handle: they run on the CURRENT device, on caller-owned device buffers, and call the same device functions as the
 * step path.  bg_last_error(NULL) gives the text of a failure. ---- */

/* Replaces: `BalatroGame._classify_hand(cards)` (balatro_game.py:40-93).  cards_dev is [M, 8] card codes (rank-2)*4+suit
 * (8-byte aligned), n_dev[i] in [0, 8] of them are valid; hand_type_dev[i] = HandType value 0..8 (scoring_engine.py:12-24). */
int bg_classify_batch(const uint8_t* cards_dev, const uint8_t* n_dev, uint8_t* hand_type_dev, int64_t m, void* stream);
/* The same with the lane mapping chosen by the caller: lanes_per_case = 1 (lane = hand, the default above) or 8 (lane = card, the
 * counts and sets by `__shfl_xor` inside 8-lane groups -- the mapping SURVEY 7.6 asks to benchmark beside the first).  Results are
 * identical.  kernel_ms_out (may be NULL): time of the kernel alone, HIP events on `stream`; non-NULL synchronises the stream. */
int bg_classify_batch_ex(const uint8_t* cards_dev, const uint8_t* n_dev, uint8_t* hand_type_dev, int64_t m, int lanes_per_case,
                         float* kernel_ms_out, void* stream);

/* Replaces: `UnifiedScorer.score_hand(ctx)` (unified_scoring.py:111-299) called with game_state['jokers'] = joker NAMES (as
 * unified_scoring.py:313-351 does) after random.seed(gseed).  One int32[BG_SCORE_CASE_WORDS] record per case:
 *   [0..23] 8 x (rank 2..14 | 0 for a STONE card, suit 0..3 = C D H S | 4 = 'Stone', chip_value()) = context.cards
 *   [24] len(cards)  [25] the first nscoring of them are context.scoring_cards  [26] hand type 0..11
 *   [27] name style: 0 = balatro_env_2.py:674 ('One Pair', 'Three Kind', 'Four Kind'), 1 = balatro_sim ('Pair', ...)
 *   [28] hand level  [29] number of jokers  [30..34] joker ids (jokers.py)  [35] hands_left  [36] discards_left
 *   [37] len(game_state['deck'])  [38] gseed (< 2**32)  [39] 0
 * and one int64[BG_SCORE_OUT_WORDS] result: score, final chips, final mult, bits of the float64 x_mult, money gained, words
 * of the global stream consumed, the word the NEXT getrandbits(32) would return (identifies the stream position), 0.
 * Synchronises the stream. */
#define BG_SCORE_CASE_WORDS 40
#define BG_SCORE_OUT_WORDS 8
int bg_score_hand_batch(const int32_t* cases_dev, int64_t* out_dev, int m, void* stream);
/* The same with lanes_per_case = 1 or 8 (8: lane = card while the hand is gathered, lane = joker for the per-card joker phase whose
 * totals are shuffle reductions, lane = card for Bloodstone's RNG words; the order-dependent main phase runs on every lane).
 * kernel_ms_out (may be NULL): time of the scoring kernel alone (the per-case random.seed() runs in a kernel of its own before it). */
int bg_score_hand_batch_ex(const int32_t* cases_dev, int64_t* out_dev, int m, int lanes_per_case, float* kernel_ms_out, void* stream);

/* balatro_sim.py (the reference's secondary, Balatro-accurate evaluator / scorer; not reachable from the live env).  A sim card
 * is six int32: rank 2..14, suit 0..3 = Clubs, Diamonds, Hearts, Spades, base_value, enhancement (0 None, 1 'bonus', 2 'mult',
 * 3 'wild', 4 'glass', 5 'steel', 6 'stone', 7 'gold', 8 'lucky'), edition (0 None, 1 'foil', 2 'holographic', 3 'polychrome',
 * 4 'negative'), seal (0 None, 1 'gold', 2 'red', 3 'blue', 4 'purple').  Hand types 0..11 = High Card, Pair, Two Pair, Three
 * of a Kind, Straight, Flush, Full House, Four of a Kind, Straight Flush, Five of a Kind, Flush House, Flush Five.
 *
 * Replaces: `BalatroSimulator.evaluate_hand(cards)` (balatro_sim.py:220-366 over get_x_same :108-126, get_flush :128-149,
 * get_straight :151-214).  hands_dev is int32 [M, 8, 6], n_dev[i] in [0, 8] cards are valid, flags_dev[i] bit 0 = a Four Fingers
 * joker is owned, bit 1 = Shortcut.  out_dev is int8 [M, BG_SIM_EVAL_BYTES]: [0] results['top']; [1..12] len(results[type]);
 * [13..24] len(results[type][0]); [32 + 8 * type + k] = position in the hand of card k of results[type][0] (-1 padded). */
#define BG_SIM_EVAL_BYTES 128
int bg_sim_evaluate_batch(const int32_t* hands_dev, const int32_t* n_dev, const int32_t* flags_dev, int8_t* out_dev, int m, void* stream);

/* Replaces: `BalatroSimulator.calculate_score(cards, game_state)` (balatro_sim.py:402-548) after random.seed(seed).  One
 * int32[BG_SIM_CASE_WORDS] record per case: [0..47] 8 sim cards, [48] number of cards, [49] number of jokers,
 * [50..54] player_state.jokers (ids; Four Fingers 18 / Shortcut 69 act on the evaluation and draw like any other joker),
 * [55] game_state['hands_left'] (1 when the key is absent, complete_joker_effects.py:46), [56] 'discards_left' (0 when absent),
 * [57] len(game_state['deck']), [58] seed (< 2**32), rest 0.  Result int64[8]: score, chips, added mult, bits of the float64
 * x_mult, money gained, words of the global stream consumed, the word the next getrandbits(32) returns, top | nscoring << 8.
 * Synchronises the stream. */
#define BG_SIM_CASE_WORDS 64
int bg_sim_score_batch(const int32_t* cases_dev, int64_t* out_dev, int m, void* stream);

#ifdef __cplusplus
}
#endif
#endif
