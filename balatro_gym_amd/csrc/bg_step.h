// bg_step.h -- the per-env step state machine (device).  One lane = one env; see bg_device.h for the layout.
// Reference: balatro_env_2.py::BalatroEnv.step and everything it calls; file:line cited per function.
#pragma once
#include "bg_device.h"

// what a step reports besides the state change
struct StepOut {
  double reward;
  int64_t final_score;
  double terms[8];
  int32_t error, flags, aux;
  int32_t hand_type, cards_played;
  bool terminated;
  // info['score_breakdown'] of an accepted play (unified_scoring.py:129-137, :293-297): what the scorer itself reports, before the env's
  // steel / boss / red-seal factors.  Only the bg_step instantiation of the engine emits these (dead code in the rollouts).
  // They are stored where they are computed (bd_dst = the step's row of bg_info_ptrs.score_breakdown, or null): carried to the end of
  // the step they would cost nine more live registers in a kernel that is out of them.
  double* bd_dst;
};

// ---------------------------------------------------------------------------------------------------------
// shop inventory storage (cold chunks 3..6): cost i32[9], (type | payload << 8) u16[9]
// ---------------------------------------------------------------------------------------------------------
// The inventory is kept in registers while an env is in the shop (ShopRegs = cold chunks 3..6): the action mask and the
// observation read it every step, and fetching it item by item would put ~9 dependent HBM round trips on every step of
// every wave that has one lane in SHOP phase.
struct ShopRegs { uint4 c3, c4, c5, c6; bool valid; };
__device__ __forceinline__ void bg_shop_load(const BgDev& d, int env, ShopRegs& sr) {
  if (sr.valid) return;
  size_t N = d.N;
  sr.c3 = bg_ld16a(&d.cold[3 * N + env]); sr.c4 = bg_ld16a(&d.cold[4 * N + env]); sr.c5 = bg_ld16a(&d.cold[5 * N + env]); sr.c6 = bg_ld16a(&d.cold[6 * N + env]); // (past the L1: bg_ld16a)
  sr.valid = true;
}
__device__ __forceinline__ void bg_shop_store(const BgDev& d, int env, const ShopRegs& sr) {
  size_t N = d.N;
  d.cold[3 * N + env] = sr.c3; d.cold[4 * N + env] = sr.c4; d.cold[5 * N + env] = sr.c5; d.cold[6 * N + env] = sr.c6;
}
__device__ __forceinline__ void bg_shop_unpack(const ShopRegs& sr, int32_t cost[9], uint32_t tp[9]) {
  cost[0] = sr.c3.x; cost[1] = sr.c3.y; cost[2] = sr.c3.z; cost[3] = sr.c3.w;
  cost[4] = sr.c4.x; cost[5] = sr.c4.y; cost[6] = sr.c4.z; cost[7] = sr.c4.w; cost[8] = sr.c5.x;
  tp[0] = sr.c5.y & 0xffffu; tp[1] = sr.c5.y >> 16; tp[2] = sr.c5.z & 0xffffu; tp[3] = sr.c5.z >> 16;
  tp[4] = sr.c5.w & 0xffffu; tp[5] = sr.c5.w >> 16; tp[6] = sr.c6.x & 0xffffu; tp[7] = sr.c6.x >> 16; tp[8] = sr.c6.y & 0xffffu;
}
__device__ __forceinline__ void bg_shop_pack(ShopRegs& sr, const int32_t cost[9], const uint32_t tp[9]) {
  sr.c3 = make_uint4(cost[0], cost[1], cost[2], cost[3]);
  sr.c4 = make_uint4(cost[4], cost[5], cost[6], cost[7]);
  sr.c5 = make_uint4(cost[8], tp[0] | (tp[1] << 16), tp[2] | (tp[3] << 16), tp[4] | (tp[5] << 16));
  sr.c6 = make_uint4(tp[6] | (tp[7] << 16), tp[8], 0, 0);
  sr.valid = true;
}
enum { IT_PACK = 1, IT_CARD = 2, IT_JOKER = 3, IT_VOUCHER = 4 }; // shop.py:17-21 ItemType (auto())
enum { PK_STANDARD = 0, PK_JOKER = 1, PK_TAROT = 2, PK_PLANET = 3, PK_SPECTRAL = 4 };

// balatro_env_2.py:1426-1471 _get_action_mask as a 60-bit set
__device__ __forceinline__ uint64_t bg_action_mask(const BgDev& d, int env, const Env& e, ShopRegs& sr) {
  uint64_t m = 0;
  if (e.phase == 0) {
    int n = e.nhand < 8 ? e.nhand : 8;
    m |= ((1ull << n) - 1) << 2;
    if (e.nsel > 0) { m |= 1ull; if (e.discards_left > 0) m |= 2ull; }
    m |= ((1ull << e.ncons) - 1) << 10;
  } else if (e.phase == 1) {
    if (e.bflags & BG_BF_SHOP_EXISTS) {
      bg_shop_load(d, env, sr);
      int32_t cost[9]; uint32_t tp[9];
      bg_shop_unpack(sr, cost, tp);
#pragma unroll
      for (int i = 0; i < 9; i++)
        if (i < e.shop_n && e.money >= cost[i]) m |= 1ull << (20 + i);
      if (e.money >= e.shop_reroll_state) m |= 1ull << 30;
    }
    m |= 1ull << 31;
    m |= ((1ull << e.njokers) - 1) << 32;
  } else if (e.phase == 2) {
    m |= 0xfull << 45;
  }
  return m;
}

// ---------------------------------------------------------------------------------------------------------
// reset (balatro_env_2.py:505-558).  The shuffled deck comes from the look-ahead ring filled by the refill kernel
// (`rng.shuffle('deck_shuffle', deck)` :525 depends on nothing but stream 0).
// ---------------------------------------------------------------------------------------------------------
// `pre`: the four 16-byte chunks of the ring's next deck, loaded ahead of time by the caller (nullptr = load them here);
// `pv`: the same in LDS (bg_engine3.h keeps the next deck of every env there), chunk k at pv[k * pv_stride]
typedef uint32_t bg_pv_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const bg_pv_u32x4 lds_cu4;
template <class DK>
__device__ __forceinline__ void bg_env_reset(const BgDev& d, int env, Env& e, DK& dk, const uint4* pre = nullptr, lds_cu4* pv = nullptr, int pv_stride = 0) {
  // the reset template (applied at the end) is requested FIRST: its HBM round trip then runs beside the cold stores and the deck copy instead of
  // behind them (the compiler cannot move a load above stores through other pointers)
  const uint4 t0 = d.tmpl[env], t1 = d.tmpl[(size_t)d.N + env];
  e.ante = 1; e.round = 1; e.phase = 2; e.chips_needed = 300; e.chips_scored = 0; e.round_chips = 0; e.money = 4;
  e.hand = 0; e.nhand = 0; e.sel = 0; e.nsel = 0; e.hands_left = 4; e.discards_left = 3; e.hand_size = 8;
  e.njokers = 0; e.jokers = 0; e.ncons = 0; e.cons0 = 0; e.cons1 = 0; e.n_magic = 0; e.n_minim = 0;
  e.shop_reroll_state = 5; e.hp_total = 0; e.hp_ante = 0; e.best_hand = 0; e.jokers_sold = 0;
  e.boss_type = 0; e.boss_types = 0; e.boss_cards = 0; e.boss_hp = 0; e.boss_req = 5; e.face_down = 0;
  e.bflags = (e.bflags & BG_BF_SHOP_EXISTS) | BG_BF_FIRST_HAND;
  e.highlighted = 0;
  e.ndrop = 0; e.nfo = 0; // a fresh 52-card deck (:519-525)
  e.levels = 0x111111111111ull; // ScoreEngine(): every level 1
  e.excess = 0;
  // hand_play_counts = 0 (cold chunks 0..2)
#pragma unroll
  for (int k = 0; k < 3; k++) d.cold[(size_t)k * d.N + env] = make_uint4(0, 0, 0, 0);
  // consume one pre-shuffled deck
  if (e.d_ready <= 0) atomicOr(d.err, BG_DEVERR_DECKRING);
  else {
#pragma unroll
    for (int k = 0; k < BG_NDECK; k++) {
      uint4 c;
      if (pv) { const bg_pv_u32x4 v = pv[k * pv_stride]; c = make_uint4(v.x, v.y, v.z, v.w); }   // (bg_engine3.h: the env's next deck waits in LDS; chunk by chunk, so that only one is in registers at a time)
      else c = pre ? pre[k] : d.ndeck[((size_t)e.d_head * BG_NDECK + k) * d.N + env];
      d.deck[(size_t)k * d.N + env] = c;
      bg_deck_set(dk, k, c);
    }
    e.d_head = (e.d_head + 1 == d.KD) ? 0 : e.d_head + 1;
    e.d_ready--; e.d_cons = (e.d_cons + 1) & 0xff;
  }
  // reset template (harness injection, applied after every reset)
  int tnj = (int)bg_b(t0.y, 1);
  if (t0.y & 0x80000000u) { e.njokers = tnj; e.jokers = (uint64_t)t0.x | ((uint64_t)(t0.y & 0xffu) << 32); }
  if (t0.y & 0x40000000u) e.money = (int32_t)t0.z;
  if (t0.y & 0x20000000u) e.ante = (int)bg_b(t0.y, 2);
  if (t0.y & 0x10000000u) e.levels = (uint64_t)t1.x | ((uint64_t)(t1.y & 0xffffu) << 32);
  if (t1.z & 0x80000000u) { e.ncons = (int)bg_b(t1.z, 0); e.cons0 = bg_b(t1.z, 1); e.cons1 = bg_b(t1.z, 2); }
  if constexpr (DK::kCards) { // card_states = {} (:511), then the harness re-applies its injected states
#pragma unroll
    for (int k = 0; k < BG_NCST; k++) d.cstate[(size_t)k * d.N + env] = d.ctmpl[(size_t)k * d.N + env];
  }
}

// ---------------------------------------------------------------------------------------------------------
// shop.py:104-139 + balatro_env_2.py:1383-1392
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double bg_shop_cost_mult(const Env& e, lds_JTables* jt) { // shop.py:104-108
  int k = e.shop_ante - 1;
  k = k < 0 ? 0 : (k > 100 ? 100 : k);
  double m = jt->pow115[k];
  if (e.n_magic > 0) m *= 0.9;
  return m;
}

// candid[j] = j-th id (0-based) of {1..145} \ owned, in library order (shop.py:123): walk the owned ids in ASCENDING
// order and step over each one that is <= the running id.
__device__ __forceinline__ uint64_t bg_sorted_jokers(const Env& e) {
  uint64_t v = e.jokers | (e.njokers < 8 ? (~0ull << (8 * e.njokers)) : 0ull); // pad with 0xff
#pragma unroll 1
  for (int pass = 0; pass < 4; pass++) { // bubble passes over 5 bytes (jokers <= 5)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint32_t a = (uint32_t)(v >> (8 * i)) & 0xffu, b2 = (uint32_t)(v >> (8 * (i + 1))) & 0xffu;
      if (a > b2) v = (v & ~(0xffffull << (8 * i))) | ((uint64_t)b2 << (8 * i)) | ((uint64_t)a << (8 * (i + 1)));
    }
  }
  // `j.id not in self.player.jokers` is a membership test: an id owned twice (Ankh's copy under scorer-level joker
  // names) is stepped over once
#pragma unroll
  for (int i = 4; i >= 1; i--)
    if (((v >> (8 * i)) & 0xffull) == ((v >> (8 * (i - 1))) & 0xffull)) v |= 0xffull << (8 * i);
  return v;
}
__device__ __forceinline__ int bg_candidate(uint64_t sorted, int j) {
  int id = j + 1;
#pragma unroll
  for (int q = 0; q < 5; q++) { int o = (int)((sorted >> (8 * q)) & 0xff); if (o <= id && o <= 145) id++; }
  return id;
}

// byte i (0..23) of six packed words, i per lane (registers cannot be indexed by a lane's value: a select chain; six scalars, not an array --
// an array handed around by reference ended up in scratch memory)
struct BgPk6 { uint32_t a, b, c, d, e, f; };
__device__ __forceinline__ uint32_t bg_pk_byte(const BgPk6 pk, int i) {
  const int g = i >> 2;
  uint32_t wv = pk.a;
  wv = g == 1 ? pk.b : wv; wv = g == 2 ? pk.c : wv; wv = g == 3 ? pk.d : wv; wv = g == 4 ? pk.e : wv; wv = g == 5 ? pk.f : wv;
  return (wv >> (8 * (i & 3))) & 0xffu;
}
// bit 7 of every byte of f -> bits 0..3 (byte k -> bit k): (f >> 7) has its bits at 0, 8, 16, 24, and the four shifted copies a multiply by
// 2^24 + 2^17 + 2^10 + 2^3 adds land on 16 different positions (no carries); bits 24..27 collect byte 0..3
__device__ __forceinline__ uint32_t bg_movemask4(uint32_t f) { return (((f >> 7) & 0x01010101u) * 0x01020408u) >> 24; }

__device__ __forceinline__ void bg_shop_inventory(const BgDev& d, int env, Env& e, RngWin& w, ShopRegs& sr) {
  BG_PROBE_BEGIN();
  bool full_state;
  const uint32_t* S = bg_sbase(d, env, e, full_state);
  const bool fresh = e.s_idx == 0 && !full_state;
  // A FRESH stream (every generate_shop): every draw looks at the TOP BYTE of its word only -- getrandbits(2) = byte >> 6, getrandbits(8) = byte,
  // getrandbits(6) = byte >> 2 -- and the slot's tail holds exactly those, packed by the seeding kernel (BG_SW_PK: six words): two 16-byte pieces of one
  // line (a winning play touched that line when it began), REQUESTED first and looked at behind the cost factor and the sorted joker list: ~200
  // instructions that do not need them.
  uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
  if (fresh) { const uint4* S4 = (const uint4*)(S + BG_SW_PK); q0 = S4[0]; q1 = S4[1]; }
  double mult = bg_shop_cost_mult(e, w.jt);
  const uint64_t sj = bg_sorted_jokers(e);
  int owned145 = 0;
#pragma unroll
  for (int q = 0; q < 5; q++) owned145 += bg_get8(sj, q) <= 145 ? 1 : 0;
  const uint32_t nc = (uint32_t)(145 - owned145); // 140..145: _randbelow(nc) looks at 8 bits
  // The seven draws -- choice of the third pack (_randbelow(3)), random.sample(candid, 3) by the selection-set method
  // (Lib/random.py sample(), n > 21: _randbelow(nc) until new), the voucher (_randbelow(2)), two randint(0, 51) -- are
  // rejection loops over consecutive words.  A loop per draw makes the wave iterate until its unluckiest lane accepts;
  // instead classify the next 24 words at once (one acceptance mask per kind of draw) and walk the masks with ffs.
  int third_r = 0, p0 = 0, p1 = 0, p2 = 0, v = 0, ca = 0, cb = 0;
  bool fast = false;
  // The 24 top bytes are classified four at a time (SWAR): no LDS window of words, no 24 dependent LDS reads.  (Round 4: ~1 000 instructions and three
  // memory phases -- ~15 k cycles in nearly every play batch, since a batch of 20-30 plays nearly always holds a won blind.)
  if (fresh) {
    const BgPk6 pk{q0.x, q0.y, q0.z, q0.w, q1.x, q1.y};
    BG_PROBE(18);
    const uint32_t H = 0x80808080u;
    const uint32_t addnc = (0x80u - (nc & 0x7fu)) * 0x01010101u;   // nc = 140..145 has bit 7 set: byte >= nc <=> bit 7 and low seven bits >= nc & 127
    uint32_t m3 = 0, m2 = 0, m8 = 0, m52 = 0;
#define BG_INV_CLASS(x, g) do { \
      const uint32_t ge192 = (x) & ((x) << 1);                                   /* bit 7: byte >= 192 (top two bits == 3) */ \
      const uint32_t ge208 = ge192 & (((x) << 2) | ((x) << 3));                  /* byte >= 208 = 0xD0 <=> byte >> 2 >= 52 */ \
      const uint32_t genc = (x) & (((x) & 0x7f7f7f7fu) + addnc);                 /* byte >= nc (no carry between bytes: 0x7f + 0x74 < 0x100) */ \
      m3 |= bg_movemask4(~ge192 & H) << (4 * (g)); \
      m2 |= bg_movemask4(~(x) & H) << (4 * (g));                                 /* byte < 128 (top two bits < 2) */ \
      m52 |= bg_movemask4(~ge208 & H) << (4 * (g)); \
      m8 |= bg_movemask4(~genc & H) << (4 * (g)); } while (0)
    BG_INV_CLASS(pk.a, 0); BG_INV_CLASS(pk.b, 1); BG_INV_CLASS(pk.c, 2); BG_INV_CLASS(pk.d, 3); BG_INV_CLASS(pk.e, 4); BG_INV_CLASS(pk.f, 5);
#undef BG_INV_CLASS
    auto nz = [H](uint32_t x) { return bg_movemask4((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & H); };   // bit k: byte k of x is not zero
    auto ne_mask = [&](uint32_t pv) {   // bit i: byte i != pv
      const uint32_t rep = pv * 0x01010101u;
      return nz(pk.a ^ rep) | (nz(pk.b ^ rep) << 4) | (nz(pk.c ^ rep) << 8) | (nz(pk.d ^ rep) << 12) | (nz(pk.e ^ rep) << 16) | (nz(pk.f ^ rep) << 20);
    };
    fast = true;
    uint32_t rem = 0xffffffu; // words not consumed yet
    int i0 = __ffs((int)(m3 & rem)) - 1; fast = fast && i0 >= 0; rem &= ~((2u << (i0 & 31)) - 1u);
    int i1 = __ffs((int)(m8 & rem)) - 1; fast = fast && i1 >= 0; rem &= ~((2u << (i1 & 31)) - 1u);
    p0 = (int)bg_pk_byte(pk, i1 & 31);
    uint32_t ne = ne_mask((uint32_t)p0);
    int i2 = __ffs((int)(m8 & ne & rem)) - 1; fast = fast && i2 >= 0; rem &= ~((2u << (i2 & 31)) - 1u);
    p1 = (int)bg_pk_byte(pk, i2 & 31);
    ne &= ne_mask((uint32_t)p1);
    int i3 = __ffs((int)(m8 & ne & rem)) - 1; fast = fast && i3 >= 0; rem &= ~((2u << (i3 & 31)) - 1u);
    p2 = (int)bg_pk_byte(pk, i3 & 31);
    int i4 = __ffs((int)(m2 & rem)) - 1; fast = fast && i4 >= 0; rem &= ~((2u << (i4 & 31)) - 1u);
    int i5 = __ffs((int)(m52 & rem)) - 1; fast = fast && i5 >= 0; rem &= ~((2u << (i5 & 31)) - 1u);
    int i6 = __ffs((int)(m52 & rem)) - 1; fast = fast && i6 >= 0;
    third_r = (int)(bg_pk_byte(pk, i0 & 31) >> 6); v = (int)(bg_pk_byte(pk, i4 & 31) >> 6);
    ca = (int)(bg_pk_byte(pk, i5 & 31) >> 2); cb = (int)(bg_pk_byte(pk, i6 & 31) >> 2);
    if (fast) e.s_idx += i6 + 1;
    w.s_start = e.s_idx; w.s_len = 0; w.g_len = 0; w.g_blk = -1;   // (as bg_sprefetch leaves the window: nothing of either stream is in it)
  }
  if (!fast) bg_sprefetch(d, env, e, w, BG_WIN); // shop.py:111-139: a rerolled stream (or the overflow block), or the one visit in thousands whose rejections outrun 24 words
  if (!fast) { // window too short for this lane's rejections (or no window): the plain loops
    third_r = (int)bg_randbelow<true>(d, env, e, w, 3u); // rng.choice([...]) is evaluated before the loop
    int guard = 0;
    p0 = (int)bg_randbelow<true>(d, env, e, w, nc);
    do { p1 = (int)bg_randbelow<true>(d, env, e, w, nc); } while (p1 == p0 && ++guard < 4096);
    do { p2 = (int)bg_randbelow<true>(d, env, e, w, nc); } while ((p2 == p0 || p2 == p1) && ++guard < 4096);
    v = (int)bg_randbelow<true>(d, env, e, w, 2u); // 0 'Voucher: Magic Trick' 600, 1 'Voucher: Minimalist' 750
    ca = (int)bg_randbelow<true>(d, env, e, w, 52u); cb = (int)bg_randbelow<true>(d, env, e, w, 52u); // randint(0, 51)
  }
  int third = PK_TAROT + third_r;
  int c_pack2 = third == PK_TAROT ? 600 : (third == PK_PLANET ? 900 : 1300);
  int32_t cost[9];
  uint32_t tp[9];
  cost[0] = (int32_t)(250.0 * mult); tp[0] = IT_PACK | (PK_STANDARD << 8);
  cost[1] = (int32_t)(500.0 * mult); tp[1] = IT_PACK | (PK_JOKER << 8);
  cost[2] = (int32_t)((double)c_pack2 * mult); tp[2] = IT_PACK | ((uint32_t)third << 8);
  int j0 = bg_candidate(sj, p0), j1 = bg_candidate(sj, p1), j2 = bg_candidate(sj, p2);
  cost[3] = (int32_t)((double)w.jt->cost[j0] * mult); tp[3] = IT_JOKER | ((uint32_t)j0 << 8);
  cost[4] = (int32_t)((double)w.jt->cost[j1] * mult); tp[4] = IT_JOKER | ((uint32_t)j1 << 8);
  cost[5] = (int32_t)((double)w.jt->cost[j2] * mult); tp[5] = IT_JOKER | ((uint32_t)j2 << 8);
  cost[6] = (int32_t)((v ? 750.0 : 600.0) * mult); tp[6] = IT_VOUCHER | ((uint32_t)v << 8);
  cost[7] = 40; tp[7] = IT_CARD | ((uint32_t)ca << 8);
  cost[8] = 40; tp[8] = IT_CARD | ((uint32_t)cb << 8);
  bg_shop_pack(sr, cost, tp);
  bg_shop_store(d, env, sr);
  e.shop_n = 9;
}

// balatro_env_2.py:1383-1392: the shop seed (one get_int on stream 2) and random.Random(seed) were produced ahead of
// time by the refill kernel; switching to the next ring slot IS "Shop(ante, player, seed=shop_seed)".
__device__ __forceinline__ void bg_generate_shop(const BgDev& d, int env, Env& e, RngWin& w, ShopRegs& sr) {
  if (e.s_ready <= 0) { atomicOr(d.err, BG_DEVERR_SHOPRING); return; }
  e.s_cur = (e.s_cur + 1 == d.KS) ? 0 : e.s_cur + 1;
  e.s_ready--; e.s_cons = (e.s_cons + 1) & 0xff;
  e.s_idx = 0;
  e.bflags = (e.bflags & ~BG_BF_SHOP_OVF) | BG_BF_SHOP_EXISTS;
  e.shop_ante = e.ante;
  e.shop_reroll_base = 50;
  w.need_inv = true; // Shop.__init__ -> _generate_inventory (shop.py:101): run by bg_env_dispatch, one code site for a wave whose
                     // lanes reach it from a won play, a skipped blind and a reroll
  e.shop_reroll_state = (int32_t)(50.0 * bg_shop_cost_mult(e, w.jt));
}

#define BG_FLAG_DEFER_ADV 0x40000000 // StepOut.flags, internal: the play was won, _advance_round is still to run
// balatro_env_2.py:1326-1381 (card-state gold money needs card states: not on this path)
template <bool CARDS = false>
__device__ __forceinline__ void bg_advance_round(const BgDev& d, int env, Env& e, RngWin& w, ShopRegs& sr) {
  if constexpr (CARDS) { // :1334-1343 every GOLD card held in hand pays $3
    int gold = 0;
#pragma unroll 1
    for (int i = 0; i < e.nhand; i++) if ((bg_cstate(d, env, bg_get8(e.hand, i)) & 0xfu) == 7u) gold += 3;
    e.money += gold;
  }
  if (e.boss_type) { e.money += 5; e.boss_type = 0; e.boss_types = 0; e.boss_cards = 0; e.face_down = 0; }
  e.round_chips = 0; e.best_hand = 0; e.hp_ante = 0;
  if (e.round == 3) {
    e.ante += 1; e.round = 1;
    if (e.ante > 100) return; // :1366-1367
  } else e.round += 1;
  e.money += 25 * e.round + (e.round == 3 ? 10 : 0);
  e.hands_left = 4; e.discards_left = 3;
  e.phase = 1;
  bg_generate_shop(d, env, e, w, sr);
}

// ---------------------------------------------------------------------------------------------------------
// Joker chain (unified_scoring.py:111-299 + complete_joker_effects.py:35-184), jokers by jokers.py id.
// Played cards are (rank, suit, chips) packed per card: rank | suit << 4 | chips << 8.
// ---------------------------------------------------------------------------------------------------------
struct JEff { int chips, mult; double x; };

// complete_joker_effects.py:35-129 (main 'scoring' phase) as a descriptor per joker id:
//   bits 0..4 condition index into a per-play condition bit set, 5..7 value kind, 8..23 constant.
// conditions: 0 always | 1 <= 3 scoring cards | 2 hands_left == 1 | 3 discards_left == 0 | 4..7 a scoring card of suit
//   C,D,H,S | 8 Blackboard | 9 Seeing Double | 10 Flower Pot | 11 a King played | 12 a Queen played | 13 / 14 / 15 the hand
//   is called 'Pair' / 'Three of a Kind' / 'Four of a Kind' | 16+t hand type == t | 31 never.
//   The env's names are 'One Pair' / 'Three Kind' / 'Four Kind' (balatro_env_2.py:674): conditions 13..15 are
//   never set on the step path, only by the operator-level entry point bg_score_hand_batch when a case uses the
//   balatro_sim-style names (unified_scoring.py:313-351 called directly).
// values: 0 +mult c | 1 +chips c | 2 x c | 3 +mult randint(0,23) (Misprint) | 4 +mult 3*len(jokers) | 5 +chips 30*discards
//   | 6 x 1.5**kings | 7 +mult 13*queens.
#define BG_JM(cond, vk, c) ((uint32_t)(cond) | ((uint32_t)(vk) << 5) | ((uint32_t)(c) << 8))
__device__ __forceinline__ uint32_t bg_jmain_desc(int id) {
  switch (id) {
    case 1: return BG_JM(0, 0, 4);      // Joker
    case 136: return BG_JM(0, 1, 250);  // Stuntman
    case 27: return BG_JM(0, 3, 0);     // Misprint
    case 38: return BG_JM(0, 0, 15);    // Gros Michel
    case 61: return BG_JM(0, 2, 3);     // Cavendish
    case 16: return BG_JM(1, 0, 20);    // Half Joker
    case 34: return BG_JM(0, 4, 0);     // Abstract Joker
    case 108: return BG_JM(2, 2, 3);    // Acrobat
    case 23: return BG_JM(3, 0, 15);    // Mystic Summit
    case 22: return BG_JM(0, 5, 0);     // Banner
    case 53: return BG_JM(0, 1, 104);   // Blue Joker: 2 * len(deck)
    case 97: return BG_JM(0, 0, 20);    // Popcorn
    case 50: return BG_JM(0, 1, 100);   // Ice Cream
    case 2: return BG_JM(5, 0, 3);      // Greedy (Diamonds)
    case 3: return BG_JM(6, 0, 3);      // Lusty (Hearts)
    case 4: return BG_JM(7, 0, 3);      // Wrathful (Spades)
    case 5: return BG_JM(4, 0, 3);      // Gluttonous (Clubs)
    case 6: return BG_JM(13, 0, 8);        // Jolly Joker ('Pair')
    case 7: return BG_JM(14, 0, 12);       // Zany Joker ('Three of a Kind')
    case 11: return BG_JM(13, 1, 50);      // Sly Joker
    case 12: return BG_JM(14, 1, 100);     // Wily Joker
    case 131: return BG_JM(13, 2, 2);      // The Duo
    case 132: return BG_JM(14, 2, 3);      // The Trio
    case 133: return BG_JM(15, 2, 4);      // The Family ('Four of a Kind')
    case 8: return BG_JM(16 + 2, 0, 10);   // Mad Joker (Two Pair)
    case 9: return BG_JM(16 + 4, 0, 12);   // Crazy Joker (Straight)
    case 10: return BG_JM(16 + 5, 0, 10);  // Droll Joker (Flush)
    case 13: return BG_JM(16 + 2, 1, 80);  // Clever Joker
    case 14: return BG_JM(16 + 4, 1, 100); // Devious Joker
    case 15: return BG_JM(16 + 5, 1, 80);  // Crafty Joker
    case 134: return BG_JM(16 + 4, 2, 3);  // The Order
    case 135: return BG_JM(16 + 5, 2, 2);  // The Tribe
    case 48: return BG_JM(8, 2, 3);     // Blackboard
    case 128: return BG_JM(9, 2, 2);    // Seeing Double
    case 122: return BG_JM(10, 2, 3);   // Flower Pot
    case 72: return BG_JM(11, 6, 0);    // Baron
    case 140: return BG_JM(12, 7, 0);   // Shoot the Moon
    default: return BG_JM(31, 0, 0);
  }
}
// name groups: 1 flush synergy (:853) | 2 pair/set synergy (:857-858) | 4 face synergy (:863) | 8 Trading Card | 16 Faceless
// Joker | 32 "discard joker" names (:1006)
__device__ __forceinline__ uint32_t bg_jflags(int id) {
  uint32_t f = 0;
  if (id == 113 || id == 18 || id == 69) f |= 1;
  if (id == 40 || id == 39 || id == 6 || id == 7) f |= 2;
  if (id == 33 || id == 104 || id == 42) f |= 4;
  if (id == 95) f |= 8;
  if (id == 57) f |= 16;
  if (id == 57 || id == 130 || id == 82 || id == 77) f |= 32;
  return f;
}

// complete_joker_effects.py:131-184 (per-card 'individual_scoring' phase).  Every effect is "rank in R" (any suit)
// or "suit == s" (any rank), so a joker's total over the played cards is (number of matching cards) x (chips, mult):
// the descriptor is  bits 0..14 ranks R | 16..18 suit+1 (0 = rank type) | 20..21 special (1 = 8 Ball: one more random()
// per played 8, :167; 2 = Bloodstone: needs the random() value, :161) | 22 x2 | 24..31 chips | 32..39 mult.
// Each (card, joker) pair consumes one random() = 2 words whatever the joker, because suit_effects (:157-162) is
// rebuilt on every call.
#define BG_JD(ranks, suit1, special, x2, chips, mult) \
  ((uint64_t)(ranks) | ((uint64_t)(suit1) << 16) | ((uint64_t)(special) << 20) | ((uint64_t)(x2) << 22) | ((uint64_t)(chips) << 24) | ((uint64_t)(mult) << 32))
#define BG_R(r) (1u << (r))
__device__ __forceinline__ uint64_t bg_jdesc(int id) {
  const uint32_t FACES = BG_R(11) | BG_R(12) | BG_R(13);
  switch (id) {
    case 31: return BG_JD(BG_R(2) | BG_R(3) | BG_R(5) | BG_R(8) | BG_R(14), 0, 0, 0, 0, 8);   // Fibonacci
    case 39: return BG_JD(BG_R(2) | BG_R(4) | BG_R(6) | BG_R(8) | BG_R(10), 0, 0, 0, 0, 4);   // Even Steven
    case 40: return BG_JD(BG_R(3) | BG_R(5) | BG_R(7) | BG_R(9) | BG_R(14), 0, 0, 0, 31, 0);  // Odd Todd
    case 41: return BG_JD(BG_R(14), 0, 0, 0, 20, 4);                                          // Scholar
    case 101: return BG_JD(BG_R(4) | BG_R(10), 0, 0, 0, 10, 4);                               // Walkie Talkie
    case 124: return BG_JD(BG_R(2), 0, 0, 0, 8, 0);                                           // Wee Joker
    case 26: return BG_JD(0, 0, 1, 0, 0, 0);                                                  // 8 Ball
    case 33: return BG_JD(FACES, 0, 0, 0, 30, 0);                                             // Scary Face
    case 104: return BG_JD(FACES, 0, 0, 0, 0, 5);                                             // Smiley Face
    case 147: return BG_JD(BG_R(12) | BG_R(13), 0, 0, 1, 0, 0);                               // Triboulet
    case 118: return BG_JD(0, 3 + 1, 0, 0, 50, 0);                                            // Arrowhead (Spades)
    case 119: return BG_JD(0, 0 + 1, 0, 0, 0, 7);                                             // Onyx Agate (Clubs)
    case 117: return BG_JD(0, 2 + 1, 2, 1, 0, 0);                                             // Bloodstone (Hearts, p = .5)
    default: return 0ull;                                                                     // incl. Rough Gem ($ only)
  }
}

// fill the workgroup's LDS tables (every thread of the block must call it; ends with a barrier)
__device__ __forceinline__ void bg_tables_init(JTables* t) {
  for (int id = threadIdx.x; id < 152; id += blockDim.x) {
    t->jd[id] = bg_jdesc(id);
    t->jm[id] = bg_jmain_desc(id);
    { uint64_t rm = 0; uint32_t r = (uint32_t)bg_jdesc(id) & 0x7fffu; for (int k = 0; k < 15; k++) if ((r >> k) & 1u) rm |= 0xfull << (4 * k); t->jr[id] = rm; }
    t->jf[id] = (uint8_t)bg_jflags(id);
    t->cost[id] = id < 151 ? BG_JOKER_COST[id] : 0;
    if (id < 101) t->pow115[id] = BG_POW115[id];
    if (id < 16) { t->pow15[id] = BG_POW15[id]; t->pow08[id] = id < 9 ? BG_POW08[id] : 0.0; }
    if (id < 64) t->inv[id] = id >= 2 ? (uint32_t)(0x100000000ull / (uint64_t)id) : 0xffffffffu;
  }
  __syncthreads();
}
// the same tables copied from the handle's prebuilt copy: a launch of one step should not pay for 152 switch statements
__device__ __forceinline__ void bg_tables_load(JTables* t, const uint32_t* __restrict__ g) {
  uint32_t* w = (uint32_t*)t;
  for (int i = threadIdx.x; i < (int)(sizeof(JTables) / 4); i += blockDim.x) w[i] = g[i];
  __syncthreads();
}
__device__ __forceinline__ uint32_t bg_joker_flags(const Env& e, lds_JTables* jt) {
  uint32_t f = 0;
#pragma unroll 1
  for (int j = 0; j < e.njokers; j++) f |= jt->jf[bg_get8(e.jokers, j)];
  return f;
}

// boss_blinds.py:343-378 on_hand_drawn as applied by balatro_env_2.py:936-948
template <class DK>
__device__ __forceinline__ void bg_boss_on_hand_drawn(const BgDev& d, int env, Env& e, RngWin& w, const DK& dk) {
  uint32_t fd = 0;
  int n = e.nhand;
  int h0 = -1, h1 = -1;
  switch (e.boss_type) {
    case 1: // The Hook: random.sample(range(n), 2), n <= 21 -> pool method
      if (n >= 2) {
        int j0 = (int)bg_randbelow<false>(d, env, e, w, (uint32_t)n);
        int j1 = (int)bg_randbelow<false>(d, env, e, w, (uint32_t)(n - 1));
        h0 = j0;                                  // pool[j] = j initially
        h1 = (j1 == j0) ? (n - 1) : j1;           // pool[j0] was overwritten with pool[n-1]
      }
      break;
    case 3: // The Wheel
      bg_gprefetch(d, env, e, w, 2 * n);
#pragma unroll 1
      for (int i = 0; i < n; i++) if (bg_grandom(d, env, e, w) < 1.0 / 7.0) fd |= 1u << i;
      break;
    case 4: if (e.bflags & BG_BF_FIRST_HAND) fd = (1u << n) - 1; break;  // The House
    case 5: // The Mark
#pragma unroll 1
      for (int i = 0; i < n; i++) { int rk = (bg_card(d, env, dk, bg_get8(e.hand, i)) >> 2) + 2; if (rk >= 11 && rk <= 13) fd |= 1u << i; }
      break;
    case 6: if (!(e.bflags & BG_BF_FIRST_HAND)) fd = (1u << n) - 1; break; // The Fish
    default: break;
  }
  e.face_down = fd & 0xffu;
  if (h0 >= 0) { // pop in descending position order (:946-948)
    int a = h0 > h1 ? h0 : h1, b = h0 > h1 ? h1 : h0;
    if (a < e.nhand) { e.hand = bg_del8(e.hand, a); e.nhand--; }
    if (b < e.nhand) { e.hand = bg_del8(e.hand, b); e.nhand--; }
  }
}

// unified_scoring.py:174-244: the joker chain of one scored hand -- individual phase (card-major, joker-minor), then the
// main phase in joker order -- with the eager RNG draws of complete_joker_effects.py:42,161.  What it needs of
// the hand is a few small histograms (ChainIn).  GENERAL = false is the step path (bg_step_play_hand: scoring cards == cards,
// the env's hand-type names, money reported to a throw-away dict); GENERAL = true is the operator-level entry point
// bg_score_hand_batch (UnifiedScorer.score_hand called directly: either name style, `cards` != `scoring_cards`, deck length and
// money as given).  Both instantiate THIS function.
struct ChainIn {
  uint64_t phist;   // 15 x 4-bit counts per rank of the SCORING cards (rank 0 = a STONE card)
  uint64_t pcodes;  // their card codes, one byte each (rank and suit of a STONE card are ignored)
  uint32_t scnt;    // 5 x 4-bit counts per suit of the scoring cards (C, D, H, S, 'Stone')
  uint32_t stone;   // bit per scoring-card index
  int n, ht;        // scoring cards, hand type
  int kings, queens; // of context['cards'] (Baron :116-120, Shoot the Moon :122-126)
  bool all_black;   // GENERAL: every card of context['cards'] is a Spade or a Club (Blackboard)
  int deck_len, style; // GENERAL: len(game_state['deck']), 1 = 'Pair' / 'Three of a Kind' / 'Four of a Kind' names
};
// The main phase's 12 candidate words, requested EARLY (bg_step_play_hand issues the loads before it gathers and classifies the
// cards, so their HBM round trip runs beside ~5k cycles of LDS work instead of after it): valid when `skip` equals the number
// of words the individual phase turns out to consume.
struct ChainPeek { uint32_t mw[12]; uint32_t avail; int skip; bool ok; };
// One joker's share of the individual phase: matching cards = sum of the histogram nibbles selected by the rank mask, or the
// suit's count; (chips, mult) = count x the descriptor's constants; x2 per matching card for Triboulet.  sp = the descriptor's
// "special" field (1 = 8 Ball, 2 = Bloodstone: settled card by card from the RNG words).
struct ChainJ { int ic, im, xexp; uint32_t sp; };
__device__ __forceinline__ ChainJ bg_chain_joker(uint64_t dsc, uint64_t jr, uint64_t phist, uint32_t scnt) {
  ChainJ r;
  r.sp = (uint32_t)(dsc >> 20) & 3u;
  uint64_t x = phist & jr;
  uint32_t y = (uint32_t)(x & 0x0f0f0f0f0f0f0f0full) + (uint32_t)((x >> 4) & 0x0f0f0f0f0f0f0f0full) +
               (uint32_t)((x & 0x0f0f0f0f0f0f0f0full) >> 32) + (uint32_t)(((x >> 4) & 0x0f0f0f0f0f0f0f0full) >> 32);
  y += y >> 16;
  int cnt = (int)((y + (y >> 8)) & 0xffu);
  uint32_t suit1 = (uint32_t)(dsc >> 16) & 7u;
  if (suit1) cnt = (int)((scnt >> (4 * (suit1 - 1))) & 0xfu);
  if (r.sp == 2u) cnt = 0; // Bloodstone is settled card by card
  r.ic = cnt * (int)((dsc >> 24) & 0xffu); r.im = cnt * (int)((dsc >> 32) & 0xffu);
  r.xexp = ((dsc >> 22) & 1u) ? cnt : 0;
  return r;
}
// Bloodstone on a Heart: x2 iff the pair's random() < 0.5.  Pair (c, jb) sits 2*(c*nj + jb) words ahead of the cursor, plus 2
// for every extra 8-Ball draw that precedes it in card-major, joker-minor order: `eights` played 8s among cards 0..c-1 times the
// nb8 8-Ball jokers owned, plus -- when card c is an 8 itself -- the 8 Balls in the slots before jb (m8 = bit per 8-Ball slot).  An id
// can be owned more than once (Ankh copies a joker), so both kinds are SETS of slots.  -1: no word to read.
__device__ __forceinline__ int bg_chain_blood_off(int c, int code, bool st, int n, int nj, int jb, uint32_t m8, int nb8, bool blood, int eights) {
  const int rk = (code >> 2) + 2;
  return (blood && c < n && !st && (code & 3) == 2) ? 2 * (c * nj + jb) + 2 * (eights * nb8 + ((!st && rk == 8) ? __popc(m8 & ((1u << jb) - 1u)) : 0)) : -1;
}

// unified_scoring.py:216-244 main phase, joker order; one randint(0, 23) per joker (mw = the 12 tempered words that follow the
// individual phase's draws, avail = which of them the ring holds).
template <bool GENERAL, class DK>
__device__ __forceinline__ void bg_chain_main(const BgDev& d, int env, Env& e, RngWin& w, const ChainIn& in, const uint32_t (&dms)[5],
                                              const uint32_t (&mw)[12], uint32_t avail, int64_t& chips, int64_t& mult, double& x_mult) {
  BG_PROBE_BEGIN();
  const int nj = e.njokers, n = in.n, ht = in.ht;
  const uint32_t scnt = in.scnt;
  const int kings = in.kings, queens = in.queens;
  // bit 4 = 'Stone': complete_joker_effects.py:98-114 compare suit STRINGS, so a STONE card is a fifth kind of suit for
  // Blackboard / Seeing Double / Flower Pot (and no suit at all for the four suit jokers)
  uint32_t suits = ((scnt & 0xfu) ? 1u : 0u) | ((scnt & 0xf0u) ? 2u : 0u) | ((scnt & 0xf00u) ? 4u : 0u) | ((scnt & 0xf000u) ? 8u : 0u) |
                   ((scnt & 0xf0000u) ? 16u : 0u);
  // Blackboard (complete_joker_effects.py:98-102) looks at context['cards'], the other suit jokers at the scoring cards; on
  // the step path the two lists are the same one (balatro_env_2.py:683-689)
  const bool blackboard = GENERAL ? in.all_black : (suits & ~9u) == 0;
  uint32_t cond = 1u | (n <= 3 ? 2u : 0u) | (e.hands_left == 1 ? 4u : 0u) | (e.discards_left == 0 ? 8u : 0u) | ((suits & 15u) << 4) |
                  (blackboard ? 1u << 8 : 0u) | (((suits & 1u) && __popc(suits) > 1) ? 1u << 9 : 0u) |
                  (__popc(suits) == 4 ? 1u << 10 : 0u) | (kings > 0 ? 1u << 11 : 0u) | (queens > 0 ? 1u << 12 : 0u) | (1u << (16 + ht));
  if constexpr (GENERAL) // 'Pair' / 'Three of a Kind' / 'Four of a Kind' (complete_joker_effects.py:64-80) only exist in the sim-style names
    if (in.style == 1) cond |= (ht == 1 ? 1u << 13 : 0u) | (ht == 3 ? 1u << 14 : 0u) | (ht == 7 ? 1u << 15 : 0u);
  double baron = w.jt->pow15[kings];
  BG_PROBE(8);
  BG_PROBE(13);
  // The nj draws are `_randbelow(24)`: 5-bit words, rejected when >= 24.  Instead of a rejection loop per joker (a wave
  // iterates until its unluckiest lane is done), look at the next 12 words at once: the j-th ACCEPTED word is joker
  // j's draw.  Only Misprint uses the value.
  // (Round 5: everything below is written as masks and selects on purpose.  As nested `?:` under `if (j < nj)` / `if (ok)` the compiler turned the five
  //  joker slots into 24 branches and 35 exec-mask saves -- 255 instructions on one source line, ~7 k cycles per play for what is ~150 arithmetic
  //  instructions; a slot beyond njokers holds id 0, whose descriptor's condition (31) is never set, so no slot needs a guard.)
  uint32_t mis_of[5] = {0, 0, 0, 0, 0};
  {
    bg_gnorm(d, e);
    uint32_t acc = 0, r5a = 0, r5b = 0; // the twelve 5-bit candidates: six per word
#pragma unroll
    for (int i = 0; i < 12; i++) {
      const uint32_t r5 = mw[i] >> 27;
      acc |= (r5 < 24u ? 1u : 0u) << i;
      if (i < 6) r5a |= r5 << (5 * i); else r5b |= r5 << (5 * (i - 6));
    }
    // usable only up to the last word the ring holds, and only if the nj-th accepted word lies inside
    uint32_t usable = acc & avail;
    bool fast = avail == 0xfffu ? __popc(acc) >= nj : false;
    if (!fast && avail) { // ring ends inside the 12 words: accepted words must all come before the first missing one
      const int first_missing = __ffs((int)(~avail & 0xfffu)) - 1;
      usable = acc & ((1u << first_missing) - 1u);
      fast = __popc(usable) >= nj;
    }
    if (fast) {
      uint32_t m = usable;
      int last = 0;
#pragma unroll
      for (int j = 0; j < 5; j++) {
        const bool on = j < nj;
        const int pos = __ffs((int)m) - 1;            // (m != 0 while j < nj: popc(usable) >= nj)
        last = on ? pos : last;
        m = on ? (m & (m - 1u)) : m;
        const uint32_t word = pos < 6 ? r5a : r5b;
        mis_of[j] = (word >> (5u * (uint32_t)(pos < 6 ? pos : pos - 6) & 31u)) & 0x1fu;   // (only looked at by a Misprint in slot j)
      }
      e.g_idx += last + 1;
    } else {
#pragma unroll 1
      for (int j = 0; j < nj; j++) {
        uint32_t v = bg_randbelow<false>(d, env, e, w, 24u);
#pragma unroll
        for (int q = 0; q < 5; q++) if (q == j) mis_of[q] = v;
      }
    }
  }
  const uint32_t k4 = 3u * (uint32_t)nj, k5 = 30u * (uint32_t)e.discards_left, k7 = 13u * (uint32_t)queens;
  uint32_t blue = 104u;   // Blue Joker: 2 * len(deck) = 104 until Immolate / Cryptid change the deck
  if constexpr (GENERAL) blue = 2u * (uint32_t)in.deck_len;
  else if constexpr (DK::kCards) blue = 2u * (uint32_t)(52 - e.ndrop + e.nfo);
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const uint32_t dm = dms[j];
    const uint32_t okm = 0u - ((cond >> (dm & 31u)) & 1u);          // all ones when the joker's condition holds
    const uint32_t vk = (dm >> 5) & 7u, c = dm >> 8;
    const uint32_t is0 = 0u - (uint32_t)(vk == 0u), is1 = 0u - (uint32_t)(vk == 1u), is3 = 0u - (uint32_t)(vk == 3u), is4 = 0u - (uint32_t)(vk == 4u),
                   is5 = 0u - (uint32_t)(vk == 5u), is7 = 0u - (uint32_t)(vk == 7u);
    const uint32_t madd = (c & is0) | (mis_of[j] & is3) | (k4 & is4) | (k7 & is7);
    const uint32_t c1 = (c == 104u) ? blue : c;                       // (+chips 104 is the Blue Joker's descriptor)
    const uint32_t cadd = (c1 & is1) | (k5 & is5);
    chips += (int64_t)(cadd & okm); mult += (int64_t)(madd & okm);
    double xf = vk == 2u ? (double)c : (vk == 6u ? baron : 1.0);
    xf = okm ? xf : 1.0;                                              // (a factor of exactly 1.0 leaves x_mult's bits alone)
    x_mult *= xf;
  }
  BG_PROBE(14);
}

template <bool GENERAL, class DK>
__device__ __forceinline__ void bg_joker_chain(const BgDev& d, int env, Env& e, RngWin& w, const ChainIn& in, int64_t& chips,
                                               int64_t& mult, double& x_mult, int& money, const ChainPeek* pre = nullptr) {
  BG_PROBE_BEGIN();
  // unified_scoring.py:174-209 individual phase.  Totals do not depend on the (card-major, joker-minor) order: chips
  // and mult add up, and every x factor is exactly 2.0.
  const int nj = e.njokers, n = in.n;
  const uint64_t phist = in.phist, pcodes = in.pcodes;
  const uint32_t scnt = in.scnt, stone = in.stone;
  int ic = 0, im = 0, xexp = 0;
  uint32_t m8 = 0, mb = 0; // slots holding an 8 Ball / a Bloodstone (an id may be owned twice: Ankh)
  // all table reads first (independent LDS reads), then straight-line arithmetic per joker slot
  uint64_t jds[5], jrs[5];
  uint32_t dms[5];
#pragma unroll
  for (int j = 0; j < 5; j++) {
    int id = j < nj ? (int)((e.jokers >> (8 * j)) & 0xff) : 0;
    jds[j] = w.jt->jd[id]; jrs[j] = w.jt->jr[id]; dms[j] = w.jt->jm[id];
  }
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const ChainJ cj = bg_chain_joker(jds[j], jrs[j], phist, scnt);
    if (cj.sp == 1u) m8 |= 1u << j;
    if (cj.sp == 2u) mb |= 1u << j;
    ic += cj.ic; im += cj.im; xexp += cj.xexp;
    if constexpr (GENERAL) if (j < nj && (int)((e.jokers >> (8 * j)) & 0xff) == 116) money += (int)((scnt >> 4) & 0xfu); // Rough Gem: $1 per Diamond (:160)
  }
  BG_PROBE(7);
  bg_gnorm(d, e);
  const int nb8 = __popc(m8);
  const int n8 = nb8 ? (int)((phist >> 32) & 0xf) : 0; // 8 Ball: one extra random() per played 8 and per 8 Ball owned (:167)
  int consumed = 2 * n * nj + 2 * n8 * nb8;
  // Every RNG word the chain looks at is requested in ONE batch of independent loads per Bloodstone owned (one, practically
  // always): its word per played Heart, and the 12 words that follow the individual phase's `consumed` (eagerly drawn, never
  // looked at) words, where the main phase's randint draws will fall (5 accepted among 12 words fails once in ~3000 plays: then
  // the loop).
  // random() < 0.5 for random() = ((a >> 5) * 2**26 + (b >> 6)) / 2**53 is decided by the top bit of the FIRST word
  // alone ((a >> 5) < 2**26), so one word per Heart is read and no float arithmetic is needed.
  uint32_t mw[12];
  uint32_t avail = 0; // main-phase words the ring already holds
  uint32_t mm = ((scnt >> 8) & 0xfu) ? mb : 0u; // Bloodstones with a Heart to look at
  // An 8 Ball owner's main-phase words could not be requested early (their place depends on the 8s played): request them NOW, raw, before the
  // Bloodstone words below are waited for -- a batch nearly always holds an owner of each, and the two HBM round trips then run side by side
  // instead of one after the other (2.9 k + 2.8 k cycles per workgroup-step, probes 2 / 3)
  const bool pre_ok = pre && pre->ok && pre->skip == consumed;
  if (pre_ok) {
#pragma unroll
    for (int i = 0; i < 12; i++) mw[i] = pre->mw[i];
    avail = pre->avail;
  } else bg_gpeek12_raw(d, env, e, consumed, mw, avail);
#pragma unroll 1
  while (mm) {
    const int jb = __ffs((int)mm) - 1;
    mm &= mm - 1u;
    int boff[8];
    int eights = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
      int code = (int)((pcodes >> (8 * c)) & 0xff);
      int rk = (code >> 2) + 2;
      const bool st = (stone >> c) & 1u; // a STONE card has no rank and no suit for the jokers
      boff[c] = bg_chain_blood_off(c, code, st, n, nj, jb, m8, nb8, true, eights);
      if (c < n && nb8 && !st && rk == 8) eights++;
    }
    // eight loads side by side (unconditional, from a harmless address where there is no word), looked at together: as eight `if (...) bg_gpeek()`
    // every Heart was a branch region with its own wait -- up to five HBM round trips in a row inside a play batch
    uint32_t ra[8];
    uint32_t vm = 0;
    bool bad = false;
    // (requesting these eight words EARLY -- right behind the gather, so that their round trip runs beside the classification -- was measured in round 5:
    //  -2 % at 20 steps, -1 % at 372, profiles/r05/play_path_ab.txt; fetched at the play's start by LDS-DMA, with no register held, in round 6: +-0,
    //  profiles/r06/lds_dma_prefetch.txt)
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const bool v = boff[c] >= 0;
      bool okc;
      const uint32_t* p = bg_gpeek_addr(d, env, e, v ? boff[c] : 0, okc);
      ra[c] = *p;
      bad = bad || (v && !okc);
      vm |= (v && okc ? 1u : 0u) << c;
    }
    if (bad) atomicOr(d.err, BG_DEVERR_GSTREAM);
#pragma unroll
    for (int c = 0; c < 8; c++) xexp += (int)(((vm >> c) & 1u) & ((bg_temper(ra[c]) >> 31) ^ 1u));
  }
  BG_PROBE(2);
#pragma unroll
  for (int i = 0; i < 12; i++) mw[i] = bg_temper(mw[i]); // requested raw (the first use of a loaded value is the wait)
  bg_gskip(d, e, consumed);
  BG_PROBE(3);
  chips += ic; mult += im;
  x_mult *= (double)(1ull << xexp);
  bg_chain_main<GENERAL, DK>(d, env, e, w, in, dms, mw, avail, chips, mult, x_mult);
}

// ---------------------------------------------------------------------------------------------------------
// PLAY_HAND  balatro_env_2.py:645-960
// ---------------------------------------------------------------------------------------------------------
template <class DK>
__device__ __forceinline__ void bg_step_play_hand(const BgDev& d, int env, Env& e, RngWin& w, ShopRegs& sr, const DK& dk, StepOut& o) {
  BG_PROBE_BEGIN();
  // :650-660 selected cards in selection order.  Everything the scorer needs is kept as small histograms:
  //   phist  15 x 4-bit counts per rank (2..14), scnt 4 x 4-bit counts per suit, pcodes the card codes by play index,
  //   pmask  bit per played DECK index (boss_blinds.py:472 id(card)), chip_sum (cards.py:52-60)
  uint64_t phist = 0, pcodes = 0, pmask = 0;
  uint32_t scnt = 0;
  int n = 0, chip_sum = 0;
  // the joker chain's main-phase words sit 2 * n * njokers (+ 2 per 8 Ball draw) words ahead: known before any card is
  // looked at unless an 8 Ball is owned -- request them now
  ChainPeek pre; pre.ok = false; pre.skip = 0; pre.avail = 0;
  if ((d.flags & 1u) && e.njokers > 0) {
    int npre = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) npre += (i < e.nsel && bg_get8(e.sel, i) < e.nhand) ? 1 : 0;
    bool ball = false;
#pragma unroll
    for (int j = 0; j < 5; j++) ball = ball || (j < e.njokers && bg_get8(e.jokers, j) == 26);
    if (!ball) {
      bg_gnorm(d, e);
      pre.skip = 2 * npre * e.njokers;
      bg_gpeek12_raw(d, env, e, pre.skip, pre.mw, pre.avail);
      pre.ok = true;
    }
  }
  // A play that beats the blind generates a shop, whose fresh inventory reads the tail of the next pre-seeded shop slot (32 bytes of one line): touch it
  // now (a dword load nobody waits for) so that it comes from L2, not from HBM, if the play wins
  uint32_t touch0 = 0;
  if (e.s_ready > 0) touch0 = bg_sblock(d, env, (e.s_cur + 1 == d.KS) ? 0 : e.s_cur + 1)[BG_SW_PK];
  // card states (CardAdapter.to_scoring_format :287-325): BONUS +30, STONE +50 and no rank / suit, FOIL +50; the seals
  // and the GLASS / LUCKY rolls are settled after the scorer (:703-734)
  uint32_t stone = 0;                      // bit per play index
  uint64_t dhist = 0;                      // rank histogram of the DECK cards played (boss Plant, face synergy)
  uint64_t cst = 0;                        // enh | seal << 4 per play index, one byte each
  const DeckHead dh = bg_deck_head(d, env, dk); // deck[0..15]: the cards under the hand's indexes AND the classifier's positions
#pragma unroll 1
  for (int i = 0; i < e.nsel; i++) {
    int pos = bg_get8(e.sel, i);
    if (pos < e.nhand) {
      int ci = bg_get8(e.hand, pos);
      int code = bg_card_h(d, env, dk, dh, ci);
      int bonus = 0;
      bool is_stone = false;
      if constexpr (DK::kCards) {
        const uint32_t cs = bg_cstate(d, env, ci);
        const uint32_t enh = cs & 0xfu;
        bonus = enh == 1u ? 30 : (enh == 6u ? 50 : 0);
        if (((cs >> 4) & 0xfu) == 1u) bonus += 50;
        is_stone = enh == 6u;
        cst |= (uint64_t)(enh | (((cs >> 8) & 0xfu) << 4)) << (8 * n);
        dhist += 1ull << (4 * ((code >> 2) + 2));
      }
      if (is_stone) { phist += 1ull; scnt += 1u << 16; stone |= 1u << n; } // rank 0, suit 'Stone'
      else { phist += 1ull << (4 * ((code >> 2) + 2)); scnt += 1u << (4 * (code & 3)); }
      pcodes |= (uint64_t)code << (8 * n);
      pmask |= 1ull << ci;
      chip_sum += bg_card_chips(code) + bonus;
      n++;
      e.highlighted |= 1u << pos; // :663-666 highlights are never cleared by a play
    }
  }
  if constexpr (!DK::kCards) dhist = phist;
  int jacks = (int)((phist >> 44) & 0xf), queens = (int)((phist >> 48) & 0xf), kings = (int)((phist >> 52) & 0xf);
  // the reward's face synergy (:862) and The Plant (boss_blinds.py:425) look at the deck card, stone or not
  const int djqk = (int)(((dhist >> 44) & 0xf) + ((dhist >> 48) & 0xf) + ((dhist >> 52) & 0xf));
  int faces = djqk + (int)((dhist >> 56) & 0xf); // rank >= 11 counts the ace (:862)
  BG_PROBE(5);
  // :669-671 classify deck[p] for highlighted POSITIONS p
  uint64_t hc = 0; int nh = 0;
#pragma unroll 1
  for (uint32_t hm = e.highlighted & 0xffffu; hm; hm &= hm - 1) {
    int p = __ffs((int)hm) - 1;
    if (nh < 8) hc |= (uint64_t)bg_card_h(d, env, dk, dh, p) << (8 * nh);
    nh++;
  }
  int ht = bg_classify(hc, nh < 8 ? nh : 8);
  BG_PROBE(6);
  // :677-680 boss restrictions (boss_blinds.py:380-407)
  if (e.boss_type) {
    int err = 0;
    if (e.boss_type == 7 && n != 5) err = 2;
    else if (e.boss_type == 12 && (e.boss_types & (1u << ht))) err = 3;
    else if (e.boss_type == 13 && e.boss_types && !(e.boss_types & (1u << ht))) err = 4;
    else if (e.boss_type == 25 && n < e.boss_req) err = 5;
    if (err) {
      o.reward = -1.0; o.error = err;
      // what the reference's message names (boss_blinds.py:393,399,405): the hand type played again, the ONE allowed type, the cards required
      o.aux = err == 3 ? ht : (err == 4 ? __ffs((int)e.boss_types) - 1 : (err == 5 ? e.boss_req : 0));
      return;
    }
  }
  // :683-692 UnifiedScorer.score_hand
  int level = bg_level(e, ht);
  int bchips, bmult;
  bg_hand_base(ht, level, bchips, bmult);
  int64_t chips = bchips + chip_sum, mult = bmult;
  double x_mult = 1.0;
  BG_PROBE(1);
  if ((d.flags & 1u) && e.njokers > 0) { // scorer-level joker names (BG_FLAG_SCORER_JOKERS); dict jokers are inert (Q6)
    ChainIn in;
    in.phist = phist; in.pcodes = pcodes; in.scnt = scnt; in.stone = stone; in.n = n; in.ht = ht; in.kings = kings; in.queens = queens;
    in.all_black = false; in.deck_len = 52; in.style = 0;
    int chain_money = 0; // game_state is state.to_dict(): the scorer's money goes nowhere (unified_scoring.py:292-294)
    bg_joker_chain<false, DK>(d, env, e, w, in, chips, mult, x_mult, chain_money, &pre);
  }
  BG_PROBE(15);
  int64_t final_score = (int64_t)((double)(chips * mult) * x_mult); // unified_scoring.py:286
  if (o.bd_dst) { // info['score_breakdown'] (bg_step only): final_chips, final_mult, final_x_mult, card_chips, base_chips, base_mult, money_gained, 0
    int gems = 0; // money_gained: Rough Gem pays $1 per Diamond scored (complete_joker_effects.py:160); the env drops it (unified_scoring.py:292-294)
    if ((d.flags & 1u) && e.njokers > 0) {
#pragma unroll
      for (int j = 0; j < 5; j++) gems += (j < e.njokers && bg_get8(e.jokers, j) == 116) ? 1 : 0;
    }
    double2* q = (double2*)o.bd_dst;
    q[0] = make_double2((double)chips, (double)mult); q[1] = make_double2(x_mult, (double)chip_sum);
    q[2] = make_double2((double)bchips, (double)bmult); q[3] = make_double2((double)(gems * (int)((scnt >> 4) & 0xfu)), 0.0);
  }
  int retriggers = 0;
  if constexpr (DK::kCards) {
    // :703-734 per played card: GLASS rolls once, LUCKY twice on the 'card_enhancement' stream (the second roll < 0.0667
    // pays $20); GOLD seal $3, RED seal a retrigger, BLUE seal a planet of the played hand type
    int extra_money = 0, blue = 0;
#pragma unroll 1
    for (int c = 0; c < n; c++) {
      const uint32_t b = (uint32_t)(cst >> (8 * c)) & 0xffu, enh = b & 0xfu, seal = b >> 4;
      if (enh == 4u) (void)bg_lazy_random(d.cardmt + (size_t)env * BG_MTS);
      else if (enh == 8u) {
        (void)bg_lazy_random(d.cardmt + (size_t)env * BG_MTS);
        if (bg_lazy_random(d.cardmt + (size_t)env * BG_MTS) < 0.0667) extra_money += 20;
      }
      if (seal == 1u) extra_money += 3; else if (seal == 2u) retriggers++; else if (seal == 3u) blue++;
    }
    if (blue && e.ncons < 2) { // :732-734 guard against the un-grown list, :765-767 append under the same guard
      const int PLANET_ID[12] = {38, 30, 31, 32, 33, 34, 35, 36, 37, 39, 40, 41};
#pragma unroll 1
      for (int q = 0; q < blue; q++)
        if (e.ncons < 2) { if (e.ncons == 0) e.cons0 = PLANET_ID[ht]; else e.cons1 = PLANET_ID[ht]; e.ncons++; }
    }
    // :741-742 STEEL cards held in hand and not selected: x1.5 each
    double steel = 1.0;
    uint32_t selpos = 0;
#pragma unroll 1
    for (int q = 0; q < e.nsel; q++) { int sp = bg_get8(e.sel, q); if (sp < e.nhand) selpos |= 1u << sp; }
#pragma unroll 1
    for (int i = 0; i < e.nhand; i++)
      if (!((selpos >> i) & 1u) && (bg_cstate(d, env, bg_get8(e.hand, i)) & 0xfu) == 5u) steel *= 1.5;
    final_score = (int64_t)((double)final_score * steel);
    e.money += extra_money;
  }
  BG_PROBE(16);
  // :745-755 boss scoring ratio (boss_blinds.py:409-445)
  if (e.boss_type) {
    int64_t mc = bchips, mm = bmult;
    if (e.boss_type == 21) { mc = mc / 2; mm = mm / 2; }
    else if (e.boss_type == 22) mc = 0;
    else if (e.boss_type == 23) { mc = (int64_t)((double)mc * 0.75); mm = (int64_t)((double)mm * 0.75); }
    int deb = 0;
    if (e.boss_type == 14) deb = djqk;                                          // The Plant: face cards (deck rank)
    else if (e.boss_type == 24) deb = n;                                        // The Violet: every card
    else if (e.boss_type == 16) deb = __popcll(e.boss_cards & pmask);           // The Pillar: played before
    if (deb > 0) {
      double pen = w.jt->pow08[deb];
      mc = (int64_t)((double)mc * pen);
      mm = (int64_t)((double)mm * pen);
    }
    double cr = (double)mc / (double)bchips, mr = (double)mm / (double)bmult;
    final_score = (int64_t)((double)final_score * cr * mr);
  }
  // :758-759 int(final * (1 + 0.5 * red seals))
  if constexpr (DK::kCards) final_score = (int64_t)((double)final_score * (1.0 + (double)retriggers * 0.5));
  BG_PROBE(17);
  // :775-786
  int64_t need1 = e.chips_needed > 1 ? e.chips_needed : 1;
  double old_progress = (double)e.round_chips / (double)need1;
  if (old_progress > 1.0) old_progress = 1.0;
  e.round_chips += final_score;
  e.chips_scored += final_score;
  e.hp_total++; e.hp_ante++;
  if (final_score > e.best_hand) e.best_hand = final_score;
  // engine.hand_play_counts[hand_type] += 1 (write-only statistic: a no-return atomic keeps it off the wait path)
  atomicAdd(((uint32_t*)&d.cold[(size_t)(ht >> 2) * d.N + env]) + (ht & 3), 1u);
  // :789-794 boss on_hand_scored (boss_blinds.py:480-507); Tooth/Serpent mutate a throw-away dict
  if (e.boss_type) {
    e.boss_types |= 1u << ht;
    e.bflags &= ~BG_BF_FIRST_HAND;
    e.boss_hp++;
    if (e.boss_type == 16) e.boss_cards |= pmask;
    if (e.boss_type == 25) e.boss_req = e.boss_req + 1 > 7 ? 7 : e.boss_req + 1;
  }
  e.nsel = 0; e.sel = 0; // :797
  BG_PROBE(10);
  // :799-892 reward shaping (float64, left to right)
  double new_progress = (double)e.round_chips / (double)need1;
  if (new_progress > 1.0) new_progress = 1.0;
  double progress_reward = 15.0 * new_progress;
  double milestone = 0.0;
  if (old_progress < 0.25 && 0.25 <= new_progress) milestone = 5.0;
  else if (old_progress < 0.5 && 0.5 <= new_progress) milestone = 10.0;
  else if (old_progress < 0.75 && 0.75 <= new_progress) milestone = 15.0;
  else if (old_progress < 1.0 && 1.0 <= new_progress) milestone = 25.0;
  double score_reward;
  if (e.ante <= 3) { score_reward = (double)final_score / 100.0; if (score_reward > 10.0) score_reward = 10.0; }
  else {
    int64_t s = final_score > 1 ? final_score : 1;
    score_reward = s >= BG_LOG10_N ? 10.0 : 3.0 * BG_LOG10[s]; // 3.0 * np.log10(s), table pinned on the reference platform
    if (score_reward > 10.0) score_reward = 10.0;
  }
  double hq;
  switch (ht) {
    case 0: hq = 0.1; break; case 1: hq = 0.5; break; case 2: hq = 1.0; break; case 3: hq = 2.0; break;
    case 4: hq = 2.5; break; case 5: hq = 2.5; break; case 6: hq = 3.5; break; case 7: hq = 5.0; break;
    case 8: hq = 7.0; break; case 9: hq = 10.0; break; default: hq = 0.0; break;
  }
  double eff = 0.0;
  if (ht >= 3 && n <= 3) eff = 2.0;
  else if (ht >= 5 && n == 5) eff = 1.0;
  else if (n <= 4 && e.hands_left <= 2) eff = 1.5;
  double syn = 0.0;
  if (e.njokers > 0) {
    uint32_t jf = bg_joker_flags(e, w.jt);
    if (ht == 5 && (jf & 1u)) syn += 2.0;
    if ((ht == 1 || ht == 2 || ht == 3) && (jf & 2u)) syn += 1.5;
    if (faces > 0 && (jf & 4u)) syn += 0.5 * (double)faces;
  }
  double strat = 0.0;
  if (new_progress > 0.7 && e.hands_left >= 3) strat = 2.0;
  else if (new_progress < 0.3 && ht >= 5) strat = 3.0;
  double ante_bonus = 0.0;
  if (e.ante >= 4) { ante_bonus = (double)(e.ante - 3) * 0.5; if (ante_bonus > 5.0) ante_bonus = 5.0; }
  double r = progress_reward + milestone;
  r = r + score_reward;
  r = r + hq * 2.0;
  r = r + eff * 1.5;
  r = r + syn * 3.0;
  r = r + strat * 2.0;
  r = r + ante_bonus;
  if (r > 100.0) r = 100.0;
  o.terms[0] = progress_reward; o.terms[1] = milestone; o.terms[2] = score_reward; o.terms[3] = hq;
  o.terms[4] = eff; o.terms[5] = syn; o.terms[6] = strat; o.terms[7] = ante_bonus;
  o.final_score = final_score; o.hand_type = ht; o.cards_played = n;
  BG_PROBE(11);
  // :914-960 outcome
  if (e.round_chips >= (int64_t)e.chips_needed) {
    double bonus = 25.0 + 10.0 * (double)e.ante;
    r += bonus < 50.0 ? bonus : 50.0;
    if (w.defer_adv) o.flags |= BG_FLAG_DEFER_ADV; // the service-wave kernel runs _advance_round as a second work item
    else bg_advance_round<DK::kCards>(d, env, e, w, sr);
    o.flags |= 1; // beat_blind
  } else if (e.hands_left <= 1) {
    r += -50.0 * (1.0 - new_progress);
    o.terminated = true;
    o.flags |= 2; // failed
  } else {
    e.hands_left -= 1;
    bg_draw_cards(e);
    if (e.boss_type) bg_boss_on_hand_drawn(d, env, e, w, dk);
  }
  BG_PROBE(12);
  o.reward = r;
  asm volatile("" ::"v"(touch0)); // the touched word is dead: this only keeps the load alive
}

// DISCARD  balatro_env_2.py:962-1050
template <class DK>
__device__ __forceinline__ void bg_step_discard(const BgDev& d, int env, Env& e, const RngWin& w, const DK& dk, StepOut& o) {
  int n = 0, nfaces = 0, purple = 0;
#pragma unroll 1
  for (int i = 0; i < e.nsel; i++) {
    int pos = bg_get8(e.sel, i);
    if (pos < e.nhand) {
      if constexpr (DK::kCards) purple += ((bg_cstate(d, env, bg_get8(e.hand, pos)) >> 8) & 0xfu) == 4u; // :975-979
      int rk = (bg_card(d, env, dk, bg_get8(e.hand, pos)) >> 2) + 2;
      nfaces += (rk >= 11 && rk <= 13);
      n++;
      e.highlighted |= 1u << pos; // :1011-1013 on top of stale play highlights
    }
  }
  bool first = e.discards_left == 3; // == game.discards (balatro_game.py:25)
  int money = 0, ndj = 0;
#pragma unroll 1
  for (int j = 0; j < e.njokers; j++) { // complete_joker_effects.py:186-209
    uint32_t f = w.jt->jf[bg_get8(e.jokers, j)];
    if ((f & 8u) && first && n == 1) money += 3;        // Trading Card
    else if ((f & 16u) && nfaces >= 3) money += 5;      // Faceless Joker
    ndj += (f & 32u) ? 1 : 0;
  }
  e.money += money;
  // balatro_game.py:111-127 discard_hand: drop every highlighted position, clear highlights, refill
  uint64_t nh = 0; int k = 0;
#pragma unroll 1
  for (int p = 0; p < e.nhand; p++)
    if (!(e.highlighted & (1u << p))) { nh |= (uint64_t)bg_get8(e.hand, p) << (8 * k); k++; }
  e.hand = nh; e.nhand = k;
  e.highlighted = 0;
  e.discards_left -= 1;
  bg_draw_cards(e);
  e.nsel = 0; e.sel = 0;
  if constexpr (DK::kCards) { // :1021-1032 purple seals -> rng.choice('seal_applications', tarots), stream 13
#pragma unroll 1
    for (int q = 0; q < purple; q++)
      if (e.ncons < 2) {
        uint32_t* S = d.sealmt + (size_t)env * BG_MTS;
        uint32_t t;
        int guard = 0;
        do { t = bg_lazy_next(S) >> 27; } while (t >= 22u && ++guard < 4096); // _randbelow(22): getrandbits(5)
        if (e.ncons == 0) e.cons0 = 1u + t; else e.cons1 = 1u + t;
        e.ncons++;
      }
  }
  double r = 0.2;
  if (ndj) r += 0.5 * (double)ndj;
  if (money > 0) r += (double)money / 5.0;
  int64_t need1 = e.chips_needed > 1 ? e.chips_needed : 1;
  double progress = (double)e.round_chips / (double)need1;
  if (progress < 0.5 && e.discards_left > 1) r += 0.5;
  else if (progress > 0.8 && e.discards_left > 1) r -= 0.3;
  o.reward = r;
}

// SHOP  balatro_env_2.py:1174-1253 + shop.py:160-205
__device__ __forceinline__ void bg_step_shop(const BgDev& d, int env, Env& e, RngWin& w, ShopRegs& sr, int action, StepOut& o) {
  if (action >= 32 && action < 37) { // sell joker :1202-1215
    int ji = action - 32;
    int id = bg_get8(e.jokers, ji);
    e.jokers = bg_del8(e.jokers, ji);
    e.njokers--;
    int v = w.jt->cost[id] / 2;
    if (v < 3) v = 3;
    e.money += v;
    e.jokers_sold++;
    o.reward = (double)v / 5.0;
    o.flags |= 128; o.aux = id;
    return;
  }
  if (action == 31) { // SKIP -> PLAY :1247-1251
    e.phase = 0;
    bg_draw_cards(e);
    o.reward = 0.0;
    return;
  }
  if (action == 30) { // REROLL shop.py:170-177
    int32_t cost = (int32_t)((double)e.shop_reroll_base * bg_shop_cost_mult(e, w.jt));
    if (e.money < cost) { o.reward = -1.0; o.error = 6; return; }
    e.money -= cost;
    e.shop_reroll_base = (int32_t)((double)e.shop_reroll_base * 1.35);
    w.need_inv = true;
    o.reward = 0.0;
    return;
  }
  // buy 20..28: the mask guarantees index < shop_n and money >= cost (shop.py:179-203)
  int idx = action - 20;
  bg_shop_load(d, env, sr);
  int32_t costs[9]; uint32_t tps[9];
  bg_shop_unpack(sr, costs, tps);
  int32_t cost = 0; uint32_t tp = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) if (i == idx) { cost = costs[i]; tp = tps[i]; }
  e.money -= cost;
#pragma unroll
  for (int i = 0; i < 8; i++) if (i >= idx) { costs[i] = costs[i + 1]; tps[i] = tps[i + 1]; } // inventory.pop(idx)
  bg_shop_pack(sr, costs, tps);
  bg_shop_store(d, env, sr);
  e.shop_n--;
  int type = tp & 0xff, payload = (tp >> 8) & 0xff;
  if (type == IT_PACK) {
    int count = payload == PK_STANDARD ? 3 : 1; // shop.py:150-157 _open_pack draws from the shop stream
    for (int i = 0; i < count; i++) { int c = (int)bg_randbelow<true>(d, env, e, w, 52u); if (i == 0) o.aux = c; }
    o.reward = 5.0; o.flags |= 8;
  } else if (type == IT_CARD) {
    o.reward = 3.0; o.flags |= 16;
  } else if (type == IT_JOKER) {
    if (e.njokers >= 5) { o.reward = -1.0; o.error = 7; return; } // chips already deducted (shop.py:187-197)
    e.jokers = bg_set8(e.jokers, e.njokers, payload);
    e.njokers++;
    o.reward = 15.0; o.flags |= 64; o.aux = payload;
  } else {
    if (payload == 0) e.n_magic++; else e.n_minim++;
    o.reward = 10.0; o.flags |= 32; o.aux = payload;
  }
}

// BLIND_SELECT  balatro_env_2.py:1255-1318
template <bool CARDS = false>
__device__ __forceinline__ void bg_step_blind(const BgDev& d, int env, Env& e, RngWin& w, ShopRegs& sr, int action, StepOut& o) {
  if (action < 48) {
    int b = action - 45;
    e.round = b + 1;
    int64_t need;
    {
      int a = e.ante;
      // balatro_env_2.py:55-74
      const int T[8][3] = {{300, 450, 600},    {450, 675, 900},    {600, 900, 1200},   {900, 1350, 1800},
                           {1350, 2025, 2700}, {2100, 3150, 4200}, {3300, 4950, 6600}, {5250, 7875, 10500}};
      if (a <= 8) need = T[a - 1][b];
      else { int k = a - 8; if (k > 92) k = 92; need = (int64_t)((double)T[7][b] * BG_POW15[k]); }
    }
    o.reward = 0.0;
    if (b == 2) {
      int boss = 1 + (int)bg_randbelow<false>(d, env, e, w, 28u); // boss_blinds.py:522-532 random.choice(list(BossBlindType))
      e.boss_type = boss; e.boss_types = 0; e.boss_cards = 0; e.boss_hp = 0; e.boss_req = 5;
      e.bflags |= BG_BF_FIRST_HAND;
      need = (int64_t)((double)need * (boss == 2 ? 2.0 : 1.0)); // The Wall
      if (boss == 9) e.discards_left = 0;   // The Water
      if (boss == 11) e.hand_size -= 1;     // The Manacle
      if (boss == 17) e.hands_left = 1;     // The Needle
      o.aux = boss;
      o.reward = 10.0;
    }
    e.chips_needed = (int32_t)need;
    e.phase = 0;
    bg_draw_cards(e);
  } else { // 48 SKIP_BLIND :1305-1316
    o.reward = -5.0;
    bg_advance_round<CARDS>(d, env, e, w, sr);
    o.flags |= 4;
  }
}

__device__ __forceinline__ void bg_step_init(StepOut& o) {
  o.reward = 0.0; o.final_score = 0; o.error = 0; o.flags = 0; o.aux = 0; o.hand_type = -1; o.cards_played = 0;
  o.terminated = false;
  o.bd_dst = nullptr;
#pragma unroll
  for (int i = 0; i < 8; i++) o.terms[i] = 0.0;
}

// :1052-1058 SELECT_CARD toggle; the selection ORDER is kept (it is the scoring / RNG order of a later play)
__device__ __forceinline__ void bg_toggle_select(Env& e, int pos) {
  // position of `pos` among the first nsel bytes of e.sel, without a loop: zero-byte test on sel ^ (pos in every byte); the
  // LOWEST flagged byte is exact (borrows only disturb bytes above a real zero) and a position is listed at most once
  const uint32_t rep = (uint32_t)pos * 0x01010101u;
  const uint64_t x = e.sel ^ (((uint64_t)rep << 32) | rep);
  uint64_t z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
  z &= e.nsel >= 8 ? ~0ull : ((1ull << (8 * e.nsel)) - 1ull);
  if (z) { e.sel = bg_del8(e.sel, (__ffsll((long long)z) - 1) >> 3); e.nsel--; }
  else { e.sel = bg_set8(e.sel, e.nsel, pos); e.nsel++; }
}

// _use_consumable (balatro_env_2.py:1066-1172) over ConsumableManager.use_consumable (consumables.py:622-652),
// TarotEffects.apply_tarot (:111-327) and SpectralEffects.apply_spectral (:354-613).  A consumable is held as its
// _get_consumable_ids id (:1545-1567: tarots 1-22, planets 30-41, spectrals 50-67); bit 7 marks a name in enum form
// ('THE_FOOL', what The Emperor creates): used like the natural name, shown as id 0.  What the reference really does:
//  - target cards are classes made by CardAdapter.to_consumable_format (:328-343): rank / suit edits never reach the deck,
//    only enhancement / edition / seal are copied back (:1122-1138), as the INTEGER values of consumables.py's enums (its
//    Seal enum is RED 1, BLUE 2, GOLD 3 while cards.py has GOLD 1, RED 2, BLUE 3: Talisman's seal acts as a blue one);
//  - to_dict()['consumables'] is the live list: created items are appended by the effect and again by :1157-1160;
//    to_dict()['jokers'] is a fresh list: only :1147-1155 adds jokers (by JOKER_LIBRARY name);
//  - The Hanged Man / Familiar / Grim / Incantation with a target and Sigil / Ouija raise: error 11, reward -1.0 (harness
//    convention), state as the exception leaves it;
//  - Cryptid's copies are appended behind the 52 real cards and draws take the lowest free index, so they are only COUNTED
//    (deck_size, Blue Joker); Immolate removes five sampled cards from the live list (error 12 below 24 real cards).
// Kernels built without card states keep planets only (bg_inject_consumables refuses anything else for them).
template <class DK>
__device__ __forceinline__ void bg_use_consumable(const BgDev& d, int env, Env& e, RngWin& w, const DK& dk, int ci, StepOut& o) {
  const int code = ci == 0 ? (int)e.cons0 : (int)e.cons1, id = code & 0x7f;
  uint32_t L = (e.cons0 & 0xffu) | ((e.cons1 & 0xffu) << 8); // the live list, one byte per entry (3 at most, transiently)
  int n = e.ncons;
  bool success = false;
  int money_gained = 0, planet = -1, naff = 0, set_enh = -1, set_edi = -1, set_seal = -1, nitems = 0, njc = 0, jc = 0, hs = 0, ncreated = 0, ndestroyed = 0;
  uint64_t aff = 0;
  uint32_t items = 0;
  if (id >= 30 && id <= 41) { planet = id - 30; success = true; } // :644-652
  else if constexpr (DK::kCards) {
    uint64_t tg = 0; int nt = 0; // :1074-1083 target cards (deck indexes) in selection order
#pragma unroll 1
    for (int i = 0; i < e.nsel; i++) {
      int pos = bg_get8(e.sel, i);
      if (pos < e.nhand) { tg |= (uint64_t)bg_get8(e.hand, pos) << (8 * nt); nt++; }
    }
    bool raises = false, unsupported = false;
    const int slots = 2;
    switch (id) {
      case 1: // The Fool :127-134 (the list holds at least the Fool itself)
        if (n > 0) {
          uint32_t c = (L >> (8 * bg_randbelow<false>(d, env, e, w, (uint32_t)n))) & 0xffu;
          L |= c << (8 * n); n++; // live list, no slot check
          items = c; nitems = 1; success = true;
        }
        break;
      case 2: case 4: case 6: // Magician LUCKY :136-143, Empress MULT :157-164, Hierophant BONUS :177-184
        if (nt > 0) { naff = nt < 2 ? nt : 2; aff = tg; set_enh = id == 2 ? 8 : (id == 4 ? 2 : 1); success = true; }
        break;
      case 3: // The High Priestess :145-155: choice first, slot check second
#pragma unroll 1
        for (int k = 0; k < 2; k++) {
          uint32_t p = 30u + bg_randbelow<false>(d, env, e, w, 9u);
          if (n < slots) { L |= p << (8 * n); n++; items |= p << (8 * nitems); nitems++; }
        }
        success = true;
        break;
      case 5: // The Emperor :166-175: slot check first; names in enum form
#pragma unroll 1
        for (int k = 0; k < 2; k++)
          if (n < slots) {
            uint32_t t = (1u + bg_randbelow<false>(d, env, e, w, 22u)) | 0x80u;
            L |= t << (8 * n); n++; items |= t << (8 * nitems); nitems++;
          }
        success = true;
        break;
      case 7: case 8: case 12: case 16: case 17: // Lovers WILD, Chariot STEEL, Justice GLASS, Devil GOLD, Tower STONE
        if (nt >= 1) { naff = 1; aff = tg; set_enh = id == 7 ? 3 : (id == 8 ? 5 : (id == 12 ? 4 : (id == 16 ? 7 : 6))); success = true; }
        break;
      case 9: // Strength :202-210: only cards below the ace are listed; the rank edit is lost
        if (nt > 0) {
#pragma unroll 1
          for (int i = 0; i < nt && i < 2; i++) {
            int ti = bg_get8(tg, i);
            if ((bg_card(d, env, dk, ti) >> 2) + 2 < 14) { aff |= (uint64_t)ti << (8 * naff); naff++; }
          }
          success = true;
        }
        break;
      case 10: // The Hermit :212-219
        money_gained = e.money < 20 ? e.money : 20; success = true;
        break;
      case 11: // Wheel of Fortune :221-231: `target_cards and random.random() < 0.25`
        if (nt > 0 && bg_grandom(d, env, e, w) < 0.25) {
          set_edi = 1 + (int)bg_randbelow<false>(d, env, e, w, 3u); naff = 1; aff = tg; success = true;
        }
        break;
      case 13: // The Hanged Man :241-251
        raises = nt > 0;
        break;
      case 14: // Death :253-261
        if (nt >= 2) { naff = 2; aff = tg; success = true; }
        break;
      case 15: // Temperance :263-273
        money_gained = 5 * e.njokers < 50 ? 5 * e.njokers : 50; success = true;
        break;
      case 18: case 19: case 20: case 22: // Star / Moon / Sun / World: suit edits are lost
        if (nt > 0) { naff = nt < 3 ? nt : 3; aff = tg; success = true; }
        break;
      case 21: { // Judgement :318-327
        uint32_t p = 30u + bg_randbelow<false>(d, env, e, w, 9u);
        if (n < slots) { L |= p << (8 * n); n++; items = p; nitems = 1; }
        success = true;
        break;
      }
      case 50: case 51: case 52: // Familiar / Grim / Incantation :373-457: deck.remove(target class)
        raises = nt >= 1;
        break;
      case 53: case 61: case 63: case 64: // Talisman 3, Deja Vu 1, Trance 2, Medium 4 (consumables.Seal values)
        if (nt >= 1) { naff = 1; aff = tg; set_seal = id == 53 ? 3 : (id == 61 ? 1 : (id == 63 ? 2 : 4)); success = true; }
        break;
      case 54: // Aura :467-474
        if (nt >= 1) { set_edi = 1 + (int)bg_randbelow<false>(d, env, e, w, 3u); naff = 1; aff = tg; success = true; }
        break;
      case 55: // Wraith :476-488 (rare_jokers by JOKER_LIBRARY name; 'Drivers License' is not a library name)
        if (e.njokers < 5) {
          uint32_t k = bg_randbelow<false>(d, env, e, w, 14u);
          jc = k == 4u ? 0 : 137 + (int)k; njc = 1; hs = -1; success = true;
        }
        break;
      case 56: case 57: // Sigil :490-498, Ouija :500-509: random.choice, then assignment to a frozen dataclass
        if (e.nhand > 0) { (void)bg_randbelow<false>(d, env, e, w, id == 56 ? 4u : 13u); raises = true; }
        break;
      case 58: // Ectoplasm :511-517
        if (e.njokers > 0) { hs = -1; success = true; }
        break;
      case 59: { // Immolate :519-531: random.sample(deck, 5) by index (Lib/random.py sample(): a selection set for n > 21, a pool of the n indexes
                 // below), then deck.remove(card) for each: every later deck index -- hand indexes, card_states keys -- now names another card
        const int nn = 52 - e.ndrop, n = nn + e.nfo;
        if (nn < 13) { unsupported = true; break; } // the hand's indexes (always 0..7) must stay valid: at least 8 real cards behind this use
        uint64_t gone = 0; int nfgone = 0;
        if (n > 21) {
          uint32_t p0 = 0xffu, p1 = 0xffu, p2 = 0xffu, p3 = 0xffu;
#pragma unroll 1
          for (int i = 0; i < 5; i++) {
            uint32_t j; int guard = 0;
            do { j = bg_randbelow<false>(d, env, e, w, (uint32_t)n); } while ((j == p0 || j == p1 || j == p2 || j == p3) && ++guard < 4096);
            p3 = p2; p2 = p1; p1 = p0; p0 = j;
            if ((int)j < nn) gone |= 1ull << j; else nfgone++;
          }
        } else { // pool = list(range(n)); result[i] = pool[j]; pool[j] = pool[n - i - 1] -- up to 21 five-bit entries in two registers (the RNG window may hold drawn words)
          uint64_t plo = 0, phi = 0;
#pragma unroll 1
          for (int i = 0; i < n; i++) { if (i < 12) plo |= (uint64_t)i << (5 * i); else phi |= (uint64_t)i << (5 * (i - 12)); }
#pragma unroll 1
          for (int i = 0; i < 5; i++) {
            const int j = (int)bg_randbelow<false>(d, env, e, w, (uint32_t)(n - i)), last = n - i - 1;
            const uint32_t pick = (uint32_t)((j < 12 ? plo >> (5 * j) : phi >> (5 * (j - 12))) & 31ull);
            const uint64_t lv = (last < 12 ? plo >> (5 * last) : phi >> (5 * (last - 12))) & 31ull;
            if (j < 12) plo = (plo & ~(31ull << (5 * j))) | (lv << (5 * j)); else phi = (phi & ~(31ull << (5 * (j - 12)))) | (lv << (5 * (j - 12)));
            if ((int)pick < nn) gone |= 1ull << pick; else nfgone++;
          }
        }
        // compact the deck through this lane's RNG window (nothing is cached in it here), then HBM copy + kernel-local copy
        DK& mdk = const_cast<DK&>(dk);
        uint32_t cur = 0; int wpos = 0; uint64_t played = 0;
#pragma unroll 1
        for (int i = 0; i < nn; i++)
          if (!((gone >> i) & 1ull)) {
            cur |= (uint32_t)bg_card(d, env, dk, i) << (8 * (wpos & 3));
            if ((e.boss_cards >> i) & 1ull) played |= 1ull << wpos; // The Pillar marks card OBJECTS (id(card)): they move along
            wpos++;
            if ((wpos & 3) == 0) { w.lds[((wpos >> 2) - 1) * BG_BLOCK] = cur; cur = 0; }
          }
#pragma unroll 1
        while (wpos < 64) { wpos++; if ((wpos & 3) == 0) { w.lds[((wpos >> 2) - 1) * BG_BLOCK] = cur; cur = 0; } } // the partial word, then zeros
        w.g_len = 0; w.s_len = 0; w.g_blk = -1;
#pragma unroll
        for (int k = 0; k < BG_NDECK; k++) {
          const uint4 c4 = make_uint4(w.lds[(4 * k) * BG_BLOCK], w.lds[(4 * k + 1) * BG_BLOCK], w.lds[(4 * k + 2) * BG_BLOCK], w.lds[(4 * k + 3) * BG_BLOCK]);
          d.deck[(size_t)k * d.N + env] = c4;
          bg_deck_set(mdk, k, c4);
        }
        e.boss_cards = played;
        e.ndrop += 5 - nfgone; e.nfo -= nfgone;
        ndestroyed = 5; money_gained = 20; success = true;
        break;
      }
      case 60: // Ankh :533-543: the "name" is a {'name','id'} dict unless the scorer-level harness hands out names
        if (e.njokers > 0) {
          uint32_t k = bg_randbelow<false>(d, env, e, w, (uint32_t)e.njokers);
          jc = (d.flags & 1u) ? bg_get8(e.jokers, (int)k) : 0; njc = 1; success = true;
        }
        break;
      case 62: // Hex :553-563
        if (e.njokers > 0) { (void)bg_randbelow<false>(d, env, e, w, (uint32_t)e.njokers); success = true; }
        break;
      case 65: // Cryptid :581-591: two consumables.Card copies appended to the live deck (counted, never drawn)
        if (nt >= 1) { if (52 - e.ndrop + e.nfo + 2 > 127) { unsupported = true; break; } e.nfo += 2; ncreated = 2; success = true; } // deck_size = np.int8(len(deck)) (:1491) stops at 127
        break;
      case 66: // The Soul :593-601
        if (e.njokers < 5) { jc = 146 + (int)bg_randbelow<false>(d, env, e, w, 5u); njc = 1; success = true; }
        break;
      case 67: // Black Hole :603-610
        success = true;
        break;
      default: break; // unknown name :654
    }
    if (raises) { o.reward = -1.0; o.error = 11; return; }
    if (unsupported) { o.reward = -1.0; o.error = 12; return; }
  }
  if (success) { // :1093-1164
    double r = 0.0;
    L = (L & ((1u << (8 * ci)) - 1u)) | ((L >> (8 * (ci + 1))) << (8 * ci)); n--; // pop(consumable_idx)
    if (money_gained > 0) { e.money += money_gained; r += (double)money_gained / 10.0; }
    if (planet >= 0) {
      const int ht = (int)((0xba9087654321ull >> (4 * planet)) & 0xfull); // Mercury..Eris -> hand type :1103-1116
      int lv = bg_level(e, ht);
      if (lv < 15) e.levels += 1ull << (4 * ht);          // engine.apply_planet (scoring_engine.py:82-85)
      else e.excess += 1ull << (4 * ht);                  // state.hand_levels[...] += 1 is uncapped (:1119)
      r += 10.0;
    }
    if constexpr (DK::kCards) {
      if (naff) {
#pragma unroll 1
        for (int i = 0; i < naff; i++) {
          const int ti = bg_get8(aff, i);
          uint16_t* q = &((uint16_t*)&d.cstate[(size_t)(ti >> 3) * d.N + env])[ti & 7];
          uint32_t v = *q;
          if (set_enh >= 0) v = (v & ~0xfu) | (uint32_t)set_enh;
          if (set_edi >= 0) v = (v & ~0xf0u) | ((uint32_t)set_edi << 4);
          if (set_seal >= 0) v = (v & ~0xf00u) | ((uint32_t)set_seal << 8);
          *q = (uint16_t)v;
        }
        r += (double)naff * 2.0;
      }
      if (ncreated) r += (double)ncreated * 3.0;     // :1140-1141
      if (ndestroyed) r += (double)ndestroyed * 1.0; // :1143-1144
      if (njc) {
        if (e.njokers < 5 && jc > 0) { e.jokers = bg_set8(e.jokers, e.njokers, jc); e.njokers++; }
        r += (double)njc * 15.0;
      }
      if (nitems) {
#pragma unroll 1
        for (int i = 0; i < nitems; i++)
          if (n < 2) { L |= ((items >> (8 * i)) & 0xffu) << (8 * n); n++; }
        r += (double)nitems * 5.0;
      }
      if (hs) e.hand_size += hs;
    }
    e.cons0 = L & 0xffu; e.cons1 = (L >> 8) & 0xffu; e.ncons = n;
    o.reward = r;
  } else { o.reward = -1.0; o.error = 8; } // :1166-1168
  e.nsel = 0; e.sel = 0; // :1171
}

// dispatch of a VALID action (balatro_env_2.py:629-637)
template <class DK>
__device__ __forceinline__ void bg_env_dispatch(const BgDev& d, int env, Env& e, RngWin& w, ShopRegs& sr, const DK& dk,
                                                int action, StepOut& o) {
  if (e.phase == 0) {
    if (action == 0) bg_step_play_hand(d, env, e, w, sr, dk, o);
    else if (action == 1) bg_step_discard(d, env, e, w, dk, o);
    else if (action < 10) bg_toggle_select(e, action - 2);
    else bg_use_consumable(d, env, e, w, dk, action - 10, o); // 10..14
  } else if (e.phase == 1) bg_step_shop(d, env, e, w, sr, action, o);
  else if (e.phase == 2) bg_step_blind<DK::kCards>(d, env, e, w, sr, action, o);
  if (w.need_inv) { BG_PROBE_BEGIN(); bg_shop_inventory(d, env, e, w, sr); w.need_inv = false; BG_PROBE(22); }
}

// the guards in front of the dispatch (balatro_env_2.py:619-627); returns true when the action must be dispatched
__device__ __forceinline__ bool bg_step_guards(const Env& e, uint64_t mask, int action, StepOut& o) {
  if (e.ante > 100) { o.terminated = true; o.error = 9; return false; }
  if (e.chips_scored > 1000000000ll) { o.terminated = true; o.error = 10; return false; }
  if (action < 0 || action >= 60 || !((mask >> action) & 1ull)) { o.reward = -1.0; o.error = 1; return false; }
  return true;
}

// balatro_env_2.py:616-637 step()
template <class DK>
__device__ __forceinline__ void bg_env_step(const BgDev& d, int env, Env& e, RngWin& w, ShopRegs& sr, const DK& dk, uint64_t mask,
                                            int action, StepOut& o) {
  bg_step_init(o);
  if (bg_step_guards(e, mask, action, o)) bg_env_dispatch(d, env, e, w, sr, dk, action, o);
  if (e.max_ante > 0 && e.ante > e.max_ante) { o.terminated = true; o.flags |= 256; }
}

// ---------------------------------------------------------------------------------------------------------
// Observation writer (balatro_env_2.py:1473-1541), reference dtypes, one array per key, row `env`.
// ---------------------------------------------------------------------------------------------------------
struct ObsPtrs {
  int8_t* hand; int8_t* hand_size; int8_t* deck_size; int64_t* selected_cards; int64_t* chips_scored;
  int32_t* round_chips_scored; float* progress_ratio; int32_t* mult; int32_t* chips_needed; int32_t* money;
  int16_t* ante; int8_t* round; int8_t* hands_left; int8_t* discards_left; int8_t* joker_count; int16_t* joker_ids;
  int8_t* joker_slots; int8_t* consumable_count; int16_t* consumables; int8_t* consumable_slots; int16_t* shop_items;
  int16_t* shop_costs; int16_t* shop_rerolls; int8_t* hand_levels; int8_t* phase; int8_t* action_mask;
  int32_t* hands_played; int32_t* best_hand_this_ante; int8_t* boss_blind_active; int8_t* boss_blind_type;
  int64_t* face_down_cards;
  // the 31 pointers above mirror bg_obs_ptrs; the packed-row output of bg_rollout_rows follows
  uint8_t* rows;        // non-null: one BG_ROW_BYTES record per (step, env) instead of the per-key arrays
  uint32_t row_stride;  // bytes between consecutive records (multiple of 16)
};
struct RowExtra { double reward; int32_t action; uint32_t terminated; bool cached = false; float prf = 0.0f; uint64_t handb = 0; uint32_t selm = 0; };
// The two observation values that only change in the heavy actions (so the service-wave kernel lets the service lane compute
// them and the env lane carry them): the hand as card codes (8 LDS byte reads) and progress_ratio (a float64 division).
template <class DK>
__device__ __forceinline__ uint64_t bg_obs_handb(const BgDev& d, int env, const Env& e, const DK& dk) {
  // (round 5: eight selects and ONE branch for the rare deck index beyond the 16 cards held in registers -- it was a branch per hand position, in
  //  every image build of every service step)
  const DeckHead dh = bg_deck_head(d, env, dk);
  uint32_t lo = 0, hi = 0;
  bool far = false;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t idx = (uint32_t)bg_get8(e.hand, i);
    const bool on = i < e.nhand;
    far = far || (on && idx >= 16u);
    const uint32_t c = (uint32_t)(((idx & 8u) ? dh.hi : dh.lo) >> (8u * (idx & 7u))) & 0xffu;
    const uint32_t v = on ? c : 0xffu;
    if (i < 4) lo |= v << (8 * i); else hi |= v << (8 * (i - 4));
  }
  uint64_t handb = ((uint64_t)hi << 32) | lo;
  if (far) { // Immolate / Cryptid workloads only
#pragma unroll 1
    for (int i = 0; i < e.nhand && i < 8; i++) {
      const int idx = bg_get8(e.hand, i);
      if (idx >= 16) handb = bg_set8(handb, i, bg_card(d, env, dk, idx));
    }
  }
  return handb;
}
__device__ __forceinline__ uint32_t bg_obs_selm(const Env& e) { // bit p = hand position p is selected
  uint32_t selm = 0;
#pragma unroll 1
  for (int i = 0; i < e.nsel; i++) selm |= 1u << bg_get8(e.sel, i);
  return selm;
}
__device__ __forceinline__ float bg_obs_prf(const Env& e) { // :1497 min(2, round_chips / max(1, chips_needed)) as float32
  int64_t need1 = e.chips_needed > 1 ? e.chips_needed : 1;
  double pr = (double)e.round_chips / (double)need1;
  return (float)(pr < 2.0 ? pr : 2.0);
}
// LDS staging of packed records (block-compacted rollout kernel): 64 slots x 6 pieces of 16 bytes + one address per slot
typedef uint32_t bg_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bg_u32x4 lds_u4;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
#define BG_STAGE_NP 11 // 16-byte pieces of a record staged at a time: runs of 176 bytes per row and store instruction
struct RowStage { lds_u4* stage; lds_u64* addr; };
// lanes of one wave exchanging data through LDS: the hardware queue is in order per wave, the compiler must not move
// LDS accesses across the hand-over point
#define BG_WAVE_SYNC() __builtin_amdgcn_wave_barrier()

// Statistics of a launch (bg_rollout_stats).  A wave folds its lanes with shuffles and adds the result to six LDS words of its workgroup; behind the
// workgroup's last barrier six lanes add those to the caller's struct.  One global atomic per wave and field -- 5 376 (engine 3) / 10 752 (bg_engine.h)
// read-modify-writes of ONE 48-byte line per launch of 65 536 envs, which the memory side serialises at ~11 ns apiece -- kept every launch ~58 us longer
// than its last workgroup (a fifth of a 20-step launch; tools/short_launches.py).
typedef __attribute__((address_space(3))) unsigned long long lds_stat;
__device__ __forceinline__ void bg_stats_wave(unsigned long long* st, uint64_t n_steps, uint64_t n_eps, uint64_t n_plays, int64_t ssum, uint64_t rbits, uint64_t ohash) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n_steps += __shfl_down(n_steps, off); n_eps += __shfl_down(n_eps, off); n_plays += __shfl_down(n_plays, off);
    ssum += __shfl_down(ssum, off); rbits ^= __shfl_down(rbits, off); ohash ^= __shfl_down(ohash, off);
  }
  if ((threadIdx.x & 63u) == 0u) {
    lds_stat* s = (lds_stat*)st;
    if (n_steps) __hip_atomic_fetch_add(&s[0], (unsigned long long)n_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (n_eps) __hip_atomic_fetch_add(&s[1], (unsigned long long)n_eps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (n_plays) __hip_atomic_fetch_add(&s[2], (unsigned long long)n_plays, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (ssum) __hip_atomic_fetch_add(&s[3], (unsigned long long)ssum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (rbits) __hip_atomic_fetch_xor(&s[4], (unsigned long long)rbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (ohash) __hip_atomic_fetch_xor(&s[5], (unsigned long long)ohash, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
// behind a __syncthreads() that follows every wave's bg_stats_wave
__device__ __forceinline__ void bg_stats_flush(bg_rollout_stats* out, const unsigned long long* st, int tid) {
  if (tid < 6) {
    const unsigned long long v = ((const lds_stat*)st)[tid];
    unsigned long long* dst = (unsigned long long*)out + tid;   // steps, episodes, plays, score_sum, reward_bits, obs_hash
    if (v) { if (tid < 4) atomicAdd(dst, v); else atomicXor(dst, v); }
  }
}
static_assert(sizeof(bg_rollout_stats) == 48, "six 64-bit fields, in the order bg_stats_flush assumes");

// `row` = env + t * N for [T, N, ...] rollout buffers.  Returns a 64-bit hash of the row (rollout checksum; the same
// value for both output layouts).
//
// Packed records (p.rows): the lanes of a workgroup sit on different steps, so with one array per key every 32-byte
// sector of the narrow keys is completed by several partial writes issued iterations apart -- measured 2.2x the
// algorithmic write traffic.  A record is 22 whole 16-byte stores owned by ONE lane: nothing is shared between lanes.
// STAGE: 0 = every lane stores its own record directly, 1 = two slices of BG_STAGE_NP pieces through LDS, 2 = three slices of
// 8 / 7 / 7 pieces (8 KB of staging per wave), 3 = the record goes to the env's IMAGE in LDS (rs.stage = its 22 pieces; the
// step engine copies images out cooperatively and patches them in place on cheap steps) -- and, when p.rows is null, the
// per-key arrays are written as well
template <bool HASH, int STAGE, class DK>
__device__ __forceinline__ uint64_t bg_write_obs_impl(const BgDev& d, int env, size_t row, const Env& e, const DK& dk,
                                                const ObsPtrs& p, uint64_t mask, ShopRegs& sr, const RowExtra& rx,
                                                const RowStage& rs) {
  uint64_t hsh = 0x9E3779B97F4A7C15ull;
#define BG_MIX(v) do { if (HASH) { hsh ^= (uint64_t)(v); hsh *= 0xBF58476D1CE4E5B9ull; hsh ^= hsh >> 29; } } while (0)
  // ---- values
  const uint64_t handb = rx.cached ? rx.handb : bg_obs_handb(d, env, e, dk);
  BG_MIX(handb);
  const uint32_t selm = rx.cached ? rx.selm : bg_obs_selm(e);
  BG_MIX(selm | ((uint64_t)e.face_down << 8) | ((uint64_t)e.nhand << 16));
  BG_MIX(e.chips_scored); BG_MIX(e.round_chips);
  int64_t need1 = e.chips_needed > 1 ? e.chips_needed : 1;
  const float prf = rx.cached ? rx.prf : bg_obs_prf(e);
  BG_MIX(__float_as_uint(prf));
  BG_MIX(((uint64_t)(uint32_t)e.chips_needed << 32) | (uint32_t)e.money);
  BG_MIX((uint64_t)e.ante | ((uint64_t)e.round << 8) | ((uint64_t)e.hands_left << 16) | ((uint64_t)e.discards_left << 24) |
         ((uint64_t)e.njokers << 32) | ((uint64_t)e.ncons << 40) | ((uint64_t)e.phase << 48));
  BG_MIX(e.jokers);
  uint32_t jq[5]; // joker_ids int16[10]: byte i of e.jokers -> halfword i (ids beyond njokers read as 0)
  {
    const uint64_t jm = e.jokers & (e.njokers >= 8 ? ~0ull : ((1ull << (8 * e.njokers)) - 1ull));
    const uint32_t jl = (uint32_t)jm, jh = (uint32_t)(jm >> 32);
    jq[0] = (jl & 0xffu) | ((jl & 0xff00u) << 8);
    jq[1] = ((jl >> 16) & 0xffu) | ((jl >> 24) << 16);
    jq[2] = (jh & 0xffu) | ((jh & 0xff00u) << 8);
    jq[3] = ((jh >> 16) & 0xffu) | ((jh >> 24) << 16);
    jq[4] = 0u;
  }
  BG_MIX(e.cons0 | (e.cons1 << 8) | ((uint32_t)e.ndrop << 16) | ((uint32_t)e.nfo << 24));
  const uint32_t cq = (uint32_t)((e.ncons > 0 && !(e.cons0 & 0x80u)) ? e.cons0 : 0) | ((uint32_t)((e.ncons > 1 && !(e.cons1 & 0x80u)) ? e.cons1 : 0) << 16); // enum-form names map to 0 (:1570)
  // shop rows only in SHOP phase (:1534-1539)
  uint32_t it[5] = {0, 0, 0, 0, 0}, co[5] = {0, 0, 0, 0, 0};
  if (e.phase == 1 && (e.bflags & BG_BF_SHOP_EXISTS)) {
    bg_shop_load(d, env, sr);
    int32_t costs[9]; uint32_t tps[9];
    bg_shop_unpack(sr, costs, tps);
#pragma unroll
    for (int i = 0; i < 9; i++)
      if (i < e.shop_n) {
        it[i >> 1] |= (tps[i] & 0xffu) << (16 * (i & 1));
        co[i >> 1] |= ((uint32_t)costs[i] & 0xffffu) << (16 * (i & 1));
      }
  }
#pragma unroll
  for (int i = 0; i < 5; i++) { BG_MIX(((uint64_t)it[i] << 32) | co[i]); }
  BG_MIX(e.shop_reroll_state);
  uint32_t lv[3]; // hand_levels int8[12]: nibble ht of (levels, excess) -> byte ht of their sum (<= 30: no carry between bytes)
#pragma unroll
  for (int w = 0; w < 3; w++) {
    uint32_t a = (uint32_t)(e.levels >> (16 * w)) & 0xffffu, b = (uint32_t)(e.excess >> (16 * w)) & 0xffffu;
    a = (a | (a << 8)) & 0x00ff00ffu; a = (a | (a << 4)) & 0x0f0f0f0fu;
    b = (b | (b << 8)) & 0x00ff00ffu; b = (b | (b << 4)) & 0x0f0f0f0fu;
    lv[w] = a + b;
    BG_MIX(lv[w]);
  }
  BG_MIX(mask);
  uint32_t mq[15];
#pragma unroll
  for (int w = 0; w < 15; w++) {
    // four mask bits -> four 0/1 bytes: x * (1 + 2^7 + 2^14 + 2^21) puts bit i at position 8 i (no two terms overlap)
    const uint32_t bits = (uint32_t)(mask >> (4 * w)) & 0xfu;
    mq[w] = __umul24(bits, 0x204081u) & 0x01010101u;
  }
  BG_MIX(((uint64_t)(uint32_t)e.hp_total << 32) | (uint32_t)e.best_hand);
  BG_MIX(e.boss_type);
#undef BG_MIX
  // ---- packed record (offsets: BG_ROW_* in include/balatro_mi355x.h)
  if (p.rows || STAGE == 3) {
    uint32_t w[88];
#pragma unroll
    for (int i = 0; i < 8; i++) { w[2 * i] = (selm >> i) & 1u; w[2 * i + 1] = 0u; }            //   0 selected_cards i64[8]
#pragma unroll
    for (int i = 0; i < 8; i++) { w[16 + 2 * i] = (e.face_down >> i) & 1u; w[17 + 2 * i] = 0u; } //  64 face_down_cards i64[8]
    w[32] = (uint32_t)(uint64_t)e.chips_scored; w[33] = (uint32_t)((uint64_t)e.chips_scored >> 32); // 128 chips_scored i64
    { const uint64_t rb = (uint64_t)__double_as_longlong(rx.reward); w[34] = (uint32_t)rb; w[35] = (uint32_t)(rb >> 32); } // 136 reward f64
    w[36] = (uint32_t)(int32_t)e.round_chips; w[37] = __float_as_uint(prf); w[38] = 1u; w[39] = (uint32_t)e.chips_needed; // 144..
    w[40] = (uint32_t)e.money; w[41] = (uint32_t)e.hp_total; w[42] = (uint32_t)(int32_t)e.best_hand; w[43] = (uint32_t)rx.action; // 160..
#pragma unroll
    for (int i = 0; i < 15; i++) w[44 + i] = mq[i];                                              // 176 action_mask i8[60]
#pragma unroll
    for (int i = 0; i < 5; i++) { w[59 + i] = jq[i]; w[64 + i] = it[i]; w[69 + i] = co[i]; }     // 236 / 256 / 276 i16[10] each
    w[74] = cq; w[75] = 0u;                                                                      // 296 consumables i16[5]
    w[76] = ((uint32_t)e.ante & 0xffffu) << 16;                                                  // 306 ante i16
    w[77] = ((uint32_t)e.shop_reroll_state & 0xffffu) | ((uint32_t)(handb & 0xffffull) << 16);   // 308 shop_rerolls i16, 310 hand i8[8]
    w[78] = (uint32_t)(handb >> 16);
    w[79] = (uint32_t)(handb >> 48) | ((lv[0] & 0xffffu) << 16);                                 // 318 hand_levels i8[12]
    w[80] = (lv[0] >> 16) | (lv[1] << 16);
    w[81] = (lv[1] >> 16) | (lv[2] << 16);
    w[82] = (lv[2] >> 16) | (((uint32_t)e.nhand & 0xffu) << 16) | ((uint32_t)(52 - e.ndrop + e.nfo) << 24);                   // 330 hand_size, deck_size
    w[83] = ((uint32_t)e.round & 0xffu) | (((uint32_t)e.hands_left & 0xffu) << 8) | (((uint32_t)e.discards_left & 0xffu) << 16) |
            (((uint32_t)e.njokers & 0xffu) << 24);                                               // 332 round, hands_left, discards_left, joker_count
    w[84] = 5u | (((uint32_t)e.ncons & 0xffu) << 8) | (2u << 16) | (((uint32_t)e.phase & 0xffu) << 24); // 336 joker_slots, consumable_count, consumable_slots, phase
    w[85] = (e.boss_type ? 1u : 0u) | (((uint32_t)e.boss_type & 0xffu) << 8) | ((rx.terminated & 1u) << 16); // 340 boss_blind_active, boss_blind_type, terminated
    w[86] = 0u; w[87] = 0u;
    uint8_t* rowp = p.rows + row * (size_t)p.row_stride;
    if constexpr (STAGE == 3) {
#pragma unroll
      for (int k = 0; k < 22; k++) rs.stage[k] = bg_u32x4{w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]};
    } else if constexpr (STAGE == 2) {
      // as below, in three slices of 8 / 7 / 7 pieces (runs of 128 / 112 bytes per row and store instruction)
      const unsigned long long act = __ballot(1);
      const uint32_t A = (uint32_t)__popcll(act);
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
      rs.addr[rank] = (unsigned long long)rowp;
#pragma unroll
      for (int sl = 0; sl < 3; sl++) {
        const int np = sl == 0 ? 8 : 7, base = sl == 0 ? 0 : (sl == 1 ? 8 : 15);
#pragma unroll
        for (int c = 0; c < np; c++) { const int k = base + c; rs.stage[rank * np + c] = bg_u32x4{w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]}; }
        BG_WAVE_SYNC();
#pragma unroll
        for (int k = 0; k < np; k++) {
          const uint32_t q = (uint32_t)k * A + rank;                                  // < np * A
          const uint32_t r = np == 8 ? q >> 3 : (q * 9363u) >> 16;                    // q / np (exact for q < 64 * np)
          const uint32_t c = q - r * (uint32_t)np;
          const unsigned long long a = rs.addr[r];
          *(__attribute__((address_space(1))) bg_u32x4*)(a + 16ull * (uint32_t)base + 16ull * c) = rs.stage[q];
        }
        BG_WAVE_SYNC();
      }
    } else if constexpr (STAGE == 1) {
      // The lanes that finished a step this iteration write their records out TOGETHER: 16 bytes per lane straight to
      // 64 different rows keeps the store path busy ~4x longer than the same bytes in row-contiguous runs (measured:
      // 22 such stores were a quarter of the kernel).  BG_STAGE_NP 16-byte pieces of every record go to LDS, then lane `rank`
      // stores pieces rank, rank + A, ... of the A x BG_STAGE_NP staged ones: consecutive lanes = consecutive pieces of a row.
      const unsigned long long act = __ballot(1);
      const uint32_t A = (uint32_t)__popcll(act);
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
      rs.addr[rank] = (unsigned long long)rowp;
#pragma unroll
      for (int sl = 0; sl < 22 / BG_STAGE_NP; sl++) {
        constexpr int np = BG_STAGE_NP; // pieces per slice (divides 22)
#pragma unroll
        for (int c = 0; c < np; c++) { const int k = np * sl + c; rs.stage[rank * np + c] = bg_u32x4{w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]}; }
        // other lanes of this wave read what this lane just wrote: make the writes of the whole wave land first (the
        // LDS queue is in order per wave; the fence keeps the compiler from moving the reads across)
        BG_WAVE_SYNC();
#pragma unroll
        for (int k = 0; k < np; k++) {
          const uint32_t q = (uint32_t)k * A + rank;                        // < np * A
          const uint32_t r = (q * (uint32_t)((65536 + np - 1) / np)) >> 16; // q / np (exact for q < 64 * np, np in {2, 11, 22})
          const uint32_t c = q - r * (uint32_t)np;
          const unsigned long long a = rs.addr[r];
          *(__attribute__((address_space(1))) bg_u32x4*)(a + 16ull * (uint32_t)(np * sl) + 16ull * c) = rs.stage[q]; // global_store, not flat
        }
        BG_WAVE_SYNC(); // ... and the reads before the next slice overwrites the staging
      }
    } else {
      uint4* q = (uint4*)rowp;
#pragma unroll
      for (int k = 0; k < 22; k++) q[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
    }
    if (STAGE != 3 || p.rows) return hsh;
  }
  // ---- one array per key
  if (p.hand) ((uint64_t*)p.hand)[row] = handb;
  if (p.hand_size) p.hand_size[row] = (int8_t)e.nhand;
  if (p.deck_size) p.deck_size[row] = (int8_t)(52 - e.ndrop + e.nfo);
  if (p.selected_cards) {
    ulonglong2* q = (ulonglong2*)(p.selected_cards + row * 8);
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = make_ulonglong2((selm >> (2 * i)) & 1u, (selm >> (2 * i + 1)) & 1u);
  }
  if (p.chips_scored) p.chips_scored[row] = e.chips_scored;
  if (p.round_chips_scored) p.round_chips_scored[row] = (int32_t)e.round_chips;
  if (p.progress_ratio) p.progress_ratio[row] = prf;
  if (p.mult) p.mult[row] = 1;
  if (p.chips_needed) p.chips_needed[row] = e.chips_needed;
  if (p.money) p.money[row] = e.money;
  if (p.ante) p.ante[row] = (int16_t)e.ante;
  if (p.round) p.round[row] = (int8_t)e.round;
  if (p.hands_left) p.hands_left[row] = (int8_t)e.hands_left;
  if (p.discards_left) p.discards_left[row] = (int8_t)e.discards_left;
  if (p.joker_count) p.joker_count[row] = (int8_t)e.njokers;
  if (p.joker_ids) {
    uint32_t* q = (uint32_t*)(p.joker_ids + row * 10);
#pragma unroll
    for (int i = 0; i < 5; i++) q[i] = jq[i];
  }
  if (p.joker_slots) p.joker_slots[row] = 5;
  if (p.consumable_count) p.consumable_count[row] = (int8_t)e.ncons;
  if (p.consumables) {
    int16_t* q = p.consumables + row * 5;
    q[0] = (int16_t)(cq & 0xffffu); q[1] = (int16_t)(cq >> 16); q[2] = 0; q[3] = 0; q[4] = 0;
  }
  if (p.consumable_slots) p.consumable_slots[row] = 2;
  if (p.shop_items) { uint32_t* q = (uint32_t*)(p.shop_items + row * 10);
#pragma unroll
    for (int i = 0; i < 5; i++) q[i] = it[i]; }
  if (p.shop_costs) { uint32_t* q = (uint32_t*)(p.shop_costs + row * 10);
#pragma unroll
    for (int i = 0; i < 5; i++) q[i] = co[i]; }
  if (p.shop_rerolls) p.shop_rerolls[row] = (int16_t)e.shop_reroll_state;
  if (p.hand_levels) { uint32_t* q = (uint32_t*)(p.hand_levels + row * 12); q[0] = lv[0]; q[1] = lv[1]; q[2] = lv[2]; }
  if (p.phase) p.phase[row] = (int8_t)e.phase;
  if (p.action_mask) {
    uint32_t* q = (uint32_t*)(p.action_mask + row * 60);
#pragma unroll
    for (int w = 0; w < 15; w++) q[w] = mq[w];
  }
  if (p.hands_played) p.hands_played[row] = e.hp_total;
  if (p.best_hand_this_ante) p.best_hand_this_ante[row] = (int32_t)e.best_hand;
  if (p.boss_blind_active) p.boss_blind_active[row] = e.boss_type ? 1 : 0;
  if (p.boss_blind_type) p.boss_blind_type[row] = (int8_t)e.boss_type;
  if (p.face_down_cards) {
    ulonglong2* q = (ulonglong2*)(p.face_down_cards + row * 8);
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = make_ulonglong2((e.face_down >> (2 * i)) & 1u, (e.face_down >> (2 * i + 1)) & 1u);
  }
  return hsh;
}

template <bool HASH, class DK>
__device__ __forceinline__ uint64_t bg_write_obs(const BgDev& d, int env, size_t row, const Env& e, const DK& dk, const ObsPtrs& p,
                                                uint64_t mask, ShopRegs& sr, const RowExtra& rx) {
  return bg_write_obs_impl<HASH, 0>(d, env, row, e, dk, p, mask, sr, rx, RowStage{nullptr, nullptr});
}

// ---- record images (LDS): one 352-byte record per env, 22 pieces of 16 bytes, kept current by the step engine
// per-key arrays of one row from an image (cheap steps of the engine in per-key mode); byte offsets: BG_ROW_*
__device__ __forceinline__ void bg_emit_keys_from_image(const lds_u4* img, const ObsPtrs& p, size_t row) {
  const lds_u32* w = (const lds_u32*)img;
  const lds_u8* b = (const lds_u8*)img;
  if (p.selected_cards) { ulonglong2* q = (ulonglong2*)(p.selected_cards + row * 8);
#pragma unroll
    for (int i = 0; i < 4; i++) { const bg_u32x4 v = img[i]; q[i] = make_ulonglong2(((unsigned long long)v.y << 32) | v.x, ((unsigned long long)v.w << 32) | v.z); } }
  if (p.face_down_cards) { ulonglong2* q = (ulonglong2*)(p.face_down_cards + row * 8);
#pragma unroll
    for (int i = 0; i < 4; i++) { const bg_u32x4 v = img[4 + i]; q[i] = make_ulonglong2(((unsigned long long)v.y << 32) | v.x, ((unsigned long long)v.w << 32) | v.z); } }
  if (p.chips_scored) p.chips_scored[row] = (int64_t)(((uint64_t)w[33] << 32) | w[32]);
  if (p.round_chips_scored) p.round_chips_scored[row] = (int32_t)w[36];
  if (p.progress_ratio) p.progress_ratio[row] = __uint_as_float(w[37]);
  if (p.mult) p.mult[row] = (int32_t)w[38];
  if (p.chips_needed) p.chips_needed[row] = (int32_t)w[39];
  if (p.money) p.money[row] = (int32_t)w[40];
  if (p.hands_played) p.hands_played[row] = (int32_t)w[41];
  if (p.best_hand_this_ante) p.best_hand_this_ante[row] = (int32_t)w[42];
  if (p.action_mask) { uint32_t* q = (uint32_t*)(p.action_mask + row * 60);
#pragma unroll
    for (int i = 0; i < 15; i++) q[i] = w[44 + i]; }
  if (p.joker_ids) { uint32_t* q = (uint32_t*)(p.joker_ids + row * 10);
#pragma unroll
    for (int i = 0; i < 5; i++) q[i] = w[59 + i]; }
  if (p.shop_items) { uint32_t* q = (uint32_t*)(p.shop_items + row * 10);
#pragma unroll
    for (int i = 0; i < 5; i++) q[i] = w[64 + i]; }
  if (p.shop_costs) { uint32_t* q = (uint32_t*)(p.shop_costs + row * 10);
#pragma unroll
    for (int i = 0; i < 5; i++) q[i] = w[69 + i]; }
  if (p.consumables) { int16_t* q = p.consumables + row * 5; const uint32_t a = w[74], c = w[75]; q[0] = (int16_t)(a & 0xffffu); q[1] = (int16_t)(a >> 16); q[2] = (int16_t)(c & 0xffffu); q[3] = (int16_t)(c >> 16); q[4] = (int16_t)(w[76] & 0xffffu); }
  if (p.ante) p.ante[row] = (int16_t)(w[76] >> 16);
  if (p.shop_rerolls) p.shop_rerolls[row] = (int16_t)(w[77] & 0xffffu);
  if (p.hand) { int8_t* q = p.hand + row * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = (int8_t)b[310 + i]; }
  if (p.hand_levels) { int8_t* q = p.hand_levels + row * 12;
#pragma unroll
    for (int i = 0; i < 12; i++) q[i] = (int8_t)b[318 + i]; }
  if (p.hand_size) p.hand_size[row] = (int8_t)b[330];
  if (p.deck_size) p.deck_size[row] = (int8_t)b[331];
  if (p.round) p.round[row] = (int8_t)b[332];
  if (p.hands_left) p.hands_left[row] = (int8_t)b[333];
  if (p.discards_left) p.discards_left[row] = (int8_t)b[334];
  if (p.joker_count) p.joker_count[row] = (int8_t)b[335];
  if (p.joker_slots) p.joker_slots[row] = (int8_t)b[336];
  if (p.consumable_count) p.consumable_count[row] = (int8_t)b[337];
  if (p.consumable_slots) p.consumable_slots[row] = (int8_t)b[338];
  if (p.phase) p.phase[row] = (int8_t)b[339];
  if (p.boss_blind_active) p.boss_blind_active[row] = (int8_t)b[340];
  if (p.boss_blind_type) p.boss_blind_type[row] = (int8_t)b[341];
}
// checksum of one observation: every byte of the image except the step's reward / action / terminated (tests: BG_POLICY_HASH_OBS)
__device__ __forceinline__ uint64_t bg_hash_image(const lds_u4* img) {
  uint64_t h = 0x9E3779B97F4A7C15ull;
#pragma unroll
  for (int k = 0; k < 22; k++) {
    bg_u32x4 v = img[k];
    if (k == 8) { v.z = 0; v.w = 0; }               // reward (bytes 136..143)
    if (k == 10) v.w = 0;                            // action (172..175)
    if (k == 21) v.y &= 0xff00ffffu;                 // terminated (342)
    h ^= ((uint64_t)v.y << 32) | v.x; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 29;
    h ^= ((uint64_t)v.w << 32) | v.z; h *= 0x94D049BB133111EBull; h ^= h >> 31;
  }
  return h;
}

// The same policy with everything that depends only on the env hoisted out of the step loop (the 64-bit `% 3` and one of
// the three 64-bit multiplies), and the k-th valid action found by a branch-free popcount descent instead of a loop whose
// trip count differs per lane.
struct PolicyLane { uint64_t seed_env; int blind; };
__device__ __forceinline__ PolicyLane bg_policy_lane(int policy, uint64_t policy_seed, uint64_t env_index) {
  PolicyLane p;
  p.seed_env = policy_seed + 0x9E3779B97F4A7C15ull * (env_index + 1);
  p.blind = policy == 2 ? 45 + (int)(env_index % 3) : 45;
  return p;
}
__device__ __forceinline__ int bg_policy_action(const Env& e, uint64_t mask, int policy, const PolicyLane& pl, uint64_t t) {
  if (policy != 0) {
    if (e.phase == 2) return pl.blind;
    if (e.phase == 1) return 31;
  }
  const int nv = __popcll(mask);
  if (!nv) return 0;
  uint64_t x = pl.seed_env + 0xD1B54A32D192ED03ull * (t + 1);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  uint32_t k = (uint32_t)(x >> 32) % (uint32_t)nv;
  const uint32_t lo = (uint32_t)mask, hi = (uint32_t)(mask >> 32);
  uint32_t c = (uint32_t)__popc(lo);
  uint32_t w = k < c ? lo : hi;
  int base = k < c ? 0 : 32;
  k = k < c ? k : k - c;
#pragma unroll
  for (int sh = 16; sh >= 1; sh >>= 1) { // k-th set bit of w
    c = (uint32_t)__popc(w & ((1u << sh) - 1u));
    const bool up = k >= c;
    w = up ? w >> sh : w;
    base += up ? sh : 0;
    k = up ? k - c : k;
  }
  return base;
}

// The same policy for a lane that keeps x0 = seed_env + psi * (t + 1) itself (+= psi per step: no 64-bit multiply for it) and
// takes h % nv from a reciprocal table: q = umulhi(h, floor(2**32 / nv)) is the quotient or one less, so one conditional
// subtraction makes the remainder exact.
#define BG_POLICY_PSI 0xD1B54A32D192ED03ull
__device__ __forceinline__ int bg_policy_action_fast(const Env& e, uint64_t mask, int policy, const PolicyLane& pl, uint64_t x0, lds_JTables* jt) {
  if (policy != 0) {
    if (e.phase == 2) return pl.blind;
    if (e.phase == 1) return 31;
  }
  const int nv = __popcll(mask);
  if (!nv) return 0;
  uint64_t x = x0;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  const uint32_t h = (uint32_t)(x >> 32);
  uint32_t k = h - __umulhi(h, jt->inv[nv]) * (uint32_t)nv;
  k = k >= (uint32_t)nv ? k - (uint32_t)nv : k;
  const uint32_t lo = (uint32_t)mask, hi = (uint32_t)(mask >> 32);
  uint32_t c = (uint32_t)__popc(lo);
  uint32_t w = k < c ? lo : hi;
  int base = k < c ? 0 : 32;
  k = k < c ? k : k - c;
#pragma unroll
  for (int sh = 16; sh >= 1; sh >>= 1) { // k-th set bit of w
    c = (uint32_t)__popc(w & ((1u << sh) - 1u));
    const bool up = k >= c;
    w = up ? w >> sh : w;
    base += up ? sh : 0;
    k = up ? k - c : k;
  }
  return base;
}

// counter-hash policy on a 60-bit action mask; phase overrides for the scripted policies
__device__ __forceinline__ int bg_policy_action(const Env& e, uint64_t mask, int policy, uint64_t policy_seed,
                                                uint64_t env_index, uint64_t t) {
  if (policy != 0) {
    if (e.phase == 2) return policy == 2 ? 45 + (int)(env_index % 3) : 45;
    if (e.phase == 1) return 31;
  }
  int nv = __popcll(mask);
  if (!nv) return 0;
  uint32_t k = bg_policy_hash(policy_seed, env_index, t) % (uint32_t)nv;
  uint64_t m = mask;
#pragma unroll 1
  for (uint32_t i = 0; i < k; i++) m &= m - 1; // clear the k lowest set bits
  return __ffsll((long long)m) - 1;
}
