// bg_sim.h -- the reference's secondary evaluator / scorer on the device: balatro_gym/balatro_sim.py::BalatroSimulator
// (SURVEY.md 8 a14): get_x_same :108-126, get_flush :128-149, get_straight :151-214, evaluate_hand :220-366 (scoring subsets,
// Four Fingers / Shortcut, flush suit priority Spades > Hearts > Clubs > Diamonds, the 5K -> 4K -> 3K -> Pair cascade) and
// calculate_score :402-548 (string enhancements / editions / seals, JOKER-major chain over complete_joker_effects.py with its
// eager RNG draws, `int(chips * ((mult + add) * x))`).  Not reachable from the live env; exposed as the operator-level entry
// points bg_sim_evaluate_batch / bg_sim_score_batch (lane = case).  The joker effects come from the SAME descriptor tables as
// the step path's chain (JTables: bg_jdesc / bg_jmain_desc); only the order of the draws differs (joker-major here).
// A card is rank | suit << 4 | enhancement << 8 | edition << 12 | seal << 16 plus its base_value; codes: include/balatro_mi355x.h.
#pragma once

struct SimHand { uint32_t cd[8]; int32_t bv[8]; int n; };
__device__ __forceinline__ int bg_sim_rank(uint32_t c) { return (int)(c & 0xfu); }
__device__ __forceinline__ int bg_sim_suit(uint32_t c) { return (int)((c >> 4) & 0x3u); }

struct SimEval {
  int top;
  int8_t nlists[12], n0[12], pos[12][8];
};
__device__ __forceinline__ void bg_sim_set(SimEval& r, int type, int nlists, const int8_t* first, int flen) {
  r.nlists[type] = (int8_t)nlists; r.n0[type] = (int8_t)flen;
  for (int i = 0; i < 8; i++) r.pos[type][i] = i < flen ? first[i] : (int8_t)-1;
}
// :108-126 groups of EXACTLY num equal ranks in descending rank order; inside a group the lowest index comes first, the others
// ascending (the reference loops i downwards, so the assignment that survives is the one made from the group's lowest index)
__device__ __forceinline__ int bg_sim_x_same(int num, const SimHand& h, int8_t groups[4][8], int8_t* glen) {
  int g = 0;
  for (int r = 14; r >= 2 && g < 4; r--) {
    int k = 0;
    int8_t cur[8];
    for (int i = 0; i < h.n; i++)
      if (bg_sim_rank(h.cd[i]) == r && k < 8) cur[k++] = (int8_t)i;
    if (k == num) { for (int i = 0; i < k; i++) groups[g][i] = cur[i]; glen[g] = (int8_t)k; g++; }
  }
  return g;
}
__device__ __forceinline__ int bg_sim_flush(const SimHand& h, bool ff, int8_t* out) { // :128-149
  const int req = ff ? 4 : 5;
  if (h.n > 5 || h.n < req) return 0;
  const int ORDER[4] = {3, 2, 0, 1}; // "Spades", "Hearts", "Clubs", "Diamonds"
  for (int q = 0; q < 4; q++) {
    int k = 0;
    for (int i = 0; i < h.n; i++)
      if (bg_sim_suit(h.cd[i]) == ORDER[q]) out[k++] = (int8_t)i;
    if (k >= req) return k;
  }
  return 0;
}
__device__ __forceinline__ int bg_sim_straight(const SimHand& h, bool ff, bool shortcut, int8_t* out) { // :151-214
  const int req = ff ? 4 : 5;
  if (h.n > 5 || h.n < req) return 0;
  int8_t t[16];
  int tl = 0, len = 0;
  bool straight = false, skipped = false;
  for (int r = 14; r > 1; r--) {
    bool any = false;
    for (int i = 0; i < h.n; i++) any = any || bg_sim_rank(h.cd[i]) == r;
    if (any) {
      len++;
      for (int i = 0; i < h.n; i++)
        if (bg_sim_rank(h.cd[i]) == r && tl < 16) t[tl++] = (int8_t)i;
    } else if (shortcut && !skipped) skipped = true;
    else { len = 0; tl = 0; skipped = false; }
    if (len >= req) { straight = true; break; }
  }
  if (!straight) { // the wheel A-2-3-4-5; `skipped` carries over from the loop above
    const int WHEEL[5] = {14, 2, 3, 4, 5};
    int8_t w[16];
    int wl = 0, wlen = 0;
    for (int q = 0; q < 5; q++) {
      bool any = false;
      for (int i = 0; i < h.n; i++) any = any || bg_sim_rank(h.cd[i]) == WHEEL[q];
      if (any) {
        wlen++;
        for (int i = 0; i < h.n; i++)
          if (bg_sim_rank(h.cd[i]) == WHEEL[q] && wl < 16) w[wl++] = (int8_t)i;
      } else if (shortcut && !skipped) skipped = true;
      else break;
    }
    if (wlen >= req) { for (int i = 0; i < wl; i++) t[i] = w[i]; tl = wl; straight = true; }
  }
  if (!straight) return 0;
  if (tl > req) tl = req; // t[:required_cards]
  for (int i = 0; i < tl; i++) out[i] = t[i];
  return tl;
}
// :220-366
__device__ __forceinline__ void bg_sim_evaluate(const SimHand& h, bool ff, bool shortcut, SimEval& r) {
  for (int t = 0; t < 12; t++) { r.nlists[t] = 0; r.n0[t] = 0; for (int i = 0; i < 8; i++) r.pos[t][i] = -1; }
  r.top = -1;
  int8_t g5[4][8], g4[4][8], g3[4][8], g2[4][8], l5[4], l4[4], l3[4], l2[4], fl[8], st[8], tmp[16];
  const int n5 = bg_sim_x_same(5, h, g5, l5), n4 = bg_sim_x_same(4, h, g4, l4), n3 = bg_sim_x_same(3, h, g3, l3), n2 = bg_sim_x_same(2, h, g2, l2);
  const int nf = bg_sim_flush(h, ff, fl), ns = bg_sim_straight(h, ff, shortcut, st);
#define BG_SIM_TOP(t) do { if (r.top < 0) r.top = (t); } while (0)
  if (n5 && nf) { bg_sim_set(r, 11, n5, g5[0], l5[0]); BG_SIM_TOP(11); }
  if (n3 && n2 && nf) { for (int i = 0; i < l3[0]; i++) tmp[i] = g3[0][i]; for (int i = 0; i < l2[0]; i++) tmp[l3[0] + i] = g2[0][i]; bg_sim_set(r, 10, 1, tmp, l3[0] + l2[0]); BG_SIM_TOP(10); }
  if (n5) { bg_sim_set(r, 9, n5, g5[0], l5[0]); BG_SIM_TOP(9); }
  if (nf && ns) { // Straight Flush: the flush cards, then the straight's cards that are not `in` them (dataclass equality: every field)
    int k = 0;
    for (int i = 0; i < nf; i++) tmp[k++] = fl[i];
    for (int i = 0; i < ns; i++) {
      bool in = false;
      for (int j = 0; j < nf; j++) in = in || (h.cd[st[i]] == h.cd[fl[j]] && h.bv[st[i]] == h.bv[fl[j]]);
      if (!in) tmp[k++] = st[i];
    }
    bg_sim_set(r, 8, 1, tmp, k > 8 ? 8 : k); BG_SIM_TOP(8);
  }
  if (n4) { bg_sim_set(r, 7, n4, g4[0], l4[0]); BG_SIM_TOP(7); }
  if (n3 && n2) { for (int i = 0; i < l3[0]; i++) tmp[i] = g3[0][i]; for (int i = 0; i < l2[0]; i++) tmp[l3[0] + i] = g2[0][i]; bg_sim_set(r, 6, 1, tmp, l3[0] + l2[0]); BG_SIM_TOP(6); }
  if (nf) { bg_sim_set(r, 5, 1, fl, nf); BG_SIM_TOP(5); }
  if (ns) { bg_sim_set(r, 4, 1, st, ns); BG_SIM_TOP(4); }
  if (n3) { bg_sim_set(r, 3, n3, g3[0], l3[0]); BG_SIM_TOP(3); }
  if (n2 == 2 || (n3 == 1 && n2 == 1)) {
    for (int i = 0; i < l2[0]; i++) tmp[i] = g2[0][i];
    if (n2 > 1) { for (int i = 0; i < l2[1]; i++) tmp[l2[0] + i] = g2[1][i]; bg_sim_set(r, 2, 1, tmp, l2[0] + l2[1]); }
    else { for (int i = 0; i < l3[0]; i++) tmp[l2[0] + i] = g3[0][i]; bg_sim_set(r, 2, 1, tmp, l2[0] + l3[0]); }
    BG_SIM_TOP(2);
  }
  if (n2) { bg_sim_set(r, 1, n2, g2[0], l2[0]); BG_SIM_TOP(1); }
  { for (int i = 0; i < h.n; i++) tmp[i] = (int8_t)i; bg_sim_set(r, 0, 1, tmp, h.n); BG_SIM_TOP(0); } // [hand], also for an empty hand
#undef BG_SIM_TOP
  // cascades :357-364, after `top` is settled
  if (r.nlists[9]) { int8_t c[8]; for (int i = 0; i < 8; i++) c[i] = r.pos[9][i]; bg_sim_set(r, 7, 1, c, r.n0[9] < 4 ? r.n0[9] : 4); }
  if (r.nlists[7]) { int8_t c[8]; for (int i = 0; i < 8; i++) c[i] = r.pos[7][i]; bg_sim_set(r, 3, 1, c, r.n0[7] < 3 ? r.n0[7] : 3); }
  if (r.nlists[3]) { int8_t c[8]; for (int i = 0; i < 8; i++) c[i] = r.pos[3][i]; bg_sim_set(r, 1, 1, c, r.n0[3] < 2 ? r.n0[3] : 2); }
}

// layouts of the batch records: include/balatro_mi355x.h
#define BG_SIM_CASE_WORDS 64
__device__ __forceinline__ void bg_sim_load_hand(const int32_t* c, int n, SimHand& h) {
  h.n = n < 0 ? 0 : (n > 8 ? 8 : n);
  for (int i = 0; i < 8; i++) {
    const int32_t* q = c + 6 * i;
    h.cd[i] = i < h.n ? ((uint32_t)(q[0] & 0xf) | ((uint32_t)(q[1] & 3) << 4) | ((uint32_t)(q[3] & 0xf) << 8) | ((uint32_t)(q[4] & 7) << 12) | ((uint32_t)(q[5] & 7) << 16)) : 0u;
    h.bv[i] = i < h.n ? q[2] : 0;
  }
}

__global__ __launch_bounds__(BG_BLOCK) void bg_sim_evaluate_batch_kernel(const int32_t* __restrict__ hands, const int32_t* __restrict__ n,
                                                                        const int32_t* __restrict__ flags, int8_t* __restrict__ out, int m) {
  const int i = blockIdx.x * BG_BLOCK + threadIdx.x;
  if (i >= m) return;
  SimHand h;
  bg_sim_load_hand(hands + (size_t)i * 48, n[i], h);
  SimEval r;
  bg_sim_evaluate(h, (flags[i] & 1) != 0, (flags[i] & 2) != 0, r);
  int8_t* o = out + (size_t)i * 128;
  o[0] = (int8_t)r.top;
  for (int t = 0; t < 12; t++) { o[1 + t] = r.nlists[t]; o[13 + t] = r.n0[t]; for (int k = 0; k < 8; k++) o[32 + 8 * t + k] = r.pos[t][k]; }
}

// calculate_score :402-548 after random.seed(seed)
__global__ __launch_bounds__(BG_BLOCK) void bg_sim_score_batch_kernel(BgDev d, const int32_t* __restrict__ cases, int64_t* __restrict__ out) {
  __shared__ uint32_t win[BG_WIN][BG_BLOCK];
  __shared__ JTables jt;
  bg_tables_init(&jt);
  const int i = blockIdx.x * BG_BLOCK + threadIdx.x;
  if (i >= d.N) return;
  const int32_t* c = cases + (size_t)i * BG_SIM_CASE_WORDS;
  uint32_t* g0 = bg_gblock(d, i, 0);
  uint32_t* g1 = bg_gblock(d, i, 1);
  bg_mt_seed(g0, (uint32_t)c[58]);
  bg_mt_twist(g0, g0);
  bg_mt_twist(g0, g1);
  for (int k = 0; k < 16; k++) g0[BG_MT_N + k] = g1[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  Env e;
  { uint4 z[BG_NHOT];
#pragma unroll
    for (int k = 0; k < BG_NHOT; k++) z[k] = make_uint4(0, 0, 0, 0);
    bg_unpack(z, e); }
  e.g_cur = 0; e.g_idx = 0; e.g_cons = 0; e.g_valid = 2;
  RngWin w;
  bg_win_init(w, &win[0][threadIdx.x], &jt);
  SimHand h;
  bg_sim_load_hand(c, c[48], h);
  int nj = c[49]; nj = nj < 0 ? 0 : (nj > 5 ? 5 : nj);
  const int hands_left = c[55], discards_left = c[56], deck_len = c[57];
  bool ff = false, shortcut = false; // :240-243 Four Fingers (jokers.py id 18) / Shortcut (69) anywhere among the player's jokers
  for (int j = 0; j < nj; j++) { ff = ff || c[50 + j] == 18; shortcut = shortcut || c[50 + j] == 69; }
  SimEval ev;
  bg_sim_evaluate(h, ff, shortcut, ev);
  const int top = ev.top, nsc = ev.nlists[top] ? ev.n0[top] : 0;
  int bchips, bmult;
  bg_hand_base(top, 1, bchips, bmult);                 // :424-441 the level-1 table (calculate_score knows no hand levels)
  int64_t chips = bchips, add_mult = 0, money = 0;
  double xm = 1.0;
  uint32_t sc[8];
  for (int q = 0; q < 8; q++) sc[q] = q < nsc ? h.cd[ev.pos[top][q]] : 0u;
  uint32_t suits = 0; int kings = 0, queens = 0; bool all_black = true;
  for (int q = 0; q < nsc; q++) { chips += h.bv[ev.pos[top][q]]; suits |= 1u << bg_sim_suit(sc[q]); }
  for (int q = 0; q < h.n; q++) { const int r = bg_sim_rank(h.cd[q]), s = bg_sim_suit(h.cd[q]); kings += r == 13; queens += r == 12; if (!(s == 3 || s == 0)) all_black = false; }
  for (int q = 0; q < nsc; q++) {                      // :452-486 enhancements, editions, seals of the scoring cards
    const uint32_t enh = (sc[q] >> 8) & 0xfu, edi = (sc[q] >> 12) & 7u, seal = (sc[q] >> 16) & 7u;
    if (enh == 1u) chips += 30; else if (enh == 2u) add_mult += 4;
    else if (enh == 4u) { xm *= 2.0; (void)bg_grandom(d, i, e, w); }
    else if (enh == 5u) xm *= 1.5; else if (enh == 6u) chips += 50; else if (enh == 7u) money += 3;
    else if (enh == 8u) { if (bg_grandom(d, i, e, w) < 0.2) money += 1; }
    if (edi == 1u) chips += 50; else if (edi == 2u) add_mult += 10; else if (edi == 3u) xm *= 1.5;
    if (seal == 1u) money += 3;
  }
  // conditions of the main-phase descriptors (bg_jmain_desc), with balatro_sim's hand-type names and its game_state
  const uint32_t cond = 1u | (nsc <= 3 ? 2u : 0u) | (hands_left == 1 ? 4u : 0u) | (discards_left == 0 ? 8u : 0u) | ((suits & 15u) << 4) |
                        (all_black ? 1u << 8 : 0u) | (((suits & 1u) && __popc(suits) > 1) ? 1u << 9 : 0u) | (__popc(suits) == 4 ? 1u << 10 : 0u) |
                        (kings > 0 ? 1u << 11 : 0u) | (queens > 0 ? 1u << 12 : 0u) | (top == 1 ? 1u << 13 : 0u) | (top == 3 ? 1u << 14 : 0u) |
                        (top == 7 ? 1u << 15 : 0u) | (1u << (16 + top));
  for (int j = 0; j < nj; j++) {                        // :491-534 joker-major: before (no effect), individual per card, main
    const int id = c[50 + j] & 0xff;
    const uint64_t dsc = jt.jd[id < 152 ? id : 0];
    const uint32_t dm = jt.jm[id < 152 ? id : 0];
    const uint32_t sp = (uint32_t)(dsc >> 20) & 3u, suit1 = (uint32_t)(dsc >> 16) & 7u;
    for (int q = 0; q < nsc; q++) {
      const double blood = bg_grandom(d, i, e, w);     // the eagerly built suit_effects dict (complete_joker_effects.py:157-162)
      const int r = bg_sim_rank(sc[q]), s = bg_sim_suit(sc[q]);
      bool match = suit1 ? (s == (int)suit1 - 1) : (((uint32_t)dsc >> r) & 1u) != 0u;
      if (sp == 1u) { if (r == 8) (void)bg_grandom(d, i, e, w); match = false; }   // 8 Ball :167
      if (sp == 2u) match = match && blood < 0.5;                                   // Bloodstone
      if (match) { chips += (int)((dsc >> 24) & 0xffu); add_mult += (int)((dsc >> 32) & 0xffu); if ((dsc >> 22) & 1u) xm *= 2.0; }
      if (id == 116 && s == 1) money += 1;                                          // Rough Gem
    }
    const uint32_t rb = bg_randbelow<false>(d, i, e, w, 24u);                       // Misprint's randint(0, 23), drawn for every joker
    if ((cond >> (dm & 31u)) & 1u) {
      const uint32_t vk = (dm >> 5) & 7u;
      const int cst = (int)(dm >> 8);
      if (vk == 0u) add_mult += cst; else if (vk == 1u) chips += (cst == 104 ? 2 * deck_len : cst);
      else if (vk == 2u) xm *= (double)cst; else if (vk == 3u) add_mult += (int)rb; else if (vk == 4u) add_mult += 3 * nj;
      else if (vk == 5u) chips += 30 * discards_left; else if (vk == 6u) xm *= jt.pow15[kings]; else if (vk == 7u) add_mult += 13 * queens;
    }
  }
  bg_gnorm(d, e);
  const double final_mult = (double)((int64_t)bmult + add_mult) * xm;                // :537
  int64_t* o = out + (size_t)i * 8;
  o[0] = (int64_t)((double)chips * final_mult);                                      // :538
  o[1] = chips; o[2] = add_mult; o[3] = __double_as_longlong(xm); o[4] = money;
  o[5] = (int64_t)e.g_cons * BG_MT_N + e.g_idx;
  o[6] = (int64_t)bg_gpeek(d, i, e, 0);
  o[7] = (int64_t)top | ((int64_t)nsc << 8);
}
