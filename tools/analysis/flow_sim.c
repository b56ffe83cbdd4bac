/* Dev tool (CPU): schedule model of the tokenizer's "splice" flow (inflate_par.hpp, round 4).
 *
 *   flow_sim <file with concatenated gzip members> [SUB bits] [CPW bits] [STEPS] [RING items] [LANES]
 *
 * The block's bitstream is cut into ITEMS of SUB bits.  An idle lane is given the next item and starts decoding AT the
 * item boundary (in general not a token boundary: its first tokens are garbage, Huffman streams self-synchronise).  It
 * records where its trajectory stands at every CPW-bit checkpoint of its own item.  When it runs over the end of its
 * item it goes on into the next one, now comparing its position with the owner's checkpoints: equal = merged, the
 * runner stops, the owner's tokens from that checkpoint on continue the stream.  A runner that reaches an unassigned
 * item stops exactly there (the item is then given out with a true start).  The true stream is the chain
 * runner(0) -> owner of the item it merged into -> ...; items are retired in order along that chain.
 * The model decodes real bits (garbage trajectories included) and counts wave-steps (a step of the 64 lanes in which at
 * least one lane decodes a token), lane-steps and scheduling points per block.
 * Nothing here is linked into the product or the tests. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint16_t count[16], first[16], offs[16], sym[320]; int maxlen; } Code;
static const uint8_t *in; static size_t n;
static inline uint32_t bit(uint64_t p) { return p < n * 8 ? (in[p >> 3] >> (p & 7)) & 1 : 0; }
static uint32_t bits(uint64_t *p, int k) { uint32_t v = 0; for (int i = 0; i < k; ++i) v |= bit((*p)++) << i; return v; }
static int build(Code *c, const uint8_t *lens, int cnt) {
  memset(c, 0, sizeof *c);
  for (int i = 0; i < cnt; ++i) c->count[lens[i]]++;
  c->count[0] = 0;
  int code = 0, off = 0; c->maxlen = 0;
  for (int l = 1; l < 16; ++l) { c->first[l] = code; c->offs[l] = off; if (c->count[l]) c->maxlen = l; code = (code + c->count[l]) << 1; off += c->count[l]; }
  uint16_t next[16]; memcpy(next, c->offs, sizeof next);
  for (int i = 0; i < cnt; ++i) if (lens[i]) c->sym[next[lens[i]]++] = i;
  return 0;
}
static int decode(const Code *c, uint64_t *p) {
  int code = 0;
  for (int l = 1; l <= c->maxlen; ++l) {
    code = (code << 1) | bit((*p)++);
    int idx = code - c->first[l];
    if (idx >= 0 && idx < c->count[l]) return c->sym[c->offs[l] + idx];
  }
  return -1;
}
static const uint8_t LX[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint8_t DX[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
/* one token at *p; 0 literal, 2 match, 1 end of block, -1 bad */
static int token(const Code *ll, const Code *dc, uint64_t *p) {
  int s = decode(ll, p);
  if (s < 0 || s > 285) return -1;
  if (s < 256) return 0;
  if (s == 256) return 1;
  *p += LX[s - 257];
  int d = decode(dc, p);
  if (d < 0 || d > 29) return -1;
  *p += DX[d];
  return 2;
}

static int SUB = 1024, CPW = 128, STEPS = 12, RING = 64, NL = 64, MULTILIT = 0, RETIRE_MIN = 16;
#define MAXL 256
enum { IDLE = 0, RUN = 1 };
enum { ST_NONE = 0, ST_MERGE, ST_EXACT, ST_EOB, ST_ERR };
typedef struct { int mode; int origin; uint64_t pos; int cur_item; } Lane;
typedef struct { int assigned, owner; uint64_t *cp; int stop_kind; int stop_item, stop_j; uint64_t stop_pos; int exact_start; uint64_t reserved, start_pos; } Item;

static uint64_t g_wave_steps, g_lane_steps, g_sched, g_tokens, g_blocks, g_exact_stops, g_miss_empty, g_merges, g_items, g_true_tokens_steps;
static uint64_t g_idle_lane_steps, g_killed, g_restarts;

static void sim_block(const Code *ll, const Code *dc, uint64_t p0, uint64_t pend, uint64_t ntok) {
  const uint64_t org = ((p0 >> 3) & ~3ull) * 8;
  const int ncp = SUB / CPW;
  const int n_items = (int)((pend - org + SUB - 1) / SUB) + 1;
  Item *it = calloc(n_items + 2, sizeof(Item));
  for (int i = 0; i < n_items + 2; ++i) it[i].cp = calloc(ncp, sizeof(uint64_t));
  Lane ln[MAXL]; memset(ln, 0, sizeof ln);
  int next_item = 0, retired = 0, chain = 0 /* origin of the runner the truth is in */, done = 0;
  it[0].reserved = p0;  /* item 0 starts at the true start */
  const int n_spec = (int)((pend - org + SUB - 1) / SUB);  /* items the member reaches into (the index's hint) */
  uint64_t guard = 0;
  while (!done && ++guard < 1000000) {
    g_sched++;
    /* ---- follow the chain / retire ---- */
    for (;;) {
      Item *r = &it[chain];
      if (!r->assigned || r->stop_kind == ST_NONE) break;
      if (r->stop_kind == ST_EOB) { done = 1; break; }
      if (r->stop_kind == ST_ERR) { fprintf(stderr, "true stream hit an error?\n"); exit(1); }
      if (r->stop_kind == ST_MERGE) { chain = r->stop_item; g_merges++; }
      else { /* EXACT: the runner stopped in front of an unassigned item, which it reserved for its continuation */
        const int s = (int)((r->stop_pos - org) / SUB);
        if (it[s].assigned && it[s].start_pos == r->stop_pos) { chain = s; }
        else if (it[s].assigned || (it[s].reserved && it[s].reserved != r->stop_pos)) { g_restarts++; fprintf(stderr, "restart needed\n"); exit(1); }
        else break;  /* reserved by this runner, not given out yet */
      }
    }
    if (done) break;
    retired = chain;  /* slots in front of the chain head are free */
    for (int l = 0; l < NL; ++l)  /* runners behind the true stream are of no use any more */
      if (ln[l].mode == RUN && ln[l].origin < chain) { ln[l].mode = IDLE; g_killed++; }
    /* ---- assign ---- */
    for (int l = 0; l < NL; ++l) {
      if (ln[l].mode != IDLE) continue;
      if (next_item < n_spec && next_item < retired + RING && next_item < n_items) {
        const int s = next_item++;
        it[s].assigned = 1; it[s].owner = l;
        ln[l].mode = RUN; ln[l].origin = s;
        if (it[s].reserved) { ln[l].pos = it[s].reserved; it[s].exact_start = 1; }
        else { ln[l].pos = org + (uint64_t)s * SUB; it[s].cp[0] = ln[l].pos; }
        it[s].start_pos = ln[l].pos;
        g_items++;
      }
    }
    /* ---- decode steps ---- */
    for (int k = 0; k < STEPS; ++k) {
      int active = 0;
      for (int l = 0; l < NL; ++l) if (ln[l].mode == RUN) active++;
      if (!active) break;
      g_wave_steps++;
      g_lane_steps += active;
      g_idle_lane_steps += NL - active;
      for (int l = 0; l < NL; ++l) {
        Lane *L = &ln[l];
        if (L->mode != RUN) continue;
        Item *me = &it[L->origin];
        uint64_t p = L->pos;
        int r = token(ll, dc, &p);
        if (MULTILIT && r == 0) {  /* a second literal in the same step */
          uint64_t q = p; int r2 = token(ll, dc, &q);
          /* (a checkpoint between the two literals is handled by the first one: conservatively do not pair across one) */
          if (r2 == 0 && ((p - org) / CPW) == ((L->pos - org) / CPW)) p = q;
        }
        if (r == 1) { me->stop_kind = ST_EOB; me->stop_pos = p; L->mode = IDLE; continue; }
        if (r < 0 || p > pend + 4096) { me->stop_kind = ST_ERR; L->mode = IDLE; continue; }
        const uint64_t old = L->pos;
        L->pos = p;
        /* crossed a checkpoint boundary?  (tokens are shorter than CPW: at most one boundary in (old, p]) */
        const uint64_t c = (p - org) / CPW;
        if (org + c * CPW <= old) continue;
        const int s = (int)(c * CPW / SUB), j = (int)((c * CPW % SUB) / CPW);
        if (s >= n_items) { me->stop_kind = ST_ERR; L->mode = IDLE; continue; }
        if (s == L->origin || (it[s].assigned && it[s].owner == l)) { it[s].cp[j] = p; continue; }
        if (!it[s].assigned) {  /* nobody decodes this item yet: stop exactly here; the item is reserved for the continuation */
          me->stop_kind = ST_EXACT; me->stop_pos = p; L->mode = IDLE; g_exact_stops++;
          if (!it[s].reserved) it[s].reserved = p;
          continue;
        }
        if (it[s].cp[j] == p) { me->stop_kind = ST_MERGE; me->stop_item = s; me->stop_j = j; me->stop_pos = p; L->mode = IDLE; continue; }
        if (it[s].cp[j] == 0) g_miss_empty++;
      }
    }
  }
  if (!done) { fprintf(stderr, "block did not finish (guard)\n"); exit(1); }
  g_tokens += ntok; g_blocks++;
  for (int i = 0; i < n_items + 2; ++i) free(it[i].cp);
  free(it);
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
  if (argc > 2) SUB = atoi(argv[2]);
  if (argc > 3) CPW = atoi(argv[3]);
  if (argc > 4) STEPS = atoi(argv[4]);
  if (argc > 5) RING = atoi(argv[5]);
  if (argc > 6) NL = atoi(argv[6]);
  if (argc > 7) MULTILIT = atoi(argv[7]);
  fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t *buf = malloc(n + 16); if (fread(buf, 1, n, f) != n) return 2; memset(buf + n, 0, 16); in = buf;
  size_t pos = 0; int members = 0;
  while (pos + 18 <= n && in[pos] == 0x1f && in[pos + 1] == 0x8b) {
    int flg = in[pos + 3]; size_t q = pos + 10;
    if (flg & 4) q += 2 + in[q] + 256 * in[q + 1];
    if (flg & 8) { while (in[q]) ++q; ++q; }
    if (flg & 16) { while (in[q]) ++q; ++q; }
    if (flg & 2) q += 2;
    uint64_t p = (uint64_t)q * 8;
    for (;;) {
      int final = bits(&p, 1), type = bits(&p, 2);
      if (type == 0) { p = (p + 7) & ~7ull; uint32_t len = bits(&p, 16); bits(&p, 16); p += 8ull * len; }
      else if (type == 3) return 3;
      else {
        uint8_t lens[320]; memset(lens, 0, sizeof lens); int hlit = 288, hdist = 30;
        if (type == 1) { for (int i = 0; i < 288; ++i) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8; for (int i = 0; i < 30; ++i) lens[288 + i] = 5; }
        else {
          hlit = bits(&p, 5) + 257; hdist = bits(&p, 5) + 1; int ncl = bits(&p, 4) + 4;
          static const uint8_t ord[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
          uint8_t cl[19] = {0}; for (int i = 0; i < ncl; ++i) cl[ord[i]] = bits(&p, 3);
          Code cc; build(&cc, cl, 19);
          for (int i = 0; i < hlit + hdist;) {
            int s = decode(&cc, &p); if (s < 0) return 4;
            if (s < 16) lens[i++] = s;
            else { int rep = s == 16 ? 3 + bits(&p, 2) : s == 17 ? 3 + bits(&p, 3) : 11 + bits(&p, 7); int v = s == 16 ? lens[i - 1] : 0; while (rep--) lens[i++] = v; }
          }
        }
        Code ll, dc; build(&ll, lens, hlit); build(&dc, lens + hlit, hdist);
        uint64_t t = p, nt = 0; int r;
        while ((r = token(&ll, &dc, &t)) == 0 || r == 2) nt++;
        if (r < 0) return 5;
        sim_block(&ll, &dc, p, t, nt);
        p = t;
      }
      if (final) break;
    }
    pos = (size_t)((p + 7) >> 3) + 8;
    members++;
  }
  printf("SUB %d CPW %d STEPS %d RING %d LANES %d MULTILIT %d: members %d blocks %llu tokens %llu (%.0f per member)\n", SUB, CPW, STEPS, RING, NL, MULTILIT, members,
         (unsigned long long)g_blocks, (unsigned long long)g_tokens, (double)g_tokens / members);
  printf("  wave-steps per member %.1f  (ideal %.1f)  lane-steps per token %.3f  lane occupancy %.1f %%  scheduling points per member %.1f\n",
         (double)g_wave_steps / members, (double)g_tokens / members / NL, (double)g_lane_steps / g_tokens,
         100.0 * g_lane_steps / (g_lane_steps + g_idle_lane_steps), (double)g_sched / members);
  printf("  items per member %.1f  merges %.1f  exact stops %.1f  checkpoint empty when compared %.2f  runners killed %.2f per member\n", (double)g_items / members,
         (double)g_merges / members, (double)g_exact_stops / members, (double)g_miss_empty / members, (double)g_killed / members);
  return 0;
}
