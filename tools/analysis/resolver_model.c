/* Dev tool (CPU): model of a TOKEN-centric resolver (64 tokens per step, no byte pass) on real token streams.
 *
 *   resolver_model <file with concatenated gzip members> [batch_bytes = 1728]
 *
 * The byte-centric resolver of inflate_par.hpp walks every output byte through a 64-lane group although ~3/4 of the
 * bytes arrive through the per-token deposit path (tools/analysis/token_stats.c).  This models the alternative and checks
 * it byte for byte against a plain sequential LZ77 replay:
 *   per batch (<= batch_bytes of output, cut at a token boundary), per chunk of 64 tokens in order:
 *     phase 1  every literal lane stores its byte; every match whose whole source is flushed output (in front of the
 *              batch) copies it (any order: those bytes are final);
 *     phase 2  the remaining matches of the chunk (source inside the batch window) in ROUNDS: the first pending match is
 *              always ready (everything in front of its destination is final; dist < len is a periodic fill), a later one
 *              is ready when its source ends at or before the first pending destination; ready matches copy at once.
 * Reported: rounds per chunk, how many matches take which path.  Nothing here is linked into the product or the tests. */
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
static int decode(const Code *c, uint64_t *p) {  /* -1: no code */
  int code = 0;
  for (int l = 1; l <= c->maxlen; ++l) {
    code = (code << 1) | bit((*p)++);
    int idx = code - c->first[l];
    if (idx >= 0 && idx < c->count[l]) return c->sym[c->offs[l] + idx];
  }
  return -1;
}
static const uint16_t LB[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t LX[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint8_t DX[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
static const uint16_t DB[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
/* one token at *p; returns 0 literal/match (len = 0 for a literal), 1 end of block, -1 bad */
static int token(const Code *ll, const Code *dc, uint64_t *p, int *len, int *dist) {
  int s = decode(ll, p);
  *len = 0; *dist = 0;
  if (s < 0 || s > 285) return -1;
  if (s < 256) return 0;
  if (s == 256) return 1;
  *len = LB[s - 257] + bits(p, LX[s - 257]);
  int d = decode(dc, p);
  if (d < 0 || d > 29) return -1;
  *dist = DB[d] + bits(p, DX[d]);
  return 0;
}


typedef struct { uint32_t len, dist; uint8_t lit; } Tok;
static Tok *tk; static size_t ntk, tk_cap;
static uint8_t *ref, *ob; static size_t ob_cap;
static uint32_t BATCH = 1728;
static uint64_t chunks, round_hist[70], n_lit, n_far, n_win, n_self, n_long, far_long, total_rounds, members, bytes_total, straddle;
static void push(uint32_t len, uint32_t dist, uint8_t lit) {
  if (ntk == tk_cap) { tk_cap = tk_cap ? tk_cap * 2 : 1 << 16; tk = realloc(tk, tk_cap * sizeof(Tok)); }
  tk[ntk].len = len; tk[ntk].dist = dist; tk[ntk].lit = lit; ntk++;
}
static int run_member(void) {
  size_t out = 0;
  for (size_t i = 0; i < ntk; ++i) out += tk[i].len ? tk[i].len : 1;
  if (out + 64 > ob_cap) { ob_cap = out * 2 + 64; ref = realloc(ref, ob_cap); ob = realloc(ob, ob_cap); }
  size_t o = 0;
  for (size_t i = 0; i < ntk; ++i) {  /* reference replay */
    if (!tk[i].len) ref[o++] = tk[i].lit;
    else { if (tk[i].dist > o) return 10; for (uint32_t k = 0; k < tk[i].len; ++k, ++o) ref[o] = ref[o - tk[i].dist]; }
  }
  memset(ob, 0xEE, out);
  size_t t0 = 0, batch0 = 0;
  static size_t off[64]; static int pend[64];
  while (t0 < ntk) {
    size_t t1 = t0, bytes = 0;
    while (t1 < ntk) { uint32_t l = tk[t1].len ? tk[t1].len : 1; if (bytes + l > BATCH && t1 > t0) break; bytes += l; ++t1; }
    size_t pos = batch0;
    for (size_t c = t0; c < t1; c += 64) {
      size_t ce = c + 64 < t1 ? c + 64 : t1;
      int np = 0;
      for (size_t i = c; i < ce; ++i) { off[i - c] = pos; pos += tk[i].len ? tk[i].len : 1; }
      for (size_t i = c; i < ce; ++i) {  /* phase 1 */
        const Tok *t = &tk[i]; size_t d = off[i - c];
        if (!t->len) { ob[d] = t->lit; n_lit++; continue; }
        size_t s = d - t->dist;
        if (s + t->len <= batch0) { memcpy(ob + d, ob + s, t->len); n_far++; if (t->len > 16) far_long++; }
        else { pend[np++] = (int)(i - c); if (s < batch0) straddle++; }
      }
      int rounds = 0;
      while (np) {  /* phase 2 */
        rounds++;
        size_t first_dst = off[pend[0]];
        static uint8_t tmp[64][258]; int ready[64], nr = 0, keep = 0;
        for (int j = 0; j < np; ++j) {
          const Tok *t = &tk[c + pend[j]]; size_t d = off[pend[j]], s = d - t->dist;
          if (j == 0 || s + t->len <= first_dst) {
            /* read now (all ready lanes read before any writes: the model of one parallel round) */
            for (uint32_t k = 0; k < t->len; ++k) tmp[nr][k] = (j == 0 && t->dist < t->len) ? ob[s + k % t->dist] : ob[s + k];
            ready[nr++] = pend[j];
          } else pend[keep++] = pend[j];
        }
        for (int r = 0; r < nr; ++r) { const Tok *t = &tk[c + ready[r]]; memcpy(ob + off[ready[r]], tmp[r], t->len); n_win++; if (t->dist < t->len) n_self++; if (t->len > 16) n_long++; }
        np = keep;
      }
      chunks++; total_rounds += rounds; round_hist[rounds < 69 ? rounds : 69]++;
    }
    batch0 = pos; t0 = t1;
  }
  if (memcmp(ob, ref, out)) return 11;
  bytes_total += out; members++;
  return 0;
}
int main(int argc, char **argv) {
  if (argc < 2) return 2;
  if (argc > 2) BATCH = atoi(argv[2]);
  FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
  fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t *buf = malloc(n + 16); if (fread(buf, 1, n, f) != n) return 2; memset(buf + n, 0, 16); in = buf;
  size_t pos = 0;
  while (pos + 18 <= n && in[pos] == 0x1f && in[pos + 1] == 0x8b) {
    int flg = in[pos + 3]; size_t q = pos + 10;
    if (flg & 4) q += 2 + in[q] + 256 * in[q + 1];
    if (flg & 8) { while (in[q]) ++q; ++q; }
    if (flg & 16) { while (in[q]) ++q; ++q; }
    if (flg & 2) q += 2;
    uint64_t p = (uint64_t)q * 8;
    ntk = 0;
    for (;;) {
      int final = bits(&p, 1), type = bits(&p, 2);
      if (type == 0) { p = (p + 7) & ~7ull; uint32_t len = bits(&p, 16); bits(&p, 16); for (uint32_t i = 0; i < len; ++i) push(0, 0, in[(p >> 3) + i]); p += 8ull * len; }
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
        int r, len, dist;
        for (;;) {
          uint64_t p_before = p;
          r = token(&ll, &dc, &p, &len, &dist);
          if (r) break;
          if (!len) { /* literal: recover the symbol */ uint64_t q2 = p_before; int s = decode(&ll, &q2); push(0, 0, (uint8_t)s); }
          else push(len, dist, 0);
        }
        if (r < 0) return 5;
      }
      if (final) break;
    }
    int rc = run_member(); if (rc) { printf("MODEL MISMATCH rc=%d in member %llu\n", rc, (unsigned long long)members); return rc; }
    pos = (size_t)((p + 7) >> 3) + 8;
  }
  uint64_t nm = n_far + n_win;
  printf("members %llu, %llu bytes: the token-centric model reproduces every byte\n", (unsigned long long)members, (unsigned long long)bytes_total);
  printf("tokens: literals %llu, matches from flushed output %llu (%.1f %% of matches; %.1f %% of them > 16 bytes), matches inside the window %llu (%.1f %%; %.1f %% of them straddle the batch start, %.2f %% self-overlapping, %.1f %% > 16 bytes)\n",
         (unsigned long long)n_lit, (unsigned long long)n_far, 100.0 * n_far / nm, 100.0 * far_long / n_far, (unsigned long long)n_win, 100.0 * n_win / nm, 100.0 * straddle / n_win, 100.0 * n_self / n_win, 100.0 * n_long / n_win);
  printf("chunks of 64 tokens: %llu (%.1f output bytes each); rounds of phase 2 per chunk: mean %.2f\n", (unsigned long long)chunks, (double)bytes_total / chunks, (double)total_rounds / chunks);
  uint64_t acc = 0;
  for (int r = 0; r < 70; ++r) { acc += round_hist[r]; if (r <= 8 || r == 12 || r == 16 || r == 24 || r == 32 || r == 69) printf("  <= %2d rounds: %.2f %%\n", r, 100.0 * acc / chunks); }
  return 0;
}
