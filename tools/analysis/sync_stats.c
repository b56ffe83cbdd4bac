/* Dev tool (CPU): how fast does a DEFLATE Huffman stream re-synchronise?
 *
 *   sync_stats <file with concatenated gzip members>
 *
 * For every dynamic / fixed Huffman block the true token boundaries are recorded (bit positions).  Then, at every
 * 512-bit item boundary B inside the block (the tokenizer's work unit, inflate_par.hpp):
 *   - a blind decode is started S bits in front of B for several S; its end (first token boundary >= B) is the
 *     PREDICTED start of the next item; the prediction is right when that position is a true boundary;
 *   - a blind decode started exactly at B is followed until it first lands on a true boundary: the distance is how
 *     far a run that started on a guess has to be repaired (the splice design of DESIGN.md section 12).
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
/* one token at *p; returns 0 literal/match, 1 end of block, -1 bad */
static int token(const Code *ll, const Code *dc, uint64_t *p) {
  int s = decode(ll, p);
  if (s < 0 || s > 285) return -1;
  if (s < 256) return 0;
  if (s == 256) return 1;
  *p += LX[s - 257];
  int d = decode(dc, p);
  if (d < 0 || d > 29) return -1;
  *p += DX[d];
  return 0;
}
#define NS 8
static const int SPEC[NS] = {32, 64, 96, 128, 192, 256, 384, 512};
static uint64_t spec_ok[NS], spec_all, hist[64], splice_n, splice_sum, splice_never, tokens, blocks, tok_bits;
static uint8_t *mark; static size_t mark_cap;
static void study(const Code *ll, const Code *dc, uint64_t p0, uint64_t p1) {  /* true boundaries marked in [p0, p1) */
  for (uint64_t B = ((p0 >> 9) + 2) << 9; B + 512 < p1; B += 512) {
    spec_all++;
    for (int k = 0; k < NS; ++k) {
      if (B < p0 + (uint64_t)SPEC[k]) continue;
      uint64_t p = B - SPEC[k]; int r = 0;
      while (p < B && (r = token(ll, dc, &p)) == 0) {}
      if (r == 0 && p < p1 && (mark[(p - p0) >> 3] >> ((p - p0) & 7) & 1)) spec_ok[k]++;
    }
    uint64_t p = B; int r = 0; int found = 0;
    while (p < B + 2048 && p < p1) {
      if (mark[(p - p0) >> 3] >> ((p - p0) & 7) & 1) { found = 1; break; }
      if ((r = token(ll, dc, &p)) != 0) break;
    }
    if (found) { uint64_t d = p - B; splice_n++; splice_sum += d; hist[d / 16 < 63 ? d / 16 : 63]++; } else splice_never++;
  }
}
int main(int argc, char **argv) {
  if (argc < 2) return 2;
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
        uint64_t p0 = p; int r;
        /* pass 1: find the end; pass 2: mark */
        uint64_t t = p; uint64_t nt = 0; while ((r = token(&ll, &dc, &t)) == 0) nt++;
        if (r < 0) return 5;
        size_t need = (t - p0) / 8 + 2; if (need > mark_cap) { mark = realloc(mark, need); mark_cap = need; }
        memset(mark, 0, need);
        t = p0; for (;;) { mark[(t - p0) >> 3] |= 1 << ((t - p0) & 7); if (token(&ll, &dc, &t)) break; }
        tokens += nt; blocks++; tok_bits += t - p0;
        study(&ll, &dc, p0, t);
        p = t;
      }
      if (final) break;
    }
    pos = (size_t)((p + 7) >> 3) + 8;
  }
  printf("blocks %llu  tokens %llu  bits/token %.2f  item boundaries studied %llu\n", (unsigned long long)blocks, (unsigned long long)tokens, (double)tok_bits / tokens, (unsigned long long)spec_all);
  printf("blind run of S bits in front of an item boundary predicts the next item's start correctly:\n");
  for (int k = 0; k < NS; ++k) printf("  S = %3d bits: %.1f %%\n", SPEC[k], 100.0 * spec_ok[k] / spec_all);
  printf("blind run from the boundary itself: first true token boundary it lands on (repair length of a spliced run):\n");
  printf("  mean %.1f bits; never within 2048 bits: %.2f %%\n", (double)splice_sum / splice_n, 100.0 * splice_never / (splice_n + splice_never));
  uint64_t acc = 0;
  for (int b = 0; b < 64; ++b) { acc += hist[b]; if (b < 12 || b % 8 == 7) printf("  <= %4d bits: %.1f %%\n", b * 16 + 15, 100.0 * acc / splice_n); }
  return 0;
}
