/* Dev tool (CPU): model of the WORKGROUP-per-member resolver (round 6) on real token streams.
 *
 *   wg_resolver_model <file with concatenated gzip members> [waves = 4] [chunk = 64]
 *
 * W waves share one member: wave w takes the chunks (of `chunk` consecutive tokens) w, w + W, ...  The whole 32 KiB of
 * history is in LDS.  A chunk first does what depends on nobody in flight (phase 1: literals, matches whose source ends in
 * front of the oldest chunk that may still be in flight, W - 1 chunks back), waits for the chunk in front of it to be
 * complete, and finishes its other matches in LEVELS of exact dependence (level of a pending match = 1 + the highest level
 * among the pending matches of the same chunk whose destination its source overlaps; a match that only needs earlier
 * chunks is level 1).  The levels of all chunks, one behind the other, are the member's dependent chain of LDS round trips.
 * Reported: tokens by class, pending matches per chunk, levels per chunk, chain links per member.  Checked byte for byte
 * against a sequential replay.  Nothing here is linked into the product or the tests. */
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
static int W = 4, CH = 64;
static uint64_t chunks, members, bytes_total, n_lit, n_match, n_p1, n_cross, n_intra, n_S, n_L, n_H, levels_total, lvl_hist[70], chain_max, chain_sum, span_max, big_chunks;
static uint64_t y_execs, y_hist[70], y_intra, len_hist[8];
static int SEG = 256; static uint64_t seg_count, seg_levels, seg_lvl_hist[70], seg_pending, cons_rounds;
static uint64_t dist_hist[16], depth_sum, depth_max_all; static uint16_t *dep; static size_t dep_cap;
static void push(uint32_t len, uint32_t dist, uint8_t lit) {
  if (ntk == tk_cap) { tk_cap = tk_cap ? tk_cap * 2 : 1 << 16; tk = realloc(tk, tk_cap * sizeof(Tok)); }
  tk[ntk].len = len; tk[ntk].dist = dist; tk[ntk].lit = lit; ntk++;
}
static void copy_match(size_t d, const Tok *t) { for (uint32_t k = 0; k < t->len; ++k) ob[d + k] = ob[d + k - t->dist]; }
static int run_member(void) {
  size_t out = 0;
  for (size_t i = 0; i < ntk; ++i) out += tk[i].len ? tk[i].len : 1;
  if (out + 64 > ob_cap) { ob_cap = out * 2 + 64; ref = realloc(ref, ob_cap); ob = realloc(ob, ob_cap); }
  size_t o = 0;
  for (size_t i = 0; i < ntk; ++i) {
    if (!tk[i].len) ref[o++] = tk[i].lit;
    else { if (tk[i].dist > o) return 10; for (uint32_t k = 0; k < tk[i].len; ++k, ++o) ref[o] = ref[o - tk[i].dist]; }
  }
  if (out > dep_cap) { dep_cap = out * 2; dep = realloc(dep, dep_cap * 2); }
  { size_t q = 0; unsigned mx = 0; for (size_t i = 0; i < ntk; ++i) { if (!tk[i].len) dep[q++] = 0; else for (uint32_t k = 0; k < tk[i].len; ++k, ++q) { dep[q] = dep[q - tk[i].dist] + 1; if (dep[q] > mx) mx = dep[q]; } } depth_sum += mx; if (mx > depth_max_all) depth_max_all = mx; }
  memset(ob, 0xEE, out);
  size_t nch = (ntk + CH - 1) / CH;
  size_t *cstart = malloc((nch + 1) * sizeof(size_t));
  size_t pos = 0;
  for (size_t c = 0; c < nch; ++c) { cstart[c] = pos; for (size_t i = c * CH; i < (c + 1) * CH && i < ntk; ++i) pos += tk[i].len ? tk[i].len : 1; }
  cstart[nch] = pos;
  uint64_t chain = 0;
  static size_t off[256]; static int lvl[256];
  for (size_t c = 0; c < nch; ++c) {
    const size_t t0 = c * CH, t1 = t0 + CH < ntk ? t0 + CH : ntk;
    const size_t safe = cstart[c >= (size_t)(W - 1) ? c - (W - 1) : 0];  /* everything in front of this is complete when the chunk starts */
    const size_t c0 = cstart[c];
    if (cstart[c + 1] - c0 > span_max) span_max = cstart[c + 1] - c0;
    if (cstart[c + 1] - c0 > 2048) big_chunks++;
    size_t p = c0; int maxl = 0;
    for (size_t i = t0; i < t1; ++i) { off[i - t0] = p; p += tk[i].len ? tk[i].len : 1; }
    /* phase 1 */
    for (size_t i = t0; i < t1; ++i) {
      const Tok *t = &tk[i]; const size_t d = off[i - t0];
      lvl[i - t0] = 0;
      if (!t->len) { ob[d] = t->lit; n_lit++; continue; }
      n_match++;
      { int b = 0; uint32_t x = t->dist; while (x > 1) { x >>= 1; b++; } dist_hist[b]++; }
      if (t->dist < 16 && t->dist < t->len) n_H++; else if (t->len > 32 || t->dist < t->len) n_L++; else n_S++;
      const size_t s = d - t->dist, need_end = s + t->len < d ? s + t->len : d;  /* the part of the source that is not its own output */
      if (need_end <= safe) { copy_match(d, t); n_p1++; }
      else lvl[i - t0] = -1;  /* pending */
    }
    /* levels: in token order (a pending match's level depends on earlier pending matches of the chunk only) */
    for (size_t i = t0; i < t1; ++i) {
      if (lvl[i - t0] != -1) continue;
      const Tok *t = &tk[i]; const size_t d = off[i - t0], s = d - t->dist, e = s + t->len < d ? s + t->len : d;
      int l = 1;
      if (e > c0) {
        for (size_t j = t0; j < i; ++j) {
          if (lvl[j - t0] <= 0 || !tk[j].len) continue;
          const size_t dj = off[j - t0], ej = dj + tk[j].len;
          if (dj < e && ej > s && lvl[j - t0] + 1 > l) l = lvl[j - t0] + 1;
        }
        n_intra++;
      } else n_cross++;
      lvl[i - t0] = l;
      if (l > maxl) maxl = l;
    }
    /* execute level by level (all reads of a level before its writes would be the parallel round; sequential order inside a level is equivalent because a level's members do not overlap one another's sources) */
    for (int l = 1; l <= maxl; ++l)
      for (size_t i = t0; i < t1; ++i) if (lvl[i - t0] == l) copy_match(off[i - t0], &tk[i]);
    {  /* design Y as built: literals early, EVERY match of the chunk waits for the chunk in front; executions = exact levels
          over all matches of the chunk (level 1: no overlap with an earlier match of the chunk) */
      static int l2[256]; int mx2 = 0;
      for (size_t i = t0; i < t1; ++i) {
        const Tok *t = &tk[i]; l2[i - t0] = 0;
        if (!t->len) continue;
        { uint32_t L = t->len; len_hist[L == 3 ? 0 : L < 8 ? 1 : L == 8 ? 2 : L <= 16 ? 3 : L <= 32 ? 4 : 5]++; }
        const size_t d = off[i - t0], s = d - t->dist, e = s + t->len < d ? s + t->len : d;
        int l = 1;
        if (e > c0) {
          y_intra++;
          for (size_t j = t0; j < i; ++j) { if (!tk[j].len) continue; const size_t dj = off[j - t0], ej = dj + tk[j].len; if (dj < e && ej > s && l2[j - t0] + 1 > l) l = l2[j - t0] + 1; }
        }
        l2[i - t0] = l; if (l > mx2) mx2 = l;
      }
      y_execs += mx2; y_hist[mx2 < 69 ? mx2 : 69]++;
    }
    chunks++; levels_total += maxl; lvl_hist[maxl < 69 ? maxl : 69]++;
    chain += maxl;
  }
  free(cstart);
  if (memcmp(ob, ref, out)) return 11;
  {  /* segments of SEG tokens, exact levels by bytes: level of a match = 1 + the highest level among its source bytes inside the segment */
    static uint8_t *bl; static size_t bl_cap;
    if (out + 64 > bl_cap) { bl_cap = out * 2 + 64; bl = realloc(bl, bl_cap); }
    size_t q = 0;
    for (size_t t0 = 0; t0 < ntk; t0 += SEG) {
      const size_t t1 = t0 + SEG < ntk ? t0 + SEG : ntk, s0 = q;
      int mx = 0;
      for (size_t i = t0; i < t1; ++i) {
        const Tok *t = &tk[i];
        if (!t->len) { bl[q++] = 0; continue; }
        const size_t s = q - t->dist; int l = 0;
        for (uint32_t k = 0; k < t->len && s + k < q; ++k) if (s + k >= s0 && bl[s + k] > l) l = bl[s + k];
        const size_t e = s + t->len < q ? s + t->len : q;
        const int lv = e > s0 ? l + 1 : 0;   /* 0: source in front of the segment (phase A) */
        if (lv) seg_pending++;
        for (uint32_t k = 0; k < t->len; ++k) bl[q++] = (uint8_t)lv;
        if (lv > mx) mx = lv;
      }
      seg_count++; seg_levels += mx; seg_lvl_hist[mx < 69 ? mx : 69]++;
    }
  }
  bytes_total += out; members++;
  chain_sum += chain; if (chain > chain_max) chain_max = chain;
  return 0;
}
int main(int argc, char **argv) {
  if (argc < 2) return 2;
  if (argc > 2) W = atoi(argv[2]);
  if (argc > 3) CH = atoi(argv[3]);
  if (argc > 4) SEG = atoi(argv[4]);
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
  printf("members %llu, %llu bytes, W = %d waves, chunks of %d tokens: the model reproduces every byte\n", (unsigned long long)members, (unsigned long long)bytes_total, W, CH);
  printf("tokens per member %.0f: literals %.1f %%, matches %.1f %% (simple %.1f %%, loop %.1f %%, hard %.2f %% of matches)\n", (double)(n_lit + n_match) / members,
         100.0 * n_lit / (n_lit + n_match), 100.0 * n_match / (n_lit + n_match), 100.0 * n_S / n_match, 100.0 * n_L / n_match, 100.0 * n_H / n_match);
  printf("matches: phase 1 (source complete when the chunk starts) %.1f %%, waiting for a chunk in flight %.1f %%, source inside the own chunk %.1f %%\n",
         100.0 * n_p1 / n_match, 100.0 * n_cross / n_match, 100.0 * n_intra / n_match);
  printf("chunks per member %.1f (%.1f bytes each, largest %llu, %llu over 2 KiB); pending matches per chunk %.2f; levels per chunk: mean %.2f\n", (double)chunks / members, (double)bytes_total / chunks,
         (unsigned long long)span_max, (unsigned long long)big_chunks, (double)(n_cross + n_intra) / chunks, (double)levels_total / chunks);
  uint64_t acc = 0;
  for (int r = 0; r < 70; ++r) { acc += lvl_hist[r]; if (r <= 6 || r == 8 || r == 12 || r == 69) printf("  <= %2d levels: %.2f %%\n", r, 100.0 * acc / chunks); }
  printf("chain links (levels, one chunk behind the other) per member: mean %.0f, max %llu\n", (double)chain_sum / members, (unsigned long long)chain_max);
  printf("segments of %d tokens (sources in front of the segment final, exact dependences inside): %.1f per member, pending matches %.1f per segment, levels per segment mean %.2f, per member %.0f\n", SEG, (double)seg_count / members, (double)seg_pending / seg_count, (double)seg_levels / seg_count, (double)seg_levels / members);
  { uint64_t a2 = 0; for (int r = 0; r < 70; ++r) { a2 += seg_lvl_hist[r]; if (r <= 8 || r == 12 || r == 16 || r == 69) printf("  <= %2d levels: %.2f %%\n", r, 100.0 * a2 / seg_count); } }
  printf("design Y (literals early, all matches behind the chunk in front): copy executions per chunk mean %.2f (intra-chunk sources %.2f per chunk)\n", (double)y_execs / chunks, (double)y_intra / chunks);
  { uint64_t a3 = 0; for (int r = 0; r < 70; ++r) { a3 += y_hist[r]; if (r <= 6) printf("  <= %2d executions: %.2f %%\n", r, 100.0 * a3 / chunks); } }
  printf("match lengths: 3: %.1f %%, 4-7: %.1f %%, 8: %.1f %%, 9-16: %.1f %%, 17-32: %.1f %%, > 32: %.1f %%\n", 100.0 * len_hist[0] / n_match, 100.0 * len_hist[1] / n_match, 100.0 * len_hist[2] / n_match, 100.0 * len_hist[3] / n_match, 100.0 * len_hist[4] / n_match, 100.0 * len_hist[5] / n_match);
  printf("true byte-level dependence depth per member: mean %.0f, max %llu\n", (double)depth_sum / members, (unsigned long long)depth_max_all);
  printf("distance histogram (log2 buckets, %% of matches):");
  for (int b = 0; b < 16; ++b) printf(" %d:%.1f", b, 100.0 * dist_hist[b] / n_match);
  printf("\n");
  return 0;
}
