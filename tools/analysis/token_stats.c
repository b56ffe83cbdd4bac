/* Dev tool (CPU): what the resolver has to do -- token mix, distances, dependencies inside a 64-byte group.
 *
 *   token_stats <file with concatenated gzip members> [batch_bytes = 1728]
 *
 * Output bytes are classed by where the resolver (inflate_par.hpp, resolve_member) finds them: a literal, an earlier byte
 * of the same batch window (LDS), or flushed output in front of the batch (HBM / L2: "far").  A 64-byte group needs the
 * pointer-doubling loop when a byte's source lies in the same group.  Nothing here is linked into the product or the tests. */
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

static uint64_t far_len_hist[6], far_len_bytes[6], win_len_hist[6], win_len_bytes[6];
static uint64_t n_lit, n_match, len_sum, dh[8], by_lit, by_win, by_far, far_tok, far_tok_le16, groups, groups_dep, depth_sum, tok_in_group_sum, members;
static uint32_t BATCH = 1728;
/* per member: source index of every output byte (-1 literal) */
static int32_t *src; static size_t src_cap;
static void member_done(size_t out) {
  /* batches of BATCH bytes (the GPU cuts at token boundaries; close enough), groups of 64 */
  for (size_t g = 0; g + 64 <= out; g += 64) {
    groups++;
    int dep = 0, depth = 0;
    int d[64];
    for (int i = 0; i < 64; ++i) {
      int32_t sidx = src[g + i];
      d[i] = 0;
      if (sidx >= 0 && (size_t)sidx >= g) { dep = 1; d[i] = d[sidx - g] + 1; if (d[i] > depth) depth = d[i]; }
    }
    if (dep) { groups_dep++; int it = 0; while ((1 << it) < depth + 1) ++it; depth_sum += it; }
  }
  members++;
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
    size_t out = 0;
    for (;;) {
      int final = bits(&p, 1), type = bits(&p, 2);
      if (type == 0) { p = (p + 7) & ~7ull; uint32_t len = bits(&p, 16); bits(&p, 16); p += 8ull * len;
        if (out + len > src_cap) { src_cap = (out + len) * 2; src = realloc(src, src_cap * 4); }
        for (uint32_t i = 0; i < len; ++i) src[out++] = -1; }
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
        while ((r = token(&ll, &dc, &p, &len, &dist)) == 0) {
          if (out + 300 > src_cap) { src_cap = (out + 300) * 2; src = realloc(src, src_cap * 4); }
          if (!len) { n_lit++; by_lit++; src[out++] = -1; continue; }
          n_match++; len_sum += len;
          dh[dist < 4 ? 0 : dist < 16 ? 1 : dist < 64 ? 2 : dist < 256 ? 3 : dist < 1728 ? 4 : dist < 8192 ? 5 : 6]++;
          size_t batch0 = out - out % BATCH;
          int far = (size_t)dist > out - batch0;  /* source starts in front of the batch window */
          { int b = len <= 8 ? 0 : len <= 16 ? 1 : len <= 32 ? 2 : len <= 64 ? 3 : len <= 128 ? 4 : 5; if (far) { far_len_hist[b]++; far_len_bytes[b] += len; } else { win_len_hist[b]++; win_len_bytes[b] += len; } }
          if (far) { far_tok++; if (len <= 16 && (size_t)dist >= out - batch0 + 16) far_tok_le16++; }
          for (int i = 0; i < len; ++i) { size_t sidx = out - dist; src[out] = (int32_t)sidx; if (sidx < batch0) by_far++; else by_win++; out++; }
        }
        if (r < 0) return 5;
      }
      if (final) break;
    }
    member_done(out);
    pos = (size_t)((p + 7) >> 3) + 8;
  }
  uint64_t bytes = by_lit + by_win + by_far, toks = n_lit + n_match;
  printf("members %llu  tokens %llu (%.1f %% literals)  bytes per token %.2f  mean match length %.2f\n", (unsigned long long)members, (unsigned long long)toks, 100.0 * n_lit / toks, (double)bytes / toks, (double)len_sum / n_match);
  const char *nm[7] = {"< 4", "< 16", "< 64", "< 256", "< 1728", "< 8192", "<= 32768"};
  printf("match distances:"); for (int i = 0; i < 7; ++i) printf("  %s: %.1f %%", nm[i], 100.0 * dh[i] / n_match); printf("\n");
  printf("output bytes: literal %.1f %%, copied inside the %u-byte batch window %.1f %%, copied from flushed output %.1f %%\n", 100.0 * by_lit / bytes, BATCH, 100.0 * by_win / bytes, 100.0 * by_far / bytes);
  printf("matches whose source starts in front of the batch: %.1f %% of the matches; %.1f %% of those are <= 16 bytes with the whole 16-byte load in flushed output (the per-token deposit path)\n", 100.0 * far_tok / n_match, 100.0 * far_tok_le16 / far_tok);
  { const char *lb[6] = {"<= 8", "<= 16", "<= 32", "<= 64", "<= 128", "<= 258"}; uint64_t ft = 0, fb = 0, wt = 0, wb = 0;
    for (int i = 0; i < 6; ++i) { ft += far_len_hist[i]; fb += far_len_bytes[i]; wt += win_len_hist[i]; wb += win_len_bytes[i]; }
    printf("match lengths (share of matches / of their bytes), source in flushed output:"); for (int i = 0; i < 6; ++i) printf("  %s: %.1f / %.1f %%", lb[i], 100.0 * far_len_hist[i] / ft, 100.0 * far_len_bytes[i] / fb); printf("\n");
    printf("match lengths (share of matches / of their bytes), source inside the batch window:"); for (int i = 0; i < 6; ++i) printf("  %s: %.1f / %.1f %%", lb[i], 100.0 * win_len_hist[i] / wt, 100.0 * win_len_bytes[i] / wb); printf("\n"); }
  printf("64-byte groups: %llu, with a source inside the same group %.1f %% (mean pointer-doubling steps when so: %.2f)\n", (unsigned long long)groups, 100.0 * groups_dep / groups, groups_dep ? (double)depth_sum / groups_dep : 0.0);
  return 0;
}
