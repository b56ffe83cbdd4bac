// How long do the markers of a chunk live?  (The chunked decode of ONE long stream, sm_inflate.hpp, resolves a chunk to 16-bit
// symbols because a match may copy bytes that lie in front of the chunk -- "markers".  Once 32 KiB of output in a row hold no
// marker, none can ever appear again in that chunk: from there on the chunk could be resolved to plain bytes, and its last
// window would not depend on the chunk before it.)  This tool inflates a raw DEFLATE stream on the CPU, remembers for every
// output byte the position of the LITERAL it is a copy of, cuts the output every `chunk` bytes and reports, per chunk, the
// first position T at which the last 32 KiB are marker-free.
//   gcc -O2 -o marker_decay marker_decay.c && ./marker_decay stream.deflate 131072
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static const uint8_t *in_; static size_t n_; static uint64_t p_;
static inline uint32_t bits(int k) { uint32_t v = 0; for (int i = 0; i < k; ++i, ++p_) v |= (uint32_t)((p_ < n_ * 8 ? (in_[p_ >> 3] >> (p_ & 7)) & 1 : 0)) << i; return v; }
typedef struct { uint16_t count[16], sym[320]; } Code;
static void build(Code *c, const uint8_t *lens, int n) {
  uint16_t offs[16]; memset(c->count, 0, sizeof c->count);
  for (int i = 0; i < n; ++i) c->count[lens[i]]++;
  c->count[0] = 0; offs[1] = 0;
  for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + c->count[l];
  for (int i = 0; i < n; ++i) if (lens[i]) c->sym[offs[lens[i]]++] = (uint16_t)i;
}
static int decode(const Code *c) {
  int code = 0, first = 0, index = 0;
  for (int l = 1; l <= 15; ++l) {
    code |= (int)bits(1);
    int cnt = c->count[l];
    if (code - cnt < first) return c->sym[index + (code - first)];
    index += cnt; first += cnt; first <<= 1; code <<= 1;
  }
  return -1;
}
static const uint16_t LB[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t LX[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint16_t DB[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const uint8_t DX[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
int main(int argc, char **argv) {
  if (argc < 3) return 2;
  FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
  fseek(f, 0, SEEK_END); n_ = ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t *buf = malloc(n_); if (fread(buf, 1, n_, f) != n_) return 2; in_ = buf;
  const size_t chunk = strtoul(argv[2], 0, 10), cap = (size_t)1 << 31;
  uint32_t *org = malloc(cap * 4); size_t o = 0;
  Code ll, dd; uint8_t lens[320];
  for (int last = 0; !last;) {
    last = bits(1); int type = bits(2);
    if (type == 0) { p_ = (p_ + 7) & ~7ull; uint32_t len = bits(16); bits(16); for (uint32_t i = 0; i < len; ++i, p_ += 8) org[o] = (uint32_t)o, ++o; continue; }
    if (type == 1) { for (int i = 0; i < 288; ++i) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8; build(&ll, lens, 288); for (int i = 0; i < 30; ++i) lens[i] = 5; build(&dd, lens, 30); }
    else {
      int hlit = bits(5) + 257, hdist = bits(5) + 1, ncl = bits(4) + 4; static const uint8_t ord[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
      uint8_t cl[19] = {0}; for (int i = 0; i < ncl; ++i) cl[ord[i]] = bits(3);
      Code cc; build(&cc, cl, 19);
      for (int i = 0; i < hlit + hdist;) { int s = decode(&cc); if (s < 16) lens[i++] = s; else { int rep = s == 16 ? 3 + bits(2) : s == 17 ? 3 + bits(3) : 11 + bits(7); uint8_t v = s == 16 ? lens[i - 1] : 0; while (rep--) lens[i++] = v; } }
      build(&ll, lens, hlit); build(&dd, lens + hlit, hdist);
    }
    for (;;) {
      int s = decode(&ll); if (s < 0) return 3;
      if (s < 256) { org[o] = (uint32_t)o; ++o; continue; }
      if (s == 256) break;
      int len = LB[s - 257] + bits(LX[s - 257]); int ds = decode(&dd); size_t dist = DB[ds] + bits(DX[ds]);
      for (int i = 0; i < len; ++i, ++o) org[o] = org[o - dist];
      if (o + 300 > cap) { last = 1; break; }
    }
  }
  // per chunk: T = first position (relative to the chunk's start) with 32 KiB of marker-free output behind it
  size_t nch = 0, hist[9] = {0}; double sumT = 0, markers = 0, bytes = 0;
  static const size_t edge[8] = {32768, 40960, 49152, 65536, 81920, 98304, 114688, 131072};
  for (size_t s = chunk; s + chunk <= o; s += chunk) {
    size_t lastm = 0, any = 0, T = chunk + 1, cnt = 0;
    for (size_t i = 0; i < chunk; ++i) {
      if (org[s + i] < s) { lastm = i; any = 1; ++cnt; }
      else if (i + 1 >= 32768 && (!any || i - lastm >= 32768)) { T = i + 1; break; }
    }
    // (the markers behind T: none by construction; count those in front)
    for (size_t i = (T > chunk ? chunk : T); i < chunk; ++i) if (org[s + i] < s) ++cnt;
    ++nch; markers += cnt; bytes += chunk;
    if (T <= chunk) sumT += T; else sumT += chunk;
    int b = 8; for (int k = 0; k < 8; ++k) if (T <= edge[k]) { b = k; break; }
    hist[b]++;
  }
  printf("%zu bytes out, %zu chunks of %zu: markers are %.1f %% of all symbols; the window is marker-free for good after T bytes of the chunk:\n", o, nch, chunk, 100 * markers / bytes);
  for (int k = 0; k < 8; ++k) printf("  T <= %6zu: %5.1f %%\n", edge[k], 100.0 * hist[k] / nch);
  printf("  never inside the chunk: %5.1f %%   (mean T, capped at the chunk: %.0f)\n", 100.0 * hist[8] / nch, sumT / nch);
  return 0;
}
