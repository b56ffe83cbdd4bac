// resolver_emu.cc -- the resolver's DEVICE code (archive_amd/csrc/inflate_par.hpp: resolve_member and everything under
// it) executed on the CPU by 64 threads (tests/emu/wave_emu.hpp), on token streams made from real DEFLATE data, and
// compared byte for byte with a sequential LZ77 replay.  Test infrastructure only.
//
//   g++ -std=c++17 -O2 -pthread -o resolver_emu tests/emu/resolver_emu.cc
//   resolver_emu <file of concatenated gzip members> [seed]
//   -DEMU_WG: the workgroup-per-member resolver (inflate_res_wg.hpp): WG_WAVES x 64 threads, one LDS ring
#define AHIP_HOST_EMU 1
#ifdef EMU_WG
#include "../../archive_amd/csrc/inflate_res_wg.hpp"
#else
#include "../../archive_amd/csrc/inflate_par.hpp"
#endif
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <thread>
#include <vector>
using namespace ahip;

// ---- a plain DEFLATE tokenizer (bit by bit, canonical codes) ----
struct Code { uint16_t count[16], first[16], offs[16], sym[320]; int maxlen; };
static const uint8_t *in_; static size_t n_;
static inline uint32_t bit(uint64_t p) { return p < n_ * 8 ? (in_[p >> 3] >> (p & 7)) & 1 : 0; }
static uint32_t bits(uint64_t *p, int k) { uint32_t v = 0; for (int i = 0; i < k; ++i) v |= bit((*p)++) << i; return v; }
static void build(Code *c, const uint8_t *lens, int cnt) {
  memset(c, 0, sizeof *c);
  for (int i = 0; i < cnt; ++i) c->count[lens[i]]++;
  c->count[0] = 0;
  int code = 0, off = 0;
  for (int l = 1; l < 16; ++l) { c->first[l] = code; c->offs[l] = off; if (c->count[l]) c->maxlen = l; code = (code + c->count[l]) << 1; off += c->count[l]; }
  uint16_t next[16]; memcpy(next, c->offs, sizeof next);
  for (int i = 0; i < cnt; ++i) if (lens[i]) c->sym[next[lens[i]]++] = i;
}
static int decode(const Code *c, uint64_t *p) {
  int code = 0;
  for (int l = 1; l <= c->maxlen; ++l) { code = (code << 1) | bit((*p)++); int idx = code - c->first[l]; if (idx >= 0 && idx < c->count[l]) return c->sym[c->offs[l] + idx]; }
  return -1;
}
static const uint16_t LB[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t LX[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint16_t DB[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const uint8_t DX[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};

#ifdef EMU_SYM   // 16-bit symbols: the member's first part plays "output in front of the chunk", the rest is resolved as a chunk
typedef u16 Elem;
#else
typedef u8 Elem;
#endif
#ifdef EMU_WG
static ResWgLds PW;               // the workgroup's LDS
static wave_emu::Ctx wave_ctx[8];  // a context per wave
#else
static ResLdsT<Elem> P;  // the wave's LDS
#endif

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1;
  FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
  fseek(f, 0, SEEK_END); n_ = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> buf(n_ + 64, 0); if (fread(buf.data(), 1, n_, f) != n_) return 2; in_ = buf.data();
  std::vector<uint8_t> out, ref;
  size_t pos = 0, members = 0; uint64_t toks_total = 0, stored_recs = 0, runs_total = 0, markers_total = 0;
  while (pos + 18 <= n_ && in_[pos] == 0x1f && in_[pos + 1] == 0x8b) {
    int flg = in_[pos + 3]; size_t q = pos + 10;
    if (flg & 4) q += 2 + in_[q] + 256 * in_[q + 1];
    if (flg & 8) { while (in_[q]) ++q; ++q; }
    if (flg & 16) { while (in_[q]) ++q; ++q; }
    if (flg & 2) q += 2;
    uint64_t p = (uint64_t)q * 8;
    std::vector<uint32_t> tok;            // the member's token stream as step words (literal / len << 16 | dist)
    std::vector<size_t> rec_at;           // indices where a stored block sits: one word STORED_MARK | len, then the offset (lo, hi)
    const uint32_t STORED_MARK = 0x60000000u;
    const size_t out0 = ref.size();
    for (;;) {
      int final = bits(&p, 1), type = bits(&p, 2);
      if (type == 0) {
        p = (p + 7) & ~7ull; uint32_t len = bits(&p, 16); bits(&p, 16);
        const uint64_t byte = p >> 3;
        if (len >= 3) { rec_at.push_back(tok.size()); tok.push_back(STORED_MARK | len); tok.push_back((uint32_t)byte); tok.push_back((uint32_t)(byte >> 32)); stored_recs++; }
        else for (uint32_t i = 0; i < len; ++i) tok.push_back(0x80000000u | ((uint32_t)in_[byte + i] << 16));
        for (uint32_t i = 0; i < len; ++i) ref.push_back(in_[byte + i]);
        p += 8ull * len;
      } else if (type == 3) return 3;
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
        for (;;) {
          int s = decode(&ll, &p);
          if (s < 0 || s > 285) return 5;
          if (s == 256) break;
          if (s < 256) { tok.push_back(0x80000000u | ((uint32_t)s << 16)); ref.push_back((uint8_t)s); continue; }
          uint32_t len = LB[s - 257] + bits(&p, LX[s - 257]);
          int d = decode(&dc, &p); if (d < 0 || d > 29) return 5;
          uint32_t dist = DB[d] + bits(&p, DX[d]);
          if (dist > ref.size() - out0) return 6;  // (reaches into an earlier member: not made by zlib)
          tok.push_back((len << 16) | dist);
          for (uint32_t i = 0; i < len; ++i) ref.push_back(ref[ref.size() - dist]);
        }
      }
      if (final) break;
    }
    // ---- runs: pseudo-random lengths, scattered over the area with gaps; a stored block is a directory entry of its
    //      own; every token is recorded as (end of the run's output so far) << 16 | payload, like the tokenizer does.
    //      Some runs are flagged DF_BIG although they are short (the flag only selects how offsets are found). ----
    std::vector<uint32_t> area; std::vector<DirEnt> dir;
    size_t t = 0, ri = 0;
    uint64_t opos = 0;
    uint64_t front = 0;  // bytes of the member in front of the resolved part
#ifdef EMU_SYM
    {  // skip whole tokens until about a third of the member (at most 40 000 bytes) lies in front of the chunk
      const uint64_t want_front = std::min<uint64_t>(40000, (ref.size() - out0) / 3);
      while (t < tok.size() && front < want_front) {
        if (ri < rec_at.size() && rec_at[ri] == t) { front += tok[t] & 0xffffu; t += 3; ++ri; continue; }
        front += (int32_t)tok[t] < 0 ? 1u : (tok[t] >> 16);
        ++t;
      }
    }
#endif
    while (t < tok.size()) {
      seed = seed * 1664525u + 1013904223u;
      while (ri < rec_at.size() && rec_at[ri] < t) ++ri;
      if (ri < rec_at.size() && rec_at[ri] == t) {  // stored block
        const uint32_t len = tok[t] & 0xffffu;
        area.resize(area.size() + (seed >> 4) % 7, 0xdeadbeefu);
        dir.push_back(make_uint4((unsigned)area.size(), len | DF_STORED, (unsigned)opos, (unsigned)(opos >> 32)));
        area.push_back(tok[t + 1]); area.push_back(tok[t + 2]);
        opos += len; t += 3;
        continue;
      }
      size_t want = 1 + (seed >> 8) % ((seed >> 28) == 0 ? 400 : 90), end = t + want < tok.size() ? t + want : tok.size();
      if (ri < rec_at.size() && rec_at[ri] < end) end = rec_at[ri];
      area.resize(area.size() + (seed >> 4) % 7, 0xdeadbeefu);  // a gap
      const size_t a0 = area.size();
      uint64_t bytes = 0;
      for (size_t i = t; i < end; ++i) {
        const uint32_t w = tok[i];
        bytes += (int32_t)w < 0 ? 1u : (w >> 16);
        area.push_back(rec_word((u32)bytes, w));
      }
      const bool big = bytes > 0xffffu || ((seed >> 20) & 7) == 0;
      dir.push_back(make_uint4((unsigned)a0, (unsigned)(end - t) | (big ? DF_BIG : 0u), (unsigned)opos, (unsigned)(opos >> 32)));
      opos += bytes;
      t = end;
    }
    runs_total += dir.size(); toks_total += tok.size();
    const size_t produced = ref.size() - out0 - front;
    // (the output starts at every alignment in turn: the ring goes out in 16-byte units aligned in global memory)
    const size_t mis = members % 16;
    std::vector<Elem> ebuf(produced + 64 + 16 + 32, (Elem)0xEE);
    Elem *eout_p = (Elem *)(((uintptr_t)ebuf.data() + 15) & ~(uintptr_t)15) + mis;
    u32 cyc_all[64][8] = {};
    std::vector<std::thread> th;
#ifdef EMU_WG
    wave_emu::wg_threads = (int)WG_THREADS;
    bool ok_all[WG_THREADS];
    for (int t = 0; t < (int)WG_THREADS; ++t)
      th.emplace_back([&, t]() {
        wave_emu::lane = t & 63; wave_emu::wave = t >> 6; wave_emu::cur = &wave_ctx[t >> 6];
        ok_all[t] = resolve_member_wg(PW, in_, area.data(), dir.data(), (u32)dir.size(), eout_p, t >> 6, t & 63);
      });
    for (auto &x : th) x.join();
    for (int t = 0; t < (int)WG_THREADS; ++t) if (!ok_all[t]) { printf("LOOP BOUND reached in member %zu (thread %d)\n", members, t); return 1; }
    // nothing in front of the member's first byte or behind its last one may have been touched
    for (size_t i = 0; i < mis; ++i) if (eout_p[-(ptrdiff_t)(i + 1)] != (Elem)0xEE) { printf("WRITE IN FRONT of member %zu\n", members); return 1; }
    for (size_t i = 0; i < 32; ++i) if (eout_p[produced + i] != (Elem)0xEE) { printf("WRITE BEHIND member %zu (+%zu)\n", members, i); return 1; }
#else
    for (int l = 0; l < 64; ++l)
      th.emplace_back([&, l]() { wave_emu::lane = l; resolve_member<Elem>(P, in_, area.data(), dir.data(), (u32)dir.size(), eout_p, cyc_all[l], l); });
    for (auto &x : th) x.join();
#endif
    const Elem *eout = eout_p;
    const uint8_t *want = ref.data() + out0 + front;
    for (size_t i = 0; i < produced; ++i) {
      uint32_t v = eout[i];
      if (sizeof(Elem) == 2 && v >= SYM_MARK) {  // a marker: byte j of the 32 KiB in front of the chunk
        const int64_t j = (int64_t)v - SYM_MARK, at = (int64_t)front - 32768 + j;
        if (at < 0 || at >= (int64_t)front) { printf("MARKER out of range in member %zu at %zu: %x (front %llu)\n", members, i, v, (unsigned long long)front); return 1; }
        v = ref[out0 + (size_t)at];
        markers_total++;
      }
      if (v != want[i]) {
        printf("MISMATCH in member %zu at element %zu of %zu (got %02x want %02x)\n", members, i, produced, v, want[i]);
        // which directory entry / token makes that byte
        uint64_t o = 0;
        for (size_t e = 0; e < dir.size(); ++e) {
          const uint32_t cnt = dir[e].y & DF_CNT;
          if (dir[e].y & DF_STORED) { if (i < o + cnt) { printf("  stored entry %zu at %llu + %u\n", e, (unsigned long long)o, cnt); break; } o += cnt; continue; }
          uint32_t pe = 0; bool hit = false;
          for (uint32_t k = 0; k < cnt; ++k) {
            const uint32_t w = area[dir[e].x + k], len = ((w >> 16) - pe) & 0xffffu;
            if (i < o + len) { printf("  entry %zu (%s, %u tokens, starts at %llu) token %u: at %llu, %u bytes, %s %u\n", e, (dir[e].y & DF_BIG) ? "BIG" : "plain", cnt, (unsigned long long)dir[e].z, k,
                                      (unsigned long long)o, len, (w & REC_LIT) ? "literal" : "distance", (w & REC_LIT) ? (w & 0xff) : (w & 0x7fff) + 1); hit = true; break; }
            o += len; pe = w >> 16;
          }
          if (hit) break;
        }
        return 1;
      }
    }
    members++;
    pos = (size_t)((p + 7) >> 3) + 8;
  }
#ifdef EMU_WG
  const char *what = "workgroup-per-member resolver";
#else
  const char *what = sizeof(Elem) == 2 ? "token-centric resolver, 16-bit symbols" : "token-centric resolver";
#endif
  if (sizeof(Elem) == 2) printf("markers checked: %llu\n", (unsigned long long)markers_total);
  printf("resolver emu ok [%s]: %zu members, %zu bytes, %llu token words in %llu runs, %llu stored records\n", what, members, ref.size(),
         (unsigned long long)toks_total, (unsigned long long)runs_total, (unsigned long long)stored_recs);
  return members ? 0 : 7;
}
