// inflate_emu.cc -- the WHOLE device path of one member on the CPU: the flow tokenizer (inflate_member<false, true>: block
// headers, table build, huffman_block_tokenize, the serial token emitter) and then the resolver (resolve_member), both
// executed by 64 threads as the 64 lanes of a wave (tests/emu/wave_emu.hpp), glued together the way
// inflate_tokenize_kernel / inflate_resolve_kernel of archive_hip.hip do it.  Test infrastructure only.
//
//   g++ -std=c++17 -O2 -pthread -o inflate_emu tests/emu/inflate_emu.cc
//   inflate_emu <gzip members> <expected plain bytes> <sizes: one decimal per member, whitespace separated, or the word auto>
// (auto: a member's window is what is left of the expected bytes -- for streams whose members are not told apart beforehand)
#define AHIP_HOST_EMU 1
#include "../../archive_amd/csrc/inflate_par.hpp"
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
using namespace ahip;

static struct TokKernelLds { WaveLds w; TokLds p; } TL;  // the tokenizer wave's LDS
static ResLds PL;                                         // the resolver wave's LDS

static std::vector<uint8_t> slurp(const char *path) {
  FILE *f = fopen(path, "rb"); if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> b(n + 64, 0); if (fread(b.data(), 1, n, f) != n) exit(2); fclose(f); b.resize(n);
  return b;
}
template <class F> static void wave(F f) {
  std::vector<std::thread> th;
  for (int l = 0; l < 64; ++l) th.emplace_back([=]() { wave_emu::lane = l; f(l); });
  for (auto &x : th) x.join();
}
int main(int argc, char **argv) {
  if (argc < 4) return 2;
  std::vector<uint8_t> comp = slurp(argv[1]), want = slurp(argv[2]), szs = slurp(argv[3]);
  const size_t n = comp.size();
  comp.resize(n + 64, 0);
  const bool auto_sizes = szs.size() >= 4 && !memcmp(szs.data(), "auto", 4);
  std::vector<uint64_t> sizes; if (!auto_sizes) { szs.push_back(0); char *p = (char *)szs.data(); for (;;) { char *e; unsigned long long v = strtoull(p, &e, 10); if (e == p) break; sizes.push_back(v); p = e; } }
  std::vector<uint8_t> out(want.size() + 64, 0xEE);
  size_t pos = 0, k = 0; uint64_t out_off = 0, flow_windows = 0, fallbacks = 0, runs = 0, late = 0, steps_total = 0, fstat[8] = {0};
  while (pos + 18 <= n && comp[pos] == 0x1f && comp[pos + 1] == 0x8b) {
    if (!auto_sizes && k >= sizes.size()) { printf("more members than sizes\n"); return 1; }
    const uint64_t limit = auto_sizes ? want.size() - out_off : sizes[k];
    int flg = comp[pos + 3]; size_t q = pos + 10;
    if (flg & 4) q += 2 + comp[q] + 256 * comp[q + 1];
    if (flg & 8) { while (comp[q]) ++q; ++q; }
    if (flg & 16) { while (comp[q]) ++q; ++q; }
    if (flg & 2) q += 2;
    MemberDesc d{};
    d.in_off = q; d.out_off = out_off; d.out_limit = limit; d.expect_end = ~0ull; d.in_end = 0;
    d.hist = out_off < 32768 ? (u32)out_off : 32768u;
    u64 toff, doff; TokSink sk{nullptr, 0, nullptr, 0, 0, ~0u, 0, 0, false, false};
    tok_layout(0, d.out_limit, 0, toff, sk.col_cap, doff, sk.dir_cap);
    std::vector<u32> area((size_t)sk.col_cap * 64 + 64, 0xdeadbeefu);
    std::vector<DirEnt> dir((size_t)sk.dir_cap + 64);
    sk.area = area.data(); sk.dir = dir.data();
    MemberResult res{};
    HeaderLds &hdr = *(HeaderLds *)((u8 *)TL.p.inbuf + 1024);  // as in inflate_tokenize_kernel
    if (getenv("EMU_SIZING_SMALL")) {
      // a sizing run that keeps its tokens in an area far too small (a false candidate right behind the member's start
      // cuts it down to this on the device): it must give the tokens up, go on at full speed, and still report the
      // member's true size and end -- flagged MR_FAR so that the decode proper tokenizes it again
      sk.sizing = true; sk.col_cap = 8; sk.dir_cap = 63;
      d.out_limit = ~0ull;
      wave([&](int lane) { inflate_member<false, true>(TL.w, hdr, &TL.p, comp.data(), n, d, (u8 *)nullptr, sk, res, lane); });
      const uint64_t want_len = auto_sizes ? 0 : sizes[k];
      if (res.status != MS_OK || (!auto_sizes && res.out_len != want_len)) { printf("member %zu (sizing, small area): status %u out_len %llu (want %llu)\n", k, res.status, (unsigned long long)res.out_len, (unsigned long long)want_len); return 1; }
      if (res.blocks & MR_FAR) late++;
      if (res.fallbacks) fallbacks += res.fallbacks;
      out_off += res.out_len;
      pos = (size_t)res.end_pos + 8;
      ++k;
      continue;
    }
    wave([&](int lane) { inflate_member<false, true>(TL.w, hdr, &TL.p, comp.data(), n, d, (u8 *)nullptr, sk, res, lane); });
    if (res.status != MS_OK || (auto_sizes ? res.out_len > limit : res.out_len != limit)) { printf("member %zu: tokenizer status %u out_len %llu (want %llu) blocks %x\n", k, res.status, (unsigned long long)res.out_len, (unsigned long long)limit, res.blocks); return 1; }
    flow_windows += res.windows; fallbacks += res.fallbacks; runs += res.tok_words; steps_total += res.partial; for (int q = 0; q < 8; ++q) fstat[q] += res.cyc[q];
    if (res.blocks & MR_FAR) late++;  // reaches into earlier members' output (q8): the late kernel resolves it after them -- as here, in order
    wave([&](int lane) { u32 cyc[8] = {}; resolve_member(PL, comp.data(), area.data(), dir.data(), (u32)res.tok_words, out.data() + out_off, cyc, lane); });
    const uint64_t got_len = res.out_len;
    if (memcmp(out.data() + out_off, want.data() + out_off, got_len)) {
      size_t i = 0; while (out[out_off + i] == want[out_off + i]) ++i;
      printf("MISMATCH in member %zu at byte %zu of %llu (got %02x want %02x)\n", k, i, (unsigned long long)got_len, out[out_off + i], want[out_off + i]);
      return 1;
    }
    out_off += got_len;
    pos = (size_t)res.end_pos + 8;
    ++k;
  }
  if (getenv("EMU_SIZING_SMALL")) {
    if (out_off != want.size()) { printf("sized %llu of %zu bytes in %zu members\n", (unsigned long long)out_off, want.size(), k); return 1; }
    printf("inflate emu sizing ok: %zu members, %llu bytes; %llu gave their tokens up; hand-overs to the serial emitter %llu\n", k, (unsigned long long)out_off,
           (unsigned long long)late, (unsigned long long)fallbacks);
    return 0;
  }
  if (out_off != want.size()) { printf("decoded %llu of %zu bytes in %zu members\n", (unsigned long long)out_off, want.size(), k); return 1; }
  printf("inflate emu ok: %zu members, %llu bytes; flow epochs %llu, fallbacks to the serial emitter %llu, directory runs %llu, members reaching into earlier ones %llu\n", k, (unsigned long long)out_off,
         (unsigned long long)flow_windows, (unsigned long long)fallbacks, (unsigned long long)runs, (unsigned long long)late);
  if (steps_total) printf("decode steps of the wave (-DAHIP_PROFILE builds): %llu, %.1f per member\n", (unsigned long long)steps_total, (double)steps_total / (double)k);
#ifdef AHIP_RES_STATS
  { const char *nm[13] = {"looks", "chunks", "passes", "passes that split", "late fetches", "deferred matches", "flushes", "pending batches", "rounds", "wave copies", "reclassified passes", "deposited matches", "literals"};
    for (int q = 0; q < 13; ++q) printf("  res %-20s %10.1f per member\n", nm[q], (double)res_stats[q] / k); }
#endif
#ifdef AHIP_FLOW_STATS
  printf("flow stats per member: lane-steps %.0f (%.1f %% of 64 x steps), speculative %.0f, scheduling points %.1f, idle lanes after assignment %.1f per point, repairs %.1f\n",
         (double)fstat[0] / k, steps_total ? 100.0 * fstat[0] / (64.0 * steps_total) : 0.0, (double)fstat[1] / k, (double)fstat[2] / k, fstat[2] ? (double)fstat[3] / fstat[2] : 0.0, (double)fstat[4] / k);
#endif
  return k ? 0 : 7;
}
