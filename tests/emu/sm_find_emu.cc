// sm_find_emu.cc -- the block finder of the single-long-member path (sm_find_wave of sm_inflate.hpp: bit-parallel
// filter, Kraft filter, header filter; stored blocks on byte boundaries) on the CPU wave emulation, against the plain
// statement of what it looks for: the first bit position in [q0, q1) where BTYPE == 2, HLIT / HDIST <= 29, the code-length
// code is complete and sm_header_plausible() agrees -- or where a byte 0 / 1 is followed by LEN and ~LEN.  Test infrastructure only.
//
//   g++ -std=c++17 -O2 -pthread -o sm_find_emu tests/emu/sm_find_emu.cc
//   sm_find_emu <raw deflate stream> <ranges: q0 q1 pairs, decimal, whitespace separated>
#define AHIP_HOST_EMU 1
#include "../../archive_amd/csrc/sm_inflate.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
using namespace ahip;

static SmFindLds LDS;
static u8 TAB[128];

template <class F> static void wave(F f) {
  std::vector<std::thread> th;
  for (int l = 0; l < 64; ++l) th.emplace_back([=]() { wave_emu::lane = l; f(l); });
  for (auto &x : th) x.join();
}
static bool plain_test(const u8 *in, u64 n, u64 q) {
  // a stored block on a byte boundary: header byte 0 / 1, LEN, ~LEN
  if ((q & 7) == 0 && q + 40 <= n * 8) {
    const u8 *p = in + (q >> 3);
    if ((p[0] & 0xfe) == 0 && ((p[1] ^ p[3]) & 0xff) == 0xff && ((p[2] ^ p[4]) & 0xff) == 0xff) {
      const u64 q2 = q + 40 + 8ull * (p[1] | ((u32)p[2] << 8));  // ... followed by another stored block
      if (q2 + 40 <= n * 8) {
        const u8 *r = in + (q2 >> 3);
        if ((r[0] & 0xfe) == 0 && ((r[1] ^ r[3]) & 0xff) == 0xff && ((r[2] ^ r[4]) & 0xff) == 0xff) return true;
      }
    }
  }
  if (q + 29 > n * 8) return false;
  const u64 v = sm_bits64(in, n, q);
  if (((v >> 1) & 3) != 2) return false;
  if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) return false;
  const u32 ncl = ((u32)(v >> 13) & 15) + 4;
  const u64 w = sm_bits64(in, n, q + 17);
  u32 kraft = 0;
  for (u32 i = 0; i < ncl; ++i) { const u32 l = (u32)(w >> (3 * i)) & 7; if (l) kraft += 128u >> l; }
  if (kraft != 128) return false;
  return sm_header_plausible(in, n, q, TAB);
}
int main(int argc, char **argv) {
  if (argc < 3) return 2;
  FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
  fseek(f, 0, SEEK_END); const size_t n = ftell(f); fseek(f, 0, SEEK_SET);
  u8 *in = (u8 *)aligned_alloc(64, (n + 191) & ~(size_t)63);
  memset(in, 0, (n + 191) & ~(size_t)63);
  if (fread(in, 1, n, f) != n) return 2;
  fclose(f);
  int checked = 0, finds = 0;
  for (int a = 2; a + 1 < argc; a += 2) {
    const u64 q0 = strtoull(argv[a], nullptr, 10), q1 = strtoull(argv[a + 1], nullptr, 10);
    u64 want = ~0ull;
    for (u64 q = q0; q < q1; ++q) if (plain_test(in, n, q)) { want = q; break; }
    u64 got = 0;
    wave([&](int lane) { const u64 r = sm_find_wave(LDS, in, n, q0, q1, lane); if (lane == 0) got = r; });
    if (got != want) { printf("range [%llu, %llu): found %lld, the plain scan says %lld\n", (unsigned long long)q0, (unsigned long long)q1, (long long)got, (long long)want); return 1; }
    ++checked; finds += want != ~0ull;
  }
  printf("sm find emu ok: %d ranges, %d with a find\n", checked, finds);
  return 0;
}
