// raw_emu.cc -- raw DEFLATE streams, valid or damaged, through the device code on the CPU (tests/emu/wave_emu.hpp) the way
// the kernels of archive_hip.hip take one member: inflate_tokenize_kernel's body, then inflate_late_kernel's for what it
// leaves behind (over-subscribed codes -> the reference's own table; a full token store -> the byte-writing serial decoder;
// a reach in front of the output), else inflate_resolve_kernel's.  Prints, per stream, the ABI status
// (member_status_to_abi), the produced length, the stream position and an FNV-1a hash of the output for the Python side to
// compare with the oracle.  Test infrastructure only.
//
//   g++ -std=c++17 -O2 -pthread -o raw_emu tests/emu/raw_emu.cc
//   raw_emu <file: u32 count, then per stream u32 length + bytes> <output file for the decoded bytes, concatenated>
#define AHIP_HOST_EMU 1
#include "../../archive_amd/csrc/inflate_par.hpp"
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
using namespace ahip;

static struct TokKernelLds { WaveLds w; TokLds p; } TL;
static ResLds PL;
static WaveLds LL;   // the late kernel's own tables / header scratch
static HeaderLds LH;

template <class F> static void wave(F f) {
  std::vector<std::thread> th;
  for (int l = 0; l < 64; ++l) th.emplace_back([=]() { wave_emu::lane = l; f(l); });
  for (auto &x : th) x.join();
}
static int abi(u32 ms) { return ms == MS_OK || ms == MS_EOS ? 0 : ms == MS_FALSE ? 1 : (ms == MS_RANGE || ms == MS_FARREF) ? 2 : ms == MS_HANG ? 3 : -100 - (int)ms; }

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
  FILE *fo = fopen(argv[2], "wb"); if (!fo) return 2;
  uint32_t count = 0; if (fread(&count, 4, 1, f) != 1) return 2;
  const u64 CAP = 1u << 20;
  std::vector<u32> exact(2 * 32768);
  for (uint32_t i = 0; i < count; ++i) {
    uint32_t len = 0; if (fread(&len, 4, 1, f) != 1) return 2;
    std::vector<uint8_t> in(len + 64, 0);
    if (len && fread(in.data(), 1, len, f) != len) return 2;
    MemberDesc d{};
    d.in_off = 0; d.out_off = 0; d.out_limit = CAP; d.expect_end = ~0ull; d.in_end = 0; d.hist = 0;
    u64 toff, doff; TokSink sk{nullptr, 0, nullptr, 0, 0, ~0u, 0, 0, false, false};
    tok_layout(0, d.out_limit, 0, toff, sk.col_cap, doff, sk.dir_cap);
    std::vector<u32> area((size_t)sk.col_cap * 64 + 64);
    std::vector<DirEnt> dir((size_t)sk.dir_cap + 64);
    std::vector<uint8_t> out(CAP + 64, 0);
    sk.area = area.data(); sk.dir = dir.data();
    MemberResult res{};
    HeaderLds &hdr = *(HeaderLds *)((u8 *)TL.p.inbuf + 1024);
    wave([&](int lane) { inflate_member<false, true>(TL.w, hdr, &TL.p, in.data(), len, d, (u8 *)nullptr, sk, res, lane); });
    const char *route = "resolve";
    if (res.status == MS_TOKFULL || res.status == MS_OVERSUB) {
      route = res.status == MS_OVERSUB ? "late:exact" : "late:serial";
      wave([&](int lane) { inflate_member<true, false>(LL, LH, nullptr, in.data(), len, d, out.data(), TokSink{nullptr, 0, nullptr, 0, 0, ~0u, 0, 0, false, false}, res, lane, nullptr, exact.data()); });
    } else {
      if (res.blocks & MR_FAR) route = "late:far";
      wave([&](int lane) { u32 cyc[8] = {}; resolve_member(PL, in.data(), area.data(), dir.data(), (u32)res.tok_words, out.data(), cyc, lane); });
    }
    const u64 n = res.out_len <= CAP ? res.out_len : 0;
    uint64_t h = 1469598103934665603ull;
    for (u64 k = 0; k < n; ++k) { h ^= out[k]; h *= 1099511628211ull; }
    printf("%u %d %llu %llu %016llx %s\n", i, abi(res.status), (unsigned long long)res.out_len, (unsigned long long)res.end_pos, (unsigned long long)h, route);
    uint64_t n64 = n; fwrite(&n64, 8, 1, fo); if (n) fwrite(out.data(), 1, n, fo);
  }
  fclose(fo);
  return 0;
}
