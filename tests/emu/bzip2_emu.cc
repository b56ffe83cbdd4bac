// bzip2_emu.cc -- phase 1 of the bzip2 block decoder executed on the CPU by 64 threads as the 64 lanes of a wave
// (tests/emu/wave_emu.hpp): bz_decode_block_wave (header, selectors, code lengths, tables, the windowed Huffman chain ->
// symbol stream), then the chunked move-to-front side glued the way archive_hip.hip launches it (bz_mtf_chunk_wave<false>
// per chunk, bz_mtf_scan_wave, bz_mtf_chunk_wave<true> per chunk).  The later phases are separate kernels on the device;
// here the host finishes the block the plain way (T^-1, pointer chase, run-length undo) so that the bytes can be
// compared.  Test infrastructure only.
//
//   g++ -std=c++17 -O2 -pthread -o bzip2_emu tests/emu/bzip2_emu.cc
//   bzip2_emu <file.bz2> <expected plain bytes>
#define AHIP_HOST_EMU 1
#include "../../archive_amd/csrc/bzip2_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
using namespace ahip;

static BzLds LDS;
static BzScanLds SLDS;

static std::vector<uint8_t> slurp(const char *path) {
  FILE *f = fopen(path, "rb"); if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> b(n); if (n && fread(b.data(), 1, n, f) != n) exit(2); fclose(f);
  return b;
}
template <class F> static void wave(F f) {
  std::vector<std::thread> th;
  for (int l = 0; l < 64; ++l) th.emplace_back([=]() { wave_emu::lane = l; f(l); });
  for (auto &x : th) x.join();
}
static uint64_t bits48(const std::vector<uint8_t> &d, uint64_t bit) {
  uint64_t v = 0;
  for (int k = 0; k < 48; ++k) { const uint64_t b = bit + k; v = (v << 1) | ((d[b >> 3] >> (7 - (b & 7))) & 1); }
  return v;
}
int main(int argc, char **argv) {
  if (argc < 3) return 2;
  std::vector<uint8_t> comp = slurp(argv[1]), want = slurp(argv[2]);
  const size_t n = comp.size();
  if (n < 14 || comp[0] != 'B' || comp[1] != 'Z' || comp[2] != 'h') { printf("not a bzip2 stream\n"); return 2; }
  const uint32_t level = comp[3] - '0';
  // device buffers are allocated: 4-byte aligned, readable a little past the end
  uint8_t *in = (uint8_t *)aligned_alloc(64, (n + 127) & ~(size_t)63);
  memset(in, 0, (n + 127) & ~(size_t)63);
  memcpy(in, comp.data(), n);
  std::vector<uint32_t> tt((size_t)100000 * level + 64, 0xdead0000u);
  std::vector<uint8_t> sel(BZ_MAX_SELECTORS + 64), out, list0(256), perms(BZ_CHUNKS * 256), lists(BZ_CHUNKS * 256);
  std::vector<uint16_t> syms(BZ_SYM_CAP);
  std::vector<BzChunk> chunks(BZ_CHUNKS);
  std::vector<uint32_t> offs(BZ_CHUNKS);
  uint64_t bit = 32;
  size_t blocks = 0;
  for (;;) {
    if (bit + 48 > n * 8) { printf("ran off the stream at bit %llu\n", (unsigned long long)bit); return 1; }
    const uint64_t magic = bits48(comp, bit);
    if (magic == 0x177245385090ull) break;
    if (magic != 0x314159265359ull) { printf("no block magic at bit %llu\n", (unsigned long long)bit); return 1; }
    BzResult R{};
    const BzCand c{bit, 0, 0};
    wave([&](int lane) { BzResult r; bz_decode_block_wave(LDS, in, n, c, syms.data(), list0.data(), sel.data(), r, (u32)lane); if (lane == 0) R = r; });
    const bool damaged_ok = argc > 3 && !strcmp(argv[3], "agree");  // damaged input: only ask that both forms of the pass agree
    if (R.status != BZ_ST_OK && !damaged_ok) { printf("block %zu: Huffman side status %u\n", blocks, R.status); return 1; }
    {
      // the position-parallel Huffman pass (bz_header's tables, bz_jump_tile for every tile, bz_walk_groups, bz_decode_group
      // for every group; these functions have no cross-lane operations and run here with one thread) must leave the
      // same symbol stream, end position and verdict
      static BzTables T;
      static BzTileLds TL;
      static BzWalkLds WL;
      BzResult H{};
      std::vector<uint8_t> sel2(BZ_MAX_SELECTORS + 64), list2(256);
      wave([&](int lane) { BzResult r; bz_decode_block_wave(LDS, in, n, c, (u16 *)nullptr, list2.data(), sel2.data(), r, (u32)lane, &T); if (lane == 0) H = r; });
      if (H.status != BZ_ST_OK) {  // header trouble: the serial wave must have stopped the same way, before any symbol
        if (damaged_ok && H.status == R.status && R.nsyms == 0) { printf("bzip2 emu agree: block %zu header status %u\n", blocks, H.status); return 0; }
        printf("block %zu: header pass status %u, serial wave %u\n", blocks, H.status, R.status); return 1;
      }
      if (T.sym_bit != H.end_bit) { printf("block %zu: header pass end\n", blocks); return 1; }
      const uint64_t lim = n * 8, bit0 = bit;  // (one block at a time here: its bits end with the input)
      const uint64_t tstride = lim - bit0 + 64;
      std::vector<uint16_t> j50((size_t)tstride * 6, 0xabcd), syms2(BZ_SYM_CAP, 0xeeee);
      for (uint64_t tb = T.sym_bit; tb < lim; tb += BZ_TW) bz_jump_tile(TL, in, n, &T, tb, lim, j50.data() + (tb - bit0), tstride, 0, 1);
      std::vector<uint32_t> gstart(BZ_MAX_SELECTORS);
      uint32_t found = 0, marked = 0;
      bz_walk_groups(WL, &T, sel2.data(), lim, j50.data(), bit0, tstride, gstart.data(), found, marked, 0, 1);
      BzGroupEnd end{};
      if (marked == 0) { end.status = BZ_ST_NEG; end.nsyms = R.nsyms; end.end_bit = R.end_bit; }  // out of selectors: _getMtfVal's -1 (bz_block_exact goes on from there like the reference)
      else if (marked == 2 && damaged_ok) { printf("bzip2 emu agree: block %zu handed to the serial wave by the walk (it ran into the end of the block's bits)\n", blocks); return 0; }
      else if (marked != 1 || found == 0) { printf("block %zu: walk found %u groups, marked %u\n", blocks, found, marked); return 1; }
      else for (uint32_t g = 0; g < found; ++g) bz_decode_group(T, T.eob, in, n, T.sym_bit, g, gstart[g], sel2[g], g + 1 == found, syms2.data(), end);
      if (end.status == BZ_ST_HUFF_SERIAL) { printf("bzip2 emu agree: block %zu handed to the serial wave\n", blocks); return 0; }
      if (damaged_ok && R.status != BZ_ST_OK) {
        // an error: the verdict must be the same; the symbols in front of it too (the list side reads them and may fail first)
        const bool same = end.status == R.status && (marked == 0 || (end.nsyms == R.nsyms && !memcmp(syms2.data(), syms.data(), (size_t)R.nsyms * 2)));
        printf(same ? "bzip2 emu agree: block %zu status %u after %u symbols\n" : "block %zu: verdicts differ (%u)\n", blocks, R.status, R.nsyms);
        if (!same) printf("  position-parallel: status %u nsyms %u; serial wave: status %u nsyms %u\n", end.status, end.nsyms, R.status, R.nsyms);
        return same ? 0 : 1;
      }
      if (end.status != R.status || end.nsyms != R.nsyms || end.end_bit != R.end_bit || memcmp(syms2.data(), syms.data(), (size_t)R.nsyms * 2) ||
          memcmp(sel2.data(), sel.data(), T.nsel) || memcmp(list2.data(), list0.data(), 256)) {
        size_t i = 0; while (i < R.nsyms && syms2[i] == syms[i]) ++i;
        printf("block %zu: position-parallel pass differs: status %u/%u nsyms %u/%u end %llu/%llu first differing symbol %zu\n", blocks, end.status, R.status,
               end.nsyms, R.nsyms, (unsigned long long)end.end_bit, (unsigned long long)R.end_bit, i);
        return 1;
      }
    }
    const uint32_t nmax = 100000u * level;
    // the list side the way archive_hip.hip launches it: a part per lane (bz_mtf_lanes<false>), the scan over the chunks, the parts
    // again with the bytes (bz_mtf_lanes<true>)
    static BzLaneLds LL;
    std::vector<uint8_t> pperms((size_t)BZ_PARTS * 256);
    std::vector<uint32_t> pcounts(BZ_PARTS);
    std::vector<uint8_t> b8(tt.size(), 0xdd);
    const BzResult R_huff = R;
    for (uint32_t k = 0; k < BZ_CHUNKS; ++k)
      wave([&](int lane) { BzChunk r; bz_mtf_lanes_wave<false>(LL, syms.data(), R.nsyms, k, nmax, nullptr, 0, b8.data(), pperms.data() + (size_t)k * 64 * 256, pcounts.data() + k * 64, perms.data() + k * 256, r, (u32)lane); if (lane == 0) chunks[k] = r; });
    wave([&](int lane) { BzResult r = R; bz_mtf_scan_wave(SLDS, r, nmax, chunks.data(), perms.data(), list0.data(), lists.data(), offs.data(), (u32)lane); wave_emu::barrier(); if (lane == 0) R = r; });
    {
      // ... and the wave-per-symbol form of the same passes (bz_mtf_chunk_wave; its chunks are cut elsewhere): same verdict, same bytes
      std::vector<uint8_t> perms_w(BZ_CHUNKS * 256), lists_w(BZ_CHUNKS * 256);
      std::vector<uint32_t> tt_w(tt.size(), 0xdead0000u), offs_w(BZ_CHUNKS);
      std::vector<BzChunk> chunks_w(BZ_CHUNKS);
      BzResult W = R_huff;
      for (uint32_t k = 0; k < BZ_CHUNKS; ++k)
        wave([&](int lane) { BzChunk r; bz_mtf_chunk_wave<false>(syms.data(), W.nsyms, k, nmax, nullptr, 0, tt_w.data(), perms_w.data() + k * 256, r, (u32)lane); if (lane == 0) chunks_w[k] = r; });
      wave([&](int lane) { BzResult r = W; bz_mtf_scan_wave(SLDS, r, nmax, chunks_w.data(), perms_w.data(), list0.data(), lists_w.data(), offs_w.data(), (u32)lane); wave_emu::barrier(); if (lane == 0) W = r; });
      if (W.status != R.status || (R.status == BZ_ST_OK && W.nblock != R.nblock)) { printf("block %zu: the two forms of the list pass differ: status %u / %u, size %u / %u\n", blocks, R.status, W.status, R.nblock, W.nblock); return 1; }
      if (R.status == BZ_ST_OK) {
        for (uint32_t k = 0; k < BZ_CHUNKS; ++k) {
          wave([&](int lane) { BzChunk r; bz_mtf_lanes_wave<true>(LL, syms.data(), R.nsyms, k, nmax, lists.data() + k * 256, offs[k], b8.data(), pperms.data() + (size_t)k * 64 * 256, pcounts.data() + k * 64, nullptr, r, (u32)lane); });
          wave([&](int lane) { BzChunk r; bz_mtf_chunk_wave<true>(syms.data(), W.nsyms, k, nmax, lists_w.data() + k * 256, offs_w[k], tt_w.data(), perms_w.data() + k * 256, r, (u32)lane); });
        }
        for (uint32_t i = 0; i < R.nblock; ++i)
          if (tt_w[i] != b8[i]) { printf("block %zu: the two forms of the list pass wrote different bytes (at %u)\n", blocks, i); return 1; }
        for (uint32_t i = 0; i < R.nblock; ++i) tt[i] = b8[i];  // (bz_tinv_scatter's first step)
      }
    }
    if (R.status != BZ_ST_OK) {
      if (damaged_ok) { printf("bzip2 emu agree: block %zu, the list side says status %u\n", blocks, R.status); return 0; }
      printf("block %zu: status %u\n", blocks, R.status); return 1;
    }
    const uint32_t nb = R.nblock;
    // T^-1 and the walk, as bzip2_decoder.dart:406-439 / :610-727 do them
    uint32_t cf[257] = {0};
    for (uint32_t i = 0; i < nb; ++i) { if (tt[i] > 255) { printf("block %zu: tt[%u] = %x is no byte\n", blocks, i, tt[i]); return 1; } cf[(tt[i] & 0xff) + 1]++; }
    for (int i = 1; i <= 256; ++i) cf[i] += cf[i - 1];
    for (uint32_t i = 0; i < nb; ++i) { const uint32_t ch = tt[i] & 0xff; tt[cf[ch]++] |= i << 8; }
    uint32_t tpos = tt[R.pad_orig_ptr] >> 8, run = 0, prev = 256;
    for (uint32_t k = 0; k < nb; ++k) {
      tpos = tt[tpos];
      const uint8_t ch = tpos & 0xff;
      tpos >>= 8;
      if (run == 4) { for (uint32_t r = 0; r < ch; ++r) out.push_back((uint8_t)prev); run = 0; prev = 256; continue; }
      out.push_back(ch);
      if (ch == prev) run++; else { run = 1; prev = ch; }
    }
    bit = R.end_bit;
    ++blocks;
  }
  if (argc > 3 && !strcmp(argv[3], "agree")) { printf("bzip2 emu agree: %zu blocks decoded by both forms of the pass alike (%zu bytes)\n", blocks, out.size()); return 0; }
  if (out.size() != want.size() || memcmp(out.data(), want.data(), out.size())) {
    size_t i = 0; while (i < out.size() && i < want.size() && out[i] == want[i]) ++i;
    printf("MISMATCH at byte %zu (sizes %zu / %zu)\n", i, out.size(), want.size());
    return 1;
  }
  printf("bzip2 emu ok: %zu blocks, %zu bytes\n", blocks, out.size());
  return 0;
}
