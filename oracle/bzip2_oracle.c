/*
 * oracle/bzip2_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's BZip2Decoder
 * (/root/reference/lib/src/codecs/bzip2_decoder.dart, bzip2/bz2_bit_reader.dart, bzip2/bzip2.dart --
 * a Dart transliteration of libbzip2's decompress.c).  Checks the HIP bzip2 path; never linked into
 * or called from the product.
 *
 * Parity pinning: the reference's fixture test/_data/bzip2/test.bz2 (813 -> 1773 bytes,
 * test/bzip2_test.dart:8-15) and test/_data/test2.tar.bz2 -> test2.tar (tests/golden/), plus CPython's
 * bz2 (libbzip2) on valid streams of every block size (tests/test_bzip2_oracle.py).
 *
 * Restated (bzip2_decoder.dart line numbers):
 *   :20-88    decodeStream (ONE stream: returns at the first end-of-stream block)  -> orc_bzip2_decode
 *   :90-111   _readBlockType                                                         -> read_block_type
 *   :113-730  _readCompressed: header :114-246, MTF/RUNA/RUNB -> tt :267-388, cftab :406-432,
 *             T^-1 :435-439, inverse BWT + un-RLE + CRC :610-727                     -> read_compressed
 *   :732-772  _getMtfVal     :774-813 _hbCreateDecodeTables     :815-823 _makeMaps
 *   bzip2/bz2_bit_reader.dart:12-44 readBits (MSB first); bzip2/bzip2.dart:11-18 CRC (MSB-first CRC-32)
 * The randomised-block branch (:441-471, :489-608) is obsolete (bzip2 >= 0.9.5 never sets it) and
 * mis-steps rNToGo in the reference; it is NOT restated: such a block returns ORC_FALSE here.
 *
 * Status codes as in inflate_oracle.c: 0 true, 1 false (output so far kept), 2 RangeError
 * (bit reader ran past the end), -1 output cap.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

enum { ORC_OK = 0, ORC_FALSE = 1, ORC_RANGE = 2, ORC_CAP = -1 };
enum { BZ_N_GROUPS = 6, BZ_G_SIZE = 50, BZ_MAX_ALPHA = 258, BZ_MAX_CODE_LEN = 23, BZ_MAX_SELECTORS = 18002,
       MTFA_SIZE = 4096, MTFL_SIZE = 16, BZ_RUNA = 0, BZ_RUNB = 1 };

typedef struct {
  const uint8_t *p; size_t n, pos;
  uint32_t buf; int bitpos; /* bits left in buf */
  int fault;
} bitreader_t;

static uint32_t br_byte_raw(bitreader_t *b) {
  if (b->pos >= b->n) { b->fault = 1; b->pos++; return 0; }
  return b->p[b->pos++];
}
/* Bz2BitReader.readBits */
static uint32_t br_bits(bitreader_t *b, int nbits) {
  static const uint32_t mask[9] = {0, 1, 3, 7, 15, 31, 63, 127, 255};
  if (nbits == 0) return 0;
  if (b->bitpos == 0) { b->bitpos = 8; b->buf = br_byte_raw(b); }
  uint32_t value = 0;
  while (nbits > b->bitpos) {
    value = (value << b->bitpos) + (b->buf & mask[b->bitpos]);
    nbits -= b->bitpos;
    b->bitpos = 8;
    b->buf = br_byte_raw(b);
  }
  if (nbits > 0) {
    if (b->bitpos == 0) { b->bitpos = 8; b->buf = br_byte_raw(b); }
    value = (value << nbits) + ((b->buf >> (b->bitpos - nbits)) & mask[nbits]);
    b->bitpos -= nbits;
  }
  return value;
}

static uint32_t g_crc_table[256];
static int g_crc_ready = 0;
static void crc_init(void) {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i << 24;
    for (int k = 0; k < 8; ++k) c = (c & 0x80000000u) ? ((c << 1) ^ 0x04c11db7u) : (c << 1);
    g_crc_table[i] = c;
  }
  g_crc_ready = 1;
}
static uint32_t crc_update(uint32_t value, uint32_t crc) { return (crc << 8) ^ g_crc_table[((crc >> 24) & 0xff) ^ (value & 0xff)]; }

typedef struct {
  bitreader_t br;
  uint8_t *out; size_t out_len, out_cap; int cap_fault;
  int block_size100k;
  uint32_t *tt;
  uint8_t in_use[256], seq_to_unseq[256];
  int num_in_use;
  uint8_t selector[BZ_MAX_SELECTORS], selector_mtf[BZ_MAX_SELECTORS];
  int num_selectors;
  uint8_t len[BZ_N_GROUPS][BZ_MAX_ALPHA];
  int32_t limit[BZ_N_GROUPS][BZ_MAX_ALPHA], base[BZ_N_GROUPS][BZ_MAX_ALPHA], perm[BZ_N_GROUPS][BZ_MAX_ALPHA];
  int32_t min_lens[BZ_N_GROUPS];
  int32_t unzftab[256], cftab[257];
  uint8_t mtfa[MTFA_SIZE];
  int32_t mtfbase[256 / MTFL_SIZE];
  int group_pos, group_no, g_sel, g_minlen;
} bz_t;

static void out_byte(bz_t *s, int c) {
  if (s->out_len >= s->out_cap) { s->cap_fault = 1; return; }
  s->out[s->out_len++] = (uint8_t)c;
}

static void hb_create_decode_tables(int32_t *limit, int32_t *base, int32_t *perm, const uint8_t *length, int min_len,
                                    int max_len, int alpha_size) {
  int pp = 0;
  for (int i = min_len; i <= max_len; i++)
    for (int j = 0; j < alpha_size; j++)
      if (length[j] == i) perm[pp++] = j;
  for (int i = 0; i < BZ_MAX_CODE_LEN; i++) base[i] = 0;
  for (int i = 0; i < alpha_size; i++) base[length[i] + 1]++;
  for (int i = 1; i < BZ_MAX_CODE_LEN; i++) base[i] += base[i - 1];
  for (int i = 0; i < BZ_MAX_CODE_LEN; i++) limit[i] = 0;
  int vec = 0;
  for (int i = min_len; i <= max_len; i++) {
    vec += base[i + 1] - base[i];
    limit[i] = vec - 1;
    vec <<= 1;
  }
  for (int i = min_len + 1; i <= max_len; i++) base[i] = ((limit[i - 1] + 1) << 1) - base[i];
}

static int get_mtf_val(bz_t *s) {
  if (s->group_pos == 0) {
    s->group_no++;
    if (s->group_no >= s->num_selectors) return -1;
    s->group_pos = BZ_G_SIZE;
    s->g_sel = s->selector[s->group_no];
    s->g_minlen = s->min_lens[s->g_sel];
  }
  s->group_pos--;
  int zn = s->g_minlen;
  int32_t zvec = (int32_t)br_bits(&s->br, zn);
  for (;;) {
    if (zn > 20) return -1;
    if (zvec <= s->limit[s->g_sel][zn]) break;
    zn++;
    zvec = (zvec << 1) | (int32_t)br_bits(&s->br, 1);
  }
  int32_t idx = zvec - s->base[s->g_sel][zn];
  if (idx < 0 || idx >= BZ_MAX_ALPHA) return -1;
  return s->perm[s->g_sel][idx];
}

/* -1 = data error (reference returns -1); otherwise the block CRC (not finalised) via *crc_out */
static int read_compressed(bz_t *s, uint32_t *crc_out) {
  bitreader_t *br = &s->br;
  /* The reference allocates these afresh -- zero-filled -- for EVERY block (_selectorMtf/_selector :155-156, _len :193-196,
   * _limit/_base/_perm :221-229, _seqToUnseq :817; only _tt lives as long as the stream, :42): a damaged block whose code
   * indexes _perm beyond the symbols its tables hold reads 0 there, never what the block before it left behind. */
  memset(s->selector, 0, sizeof s->selector); memset(s->selector_mtf, 0, sizeof s->selector_mtf);
  memset(s->len, 0, sizeof s->len); memset(s->limit, 0, sizeof s->limit); memset(s->base, 0, sizeof s->base);
  memset(s->perm, 0, sizeof s->perm); memset(s->min_lens, 0, sizeof s->min_lens); memset(s->seq_to_unseq, 0, sizeof s->seq_to_unseq);
  int randomized = (int)br_bits(br, 1);
  int32_t orig_ptr = (int32_t)br_bits(br, 8);
  orig_ptr = (orig_ptr << 8) | (int32_t)br_bits(br, 8);
  orig_ptr = (orig_ptr << 8) | (int32_t)br_bits(br, 8);
  uint8_t in_use16[16];
  for (int i = 0; i < 16; ++i) in_use16[i] = (uint8_t)br_bits(br, 1);
  memset(s->in_use, 0, 256);
  for (int i = 0, k = 0; i < 16; ++i, k += 16)
    if (in_use16[i])
      for (int j = 0; j < 16; ++j) s->in_use[k + j] = (uint8_t)br_bits(br, 1);
  s->num_in_use = 0;
  for (int i = 0; i < 256; ++i)
    if (s->in_use[i]) s->seq_to_unseq[s->num_in_use++] = (uint8_t)i;
  if (s->num_in_use == 0) return -1;
  int alpha_size = s->num_in_use + 2;
  int num_groups = (int)br_bits(br, 3);
  if (num_groups < 2 || num_groups > 6) return -1;
  s->num_selectors = (int)br_bits(br, 15);
  if (s->num_selectors < 1) return -1;
  for (int i = 0; i < s->num_selectors; ++i) {
    int j = 0;
    for (;;) {
      if (br_bits(br, 1) == 0) break;
      j++;
      if (j >= num_groups) return -1;
    }
    if (i < BZ_MAX_SELECTORS) s->selector_mtf[i] = (uint8_t)j;
    else { br->fault = 1; return -1; } /* Dart: index past the Uint8List -> RangeError (reported like the reader's own); libbzip2 >= 1.0.8 discards */
    if (br->fault) return -1;
  }
  uint8_t pos[BZ_N_GROUPS];
  for (int i = 0; i < num_groups; ++i) pos[i] = (uint8_t)i;
  for (int i = 0; i < s->num_selectors; ++i) {
    int v = s->selector_mtf[i];
    uint8_t tmp = pos[v];
    while (v > 0) { pos[v] = pos[v - 1]; v--; }
    pos[0] = tmp;
    s->selector[i] = tmp;
  }
  for (int t = 0; t < num_groups; ++t) {
    int c = (int)br_bits(br, 5);
    for (int i = 0; i < alpha_size; ++i) {
      for (;;) {
        if (c < 1 || c > 20) return -1;
        if (br_bits(br, 1) == 0) break;
        if (br_bits(br, 1) == 0) c++; else c--;
        if (br->fault) return -1;
      }
      s->len[t][i] = (uint8_t)c;
    }
  }
  for (int t = 0; t < num_groups; t++) {
    int min_len = 32, max_len = 0;
    for (int i = 0; i < alpha_size; ++i) {
      if (s->len[t][i] > max_len) max_len = s->len[t][i];
      if (s->len[t][i] < min_len) min_len = s->len[t][i];
    }
    hb_create_decode_tables(s->limit[t], s->base[t], s->perm[t], s->len[t], min_len, max_len, alpha_size);
    s->min_lens[t] = min_len;
  }
  int eob = s->num_in_use + 1;
  int nblock_max = 100000 * s->block_size100k;
  memset(s->unzftab, 0, sizeof s->unzftab);
  memset(s->mtfa, 0, sizeof s->mtfa); /* _mtfa = Uint8List(mtfaSize), :255 */
  {
    int kk = MTFA_SIZE - 1;
    for (int ii = 256 / MTFL_SIZE - 1; ii >= 0; ii--) {
      for (int jj = MTFL_SIZE - 1; jj >= 0; jj--) { s->mtfa[kk] = (uint8_t)(ii * MTFL_SIZE + jj); kk--; }
      s->mtfbase[ii] = kk + 1;
    }
  }
  int nblock = 0;
  s->group_pos = 0;
  s->group_no = -1;
  int next_sym = get_mtf_val(s);
  if (next_sym < 0) return -1;
  int uc = 0;
  for (;;) {
    if (br->fault) return -1;
    if (next_sym == eob) break;
    if (next_sym == BZ_RUNA || next_sym == BZ_RUNB) {
      int es = -1, N = 1;
      do {
        if (N >= 2 * 1024 * 1024) return -1;
        if (next_sym == BZ_RUNA) es += N; else es += 2 * N;
        N *= 2;
        next_sym = get_mtf_val(s);
      } while (next_sym == BZ_RUNA || next_sym == BZ_RUNB);
      es++;
      uc = s->seq_to_unseq[s->mtfa[s->mtfbase[0]]];
      s->unzftab[uc] += es;
      while (es > 0) {
        if (nblock >= nblock_max) return -1;
        s->tt[nblock++] = (uint32_t)uc;
        es--;
      }
      continue;
    }
    if (nblock >= nblock_max) return -1;
    {
      /* next_sym may be -1 here: _getMtfVal's failures (out of selectors, a code longer than 20 bits, an index outside
       * the alphabet: :735, :753, :766) are NOT checked by the loop (:304, :385 -- only the very first call is, :271).
       * The reference goes on with -1 as a symbol: nn = -2 takes the short-list branch, reads _mtfa[_mtfbase[0] - 2]
       * (whatever that cell of the freshly allocated array holds: zero unless this block's list has been there), shifts
       * nothing, makes that byte the front of the list, stores it, and decodes on from wherever the bit reader stands --
       * until an end-of-block symbol, nblockMAX (`false`), the end of the input or a negative index (RangeError). */
      int nn = next_sym - 1;
      if (nn < MTFL_SIZE) {
        int pp = s->mtfbase[0];
        if (pp + nn < 0) { br->fault = 1; return -1; } /* Uint8List[-1]: RangeError */
        uc = s->mtfa[pp + nn];
        while (nn > 0) { s->mtfa[pp + nn] = s->mtfa[pp + nn - 1]; nn--; }
        s->mtfa[pp] = (uint8_t)uc;
      } else {
        int lno = nn / MTFL_SIZE, off = nn % MTFL_SIZE;
        int pp = s->mtfbase[lno] + off;
        uc = s->mtfa[pp];
        while (pp > s->mtfbase[lno]) { s->mtfa[pp] = s->mtfa[pp - 1]; pp--; }
        s->mtfbase[lno]++;
        while (lno > 0) {
          s->mtfbase[lno]--;
          s->mtfa[s->mtfbase[lno]] = s->mtfa[s->mtfbase[lno - 1] + MTFL_SIZE - 1];
          lno--;
        }
        s->mtfbase[0]--;
        s->mtfa[s->mtfbase[0]] = (uint8_t)uc;
        if (s->mtfbase[0] == 0) {
          int kk = MTFA_SIZE - 1;
          for (int ii = 256 / MTFL_SIZE - 1; ii >= 0; ii--) {
            for (int jj = MTFL_SIZE - 1; jj >= 0; jj--) { s->mtfa[kk] = s->mtfa[s->mtfbase[ii] + jj]; kk--; }
            s->mtfbase[ii] = kk + 1;
          }
        }
      }
    }
    s->unzftab[s->seq_to_unseq[uc]]++;
    s->tt[nblock++] = s->seq_to_unseq[uc];
    next_sym = get_mtf_val(s); /* (not checked: see above) */
  }
  if (orig_ptr < 0 || orig_ptr >= nblock) return -1;
  for (int i = 0; i <= 255; i++) if (s->unzftab[i] < 0 || s->unzftab[i] > nblock) return -1;
  s->cftab[0] = 0;
  for (int i = 1; i <= 256; i++) s->cftab[i] = s->unzftab[i - 1];
  for (int i = 1; i <= 256; i++) s->cftab[i] += s->cftab[i - 1];
  for (int i = 0; i <= 256; i++) if (s->cftab[i] < 0 || s->cftab[i] > nblock) return -1;
  for (int i = 1; i <= 256; i++) if (s->cftab[i - 1] > s->cftab[i]) return -1;
  for (int i = 0; i < nblock; i++) {
    uc = (int)(s->tt[i] & 0xff);
    s->tt[s->cftab[uc]] |= ((uint32_t)i << 8);
    s->cftab[uc]++;
  }
  uint32_t crc = 0xffffffffu;
  if (randomized) return -1; /* not restated (see header) */
  uint32_t t_pos = s->tt[orig_ptr] >> 8;
  int n_used = 0;
  if (t_pos >= (uint32_t)nblock_max) { *crc_out = crc; return 0; }
  t_pos = s->tt[t_pos];
  int k0 = (int)(t_pos & 0xff);
  t_pos >>= 8;
  n_used++;
  int out_len = 0, out_ch = 0, save_nblock_pp = nblock + 1, c_k0 = k0, k1;
  for (;;) {
    if (out_len > 0) {
      for (;;) {
        if (out_len == 1) break;
        out_byte(s, out_ch);
        crc = crc_update((uint32_t)out_ch, crc);
        out_len--;
      }
      out_byte(s, out_ch);
      crc = crc_update((uint32_t)out_ch, crc);
    }
    if (s->cap_fault) return -1;
    if (n_used > save_nblock_pp) return -1;
    if (n_used == save_nblock_pp) { *crc_out = crc; return 0; }
    out_ch = c_k0;
#define BZ_NEXT(var)                                                 \
  do {                                                               \
    if (t_pos >= (uint32_t)nblock_max) return -1;                    \
    t_pos = s->tt[t_pos];                                            \
    (var) = (int)(t_pos & 0xff);                                     \
    t_pos >>= 8;                                                     \
    n_used++;                                                        \
  } while (0)
    BZ_NEXT(k1);
    if (k1 != c_k0) { c_k0 = k1; out_byte(s, out_ch); crc = crc_update((uint32_t)out_ch, crc); out_len = 0; continue; }
    if (n_used == save_nblock_pp) { out_byte(s, out_ch); crc = crc_update((uint32_t)out_ch, crc); out_len = 0; continue; }
    out_len = 2;
    BZ_NEXT(k1);
    if (n_used == save_nblock_pp) continue;
    if (k1 != c_k0) { c_k0 = k1; continue; }
    out_len = 3;
    BZ_NEXT(k1);
    if (n_used == save_nblock_pp) continue;
    if (k1 != c_k0) { c_k0 = k1; continue; }
    BZ_NEXT(k1);
    out_len = k1 + 4;
    BZ_NEXT(c_k0);
  }
}

static int read_block_type(bz_t *s) {
  static const uint8_t cm[6] = {0x31, 0x41, 0x59, 0x26, 0x53, 0x59}, em[6] = {0x17, 0x72, 0x45, 0x38, 0x50, 0x90};
  int eos = 1, compressed = 1;
  for (int i = 0; i < 6; ++i) {
    uint32_t b = br_bits(&s->br, 8);
    if (b != cm[i]) compressed = 0;
    if (b != em[i]) eos = 0;
    if (!eos && !compressed) return -1;
  }
  return compressed ? 0 : 2;
}

/* ONE candidate block of a stream on its own -- what the chain test (tests/emu/bzip2_chain_emu.cc) feeds the product's
 * host chain with instead of the GPU's per-block verdicts: the bit reader is put at `bit` (any bit position), then
 * _readBlockType, the stored CRC and, for a compressed block, _readCompressed run exactly as decodeStream runs them
 * (bzip2_decoder.dart:47-58, :70-75).  kind: 0 compressed, 2 end of stream, -1 no magic.  Returns ORC_OK / ORC_FALSE
 * (bytes written before the failure are in out[0, *out_len)) / ORC_RANGE / ORC_CAP; 17 = the randomised flag is set (the
 * product reports that as unsupported before anything else of the block). */
int orc_bzip2_block(const uint8_t *in, size_t n, uint64_t bit, int level, uint8_t *out, size_t cap, uint64_t *end_bit,
                    size_t *out_len, uint32_t *crc_out, uint32_t *stored_out, int *kind_out) {
  if (!g_crc_ready) crc_init();
  bz_t *s = (bz_t *)calloc(1, sizeof(bz_t));
  if (!s) return ORC_CAP;
  int st = ORC_OK;
  s->br.p = in; s->br.n = n; s->out = out; s->out_cap = cap;
  s->br.pos = (size_t)(bit >> 3);
  if (bit & 7) { s->br.buf = br_byte_raw(&s->br); s->br.bitpos = 8 - (int)(bit & 7); }
  s->block_size100k = level;
  s->tt = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(level * 100000 + 1));
  uint32_t crc = 0, stored = 0;
  int type = read_block_type(s);
  *kind_out = type;
  if (s->br.fault) { st = ORC_RANGE; goto done; }
  if (type < 0) { st = ORC_FALSE; goto done; }
  for (int k = 0; k < 4; ++k) stored = (stored << 8) | br_bits(&s->br, 8);
  if (s->br.fault) { st = ORC_RANGE; goto done; }
  if (type == 0) {
    bitreader_t look = s->br;
    if (br_bits(&look, 1) == 1 && !look.fault) { st = 17; goto done; }
    int r = read_compressed(s, &crc);
    if (s->br.fault) st = ORC_RANGE;
    else if (s->cap_fault) st = ORC_CAP;
    else if (r < 0) st = ORC_FALSE;
    crc ^= 0xffffffffu;
  }
done:
  *end_bit = (uint64_t)s->br.pos * 8 - (uint64_t)s->br.bitpos;
  *out_len = s->out_len; *crc_out = crc; *stored_out = stored;
  free(s->tt);
  free(s);
  return st;
}

/* where decodeStream left its InputStream (the bytes its Bz2BitReader had pulled, bz2_bit_reader.dart:12-44) in the last
 * orc_bzip2_decode() of this thread */
static __thread size_t g_bz_last_pos;
size_t orc_bzip2_last_position(void) { return g_bz_last_pos; }

/* BZip2Decoder().decodeBytes(data, verify) */
int orc_bzip2_decode(const uint8_t *in, size_t n, int verify, uint8_t *out, size_t cap, size_t *out_len) {
  if (!g_crc_ready) crc_init();
  bz_t *s = (bz_t *)calloc(1, sizeof(bz_t));
  if (!s) return ORC_CAP;
  s->br.p = in; s->br.n = n; s->out = out; s->out_cap = cap;
  int st = ORC_OK;
  uint32_t combined = 0;
  /* `a != .. || b != .. || c != ..` short-circuits; a read past the end throws before the compare */
  for (int k = 0; k < 3; ++k) {
    uint32_t v = br_bits(&s->br, 8);
    if (s->br.fault) { st = ORC_RANGE; goto done; }
    if (v != (uint32_t)"BZh"[k]) { st = ORC_FALSE; goto done; }
  }
  s->block_size100k = (int)br_bits(&s->br, 8) - 0x30;
  if (s->br.fault) { st = ORC_RANGE; goto done; }
  if (s->block_size100k < 0 || s->block_size100k > 9) { st = ORC_FALSE; goto done; }
  s->tt = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(s->block_size100k * 100000 + 1));
  while (s->br.pos < n) { /* while (!input.isEOS) */
    int type = read_block_type(s);
    if (s->br.fault) { st = ORC_RANGE; goto done; }
    if (type < 0) { st = ORC_FALSE; goto done; }
    uint32_t stored = 0;
    for (int k = 0; k < 4; ++k) stored = (stored << 8) | br_bits(&s->br, 8);
    if (type == 0) {
      uint32_t crc = 0;
      int r = read_compressed(s, &crc);
      if (s->br.fault) { st = ORC_RANGE; goto done; }
      if (s->cap_fault) { st = ORC_CAP; goto done; }
      if (r < 0) { st = ORC_FALSE; goto done; }
      crc ^= 0xffffffffu;
      if (verify && crc != stored) { st = ORC_FALSE; goto done; }
      combined = ((combined << 1) | (combined >> 31)) & 0xffffffffu;
      combined ^= crc;
    } else {
      if (s->br.fault) { st = ORC_RANGE; goto done; }
      if (verify && stored != combined) { st = ORC_FALSE; goto done; }
      st = ORC_OK;
      goto done;
    }
  }
done:
  if (s->br.fault && st == ORC_OK) st = ORC_RANGE;
  if (out_len) *out_len = s->out_len;
  g_bz_last_pos = s->br.pos < n ? s->br.pos : n;
  free(s->tt);
  free(s);
  return st;
}
