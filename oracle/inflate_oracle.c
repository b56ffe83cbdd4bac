/*
 * oracle/inflate_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, single thread) of the reference's pure-Dart Inflate path.
 * It exists to CHECK the HIP product path; it is never linked into, imported by, or
 * called from libarchive_hip.so or the archive_amd package.  Only tests/, the smoke()
 * entry and bench.py's cpu_baseline leg may load it.
 *
 * Parity pinning: validated against the reference's own fixtures (tests/golden/, made by
 * tests/golden/make_golden.py from /root/reference/test/_data) and against C zlib on
 * valid streams (tests/test_oracle.py).  The reference is Dart and cannot run in this
 * image (no dart SDK), so there is no oracle/_ref build for this path.
 *
 * What is restated (all paths relative to /root/reference/lib/src):
 *   codecs/zlib/inflate.dart:104-116   _inflate            -> inf_run
 *   codecs/zlib/inflate.dart:120-156   _parseBlock         -> inf_block
 *   codecs/zlib/inflate.dart:159-184   _readBits           -> inf_bits
 *   codecs/zlib/inflate.dart:187-211   _readCodeByTable    -> inf_code
 *   codecs/zlib/inflate.dart:213-234   _parseUncompressedBlock -> inf_stored
 *   codecs/zlib/inflate.dart:239-298   _parseDynamicHuffmanBlock -> inf_dynamic
 *   codecs/zlib/inflate.dart:300-343   _decodeHuffman      -> inf_huffman
 *   codecs/zlib/inflate.dart:345-401   _decode             -> inf_lengths
 *   codecs/zlib/_huffman_table.dart:9-46  HuffmanTable     -> huff_build
 *   util/output_memory_stream.dart:79-98  writeBackReference -> out_backref
 *   codecs/zlib/_gzip_decoder_web.dart:27-138  decodeStream/_readHeader -> orc_gzip_decode
 *   codecs/zlib/_zlib_decoder_web.dart:31-107  decodeStream -> orc_zlib_decode
 *   util/crc32.dart:6-27, util/adler32.dart:29-52 -> orc_crc32 / orc_adler32
 *
 * Status codes (the reference itself is silent on errors; these make its behaviour visible):
 *   0  ORC_OK     decodeStream returned true / inflate ran to a final block or clean EOS
 *   1  ORC_FALSE  the reference stopped early (returned -1/false); output so far is kept
 *   2  ORC_RANGE  the reference would throw RangeError (read past the buffer, negative
 *                 back-reference source, code-length repeat overflow); caller gets no output
 *   3  ORC_HANG   the reference would not terminate (zero-length litlen table entry, quirk q3)
 *  -1  ORC_CAP    caller's output buffer too small (oracle artefact, not a reference state)
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

enum { ORC_OK = 0, ORC_FALSE = 1, ORC_RANGE = 2, ORC_HANG = 3, ORC_CAP = -1 };

/* ---- byte streams (util/input_memory_stream.dart, util/output_memory_stream.dart) ---- */
typedef struct { const uint8_t *p; size_t n; size_t pos; int big_endian; int fault; } istream_t;
typedef struct { uint8_t *p; size_t len; size_t cap; size_t base; int fault; } ostream_t;
/* ostream_t.base: index of the first byte of the current OutputMemoryStream inside p.
 * gzip shares one stream across members (base stays 0); zlib uses a fresh stream per member. */

static int is_eos(const istream_t *s) { return s->pos >= s->n; }
static size_t is_left(const istream_t *s) { return s->pos >= s->n ? 0 : s->n - s->pos; }
/* readByte with Dart's bounds check: out of range -> RangeError */
static int is_byte(istream_t *s) {
  if (s->pos >= s->n) { s->fault = ORC_RANGE; s->pos++; return 0; }
  return s->p[s->pos++];
}
static uint32_t is_u16(istream_t *s) {
  uint32_t a = (uint32_t)is_byte(s), b = (uint32_t)is_byte(s);
  return s->big_endian ? ((a << 8) | b) : ((b << 8) | a);
}
static uint32_t is_u32(istream_t *s) {
  uint32_t a = (uint32_t)is_byte(s), b = (uint32_t)is_byte(s), c = (uint32_t)is_byte(s), d = (uint32_t)is_byte(s);
  return s->big_endian ? ((a << 24) | (b << 16) | (c << 8) | d) : ((d << 24) | (c << 16) | (b << 8) | a);
}

static void out_byte(ostream_t *o, int v) {
  if (o->fault) return;
  if (o->len >= o->cap) { o->fault = ORC_CAP; return; }
  o->p[o->len++] = (uint8_t)v;
}
static void out_bytes(ostream_t *o, const uint8_t *src, size_t n) {
  if (o->fault) return;
  if (o->len + n > o->cap) { o->fault = ORC_CAP; return; }
  memcpy(o->p + o->len, src, n);
  o->len += n;
}
/* writeBackReference: forward byte copy (the bulk branch gives the same bytes when
 * distance >= count).  A source before the start of the stream is a Dart RangeError. */
static void out_backref(ostream_t *o, long distance, long count) {
  if (o->fault) return;
  if (count <= 0) return; /* setRange with an empty/negative range copies nothing */
  if ((long)(o->len - o->base) - distance < 0) { o->fault = ORC_RANGE; return; }
  if (o->len + (size_t)count > o->cap) { o->fault = ORC_CAP; return; }
  size_t s = o->len - (size_t)distance, d = o->len, e = o->len + (size_t)count;
  if (distance == 0) { o->len = e; return; } /* src == dst: bytes keep their (zero-filled) value */
  while (d < e) o->p[d++] = o->p[s++];
  o->len = e;
}

/* ---- HuffmanTable (_huffman_table.dart:9-46) ---- */
typedef struct { uint32_t table[1 << 15]; int maxlen; } huff_t;

static void huff_build(huff_t *h, const uint8_t *lengths, int n) {
  int maxlen = 0;
  for (int i = 0; i < n; ++i) if (lengths[i] > maxlen) maxlen = lengths[i];
  h->maxlen = maxlen;
  uint32_t size = 1u << maxlen;
  memset(h->table, 0, size * sizeof(uint32_t));
  uint32_t code = 0, skip = 2;
  for (int bl = 1; bl <= maxlen; ++bl) {
    for (int i = 0; i < n; ++i) {
      if (lengths[i] != bl) continue;
      uint32_t rev = 0, t = code;
      for (int j = 0; j < bl; ++j) { rev = (rev << 1) | (t & 1); t >>= 1; }
      for (uint32_t j = rev; j < size; j += skip) h->table[j] = ((uint32_t)bl << 16) | (uint32_t)i;
      ++code;
    }
    code <<= 1;
    skip <<= 1;
  }
}

/* ---- constant tables (inflate.dart:738-894; RFC 1951 values) ---- */
static const uint8_t k_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
static const uint16_t k_len_base[31] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59,
                                        67, 83, 99, 115, 131, 163, 195, 227, 258, 258, 258};
static const uint8_t k_len_extra[31] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3,
                                        4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0};
static const uint16_t k_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769,
                                         1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t k_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8,
                                         9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

/* ---- Inflate state ---- */
struct tables_s;
typedef struct {
  istream_t *in;
  ostream_t *out;
  uint64_t bitbuf; /* Dart int is 64-bit; at most 23 valid bits live here (pitfall p1) */
  int bitlen;
  int status;      /* ORC_OK until something notable happens */
  struct tables_s *T;
} inflate_t;

/* per-call table storage (heap; keeps the oracle re-entrant for the threaded CPU baseline) */
typedef struct tables_s { huff_t fixed_ll, fixed_d, cl, ll, d; } tables_t;

static void fixed_init(tables_t *T) {
  uint8_t l[288], d[30];
  for (int i = 0; i < 144; ++i) l[i] = 8;
  for (int i = 144; i < 256; ++i) l[i] = 9;
  for (int i = 256; i < 280; ++i) l[i] = 7;
  for (int i = 280; i < 288; ++i) l[i] = 8;
  for (int i = 0; i < 30; ++i) d[i] = 5;
  huff_build(&T->fixed_ll, l, 288);
  huff_build(&T->fixed_d, d, 30);
}

/* _readBits: -1 at end of input, 0 when length == 0 */
static int inf_bits(inflate_t *z, int length) {
  if (length == 0) return 0;
  while (z->bitlen < length) {
    if (is_eos(z->in)) return -1;
    uint64_t octet = z->in->p[z->in->pos++];
    z->bitbuf |= octet << z->bitlen;
    z->bitlen += 8;
  }
  int v = (int)(z->bitbuf & ((1ull << length) - 1));
  z->bitbuf >>= length;
  z->bitlen -= length;
  return v;
}

/* _readCodeByTable: needs maxCodeLength bits available even when the code is shorter (q2) */
static int inf_code(inflate_t *z, const huff_t *h, int *codelen_out) {
  while (z->bitlen < h->maxlen) {
    if (is_eos(z->in)) return -1;
    uint64_t octet = z->in->p[z->in->pos++];
    z->bitbuf |= octet << z->bitlen;
    z->bitlen += 8;
  }
  uint32_t e = h->table[z->bitbuf & ((1ull << h->maxlen) - 1)];
  int cl = (int)(e >> 16);
  z->bitbuf >>= cl;
  z->bitlen -= cl;
  if (codelen_out) *codelen_out = cl;
  return (int)(e & 0xffff);
}

static int inf_stored(inflate_t *z) {
  z->bitbuf = 0;
  z->bitlen = 0;
  int len = inf_bits(z, 16);
  int nlen = inf_bits(z, 16) ^ 0xffff;
  if (len != 0 && len != nlen) return -1;
  if (len > 0 && (size_t)len > is_left(z->in)) return -1;
  if (len < 0) len = 0; /* unreachable: len == -1 fails the check above */
  out_bytes(z->out, z->in->p + z->in->pos, (size_t)len);
  z->in->pos += (size_t)len;
  return 0;
}

static int inf_huffman(inflate_t *z, const huff_t *ll, const huff_t *d) {
  for (;;) {
    int cl;
    int code = inf_code(z, ll, &cl);
    if (code < 0 || code > 285) return -1;
    if (code == 256) break;
    if (code < 256) {
      if (cl == 0) { z->status = ORC_HANG; return -1; } /* q3: literal 0 forever */
      out_byte(z->out, code);
      if (z->out->fault) return -1;
      continue;
    }
    int ti = code - 257;
    long length = (long)k_len_base[ti] + inf_bits(z, k_len_extra[ti]); /* -1 at EOS is added as is */
    int dc = inf_code(z, d, NULL);
    if (dc < 0 || dc > 29) return -1;
    long distance = (long)k_dist_base[dc] + inf_bits(z, k_dist_extra[dc]);
    out_backref(z->out, distance, length);
    if (z->out->fault) return -1;
  }
  while (z->bitlen >= 8) { /* un-read whole bytes (inflate.dart:337-340) */
    z->bitlen -= 8;
    if (z->in->pos > 0) z->in->pos--;
  }
  return 0;
}

/* _decode: code-length RLE.  Writing past `num` entries is a Dart RangeError. */
static int inf_lengths(inflate_t *z, int num, const huff_t *cl, uint8_t *lens) {
  int prev = 0, i = 0;
  while (i < num) {
    int code = inf_code(z, cl, NULL);
    if (code == -1) return -1;
    int repeat, fill;
    switch (code) {
      case 16: repeat = inf_bits(z, 2); if (repeat == -1) return -1; repeat += 3; fill = prev; break;
      case 17: repeat = inf_bits(z, 3); if (repeat == -1) return -1; repeat += 3; fill = 0; prev = 0; break;
      case 18: repeat = inf_bits(z, 7); if (repeat == -1) return -1; repeat += 11; fill = 0; prev = 0; break;
      default:
        if (code < 0 || code > 15) return -1;
        repeat = 1; fill = code; prev = code; break;
    }
    while (repeat-- > 0) {
      if (i >= num) { z->status = ORC_RANGE; return -1; }
      lens[i++] = (uint8_t)fill;
    }
  }
  return 0;
}

static int inf_dynamic(inflate_t *z) {
  int hlit = inf_bits(z, 5);
  if (hlit == -1) return -1;
  hlit += 257;
  if (hlit > 288) return -1;
  int hdist = inf_bits(z, 5);
  if (hdist == -1) return -1;
  hdist += 1;
  if (hdist > 32) return -1;
  int hclen = inf_bits(z, 4);
  if (hclen == -1) return -1;
  hclen += 4;
  if (hclen > 19) return -1;
  uint8_t cl_lens[19];
  memset(cl_lens, 0, sizeof cl_lens);
  for (int i = 0; i < hclen; ++i) {
    int len = inf_bits(z, 3);
    if (len == -1) return -1;
    cl_lens[k_order[i]] = (uint8_t)len;
  }
  huff_build(&z->T->cl, cl_lens, 19);
  uint8_t lens[288 + 32];
  memset(lens, 0, sizeof lens);
  if (inf_lengths(z, hlit + hdist, &z->T->cl, lens) == -1) return -1;
  huff_build(&z->T->ll, lens, hlit);
  huff_build(&z->T->d, lens + hlit, hdist);
  return inf_huffman(z, &z->T->ll, &z->T->d);
}

/* _parseBlock: 1 = more blocks follow, 0 = stop */
static int inf_block(inflate_t *z, int *failed) {
  if (is_eos(z->in)) return 0;
  int hdr = inf_bits(z, 3); /* never -1 here: at least one byte is left */
  int final = (hdr & 1) != 0;
  int r;
  switch (hdr >> 1) {
    case 0: r = inf_stored(z); break;
    case 1: r = inf_huffman(z, &z->T->fixed_ll, &z->T->fixed_d); break;
    case 2: r = inf_dynamic(z); break;
    default: *failed = 1; return 0;
  }
  if (r == -1) { *failed = 1; return 0; }
  return !final;
}

/* Inflate / Inflate.stream constructor body: returns ORC_* */
static int inf_run(tables_t *T, istream_t *in, ostream_t *out) {
  inflate_t z;
  z.T = T;
  z.in = in; z.out = out; z.bitbuf = 0; z.bitlen = 0; z.status = ORC_OK;
  int failed = 0;
  while (!is_eos(in)) {
    if (!inf_block(&z, &failed)) break;
  }
  if (out->fault) return out->fault;
  if (z.status != ORC_OK) return z.status;
  return failed ? ORC_FALSE : ORC_OK;
}

static tables_t *tables_new(void) {
  tables_t *T = (tables_t *)malloc(sizeof(tables_t));
  fixed_init(T);
  return T;
}

/* ---- public oracle API ---- */
int orc_inflate_raw(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len, size_t *in_pos) {
  istream_t is = {in, n, 0, 0, 0};
  ostream_t os = {out, 0, cap, 0, 0};
  tables_t *T = tables_new();
  int st = inf_run(T, &is, &os);
  free(T);
  if (out_len) *out_len = os.len;
  if (in_pos) *in_pos = is.pos;
  return st;
}

/* _GZipDecoder._readHeader */
static int gz_header(istream_t *in) {
  if (is_u16(in) != 0x8b1f) return 0;
  if (is_byte(in) != 8) return 0;
  int flags = is_byte(in);
  (void)is_u32(in);
  (void)is_byte(in);
  (void)is_byte(in);
  if (flags & 0x04) {
    uint32_t t = is_u16(in);
    /* readBytes clamps to what is left */
    size_t left = is_left(in);
    in->pos += (t > left) ? left : t;
  }
  if (flags & 0x08) { while (!is_eos(in)) { if (is_byte(in) == 0) break; } }
  if (flags & 0x10) { while (!is_eos(in)) { if (is_byte(in) == 0) break; } }
  if (flags & 0x02) (void)is_u16(in);
  return 1;
}

static int zlib_stream(tables_t *T, istream_t *in, ostream_t *out, int verify, int raw);
uint32_t orc_adler32(const uint8_t *p, size_t n, uint32_t adler);

/* where the last orc_gzip_decode / orc_zlib_decode of this thread left its InputStream (what decodeStream consumed) */
static __thread size_t g_last_pos = 0;
size_t orc_last_position(void) { return g_last_pos; }

/* _GZipDecoder.decodeStream */
int orc_gzip_decode(const uint8_t *in, size_t n, int verify, int raw, uint8_t *out, size_t cap, size_t *out_len) {
  istream_t is = {in, n, 0, 0, 0};
  ostream_t os = {out, 0, cap, 0, 0};
  tables_t *T = tables_new();
  int st = ORC_OK;
  while (!is_eos(&is)) {
    size_t start = is.pos;
    int ok = gz_header(&is);
    if (is.fault) { st = is.fault; break; }
    if (!ok) {
      is.pos = start;
      /* falls back to the zlib decoder on the SAME (little-endian) stream */
      st = zlib_stream(T, &is, &os, verify, raw);
      break;
    }
    int r = inf_run(T, &is, &os);
    if (r == ORC_CAP || r == ORC_RANGE || r == ORC_HANG) { st = r; break; }
    (void)is_u32(&is);
    (void)is_u32(&is);
    if (is.fault) { st = is.fault; break; }
  }
  free(T);
  if (out_len) *out_len = os.len;
  g_last_pos = is.pos;
  return st;
}

/* _ZLibDecoder.decodeStream; members are inflated into a fresh buffer and appended only
 * after the next header passed its checks (deferred flush, quirk q7). */
static int zlib_stream(tables_t *T, istream_t *in, ostream_t *out, int verify, int raw) {
  size_t committed = out->len; /* bytes already written to `output` */
  int have_buffer = 0;
  size_t buf_start = out->len, buf_len = 0;
  while (!is_eos(in)) {
    if (!raw) {
      int cmf = is_byte(in);
      int flg = is_byte(in);
      if (in->fault) { out->len = committed; return in->fault; }
      if ((cmf & 8) != 8) { out->len = committed; return ORC_FALSE; }
      if (((cmf * 256) + flg) % 31 != 0) { out->len = committed; return ORC_FALSE; }
      if ((flg & 32) >> 5) {
        (void)is_u32(in);
        out->len = committed;
        return in->fault ? in->fault : ORC_FALSE;
      }
    }
    if (have_buffer) committed = buf_start + buf_len; /* output.writeBytes(buffer) */
    /* buffer = Inflate.stream(input).getBytes(): a fresh OutputMemoryStream */
    ostream_t member = {out->p, committed, out->cap, committed, 0};
    int r = inf_run(T, in, &member);
    if (r == ORC_CAP || r == ORC_RANGE || r == ORC_HANG) { out->len = committed; return r; }
    have_buffer = 1; buf_start = committed; buf_len = member.len - committed;
    if (!raw) {
      uint32_t want = is_u32(in);
      if (in->fault) { out->len = committed; return in->fault; }
      if (verify) {
        uint32_t got = orc_adler32(out->p + buf_start, buf_len, 1);
        if (want != got) { out->len = committed; return ORC_FALSE; }
      }
    }
  }
  if (have_buffer) committed = buf_start + buf_len;
  out->len = committed;
  return ORC_OK;
}

int orc_zlib_decode(const uint8_t *in, size_t n, int verify, int raw, uint8_t *out, size_t cap, size_t *out_len) {
  istream_t is = {in, n, 0, 1 /* decodeBytes builds a big-endian stream, p8 */, 0};
  ostream_t os = {out, 0, cap, 0, 0};
  tables_t *T = tables_new();
  int st = zlib_stream(T, &is, &os, verify, raw);
  free(T);
  if (out_len) *out_len = os.len;
  g_last_pos = is.pos;
  return st;
}

/* ---- checksums ---- */
static uint32_t g_crc_table[256];
static int g_crc_ready = 0;
static void crc_init(void) {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
    g_crc_table[i] = c;
  }
  g_crc_ready = 1;
}
/* getCrc32(array, crc): chainable */
uint32_t orc_crc32(const uint8_t *p, size_t n, uint32_t crc) {
  if (!g_crc_ready) crc_init();
  crc ^= 0xffffffffu;
  for (size_t i = 0; i < n; ++i) crc = g_crc_table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
  return crc ^ 0xffffffffu;
}
/* getAdler32(array, adler): 3800-byte deferral of the modulo */
uint32_t orc_adler32(const uint8_t *p, size_t n, uint32_t adler) {
  uint64_t s1 = adler & 0xffff, s2 = adler >> 16;
  size_t i = 0;
  while (n > 0) {
    size_t k = n < 3800 ? n : 3800;
    n -= k;
    while (k--) { s1 += p[i++]; s2 += s1; }
    s1 %= 65521; s2 %= 65521;
  }
  return (uint32_t)((s2 << 16) | s1);
}
