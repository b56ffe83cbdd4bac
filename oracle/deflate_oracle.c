/*
 * oracle/deflate_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, single thread) of the reference's pure-Dart Deflate encoder
 * (/root/reference/lib/src/codecs/zlib/deflate.dart -- a Dart transliteration of zlib 1.1.x /
 * JZlib).  It defines "the reference's compressed size at level L" for the size-tolerance
 * check of the HIP encoder and produces reference-shaped streams for decode tests.  It is never
 * linked into or called from the product.
 *
 * Parity pinning: NO test of the reference pins compressed bytes (only round trips,
 * test/deflate_test.dart:12-44) and Dart cannot run here, so byte-level parity of this file
 * with the Dart code is *unpinned*.  What is pinned (tests/test_deflate_oracle.py):
 *   - with the reference's block-truncation heuristic switched OFF (it is `#ifdef
 *     TRUNCATE_BLOCK`-ed out of stock zlib) the output is byte-identical to C zlib 1.2.11
 *     (`zlib.compressobj(level, DEFLATED, -15, 8)`) at levels 1..9 on every test corpus --
 *     i.e. hash, lazy matching, longest_match, tree construction and block emission are right;
 *   - with the heuristic ON (the reference's behaviour, deflate.dart:549-562) every stream
 *     still inflates to the input through both the Inflate oracle and zlib.
 *
 * What is restated (deflate.dart line numbers):
 *   :102-169  _init            -> df_init          :816-888   _fillWindow   -> fill_window
 *   :172-239  _deflate         -> orc_deflate_raw  :895-992   _deflateFast  -> deflate_fast
 *   :290-314  _pqdownheap/_smaller                 :997-1118  _deflateSlow  -> deflate_slow
 *   :318-392  _scanTree/_buildBitLengthTree        :1120-1206 _longestMatch -> longest_match
 *   :397-462  _sendAllTrees/_sendTree              :691-737   _deflateStored-> deflate_stored
 *   :483-499  _sendCode/_sendBits                  :740-807   _trStoredBlock/_trFlushBlock
 *   :531-568  _trTally (incl. truncation heuristic):2567-2648 _genBitlen
 *   :571-614  _compressBlock                       :2656-2736 _buildTree
 *   :640-675  biFlush/_biWindup/_copyBlock         :2746-2784 _genCodes/_reverseBits
 *   :1250-1275 _getConfig (level table)
 *   codecs/zlib/_gzip_encoder_web.dart:27-100, _zlib_encoder_web.dart:27-73 -> orc_gzip_encode / orc_zlib_encode
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

uint32_t orc_crc32(const uint8_t *p, size_t n, uint32_t crc);
uint32_t orc_adler32(const uint8_t *p, size_t n, uint32_t adler);

enum { MIN_MATCH = 3, MAX_MATCH = 258, MIN_LOOKAHEAD = 262, W_SIZE_MAX = 32768, HASH_SIZE = 32768,
       HASH_MASK = 32767, HASH_SHIFT = 5, LIT_BUFSIZE = 16384, L_CODES = 286, D_CODES = 30, BL_CODES = 19,
       HEAP_SIZE = 573, MAX_BITS = 15, MAX_BL_BITS = 7, END_BLOCK = 256, LITERALS = 256, TOO_FAR = 4096 };
enum { FN_STORED = 0, FN_FAST = 1, FN_SLOW = 2 };

static const uint8_t extra_lbits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint8_t extra_dbits[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t extra_blbits[19] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 3, 7};
static const uint8_t bl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* static tables, generated the way zlib's tr_static_init does (deflate.dart:1711-2548, 2801-3441 hold the
 * same values as literals) */
static uint8_t g_length_code[256], g_dist_code[512];
static uint16_t g_base_length[29], g_base_dist[30];
static uint16_t g_static_ltree[288 * 2], g_static_dtree[30 * 2];
static int g_static_ready = 0;

static unsigned bi_reverse(unsigned code, int len) {
  unsigned res = 0;
  do { res |= code & 1; code >>= 1; res <<= 1; } while (--len > 0);
  return res >> 1;
}
static void gen_codes(uint16_t *tree, int max_code, const uint16_t *bl_count) {
  uint16_t next_code[MAX_BITS + 1];
  unsigned code = 0;
  next_code[0] = 0;
  for (int bits = 1; bits <= MAX_BITS; bits++) { code = (code + bl_count[bits - 1]) << 1; next_code[bits] = (uint16_t)code; }
  for (int n = 0; n <= max_code; n++) {
    int len = tree[n * 2 + 1];
    if (len == 0) continue;
    tree[n * 2] = (uint16_t)bi_reverse(next_code[len]++, len);
  }
}
static void static_init(void) {
  if (g_static_ready) return;
  int length = 0, code, n, dist = 0;
  for (code = 0; code < 28; code++) {
    g_base_length[code] = (uint16_t)length;
    for (n = 0; n < (1 << extra_lbits[code]); n++) g_length_code[length++] = (uint8_t)code;
  }
  g_length_code[length - 1] = (uint8_t)code; /* length 258 -> code 28 */
  g_base_length[28] = 0;                     /* deflate.dart:2514: last entry is 0 */
  for (code = 0; code < 16; code++) {
    g_base_dist[code] = (uint16_t)dist;
    for (n = 0; n < (1 << extra_dbits[code]); n++) g_dist_code[dist++] = (uint8_t)code;
  }
  dist >>= 7;
  for (; code < D_CODES; code++) {
    g_base_dist[code] = (uint16_t)(dist << 7);
    for (n = 0; n < (1 << (extra_dbits[code] - 7)); n++) g_dist_code[256 + dist++] = (uint8_t)code;
  }
  uint16_t bl_count[MAX_BITS + 1];
  memset(bl_count, 0, sizeof bl_count);
  n = 0;
  while (n <= 143) { g_static_ltree[n++ * 2 + 1] = 8; bl_count[8]++; }
  while (n <= 255) { g_static_ltree[n++ * 2 + 1] = 9; bl_count[9]++; }
  while (n <= 279) { g_static_ltree[n++ * 2 + 1] = 7; bl_count[7]++; }
  while (n <= 287) { g_static_ltree[n++ * 2 + 1] = 8; bl_count[8]++; }
  gen_codes(g_static_ltree, 287, bl_count);
  for (n = 0; n < D_CODES; n++) { g_static_dtree[n * 2 + 1] = 5; g_static_dtree[n * 2] = (uint16_t)bi_reverse((unsigned)n, 5); }
  g_static_ready = 1;
}
static int d_code(int dist) { return dist < 256 ? g_dist_code[dist] : g_dist_code[256 + (dist >> 7)]; }

typedef struct {
  uint16_t *dyn_tree;
  int max_code;
  const uint16_t *static_tree; /* or NULL */
  const uint8_t *extra_bits;
  int extra_base, elems, max_length;
} tree_desc;

typedef struct {
  const uint8_t *in; size_t in_len, in_pos;
  uint8_t *out; size_t out_len, out_cap; int overflow;
  int level, good_length, max_lazy, nice_length, max_chain, func, truncate;
  uint8_t window[2 * W_SIZE_MAX];
  uint16_t prev[W_SIZE_MAX], head[HASH_SIZE];
  int w_size, w_mask; /* 1 << windowBits (deflate.dart:110-124: windowBits 9..15) */
  int ins_h, strstart, block_start, lookahead, match_length, prev_length, match_available, match_start, prev_match;
  uint16_t dyn_ltree[HEAP_SIZE * 2], dyn_dtree[(2 * D_CODES + 1) * 2], bl_tree[(2 * BL_CODES + 1) * 2];
  tree_desc l_desc, d_desc, bl_desc;
  uint16_t bl_count[MAX_BITS + 1];
  uint32_t heap[2 * L_CODES + 1];
  int heap_len, heap_max;
  uint8_t depth[2 * L_CODES + 1];
  uint16_t d_buf[LIT_BUFSIZE];
  uint8_t l_buf[LIT_BUFSIZE];
  int last_lit, matches;
  long opt_len, static_len;
  unsigned bi_buf; int bi_valid;
  uint32_t crc; size_t total;
} deflate_t;

/* ---- output ---- */
static void put_byte(deflate_t *s, int c) {
  if (s->out_len >= s->out_cap) { s->overflow = 1; return; }
  s->out[s->out_len++] = (uint8_t)c; /* Uint8List store truncates (pitfall p2) */
}
static void put_short(deflate_t *s, unsigned w) { put_byte(s, (int)(w & 0xff)); put_byte(s, (int)((w >> 8) & 0xff)); }
static void send_bits(deflate_t *s, unsigned value, int length) {
  if (s->bi_valid > 16 - length) {
    s->bi_buf |= (value << s->bi_valid) & 0xffff; /* p3: a | ((v << n) & 0xffff) */
    put_short(s, s->bi_buf);
    s->bi_buf = value >> (16 - s->bi_valid);
    s->bi_valid += length - 16;
  } else {
    s->bi_buf |= (value << s->bi_valid) & 0xffff;
    s->bi_valid += length;
  }
}
static void send_code(deflate_t *s, int c, const uint16_t *tree) { send_bits(s, tree[c * 2], tree[c * 2 + 1]); }
static void bi_windup(deflate_t *s) {
  if (s->bi_valid > 8) put_short(s, s->bi_buf);
  else if (s->bi_valid > 0) put_byte(s, (int)s->bi_buf);
  s->bi_buf = 0; s->bi_valid = 0;
}
static void copy_block(deflate_t *s, int buf, int len, int header) {
  bi_windup(s);
  if (header) { put_short(s, (unsigned)len); put_short(s, (unsigned)((~len + 0x10000) & 0xffff)); }
  for (int i = 0; i < len; i++) put_byte(s, s->window[buf + i]);
}

/* ---- trees ---- */
static void init_block(deflate_t *s) {
  for (int n = 0; n < L_CODES; n++) s->dyn_ltree[n * 2] = 0;
  for (int n = 0; n < D_CODES; n++) s->dyn_dtree[n * 2] = 0;
  for (int n = 0; n < BL_CODES; n++) s->bl_tree[n * 2] = 0;
  s->dyn_ltree[END_BLOCK * 2] = 1;
  s->opt_len = s->static_len = 0;
  s->last_lit = s->matches = 0;
}
static int smaller(const uint16_t *tree, int n, int m, const uint8_t *depth) {
  return tree[n * 2] < tree[m * 2] || (tree[n * 2] == tree[m * 2] && depth[n] <= depth[m]); /* p6 */
}
static void pqdownheap(deflate_t *s, const uint16_t *tree, int k) {
  int v = (int)s->heap[k], j = k << 1;
  while (j <= s->heap_len) {
    if (j < s->heap_len && smaller(tree, (int)s->heap[j + 1], (int)s->heap[j], s->depth)) j++;
    if (smaller(tree, v, (int)s->heap[j], s->depth)) break;
    s->heap[k] = s->heap[j];
    k = j;
    j <<= 1;
  }
  s->heap[k] = (uint32_t)v;
}
static void gen_bitlen(deflate_t *s, tree_desc *desc) {
  uint16_t *tree = desc->dyn_tree;
  const uint16_t *stree = desc->static_tree;
  int h, n, m, bits, xbits, overflow = 0, max_length = desc->max_length;
  for (bits = 0; bits <= MAX_BITS; bits++) s->bl_count[bits] = 0;
  tree[s->heap[s->heap_max] * 2 + 1] = 0;
  for (h = s->heap_max + 1; h < HEAP_SIZE; h++) {
    n = (int)s->heap[h];
    bits = tree[tree[n * 2 + 1] * 2 + 1] + 1;
    if (bits > max_length) { bits = max_length; overflow++; }
    tree[n * 2 + 1] = (uint16_t)bits;
    if (n > desc->max_code) continue;
    s->bl_count[bits]++;
    xbits = 0;
    if (n >= desc->extra_base) xbits = desc->extra_bits[n - desc->extra_base];
    long f = tree[n * 2];
    s->opt_len += f * (bits + xbits);
    if (stree) s->static_len += f * (stree[n * 2 + 1] + xbits);
  }
  if (overflow == 0) return;
  do {
    bits = max_length - 1;
    while (s->bl_count[bits] == 0) bits--;
    s->bl_count[bits]--;
    s->bl_count[bits + 1] = (uint16_t)(s->bl_count[bits + 1] + 2);
    s->bl_count[max_length]--;
    overflow -= 2;
  } while (overflow > 0);
  for (bits = max_length; bits != 0; bits--) {
    n = s->bl_count[bits];
    while (n != 0) {
      m = (int)s->heap[--h];
      if (m > desc->max_code) continue;
      if (tree[m * 2 + 1] != bits) {
        s->opt_len += ((long)bits - (long)tree[m * 2 + 1]) * (long)tree[m * 2];
        tree[m * 2 + 1] = (uint16_t)bits;
      }
      n--;
    }
  }
}
static void build_tree(deflate_t *s, tree_desc *desc) {
  uint16_t *tree = desc->dyn_tree;
  const uint16_t *stree = desc->static_tree;
  int elems = desc->elems, n, m, max_code = -1, node;
  s->heap_len = 0;
  s->heap_max = HEAP_SIZE;
  for (n = 0; n < elems; n++) {
    if (tree[n * 2] != 0) { s->heap[++s->heap_len] = (uint32_t)(max_code = n); s->depth[n] = 0; }
    else tree[n * 2 + 1] = 0;
  }
  while (s->heap_len < 2) {
    node = (int)(s->heap[++s->heap_len] = (uint32_t)(max_code < 2 ? ++max_code : 0));
    tree[node * 2] = 1;
    s->depth[node] = 0;
    s->opt_len--;
    if (stree) s->static_len -= stree[node * 2 + 1];
  }
  desc->max_code = max_code;
  for (n = s->heap_len / 2; n >= 1; n--) pqdownheap(s, tree, n);
  node = elems;
  do {
    n = (int)s->heap[1];
    s->heap[1] = s->heap[s->heap_len--];
    pqdownheap(s, tree, 1);
    m = (int)s->heap[1];
    s->heap[--s->heap_max] = (uint32_t)n;
    s->heap[--s->heap_max] = (uint32_t)m;
    tree[node * 2] = (uint16_t)(tree[n * 2] + tree[m * 2]);
    s->depth[node] = (uint8_t)((s->depth[n] > s->depth[m] ? s->depth[n] : s->depth[m]) + 1);
    tree[n * 2 + 1] = tree[m * 2 + 1] = (uint16_t)node;
    s->heap[1] = (uint32_t)node++;
    pqdownheap(s, tree, 1);
  } while (s->heap_len >= 2);
  s->heap[--s->heap_max] = s->heap[1];
  gen_bitlen(s, desc);
  gen_codes(tree, max_code, s->bl_count);
}
static void scan_tree(deflate_t *s, uint16_t *tree, int max_code) {
  int n, prevlen = -1, curlen, nextlen = tree[0 * 2 + 1], count = 0, max_count = 7, min_count = 4;
  if (nextlen == 0) { max_count = 138; min_count = 3; }
  tree[(max_code + 1) * 2 + 1] = 0xffff;
  for (n = 0; n <= max_code; n++) {
    curlen = nextlen;
    nextlen = tree[(n + 1) * 2 + 1];
    if (++count < max_count && curlen == nextlen) continue;
    else if (count < min_count) s->bl_tree[curlen * 2] = (uint16_t)(s->bl_tree[curlen * 2] + count);
    else if (curlen != 0) { if (curlen != prevlen) s->bl_tree[curlen * 2]++; s->bl_tree[16 * 2]++; }
    else if (count <= 10) s->bl_tree[17 * 2]++;
    else s->bl_tree[18 * 2]++;
    count = 0; prevlen = curlen;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
    else if (curlen == nextlen) { max_count = 6; min_count = 3; }
    else { max_count = 7; min_count = 4; }
  }
}
static void send_tree(deflate_t *s, const uint16_t *tree, int max_code) {
  int n, prevlen = -1, curlen, nextlen = tree[0 * 2 + 1], count = 0, max_count = 7, min_count = 4;
  if (nextlen == 0) { max_count = 138; min_count = 3; }
  for (n = 0; n <= max_code; n++) {
    curlen = nextlen;
    nextlen = tree[(n + 1) * 2 + 1];
    if (++count < max_count && curlen == nextlen) continue;
    else if (count < min_count) { do { send_code(s, curlen, s->bl_tree); } while (--count != 0); }
    else if (curlen != 0) {
      if (curlen != prevlen) { send_code(s, curlen, s->bl_tree); count--; }
      send_code(s, 16, s->bl_tree); send_bits(s, (unsigned)(count - 3), 2);
    } else if (count <= 10) { send_code(s, 17, s->bl_tree); send_bits(s, (unsigned)(count - 3), 3); }
    else { send_code(s, 18, s->bl_tree); send_bits(s, (unsigned)(count - 11), 7); }
    count = 0; prevlen = curlen;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
    else if (curlen == nextlen) { max_count = 6; min_count = 3; }
    else { max_count = 7; min_count = 4; }
  }
}
static int build_bl_tree(deflate_t *s) {
  int max_blindex;
  scan_tree(s, s->dyn_ltree, s->l_desc.max_code);
  scan_tree(s, s->dyn_dtree, s->d_desc.max_code);
  build_tree(s, &s->bl_desc);
  for (max_blindex = BL_CODES - 1; max_blindex >= 3; max_blindex--)
    if (s->bl_tree[bl_order[max_blindex] * 2 + 1] != 0) break;
  s->opt_len += 3 * (max_blindex + 1) + 5 + 5 + 4;
  return max_blindex;
}
static void send_all_trees(deflate_t *s, int lcodes, int dcodes, int blcodes) {
  send_bits(s, (unsigned)(lcodes - 257), 5);
  send_bits(s, (unsigned)(dcodes - 1), 5);
  send_bits(s, (unsigned)(blcodes - 4), 4);
  for (int rank = 0; rank < blcodes; rank++) send_bits(s, s->bl_tree[bl_order[rank] * 2 + 1], 3);
  send_tree(s, s->dyn_ltree, lcodes - 1);
  send_tree(s, s->dyn_dtree, dcodes - 1);
}
static void compress_block(deflate_t *s, const uint16_t *ltree, const uint16_t *dtree) {
  int lx = 0;
  if (s->last_lit != 0) {
    do {
      int dist = s->d_buf[lx], lc = s->l_buf[lx];
      lx++;
      if (dist == 0) send_code(s, lc, ltree);
      else {
        int code = g_length_code[lc];
        send_code(s, code + LITERALS + 1, ltree);
        int extra = extra_lbits[code];
        if (extra != 0) { lc -= g_base_length[code]; send_bits(s, (unsigned)lc, extra); }
        dist--;
        code = d_code(dist);
        send_code(s, code, dtree);
        extra = extra_dbits[code];
        if (extra != 0) { dist -= g_base_dist[code]; send_bits(s, (unsigned)dist, extra); }
      }
    } while (lx < s->last_lit);
  }
  send_code(s, END_BLOCK, ltree);
}
static void tr_stored_block(deflate_t *s, int buf, int stored_len, int eof) {
  send_bits(s, (unsigned)((0 << 1) + (eof ? 1 : 0)), 3);
  copy_block(s, buf, stored_len, 1);
}
static void tr_flush_block(deflate_t *s, int buf, int stored_len, int eof) {
  long opt_lenb, static_lenb;
  int max_blindex = 0;
  if (s->level > 0) {
    build_tree(s, &s->l_desc);
    build_tree(s, &s->d_desc);
    max_blindex = build_bl_tree(s);
    opt_lenb = (s->opt_len + 3 + 7) >> 3;
    static_lenb = (s->static_len + 3 + 7) >> 3;
    if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
  } else {
    opt_lenb = static_lenb = stored_len + 5;
  }
  if (stored_len + 4 <= opt_lenb && buf != -1) {
    tr_stored_block(s, buf, stored_len, eof);
  } else if (static_lenb == opt_lenb) {
    send_bits(s, (unsigned)((1 << 1) + (eof ? 1 : 0)), 3);
    compress_block(s, g_static_ltree, g_static_dtree);
  } else {
    send_bits(s, (unsigned)((2 << 1) + (eof ? 1 : 0)), 3);
    send_all_trees(s, s->l_desc.max_code + 1, s->d_desc.max_code + 1, max_blindex + 1);
    compress_block(s, s->dyn_ltree, s->dyn_dtree);
  }
  init_block(s);
  if (eof) bi_windup(s);
}
static void flush_block_only(deflate_t *s, int eof) {
  tr_flush_block(s, s->block_start >= 0 ? s->block_start : -1, s->strstart - s->block_start, eof);
  s->block_start = s->strstart;
}
/* _trTally: true when the block must be flushed */
static int tr_tally(deflate_t *s, int dist, int lc) {
  s->d_buf[s->last_lit] = (uint16_t)dist;
  s->l_buf[s->last_lit] = (uint8_t)lc;
  s->last_lit++;
  if (dist == 0) s->dyn_ltree[lc * 2]++;
  else {
    s->matches++;
    dist--;
    s->dyn_ltree[(g_length_code[lc] + LITERALS + 1) * 2]++;
    s->dyn_dtree[d_code(dist) * 2]++;
  }
  if (s->truncate && (s->last_lit & 0x1fff) == 0 && s->level > 2) {
    long out_length = (long)s->last_lit * 8;
    long in_length = (long)s->strstart - s->block_start;
    for (int dcode = 0; dcode < D_CODES; dcode++) out_length += (long)s->dyn_dtree[dcode * 2] * (5 + extra_dbits[dcode]);
    out_length >>= 3;
    /* Dart: `_matches < (_lastLit / 2)` and `outLength < inLength / 2` are double divisions */
    if ((double)s->matches < (double)s->last_lit / 2.0 && (double)out_length < (double)in_length / 2.0) return 1;
  }
  return s->last_lit == LIT_BUFSIZE - 1;
}

/* ---- window ---- */
static int read_buf(deflate_t *s, int start, int size) {
  if (size == 0 || s->in_pos >= s->in_len) return 0;
  size_t len = s->in_len - s->in_pos;
  if (len > (size_t)size) len = (size_t)size;
  memcpy(s->window + start, s->in + s->in_pos, len);
  s->crc = orc_crc32(s->in + s->in_pos, len, s->crc);
  s->in_pos += len;
  s->total += len;
  return (int)len;
}
static void fill_window(deflate_t *s) {
  do {
    int more = 2 * s->w_size - s->lookahead - s->strstart;
    if (more == 0 && s->strstart == 0 && s->lookahead == 0) more = s->w_size;
    else if (s->strstart >= s->w_size + s->w_size - MIN_LOOKAHEAD) {
      memcpy(s->window, s->window + s->w_size, s->w_size);
      s->match_start -= s->w_size;
      s->strstart -= s->w_size;
      s->block_start -= s->w_size;
      for (int p = 0; p < HASH_SIZE; p++) { unsigned m = s->head[p]; s->head[p] = (uint16_t)(m >= s->w_size ? m - s->w_size : 0); }
      for (int p = 0; p < s->w_size; p++) { unsigned m = s->prev[p]; s->prev[p] = (uint16_t)(m >= s->w_size ? m - s->w_size : 0); }
      more += s->w_size;
    }
    if (s->in_pos >= s->in_len) return;
    int n = read_buf(s, s->strstart + s->lookahead, more);
    s->lookahead += n;
    if (s->lookahead >= MIN_MATCH) {
      s->ins_h = s->window[s->strstart];
      s->ins_h = ((s->ins_h << HASH_SHIFT) ^ s->window[s->strstart + 1]) & HASH_MASK;
    }
  } while (s->lookahead < MIN_LOOKAHEAD && s->in_pos < s->in_len);
}
#define INSERT_STRING(s, hash_head)                                                              \
  do {                                                                                           \
    (s)->ins_h = (((s)->ins_h << HASH_SHIFT) ^ (s)->window[(s)->strstart + (MIN_MATCH - 1)]) & HASH_MASK; \
    (hash_head) = (s)->head[(s)->ins_h];                                                         \
    (s)->prev[(s)->strstart & s->w_mask] = (s)->head[(s)->ins_h];                                   \
    (s)->head[(s)->ins_h] = (uint16_t)(s)->strstart;                                             \
  } while (0)

static int longest_match(deflate_t *s, int cur_match) {
  int chain_length = s->max_chain, scan = s->strstart, match, len, best_len = s->prev_length;
  int limit = s->strstart > (s->w_size - MIN_LOOKAHEAD) ? s->strstart - (s->w_size - MIN_LOOKAHEAD) : 0;
  int nice_match = s->nice_length;
  const uint8_t *w = s->window;
  int strend = s->strstart + MAX_MATCH;
  uint8_t scan_end1 = w[scan + best_len - 1], scan_end = w[scan + best_len];
  if (s->prev_length >= s->good_length) chain_length >>= 2;
  if (nice_match > s->lookahead) nice_match = s->lookahead;
  do {
    match = cur_match;
    if (w[match + best_len] != scan_end || w[match + best_len - 1] != scan_end1 || w[match] != w[scan] ||
        w[++match] != w[scan + 1])
      continue;
    scan += 2;
    match++;
    do {
    } while (w[++scan] == w[++match] && w[++scan] == w[++match] && w[++scan] == w[++match] && w[++scan] == w[++match] &&
             w[++scan] == w[++match] && w[++scan] == w[++match] && w[++scan] == w[++match] && w[++scan] == w[++match] &&
             scan < strend);
    len = MAX_MATCH - (strend - scan);
    scan = strend - MAX_MATCH;
    if (len > best_len) {
      s->match_start = cur_match;
      best_len = len;
      if (len >= nice_match) break;
      scan_end1 = w[scan + best_len - 1];
      scan_end = w[scan + best_len];
    }
  } while ((cur_match = s->prev[cur_match & s->w_mask]) > limit && --chain_length != 0);
  return best_len <= s->lookahead ? best_len : s->lookahead;
}

static void deflate_stored(deflate_t *s) {
  int max_block_size = 0xffff;
  if (max_block_size > LIT_BUFSIZE * 4 - 5) max_block_size = LIT_BUFSIZE * 4 - 5;
  for (;;) {
    if (s->lookahead <= 1) {
      fill_window(s);
      if (s->lookahead == 0) break;
    }
    s->strstart += s->lookahead;
    s->lookahead = 0;
    int max_start = s->block_start + max_block_size;
    if (s->strstart >= max_start) {
      s->lookahead = s->strstart - max_start;
      s->strstart = max_start;
      flush_block_only(s, 0);
    }
    if (s->strstart - s->block_start >= s->w_size - MIN_LOOKAHEAD) flush_block_only(s, 0);
  }
  flush_block_only(s, 1);
}
static void deflate_fast(deflate_t *s) {
  int hash_head = 0, bflush;
  for (;;) {
    if (s->lookahead < MIN_LOOKAHEAD) {
      fill_window(s);
      if (s->lookahead == 0) break;
    }
    if (s->lookahead >= MIN_MATCH) INSERT_STRING(s, hash_head);
    if (hash_head != 0 && ((s->strstart - hash_head) & 0xffff) <= s->w_size - MIN_LOOKAHEAD)
      s->match_length = longest_match(s, hash_head);
    if (s->match_length >= MIN_MATCH) {
      bflush = tr_tally(s, s->strstart - s->match_start, s->match_length - MIN_MATCH);
      s->lookahead -= s->match_length;
      if (s->match_length <= s->max_lazy && s->lookahead >= MIN_MATCH) {
        s->match_length--;
        do { s->strstart++; INSERT_STRING(s, hash_head); } while (--s->match_length != 0);
        s->strstart++;
      } else {
        s->strstart += s->match_length;
        s->match_length = 0;
        s->ins_h = s->window[s->strstart];
        s->ins_h = ((s->ins_h << HASH_SHIFT) ^ s->window[s->strstart + 1]) & HASH_MASK;
      }
    } else {
      bflush = tr_tally(s, 0, s->window[s->strstart]);
      s->lookahead--;
      s->strstart++;
    }
    if (bflush) flush_block_only(s, 0);
  }
  flush_block_only(s, 1);
}
static void deflate_slow(deflate_t *s) {
  int hash_head = 0, bflush;
  for (;;) {
    if (s->lookahead < MIN_LOOKAHEAD) {
      fill_window(s);
      if (s->lookahead == 0) break;
    }
    if (s->lookahead >= MIN_MATCH) INSERT_STRING(s, hash_head);
    s->prev_length = s->match_length;
    s->prev_match = s->match_start;
    s->match_length = MIN_MATCH - 1;
    if (hash_head != 0 && s->prev_length < s->max_lazy &&
        ((s->strstart - hash_head) & 0xffff) <= s->w_size - MIN_LOOKAHEAD) {
      s->match_length = longest_match(s, hash_head);
      if (s->match_length <= 5 && (s->match_length == MIN_MATCH && s->strstart - s->match_start > TOO_FAR))
        s->match_length = MIN_MATCH - 1;
    }
    if (s->prev_length >= MIN_MATCH && s->match_length <= s->prev_length) {
      int max_insert = s->strstart + s->lookahead - MIN_MATCH;
      bflush = tr_tally(s, s->strstart - 1 - s->prev_match, s->prev_length - MIN_MATCH);
      s->lookahead -= s->prev_length - 1;
      s->prev_length -= 2;
      do {
        if (++s->strstart <= max_insert) INSERT_STRING(s, hash_head);
      } while (--s->prev_length != 0);
      s->match_available = 0;
      s->match_length = MIN_MATCH - 1;
      s->strstart++;
      if (bflush) flush_block_only(s, 0);
    } else if (s->match_available != 0) {
      bflush = tr_tally(s, 0, s->window[s->strstart - 1]);
      if (bflush) flush_block_only(s, 0);
      s->strstart++;
      s->lookahead--;
    } else {
      s->match_available = 1;
      s->strstart++;
      s->lookahead--;
    }
  }
  if (s->match_available != 0) {
    (void)tr_tally(s, 0, s->window[s->strstart - 1]);
    s->match_available = 0;
  }
  flush_block_only(s, 1);
}

static int df_init(deflate_t *s, int level, int window_bits) {
  static const int cfg[10][5] = {{0, 0, 0, 0, FN_STORED},     {4, 4, 8, 4, FN_FAST},        {4, 5, 16, 8, FN_FAST},
                                 {4, 6, 32, 32, FN_FAST},     {4, 4, 16, 16, FN_SLOW},      {8, 16, 32, 32, FN_SLOW},
                                 {8, 16, 128, 128, FN_SLOW},  {8, 32, 128, 256, FN_SLOW},   {32, 128, 258, 1024, FN_SLOW},
                                 {32, 258, 258, 4096, FN_SLOW}};
  if (window_bits < 9 || window_bits > 15) return 0; /* deflate.dart:110-111 */
  s->w_size = 1 << window_bits; s->w_mask = s->w_size - 1;
  if (level < 0 || level > 9) return 0;
  static_init();
  s->level = level;
  s->good_length = cfg[level][0]; s->max_lazy = cfg[level][1]; s->nice_length = cfg[level][2];
  s->max_chain = cfg[level][3]; s->func = cfg[level][4];
  memset(s->head, 0, sizeof s->head);
  s->strstart = 0; s->block_start = 0; s->lookahead = 0;
  s->match_length = s->prev_length = MIN_MATCH - 1;
  s->match_available = 0; s->ins_h = 0; s->match_start = 0; s->prev_match = 0;
  s->l_desc = (tree_desc){s->dyn_ltree, 0, g_static_ltree, extra_lbits, LITERALS + 1, L_CODES, MAX_BITS};
  s->d_desc = (tree_desc){s->dyn_dtree, 0, g_static_dtree, extra_dbits, 0, D_CODES, MAX_BITS};
  s->bl_desc = (tree_desc){s->bl_tree, 0, NULL, extra_blbits, 0, BL_CODES, MAX_BL_BITS};
  s->bi_buf = 0; s->bi_valid = 0;
  s->crc = 0; s->total = 0;
  memset(s->dyn_ltree, 0, sizeof s->dyn_ltree);
  memset(s->dyn_dtree, 0, sizeof s->dyn_dtree);
  memset(s->bl_tree, 0, sizeof s->bl_tree);
  init_block(s);
  return 1;
}

/* Deflate(bytes, level: L).getBytes().  truncate_heuristic: 1 = the reference's behaviour
 * (deflate.dart:549-562), 0 = stock zlib.  Returns 0, or -1 if `cap` is too small; with an
 * invalid level nothing is written (the reference's _init returns false, deflate.dart:108-121). */
int orc_deflate_raw_wb(const uint8_t *in, size_t n, int level, int window_bits, int truncate_heuristic, uint8_t *out, size_t cap,
                       size_t *out_len, uint32_t *crc_out);
int orc_deflate_raw(const uint8_t *in, size_t n, int level, int truncate_heuristic, uint8_t *out, size_t cap,
                    size_t *out_len, uint32_t *crc_out) {
  return orc_deflate_raw_wb(in, n, level, 15, truncate_heuristic, out, cap, out_len, crc_out);
}
/* Deflate(bytes, level: L, windowBits: W).getBytes()  (deflate.dart:39-48,102-124) */
int orc_deflate_raw_wb(const uint8_t *in, size_t n, int level, int window_bits, int truncate_heuristic, uint8_t *out, size_t cap,
                       size_t *out_len, uint32_t *crc_out) {
  deflate_t *s = (deflate_t *)calloc(1, sizeof(deflate_t));
  if (!s) return -1;
  s->in = in; s->in_len = n; s->out = out; s->out_cap = cap; s->truncate = truncate_heuristic;
  if (df_init(s, level, window_bits)) {
    /* _deflate(finish): runs when there is input, lookahead, or a pending finish (always, on the first call) */
    switch (s->func) {
      case FN_STORED: deflate_stored(s); break;
      case FN_FAST: deflate_fast(s); break;
      default: deflate_slow(s); break;
    }
  }
  if (out_len) *out_len = s->out_len;
  if (crc_out) *crc_out = s->crc;
  int ovf = s->overflow;
  free(s);
  return ovf ? -1 : 0;
}

/* _GZipEncoder.encodeStream: header `1f 8b 08 00 <mtime LE> 00 ff`, deflate, CRC-32, ISIZE (both LE).
 * The reference stamps the current time; the caller supplies it here. */
int orc_gzip_encode(const uint8_t *in, size_t n, int level, uint32_t mtime, uint8_t *out, size_t cap, size_t *out_len) {
  if (cap < 18) return -1;
  static const uint8_t h[4] = {0x1f, 0x8b, 8, 0};
  memcpy(out, h, 4);
  for (int k = 0; k < 4; k++) out[4 + k] = (uint8_t)(mtime >> (8 * k));
  out[8] = 0; out[9] = 0xff;
  size_t clen = 0;
  uint32_t crc = 0;
  if (orc_deflate_raw(in, n, level, 1, out + 10, cap - 18, &clen, &crc) != 0) return -1;
  uint8_t *t = out + 10 + clen;
  for (int k = 0; k < 4; k++) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)(((uint32_t)n) >> (8 * k)); }
  *out_len = 18 + clen;
  return 0;
}
/* _ZLibEncoder.encodeStream: `78 01` (FLEVEL always 0 -> FCHECK 1), deflate, Adler-32 big-endian */
int orc_zlib_encode(const uint8_t *in, size_t n, int level, uint8_t *out, size_t cap, size_t *out_len) {
  if (cap < 6) return -1;
  out[0] = 0x78; out[1] = 0x01;
  size_t clen = 0;
  if (orc_deflate_raw(in, n, level, 1, out + 2, cap - 6, &clen, NULL) != 0) return -1;
  uint32_t a = orc_adler32(in, n, 1);
  uint8_t *t = out + 2 + clen;
  t[0] = (uint8_t)(a >> 24); t[1] = (uint8_t)(a >> 16); t[2] = (uint8_t)(a >> 8); t[3] = (uint8_t)a;
  *out_len = 6 + clen;
  return 0;
}
