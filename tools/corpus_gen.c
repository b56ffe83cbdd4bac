/*
 * tools/corpus_gen.c -- synthetic workload generator for tests and bench.py (NOT product code,
 * NOT the oracle).  Produces the inputs BASELINE.md section 3 names:
 *   - config 3/4 "log text": `YYYY-MM-DD hh:mm:ss.mmm [LEVEL] word: 4-14 words\n`, 4096-word
 *     vocabulary, deterministic in (seed, chunk index)
 *   - config 2 "enwik-like" text: Zipf(1.0) over a 50k-word vocabulary with ~8 % XML-ish markup
 *   - multi-member gzip packing of fixed-size chunks (one member per chunk, optional BGZF-style
 *     `BC` FEXTRA subfield carrying the member size, mtime 0), compressed with C zlib -- the
 *     same library the reference's default dart:io path delegates to
 *     (/root/reference/lib/src/codecs/zlib/_gzip_encoder_io.dart:17,31).  Inflate output is
 *     unique for a valid stream, so the compressor choice does not affect decode parity.
 * Threads: chunks are independent; `threads` workers pull chunk indices from a shared counter.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

typedef struct { uint64_t s; } rng_t;
static inline uint64_t rng_next(rng_t *r) { /* splitmix64 */
  uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline uint32_t rng_below(rng_t *r, uint32_t n) { return (uint32_t)((rng_next(r) >> 11) % n); }

/* ---- vocabulary ---- */
typedef struct { char w[16]; uint8_t len; } word_t;
static void make_vocab(word_t *v, int n, uint64_t seed, int minlen, int maxlen) {
  rng_t r = {seed * 0x1234567ull + 99};
  static const char cons[] = "bcdfghjklmnprstvwz", vow[] = "aeiou";
  for (int i = 0; i < n; ++i) {
    int len = minlen + (int)rng_below(&r, (uint32_t)(maxlen - minlen + 1));
    for (int k = 0; k < len; ++k) v[i].w[k] = (k & 1) ? vow[rng_below(&r, 5)] : cons[rng_below(&r, 18)];
    v[i].len = (uint8_t)len;
  }
}

/* ---- log text (configs 3 and 4) ---- */
#define LOG_VOCAB 4096
static word_t g_log_vocab[LOG_VOCAB];
static uint64_t g_log_vocab_seed = ~0ull;
static pthread_mutex_t g_vocab_mu = PTHREAD_MUTEX_INITIALIZER;

static void put2(uint8_t *p, int v) { p[0] = (uint8_t)('0' + v / 10); p[1] = (uint8_t)('0' + v % 10); }

/* Fill out[0..n) with log lines; content depends only on (seed, chunk). */
void corpus_log_text(uint64_t seed, uint64_t chunk, uint8_t *out, size_t n) {
  pthread_mutex_lock(&g_vocab_mu);
  if (g_log_vocab_seed != seed) { make_vocab(g_log_vocab, LOG_VOCAB, seed, 2, 12); g_log_vocab_seed = seed; }
  pthread_mutex_unlock(&g_vocab_mu);
  static const char *levels[5] = {"INFO", "DEBUG", "WARN", "ERROR", "TRACE"};
  rng_t r = {seed ^ (chunk * 0xD1B54A32D192ED03ull + 0x8CB92BA72F3D8DD7ull)};
  uint64_t t = 1700000000000ull + chunk * 977ull * 64ull; /* ms */
  size_t o = 0;
  uint8_t line[512];
  while (o < n) {
    t += 1 + rng_below(&r, 900);
    uint64_t ms = t % 1000, s = t / 1000;
    int sec = (int)(s % 60), min = (int)((s / 60) % 60), hr = (int)((s / 3600) % 24);
    uint64_t day = s / 86400;
    int dd = 1 + (int)(day % 28), mm = 1 + (int)((day / 28) % 12), yy = 2020 + (int)((day / 336) % 10);
    size_t k = 0;
    line[k++] = (uint8_t)('0' + yy / 1000); line[k++] = (uint8_t)('0' + (yy / 100) % 10);
    put2(line + k, yy % 100); k += 2; line[k++] = '-';
    put2(line + k, mm); k += 2; line[k++] = '-';
    put2(line + k, dd); k += 2; line[k++] = ' ';
    put2(line + k, hr); k += 2; line[k++] = ':';
    put2(line + k, min); k += 2; line[k++] = ':';
    put2(line + k, sec); k += 2; line[k++] = '.';
    line[k++] = (uint8_t)('0' + ms / 100); put2(line + k, (int)(ms % 100)); k += 2;
    line[k++] = ' '; line[k++] = '[';
    const char *lv = levels[rng_below(&r, 5)];
    size_t ll = strlen(lv); memcpy(line + k, lv, ll); k += ll;
    line[k++] = ']'; line[k++] = ' ';
    const word_t *w = &g_log_vocab[rng_below(&r, LOG_VOCAB)];
    memcpy(line + k, w->w, w->len); k += w->len; line[k++] = ':';
    int nw = 4 + (int)rng_below(&r, 11);
    for (int i = 0; i < nw; ++i) {
      w = &g_log_vocab[rng_below(&r, LOG_VOCAB)];
      line[k++] = ' '; memcpy(line + k, w->w, w->len); k += w->len;
    }
    line[k++] = '\n';
    size_t c = (k < n - o) ? k : n - o;
    memcpy(out + o, line, c);
    o += c;
  }
}

/* ---- enwik-like text (config 2) ---- */
#define WIKI_VOCAB 50000
static word_t *g_wiki_vocab = NULL;
static double *g_wiki_cdf = NULL;
static uint64_t g_wiki_seed = ~0ull;
void corpus_wiki_text(uint64_t seed, uint64_t chunk, uint8_t *out, size_t n) {
  pthread_mutex_lock(&g_vocab_mu);
  if (g_wiki_seed != seed) {
    if (!g_wiki_vocab) { g_wiki_vocab = malloc(sizeof(word_t) * WIKI_VOCAB); g_wiki_cdf = malloc(sizeof(double) * WIKI_VOCAB); }
    make_vocab(g_wiki_vocab, WIKI_VOCAB, seed + 7, 2, 12);
    double acc = 0;
    for (int i = 0; i < WIKI_VOCAB; ++i) { acc += 1.0 / (double)(i + 1); g_wiki_cdf[i] = acc; }
    for (int i = 0; i < WIKI_VOCAB; ++i) g_wiki_cdf[i] /= acc;
    g_wiki_seed = seed;
  }
  pthread_mutex_unlock(&g_vocab_mu);
  static const char *tags[8] = {"<page>", "</page>", "<title>", "</title>", "<text xml:space=\"preserve\">", "</text>", "&quot;", "&amp;"};
  rng_t r = {seed ^ (chunk * 0xA24BAED4963EE407ull + 0x9FB21C651E98DF25ull)};
  size_t o = 0, col = 0, wrap = 60 + rng_below(&r, 61);
  int cap_next = 1;
  while (o < n) {
    const char *src; size_t len; char tmp[20];
    if (rng_below(&r, 100) < 8) { src = tags[rng_below(&r, 8)]; len = strlen(src); }
    else {
      double u = (double)(rng_next(&r) >> 11) * (1.0 / 9007199254740992.0);
      int lo = 0, hi = WIKI_VOCAB - 1;
      while (lo < hi) { int mid = (lo + hi) >> 1; if (g_wiki_cdf[mid] < u) lo = mid + 1; else hi = mid; }
      const word_t *w = &g_wiki_vocab[lo];
      memcpy(tmp, w->w, w->len); len = w->len;
      if (cap_next) { tmp[0] = (char)(tmp[0] - 32); cap_next = 0; }
      if (rng_below(&r, 12) == 0) { tmp[len++] = '.'; cap_next = 1; }
      else if (rng_below(&r, 10) == 0) tmp[len++] = ',';
      src = tmp;
    }
    for (size_t k = 0; k < len && o < n; ++k) out[o++] = (uint8_t)src[k];
    col += len + 1;
    if (o < n) { if (col >= wrap) { out[o++] = '\n'; col = 0; wrap = 60 + rng_below(&r, 61); } else out[o++] = ' '; }
  }
}

/* ---- gzip member packing ---- */
/* One gzip member around `src`.  Returns member size, 0 if `cap` is too small.
 * bc != 0 adds FEXTRA {'B','C',2,BSIZE} when the member fits in 64 KiB (BGZF convention). */
size_t corpus_gzip_member(const uint8_t *src, size_t n, int level, int bc, uint8_t *dst, size_t cap) {
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return 0;
  size_t hdr = bc ? 18 : 10;
  if (cap < hdr + 8) { deflateEnd(&zs); return 0; }
  zs.next_in = (Bytef *)src; zs.avail_in = (uInt)n;
  zs.next_out = dst + hdr; zs.avail_out = (uInt)(cap - hdr - 8);
  int rc = deflate(&zs, Z_FINISH);
  size_t clen = zs.total_out;
  deflateEnd(&zs);
  if (rc != Z_STREAM_END) return 0;
  size_t total = hdr + clen + 8;
  if (bc && total > 65536) { /* does not fit BSIZE: re-emit without the subfield */
    memmove(dst + 10, dst + 18, clen);
    hdr = 10; bc = 0; total = hdr + clen + 8;
  }
  static const uint8_t h10[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
  memcpy(dst, h10, 10);
  if (bc) {
    dst[3] = 4;
    dst[10] = 6; dst[11] = 0; dst[12] = 'B'; dst[13] = 'C'; dst[14] = 2; dst[15] = 0;
    dst[16] = (uint8_t)((total - 1) & 0xff); dst[17] = (uint8_t)((total - 1) >> 8);
  }
  uint32_t crc = (uint32_t)crc32(0L, src, (uInt)n);
  uint8_t *t = dst + hdr + clen;
  for (int k = 0; k < 4; ++k) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)(((uint32_t)n) >> (8 * k)); }
  return total;
}

typedef struct {
  int kind; uint64_t seed; uint64_t first_chunk, n_chunks; size_t chunk_bytes; int level, bc;
  uint8_t *slots; size_t slot_cap; size_t *sizes; uint8_t *plain; /* optional: uncompressed copy */
  volatile uint64_t next;
} job_t;

static void *worker(void *arg) {
  job_t *j = (job_t *)arg;
  uint8_t *buf = malloc(j->chunk_bytes);
  for (;;) {
    uint64_t i = __sync_fetch_and_add(&j->next, 1);
    if (i >= j->n_chunks) break;
    uint8_t *text = j->plain ? j->plain + i * j->chunk_bytes : buf;
    if (j->kind == 0) corpus_log_text(j->seed, j->first_chunk + i, text, j->chunk_bytes);
    else corpus_wiki_text(j->seed, j->first_chunk + i, text, j->chunk_bytes);
    j->sizes[i] = corpus_gzip_member(text, j->chunk_bytes, j->level, j->bc, j->slots + i * j->slot_cap, j->slot_cap);
  }
  free(buf);
  return NULL;
}

/* Build `n_chunks` members of `chunk_bytes` each (chunk indices first_chunk..), concatenated
 * into dst.  kind 0 = log text, 1 = wiki text.  plain (may be NULL) receives the uncompressed
 * bytes.  Returns total compressed size, 0 on failure (dst too small / zlib error). */
size_t corpus_make_gzip(int kind, uint64_t seed, uint64_t first_chunk, uint64_t n_chunks, size_t chunk_bytes,
                        int level, int bc, int threads, uint8_t *dst, size_t dst_cap, uint8_t *plain) {
  size_t slot_cap = chunk_bytes + chunk_bytes / 8 + 256;
  uint8_t *slots = malloc(slot_cap * n_chunks);
  size_t *sizes = calloc(n_chunks, sizeof(size_t));
  if (!slots || !sizes) { free(slots); free(sizes); return 0; }
  job_t j = {kind, seed, first_chunk, n_chunks, chunk_bytes, level, bc, slots, slot_cap, sizes, plain, 0};
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  pthread_t th[256];
  for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, worker, &j);
  for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  size_t total = 0;
  int ok = 1;
  for (uint64_t i = 0; i < n_chunks; ++i) {
    if (!sizes[i] || total + sizes[i] > dst_cap) { ok = 0; break; }
    memcpy(dst + total, slots + i * slot_cap, sizes[i]);
    total += sizes[i];
  }
  free(slots); free(sizes);
  return ok ? total : 0;
}
