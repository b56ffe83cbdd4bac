/* sanitize_main.c -- test infrastructure: runs the oracle's decoders over files under AddressSanitizer +
 * UndefinedBehaviorSanitizer (`make -C oracle sanitize`; SURVEY.md section 5 "sanitizer build of the oracle").
 *
 *   oracle_san <kind> <file> [<file> ...]     kind: raw | gzip | zlib | bzip2 | deflate
 *
 * Every file is decoded (deflate: encoded at levels 1, 6, 9 and decoded back) with exactly-sized heap buffers, so a
 * read or write one byte outside of them aborts the process; a clean run prints one line per file and exits 0.
 * Damaged inputs are welcome: the verdict does not matter, the memory discipline does. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_inflate_raw(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len, size_t *in_pos);
int orc_gzip_decode(const uint8_t *in, size_t n, int verify, int raw, uint8_t *out, size_t cap, size_t *out_len);
int orc_zlib_decode(const uint8_t *in, size_t n, int verify, int raw, uint8_t *out, size_t cap, size_t *out_len);
int orc_bzip2_decode(const uint8_t *in, size_t n, int verify, uint8_t *out, size_t cap, size_t *out_len);
int orc_deflate_raw(const uint8_t *in, size_t n, int level, int truncate_heuristic, uint8_t *out, size_t cap, size_t *out_len, uint32_t *crc);

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  const char *kind = argv[1];
  for (int a = 2; a < argc; ++a) {
    FILE *f = fopen(argv[a], "rb");
    if (!f) { perror(argv[a]); return 2; }
    fseek(f, 0, SEEK_END);
    size_t n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *in = malloc(n ? n : 1);  /* exactly n bytes: an over-read is caught */
    if (fread(in, 1, n, f) != n) return 2;
    fclose(f);
    size_t cap = 64u << 20, got = 0, pos = 0;
    uint8_t *out = malloc(cap);
    int st = -9;
    if (!strcmp(kind, "raw")) st = orc_inflate_raw(in, n, out, cap, &got, &pos);
    else if (!strcmp(kind, "gzip")) st = orc_gzip_decode(in, n, 0, 0, out, cap, &got);
    else if (!strcmp(kind, "zlib")) st = orc_zlib_decode(in, n, 1, 0, out, cap, &got);
    else if (!strcmp(kind, "bzip2")) st = orc_bzip2_decode(in, n, 1, out, cap, &got);
    else if (!strcmp(kind, "deflate")) {
      st = 0;
      for (int level = 1; level <= 9 && st == 0; level += (level == 1 ? 5 : 3)) {
        size_t clen = 0, back = 0, p2 = 0;
        uint32_t crc = 0;
        uint8_t *c = malloc(n + n / 8 + 1024);
        if (orc_deflate_raw(in, n, level, 1, c, n + n / 8 + 1024, &clen, &crc) != 0) st = 10 + level;
        uint8_t *cc = malloc(clen ? clen : 1);  /* the decoder sees exactly clen bytes */
        memcpy(cc, c, clen);
        if (st == 0 && (orc_inflate_raw(cc, clen, out, cap, &back, &p2) != 0 || back != n || memcmp(out, in, n))) st = 20 + level;
        got = clen;
        free(c); free(cc);
      }
    } else return 2;
    printf("%s %s: status %d, %zu bytes\n", kind, argv[a], st, got);
    free(in); free(out);
    if (!strcmp(kind, "deflate") && st != 0) return 1;
  }
  return 0;
}
