/* exercises pecaller_amd/csrc/fast_inflate.h and the gzsrc layer of host_io.h on the CPU
     fast_inflate_check cat <file> <out> [block]   the file through gzsrc (as the readers use it), bytes written to <out>; exit 3 = damaged
     fast_inflate_check crc                          fi_crc32 against zlib's crc32 on many lengths, offsets and split points
     fast_inflate_check time <file>                  MB/s of the decoder and of zlib on the file */
#include <stdint.h>
#include <time.h>
#include "../../pecaller_amd/csrc/host_io.h"

static double now (void) { struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec; }

int main (int argc, char **argv)
{
  if (argc >= 2 && strcmp (argv[1], "crc") == 0)
    {
      uint8_t *b = (uint8_t *) malloc (1 << 20);
      uint32_t x = 12345;
      for (int i = 0; i < (1 << 20); i++) { x = x * 1664525u + 1013904223u; b[i] = (uint8_t) (x >> 24); }
      long checked = 0;
      for (int len = 0; len < 700; len++)
        for (int off = 0; off < 5; off++)
          {
            if (fi_crc32 (0u, b + off, (size_t) len) != (uint32_t) crc32 (0L, b + off, (uInt) len)) { printf ("crc mismatch len %d off %d\n", len, off); return 1; }
            checked++;
          }
      for (int k = 0; k < 400; k++)
        {
          x = x * 1664525u + 1013904223u;
          const size_t len = (x >> 8) % ((1 << 20) - 8), off = k % 7, cut = len ? (x >> 3) % len : 0;
          uint32_t a = fi_crc32 (fi_crc32 (0u, b + off, cut), b + off + cut, len - cut);
          if (a != (uint32_t) crc32 (0L, b + off, (uInt) len)) { printf ("crc mismatch len %zu cut %zu\n", len, cut); return 1; }
          checked++;
        }
      printf ("crc ok %ld\n", checked);
      return 0;
    }
  if (argc >= 3 && strcmp (argv[1], "time") == 0)
    {
      gzsrc g;
      for (int pass = 0; pass < 2; pass++)
        {
          if (pass) setenv ("PEMAP_ZLIB_INFLATE", "1", 1);
          if (gzsrc_open (&g, argv[2], 1 << 20, 8)) return 2;
          char *own = (char *) malloc (1 << 20), *p; int n; size_t tot = 0; int rc;
          const double t0 = now ();
          while ((rc = gzsrc_next (&g, own, &p, &n)) == 1) tot += (size_t) n;
          const double dt = now () - t0;
          printf ("%s: mode %d, %zu bytes out in %.3f s = %.0f MB/s%s\n", pass ? "zlib" : "fast_inflate", g.mode, tot, dt, (double) tot / dt / 1e6, rc < 0 ? " (ERROR)" : "");
          gzsrc_close (&g);
          free (own);
        }
      return 0;
    }
  if (argc >= 4 && strcmp (argv[1], "cat") == 0)
    {
      const size_t block = argc > 4 ? (size_t) atol (argv[4]) : (size_t) 1 << 16;
      gzsrc g;
      if (gzsrc_open (&g, argv[2], block, 3)) return 2;
      FILE *o = fopen (argv[3], "wb");
      char *own = (char *) malloc (block), *p; int n, rc;
      /* the consumer may hold 3 blocks: keep the last 3 and check they were not overwritten by the time the 4th arrives */
      char *held[3] = { NULL, NULL, NULL }; int held_n[3] = { 0, 0, 0 }; uint32_t held_crc[3] = { 0, 0, 0 }; int h = 0;
      while ((rc = gzsrc_next (&g, own, &p, &n)) == 1)
        {
          if ((size_t) n > block) { printf ("block of %d bytes\n", n); return 4; }
          if (held[h] && g.mode != 0 && fi_crc32 (0u, (const uint8_t *) held[h], (size_t) held_n[h]) != held_crc[h]) { printf ("a held block changed\n"); return 5; }
          held[h] = p; held_n[h] = n; held_crc[h] = fi_crc32 (0u, (const uint8_t *) p, (size_t) n); h = (h + 1) % 3;
          fwrite (p, 1, (size_t) n, o);
        }
      fclose (o);
      printf ("mode %d rc %d%s%s\n", g.mode, rc, g.err ? " : " : "", g.err ? g.err : "");
      gzsrc_close (&g);
      return rc < 0 ? 3 : 0;
    }
  return 1;
}
