/* exercises pecaller_amd/csrc/host_io.h on the CPU: pgz writes argv[2] bytes of a pattern to argv[1] with argv[3] threads in
   writes of odd sizes; then zreader reads the file back in 4- and 12-byte reads (pecaller's record pattern) and odd sizes.
   With argv[2] == "read" it only reads argv[1] to its end and prints the byte count (a broken stream must end the program). */
#include <stdint.h>
#include "../../pecaller_amd/csrc/host_io.h"
static unsigned char pat (size_t i) { return (unsigned char) ((i * 2654435761u) >> 13 ^ (i >> 7)); }
int main (int argc, char **argv)
{
  if (strcmp (argv[2], "read") == 0)
    {
      zreader r;
      if (zr_open (&r, argv[1])) return 5;
      size_t n = 0, got;
      unsigned char rec[16];
      while ((got = zr_read (&r, rec, (n % 16) == 0 ? 4 : 12)) != 0) n += got;
      zr_close (&r);
      printf ("read %zu\n", n);
      return 0;
    }
  const size_t total = (size_t) atol (argv[2]);
  pgz w;
  if (pgz_open (&w, argv[1], atoi (argv[3]))) return 2;
  char *buf = (char *) malloc ((size_t) 100 << 20);
  size_t done = 0, k = 0;
  while (done < total)
    {
      size_t n = ((k++ % 3) == 0 ? (size_t) 70000123 : (k % 3) == 1 ? 17 : (size_t) 33554432 + 5);
      if (n > total - done) n = total - done;
      for (size_t i = 0; i < n; i++) buf[i] = (char) pat (done + i);
      if (pgz_write (&w, buf, n)) return 3;
      done += n;
    }
  if (pgz_close (&w)) return 4;
  zreader r;
  if (zr_open (&r, argv[1])) return 5;
  size_t pos = 0;
  unsigned char rec[4096];
  for (;;)
    {
      const size_t want = (pos / 16) % 97 == 0 ? 4093 : ((pos % 16) == 0 ? 4 : 12);
      const size_t got = zr_read (&r, rec, want);
      for (size_t i = 0; i < got; i++)
        if (rec[i] != pat (pos + i)) { printf ("mismatch at %zu\n", pos + i); return 6; }
      pos += got;
      if (got < want) break;
    }
  zr_close (&r);
  if (pos != total) { printf ("read %zu of %zu\n", pos, total); return 7; }
  printf ("ok %zu\n", total);
  return 0;
}
