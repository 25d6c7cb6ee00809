/*
 * host_io.h -- the host programs' gz plumbing (plain C, zlib + pthreads), outside the hot path.
 *
 *   zreader   a gz stream inflated by a thread of its own into a ring of blocks; the consumer copies bytes out of the ring.
 *             The reference inflates every input on its main thread (pemapper.c:626, 2438-2444; pecaller.c:891-907 reads its
 *             pileup streams 4 and 12 bytes at a time with gzread): with one thread per stream the 64 pileup files of a
 *             pecaller run, or the two mate files of a mapper run, inflate beside each other and beside the consumer.
 *   pgz       a gz FILE written as a sequence of gzip members deflated by several threads (like pigz -i): the bytes a reader
 *             inflates are exactly the bytes given, in order; zlib's gzread -- what the reference's readers use
 *             (pecaller.c:891-907) -- reads concatenated members as one stream.  Level = zlib's default, as gzopen "wb".
 */
#ifndef HOST_IO_H
#define HOST_IO_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <pthread.h>

#define ZR_BLOCK (1 << 20)
#define ZR_RING 8
typedef struct
{
  gzFile f;
  pthread_t th;
  pthread_mutex_t mu;
  pthread_cond_t cv;
  char *ring[ZR_RING];
  int ring_len[ZR_RING];
  int head, count, done, stop, error;     /* shared: only under mu */
  int pos;                      /* consumer's own: bytes of ring[head] already consumed */
  int cur_len;                  /* consumer's own: length of ring[head] once it holds the block (0 = no block held) */
  const char *path;
} zreader;

static void *
zr_inflate (void *arg)
{
  zreader *r = (zreader *) arg;
  int tail = 0;
  for (;;)
    {
      pthread_mutex_lock (&r->mu);
      while (r->count == ZR_RING && !r->stop)
        pthread_cond_wait (&r->cv, &r->mu);
      const int stop = r->stop;
      if (stop)
        {
          r->done = 1;
          pthread_cond_broadcast (&r->cv);
        }
      pthread_mutex_unlock (&r->mu);
      if (stop)
        return NULL;
      int got = gzread (r->f, r->ring[tail], ZR_BLOCK);
      int bad = 0;
      if (got <= 0)
        {
          /* a short or corrupt stream is an error, not an end (gzread gives 0 after a truncated member and keeps Z_BUF_ERROR) */
          int en = Z_OK;
          gzerror (r->f, &en);
          bad = got < 0 || (en != Z_OK && en != Z_STREAM_END);
        }
      pthread_mutex_lock (&r->mu);
      if (bad)
        r->error = 1;
      if (got <= 0)
        r->done = 1;
      else
        {
          r->ring_len[tail] = got;
          r->count++;
        }
      pthread_cond_broadcast (&r->cv);
      pthread_mutex_unlock (&r->mu);
      if (got <= 0)
        return NULL;
      tail = (tail + 1) % ZR_RING;
    }
}

/* 0 = ok, -1 = the file cannot be opened */
__attribute__ ((unused)) static int
zr_open (zreader * r, const char *path)
{
  memset (r, 0, sizeof *r);
  r->f = gzopen (path, "rb");
  if (!r->f)
    return -1;
  r->path = path;
  gzbuffer (r->f, 1 << 20);
  for (int i = 0; i < ZR_RING; i++)
    r->ring[i] = (char *) malloc (ZR_BLOCK);
  pthread_mutex_init (&r->mu, NULL);
  pthread_cond_init (&r->cv, NULL);
  if (pthread_create (&r->th, NULL, zr_inflate, r))
    return -1;
  return 0;
}

/* Let go of the block that has been read to its end and take hold of the next one: afterwards cur_len/pos describe ring[head], or
 * cur_len is 0 at the end of the stream.  The ring's shared fields are touched under the lock only; between two calls the consumer
 * works on cur_len and pos, which are its own.  A stream that ended in an inflate error ends the program: calling on the part
 * that could be read would look like a result. */
__attribute__ ((unused)) static void
zr_next_block (zreader * r)
{
  pthread_mutex_lock (&r->mu);
  if (r->cur_len > 0)
    {
      r->head = (r->head + 1) % ZR_RING;
      r->count--;
      pthread_cond_broadcast (&r->cv);
    }
  while (r->count == 0 && !r->done)
    pthread_cond_wait (&r->cv, &r->mu);
  r->cur_len = r->count > 0 ? r->ring_len[r->head] : 0;
  const int failed = r->count == 0 && r->error;
  pthread_mutex_unlock (&r->mu);
  r->pos = 0;
  if (failed)
    {
      fprintf (stderr, "\n %s is truncated or not a valid gzip stream \n", r->path ? r->path : "input");
      exit (1);
    }
}

/* copy up to n bytes to dst; returns the bytes copied (less than n only at the end of the stream) */
__attribute__ ((unused)) static size_t
zr_read (zreader * r, void *dst, size_t n)
{
  size_t got = 0;
  while (got < n)
    {
      if (r->pos == r->cur_len)
        {
          zr_next_block (r);
          if (r->cur_len == 0)
            break;
        }
      size_t m = (size_t) (r->cur_len - r->pos);
      if (m > n - got)
        m = n - got;
      memcpy ((char *) dst + got, r->ring[r->head] + r->pos, m);
      r->pos += (int) m;
      got += m;
    }
  return got;
}

__attribute__ ((unused)) static void
zr_close (zreader * r)
{
  pthread_mutex_lock (&r->mu);
  r->stop = 1;
  pthread_cond_broadcast (&r->cv);
  pthread_mutex_unlock (&r->mu);
  pthread_join (r->th, NULL);
  gzclose (r->f);
  for (int i = 0; i < ZR_RING; i++)
    free (r->ring[i]);
  pthread_mutex_destroy (&r->mu);
  pthread_cond_destroy (&r->cv);
}

/* ---- parallel gz writer */
#define PGZ_PIECE ((size_t) 32 << 20)
typedef struct
{
  FILE *f;
  int threads;
  int wrote_any;
} pgz;

typedef struct
{
  const char *src;
  size_t n_pieces, bytes;
  unsigned char **out;
  size_t *out_len;
  int next;
  pthread_mutex_t mu;
  int failed;
} pgz_job;

static int
pgz_member (const char *src, size_t n, unsigned char **out, size_t *out_len)
{
  z_stream z;
  memset (&z, 0, sizeof z);
  if (deflateInit2 (&z, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK)
    return -1;
  const size_t cap = deflateBound (&z, (uLong) n) + 64;
  unsigned char *o = (unsigned char *) malloc (cap);
  z.next_in = (Bytef *) src;
  z.avail_in = (uInt) n;
  z.next_out = o;
  z.avail_out = (uInt) cap;
  const int rc = deflate (&z, Z_FINISH);
  *out_len = cap - z.avail_out;
  deflateEnd (&z);
  if (rc != Z_STREAM_END)
    {
      free (o);
      return -1;
    }
  *out = o;
  return 0;
}

static void *
pgz_worker (void *arg)
{
  pgz_job *j = (pgz_job *) arg;
  for (;;)
    {
      pthread_mutex_lock (&j->mu);
      const int k = j->next++;
      pthread_mutex_unlock (&j->mu);
      if ((size_t) k >= j->n_pieces)
        return NULL;
      const size_t off = (size_t) k * PGZ_PIECE;
      const size_t n = j->bytes - off < PGZ_PIECE ? j->bytes - off : PGZ_PIECE;
      if (pgz_member (j->src + off, n, &j->out[k], &j->out_len[k]))
        j->failed = 1;
    }
}

__attribute__ ((unused)) static int
pgz_open (pgz * p, const char *path, int threads)
{
  p->f = fopen (path, "wb");
  p->threads = threads < 1 ? 1 : threads > 64 ? 64 : threads;
  p->wrote_any = 0;
  return p->f ? 0 : -1;
}

/* append `bytes` bytes: cut into pieces, each piece one gzip member, the members written in order */
__attribute__ ((unused)) static int
pgz_write (pgz * p, const void *buf, size_t bytes)
{
  if (bytes == 0)
    return 0;
  pgz_job j;
  memset (&j, 0, sizeof j);
  j.src = (const char *) buf;
  j.bytes = bytes;
  j.n_pieces = (bytes + PGZ_PIECE - 1) / PGZ_PIECE;
  j.out = (unsigned char **) calloc (j.n_pieces, sizeof (unsigned char *));
  j.out_len = (size_t *) calloc (j.n_pieces, sizeof (size_t));
  pthread_mutex_init (&j.mu, NULL);
  int nt = p->threads;
  if ((size_t) nt > j.n_pieces)
    nt = (int) j.n_pieces;
  pthread_t th[64];
  int started = 0;              /* the pieces are handed out by a counter: fewer workers than asked for still do all of them */
  for (int t = 1; t < nt; t++)
    if (pthread_create (&th[started], NULL, pgz_worker, &j) == 0)
      started++;
  pgz_worker (&j);
  for (int t = 0; t < started; t++)
    pthread_join (th[t], NULL);
  pthread_mutex_destroy (&j.mu);
  int rc = j.failed ? -1 : 0;
  for (size_t k = 0; k < j.n_pieces; k++)
    {
      if (!rc && fwrite (j.out[k], 1, j.out_len[k], p->f) != j.out_len[k])
        rc = -1;
      free (j.out[k]);
    }
  free (j.out);
  free (j.out_len);
  p->wrote_any = 1;
  return rc;
}

__attribute__ ((unused)) static int
pgz_close (pgz * p)
{
  int rc = 0;
  if (!p->wrote_any)
    {
      /* an empty gz stream, as gzopen + gzclose leave behind */
      unsigned char *o = NULL;
      size_t n = 0;
      if (pgz_member ("", 0, &o, &n) || fwrite (o, 1, n, p->f) != n)
        rc = -1;
      free (o);
    }
  if (fclose (p->f))
    rc = -1;
  return rc;
}
#endif
