/*
 * host_io.h -- the host programs' gz plumbing (plain C, zlib + pthreads), outside the hot path.
 *
 *   zreader   a gz stream inflated by a thread of its own into a ring of blocks; the consumer copies bytes out of the ring.
 *             The reference inflates every input on its main thread (pemapper.c:626, 2438-2444; pecaller.c:891-907 reads its
 *             pileup streams 4 and 12 bytes at a time with gzread): with one thread per stream the 64 pileup files of a
 *             pecaller run, or the two mate files of a mapper run, inflate beside each other and beside the consumer.
 *   pgz       a gz FILE written as a sequence of gzip members deflated by several threads (like pigz -i): the bytes a reader
 *             inflates are exactly the bytes given, in order; zlib's gzread -- what the reference's readers use
 *             (pecaller.c:891-907) -- reads concatenated members as one stream.  Level = zlib's default, as gzopen "wb", unless the
 *             program sets another (pecaller_hip's rows: level 2, half the deflate time for 1.4x the file) or PEMAP_GZ_LEVEL does.
 *   gzsrc     where the readers' bytes come from: a regular file is mapped and, if it is gzip, inflated by fast_inflate.h (2-3
 *             times zlib's rate; every member's CRC and length checked); a file that is not gzip is handed out as it lies in the
 *             mapping (gzopen reads such files too, pemapper.c:626); anything else -- a pipe, or PEMAP_ZLIB_INFLATE=1 -- goes
 *             through zlib's gzread as before.
 */
#ifndef HOST_IO_H
#define HOST_IO_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <pthread.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include "fast_inflate.h"

/* ---- the source of a reader's bytes */
typedef struct
{
  int mode;                     /* 0 zlib, 1 gzip through fast_inflate, 2 not compressed */
  gzFile f;
  int fd;
  const uint8_t *map;
  size_t map_len, plain_pos;
  fi_state *fi;
  uint8_t *buf, *out;           /* mode 1: FI_WINDOW bytes of room for the window, then the blocks */
  size_t buf_len, block;
  const char *err;
} gzsrc;

static pthread_once_t gzsrc_once = PTHREAD_ONCE_INIT;
static void
gzsrc_global_init (void)
{
  fi_crc32 (0u, (const uint8_t *) "", 0);       /* tables, CPU features */
}

/* 0 = ok, -1 = the file cannot be opened.  `block` = bytes handed out per call at most, `ring` = blocks the consumer may hold. */
__attribute__ ((unused)) static int
gzsrc_open (gzsrc * g, const char *path, size_t block, int ring)
{
  memset (g, 0, sizeof *g);
  g->fd = -1;
  g->block = block < 4096 ? 4096 : block;       /* (a block must hold more than the decoder's overrun, FI_SLACK) */
  block = g->block;
  pthread_once (&gzsrc_once, gzsrc_global_init);
  const char *force = getenv ("PEMAP_ZLIB_INFLATE");
  if (!(force && atoi (force)))
    {
      g->fd = open (path, O_RDONLY);
      if (g->fd < 0)
        return -1;
      struct stat st;
      if (fstat (g->fd, &st) == 0 && S_ISREG (st.st_mode))
        {
          if (st.st_size == 0)
            {
              g->mode = 2;      /* an empty file: no bytes (gzread gives 0) */
              return 0;
            }
          void *m = mmap (NULL, (size_t) st.st_size, PROT_READ, MAP_PRIVATE, g->fd, 0);
          if (m != MAP_FAILED)
            {
              madvise (m, (size_t) st.st_size, MADV_SEQUENTIAL);
              g->map = (const uint8_t *) m;
              g->map_len = (size_t) st.st_size;
              if (g->map_len >= 2 && g->map[0] == 0x1f && g->map[1] == 0x8b)
                {
                  g->mode = 1;
                  g->fi = (fi_state *) malloc (sizeof (fi_state));
                  g->buf_len = FI_WINDOW + (size_t) (2 * ring + 2) * block + FI_SLACK;
                  g->buf = (uint8_t *) malloc (g->buf_len);
                  if (!g->fi || !g->buf)
                    {
                      free (g->fi);
                      free (g->buf);
                      g->fi = NULL;
                      g->buf = NULL;
                      munmap (m, (size_t) st.st_size);
                      g->map = NULL;
                      close (g->fd);
                      g->fd = -1;
                      return -1;
                    }
                  fi_init (g->fi, g->map, g->map_len);
                  g->out = g->buf + FI_WINDOW;
                }
              else
                g->mode = 2;
              return 0;
            }
        }
      /* not a regular file (a pipe, a FIFO) or not mappable: zlib reads the descriptor that is open already -- a second open of a
         FIFO could find the writer gone.  (A mapped regular file that shrinks while it is read raises SIGBUS, not a read error.) */
      g->mode = 0;
      g->f = gzdopen (g->fd, "rb");
      if (!g->f)
        {
          close (g->fd);
          g->fd = -1;
          return -1;
        }
      g->fd = -1;               /* (zlib's now: gzclose closes it) */
      gzbuffer (g->f, 1 << 20);
      return 0;
    }
  g->mode = 0;
  g->f = gzopen (path, "rb");
  if (!g->f)
    return -1;
  gzbuffer (g->f, 1 << 20);
  return 0;
}

/* the next block: 1 = *ptr / *len describe it (mode 0: read into own_buf, `block` bytes of room), 0 = end of data, -1 = the stream
   is damaged or cut short */
__attribute__ ((unused)) static int
gzsrc_next (gzsrc * g, char *own_buf, char **ptr, int *len)
{
  if (g->mode == 0)
    {
      const int got = gzread (g->f, own_buf, (unsigned) g->block);
      if (got <= 0)
        {
          /* (gzread gives 0 after a truncated member and keeps Z_BUF_ERROR) */
          int en = Z_OK;
          gzerror (g->f, &en);
          return (got < 0 || (en != Z_OK && en != Z_STREAM_END)) ? -1 : 0;
        }
      *ptr = own_buf;
      *len = got;
      return 1;
    }
  if (g->mode == 2)
    {
      if (g->plain_pos >= g->map_len)
        return 0;
      size_t m = g->map_len - g->plain_pos;
      if (m > g->block)
        m = g->block;
      *ptr = (char *) (g->map + g->plain_pos);  /* (read-only: the consumers do not write into their blocks) */
      *len = (int) m;
      g->plain_pos += m;
      return 1;
    }
  for (;;)
    {
      if (g->out + g->block + FI_SLACK > g->buf + g->buf_len)
        {
          /* back to the start of the buffer: the window moves with the writer (the blocks the consumer may still hold are the last
             `ring`, all of them further up than the 2 blocks' room the writer has before it waits for the consumer) */
          memmove (g->buf, g->out - FI_WINDOW, FI_WINDOW);      /* (they overlap when the blocks are small) */
          fi_moved (g->fi, g->out, g->buf + FI_WINDOW);
          g->out = g->buf + FI_WINDOW;
        }
      uint8_t *start = g->out;
      const int rc = fi_run (g->fi, &g->out, start + g->block - FI_SLACK);
      /* (what was decoded before the damage is handed out first; the next call reports the damage) */
      if (rc == FI_ERROR && g->out == start)
        {
          g->err = g->fi->msg;
          return -1;
        }
      if (g->out > start)
        {
          *ptr = (char *) start;
          *len = (int) (g->out - start);
          return 1;
        }
      if (rc == FI_END)
        return 0;
    }
}

__attribute__ ((unused)) static void
gzsrc_close (gzsrc * g)
{
  if (g->f)
    gzclose (g->f);
  if (g->map)
    munmap ((void *) g->map, g->map_len);
  if (g->fd >= 0)
    close (g->fd);
  free (g->fi);
  free (g->buf);
  memset (g, 0, sizeof *g);
  g->fd = -1;
}

#define ZR_BLOCK (1 << 18)     /* (64 to 256 of these streams are open at once: 18 blocks of buffer each) */
#define ZR_RING 8
typedef struct
{
  gzsrc src;
  char *own[ZR_RING];           /* the ring's own buffers (zlib mode only) */
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
      /* a short or corrupt stream is an error, not an end */
      char *blk = NULL;
      int got = 0;
      const int rc = gzsrc_next (&r->src, r->own[tail], &blk, &got);
      const int bad = rc < 0;
      if (rc <= 0)
        got = 0;
      pthread_mutex_lock (&r->mu);
      if (bad)
        r->error = 1;
      if (got <= 0)
        r->done = 1;
      else
        {
          r->ring[tail] = blk;
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
  if (gzsrc_open (&r->src, path, ZR_BLOCK, ZR_RING))
    return -1;
  r->path = path;
  if (r->src.mode == 0)
    for (int i = 0; i < ZR_RING; i++)
      r->own[i] = (char *) malloc (ZR_BLOCK);
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
  gzsrc_close (&r->src);
  for (int i = 0; i < ZR_RING; i++)
    free (r->own[i]);
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
  int level;                    /* zlib's levels; Z_DEFAULT_COMPRESSION unless the caller or PEMAP_GZ_LEVEL says otherwise */
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
  int level;
} pgz_job;

static int
pgz_member (const char *src, size_t n, unsigned char **out, size_t *out_len, int level)
{
  z_stream z;
  memset (&z, 0, sizeof z);
  if (deflateInit2 (&z, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK)
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
      if (pgz_member (j->src + off, n, &j->out[k], &j->out_len[k], j->level))
        j->failed = 1;
    }
}

__attribute__ ((unused)) static int
pgz_open (pgz * p, const char *path, int threads)
{
  p->f = fopen (path, "wb");
  p->threads = threads < 1 ? 1 : threads > 64 ? 64 : threads;
  p->wrote_any = 0;
  p->level = Z_DEFAULT_COMPRESSION;
  {
    const char *e = getenv ("PEMAP_GZ_LEVEL");
    if (e && atoi (e) >= 1 && atoi (e) <= 9)
      p->level = atoi (e);
  }
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
  j.level = p->level;
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
      if (pgz_member ("", 0, &o, &n, p->level) || fwrite (o, 1, n, p->f) != n)
        rc = -1;
      free (o);
    }
  if (fclose (p->f))
    rc = -1;
  return rc;
}
#endif
