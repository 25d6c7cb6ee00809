/*
 * fast_inflate.h -- a gzip (RFC 1952 / 1951) decoder for the host programs' input streams: the whole compressed file in memory
 * (mmap), the output produced in pieces into one contiguous buffer, CRC-32 and length of every member checked.
 *
 * Why: both host programs are bound by inflate once the device side is fast (the reference inflates on its
 * main thread, pemapper.c:626 and 2438-2444, pecaller.c:891-907).  zlib 1.2.11 inflates fastq text at ~260 MB/s per thread here;
 * a file pair is two such streams.  This decoder keeps 56+ bits in a 64-bit buffer (one refill per length/distance pair), looks
 * codes up in one 2048-entry table (longer codes: a second look-up), copies matches 8 bytes at a time, and takes the CRC with
 * carry-less multiplies where the CPU has them.  The bytes it produces are zlib's (tests/test_host_io.py compares them, and feeds it
 * damaged streams under the address sanitizer).
 *
 * Use:   fi_state s;  fi_init (&s, in, in_len);
 *        while ((rc = fi_run (&s, &out, soft_end)) == FI_MORE) { ...hand [old out, out) over; pick the next output range... }
 *   - output before `out` must stay in place and unmodified for 32 KiB back (the match window); a caller that wraps its buffer
 *     copies the last 32 KiB in front of the new position and calls fi_moved (&s, old_out, new_out);
 *   - fi_run stops at the first symbol boundary at or after soft_end: at most FI_SLACK bytes beyond it are written;
 *   - FI_END: every member decoded and verified, input used up (bytes after the last member that are not a gzip header are ignored,
 *     as zlib's gzread ignores them);  FI_ERROR: the stream is damaged or cut short (s.msg says where).
 */
#ifndef FAST_INFLATE_H
#define FAST_INFLATE_H
#include <stdint.h>
#include <string.h>
#include <stddef.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#define FI_MORE   0
#define FI_END    1
#define FI_ERROR (-1)
#define FI_SLACK  320           /* bytes fi_run may write beyond soft_end (a 258-byte match + the copy's granule) */
#define FI_WINDOW 32768

#define FI_LBITS 11             /* primary bits of the literal/length table */
#define FI_DBITS 8              /* ... of the distance table */
#define FI_LSIZE ((1 << FI_LBITS) + 288 * 16)
#define FI_DSIZE ((1 << FI_DBITS) + 32 * 128)

/* table entry: bits 0-5 bits to take from the stream (code + extra), 8-11 code length (or, for a link, the sub-table's index bits),
   12-15 kind, 16-31 value (literal, length base, distance base, or a link's offset).  0 = no such code. */
#define FI_E_LIT 0x8000u
#define FI_E_EOB 0x4000u
#define FI_E_SUB 0x2000u

typedef struct
{
  const uint8_t *in, *in_end;
  uint64_t bb;                  /* bit buffer: the next stream bits from bit 0 up; bits above bc may already hold what follows */
  int bc;                       /* valid bits in bb */
  int phase;                    /* 0 member header, 1 block header, 2 inside a Huffman block, 3 inside a stored block, 5 member trailer, 4 done */
  int last_block;
  uint32_t stored_left;
  uint32_t crc;                 /* running CRC-32 of the member (zlib's convention: 0 for no data) */
  uint64_t member_out;          /* bytes of the member so far */
  uint64_t members;
  const uint8_t *floor;         /* lowest address a match may start at (the member's first byte, or 32 KiB back) */
  const char *msg;
  uint32_t lt[FI_LSIZE];
  uint32_t dt[FI_DSIZE];
} fi_state;

/* ---------------------------------------------------------------------------------------------------- CRC-32 */
static uint32_t fi_crc_tab[8][256];
static int fi_crc_ready;

static void
fi_crc_init (void)
{
  for (uint32_t i = 0; i < 256; i++)
    {
      uint32_t c = i;
      for (int k = 0; k < 8; k++)
        c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
      fi_crc_tab[0][i] = c;
    }
  for (uint32_t i = 0; i < 256; i++)
    for (int t = 1; t < 8; t++)
      fi_crc_tab[t][i] = (fi_crc_tab[t - 1][i] >> 8) ^ fi_crc_tab[0][fi_crc_tab[t - 1][i] & 0xFF];
  __atomic_store_n (&fi_crc_ready, 1, __ATOMIC_RELEASE);
}

/* register in, register out (no inversion): eight bytes per step */
static uint32_t
fi_crc_bytes (uint32_t r, const uint8_t * p, size_t n)
{
  while (n >= 8)
    {
      uint64_t v;
      memcpy (&v, p, 8);
      v ^= r;
      r = fi_crc_tab[7][v & 0xFF] ^ fi_crc_tab[6][(v >> 8) & 0xFF] ^ fi_crc_tab[5][(v >> 16) & 0xFF] ^ fi_crc_tab[4][(v >> 24) & 0xFF]
        ^ fi_crc_tab[3][(v >> 32) & 0xFF] ^ fi_crc_tab[2][(v >> 40) & 0xFF] ^ fi_crc_tab[1][(v >> 48) & 0xFF] ^ fi_crc_tab[0][v >> 56];
      p += 8;
      n -= 8;
    }
  while (n--)
    r = (r >> 8) ^ fi_crc_tab[0][(r ^ *p++) & 0xFF];
  return r;
}

#if defined(__x86_64__)
/* Folding with carry-less multiplies.  A 128-bit lane A that stands D bits ahead of the data it is xor-ed into is replaced by
   A.lo * k(D + 32) ^ A.hi * k(D - 32), k(n) = bit-reflected (x^n mod P) << 1 (P = the gzip polynomial): D = 512 for the four
   lanes of the main loop (k = 0x154442bd4, 0x1c6e41596), D = 128 when lanes are folded into one (0x1751997d0, 0x0ccaa009e).  The
   lane that is left stands for the whole message so far: the byte-wise CRC of its 16 bytes, from register 0, is the register. */
__attribute__ ((target ("pclmul,sse4.1")))
static uint32_t
fi_crc_clmul (uint32_t r, const uint8_t * p, size_t n)
{
  if (n < 128)
    return fi_crc_bytes (r, p, n);
  const __m128i k512 = _mm_set_epi64x (0x1c6e41596LL, 0x154442bd4LL), k128 = _mm_set_epi64x (0x0ccaa009eLL, 0x1751997d0LL);
  __m128i x0 = _mm_loadu_si128 ((const __m128i *) p), x1 = _mm_loadu_si128 ((const __m128i *) (p + 16));
  __m128i x2 = _mm_loadu_si128 ((const __m128i *) (p + 32)), x3 = _mm_loadu_si128 ((const __m128i *) (p + 48));
  x0 = _mm_xor_si128 (x0, _mm_cvtsi32_si128 ((int) r));
  p += 64;
  n -= 64;
#define FI_FOLD(x, k, d) _mm_xor_si128 (_mm_xor_si128 (_mm_clmulepi64_si128 (x, k, 0x00), _mm_clmulepi64_si128 (x, k, 0x11)), d)
  while (n >= 64)
    {
      x0 = FI_FOLD (x0, k512, _mm_loadu_si128 ((const __m128i *) p));
      x1 = FI_FOLD (x1, k512, _mm_loadu_si128 ((const __m128i *) (p + 16)));
      x2 = FI_FOLD (x2, k512, _mm_loadu_si128 ((const __m128i *) (p + 32)));
      x3 = FI_FOLD (x3, k512, _mm_loadu_si128 ((const __m128i *) (p + 48)));
      p += 64;
      n -= 64;
    }
  x1 = FI_FOLD (x0, k128, x1);
  x2 = FI_FOLD (x1, k128, x2);
  x3 = FI_FOLD (x2, k128, x3);
  while (n >= 16)
    {
      x3 = FI_FOLD (x3, k128, _mm_loadu_si128 ((const __m128i *) p));
      p += 16;
      n -= 16;
    }
#undef FI_FOLD
  uint8_t lane[16];
  _mm_storeu_si128 ((__m128i *) lane, x3);
  return fi_crc_bytes (fi_crc_bytes (0u, lane, 16), p, n);
}
#endif

/* zlib's crc32 (): crc of nothing is 0, feed the previous value to go on */
static uint32_t
fi_crc32 (uint32_t crc, const uint8_t * p, size_t n)
{
  if (!__atomic_load_n (&fi_crc_ready, __ATOMIC_ACQUIRE))
    fi_crc_init ();
#if defined(__x86_64__)
  static int have = -1;
  if (have < 0)
    have = __builtin_cpu_supports ("pclmul") && __builtin_cpu_supports ("sse4.1");
  if (have)
    return ~fi_crc_clmul (~crc, p, n);
#endif
  return ~fi_crc_bytes (~crc, p, n);
}

/* ---------------------------------------------------------------------------------------------------- tables */
static const uint16_t fi_len_base[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
static const uint8_t fi_len_extra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
static const uint16_t fi_dist_base[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289,
  16385, 24577
};
static const uint8_t fi_dist_extra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

/* what a code of `len` bits for symbol `sym` stands for: kind 0 = literal/length alphabet, 1 = distances, 2 = code lengths */
static inline uint32_t
fi_entry (int kind, int sym, int len)
{
  if (kind == 0)
    {
      if (sym < 256)
        return ((uint32_t) sym << 16) | FI_E_LIT | ((uint32_t) len << 8) | (uint32_t) len;
      if (sym == 256)
        return FI_E_EOB | ((uint32_t) len << 8) | (uint32_t) len;
      if (sym > 285)
        return 0u;
      return ((uint32_t) fi_len_base[sym - 257] << 16) | ((uint32_t) len << 8) | (uint32_t) (len + fi_len_extra[sym - 257]);
    }
  if (kind == 1)
    {
      if (sym > 29)
        return 0u;
      return ((uint32_t) fi_dist_base[sym] << 16) | ((uint32_t) len << 8) | (uint32_t) (len + fi_dist_extra[sym]);
    }
  return ((uint32_t) sym << 16) | ((uint32_t) len << 8) | (uint32_t) len;
}

/* canonical Huffman code of lens[0..n) -> table with `pbits` primary bits.  0 ok; -1 over-subscribed, or incomplete with more than
   one code (what zlib refuses too); cap = entries the table holds */
static int
fi_build (uint32_t * tab, int cap, int pbits, const uint8_t * lens, int n, int kind)
{
  int count[16] = { 0 };
  for (int i = 0; i < n; i++)
    count[lens[i]]++;
  const int used = n - count[0];
  memset (tab, 0, sizeof (uint32_t) << pbits);  /* (the sub-tables are cleared as they are laid out) */
  if (used == 0)
    return 0;                   /* no codes: every look-up is an error (a block of literals only needs no distance code) */
  int left = 1;
  for (int l = 1; l <= 15; l++)
    {
      left = (left << 1) - count[l];
      if (left < 0)
        return -1;
    }
  /* (an incomplete code: only the single one-bit code, and not for the code lengths' own code -- zlib's rule) */
  if (left > 0 && (kind == 2 || used != 1 || count[1] != 1))
    return -1;
  uint16_t next[16];
  {
    unsigned code = 0;
    count[0] = 0;
    for (int l = 1; l <= 15; l++)
      {
        code = (code + (unsigned) count[l - 1]) << 1;
        next[l] = (uint16_t) code;
      }
  }
  const int psize = 1 << pbits;
  /* sub-tables: the longest code behind every primary index decides its sub-table's size */
  uint8_t sub_bits[1 << FI_LBITS];
  uint16_t codes[288];
  int any_long = 0;
  for (int i = 0; i < n; i++)
    {
      const int l = lens[i];
      if (!l)
        continue;
      unsigned c = next[l]++, rev = 0;
      for (int b = 0; b < l; b++)
        rev |= ((c >> b) & 1u) << (l - 1 - b);
      codes[i] = (uint16_t) rev;
      if (l > pbits)
        {
          if (!any_long)
            memset (sub_bits, 0, (size_t) psize);
          any_long = 1;
          const unsigned pi = rev & (unsigned) (psize - 1);
          if (l - pbits > sub_bits[pi])
            sub_bits[pi] = (uint8_t) (l - pbits);
        }
    }
  int top = psize;
  if (any_long)
    for (int pi = 0; pi < psize; pi++)
      if (sub_bits[pi])
        {
          if (top + (1 << sub_bits[pi]) > cap)
            return -1;
          tab[pi] = ((uint32_t) top << 16) | FI_E_SUB | ((uint32_t) sub_bits[pi] << 8) | (uint32_t) pbits;
          memset (tab + top, 0, sizeof (uint32_t) << sub_bits[pi]);
          top += 1 << sub_bits[pi];
        }
  for (int i = 0; i < n; i++)
    {
      const int l = lens[i];
      if (!l)
        continue;
      const uint32_t e = fi_entry (kind, i, l);
      const unsigned rev = codes[i];
      if (l <= pbits)
        for (unsigned k = rev; k < (unsigned) psize; k += 1u << l)
          tab[k] = e;
      else
        {
          const unsigned pi = rev & (unsigned) (psize - 1);
          const uint32_t link = tab[pi];
          const unsigned base = link >> 16, sb = (link >> 8) & 15u;
          for (unsigned k = rev >> pbits; k < (1u << sb); k += 1u << (l - pbits))
            tab[base + k] = e;
        }
    }
  return 0;
}

/* ---------------------------------------------------------------------------------------------------- the decoder */
static void
fi_init (fi_state * s, const uint8_t * in, size_t in_len)
{
  s->in = in;
  s->in_end = in + in_len;
  s->bb = 0;
  s->bc = 0;
  s->phase = 0;
  s->last_block = 0;
  s->stored_left = 0;
  s->crc = 0;
  s->member_out = 0;
  s->members = 0;
  s->floor = NULL;
  s->msg = NULL;
}

/* the caller moved the window: what was at old_out is now at new_out */
static inline void
fi_moved (fi_state * s, const uint8_t * old_out, uint8_t * new_out)
{
  if (s->floor)
    {
      size_t h = (size_t) (old_out - s->floor);
      if (h > FI_WINDOW)
        h = FI_WINDOW;
      s->floor = new_out - h;
    }
}

#define FI_FAIL(text) do { s->msg = (text); s->phase = 4; s->in = in; s->bb = bb; s->bc = bc; *outp = out; return FI_ERROR; } while (0)
/* at least 56 valid bits, or everything the input still has */
#define FI_REFILL() do { \
    if (__builtin_expect (in_end - in >= 8, 1)) { uint64_t v_; memcpy (&v_, in, 8); bb |= v_ << bc; in += (63 - bc) >> 3; bc |= 56; } \
    else while (bc < 56 && in < in_end) { bb |= (uint64_t) *in++ << bc; bc += 8; } } while (0)
#define FI_NEED(nbits, text) do { if (bc < (nbits)) { FI_REFILL (); if (bc < (nbits)) FI_FAIL (text); } } while (0)
#define FI_DROP(nbits) do { bb >>= (nbits); bc -= (nbits); } while (0)
/* the buffered whole bytes go back to the input (after the bits up to a byte boundary were dropped) */
#define FI_UNREAD() do { FI_DROP (bc & 7); in -= bc >> 3; bb = 0; bc = 0; } while (0)

/* (the body, compiled twice below: as it stands, and with BMI2's one-instruction variable shifts where the CPU has them -- the symbol loop
   is a chain of shifts by amounts read from the tables: +11 % on fastq with real quality lines, +17 % on text that matches well) */
static inline __attribute__ ((always_inline)) int
fi_run_body (fi_state * s, uint8_t ** outp, uint8_t * soft_end)
{
  const uint8_t *in = s->in, *const in_end = s->in_end;
  uint64_t bb = s->bb;
  int bc = s->bc;
  uint8_t *out = *outp;
  uint8_t *crc_from = out;
#define FI_ACCOUNT() do { if (out > crc_from) { s->crc = fi_crc32 (s->crc, crc_from, (size_t) (out - crc_from)); s->member_out += (uint64_t) (out - crc_from); crc_from = out; } } while (0)
  for (;;)
    {
      if (s->phase == 4)
        {
          *outp = out;
          return s->msg ? FI_ERROR : FI_END;
        }
      if (s->phase == 0)
        {
          /* ---- member header (RFC 1952); nothing buffered here */
          if (in == in_end && s->members > 0)
            {
              s->phase = 4;
              continue;
            }
          if (in_end - in < 2 || in[0] != 0x1f || in[1] != 0x8b)
            {
              if (s->members > 0)
                {
                  s->phase = 4; /* bytes behind the last member that are no header: ignored, as gzread ignores them */
                  continue;
                }
              FI_FAIL ("not a gzip stream");
            }
          if (in_end - in < 10 || in[2] != 8 || (in[3] & 0xE0))
            FI_FAIL ("gzip header damaged");
          const int flg = in[3];
          in += 10;
          if (flg & 4)
            {
              if (in_end - in < 2)
                FI_FAIL ("gzip header cut short");
              const size_t xlen = (size_t) in[0] | ((size_t) in[1] << 8);
              in += 2;
              if ((size_t) (in_end - in) < xlen)
                FI_FAIL ("gzip header cut short");
              in += xlen;
            }
          for (int f = 8; f <= 16; f <<= 1)
            if (flg & f)
              {
                const uint8_t *z = (const uint8_t *) memchr (in, 0, (size_t) (in_end - in));
                if (!z)
                  FI_FAIL ("gzip header cut short");
                in = z + 1;
              }
          if (flg & 2)
            {
              if (in_end - in < 2)
                FI_FAIL ("gzip header cut short");
              in += 2;
            }
          s->crc = 0;
          s->member_out = 0;
          s->floor = out;
          crc_from = out;
          s->phase = 1;
        }
      if (s->phase == 1)
        {
          /* ---- block header */
          FI_NEED (3, "stream ends before a block header");
          s->last_block = (int) (bb & 1u);
          const int type = (int) ((bb >> 1) & 3u);
          FI_DROP (3);
          if (type == 0)
            {
              FI_UNREAD ();
              if (in_end - in < 4)
                FI_FAIL ("stored block cut short");
              const uint32_t len = (uint32_t) in[0] | ((uint32_t) in[1] << 8), nlen = (uint32_t) in[2] | ((uint32_t) in[3] << 8);
              if ((len ^ nlen) != 0xFFFFu)
                FI_FAIL ("stored block length damaged");
              in += 4;
              s->stored_left = len;
              s->phase = 3;
            }
          else if (type == 1)
            {
              uint8_t lens[288 + 32];
              int i = 0;
              for (; i < 144; i++)
                lens[i] = 8;
              for (; i < 256; i++)
                lens[i] = 9;
              for (; i < 280; i++)
                lens[i] = 7;
              for (; i < 288; i++)
                lens[i] = 8;
              for (i = 0; i < 32; i++)
                lens[288 + i] = 5;
              fi_build (s->lt, FI_LSIZE, FI_LBITS, lens, 288, 0);
              fi_build (s->dt, FI_DSIZE, FI_DBITS, lens + 288, 32, 1);
              s->phase = 2;
            }
          else if (type == 2)
            {
              FI_NEED (14, "dynamic block header cut short");
              const int hlit = (int) (bb & 31u) + 257, hdist = (int) ((bb >> 5) & 31u) + 1, hclen = (int) ((bb >> 10) & 15u) + 4;
              FI_DROP (14);
              if (hlit > 286 || hdist > 30)
                FI_FAIL ("too many length or distance codes");
              static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
              uint8_t cl[19] = { 0 };
              for (int i = 0; i < hclen; i++)
                {
                  FI_NEED (3, "dynamic block header cut short");
                  cl[order[i]] = (uint8_t) (bb & 7u);
                  FI_DROP (3);
                }
              uint32_t ct[128];
              if (fi_build (ct, 128, 7, cl, 19, 2))
                FI_FAIL ("code lengths set damaged");
              uint8_t lens[286 + 30 + 140];
              int i = 0;
              while (i < hlit + hdist)
                {
                  FI_NEED (7 + 7, "dynamic block header cut short");
                  const uint32_t e = ct[bb & 127u];
                  if (!e)
                    FI_FAIL ("code lengths damaged");
                  FI_DROP ((int) (e & 63u));
                  const int sym = (int) (e >> 16);
                  if (sym < 16)
                    lens[i++] = (uint8_t) sym;
                  else
                    {
                      int rep, val = 0;
                      if (sym == 16)
                        {
                          if (i == 0)
                            FI_FAIL ("code lengths repeat nothing");
                          val = lens[i - 1];
                          rep = 3 + (int) (bb & 3u);
                          FI_DROP (2);
                        }
                      else if (sym == 17)
                        {
                          rep = 3 + (int) (bb & 7u);
                          FI_DROP (3);
                        }
                      else
                        {
                          rep = 11 + (int) (bb & 127u);
                          FI_DROP (7);
                        }
                      if (i + rep > hlit + hdist)
                        FI_FAIL ("code lengths run past the end");
                      memset (lens + i, val, (size_t) rep);
                      i += rep;
                    }
                }
              if (lens[256] == 0)
                FI_FAIL ("no end-of-block code");
              if (fi_build (s->lt, FI_LSIZE, FI_LBITS, lens, hlit, 0))
                FI_FAIL ("literal/length code damaged");
              if (fi_build (s->dt, FI_DSIZE, FI_DBITS, lens + hlit, hdist, 1))
                FI_FAIL ("distance code damaged");
              s->phase = 2;
            }
          else
            FI_FAIL ("block type 3");
        }
      if (s->phase == 3)
        {
          /* ---- stored bytes */
          while (s->stored_left)
            {
              if (out >= soft_end)
                goto suspend;
              size_t m = s->stored_left;
              if (m > (size_t) (soft_end - out))
                m = (size_t) (soft_end - out);
              if ((size_t) (in_end - in) < m)
                FI_FAIL ("stored block cut short");
              memcpy (out, in, m);
              out += m;
              in += m;
              s->stored_left -= (uint32_t) m;
            }
          s->phase = s->last_block ? 5 : 1;
        }
      if (s->phase == 2)
        {
          /* ---- Huffman-coded symbols, until the block ends or the output reaches soft_end */
          const uint32_t *const lt = s->lt, *const dt = s->dt;
          const uint8_t *const floor = s->floor;
          for (;;)
            {
              if (out >= soft_end)
                goto suspend;
              FI_REFILL ();
              uint32_t e = lt[bb & ((1u << FI_LBITS) - 1u)];
              if (e & FI_E_SUB)
                e = lt[(e >> 16) + ((bb >> FI_LBITS) & ((1u << ((e >> 8) & 15u)) - 1u))];
              if (e & FI_E_LIT)
                {
                  /* up to three literals per refill (3 x 15 bits of the 56) */
                  if ((int) (e & 63u) > bc)
                    FI_FAIL ("stream cut short");
                  FI_DROP ((int) (e & 63u));
                  *out++ = (uint8_t) (e >> 16);
                  e = lt[bb & ((1u << FI_LBITS) - 1u)];
                  if (e & FI_E_SUB)
                    e = lt[(e >> 16) + ((bb >> FI_LBITS) & ((1u << ((e >> 8) & 15u)) - 1u))];
                  if (!(e & FI_E_LIT) || (int) (e & 63u) > bc)
                    continue;
                  FI_DROP ((int) (e & 63u));
                  *out++ = (uint8_t) (e >> 16);
                  e = lt[bb & ((1u << FI_LBITS) - 1u)];
                  if (e & FI_E_SUB)
                    e = lt[(e >> 16) + ((bb >> FI_LBITS) & ((1u << ((e >> 8) & 15u)) - 1u))];
                  if (!(e & FI_E_LIT) || (int) (e & 63u) > bc)
                    continue;
                  FI_DROP ((int) (e & 63u));
                  *out++ = (uint8_t) (e >> 16);
                  continue;
                }
              if (e == 0u)
                FI_FAIL ("invalid literal/length code");
              if ((int) (e & 63u) > bc)
                FI_FAIL ("stream cut short");
              if (e & FI_E_EOB)
                {
                  FI_DROP ((int) (e & 63u));
                  s->phase = s->last_block ? 5 : 1;
                  break;
                }
              /* a length, then a distance: 20 + 28 bits at most, all in the buffer since the refill */
              const unsigned cl = (e >> 8) & 15u, tl = e & 63u;
              const unsigned len = (e >> 16) + (unsigned) ((bb >> cl) & ((1u << (tl - cl)) - 1u));
              FI_DROP ((int) tl);
              uint32_t d = dt[bb & ((1u << FI_DBITS) - 1u)];
              if (d & FI_E_SUB)
                d = dt[(d >> 16) + ((bb >> FI_DBITS) & ((1u << ((d >> 8) & 15u)) - 1u))];
              if (d == 0u)
                FI_FAIL ("invalid distance code");
              const unsigned cd = (d >> 8) & 15u, td = d & 63u;
              if ((int) td > bc)
                FI_FAIL ("stream cut short");
              const size_t dist = (size_t) (d >> 16) + (size_t) ((bb >> cd) & ((1u << (td - cd)) - 1u));
              FI_DROP ((int) td);
              if (dist > (size_t) (out - floor))
                FI_FAIL ("distance reaches before the start of the data");
              const uint8_t *src = out - dist;
              uint8_t *const end = out + len;
              if (dist >= 8)
                {
                  do
                    {
                      memcpy (out, src, 8);
                      out += 8;
                      src += 8;
                    }
                  while (out < end);
                }
              else if (dist == 1)
                {
                  const uint64_t v = 0x0101010101010101ull * (uint64_t) *src;
                  do
                    {
                      memcpy (out, &v, 8);
                      out += 8;
                    }
                  while (out < end);
                }
              else
                {
                  do
                    *out++ = *src++;
                  while (out < end);
                }
              out = end;
            }
        }
      if (s->phase == 5)
        {
          /* ---- member trailer: CRC-32 and length of what was produced */
          FI_UNREAD ();
          FI_ACCOUNT ();
          if (in_end - in < 8)
            FI_FAIL ("gzip trailer cut short");
          const uint32_t crc = (uint32_t) in[0] | ((uint32_t) in[1] << 8) | ((uint32_t) in[2] << 16) | ((uint32_t) in[3] << 24);
          const uint32_t isize = (uint32_t) in[4] | ((uint32_t) in[5] << 8) | ((uint32_t) in[6] << 16) | ((uint32_t) in[7] << 24);
          in += 8;
          if (crc != s->crc)
            FI_FAIL ("CRC of the data does not match the gzip trailer");
          if (isize != (uint32_t) s->member_out)
            FI_FAIL ("length of the data does not match the gzip trailer");
          s->members++;
          s->phase = 0;
        }
    }
suspend:
  FI_ACCOUNT ();
  s->in = in;
  s->bb = bb;
  s->bc = bc;
  *outp = out;
  return FI_MORE;
}

#undef FI_ACCOUNT

static int
fi_run_plain (fi_state * s, uint8_t ** outp, uint8_t * soft_end)
{
  return fi_run_body (s, outp, soft_end);
}

#if defined(__x86_64__)
__attribute__ ((target ("bmi,bmi2")))
static int
fi_run_bmi2 (fi_state * s, uint8_t ** outp, uint8_t * soft_end)
{
  return fi_run_body (s, outp, soft_end);
}
#endif

static int
fi_run (fi_state * s, uint8_t ** outp, uint8_t * soft_end)
{
#if defined(__x86_64__)
  static int have = -1;
  if (have < 0)
    have = __builtin_cpu_supports ("bmi2") && __builtin_cpu_supports ("bmi");
  if (have)
    return fi_run_bmi2 (s, outp, soft_end);
#endif
  return fi_run_plain (s, outp, soft_end);
}
#endif
