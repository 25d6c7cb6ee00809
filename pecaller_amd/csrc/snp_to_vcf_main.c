/* snp_to_vcf: a (merged) PECaller .snp file as VCF 4.0 on stdout.
 *
 *     snp_to_vcf  sdx_file  snpfile  min_prob_to_make_call  > out.vcf
 *
 * Written from the grammar of the .snp rows, as three small parts:
 *
 *   genome      the .sdx contig table and the .seq letters (index_genome's files): "the n letters before position p of contig c"
 *   allele list the row's 4th column taken apart into items -- LETTER | '+' INSERTED | '-' COUNT, separated by commas -- what
 *               pecaller prints (pecaller.c:1675) after merge_indel_snp has put the inserted letters / deleted lengths in
 *   site        REF, the ALT strings (kept as a list, joined when printed), FILTER and the names "a/b" of the genotype letters
 *
 * and one function per row Type that fills a site from the allele list.  The text printed is the reference program's
 * (src/snp_to_vcf.c:173-181 header, 505-519 rows; compared byte for byte in tests/test_downstream_tools.py), including what that
 * program does with rows a caller would not normally print.  Those behaviours, kept because a drop-in must print the same file:
 *   - Types LOW and MESS are not printed; SNP, MULTIALLELIC and INS have forms of their own; every other Type (DEL and all
 *     DENOVO_*) is printed as a deletion whose length is the number behind the sign of the row's second item (0 if none);
 *   - the names of the 14 genotype letters A C G T I D / Y R S W K M E H are reset for every printed row (homozygotes "1/1",
 *     heterozygotes "0/1", the reference letter "0/0"); names given to any other byte stay until a later row renames it;
 *   - inside a MULTIALLELIC row a letter equal to the FIRST letter of REF as it stands is not an allele -- after a deletion item
 *     REF starts one base earlier, so a letter equal to that base is passed over too;
 *   - a deletion item rewrites the alleles before it as REF with the second letter replaced; the reference finds that letter at
 *     every second character of the joined ALT text, so an inserted allele before a deletion contributes its first letter only
 *     if it is the first allele (alt_letter_at);
 *   - "##phasing=none" and the ##INFO line share a line; the date is printed without padding.
 * With three arguments the reference reads argv[3] == NULL; this program takes min_prob = 0 then.
 *
 * Plain C on the host; nothing here touches the GPU. */
#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>

/* ---------------------------------------------------------------- small growing strings */
typedef struct
{
  char *s;
  size_t n, cap;
} Str;

static void
str_room (Str * t, size_t more)
{
  if (t->n + more + 1 <= t->cap)
    return;
  while (t->n + more + 1 > t->cap)
    t->cap = t->cap ? 2 * t->cap : 64;
  t->s = (char *) realloc (t->s, t->cap);
  if (!t->s)
    {
      fprintf (stderr, "\n out of memory\n");
      exit (1);
    }
}

static void
str_clear (Str * t)
{
  str_room (t, 0);
  t->n = 0;
  t->s[0] = '\0';
}

static void
str_put (Str * t, const char *p, size_t len)
{
  str_room (t, len);
  memcpy (t->s + t->n, p, len);
  t->n += len;
  t->s[t->n] = '\0';
}

static void
str_putc (Str * t, char c)
{
  str_put (t, &c, 1);
}

/* ---------------------------------------------------------------- genome: .sdx + .seq */
typedef struct
{
  int n_contigs;
  char **name;
  long *first;                  /* offset of the contig's first letter in .seq: the lengths before it + 15 per contig before it */
  char *letters;
  long size;
  int cached;                   /* the contig found last (rows come contig by contig) */
} Genome;

static int
genome_load (Genome * g, const char *sdx_arg)
{
  FILE *f = fopen (sdx_arg, "r");
  if (!f)
    {
      printf ("\n Can not open file %s\n", sdx_arg);
      return 0;
    }
  char line[1100];
  g->n_contigs = fgets (line, 256, f) ? atoi (line) : 0;
  if (g->n_contigs < 0)
    g->n_contigs = 0;
  g->name = (char **) calloc ((size_t) g->n_contigs + 1, sizeof (char *));
  g->first = (long *) calloc ((size_t) g->n_contigs + 2, sizeof (long));
  long at = 0;
  for (int c = 0; c < g->n_contigs; c++)
    {
      long len = 0;
      char nm[1024] = "";
      if (fgets (line, 1024, f))
        sscanf (line, "%ld %1023s", &len, nm);
      g->name[c] = strdup (nm);
      g->first[c] = at;
      at += (long) (unsigned int) len + 15;     /* the .sdx line holds length - 15, the .seq has the whole contig + nothing between */
    }
  fclose (f);
  g->size = at;
  g->cached = -1;
  /* <base>.seq, where <base> is the argument up to its last '.' if it names an .sdx file */
  Str path = { 0 };
  str_put (&path, sdx_arg, strlen (sdx_arg));
  if (strstr (path.s, ".sdx"))
    {
      char *dot = strrchr (path.s + 1, '.');
      if (dot)
        {
          *dot = '\0';
          path.n = strlen (path.s);
        }
    }
  str_put (&path, ".seq", 4);
  g->letters = (char *) calloc ((size_t) g->size + 16, 1);
  if (!g->letters)
    {
      fprintf (stderr, "\n Failed to allocate memory for the Genome Buffer \n");
      return 0;
    }
  gzFile z = gzopen (path.s, "r");
  if (!z)
    {
      printf ("\n Can not open file %s for reading\n", path.s);
      return 0;
    }
  gzbuffer (z, 1 << 25);
  for (long have = 0; have < g->size;)
    {
      const long ask = g->size - have < (1L << 28) ? g->size - have : (1L << 28);
      const int got = gzread (z, g->letters + have, (unsigned) ask);
      if (got <= 0)
        break;
      have += got;
    }
  gzclose (z);
  free (path.s);
  return 1;
}

/* -> contig number, or -1 */
static int
genome_contig (Genome * g, const char *name)
{
  if (g->cached >= 0 && strcmp (g->name[g->cached], name) == 0)
    return g->cached;
  for (int c = 0; c < g->n_contigs; c++)
    if (strcmp (g->name[c], name) == 0)
      return g->cached = c;
  return -1;
}

/* the letters of contig c from 1-based position p on, at most n of them (a position outside the file reads as nothing) */
static void
genome_copy (const Genome * g, int c, long p, long n, Str * out)
{
  str_clear (out);
  long at = g->first[c] + p - 1;
  if (at < 0 || at >= g->size || n <= 0)
    return;
  const char *s = g->letters + at;
  size_t len = strnlen (s, (size_t) n);
  str_put (out, s, len);
}

/* ---------------------------------------------------------------- rows and their fields */
typedef struct
{
  char **f;
  int n, cap;
} Fields;

/* cuts the line into its fields in place: runs of tab / space / newline separate (so the empty column pecaller prints after a
   sample's name does not count) */
static void
fields_split (Fields * fl, char *line)
{
  fl->n = 0;
  char *p = line;
  for (;;)
    {
      while (*p == '\t' || *p == ' ' || *p == '\n')
        p++;
      if (!*p)
        break;
      if (fl->n == fl->cap)
        {
          fl->cap = fl->cap ? 2 * fl->cap : 256;
          fl->f = (char **) realloc (fl->f, (size_t) fl->cap * sizeof (char *));
        }
      fl->f[fl->n++] = p;
      while (*p && *p != '\t' && *p != ' ' && *p != '\n')
        p++;
      if (*p)
        *p++ = '\0';
    }
}

static int
read_line (gzFile z, Str * line)
{
  str_clear (line);
  for (;;)
    {
      str_room (line, 1 << 16);
      if (!gzgets (z, line->s + line->n, (int) (line->cap - line->n)))
        break;
      line->n += strlen (line->s + line->n);
      if (line->n && line->s[line->n - 1] == '\n')
        break;
    }
  return line->n > 0;
}

/* ---------------------------------------------------------------- the allele column */
enum
{ ITEM_LETTER, ITEM_INS, ITEM_DEL };
typedef struct
{
  int kind;
  const char *text;             /* the item without its sign */
  int len;
} Item;

#define MAX_ITEMS 64
static int
items_parse (const char *col, Item * it)
{
  int n = 0;
  const char *p = col;
  while (*p && n < MAX_ITEMS)
    {
      const char *q = strchr (p, ',');
      const int len = q ? (int) (q - p) : (int) strlen (p);
      if (*p == '+' || *p == '-')
        {
          it[n].kind = *p == '+' ? ITEM_INS : ITEM_DEL;
          it[n].text = p + 1;
          it[n].len = len - 1;
        }
      else
        {
          it[n].kind = ITEM_LETTER;
          it[n].text = p;
          it[n].len = len;
        }
      n++;
      if (!q)
        break;
      p = q + 1;
    }
  return n;
}

/* ---------------------------------------------------------------- a VCF site */
#define MAX_ALT 32
typedef struct
{
  long pos;
  Str ref;
  Str alt[MAX_ALT];
  int n_alt;
  const char *filter;
} Site;

/* names of the genotype letters; lives across rows (see the header) */
static char gt_name[256][24];

static void
gt_set (unsigned char letter, int a, int b)
{
  snprintf (gt_name[letter], sizeof gt_name[0], "%d/%d", a, b);
}

static void
gt_start_row (unsigned char ref)
{
  for (const char *c = "ACGTID"; *c; c++)
    gt_set ((unsigned char) *c, 1, 1);
  for (const char *c = "YRSWKMEH"; *c; c++)
    gt_set ((unsigned char) *c, 0, 1);
  gt_set (ref, 0, 0);
}

/* the letter pecaller prints for a sample carrying alleles x and y (IUPAC for two bases, E = a base with the deletion, H = a base
   with the insertion; D with I reads E, I with D reads H); 'N' for anything else */
static unsigned char
pair_letter (unsigned char x, unsigned char y)
{
  static const char bases[] = "ACGT";
  static const char iupac[4][4] = { {'N', 'M', 'R', 'W'}, {'M', 'N', 'S', 'Y'}, {'R', 'S', 'N', 'K'}, {'W', 'Y', 'K', 'N'} };
  const char *bx = x ? strchr (bases, x) : NULL, *by = y ? strchr (bases, y) : NULL;
  if (bx && by)
    return (unsigned char) iupac[bx - bases][by - bases];
  if (x == 'D' && (by || y == 'I'))
    return 'E';
  if (x == 'I' && (by || y == 'D'))
    return 'H';
  if (bx && y == 'D')
    return 'E';
  if (bx && y == 'I')
    return 'H';
  return 'N';
}

static Str *
site_new_alt (Site * s)
{
  Str *a = &s->alt[s->n_alt < MAX_ALT ? s->n_alt++ : MAX_ALT - 1];
  str_clear (a);
  return a;
}

/* character k of the ALT list as it would be printed (alleles joined by commas); 0 past its end */
static char
alt_letter_at (const Site * s, size_t k)
{
  for (int a = 0; a < s->n_alt; a++)
    {
      if (k < s->alt[a].n)
        return s->alt[a].s[k];
      k -= s->alt[a].n;
      if (a + 1 < s->n_alt)
        {
          if (k == 0)
            return ',';
          k--;
        }
    }
  return '\0';
}

/* REF becomes the base before the site plus `count` deleted bases; the site moves one base back.  -> 0 if the contig is unknown */
static int
site_open_deletion (Site * s, Genome * g, const char *contig, long count, long most)
{
  const int c = genome_contig (g, contig);
  if (c < 0)
    {
      printf ("\n Failed to find chrom = %s \n", contig);
      return 0;
    }
  s->pos--;
  long n = count + 1;
  if (n > most)
    n = most;                   /* (the reference's buffers hold this many letters) */
  genome_copy (g, c, s->pos, n, &s->ref);
  return 1;
}

static long
number_at (const char *p, int len)
{
  char tmp[24];
  if (len < 0)
    len = 0;
  if (len > 23)
    len = 23;
  memcpy (tmp, p, (size_t) len);
  tmp[len] = '\0';
  return atol (tmp);
}

/* Type SNP: two letters, one of them the reference's */
static int
fill_snp (Site * s, char ref, const Item * it, int n)
{
  char a = 0;
  if (n > 0 && it[0].len > 0 && it[0].text[0] != ref)
    a = it[0].text[0];
  else if (n > 1 && it[1].len > 0)
    a = it[1].kind == ITEM_LETTER ? it[1].text[0] : it[1].text[-1];
  str_putc (site_new_alt (s), a);
  gt_set ((unsigned char) a, 1, 1);
  return 1;
}

/* the text of a row's one insertion or deletion behind its sign.  The caller prints such a row as "X,+LETTERS" / "X,-COUNT", or as the
   signed item alone when no sample carries the reference's letter; the reference program does not look for the sign but counts
   characters -- one if the column has no comma, three if it has -- and so does this */
static const char *
indel_body (const char *col)
{
  const size_t skip = (col[0] && strchr (col + 1, ',')) ? 3 : 1;
  return strlen (col) >= skip ? col + skip : "";
}

/* Type INS: the inserted letters follow the reference base */
static int
fill_ins (Site * s, char ref, const char *col)
{
  Str *a = site_new_alt (s);
  const char *body = indel_body (col);
  str_putc (a, ref);
  str_put (a, body, strlen (body));
  s->filter = ".";
  return 1;
}

/* Type DEL and every Type without a form of its own */
static int
fill_del (Site * s, Genome * g, const char *contig, const char *col)
{
  if (!site_open_deletion (s, g, contig, atol (indel_body (col)), 8190))
    return 0;
  str_putc (site_new_alt (s), s->ref.n ? s->ref.s[0] : '\0');
  s->filter = ".";
  return 1;
}

/* Type MULTIALLELIC: the items in order, each one allele number `k` (1, 2, ...) unless it is the reference's letter */
static int
fill_multi (Site * s, Genome * g, const char *contig, char ref, const Item * it, int n)
{
  unsigned char allele_of[MAX_ALT + 1];         /* the letter a sample homozygous for allele k is printed with */
  allele_of[0] = (unsigned char) ref;
  int k = 1, deleted = 0;
  for (int i = 0; i < n && k < 29; i++)
    {
      const char anchor = deleted ? (s->ref.n ? s->ref.s[0] : '\0') : ref;
      if (it[i].kind == ITEM_LETTER)
        {
          const char c = it[i].len > 0 ? it[i].text[0] : '\0';
          if (c == anchor)
            continue;
          gt_set ((unsigned char) c, k, k);
          allele_of[k] = (unsigned char) c;
          for (int x = 0; x <= k; x++)
            for (int y = x + 1; y <= k; y++)
              gt_set (pair_letter (allele_of[x], allele_of[y]), x, y);
          str_putc (site_new_alt (s), c);
        }
      else if (it[i].kind == ITEM_INS)
        {
          allele_of[k] = 'I';
          gt_set ('I', k, k);
          gt_set ('H', 0, k);
          Str *a = site_new_alt (s);
          if (deleted)
            str_put (a, s->ref.s, s->ref.n);
          else
            str_putc (a, ref);
          for (int j = 0; j < it[i].len; j++)
            if (isalpha ((unsigned char) it[i].text[j]))
              str_putc (a, it[i].text[j]);
          s->filter = ".";
        }
      else
        {
          allele_of[k] = 'D';
          gt_set ('D', k, k);
          gt_set ('E', 0, k);
          /* the letters the earlier alleles are known by, before REF changes under them */
          char known[MAX_ALT];
          for (int a = 0; a < s->n_alt; a++)
            known[a] = alt_letter_at (s, 2 * (size_t) a);
          if (!site_open_deletion (s, g, contig, number_at (it[i].text, it[i].len), 4190))
            return 0;
          deleted = 1;
          for (int a = 0; a < s->n_alt; a++)
            {
              str_clear (&s->alt[a]);
              if (s->ref.n == 0)
                continue;
              str_putc (&s->alt[a], s->ref.s[0]);
              if (known[a])
                {
                  str_putc (&s->alt[a], known[a]);
                  if (s->ref.n > 2)
                    str_put (&s->alt[a], s->ref.s + 2, s->ref.n - 2);
                }
            }
          str_putc (site_new_alt (s), s->ref.n ? s->ref.s[0] : '\0');
          s->filter = ".";
        }
      k++;
    }
  return 1;
}

int
main (int argc, char **argv)
{
  if (argc != 4 && argc != 3)
    {
      printf ("\nUsage: %s sdx_file snpfile [min_prob_to_make_call] \n", argv[0]);
      return 1;
    }
  double min_prob = 0.0;
  if (argc == 4)
    {
      const double v = atof (argv[3]);
      if (v >= 0.0 && v <= 1.0)
        min_prob = v;
    }
  Genome g;
  memset (&g, 0, sizeof g);
  if (!genome_load (&g, argv[1]))
    return 1;

  printf ("##fileformat=VCFv4.0\n");
  {
    const time_t now = time (NULL);
    const struct tm *tm = localtime (&now);
    printf ("##fileDate=%d%d%d\n", tm->tm_year + 1900, tm->tm_mon + 1, tm->tm_mday);
  }
  printf ("##reference=%s\n", argv[1]);
  printf ("##phasing=none" "##INFO=<ID=NS,Number=1,Type=Integer,Description=\"Number of Samples With Data\">\n");
  printf ("##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype Quality\">\n");
  printf ("##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n");
  printf ("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT");

  gzFile in = gzopen (argv[2], "r");
  if (!in)
    {
      printf ("\n Can not open file %s for reading\n", argv[2]);
      return 1;
    }
  gzbuffer (in, 1 << 25);
  Str line = { 0 };
  Fields fl = { 0 };
  /* header: six fixed columns, then the samples' names */
  read_line (in, &line);
  fields_split (&fl, line.s);
  const int n_samples = fl.n > 6 ? fl.n - 6 : 0;
  for (int i = 6; i < fl.n; i++)
    printf ("\t%s", fl.f[i]);

  for (int j = 0; j < 256; j++)
    strcpy (gt_name[j], "./.");
  static Site site;
  Item items[MAX_ITEMS];
  /* rows: Fragment Position Reference Alleles Allele_Counts Type, then letter and posterior per sample.  The reference stops at the
     first line of five characters or fewer */
  while (read_line (in, &line) && line.n > 5)
    {
      fields_split (&fl, line.s);
      if (fl.n < 6)
        break;
      const char *contig = fl.f[0], *type = fl.f[5];
      if (strcmp (type, "LOW") == 0 || strcmp (type, "MESS") == 0)
        continue;
      const char ref = fl.f[2][0];
      const int n_items = items_parse (fl.f[3], items);
      site.pos = atoi (fl.f[1]);
      site.n_alt = 0;
      site.filter = "PASS";
      str_clear (&site.ref);
      str_putc (&site.ref, ref);
      gt_start_row ((unsigned char) ref);
      int ok;
      if (strcmp (type, "SNP") == 0)
        ok = fill_snp (&site, ref, items, n_items);
      else if (strcmp (type, "MULTIALLELIC") == 0)
        ok = fill_multi (&site, &g, contig, ref, items, n_items);
      else if (strcmp (type, "INS") == 0)
        ok = fill_ins (&site, ref, fl.f[3]);
      else
        ok = fill_del (&site, &g, contig, fl.f[3]);
      if (!ok)
        return 1;
      printf ("\n%s\t%ld\t.\t%s\t", contig, site.pos, site.ref.s);
      for (int a = 0; a < site.n_alt; a++)
        printf ("%s%s", a ? "," : "", site.alt[a].s);
      printf ("\t.\t%s\tNS=%d\tGT:GQ", site.filter, n_samples);
      for (int i = 0; i < n_samples && 6 + 2 * i + 1 < fl.n; i++)
        {
          const char *letter = fl.f[6 + 2 * i], *post = fl.f[6 + 2 * i + 1];
          printf ("\t%s:%s", atof (post) >= min_prob ? gt_name[(unsigned char) letter[0]] : "./.", post);
        }
    }
  printf ("\n");
  return 0;
}
