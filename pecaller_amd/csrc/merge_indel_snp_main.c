/* merge_indel_snp: sorts a PECaller .snp file by the .sdx's contig order and position, names the deletions by their length and
 * the insertions by the sequence most samples' reads carry (read from the mapper's <sample>.indel.txt.gz files).
 *
 * Same command line, same output file as the reference's script (src/merge_indel_snp.pl):
 *
 *     merge_indel_snp  sdx_file  snp_file  directory_indel_files  outname
 *
 * What the script does, and this program after it (line numbers of merge_indel_snp.pl):
 *   - contig order = order of the .sdx's lines, names in the second column (30-45); a contig the .sdx does not name sorts as the
 *     first one (an undefined number compares as 0).  The per-contig position offset of the script stays 0 (its increment is
 *     commented out, 44): indel-file positions are used as they are (115).
 *   - snp file: the header line is kept; samples are its fields 6, 8, 10, ... (52-54).  A row is a deletion when Type is DEL or
 *     DENOVO_DEL, or MULTIALLELIC / DENOVO_MULTIALLELIC with a D among the Alleles; an insertion when INS / DENOVO_INS or
 *     multi-allelic with an I and no D; rows with an I need the consensus of their position (59-93).
 *   - every sample's <dir>/<sample>.indel.txt.gz (a file that cannot be opened ends the program, 101-103): first line skipped,
 *     fields 7... of a row at a needed "contig_position" are counted, all samples together (109-126).
 *   - consensus = the sequence with the highest count (130-141).  The script walks a hash, so among sequences with EQUAL counts
 *     its pick changes from run to run; this program takes the one seen first.
 *   - stable sort by (contig number, position) (143-148, 186-203; Perl's sort is a merge sort).
 *   - a deletion row absorbs the deletion rows that follow it at consecutive positions (contigs are not compared, 158-165): they
 *     are not written, and the first D of its Alleles becomes -<rows absorbed + 1>; its first I, if its position has a
 *     consensus, becomes +<sequence>.  An insertion row's first I becomes +<sequence>.  The new Alleles replace the FIRST place in
 *     the row's text where the old Alleles string occurs (172, 184: a substitution on the whole line, not on the field).
 *   - looking one row past the end of the sorted list reads row 0 of the file (an undefined index, 160-161); kept.
 *
 * Plain C on the host; nothing here touches the GPU ("cheap text transforms"). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

static void
die (const char *msg, const char *arg)
{
  printf (msg, arg);
  exit (1);
}

/* ---- whole-line reader (lines of any length) */
static char *
read_line (FILE * f, gzFile g, size_t *len_out)
{
  size_t cap = 1 << 16, n = 0;
  char *b = (char *) malloc (cap);
  for (;;)
    {
      char *r = f ? fgets (b + n, (int) (cap - n), f) : gzgets (g, b + n, (int) (cap - n));
      if (!r)
        break;
      n += strlen (b + n);
      if (n && b[n - 1] == '\n')
        break;
      if (cap - n < 2)
        b = (char *) realloc (b, cap *= 2);
    }
  if (n == 0)
    {
      free (b);
      return NULL;
    }
  if (b[n - 1] == '\n')
    b[--n] = '\0';
  if (len_out)
    *len_out = n;
  return b;
}

/* split at tabs in place (Perl's split: trailing empty fields are dropped) */
static int
split_tabs (char *s, char ***out)
{
  int cap = 64, n = 0;
  char **f = (char **) malloc (sizeof (char *) * cap);
  char *p = s;
  for (;;)
    {
      if (n == cap)
        f = (char **) realloc (f, sizeof (char *) * (cap *= 2));
      f[n++] = p;
      char *t = strchr (p, '\t');
      if (!t)
        break;
      *t = '\0';
      p = t + 1;
    }
  while (n > 0 && f[n - 1][0] == '\0')
    n--;
  *out = f;
  return n;
}

/* ---- string -> slot hash (open addressing; keys are owned copies) */
typedef struct
{
  char **key;
  long *val;
  size_t cap, n;
} smap;

static unsigned long
hash_str (const char *s)
{
  unsigned long h = 1469598103934665603ul;
  for (; *s; s++)
    h = (h ^ (unsigned char) *s) * 1099511628211ul;
  return h;
}

static void
smap_init (smap * m, size_t cap)
{
  m->cap = cap;
  m->n = 0;
  m->key = (char **) calloc (cap, sizeof (char *));
  m->val = (long *) calloc (cap, sizeof (long));
}

static long *smap_slot (smap * m, const char *k, int create);

static void
smap_grow (smap * m)
{
  smap old = *m;
  smap_init (m, old.cap * 2);
  for (size_t i = 0; i < old.cap; i++)
    if (old.key[i])
      {
        size_t h = hash_str (old.key[i]) & (m->cap - 1);
        while (m->key[h])
          h = (h + 1) & (m->cap - 1);
        m->key[h] = old.key[i];
        m->val[h] = old.val[i];
        m->n++;
      }
  free (old.key);
  free (old.val);
}

static long *
smap_slot (smap * m, const char *k, int create)
{
  if (create && (m->n + 1) * 2 > m->cap)
    smap_grow (m);
  size_t h = hash_str (k) & (m->cap - 1);
  while (m->key[h])
    {
      if (strcmp (m->key[h], k) == 0)
        return &m->val[h];
      h = (h + 1) & (m->cap - 1);
    }
  if (!create)
    return NULL;
  m->key[h] = strdup (k);
  m->val[h] = -1;
  m->n++;
  return &m->val[h];
}

/* ---- the insertions seen at one needed position */
typedef struct
{
  char **seq;
  long *count;
  int n, cap;
} ins_site;

enum
{ TYPE_SNP, TYPE_DEL, TYPE_INS };

typedef struct
{
  char *text;
  char *chr;                    /* copies of fields 0, 1 */
  long pos;
  long chr_no;
  int type;
} row_t;

static row_t *rows;

static int
row_less_eq (long a, long b)
{
  /* (a <= b in the script's order; the merge keeps equal rows in file order) */
  if (rows[a].chr_no != rows[b].chr_no)
    return rows[a].chr_no < rows[b].chr_no;
  return rows[a].pos <= rows[b].pos;
}

static void
merge_sort (long *v, long *tmp, long n)
{
  if (n < 2)
    return;
  const long h = n / 2;
  merge_sort (v, tmp, h);
  merge_sort (v + h, tmp, n - h);
  long i = 0, j = h, k = 0;
  while (i < h && j < n)
    tmp[k++] = row_less_eq (v[i], v[j]) ? v[i++] : v[j++];
  while (i < h)
    tmp[k++] = v[i++];
  while (j < n)
    tmp[k++] = v[j++];
  memcpy (v, tmp, sizeof (long) * n);
}

/* the first occurrence of `what` in s replaced by `with` (new string); s itself when `what` does not occur */
static char *
replace_first (const char *s, const char *what, const char *with)
{
  const char *at = strstr (s, what);
  if (!at)
    return strdup (s);
  const size_t a = (size_t) (at - s), lw = strlen (what), lr = strlen (with), ls = strlen (s);
  char *o = (char *) malloc (ls - lw + lr + 1);
  memcpy (o, s, a);
  memcpy (o + a, with, lr);
  memcpy (o + a + lr, at + lw, ls - a - lw + 1);
  return o;
}

int
main (int argc, char **argv)
{
  if (argc != 5)
    {
      printf ("\n Usage: %s sdx_file snp_file directory_indel_files outname \n\n ", argv[0]);
      return 1;
    }
  /* ---- contig numbers */
  FILE *f = fopen (argv[1], "r");
  if (!f)
    die ("\nCan't open file %s which should contain genome_sdx_file\n", argv[1]);
  smap chr_num;
  smap_init (&chr_num, 1024);
  {
    char *l = read_line (f, NULL, NULL);
    const long chr_count = l ? atol (l) : 0;
    free (l);
    for (long i = 0; i < chr_count; i++)
      {
        if (!(l = read_line (f, NULL, NULL)))
          break;
        char **fl;
        const int nf = split_tabs (l, &fl);
        if (nf > 1)
          *smap_slot (&chr_num, fl[1], 1) = i;  /* (a name listed twice keeps its last number, as the script's hash does) */
        free (fl);
        free (l);
      }
  }
  fclose (f);
  printf ("\n Finished Reading Genome File \n");

  /* ---- the snp file */
  if (!(f = fopen (argv[2], "r")))
    die ("\nCan't open file %s which should contain sequencing data\n", argv[2]);
  char *header = read_line (f, NULL, NULL);
  if (!header)
    header = strdup ("");
  char **sample = NULL;
  int n_samples = 0;
  {
    char *h = strdup (header), **fl;
    const int nf = split_tabs (h, &fl);
    sample = (char **) malloc (sizeof (char *) * (nf / 2 + 1));
    for (int i = 6; i < nf; i += 2)
      sample[n_samples++] = strdup (fl[i]);
    free (fl);
    free (h);
  }
  smap need;                    /* "contig_position" -> index into sites[] */
  smap_init (&need, 1 << 16);
  ins_site *sites = NULL;
  long n_sites = 0, cap_sites = 0;
  long n_rows = 0, cap_rows = 1 << 16;
  rows = (row_t *) malloc (sizeof (row_t) * cap_rows);
  char *l;
  while ((l = read_line (f, NULL, NULL)))     /* (an empty line is a row of the script too) */
    {
      if (n_rows == cap_rows)
        rows = (row_t *) realloc (rows, sizeof (row_t) * (cap_rows *= 2));
      row_t *r = &rows[n_rows];
      r->text = l;
      char *c = strdup (l), **fl;
      const int nf = split_tabs (c, &fl);
      r->chr = strdup (nf > 0 ? fl[0] : "");
      const char *pos_s = nf > 1 ? fl[1] : "";
      r->pos = atol (pos_s);
      const long *cn = smap_slot (&chr_num, r->chr, 0);
      r->chr_no = cn ? *cn : 0;
      r->type = TYPE_SNP;
      const char *ty = nf > 5 ? fl[5] : "";
      int needs = 0;
      if (!strcmp (ty, "INS") || !strcmp (ty, "DENOVO_INS"))
        {
          r->type = TYPE_INS;
          needs = 1;
        }
      else if (!strcmp (ty, "DEL") || !strcmp (ty, "DENOVO_DEL"))
        r->type = TYPE_DEL;
      else if (!strcmp (ty, "MULTIALLELIC") || !strcmp (ty, "DENOVO_MULTIALLELIC"))
        {
          /* the Alleles, split at commas */
          char *a = strdup (nf > 3 ? fl[3] : "");
          for (char *t = a, *e; t; t = e ? e + 1 : NULL)
            {
              if ((e = strchr (t, ',')))
                *e = '\0';
              if (!strcmp (t, "I"))
                {
                  if (r->type != TYPE_DEL)
                    r->type = TYPE_INS;
                  needs = 1;
                }
              else if (!strcmp (t, "D"))
                r->type = TYPE_DEL;
            }
          free (a);
        }
      if (needs)
        {
          char *name = (char *) malloc (strlen (r->chr) + strlen (pos_s) + 2);
          sprintf (name, "%s_%s", r->chr, pos_s);
          long *slot = smap_slot (&need, name, 1);
          if (*slot < 0)
            {
              if (n_sites == cap_sites)
                sites = (ins_site *) realloc (sites, sizeof (ins_site) * (cap_sites = cap_sites ? cap_sites * 2 : 1024));
              memset (&sites[n_sites], 0, sizeof (ins_site));
              *slot = n_sites++;
            }
          free (name);
        }
      free (fl);
      free (c);
      n_rows++;
      if (n_rows % 100000 == 0)
        printf ("\n Read %ld lines of the SNP file \n", n_rows);
    }
  fclose (f);

  /* ---- the samples' insertion files */
  for (int s = 0; s < n_samples; s++)
    {
      char *path = (char *) malloc (strlen (argv[3]) + strlen (sample[s]) + 32);
      sprintf (path, "%s/%s.indel.txt.gz", argv[3], sample[s]);
      gzFile g = gzopen (path, "rb");
      if (!g)
        {
          fprintf (stderr, "\n Can not open %s.indel.txt.gz \n", sample[s]);
          return 2;
        }
      gzbuffer (g, 1 << 20);
      printf ("\n Working on file %s.indel.txt.gz \n", sample[s]);
      long n_lines = 0;
      while ((l = read_line (NULL, g, NULL)))
        {
          if (n_lines++ > 0)
            {
              char **fl;
              const int nf = split_tabs (l, &fl);
              if (nf > 7)
                {
                  /* (the script's offset is 0: "contig_" . (position - 0); a position is printed back as the number it reads as) */
                  char *name = (char *) malloc (strlen (fl[0]) + 32);
                  sprintf (name, "%s_%ld", fl[0], atol (fl[1]));
                  const long *slot = smap_slot (&need, name, 0);
                  free (name);
                  if (slot)
                    {
                      ins_site *is = &sites[*slot];
                      for (int j = 7; j < nf; j++)
                        {
                          int q = 0;
                          while (q < is->n && strcmp (is->seq[q], fl[j]))
                            q++;
                          if (q == is->n)
                            {
                              if (is->n == is->cap)
                                {
                                  is->cap = is->cap ? is->cap * 2 : 4;
                                  is->seq = (char **) realloc (is->seq, sizeof (char *) * is->cap);
                                  is->count = (long *) realloc (is->count, sizeof (long) * is->cap);
                                }
                              is->seq[q] = strdup (fl[j]);
                              is->count[q] = 0;
                              is->n++;
                            }
                          is->count[q]++;
                        }
                    }
                }
              free (fl);
            }
          free (l);
        }
      gzclose (g);
      printf ("\n Found a total of %ld lines \n", n_lines);
      free (path);
    }
  printf ("\n Making Consensus insertions \n");
  /* consensus of a site = index of its most frequent sequence, -1 when its position was in no file */
  int *best = (int *) malloc (sizeof (int) * (n_sites ? n_sites : 1));
  for (size_t h = 0; h < need.cap; h++)
    if (need.key[h])
      {
        const ins_site *is = &sites[need.val[h]];
        if (is->n == 0)
          printf ("\n This is impossible.  We fail to find insertion %s \n", need.key[h]);
        long top = -50;
        int b = -1;
        for (int q = 0; q < is->n; q++)
          if (is->count[q] > top)
            {
              top = is->count[q];
              b = q;
            }
        best[need.val[h]] = b;
      }

  /* ---- sort, rewrite, write */
  long *order = (long *) malloc (sizeof (long) * (n_rows + 1)), *tmp = (long *) malloc (sizeof (long) * (n_rows + 1));
  for (long i = 0; i < n_rows; i++)
    order[i] = i;
  printf ("\n About to Sort SNPs \n");
  merge_sort (order, tmp, n_rows);
  FILE *o = fopen (argv[4], "w");
  if (!o)
    die ("\nCan not open %s for writing \n", argv[4]);
  printf ("\n Writing output \n");
  fprintf (o, "%s\n", header);
  /* (one past the end of the list the script reads row 0) */
#define ORDER(k) ((k) < n_rows ? order[k] : 0)
  for (long i = 0; i < n_rows; i++)
    {
      const long j = order[i];
      row_t *r = &rows[j];
      if (r->type == TYPE_DEL || r->type == TYPE_INS)
        {
          char *c = strdup (r->text), **fl;
          const int nf = split_tabs (c, &fl);
          const char *old3 = nf > 3 ? fl[3] : "";
          char *name = (char *) malloc (strlen (nf > 0 ? fl[0] : "") + strlen (nf > 1 ? fl[1] : "") + 2);
          sprintf (name, "%s_%s", nf > 0 ? fl[0] : "", nf > 1 ? fl[1] : "");
          const long *slot = smap_slot (&need, name, 0);
          const char *cons = slot && best[*slot] >= 0 ? sites[*slot].seq[best[*slot]] : NULL;
          free (name);
          char *new3 = strdup (old3);
          if (r->type == TYPE_DEL)
            {
              long allele = 1, k = i + 1;
              while (k <= n_rows && rows[ORDER (k)].pos - rows[ORDER (k - 1)].pos == 1 && rows[ORDER (k)].type == TYPE_DEL)
                {
                  allele++;
                  k++;
                }
              char a[64];
              sprintf (a, "-%ld", allele);
              char *t = replace_first (new3, "D", a);
              free (new3);
              new3 = t;
              i = k - 1;
            }
          if (cons)
            {
              char *a = (char *) malloc (strlen (cons) + 2);
              sprintf (a, "+%s", cons);
              char *t = replace_first (new3, "I", a);
              free (new3);
              new3 = t;
              free (a);
            }
          /* (an empty old Alleles string matches at the start of the line) */
          char *t = replace_first (r->text, old3, new3);
          fprintf (o, "%s\n", t);
          free (t);
          free (new3);
          free (fl);
          free (c);
        }
      else
        fprintf (o, "%s\n", r->text);
    }
  if (fclose (o))
    die ("\nCan not write %s \n", argv[4]);
  return 0;
}
