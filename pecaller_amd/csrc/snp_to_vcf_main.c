/* snp_to_vcf: a (merged) PECaller .snp file as VCF 4.0 on stdout.
 *
 * Same command line and the same text as the reference's program (src/snp_to_vcf.c):
 *
 *     snp_to_vcf  sdx_file  snpfile  min_prob_to_make_call  > out.vcf
 *
 * The reference's rules, restated (line numbers of snp_to_vcf.c):
 *   - genome = <sdx base>.seq (gz), contig c begins at sum of the lengths before it + 15 c (117-171);
 *   - header block 173-181 as printed there ("##phasing=none" and the ##INFO line share a line; the date has no padding);
 *     the sample columns are the header's tokens from the 7th on (194-205: tokens, so the empty column after each name is skipped);
 *   - rows of Type LOW and MESS are dropped (284-288); SNP, MULTIALLELIC and INS have their own forms; EVERY other Type -- DEL and
 *     all DENOVO_* -- is read as a deletion (467-504);
 *   - a genotype letter prints as the entry of a 256-entry table of "a/b" strings (289-303 reset the entries of
 *     A C G T I D Y R S W K M E H on every kept row; what a MULTIALLELIC row wrote into other entries -- 'N' through an unknown
 *     pair -- stays for the rows after it), "./." when the posterior is below min_prob (507-519);
 *   - the allele letters of earlier rows stay in allele_char[] (266-268 are outside the loop).
 * The reference builds ALT with sprintf (s, "%s,...", s, ...): on glibc that appends, and appending is what this program does.
 * With three arguments the reference reads argv[3] == NULL; this program takes min_prob = 0 then.
 *
 * Plain C on the host; nothing here touches the GPU. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <time.h>
#include <zlib.h>

#define TOK "\n\t "

static char *
gz_line (gzFile g, size_t *cap, char **buf)
{
  size_t n = 0;
  (*buf)[0] = '\0';
  for (;;)
    {
      if (!gzgets (g, *buf + n, (int) (*cap - n)))
        break;
      n += strlen (*buf + n);
      if (n && (*buf)[n - 1] == '\n')
        break;
      if (*cap - n < 2)
        *buf = (char *) realloc (*buf, *cap *= 2);
    }
  return *buf;
}

static void
append (char *dst, const char *fmt, ...)
  __attribute__ ((format (printf, 2, 3)));
#include <stdarg.h>
static void
append (char *dst, const char *fmt, ...)
{
  va_list ap;
  va_start (ap, fmt);
  vsprintf (dst + strlen (dst), fmt, ap);
  va_end (ap);
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
      const double tp = atof (argv[3]);
      if (tp >= 0.0 && tp <= 1.0)
        min_prob = tp;
    }
  char sdxname[4096];
  strncpy (sdxname, argv[1], sizeof sdxname - 8);
  sdxname[sizeof sdxname - 8] = '\0';
  FILE *sf = fopen (sdxname, "r");
  if (!sf)
    {
      printf ("\n Can not open file %s\n", sdxname);
      return 1;
    }
  if (strstr (sdxname, ".sdx"))
    for (int i = (int) strlen (sdxname) - 1; i > 0; i--)
      if (sdxname[i] == '.')
        {
          sdxname[i] = '\0';
          break;
        }
  char line[4200];
  fgets (line, 256, sf);
  const int no_contigs = atoi (line);
  unsigned int *contig_starts = (unsigned int *) calloc ((size_t) no_contigs + 2, sizeof (unsigned int));
  char **contig_names = (char **) calloc ((size_t) no_contigs + 1, sizeof (char *));
  for (int i = 0; i < no_contigs; i++)
    {
      if (!fgets (line, 1024, sf))
        line[0] = '\0';
      char *t = strtok (line, "\t \n");
      contig_starts[i + 1] = t ? (unsigned int) atoi (t) : 0u;
      t = strtok (NULL, "\t \n");
      contig_names[i] = strdup (t ? t : "");
    }
  fclose (sf);
  for (int i = 1; i <= no_contigs; i++)
    contig_starts[i] += contig_starts[i - 1];
  const long genome_size = (long) contig_starts[no_contigs] + 15L * no_contigs;
  char *genome = (char *) calloc ((size_t) genome_size + 4200, 1);
  if (!genome)
    {
      fprintf (stderr, "\n Failed to allocate memory for the Genome Buffer \n");
      return 1;
    }
  {
    char seqname[4200];
    sprintf (seqname, "%s.seq", sdxname);
    gzFile rf = gzopen (seqname, "r");
    if (!rf)
      {
        printf ("\n Can not open file %s for reading\n", seqname);
        return 1;
      }
    gzbuffer (rf, 1 << 25);
    long count = 0;
    while (count < genome_size)
      {
        const long want = genome_size - count < (1L << 28) ? genome_size - count : (1L << 28);
        const int got = gzread (rf, genome + count, (unsigned) want);
        if (got <= 0)
          break;
        count += got;
      }
    gzclose (rf);
  }
  for (int i = 1, j = 15; i <= no_contigs; i++, j += 15)
    contig_starts[i] += (unsigned int) j;

  printf ("##fileformat=VCFv4.0\n");
  {
    time_t now = time (NULL);
    struct tm tm = *localtime (&now);
    printf ("##fileDate=%d%d%d\n", tm.tm_year + 1900, tm.tm_mon + 1, tm.tm_mday);
  }
  printf ("##reference=%s\n", argv[1]);
  printf ("##phasing=none");
  printf ("##INFO=<ID=NS,Number=1,Type=Integer,Description=\"Number of Samples With Data\">\n");
  printf ("##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype Quality\">\n");
  printf ("##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n");
  printf ("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT");
  gzFile snp = gzopen (argv[2], "r");
  if (!snp)
    {
      printf ("\n Can not open file %s for reading\n", argv[2]);
      return 1;
    }
  gzbuffer (snp, 1 << 25);
  size_t cap = 1 << 20;
  char *buffer = (char *) malloc (cap);
  gz_line (snp, &cap, &buffer);
  int tot_samples = 0;
  char *token = strtok (buffer, TOK);
  for (int i = 0; i < 6; i++)
    token = strtok (NULL, TOK);
  while (token)
    {
      printf ("\t%s", token);
      tot_samples++;
      token = strtok (NULL, TOK);
    }
  gz_line (snp, &cap, &buffer);
  size_t len = strlen (buffer);

  /* the tables that live across rows */
  static char call_map[256][24];
  static char het_map[256][256];
  for (int j = 0; j < 256; j++)
    strcpy (call_map[j], "./.");
  memset (het_map, 'N', sizeof het_map);
  {
    static const char *pairs[] = { "ACM", "AGR", "ATW", "ADE", "AIH", "CGS", "CTY", "CAM", "CDE", "CIH", "GTK", "GAR", "GCS", "GDE", "GIH",
      "TAW", "TCY", "TGK", "TDE", "TIH", "DAE", "DCE", "DGE", "DTE", "DIE", "IAH", "ICH", "IGH", "ITH", "IDH"
    };
    for (size_t k = 0; k < sizeof pairs / sizeof pairs[0]; k++)
      het_map[(int) pairs[k][0]][(int) pairs[k][1]] = pairs[k][2];
  }
  char allele_char[30];
  memset (allele_char, 'N', sizeof allele_char);
  char *last_chr = (char *) malloc (cap);
  strcpy (last_chr, "!!!!!!");
  int last_chr_no = 0;
  char *chrom = (char *) malloc (cap), *alt_tmp = (char *) malloc (cap), *alt_final = (char *) malloc (2 * cap + 8400), *ref_string =
    (char *) malloc (8400), *gb = (char *) malloc (2 * cap + 8400), *sn = (char *) malloc (cap + 8400);

  size_t row_cap = cap;
#define FIND_CHROM() do { if (strcmp (chrom, last_chr) != 0) { last_chr_no = -1; \
    for (int c_ = 0; c_ < no_contigs; c_++) if (strcmp (chrom, contig_names[c_]) == 0) { strcpy (last_chr, chrom); last_chr_no = c_; break; } \
    if (last_chr_no < 0) { printf ("\n Failed to find chrom = %s \n", chrom); return 1; } } } while (0)
  /* (a position that points outside the .seq reads as an empty reference there, not as foreign memory) */
#define GENOME_AT(off) ((off) >= 0 && (off) < genome_size ? (off) : genome_size)

  while (len > 5)
    {
      if (cap != row_cap)
        {
          /* (the row's pieces never exceed the row) */
          row_cap = cap;
          chrom = (char *) realloc (chrom, cap);
          last_chr = (char *) realloc (last_chr, cap);
          alt_tmp = (char *) realloc (alt_tmp, cap);
          alt_final = (char *) realloc (alt_final, 2 * cap + 8400);
          gb = (char *) realloc (gb, 2 * cap + 8400);
          sn = (char *) realloc (sn, cap + 8400);
        }
      token = strtok (buffer, TOK);
      strcpy (chrom, token ? token : "");
      token = strtok (NULL, TOK);
      int pos = token ? atoi (token) : 0;
      token = strtok (NULL, TOK);
      char ref = token ? token[0] : 'N';
      token = strtok (NULL, TOK);
      strcpy (alt_tmp, token ? token : "");
      token = strtok (NULL, TOK);
      token = strtok (NULL, TOK);
      if (!token)
        break;
      const int drop_it = strcmp (token, "LOW") == 0 || strcmp (token, "MESS") == 0;
      if (!drop_it)
        {
          static const char hom[] = "ACGTID", het[] = "YRSWKMEH";
          for (const char *c = hom; *c; c++)
            strcpy (call_map[(int) *c], "1/1");
          for (const char *c = het; *c; c++)
            strcpy (call_map[(int) *c], "0/1");
          strcpy (call_map[(unsigned char) ref], "0/0");
          char slabel[16] = "PASS";
          sprintf (ref_string, "%c", ref);
          alt_final[0] = '\0';
          allele_char[0] = ref;
          if (strcmp (token, "SNP") == 0)
            {
              const char a = alt_tmp[0] == ref ? alt_tmp[2] : alt_tmp[0];
              sprintf (alt_final, "%c", a);
              strcpy (call_map[(unsigned char) a], "1/1");
              allele_char[1] = a;
            }
          else if (strcmp (token, "MULTIALLELIC") == 0)
            {
              int this_a = 1, this_a_pos = 0, has_del = 0;
              const int this_stop = (int) strlen (alt_tmp);
              while (this_a_pos < this_stop && this_a < 29)
                {
                  if (alt_tmp[this_a_pos] == ref)
                    this_a_pos += 2;
                  else if (alt_tmp[this_a_pos] == '+')
                    {
                      allele_char[this_a] = 'I';
                      snprintf (call_map[(int) 'I'], sizeof call_map[0], "%d/%d", this_a, this_a);
                      sprintf (call_map[(int) 'H'], "0/%d", this_a);
                      if (!has_del)
                        {
                          if (this_a == 1)
                            sprintf (alt_final, "%c", ref);
                          else
                            append (alt_final, ",%c", ref);
                        }
                      else
                        append (alt_final, ",%s", ref_string);
                      this_a_pos++;
                      while (this_a_pos < this_stop && alt_tmp[this_a_pos] != ',')
                        {
                          if (isalpha ((unsigned char) alt_tmp[this_a_pos]))
                            append (alt_final, "%c", alt_tmp[this_a_pos]);
                          this_a_pos++;
                        }
                      this_a_pos++;
                      this_a++;
                      strcpy (slabel, ".");
                    }
                  else if (alt_tmp[this_a_pos] == '-')
                    {
                      allele_char[this_a] = 'D';
                      snprintf (call_map[(int) 'D'], sizeof call_map[0], "%d/%d", this_a, this_a);
                      sprintf (call_map[(int) 'E'], "0/%d", this_a);
                      FIND_CHROM ();
                      pos--;
                      const long this_offset = GENOME_AT ((long) pos + (long) contig_starts[last_chr_no] - 1);
                      has_del = 1;
                      ref = genome[this_offset];
                      this_a_pos++;
                      int i = 0;
                      while (this_a_pos < this_stop && alt_tmp[this_a_pos] != ',')
                        sn[i++] = alt_tmp[this_a_pos++];
                      sn[i] = '\0';
                      int del_len = atoi (sn) + 1;
                      if (del_len > 4190)
                        del_len = 4190;         /* (the reference's gb[] holds 4196 characters) */
                      if (del_len < 0)
                        del_len = 0;
                      strncpy (ref_string, &genome[this_offset], (size_t) del_len);
                      ref_string[del_len] = '\0';
                      if (this_a == 1)
                        sprintf (alt_final, "%c", ref);
                      else
                        {
                          /* the alleles so far were single letters "X,Y,...": each becomes the deleted stretch with its second letter
                             replaced (364-378) */
                          strcpy (gb, alt_final);
                          const size_t gl = strlen (gb);
                          strcpy (sn, ref_string);
                          sn[1] = gb[0];
                          sprintf (alt_final, "%s", sn);
                          for (int a = 2, j = 2; a < this_a; a++, j += 2)
                            {
                              strcpy (sn, ref_string);
                              sn[1] = (size_t) j < gl ? gb[j] : '\0';
                              append (alt_final, ",%s", sn);
                            }
                          append (alt_final, ",%c", ref);
                        }
                      this_a_pos++;
                      this_a++;
                      strcpy (slabel, ".");
                    }
                  else
                    {
                      const char alt_a = alt_tmp[this_a_pos];
                      sprintf (call_map[(unsigned char) alt_a], "%d/%d", this_a, this_a);
                      allele_char[this_a] = alt_a;
                      for (int i = 0; i <= this_a; i++)
                        for (int j = i + 1; j <= this_a; j++)
                          sprintf (call_map[(unsigned char) het_map[(unsigned char) allele_char[i]][(unsigned char) allele_char[j]]], "%d/%d", i, j);
                      if (this_a == 1)
                        sprintf (alt_final, "%c", alt_a);
                      else
                        append (alt_final, ",%c", alt_a);
                      this_a++;
                      this_a_pos += 2;
                    }
                }
            }
          else if (strcmp (token, "INS") == 0)
            {
              const int mono = strchr (alt_tmp + (alt_tmp[0] ? 1 : 0), ',') == NULL;
              const size_t al = strlen (alt_tmp), skip = mono ? 1 : 3;
              sprintf (alt_final, "%c%s", ref, al >= skip ? alt_tmp + skip : "");
              strcpy (slabel, ".");
            }
          else                  /* a deletion, and every DENOVO_* row */
            {
              FIND_CHROM ();
              pos--;
              const long this_offset = GENOME_AT ((long) pos + (long) contig_starts[last_chr_no] - 1);
              ref = genome[this_offset];
              const int mono = strchr (alt_tmp + (alt_tmp[0] ? 1 : 0), ',') == NULL;
              const size_t al = strlen (alt_tmp), skip = mono ? 1 : 3;
              int del_len = atoi (al >= skip ? alt_tmp + skip : "") + 1;
              if (del_len > 8190)
                del_len = 8190;                 /* (the reference's ref_string[] holds 8196 characters) */
              if (del_len < 0)
                del_len = 0;
              strncpy (ref_string, &genome[this_offset], (size_t) del_len);
              ref_string[del_len] = '\0';
              strcpy (slabel, ".");
              sprintf (alt_final, "%c", ref);
            }
          printf ("\n%s\t%d\t.\t%s\t%s\t.\t%s\tNS=%d\tGT:GQ", chrom, pos, ref_string, alt_final, slabel, tot_samples);
          for (int i = 0; i < tot_samples; i++)
            {
              token = strtok (NULL, TOK);
              char *token2 = strtok (NULL, TOK);
              if (!token || !token2)
                break;          /* (a short row: the reference reads through NULL here) */
              if (atof (token2) >= min_prob)
                printf ("\t%s", call_map[(unsigned char) token[0]]);
              else
                printf ("\t./.");
              printf (":%s", token2);
            }
        }
      buffer[0] = '\0';
      if (!gzeof (snp))
        gz_line (snp, &cap, &buffer);
      len = strlen (buffer);
    }
  printf ("\n");
  return 0;
}
