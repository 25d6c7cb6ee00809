/*
 * index_genome_main.c -- host program (plain C) with the dialogue and the output files of the reference's
 * index_genome_whole (src/index_genome_whole.c:93-354): it answers the same five prompts from stdin
 *
 *   Send Output to Screen or Disk? [S,D]            (D: then a log file name)
 *   Maximum Number of Contig Fasta Files to Process
 *   Please Enter Name For Fastaq File
 *   Basename to save compressed Genome and Indexes
 *   Will the target DNA be bisulfite converted?
 *
 * and writes <base>.sdx (contig count, "len-15 <tab> name" rows, 16), <base>.seq (gz of the upper-cased letters),
 * <base>.mdx (raw u32 positions grouped by 16-mer) and <base>.idx (gz of the 2^32+1 prefix counts).  The fasta is read
 * with the reference's rules (header: trailing non-alphanumerics dropped, white space -> '_'; every alphabetic character
 * of the other lines is a base; lines are taken in 255-character pieces as the reference's fgets does).  The k-mer table
 * is built on the GPU (pemap_dev_build_index: seconds, where the reference needs > 64 GB of host memory for anything
 * beyond a few Mbp) and copied back for the two index files.
 *
 * Decompressed, the four files equal the reference builder's byte for byte (tests/test_gpu_cli.py); the .idx is
 * deflated at level 1 instead of 6 (16 GiB of input).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <stdint.h>
#include <zlib.h>
#include "../../include/pemap_hip.h"

static FILE *logfile;

static void
read_var (const char *prompt, char *result)    /* index_genome_whole.c:860-879 */
{
  printf ("%s", prompt);
  result[0] = '\0';
  if (!fgets (result, 250, stdin))
    result[0] = '\0';
  size_t n = strlen (result);
  if (n)
    result[n - 1] = '\0';
  if (logfile)
    {
      char line[256];
      snprintf (line, sizeof line, "%s", prompt);
      for (char *q = line; *q; q++)
        if (*q == '\n')
          *q = '\0';
      fprintf (logfile, "\"%s\",%s\n", line, result);
    }
}

static void
die (const char *fmt, const char *arg)
{
  printf (fmt, arg);
  printf ("\n");
  exit (1);
}

int
main (void)
{
  char ss[4196], sss[4196], fasta_name[4196], basename[1024];
  read_var ("\nSend Output to Screen or Disk? [S,D]\n", ss);
  if (strchr (ss, 'D') || strchr (ss, 'd'))
    {
      read_var ("Please Enter File Name for Output\n", ss);
      if (!(logfile = fopen (ss, "a")))
        die ("\n Can not open file %s", ss);
    }
  read_var ("Maximum Number of Contig Fasta Files to Process\n", ss);
  const int N = atoi (ss) > 0 ? atoi (ss) : 1;
  read_var ("Please Enter Name For Fastaq File\n", fasta_name);
  FILE *in = fopen (fasta_name, "r");
  if (!in)
    die ("\n Can not open file %s", fasta_name);
  read_var ("Basename to save compressed Genome and Indexes\n", basename);
  snprintf (sss, sizeof sss, "%s.sdx", basename);
  FILE *sfile = fopen (sss, "w");
  if (!sfile)
    die ("\nCould Not Open file %s", sss);
  snprintf (ss, sizeof ss, "%s.seq", basename);
  gzFile seqfile = gzopen (ss, "w");
  if (!seqfile)
    die ("\nCould Not Open file %s", ss);
  gzbuffer (seqfile, 1 << 20);
  read_var ("Will the target DNA be bisulfite converted?\n", ss);
  const int is_bisulf = (strchr (ss, 'Y') || strchr (ss, 'y')) ? 1 : 0;

  /* ---- the fasta, with the reference's line handling (index_genome_whole.c:191-303) */
  size_t cap = 1 << 26, gsize = 0;
  char *genome = (char *) malloc (cap);
  char **names = (char **) calloc ((size_t) N + 1, sizeof (char *));
  uint32_t *contig_len = (uint32_t *) calloc ((size_t) N + 1, sizeof (uint32_t));
  int fasta = -1;
  if (!fgets (sss, 4195, in))
    sss[0] = '\0';
  int not_done = 1;
  while (not_done)
    {
      if (sss[0] == '>')
        {
          fasta++;
          if (fasta >= N)
            die ("\n index_genome_hip: more contigs in %s than the maximum given", fasta_name);
          unsigned j = (unsigned) strlen (sss);
          while (j > 1 && !isalnum ((unsigned char) sss[j]))
            {
              sss[j] = '\0';
              j--;
            }
          names[fasta] = (char *) calloc (j + 2, 1);
          for (unsigned i = 1; i <= j; i++)
            names[fasta][i - 1] = isspace ((unsigned char) sss[i]) ? '_' : sss[i];
          names[fasta][j] = '\0';
          if (!fgets (sss, 256, in))
            sss[0] = '\0';
        }
      for (size_t i = 0, j = strlen (sss); i < j; i++)
        if (isalpha ((unsigned char) sss[i]))
          {
            if (fasta < 0)
              die ("\n index_genome_hip: %s does not start with a '>' line", fasta_name);
            if (gsize + 1 > cap)
              genome = (char *) realloc (genome, cap *= 2);
            const char c = (char) toupper ((unsigned char) sss[i]);
            genome[gsize++] = c;
            contig_len[fasta]++;
          }
      not_done = !feof (in);
      if (not_done)
        {
          sss[0] = '\0';
          if (!fgets (sss, 256, in) || strlen (sss) < 1)
            not_done = 0;
        }
    }
  fclose (in);
  fasta++;
  for (size_t o = 0; o < gsize;)
    {
      const unsigned n = (gsize - o) > (1u << 30) ? (1u << 30) : (unsigned) (gsize - o);
      if (gzwrite (seqfile, genome + o, n) <= 0)
        die ("\n write error on %s.seq", basename);
      o += n;
    }
  gzclose (seqfile);

  /* ---- the k-mer table on the GPU */
  pemap_dev *gpu;
  if (pemap_dev_create (&gpu, getenv ("PEMAP_DEVICE") ? atoi (getenv ("PEMAP_DEVICE")) : 0))
    die ("\n index_genome_hip: %s", pemap_dev_last_error (NULL));
  if (pemap_dev_build_index (gpu, genome, gsize, contig_len, fasta, is_bisulf))
    die ("\n index_genome_hip: %s", pemap_dev_last_error (gpu));
  uint64_t n_mers = 0, gs = 0;
  int nc = 0, idepth = 16;
  pemap_dev_index_info (gpu, &n_mers, &gs, &nc, &idepth);

  const size_t CH = (size_t) 1 << 28;   /* bytes per copy */
  char *buf = (char *) malloc (CH);
  snprintf (ss, sizeof ss, "%s.mdx", basename);
  FILE *mfile = fopen (ss, "w");
  if (!mfile)
    die ("\nCould Not Open file %s", ss);
  for (uint64_t o = 0; o < n_mers * 4; o += CH)
    {
      const uint64_t n = n_mers * 4 - o < CH ? n_mers * 4 - o : CH;
      if (pemap_dev_read_buffer (gpu, 1, o, buf, n))
        die ("\n index_genome_hip: %s", pemap_dev_last_error (gpu));
      fwrite (buf, 1, n, mfile);
    }
  fclose (mfile);
  snprintf (ss, sizeof ss, "%s.idx", basename);
  gzFile ifile = gzopen (ss, "w1");
  if (!ifile)
    die ("\nCould Not Open file %s", ss);
  gzbuffer (ifile, 1 << 20);
  const uint64_t idx_bytes = (((uint64_t) 1 << 32) + 1) * 4;
  for (uint64_t o = 0; o < idx_bytes; o += CH)
    {
      const uint64_t n = idx_bytes - o < CH ? idx_bytes - o : CH;
      if (pemap_dev_read_buffer (gpu, 0, o, buf, n))
        die ("\n index_genome_hip: %s", pemap_dev_last_error (gpu));
      if (gzwrite (ifile, buf, (unsigned) n) <= 0)
        die ("\n write error on %s.idx", basename);
    }
  gzclose (ifile);
  fprintf (sfile, "%d\n", fasta);
  for (int i = 0; i < fasta; i++)
    fprintf (sfile, "%d\t%s\n", (int) contig_len[i] - 15, names[i]);
  fprintf (sfile, "%d\n", idepth);
  fclose (sfile);
  if (logfile)
    fclose (logfile);
  pemap_dev_destroy (gpu);
  return 0;
}
