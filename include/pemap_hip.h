/*
 * pemap_hip.h -- C-ABI of the MI355X (gfx950) PEMapper / PECaller hot path.
 *
 * Plain C: opaque handles, plain pointers and sizes, int return codes (0 = ok, non-zero = error; the text is in
 * pemap_dev_last_error()).  Nothing here depends on torch or on C++ types.  A host program written in C (the
 * reference's language) links libpemap_hip.so and calls these; pecaller_amd/csrc/pemapper_main.c does exactly that.
 *
 * What each entry replaces in the reference (wingolab-org/pecaller, paths under src/):
 *
 *   pemap_dev_create / _destroy      the process-wide globals of pemapper.c:140-176 (one object per GPU instead)
 *   pemap_dev_load_index             init_index_buffer + genome/.sdx load, pemapper.c:411-494, 2129-2155
 *   pemap_dev_build_index            index_genome_whole.c:93-354 (rolling 16-mer, N resets, len-15 coordinates), on device
 *   pemap_dev_set_params             argv parsing of max_dist/min_dist/is_bisulfite/min_match, pemapper.c:233-297
 *   pemap_dev_map_batch              pthread_create(..., map_everything, batch) + the result fold,
 *                                    pemapper.c:684, 759, 907-1309 (initial_map, find_matches, smith_waterman_align,
 *                                    find_mate_pairs, smith_waterman_backtrack)
 *   pemap_dev_submit_batch / _wait_batch  the same seam, asynchronous like the reference's (pthread_create returns at once, the
 *                                    batch mutex is released at pemapper.c:1307)
 *   pemap_dev_stage_reads/_run/_collect   the same call split in three, so that a caller can time the device part
 *   pemap_dev_fetch_pileup           the final genome walk that feeds the pileup / indel writers, pemapper.c:819-866
 *   pemap_dev_summary                total_reads/total_bases/total_dist/no_dists/mate_counts, pemapper.c:144-149, 1238-1265
 *   pecall_dev_*                     fill_sample_like and its callers, pecaller.c:2448-2507 (see below)
 *
 * Threading: calls on different pemap_dev objects may run concurrently from different host threads; calls on one
 * object must be serialised by the caller (the reference serialises a batch behind its own mutex, pemapper.c:661),
 * except pemap_dev_submit_batch / _wait_batch / _map_batch, which several host threads may call on one object.
 * Ownership: the caller owns every host buffer and may reuse it as soon as a call returns; the object owns all
 * device memory.
 */
#ifndef PEMAP_HIP_H
#define PEMAP_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PEMAP_MAX_HITS   200    /* max_hits, pemapper.c:162 */
#define PEMAP_MIN_READ   16     /* one full 16-mer segment (pemapper.c:1573-1587 reads garbage below this) */
#define PEMAP_MAX_READ   278    /* L + 21 window rows must fit the reference's 300 x 300 matrices (pemapper.c:155, 932-956) */

/* mapping_type values, pemapper.c:37-45 */
enum
{
  PEMAP_UNIQUE_MATE = 0, PEMAP_UNIQUE_SLIP = 1, PEMAP_UNIQUE_SINGLE = 2, PEMAP_UNIQUE_MIS = 3, PEMAP_NON_MATE = 4,
  PEMAP_NON_MIS = 5, PEMAP_FRAG_MIS = 6, PEMAP_NON_NO = 7, PEMAP_NEITHER_MAP = 8
};

typedef struct pemap_dev pemap_dev;

/* insertion callback for pemap_dev_fetch_pileup: pos = 0-based index into .seq, seq = inserted bases in read order */
typedef void (*pemap_ins_cb) (void *user, uint32_t pos, const char *seq, int len);

const char *pemap_dev_last_error (const pemap_dev * dev);       /* dev may be NULL: error of the last failed create */

int pemap_dev_create (pemap_dev ** out, int device_id);
void pemap_dev_destroy (pemap_dev * dev);

/* Index from the reference's on-disk arrays already in host memory.  pos_index has 2^32 + 1 entries (.idx inflated),
 * mers has n_mers entries (.mdx), genome is the upper-cased .seq stream, contig_starts has n_contigs + 1 entries:
 * prefix sums of the .sdx "len-15" column (pemapper.c:434-448).  idepth is the last .sdx line (16). */
int pemap_dev_load_index (pemap_dev * dev, const uint32_t * pos_index, const uint32_t * mers, uint64_t n_mers,
                          const char *genome, uint64_t genome_size, const uint32_t * contig_starts, int n_contigs,
                          int idepth);

/* Index built on the device from the concatenated genome letters (what .seq holds) and the contig lengths
 * (full lengths, not len-15).  Produces exactly the arrays the reference builder writes.  bisulfite = the builder's
 * "bisulfite converted" answer (C indexed as T, index_genome_whole.c:174-177). */
int pemap_dev_build_index (pemap_dev * dev, const char *genome, uint64_t genome_size, const uint32_t * contig_len,
                           int n_contigs, int bisulfite);

/* Same, from letters already in device memory (d_genome is a device pointer; the object takes a copy). */
int pemap_dev_build_index_resident (pemap_dev * dev, const void *d_genome, uint64_t genome_size,
                                    const uint32_t * contig_len, int n_contigs, int bisulfite);

/* Multi-GPU replica set-up: allocate empty index arrays of the given sizes so that a collective (RCCL broadcast
 * from the rank that loaded or built the index) can fill them, then pemap_dev_index_commit(). */
int pemap_dev_index_alloc (pemap_dev * dev, uint64_t n_mers, uint64_t genome_size, int n_contigs, int idepth);
int pemap_dev_index_commit (pemap_dev * dev);

/* Look-up replicas -- an MI355X-side layout with no counterpart in the reference.  The look-ups of a read-end
 * (2 x segments x 49 buckets, fill_mers / get_mers, pemapper.c:1969-2003, 2158-2165) cost one 64-byte HBM request each
 * in the reference's table; index_commit therefore also builds 8 re-ordered copies of the table (128 GiB + the records
 * of the multi-position buckets) in which a k-mer and its 48 neighbours share 8 lines.  Results are identical either
 * way.  n = -1: build them when the device has the memory (default; PEMAP_REPLICAS=0/1 in the environment overrides),
 * 0: never, 8: fail if they cannot be built.  May be called before or after the index is in place. */
int pemap_dev_set_lookup_replicas (pemap_dev * dev, int n);
/* how many replicas serve the look-ups now (0 or 8), and the bytes of the multi-position records */
int pemap_dev_lookup_replicas (pemap_dev * dev, int *n_replicas, uint64_t * record_bytes);

/* Device pointers and sizes of the resident arrays: which = 0 pos_index (u32[2^32+1]), 1 mers (u32[n_mers]),
 * 2 genome (u8[genome_size]), 3 contig_starts (u32[n_contigs+1]), 4 pileup counters (six planes A C G T Del Ins of 16-bit
 * wrapping counters, two positions to a 32-bit word, each plane padded to 256-byte blocks: n_bytes / 6 bytes per plane),
 * 5 the look-up replicas (u32[8][2^32]; replica p holds the entry of k-mer k at k with its 4-bit fields 0 and p
 * swapped), 6 the records of the multi-position buckets (16-byte units {count, positions...}); 5 and 6 are empty
 * (n_bytes = 0) when no replicas are in use.
 * For collectives (broadcast of 0..3 at start-up, sum of 4 at the end) and for copying the index back to the host. */
int pemap_dev_buffer (pemap_dev * dev, int which, void **d_ptr, uint64_t * n_bytes);
int pemap_dev_index_info (pemap_dev * dev, uint64_t * n_mers, uint64_t * genome_size, int *n_contigs, int *idepth);
/* copy one of the buffers above (or a byte range of it) to host memory */
int pemap_dev_read_buffer (pemap_dev * dev, int which, uint64_t byte_offset, void *host_dst, uint64_t n_bytes);

int pemap_dev_set_params (pemap_dev * dev, int paired, int min_dist, int max_dist, double min_align, int bisulfite);

/* One batch.  reads are `stride`-spaced byte rows (not necessarily NUL terminated), len1/len2 their lengths,
 * PEMAP_MIN_READ <= len <= PEMAP_MAX_READ, reads2/len2/m2 NULL in single-end mode.  m1/m2 receive
 * 0 (unmapped) or the reference's coordinate (0-based index of the last aligned reference base, + 2),
 * mapping_type the class.  The pileup and the summary counters accumulate on the object. */
int pemap_dev_map_batch (pemap_dev * dev, const char *reads1, const int *len1, const char *reads2, const int *len2,
                         int n, int stride, uint32_t * m1, uint32_t * m2, int *mapping_type);

/* The same call split at the point where the reference's reader thread lets go of a batch: pthread_create (pemapper.c:684)
 * returns at once and the batch's mutex is released when the worker is done (1307).  submit queues the batch (host-to-device
 * copies on a copy stream, kernels on the object's pipeline, device-to-host copy of the results) and returns a ticket; wait
 * blocks until that batch's m1 / m2 / mapping_type are in the buffers given to submit and its summary counters are folded.
 * Up to 3 batches are in flight per object; a 4th submit first delivers the oldest.  With at least 2 in flight the copies hide
 * under the kernels and the kernel pipeline never drains between batches: the host sees the device-resident rate.
 * The caller's buffers (reads, lengths, results) belong to the library from submit until the batch is delivered.
 * submit / wait / map_batch may be called from several host threads on one object (they serialise on an internal lock
 * that is released while a thread blocks in wait); everything else on the object still needs the caller's serialisation.
 * pemap_dev_map_batch == submit + wait. */
int pemap_dev_submit_batch (pemap_dev * dev, const char *reads1, const int *len1, const char *reads2, const int *len2,
                            int n, int stride, uint32_t * m1, uint32_t * m2, int *mapping_type, uint64_t * ticket);
int pemap_dev_wait_batch (pemap_dev * dev, uint64_t ticket);
/* Optional: page-lock a host range that read rows will be submitted from, so that they move by DMA straight out of it (the
 * reference allocates its batch buffers once, pd_node_alloc at pemapper.c:2340-2372, and reuses them: pin them once after
 * that, unpin before freeing them).  Rows submitted from memory that is not pinned are first copied into a pinned staging
 * buffer of the library (a few host threads, ~10 ms per million pairs).  The library never page-locks caller memory by itself. */
int pemap_dev_pin_host (pemap_dev * dev, const void *host_ptr, uint64_t n_bytes);
int pemap_dev_unpin_host (pemap_dev * dev, const void *host_ptr);

/* The same in three steps.  stage: host -> device copy of a batch; run: the kernels (asynchronous on the object's
 * stream unless sync != 0); collect: device -> host copy of the results + summary fold.  After stage, run may be
 * called repeatedly (each run maps the staged batch again and adds to the pileup): that is what bench.py times. */
int pemap_dev_stage_reads (pemap_dev * dev, const char *reads1, const int *len1, const char *reads2, const int *len2,
                           int n, int stride);
int pemap_dev_run (pemap_dev * dev, int sync);
/* map only rows [first, first + n) of the staged reads (one bench "step" = one slice of a resident read set) */
int pemap_dev_run_slice (pemap_dev * dev, int first, int n, int sync);
int pemap_dev_collect (pemap_dev * dev, uint32_t * m1, uint32_t * m2, int *mapping_type);
int pemap_dev_sync (pemap_dev * dev);

/* Synthetic workload, generated on the device (bench and scale tests): a seeded genome of
 * n_contigs contigs (repeat families, N runs at contig ends), and a staged batch of n read pairs of length read_len
 * sampled from the resident genome with substitutions/indels.  synth_genome fills contig_len[n_contigs]. */
int pemap_dev_synth_genome (pemap_dev * dev, uint64_t seed, uint64_t genome_size, int n_contigs, double repeat_frac,
                            void **d_genome, uint32_t * contig_len);
int pemap_dev_synth_reads (pemap_dev * dev, uint64_t seed, int n, int read_len, int paired, double sub_rate,
                           double indel_rate, uint64_t first_read);
/* the same with ONE insertion or deletion of 1..10 bases in the given share of the read-ends instead of a per-base indel rate
 * (BASELINE config "5 % indel-enriched 2x250bp reads") */
int pemap_dev_synth_reads_indel (pemap_dev * dev, uint64_t seed, int n, int read_len, int paired, double sub_rate,
                                 double indel_read_frac, uint64_t first_read);
/* copy the staged batch back (for the CPU baseline): reads as stride-spaced rows */
int pemap_dev_staged_reads (pemap_dev * dev, char *reads1, int *len1, char *reads2, int *len2, int stride);
int pemap_dev_staged_info (pemap_dev * dev, int *n, int *stride, int *paired);
/* release a device allocation handed out by pemap_dev_synth_genome */
int pemap_dev_free (pemap_dev * dev, void *d_ptr);

/* counts[genome_size][6] u16 (A,C,G,T,Del,Ins), wrapping as the reference's unsigned short counters do
 * (pemapper.c:53-58); cb (may be NULL) is called once per logged insertion, in unspecified order. */
int pemap_dev_fetch_pileup (pemap_dev * dev, uint16_t * counts, pemap_ins_cb cb, void *user);
/* compacted 16-byte records {u32 pos; u16 A,C,G,T,Del,Ins} of the non-zero sites in [first, first+count), ascending;
 * returns the number of records through n_records (out may be NULL to count only). */
int pemap_dev_fetch_records (pemap_dev * dev, uint64_t first, uint64_t count, void *out, uint64_t out_capacity,
                             uint64_t * n_records);
int pemap_dev_reset_pileup (pemap_dev * dev);

/* out13: total_reads, total_bases, total_dist, no_dists, mate_counts[0..8] */
int pemap_dev_summary (pemap_dev * dev, long *out13);

/* Counters the reference does not have, summed over the last run --
 * stats[0] = read-ends, [1] = positions gathered from .mdx (P), [2] = SW problems scored (H), [3] = SW problems
 * scored with direction nibbles (single-hit ends + re-scored winners), [4] = DP cells without nibbles, [5] = DP cells
 * with nibbles, [6] = pileup increments, [7] = insertions logged, [8] = alignments walked back, [9] = winners re-scored,
 * [10] = read-ends the fused seed kernel's first tier passed over (more positions than its list holds, or too many next to candidate
 *        anchors): taken by its second tier; [15] = of them, the ones left to the monolithic seed kernel,
 * [11] = chunks the run was cut into (= launches of every kernel),
 * [12] = problems decided without the DP (a diagonal with at most one mismatch; with PEMAP_GAPLESS=0 none):
 *        [3], [5] count the DP's share only.
 * [13] = problems scored by the DP restricted to a band of 32 diagonals (exact for them: pemap_band.hip.h), part of [2] and,
 *        for the single-hit ends, of [3]; [14] = band cells computed (not part of [4], [5]).
 * times_ms[0..7] = seed stage (look-up + vote), SW single-hit (with nibbles), SW multi-hit, select, SW re-score,
 * walk+pileup, look-up kernel alone, vote kernel alone (the list-mode remainder of the big read-ends and the emit kernel count
 * towards [0] only): kernel durations from HIP events on the object's streams, summed over the run's chunks (the streams of
 * the pipeline overlap in time, so their sum exceeds the wall time). */
int pemap_dev_run_stats (pemap_dev * dev, uint64_t * stats16, float *times_ms8);

/* Debug/parity taps: per read-end hit lists and per-hit SW results of the last run.
 * n_hits[n_ends]; the other arrays are [n_ends][PEMAP_MAX_HITS]. Any pointer may be NULL. end = 2*pair + mate in paired mode. */
int pemap_dev_debug_hits (pemap_dev * dev, int *n_hits, uint32_t * spot, uint8_t * orient, uint32_t * win_start,
                          int *win_len, double *score, int *start_k, int *start_i);

/* ---- PECaller: per-(site, sample) Dirichlet-multinomial genotype log-likelihood ----
 * fill_sample_like (pecaller.c:2448-2507) together with the per-sample set-up it depends on (pecaller.c:1230-1260:
 * tot = A+C+G+T+Del, coef = ln tot! - sum ln reads[i]! over all six counts).
 *   reads[n_sites][indiv][6] u16        one pileup column per sample, the 6 counters of the pileup record
 *   alpha_mean[n_sites][14][6] f64      d_alpha_mean of the pass (pecaller.c:1354-1364)
 *   max_gen / min_depth                 14 / 2 diploid, as set at pecaller.c:326-336 for haploid
 *   norm                                new_norm[pass] (pecaller.c:1339-1344)
 *   like[n_sites][indiv][14]            log-likelihoods (0 where the reference skips the sample: tot <= min_depth)
 *   best[n_sites][indiv]  (may be NULL) initial_call, 14 where skipped;  margin (may be NULL) initial_p
 * The confidence ordering (qsort at 2506) and the configuration search stay on the host. */
typedef struct pecall_dev pecall_dev;
int pecall_dev_create (pecall_dev ** out, int device_id);
void pecall_dev_destroy (pecall_dev * dev);
const char *pecall_dev_last_error (const pecall_dev * dev);
int pecall_dev_site_like (pecall_dev * dev, const uint16_t * reads, const double *alpha_mean, int n_sites, int indiv,
                          int max_gen, int min_depth, double norm, double *like, int8_t * best, double *margin);
/* the same call in three steps (host->device, kernel, device->host), so that the kernel can be timed on resident data */
int pecall_dev_stage (pecall_dev * dev, const uint16_t * reads, const double *alpha_mean, int n_sites, int indiv);
int pecall_dev_run (pecall_dev * dev, int n_sites, int indiv, int max_gen, int min_depth, double norm, int sync);
int pecall_dev_collect (pecall_dev * dev, int n_sites, int indiv, double *like, int8_t * best, double *margin);

/* ---- PECaller: the whole per-site caller ----
 * The body of call_single_base for one pileup column (pecaller.c:1207-1691), with or without a pedigree: site filters,
 * up to five passes of { Dirichlet means, fill_sample_like, confidence order, the beam over joint configurations
 * (fill_config_like / fill_config_probs / clean_config_probs with the exact Hardy-Weinberg prior), configuration
 * posteriors, per-sample marginal posteriors and calls, moment-matched alpha re-estimation + check_alpha_sanity }, then the
 * site classification of pecaller.c:1565-1636.
 *   reads[n_sites][indiv][6] u16     as above, samples in the reference's column order (it matters: ties, beam order)
 *   ref_base[n_sites]                0..3 = A C G T (gen_to_int of the .seq letter); anything else = a site the reference
 *                                    skips (pecaller.c:1208, 1718): calls 'N', posterior 1, site_type -1
 *   chrom_type[n_sites] (may be NULL) 0 autosome, 1 chrX, 2 chrY, 3 chrMT: the lower-cased contig-name prefix rule of
 *                                    pecaller.c:474-482 (chrY lifts the half-the-samples filter, 1303; X/Y/MT change add_denovo);
 *                                    + 16 = HAPLOID forced for the column, as the BED guide mode does on chrY / chrMT (955-957)
 *   haploid / threshold / theta      argv[7] / argv[5] (Prob_to_call) / argv[6] of pecaller
 *   call[n_sites][indiv]             0..13 = A C G T D I M R W S Y K E H (int_to_gen), 14 = 'N'
 *   posterior[n_sites][indiv]        final_p, what the reference prints with %g
 *   site_type[n_sites]  (may be NULL) 0 reference, 1 SNP, 2 DEL, 3 INS, 4 LOW, 5 MULTIALLELIC, 6 MESS
 *   allele_count[n_sites][6], n_pass[n_sites]  (may be NULL) Allele_Counts of the .snp row; passes run
 *   denovo[n_sites]     (may be NULL) d_count of the row (pecaller.c:1650-1671): > 0 = the type is printed as DENOVO_<type>
 * indiv <= 512: up to 64 samples one lane per sample (the fast case: shortcut kernel + beam search of the columns it lists);
 * 65..512 a lane stands for a sample of each chunk of 64 -- the shortcut kernel holds two chunks in registers (up to 128 samples) or works a
 * chunk at a time; the beam search of the columns it lists runs one wave per CU from 257 samples on.  Text formatting and
 * the merge of the pileup streams stay on the host.
 * The columns travel in chunks of 2^18 (PECALL_CHUNK_LOG2): the host-to-device copy of chunk k + 1, the kernels of chunk k and the
 * device-to-host copy of chunk k - 1 run side by side.  Arrays the caller page-locked with pecall_dev_pin_host are copied from and
 * to directly; the others pass through pinned staging buffers of the object (host copies on a few threads). */
int pecall_dev_call_sites (pecall_dev * dev, const uint16_t * reads, const uint8_t * ref_base, const uint8_t * chrom_type, long n_sites,
                           int indiv, int haploid, double threshold, double theta, int8_t * call, double *posterior,
                           int8_t * site_type, int32_t * allele_count, int8_t * n_pass, int32_t * denovo);
/* The same call with the posteriors as a list: nearly every column of real data has posterior 1 for every sample (the shortcut of
 * the pass loop), and of the 606 bytes a 64-sample column sends back 512 are these doubles -- at the rate of the PCIe link they are
 * what the call takes.  Here only the columns in which SOME posterior differs from 1 come back: post_site[k] = the column's number
 * (ascending), post_rows[k][indiv] = its samples' posteriors; every column that is not listed has posterior exactly 1 for every
 * sample.  post_cap = rows the two arrays hold; *n_post = columns listed.  More columns than post_cap: the call fails (its message
 * says so) with *n_post = the number needed; nothing else of the results is to be used then.  The other arrays are as above. */
int pecall_dev_call_sites_sparse (pecall_dev * dev, const uint16_t * reads, const uint8_t * ref_base, const uint8_t * chrom_type, long n_sites,
                                  int indiv, int haploid, double threshold, double theta, int8_t * call, uint32_t * post_site,
                                  double *post_rows, uint64_t post_cap, uint64_t * n_post, int8_t * site_type, int32_t * allele_count,
                                  int8_t * n_pass, int32_t * denovo);
/* Page-lock a host range the caller will hand to pecall_dev_call_sites again and again (its tile buffers): the same table of
 * ranges as pemap_dev_pin_host's; a range stays until its last unpin.  Unpin before freeing the memory. */
int pecall_dev_pin_host (pecall_dev * dev, const void *host_ptr, uint64_t n_bytes);
int pecall_dev_unpin_host (pecall_dev * dev, const void *host_ptr);
/* the same call in three steps (host->device, kernel, device->host): bench.py times the kernel on resident columns with them.
 * kernel_ms (may be NULL) receives the kernel's duration from HIP events on the object's stream. */
int pecall_dev_sites_stage (pecall_dev * dev, const uint16_t * reads, const uint8_t * ref_base, const uint8_t * chrom_type, long n_sites,
                            int indiv);
int pecall_dev_sites_run (pecall_dev * dev, int haploid, double threshold, double theta, float *kernel_ms);
int pecall_dev_sites_collect (pecall_dev * dev, int8_t * call, double *posterior, int8_t * site_type, int32_t * allele_count,
                              int8_t * n_pass, int32_t * denovo);
/* use_pedfile = y (pecaller.c:376-392, 561-604): parents as sample indices (-1 = not sampled), sex (1 male, 2 female), and each
 * sample's kids in ped-file order: kids of i = kid_list[kid_off[i] .. kid_off[i + 1]).  denovo_rate = argv[11] (<= theta).
 * The configuration prior then carries no_denovo * ln(denovo_rate) (add_denovo, 2396-2445, tables of main 312-374).
 * dad == NULL clears the pedigree. */
int pecall_dev_set_pedigree (pecall_dev * dev, int indiv, const int *dad, const int *mom, const int *sex, const int *kid_off,
                             const int *kid_list, double denovo_rate);

#ifdef __cplusplus
}
#endif
#endif
