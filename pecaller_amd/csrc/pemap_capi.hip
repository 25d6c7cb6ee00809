// pemap_capi.hip -- the C-ABI of include/pemap_hip.h on top of the gfx950 kernels.
//
// Built as libpemap_hip.so:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared ...
// There is no CPU fallback in this library: every entry either runs on the GPU or returns an error.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <unistd.h>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/pemap_hip.h"
#include "pemap_kernels.hip.h"
#include "pemap_aux.hip.h"
#include <rocprim/rocprim.hpp>

static char g_create_err[512] = "";
struct PmChunkCtr;

// ---- batches in flight (pemap_dev_submit_batch / pemap_dev_wait_batch).  The staged-read arrays of the object are cut in
// PM_RING slots of ring_cap rows; a submitted batch owns one slot from its host-to-device copy to the device-to-host copy of
// its m1 / m2 / mapping_type.  Lengths and results pass through pinned staging buffers of the slot; the read rows are copied
// straight from the caller's buffers (pinned through pemap_dev_pin_host, or registered on first use).
#define PM_RING 3
struct PmRingSlot
{
  bool active;
  unsigned long long seq;       // ticket of the batch that owns the slot
  int n, first;
  uint32_t *m1, *m2;            // the caller's result buffers
  int *mt;
  int *h_len;                   // pinned: len1 | len2
  uint32_t *h_res;              // pinned: m1 | m2 | mapping_type
  char *h_rows;                 // pinned staging for read rows submitted from pageable memory: reads1 | reads2 (allocated on first need)
  size_t h_rows_bytes;
  hipEvent_t ev_done;           // recorded behind the device-to-host copy of the results
  std::vector < hipEvent_t > ev_copy;   // one per slice: the slice's rows are on the device
};

// Host ranges page-locked through pemap_dev_pin_host.  HIP's registrations are process-wide and keyed by the pointer, so the
// table is too (one object must not undo what another still copies from); a range stays until its last unpin.  The library
// never registers memory on its own: a registration outlives the buffer it was made for, and any later copy of the process
// that touches part of such a stale range is refused by the runtime (a copy must lie inside ONE registration or outside all:
// tools/micro/hostreg.hip).  Rows submitted from memory that is not pinned go through the slot's own pinned staging buffer.
struct PmPinned
{
  char *base;
  size_t bytes;
  int users;                    // pemap_dev_pin_host calls not yet undone
};
static std::mutex g_pin_mu;
static std::vector < PmPinned > g_pinned;


// Tuning knobs.  The environment is read ONCE, by pemap_dev_create, into the object: nothing below
// calls getenv again, so two objects of one process may run with different settings and a setting cannot change under a
// run.  None is needed in normal use.  The two timing probes that cut kernels short (and so return wrong results) exist
// only in a library built with -DPEMAP_TIMING_PROBES, which the product build does not define.
struct PmKnobs
{
  int seed_blocks_per_cu, big_blocks_per_cu, sw_waves_per_cu;
  int replicas;                 // -1 unset, 0 never, 1 as the default
  int gapless;                  // 0 off, 1 first case only, 2 both
  double dir_budget_gb;
  int lookup_waves /* -1 unset */;
  int lookup_prio, vote_prio, sw_prio;
  int vote_waves;
  int walk_blocks_per_cu, pile_blocks_per_cu;
  int pipeline;
  int chunk_pairs;
  int gapless_blocks_per_cu;
  int band, band_waves_per_cu;  // the banded DP (pm_band_kernel) for the problems it is exact for; its waves per CU
  int seed_phase;               // always 0 without PEMAP_TIMING_PROBES
  int tier2_waves;              // persistent waves per CU of the fused seed kernel's second tier (the first tier's big-end list)
};

static int env_int (const char *name, int dflt)
{
  const char *e = getenv (name);
  return (e && *e) ? atoi (e) : dflt;
}

static void read_knobs (PmKnobs & k)
{
  k.seed_blocks_per_cu = env_int ("PEMAP_SEED_BLOCKS_PER_CU", 8);
  k.big_blocks_per_cu = env_int ("PEMAP_BIG_BLOCKS_PER_CU", 8);
  k.sw_waves_per_cu = env_int ("PEMAP_SW_WAVES_PER_CU", 16);
  // a grid of zero or fewer blocks is not a setting
  if (k.seed_blocks_per_cu < 1) k.seed_blocks_per_cu = 1;
  if (k.big_blocks_per_cu < 1) k.big_blocks_per_cu = 1;
  if (k.sw_waves_per_cu < 1) k.sw_waves_per_cu = 1;
  k.replicas = getenv ("PEMAP_REPLICAS") ? (env_int ("PEMAP_REPLICAS", 1) ? 1 : 0) : -1;
  k.gapless = env_int ("PEMAP_GAPLESS", 2);
  { const char *e = getenv ("PEMAP_DIR_BUDGET_GB"); k.dir_budget_gb = e ? atof (e) : 40.0; if (k.dir_budget_gb < 0.25) k.dir_budget_gb = 0.25; }
  k.lookup_waves = env_int ("PEMAP_LOOKUP_WAVES", -1);
  k.lookup_prio = env_int ("PEMAP_LOOKUP_PRIO", 0);
  k.tier2_waves = env_int ("PEMAP_TIER2_WAVES", 3);
  k.vote_prio = env_int ("PEMAP_VOTE_PRIO", 0);
  k.sw_prio = env_int ("PEMAP_SW_PRIO", 0);
  k.vote_waves = env_int ("PEMAP_VOTE_WAVES", 1024);
  if (k.vote_waves < 1) k.vote_waves = 1;
  k.walk_blocks_per_cu = env_int ("PEMAP_WALK_BLOCKS_PER_CU", 4);
  k.pile_blocks_per_cu = env_int ("PEMAP_PILE_BLOCKS_PER_CU", 8);
  if (k.walk_blocks_per_cu < 1) k.walk_blocks_per_cu = 1;
  if (k.pile_blocks_per_cu < 1) k.pile_blocks_per_cu = 1;
  k.pipeline = env_int ("PEMAP_PIPELINE", 1);
  k.chunk_pairs = env_int ("PEMAP_CHUNK_PAIRS", 262144);
  k.band = env_int ("PEMAP_BAND", 1);
  k.gapless_blocks_per_cu = env_int ("PEMAP_GAPLESS_BLOCKS_PER_CU", -1);     // one-wave workgroups of pm_gapless_kernel launched per CU at most; -1: by read length
  if (k.gapless_blocks_per_cu == 0) k.gapless_blocks_per_cu = 1;
  k.band_waves_per_cu = env_int ("PEMAP_BAND_WAVES_PER_CU", 16);
  if (k.band_waves_per_cu < 1) k.band_waves_per_cu = 1;
  k.seed_phase = 0;
#ifdef PEMAP_TIMING_PROBES
  k.seed_phase = env_int ("PEMAP_SEED_PHASE", 0);
#endif
}

struct pemap_dev
{
  int device;
  PmKnobs kn;
  hipStream_t stream;
  char err[512];
  // index
  uint32_t *d_pos_index, *d_mers;
  uint32_t *d_rep, *d_multi;    // look-up replicas (pemap_aux.hip.h), built by index_commit
  uint32_t multi_base;
  int n_rep, rep_want;          // rep_want: -1 = when the memory is there (default), 0 = never, 8 = required
  uint64_t multi_units;
  uint8_t *d_genome, *d_genome_alloc;   // the letters, and the allocation they sit in (padded in front)
  uint32_t *d_contig_starts;
  uint64_t n_mers, gsize;
  int n_contigs, idepth;
  bool index_ready;
  uint32_t *d_counts;           // the pileup counter planes (PmPile): 6 planes of pile_plane_words words
  size_t pile_plane_words;
  bool rest_on_alu;
  // each chunk's seed-stage remainder (big read-ends + emit) is enqueued exactly once: checked in launch_vote (the round-3 fault)
  uint64_t run_serial, rest_id[2];
  bool rest_twice;
  // params
  int paired, min_dist, max_dist, bisulfite;
  double min_align;
  // staged reads (capacity cap_reads rows each)
  uint8_t *d_reads1, *d_reads2;
  int *d_len1, *d_len2;
  int cap_reads, n_staged, stride, staged_paired, max_len_staged, min_len_staged;
  std::vector < int >h_len1, h_len2;
  // per-run work arrays (capacity cap_ends read-ends)
  int cap_ends;
  PmHits hits;
  uint32_t *d_tasks_s, *d_tasks_m, *d_redo, *d_wins;
  uint32_t *d_tasks_s2, *d_tasks_m2;      // second set: the vote of the next chunk runs beside the SW of this one
  bool vote_on_mem;
  // second set of the arrays the walk kernel reads, so that walk(chunk k) can run beside vote/SW(chunk k+1)
  PmHits hits2;
  uint32_t *d_wins2, *d_dirbuf2;
  unsigned long long *d_path, *d_path2;        // recorded traceback steps per winning alignment (two sets)
  uint16_t *d_nsteps, *d_nsteps2;
  int path_words, path_cap_ends;
  hipEvent_t ev_walk_done[2];
  uint32_t *d_m1, *d_m2;
  int *d_mtype;
  int cap_out;
  PmCounters *d_ctr;
  PmInsCursor *d_cur;
  uint32_t *d_seed_scratch;
  uint32_t *d_dirbuf;
  size_t dirbuf_dwords;
  size_t dir_slabs;             // slabs d_dirbuf holds for the staged read length; the last one is the dump slab of task-less lane groups
  uint8_t *d_ins_log;
  unsigned ins_cap;
  int seed_grid, sw_grid;
  // run bookkeeping
  int run_first, run_n;
  bool run_pending;             // kernels of the last run still in flight / not yet accounted
  bool run_split, serial_split;
  int run_chunks, run_chunk_pairs, run_L;
  uint64_t run_ends;
  hipEvent_t ev[7];
  // two-stream pipeline: the look-up kernel of chunk k+1 (memory stream) runs beside vote/SW/walk of chunk k
  hipStream_t stream2;
  hipStream_t stream3;          // the vote's own stream (PEMAP_VOTE_ON_MEM=2)
  hipEvent_t ev_lookup_done[2];
  int vote_stream;
  int n_cus;
  hipEvent_t ev_lists_ready[2], ev_lists_free[2];
  PmLists lists[2];
  int lists_cap;
  bool lists_arrays;            // the (key, segment) lists exist (not needed, and not allocated, while the fused seed kernel serves)
  PmChunkCtr *d_chunk_ctr;
  std::vector < hipEvent_t > evs;
  int big_grid, scratch_blocks;
  uint64_t last_big, last_big2;  // read-ends the fused seed kernel's first tier passed over; of them, left to the monolithic kernel
  PmCounters last_ctr;          // summed over the chunks of the last run
  PmInsCursor last_cur;
  // batches in flight
  std::mutex mu;                // guards the enqueue state: submit / wait may be called from several host threads
  // Submitters are serialised for the whole of a submit (taken BEFORE mu).  ring_finish drops mu while the host blocks on the old
  // batch's event; with mu alone a second submitter saw the same ring_seq there, chose the same slot, and whichever came second went
  // on with a stale slot and `first` -- overwriting an active slot's staging and handing out a ticket of another slot (round-2 review).
  // Waiters take mu only, so a wait is never held up by a submit that blocks on a full ring.
  std::mutex submit_mu;
  PmRingSlot ring[PM_RING];
  int ring_cap;                 // rows per slot, 0 = the ring is not set up (the staged arrays hold a resident read set)
  unsigned long long ring_seq;
  hipStream_t stream_h2d;
  float last_ms[8];
  std::vector < uint8_t > h_ins;        // host copy of all insertion-log bytes so far
  long summary[13];
};

static inline PmPile pile_of (const pemap_dev * d)
{
  PmPile p;
  p.w = d->d_counts;
  p.plane_words = d->pile_plane_words;
  p.genome = d->d_genome;
  return p;
}

static int fail (pemap_dev * d, const char *fmt, ...)
{
  va_list ap;
  va_start (ap, fmt);
  vsnprintf (d ? d->err : g_create_err, 512, fmt, ap);
  va_end (ap);
  return 1;
}

#define HIPCHK(d, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail (d, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString (e_)); } while (0)

template < class T > static int dev_alloc (pemap_dev * d, T ** p, size_t n)
{
  *p = nullptr;
  if (n == 0)
    n = 1;
  HIPCHK (d, hipMalloc ((void **) p, n * sizeof (T)));
  return 0;
}

#define TRY(x) do { int r_ = (x); if (r_) return r_; } while (0)

extern "C" const char *pemap_dev_last_error (const pemap_dev * dev)
{
  return dev ? dev->err : g_create_err;
}

extern "C" int pemap_dev_create (pemap_dev ** out, int device_id)
{
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount (&n) != hipSuccess || n <= 0)
    return fail (nullptr, "no HIP device visible: this library has no CPU path");
  if (device_id < 0 || device_id >= n)
    return fail (nullptr, "device %d out of range (0..%d)", device_id, n - 1);
  pemap_dev *d = new pemap_dev ();
  memset (d->err, 0, sizeof (d->err));
  d->device = device_id;
  d->d_pos_index = d->d_mers = nullptr;
  d->d_rep = d->d_multi = nullptr;
  d->multi_base = 0;
  d->n_rep = 0;
  d->rep_want = -1;
  d->multi_units = 0;
  d->d_genome = d->d_genome_alloc = nullptr;
  d->d_contig_starts = nullptr;
  d->d_counts = nullptr;
  d->n_mers = d->gsize = 0;
  d->n_contigs = 0;
  d->idepth = 16;
  d->index_ready = false;
  d->paired = 1;
  d->min_dist = 0;
  d->max_dist = 500;
  d->bisulfite = 0;
  d->min_align = 0.9;           // MIN_ALIGN default, pemapper.c:151
  d->d_reads1 = d->d_reads2 = nullptr;
  d->d_len1 = d->d_len2 = nullptr;
  d->cap_reads = d->n_staged = d->stride = 0;
  d->staged_paired = 0;
  d->cap_ends = 0;
  memset (&d->hits, 0, sizeof (d->hits));
  memset (&d->hits2, 0, sizeof (d->hits2));
  d->d_wins2 = d->d_dirbuf2 = nullptr;
  d->d_path = d->d_path2 = nullptr;
  d->d_nsteps = d->d_nsteps2 = nullptr;
  d->path_words = d->path_cap_ends = 0;
  d->d_tasks_s = d->d_tasks_m = d->d_redo = d->d_wins = d->d_m1 = d->d_m2 = nullptr;
  d->d_tasks_s2 = d->d_tasks_m2 = nullptr;
  d->vote_on_mem = false;
  d->stream3 = nullptr;
  d->vote_stream = 0;
  d->d_cur = nullptr;
  d->dirbuf_dwords = 0;
  memset (&d->last_cur, 0, sizeof (d->last_cur));
  d->d_mtype = nullptr;
  d->cap_out = 0;
  d->d_ctr = nullptr;
  d->d_seed_scratch = nullptr;
  d->d_dirbuf = nullptr;
  d->d_ins_log = nullptr;
  d->ins_cap = 0;
  d->run_first = d->run_n = 0;
  d->run_pending = false;
  d->run_split = false;
  d->run_chunks = 0;
  d->stream2 = nullptr;
  memset (d->lists, 0, sizeof (d->lists));
  d->lists_cap = 0;
  d->lists_arrays = false;
  d->d_chunk_ctr = nullptr;
  d->last_big = 0;
  d->ring_cap = 0;
  d->ring_seq = 0;
  d->stream_h2d = nullptr;
  for (int i = 0; i < PM_RING; i++)
    {
      d->ring[i].active = false;
      d->ring[i].seq = 0;
      d->ring[i].h_len = nullptr;
      d->ring[i].h_res = nullptr;
      d->ring[i].h_rows = nullptr;
      d->ring[i].h_rows_bytes = 0;
      d->ring[i].ev_done = nullptr;
    }
  memset (&d->last_ctr, 0, sizeof (d->last_ctr));
  memset (d->last_ms, 0, sizeof (d->last_ms));
  memset (d->summary, 0, sizeof (d->summary));
  if (hipSetDevice (device_id) != hipSuccess)
    {
      delete d;
      return fail (nullptr, "hipSetDevice(%d) failed", device_id);
    }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties (&prop, device_id) != hipSuccess)
    {
      delete d;
      return fail (nullptr, "hipGetDeviceProperties failed");
    }
  if (strncmp (prop.gcnArchName, "gfx950", 6) != 0)
    {
      fail (nullptr, "device %d is %s: this library is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
      delete d;
      return 1;
    }
  int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  read_knobs (d->kn);
  // pm_seed_kernel is launched with seed_grid blocks (monolithic form) or big_grid blocks (list mode, the big read-ends of the
  // split pipeline); every block owns one spill area of d_seed_scratch, which is sized for the LARGER of the two grids
  // (scratch_blocks) and passed to the kernel as its capacity.  (Round 1: the area was sized by seed_grid alone, an A/B run
  // raised big_grid above it through an environment knob, and the blocks beyond it wrote past the allocation: the memory
  // access fault of gpurun_out/ab_rep15.log.  DESIGN.md section 8.)
  d->seed_grid = cus * d->kn.seed_blocks_per_cu;
  d->sw_grid = cus * d->kn.sw_waves_per_cu;
  d->big_grid = cus * d->kn.big_blocks_per_cu;
  d->scratch_blocks = d->seed_grid > d->big_grid ? d->seed_grid : d->big_grid;
  d->n_cus = cus;
  if (hipStreamCreateWithFlags (&d->stream, hipStreamNonBlocking) != hipSuccess)
    {
      delete d;
      return fail (nullptr, "hipStreamCreate failed");
    }
  for (int i = 0; i < 7; i++)
    hipEventCreate (&d->ev[i]);
  if (hipMalloc ((void **) &d->d_ctr, sizeof (PmCounters)) != hipSuccess || hipMalloc ((void **) &d->d_cur, sizeof (PmInsCursor)) != hipSuccess)
    {
      delete d;
      return fail (nullptr, "hipMalloc failed");
    }
  hipMemset (d->d_ctr, 0, sizeof (PmCounters));
  hipMemset (d->d_cur, 0, sizeof (PmInsCursor));
  *out = d;
  return 0;
}

static void free_index (pemap_dev * d)
{
  hipFree (d->d_pos_index);
  hipFree (d->d_mers);
  hipFree (d->d_multi);         // d_rep is kept for the next index (pemap_dev_destroy frees it)
  d->d_multi = nullptr;
  d->n_rep = 0;
  hipFree (d->d_genome_alloc);
  d->d_genome_alloc = nullptr;
  hipFree (d->d_contig_starts);
  hipFree (d->d_counts);
  d->d_pos_index = d->d_mers = nullptr;
  d->d_genome = nullptr;
  d->d_contig_starts = nullptr;
  d->d_counts = nullptr;
  d->index_ready = false;
}

static void free_hits (PmHits & h)
{
  hipFree (h.n_hits);
  hipFree (h.spot);
  hipFree (h.gpos);
  hipFree (h.nn);
  hipFree (h.orient);
  hipFree (h.score);
  hipFree (h.sti);
  hipFree (h.stk);
  hipFree (h.slot);
  memset (&h, 0, sizeof (h));
}

static void free_work (pemap_dev * d)
{
  free_hits (d->hits);
  free_hits (d->hits2);
  hipFree (d->d_tasks_s);
  hipFree (d->d_tasks_m);
  hipFree (d->d_tasks_s2);
  hipFree (d->d_tasks_m2);
  d->d_tasks_s2 = d->d_tasks_m2 = nullptr;
  hipFree (d->d_redo);
  hipFree (d->d_wins);
  hipFree (d->d_wins2);
  hipFree (d->d_path);
  hipFree (d->d_path2);
  hipFree (d->d_nsteps);
  hipFree (d->d_nsteps2);
  d->d_path = d->d_path2 = nullptr;
  d->d_nsteps = d->d_nsteps2 = nullptr;
  d->path_words = d->path_cap_ends = 0;
  d->d_tasks_s = d->d_tasks_m = d->d_redo = d->d_wins = d->d_wins2 = nullptr;
  d->cap_ends = 0;
}

extern "C" void pemap_dev_destroy (pemap_dev * d)
{
  if (!d)
    return;
  hipSetDevice (d->device);
  hipStreamSynchronize (d->stream);
  free_index (d);
  hipFree (d->d_rep);
  d->d_rep = nullptr;
  free_work (d);
  hipFree (d->d_reads1);
  hipFree (d->d_reads2);
  hipFree (d->d_len1);
  hipFree (d->d_len2);
  hipFree (d->d_m1);
  hipFree (d->d_m2);
  hipFree (d->d_mtype);
  hipFree (d->d_ctr);
  hipFree (d->d_cur);
  hipFree (d->d_seed_scratch);
  hipFree (d->d_dirbuf);
  hipFree (d->d_dirbuf2);
  hipFree (d->d_ins_log);
  for (int i = 0; i < PM_RING; i++)
    {
      if (d->ring[i].h_len)
        hipHostFree (d->ring[i].h_len);
      if (d->ring[i].h_res)
        hipHostFree (d->ring[i].h_res);
      if (d->ring[i].h_rows)
        hipHostFree (d->ring[i].h_rows);
      if (d->ring[i].ev_done)
        hipEventDestroy (d->ring[i].ev_done);
      for (size_t k = 0; k < d->ring[i].ev_copy.size (); k++)
        hipEventDestroy (d->ring[i].ev_copy[k]);
    }
  if (d->stream_h2d)
    {
      hipStreamDestroy (d->stream_h2d);
    }
  for (int i = 0; i < 7; i++)
    hipEventDestroy (d->ev[i]);
  if (d->stream2)
    {
      hipStreamSynchronize (d->stream2);
      for (int i = 0; i < 2; i++)
        {
          hipEventDestroy (d->ev_lists_ready[i]);
          hipEventDestroy (d->ev_lists_free[i]);
          hipEventDestroy (d->ev_walk_done[i]);
          hipFree (d->lists[i].hdr);
          hipFree (d->lists[i].key);
          hipFree (d->lists[i].seg);
          hipFree (d->lists[i].big_list);
        }
      for (size_t i = 0; i < d->evs.size (); i++)
        hipEventDestroy (d->evs[i]);
      hipFree (d->d_chunk_ctr);
      hipStreamDestroy (d->stream2);
      if (d->stream3)
        {
          hipStreamSynchronize (d->stream3);
          hipStreamDestroy (d->stream3);
          hipEventDestroy (d->ev_lookup_done[0]);
          hipEventDestroy (d->ev_lookup_done[1]);
          d->stream3 = nullptr;
        }
    }
  hipStreamDestroy (d->stream);
  delete d;
}

// ------------------------------------------------------------------------------------------------------------
static const uint64_t POS_INDEX_N = (1ull << 32) + 1ull;
static int drain_ins (pemap_dev * d);
static int ring_leave (pemap_dev * d);
static void fold_summary (pemap_dev * d, int first, int n, const uint32_t * m1, const uint32_t * m2, const int *mapping_type);

extern "C" int pemap_dev_index_alloc (pemap_dev * d, uint64_t n_mers, uint64_t genome_size, int n_contigs, int idepth)
{
  HIPCHK (d, hipSetDevice (d->device));
  if (genome_size == 0 || genome_size >= (1ull << 32) - 400 || n_contigs < 1)
    return fail (d, "index_alloc: genome_size %llu / n_contigs %d not supported (positions are u32)",
                 (unsigned long long) genome_size, n_contigs);
  if (idepth != 16)
    return fail (d, "index_alloc: idepth %d: the .idx table is addressed by 16-mers (index_genome_whole.c:149)", idepth);
  free_index (d);
  d->n_mers = n_mers;
  d->gsize = genome_size;
  d->n_contigs = n_contigs;
  d->idepth = idepth;
  TRY (dev_alloc (d, &d->d_pos_index, POS_INDEX_N));
  TRY (dev_alloc (d, &d->d_mers, n_mers + 128));
  // (256 bytes in front, 512 behind: pm_band_kernel's window loads start a few bytes before a window and the kernels' 8-byte loads
  // end a few bytes behind one; d_genome points at the first letter)
  TRY (dev_alloc (d, &d->d_genome_alloc, genome_size + 768));
  HIPCHK (d, hipMemset (d->d_genome_alloc, 0, 256));
  d->d_genome = d->d_genome_alloc + 256;
  TRY (dev_alloc (d, &d->d_contig_starts, (size_t) n_contigs + 2));
  // (a plane is a whole number of 256-byte blocks: planes start line-aligned)
  d->pile_plane_words = (((size_t) genome_size + 2) / 2 + 63) & ~(size_t) 63;
  TRY (dev_alloc (d, &d->d_counts, 6 * d->pile_plane_words));
  HIPCHK (d, hipMemset (d->d_genome + genome_size, 0, 512));
  HIPCHK (d, hipMemset (d->d_counts, 0, 6 * d->pile_plane_words * sizeof (uint32_t)));
  return 0;
}

// The 8 look-up replicas and the records of the multi-position buckets (pemap_aux.hip.h), from pos_index / mers.
static int build_replicas (pemap_dev * d)
{
  hipFree (d->d_multi);
  d->d_multi = nullptr;
  d->n_rep = 0;
  int want = d->rep_want;
  if (want < 0 && d->kn.replicas >= 0)
    want = d->kn.replicas ? 8 : 0;
  if (want == 0)
    {
      hipFree (d->d_rep);
      d->d_rep = nullptr;
      return 0;
    }
  const size_t rep_bytes = 8ull * (1ull << 32) * sizeof (uint32_t);
  if (!d->d_rep)
    {
      // The replicas' 128 GiB are allocated once per object and kept across indexes.  What must stay free beside them for
      // the work arrays of a run (direction slabs, hit records, lists) and the records is probed by allocating it.  The
      // driver hands freed memory back lazily (an allocation of this size right after a hipFree of the same size fails,
      // and hipMemGetInfo lags too), hence the retries.
      const size_t reserve = (size_t) 48 << 30;
      bool ok = false;
      {
        // the common case, a fresh process: the free-memory figure is accurate, no probe needed (a 48 GiB allocation costs ~1 s)
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo (&fr, &tot) == hipSuccess && fr >= rep_bytes + reserve)
          {
            if (hipMalloc ((void **) &d->d_rep, rep_bytes) == hipSuccess)
              ok = true;
            else
              {
                (void) hipGetLastError ();
                d->d_rep = nullptr;
              }
          }
      }
      for (int attempt = 0; attempt < 10 && !ok; attempt++)
        {
          void *probe = nullptr;
          if (attempt)
            {
              (void) hipDeviceSynchronize ();
              usleep (300000);
            }
          if (hipMalloc ((void **) &d->d_rep, rep_bytes) == hipSuccess)
            {
              if (hipMalloc (&probe, reserve) == hipSuccess)
                ok = true;
              else
                {
                  hipFree (d->d_rep);
                  d->d_rep = nullptr;
                }
              hipFree (probe);
            }
          if (!ok)
            {
              (void) hipGetLastError ();
              d->d_rep = nullptr;
            }
        }
      if (!ok)
        {
          if (want == 8)
            return fail (d, "look-up replicas: the device cannot hold %.1f GB beside the index and %.1f GB of work arrays", rep_bytes / 1e9, reserve / 1e9);
          // the reference's layout serves the look-ups (pm_lookup_wave_kernel): same results, about 1.6x the time per step
          fprintf (stderr, "libpemap_hip: device %d has no room for the %.0f GB of look-up replicas beside the index and %.0f GB of work arrays; "
                   "the look-ups read the reference's table instead (slower, same results; PEMAP_REPLICAS=0 silences this)\n", d->device,
                   rep_bytes / 1e9, reserve / 1e9);
          return 0;
        }
    }
  uint32_t *units = d->d_rep + (1ull << 32);    // replica 1's place holds the record offsets until replica 0 is encoded
  const uint64_t n = 1ull << 32;
  const unsigned grid = (unsigned) d->n_cus * 64u;
  hipLaunchKernelGGL (ix_rep_units_kernel, dim3 (grid), dim3 (256), 0, d->stream, d->d_pos_index, units);
  const uint64_t sc_tiles = (n + SC_TILE - 1) / SC_TILE;
  uint32_t *d_tsum = nullptr;
  unsigned long long *d_total = nullptr;
  TRY (dev_alloc (d, &d_tsum, sc_tiles));
  TRY (dev_alloc (d, &d_total, 1));
  hipLaunchKernelGGL (ix_sumscan_reduce_kernel, dim3 ((unsigned) sc_tiles), dim3 (SC_BLOCK), 0, d->stream, units, n, d_tsum);
  hipLaunchKernelGGL (ix_sumscan_tiles_kernel, dim3 (1), dim3 (1024), 0, d->stream, d_tsum, sc_tiles, d_total);
  unsigned long long total = 0;
  HIPCHK (d, hipMemcpyAsync (&total, d_total, sizeof (total), hipMemcpyDeviceToHost, d->stream));
  HIPCHK (d, hipStreamSynchronize (d->stream));
  // entries below multi_base are positions (compressed coordinates < genome size); codes up to 0xFFFFFFFD address the records
  d->multi_base = (uint32_t) d->gsize;
  const bool fits = total < (unsigned long long) (0xFFFFFFFEu - d->multi_base);
  if (fits)
    {
      hipLaunchKernelGGL (ix_sumscan_apply_kernel, dim3 ((unsigned) sc_tiles), dim3 (SC_BLOCK), 0, d->stream, units, n, d_tsum);
      if (dev_alloc (d, &d->d_multi, (size_t) total * 4 + 64))
        {
          hipFree (d_tsum);
          hipFree (d_total);
          return 1;
        }
      hipLaunchKernelGGL (ix_rep_encode_kernel, dim3 (grid), dim3 (256), 0, d->stream, d->d_pos_index, d->d_mers, units, d->multi_base, d->d_rep,
                          d->d_multi);
      for (int p = 1; p < 8; p++)
        hipLaunchKernelGGL (ix_rep_permute_kernel, dim3 ((unsigned) d->n_cus * 32u), dim3 (256), 0, d->stream, d->d_rep,
                            d->d_rep + ((size_t) p << 32), p);
    }
  HIPCHK (d, hipStreamSynchronize (d->stream));
  HIPCHK (d, hipGetLastError ());
  hipFree (d_tsum);
  hipFree (d_total);
  if (!fits)
    {
      if (want == 8)
        return fail (d, "look-up replicas: %llu record units do not fit the codes above genome size %llu", total, (unsigned long long) d->gsize);
      return 0;
    }
  d->multi_units = total;
  d->n_rep = 8;
  return 0;
}

extern "C" int pemap_dev_set_lookup_replicas (pemap_dev * d, int n)
{
  if (n != -1 && n != 0 && n != 8)
    return fail (d, "set_lookup_replicas: %d (-1 = when the memory is there, 0 = never, 8 = required)", n);
  d->rep_want = n;
  if (d->index_ready)
    {
      HIPCHK (d, hipSetDevice (d->device));
      HIPCHK (d, hipDeviceSynchronize ());
      return build_replicas (d);
    }
  return 0;
}

extern "C" int pemap_dev_lookup_replicas (pemap_dev * d, int *n_replicas, uint64_t * record_bytes)
{
  *n_replicas = d->n_rep;
  if (record_bytes)
    *record_bytes = d->n_rep ? d->multi_units * 16ull : 0ull;
  return 0;
}

extern "C" int pemap_dev_index_commit (pemap_dev * d)
{
  if (!d->d_pos_index)
    return fail (d, "index_commit: no index arrays allocated");
  HIPCHK (d, hipSetDevice (d->device));
  HIPCHK (d, hipDeviceSynchronize ());
  TRY (build_replicas (d));
  d->index_ready = true;
  return 0;
}

extern "C" int pemap_dev_load_index (pemap_dev * d, const uint32_t * pos_index, const uint32_t * mers, uint64_t n_mers,
                                     const char *genome, uint64_t genome_size, const uint32_t * contig_starts, int n_contigs,
                                     int idepth)
{
  TRY (pemap_dev_index_alloc (d, n_mers, genome_size, n_contigs, idepth));
  HIPCHK (d, hipMemcpy (d->d_pos_index, pos_index, POS_INDEX_N * sizeof (uint32_t), hipMemcpyHostToDevice));
  HIPCHK (d, hipMemcpy (d->d_mers, mers, n_mers * sizeof (uint32_t), hipMemcpyHostToDevice));
  HIPCHK (d, hipMemcpy (d->d_genome, genome, genome_size, hipMemcpyHostToDevice));
  HIPCHK (d, hipMemcpy (d->d_contig_starts, contig_starts, ((size_t) n_contigs + 1) * sizeof (uint32_t), hipMemcpyHostToDevice));
  return pemap_dev_index_commit (d);
}

static int build_from_device_genome (pemap_dev * d, const uint32_t * contig_len, int n_contigs, int bisulfite)
{
  const uint64_t gsize = d->gsize;
  std::vector < uint64_t > real_starts (n_contigs + 1);
  std::vector < uint32_t > cstarts (n_contigs + 1);
  real_starts[0] = 0;
  cstarts[0] = 0;
  for (int c = 0; c < n_contigs; c++)
    {
      if (contig_len[c] < 16)
        return fail (d, "build_index: contig %d has %u letters; fewer than 16 is not indexable", c, contig_len[c]);
      real_starts[c + 1] = real_starts[c] + contig_len[c];
      cstarts[c + 1] = cstarts[c] + (contig_len[c] - 15);       // index_genome_whole.c:213, 316, 349
    }
  if (real_starts[n_contigs] != gsize)
    return fail (d, "build_index: contig lengths sum to %llu, genome has %llu letters", (unsigned long long) real_starts[n_contigs],
                 (unsigned long long) gsize);
  HIPCHK (d, hipMemcpy (d->d_contig_starts, cstarts.data (), (n_contigs + 1) * sizeof (uint32_t), hipMemcpyHostToDevice));
  uint64_t *d_real = nullptr;
  TRY (dev_alloc (d, &d_real, (size_t) n_contigs + 1));
  HIPCHK (d, hipMemcpy (d_real, real_starts.data (), (n_contigs + 1) * sizeof (uint64_t), hipMemcpyHostToDevice));
  const uint64_t n_tiles = (gsize + IX_PER_BLOCK - 1) / IX_PER_BLOCK;
  uint32_t *d_tc = nullptr;
  uint64_t *d_to = nullptr, *d_total = nullptr;
  TRY (dev_alloc (d, &d_tc, n_tiles));
  TRY (dev_alloc (d, &d_to, n_tiles));
  TRY (dev_alloc (d, &d_total, 1));
  hipLaunchKernelGGL (ix_count_kernel, dim3 ((unsigned) n_tiles), dim3 (IX_BLOCK), 0, d->stream, d->d_genome, d_real, n_contigs, gsize,
                      bisulfite, d_tc);
  hipLaunchKernelGGL (ix_scan_tiles_kernel, dim3 (1), dim3 (1024), 0, d->stream, d_tc, d_to, n_tiles, d_total);
  uint64_t n_mers = 0;
  HIPCHK (d, hipMemcpyAsync (&n_mers, d_total, sizeof (uint64_t), hipMemcpyDeviceToHost, d->stream));
  HIPCHK (d, hipStreamSynchronize (d->stream));
  if (n_mers == 0)
    return fail (d, "build_index: the genome has no 16-mer free of N");
  uint32_t *d_keys = nullptr, *d_vals = nullptr, *d_keys2 = nullptr;
  TRY (dev_alloc (d, &d_keys, n_mers));
  TRY (dev_alloc (d, &d_vals, n_mers));
  TRY (dev_alloc (d, &d_keys2, n_mers));
  hipFree (d->d_mers);
  d->d_mers = nullptr;
  TRY (dev_alloc (d, &d->d_mers, n_mers + 128));
  d->n_mers = n_mers;
  hipLaunchKernelGGL (ix_emit_kernel, dim3 ((unsigned) n_tiles), dim3 (IX_BLOCK), 0, d->stream, d->d_genome, d_real, n_contigs, gsize,
                      bisulfite, d_to, d_keys, d_vals);
  // stable LSD radix sort by k-mer: equal k-mers keep genome order, which is the .mdx order
  size_t tmp_bytes = 0;
  HIPCHK (d, rocprim::radix_sort_pairs (nullptr, tmp_bytes, d_keys, d_keys2, d_vals, d->d_mers, (size_t) n_mers, 0, 32, d->stream));
  void *d_tmp = nullptr;
  HIPCHK (d, hipMalloc (&d_tmp, tmp_bytes ? tmp_bytes : 1));
  HIPCHK (d, rocprim::radix_sort_pairs (d_tmp, tmp_bytes, d_keys, d_keys2, d_vals, d->d_mers, (size_t) n_mers, 0, 32, d->stream));
  // prefix table: bucket ends scattered, then a running maximum over 2^32 + 1 entries
  HIPCHK (d, hipMemsetAsync (d->d_pos_index, 0, POS_INDEX_N * sizeof (uint32_t), d->stream));
  hipLaunchKernelGGL (ix_run_ends_kernel, dim3 ((unsigned) ((n_mers + 255) / 256)), dim3 (256), 0, d->stream, d_keys2, n_mers,
                      d->d_pos_index);
  const uint64_t sc_tiles = (POS_INDEX_N + SC_TILE - 1) / SC_TILE;
  uint32_t *d_tmax = nullptr;
  TRY (dev_alloc (d, &d_tmax, sc_tiles));
  hipLaunchKernelGGL (ix_maxscan_reduce_kernel, dim3 ((unsigned) sc_tiles), dim3 (SC_BLOCK), 0, d->stream, d->d_pos_index, POS_INDEX_N, d_tmax);
  hipLaunchKernelGGL (ix_maxscan_tiles_kernel, dim3 (1), dim3 (1024), 0, d->stream, d_tmax, sc_tiles);
  hipLaunchKernelGGL (ix_maxscan_apply_kernel, dim3 ((unsigned) sc_tiles), dim3 (SC_BLOCK), 0, d->stream, d->d_pos_index, POS_INDEX_N, d_tmax);
  HIPCHK (d, hipStreamSynchronize (d->stream));
  HIPCHK (d, hipGetLastError ());
  hipFree (d_tmp);
  hipFree (d_tmax);
  hipFree (d_keys);
  hipFree (d_keys2);
  hipFree (d_vals);
  hipFree (d_tc);
  hipFree (d_to);
  hipFree (d_total);
  hipFree (d_real);
  return pemap_dev_index_commit (d);
}

extern "C" int pemap_dev_build_index (pemap_dev * d, const char *genome, uint64_t genome_size, const uint32_t * contig_len,
                                      int n_contigs, int bisulfite)
{
  TRY (pemap_dev_index_alloc (d, 0, genome_size, n_contigs, 16));
  HIPCHK (d, hipMemcpy (d->d_genome, genome, genome_size, hipMemcpyHostToDevice));
  return build_from_device_genome (d, contig_len, n_contigs, bisulfite);
}

extern "C" int pemap_dev_build_index_resident (pemap_dev * d, const void *d_genome, uint64_t genome_size,
                                               const uint32_t * contig_len, int n_contigs, int bisulfite)
{
  TRY (pemap_dev_index_alloc (d, 0, genome_size, n_contigs, 16));
  HIPCHK (d, hipMemcpy (d->d_genome, d_genome, genome_size, hipMemcpyDeviceToDevice));
  return build_from_device_genome (d, contig_len, n_contigs, bisulfite);
}

extern "C" int pemap_dev_buffer (pemap_dev * d, int which, void **d_ptr, uint64_t * n_bytes)
{
  if (!d->d_pos_index)
    return fail (d, "buffer: no index");
  switch (which)
    {
    case 0: *d_ptr = d->d_pos_index; *n_bytes = POS_INDEX_N * 4; break;
    case 1: *d_ptr = d->d_mers; *n_bytes = d->n_mers * 4; break;
    case 2: *d_ptr = d->d_genome; *n_bytes = d->gsize; break;
    case 3: *d_ptr = d->d_contig_starts; *n_bytes = ((uint64_t) d->n_contigs + 1) * 4; break;
    case 4: *d_ptr = d->d_counts; *n_bytes = 6 * d->pile_plane_words * 4; break;
    case 5: *d_ptr = d->d_rep; *n_bytes = d->n_rep ? 8ull * (1ull << 32) * 4ull : 0ull; break;
    case 6: *d_ptr = d->d_multi; *n_bytes = d->n_rep ? d->multi_units * 16ull : 0ull; break;
    default: return fail (d, "buffer: which = %d", which);
    }
  return 0;
}

extern "C" int pemap_dev_index_info (pemap_dev * d, uint64_t * n_mers, uint64_t * genome_size, int *n_contigs, int *idepth)
{
  if (!d->index_ready)
    return fail (d, "index_info: no index");
  *n_mers = d->n_mers;
  *genome_size = d->gsize;
  *n_contigs = d->n_contigs;
  *idepth = d->idepth;
  return 0;
}

extern "C" int pemap_dev_read_buffer (pemap_dev * d, int which, uint64_t byte_offset, void *host_dst, uint64_t n_bytes)
{
  void *p;
  uint64_t nb;
  TRY (pemap_dev_buffer (d, which, &p, &nb));
  if (byte_offset + n_bytes > nb)
    return fail (d, "read_buffer: range beyond buffer %d (%llu bytes)", which, (unsigned long long) nb);
  HIPCHK (d, hipSetDevice (d->device));
  HIPCHK (d, hipStreamSynchronize (d->stream));
  HIPCHK (d, hipMemcpy (host_dst, (const char *) p + byte_offset, n_bytes, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int pemap_dev_set_params (pemap_dev * d, int paired, int min_dist, int max_dist, double min_align, int bisulfite)
{
  d->paired = paired ? 1 : 0;
  d->min_dist = min_dist;
  d->max_dist = max_dist;
  d->min_align = min_align;
  d->bisulfite = bisulfite ? 1 : 0;
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
static int ensure_reads (pemap_dev * d, int n, int stride, int paired)
{
  if (n > d->cap_reads || stride != d->stride || (paired && !d->d_reads2))
    {
      hipFree (d->d_reads1);
      hipFree (d->d_reads2);
      hipFree (d->d_len1);
      hipFree (d->d_len2);
      d->d_reads1 = d->d_reads2 = nullptr;
      d->d_len1 = d->d_len2 = nullptr;
      int cap = n > d->cap_reads ? n : d->cap_reads;
      TRY (dev_alloc (d, &d->d_reads1, (size_t) cap * stride + 64));
      TRY (dev_alloc (d, &d->d_len1, (size_t) cap));
      TRY (dev_alloc (d, &d->d_reads2, (size_t) cap * stride + 64));
      TRY (dev_alloc (d, &d->d_len2, (size_t) cap));
      d->cap_reads = cap;
      d->stride = stride;
    }
  if (n > d->cap_out)
    {
      hipFree (d->d_m1);
      hipFree (d->d_m2);
      hipFree (d->d_mtype);
      TRY (dev_alloc (d, &d->d_m1, (size_t) n));
      TRY (dev_alloc (d, &d->d_m2, (size_t) n));
      TRY (dev_alloc (d, &d->d_mtype, (size_t) n));
      d->cap_out = n;
    }
  return 0;
}

// SW geometry for the longest staged read L: lanes per alignment and W columns per lane, the smallest instantiation with
// lanes * W >= L.  8 lanes x 13 columns up to 104 bases.  Beyond that 16 lanes (x 10 / 13 / 16 / 19 columns): beside the gapless
// rule the DP sees few problems per launch and the finer form wins (2 x 150 bp: 44.8 ms per step against 46.2 with 8 x 19;
// 2 x 250 bp: 111.8 against 120.5 with 8 x 32); with the rule off (PEMAP_GAPLESS=0: every problem through the DP) reads of
// 105..152 bases take 8 x 19, which was 18 % faster there.  (8 x 26 / 32 / 38 and 12 x 13 were measured equal or slower and are gone.)
// PEMAP_GAPLESS=0: every problem goes through the DP (the rule of pm_gapless_kernel off); 1: its first case only (diagonals
// with at most one mismatch); default 2: both cases
static int pm_gapless_max_x (const pemap_dev * d)
{
  return d->kn.gapless;
}

static bool pm_gapless_on (const pemap_dev * d)
{
  return d->kn.gapless != 0;
}

static void pick_geom (const pemap_dev * d, int L, int *lanes, int *w)
{
  if (L <= 8 * 13) { *lanes = 8; *w = 13; }
  else if (!pm_gapless_on (d) && L <= 8 * 19) { *lanes = 8; *w = 19; }
  else if (L <= 16 * 10) { *lanes = 16; *w = 10; }
  else if (L <= 16 * 13) { *lanes = 16; *w = 13; }
  else if (L <= 16 * 16) { *lanes = 16; *w = 16; }
  else { *lanes = 16; *w = 19; }
}

// rows of one lane's region in a direction slab: window rows + the skew steps, rounded up to the 16-step flush unit
static int tstride_for (const pemap_dev * d, int L)
{
  int lanes, w;
  pick_geom (d, L, &lanes, &w);
  return (L + 21 + lanes + 15) & ~15;
}

static size_t slab_dwords_for (const pemap_dev * d, int L)
{
  int lanes, W;
  pick_geom (d, L, &lanes, &W);
  return (size_t) lanes * (size_t) tstride_for (d, L) * (size_t) ((W * 4 + 31) / 32);
}

// device bytes the direction slabs of one chunk may take (one slab per read-end); PEMAP_DIR_BUDGET_GB overrides
static size_t dir_budget_bytes (const pemap_dev * d)
{
  return (size_t) (d->kn.dir_budget_gb * 1073741824.0);
}

static int alloc_hits (pemap_dev * d, PmHits & h, int n_ends)
{
  size_t nh = (size_t) n_ends * PM_MAX_HITS;
  TRY (dev_alloc (d, &h.n_hits, (size_t) n_ends));
  TRY (dev_alloc (d, &h.slot, (size_t) n_ends));
  TRY (dev_alloc (d, &h.spot, nh));
  TRY (dev_alloc (d, &h.gpos, nh));
  TRY (dev_alloc (d, &h.nn, nh));
  TRY (dev_alloc (d, &h.orient, nh));
  TRY (dev_alloc (d, &h.score, nh));
  TRY (dev_alloc (d, &h.sti, nh));
  TRY (dev_alloc (d, &h.stk, nh));
  return 0;
}

static int ensure_work (pemap_dev * d, int n_ends, bool two_sets)
{
  if (n_ends > d->cap_ends || (two_sets && !d->hits2.n_hits))
    {
      free_work (d);
      size_t nh = (size_t) n_ends * PM_MAX_HITS;
      TRY (alloc_hits (d, d->hits, n_ends));
      TRY (dev_alloc (d, &d->d_wins, (size_t) n_ends));
      if (two_sets)
        {
          TRY (alloc_hits (d, d->hits2, n_ends));
          TRY (dev_alloc (d, &d->d_wins2, (size_t) n_ends));
          TRY (dev_alloc (d, &d->d_tasks_s2, (size_t) n_ends * 3));       // second third: the problems left to the DP, last third: to the banded DP
          TRY (dev_alloc (d, &d->d_tasks_m2, nh * 3));
        }
      TRY (dev_alloc (d, &d->d_tasks_s, (size_t) n_ends * 3));
      TRY (dev_alloc (d, &d->d_tasks_m, nh * 3));
      TRY (dev_alloc (d, &d->d_redo, (size_t) n_ends));
      d->cap_ends = n_ends;
    }
  if (!d->d_seed_scratch)
    TRY (dev_alloc (d, &d->d_seed_scratch, (size_t) d->scratch_blocks * 6 * PM_MAX_SEG * PM_SEG_LIST_MAX));
  size_t need = ((size_t) n_ends + 1) * slab_dwords_for (d, d->max_len_staged);       // + 1: dump slab for task-less lane groups
  if (need > d->dirbuf_dwords || (two_sets && !d->d_dirbuf2))
    {
      hipFree (d->d_dirbuf);
      hipFree (d->d_dirbuf2);
      d->d_dirbuf = d->d_dirbuf2 = nullptr;
      d->dirbuf_dwords = 0;
      TRY (dev_alloc (d, &d->d_dirbuf, need));
      if (two_sets)
        TRY (dev_alloc (d, &d->d_dirbuf2, need));
      d->dirbuf_dwords = need;
    }
  d->dir_slabs = d->dirbuf_dwords / slab_dwords_for (d, d->max_len_staged);
  // recorded traceback steps: PM_PATH_WORDS words of 32 two-bit steps per read-end
  const int pwords = PM_PATH_WORDS (d->max_len_staged);
  if (!d->d_path || n_ends > d->path_cap_ends || pwords != d->path_words || (two_sets && !d->d_path2))
    {
      hipFree (d->d_path);
      hipFree (d->d_path2);
      hipFree (d->d_nsteps);
      hipFree (d->d_nsteps2);
      d->d_path = d->d_path2 = nullptr;
      d->d_nsteps = d->d_nsteps2 = nullptr;
      d->path_words = pwords;
      d->path_cap_ends = n_ends;
      TRY (dev_alloc (d, &d->d_path, (size_t) n_ends * pwords));
      TRY (dev_alloc (d, &d->d_nsteps, (size_t) n_ends));
      if (two_sets)
        {
          TRY (dev_alloc (d, &d->d_path2, (size_t) n_ends * pwords));
          TRY (dev_alloc (d, &d->d_nsteps2, (size_t) n_ends));
        }
    }
  // insertion log: 64 bytes per read-end of a chunk is ample for real data, and at least 512 MB so that runs queued back to
  // back (the log is drained when a run is absorbed) do not fill it; overflow is reported as an error
  size_t want = (size_t) n_ends * 64 + (1u << 20);
  if (want < ((size_t) 512 << 20))
    want = (size_t) 512 << 20;
  if (want > 0xF0000000ull)
    want = 0xF0000000ull;
  if (want > d->ins_cap)
    {
      TRY (drain_ins (d));
      hipFree (d->d_ins_log);
      TRY (dev_alloc (d, &d->d_ins_log, want));
      d->ins_cap = (unsigned) want;
    }
  return 0;
}

static int check_lengths (pemap_dev * d, const int *len, int n, int *mx, int *mn)
{
  for (int i = 0; i < n; i++)
    {
      if (len[i] < PEMAP_MIN_READ || len[i] > PEMAP_MAX_READ)
        return fail (d, "read %d has length %d: supported range is %d..%d", i, len[i], PEMAP_MIN_READ, PEMAP_MAX_READ);
      if (len[i] > *mx)
        *mx = len[i];
      if (len[i] < *mn)
        *mn = len[i];
    }
  return 0;
}

extern "C" int pemap_dev_stage_reads (pemap_dev * d, const char *reads1, const int *len1, const char *reads2, const int *len2,
                                      int n, int stride)
{
  HIPCHK (d, hipSetDevice (d->device));
  if (n <= 0)
    return fail (d, "stage_reads: n = %d", n);
  if (d->paired && (!reads2 || !len2))
    return fail (d, "stage_reads: paired mode needs reads2/len2");
  if (stride < PEMAP_MIN_READ)
    return fail (d, "stage_reads: stride %d", stride);
  int mx = 0, mn = 1 << 30;
  TRY (check_lengths (d, len1, n, &mx, &mn));
  if (d->paired)
    TRY (check_lengths (d, len2, n, &mx, &mn));
  if (mx > stride)
    return fail (d, "stage_reads: a read is longer than the row stride %d", stride);
  TRY (ring_leave (d));
  TRY (pemap_dev_sync (d));
  TRY (ensure_reads (d, n, stride, d->paired));
  HIPCHK (d, hipMemcpy (d->d_reads1, reads1, (size_t) n * stride, hipMemcpyHostToDevice));
  HIPCHK (d, hipMemcpy (d->d_len1, len1, (size_t) n * sizeof (int), hipMemcpyHostToDevice));
  d->h_len1.assign (len1, len1 + n);
  d->h_len2.clear ();
  if (d->paired)
    {
      HIPCHK (d, hipMemcpy (d->d_reads2, reads2, (size_t) n * stride, hipMemcpyHostToDevice));
      HIPCHK (d, hipMemcpy (d->d_len2, len2, (size_t) n * sizeof (int), hipMemcpyHostToDevice));
      d->h_len2.assign (len2, len2 + n);
    }
  d->n_staged = n;
  d->staged_paired = d->paired;
  d->max_len_staged = mx;
  d->min_len_staged = mn;
  return 0;
}

__global__ void pm_nop_kernel ()
{
}

struct RunCtx
{
  PmIndex ix;
  PmBatch b;
  PmParams prm;
  int tstride, L;
  uint32_t *dump_slab;
};

// per-chunk device counters: the kernels' PmCounters plus what the look-up kernel counts
struct PmChunkCtr
{
  PmCounters c;
  unsigned long long positions;
  unsigned n_big;
  unsigned next_end;            // work counter of the persistent look-up waves
  unsigned n_big2;              // read-ends the second tier of the fused seed kernel leaves to the monolithic kernel
  unsigned next_end2;           // the second tier's work counter
};

#define PM_MAX_CHUNKS 256
#define PM_NEV 12               // events per chunk: lookup start/end, the vote kernel's end, the seed stage's end, then the ALU stream's kernel boundaries

static int seg_template (int L)
{
  const int segs = L / 16 + ((L % 16) ? 1 : 0);      // len/16 (+1 unless divisible), pemapper.c:1573-1587
  return segs <= 7 ? 7 : segs <= 10 ? 10 : segs <= 13 ? 13 : segs <= 16 ? 16 : 19;
}

// ---- the memory stream's work for one chunk: look-ups + slice gather into the slot's lists
// is the seed stage of this run the fused kernel (pm_seed4_kernel: look-ups and vote of a read-end in one wave)?
static bool pm_fused (const pemap_dev * d)
{
  return d->n_rep == 8;
}

static void launch_lookup (pemap_dev * d, const RunCtx & c, int slot, PmChunkCtr * cc, hipEvent_t * ev, bool split)
{
  PmLists L = d->lists[slot];
  L.n_big = &cc->n_big;
  L.positions = &cc->positions;
  L.next_end = &cc->next_end;
  hipStream_t st = d->serial_split ? d->stream : d->stream2;
  // (an event recorded straight after a stream wait is stamped when the wait is queued, not when it is satisfied: the empty
  // kernel makes the stamp the moment the look-up kernel can start, so that ev[0]..ev[1] is the kernel's own duration)
  hipLaunchKernelGGL (pm_nop_kernel, dim3 (1), dim3 (1), 0, st);
  hipEventRecord (ev[0], st);
  // PEMAP_LOOKUP_WAVES=n: n persistent one-wave workgroups per CU.  Default 9 (pm_seed4_kernel: 14.6 KB of LDS and 128 VGPRs a wave): 8
  // steps of the default workload take 23.4 / 21.3 / 20.7 / 21.3 ms each on resident reads with 7 / 8 / 9 / 10 (profiles/r04_ab_sweeps.txt;
  // the third form, 19.6 KB and 168 VGPRs, took 25.2 at its best, 7) -- the kernel itself keeps gaining (5.06 / 4.53 / 4.05 / 3.84 ms per
  // launch) while the other stream's DP kernels lose the SIMDs' registers to it
  int lw = d->kn.lookup_waves > 0 ? d->kn.lookup_waves : 9;
  if (c.ix.n_rep == 8)
    {
      // (no more workgroups than are resident at once: a workgroup owns its first ends by its number, and one that starts when
      // another ends -- at the launch's end -- would be the launch's tail)
      size_t lds = 0;
      int per_simd = 2;
      switch (seg_template (c.L))
        {
        case 7: lds = sizeof (PmSeed4Shared < 7, 0 >); per_simd = PM_S4_WAVES_PER_EU; break;
        case 10: lds = sizeof (PmSeed4Shared < 10, 0 >); per_simd = PM_S4_WAVES_PER_EU; break;
        case 13: lds = sizeof (PmSeed4Shared < 13, 0 >); break;
        case 16: lds = sizeof (PmSeed4Shared < 16, 0 >); break;
        default: lds = sizeof (PmSeed4Shared < 19, 0 >); break;
        }
      const int fit = (int) ((size_t) 160 * 1024 / lds);
      if (lw > fit)
        lw = fit;
      if (lw > 4 * per_simd)
        lw = 4 * per_simd;
    }
  int lgrid = lw * d->n_cus;
  if (lgrid > c.b.n_ends)
    lgrid = c.b.n_ends;
  const int lprio = d->kn.lookup_prio;
  const bool set2 = split && slot;
  const PmHits & H = set2 ? d->hits2 : d->hits;
  // the fused seed kernel (pm_seed4_kernel, tier 0) ...
#define PM_LK(SM) do { if (c.ix.n_rep == 8) \
      hipLaunchKernelGGL (HIP_KERNEL_NAME (pm_seed4_kernel < SM, 0 >), dim3 (lgrid), dim3 (64), sizeof (PmSeed4Shared < SM, 0 >), st, c.ix, c.b, c.prm, H, L, \
                          (const uint32_t *) nullptr, (const unsigned *) nullptr, lprio); \
    else hipLaunchKernelGGL (HIP_KERNEL_NAME (pm_lookup_wave_kernel < SM, 4 >), dim3 (lgrid), dim3 (64), 0, st, c.ix, c.b, c.prm, L, lprio); } while (0)
  switch (seg_template (c.L))
    {
    case 7: PM_LK (7); break;
    case 10: PM_LK (10); break;
    case 13: PM_LK (13); break;
    case 16: PM_LK (16); break;
    default: PM_LK (19); break;
    }
#undef PM_LK
  hipEventRecord (ev[1], st);
  if (pm_fused (d))
    {
      // ... then its second tier over the ends it passed over: the same kernel with four times the list (twice, for reads over 160
      // bases), a few waves per CU -- most launches find a few thousand ends; what THAT leaves goes to the monolithic kernel (launch_vote)
      PmLists L2 = L;
      L2.big_list = L.big_list + d->lists_cap;
      L2.n_big = &cc->n_big2;
      L2.next_end = &cc->next_end2;
      const int grid2 = d->kn.tier2_waves * d->n_cus;
#define PM_LK2(SM) hipLaunchKernelGGL (HIP_KERNEL_NAME (pm_seed4_kernel < SM, 1 >), dim3 (grid2), dim3 (64), sizeof (PmSeed4Shared < SM, 1 >), st, c.ix, c.b, c.prm, \
                                       H, L2, (const uint32_t *) L.big_list, (const unsigned *) L.n_big, lprio)
      switch (seg_template (c.L))
        {
        case 7: PM_LK2 (7); break;
        case 10: PM_LK2 (10); break;
        case 13: PM_LK2 (13); break;
        case 16: PM_LK2 (16); break;
        default: PM_LK2 (19); break;
        }
#undef PM_LK2
    }
  if (pm_fused (d))
    {
      // the vote is part of the kernel: its interval is empty
      hipEventRecord (ev[2], st);
      hipEventRecord (ev[3], st);
    }
}

// ---- the seed stage after the look-ups, on stream `st`: vote + list-mode remainder on the slot's lists (split), or the
//      monolithic seed kernel; then the emit kernel (windows, slab numbers, SW task lists).
// PEMAP_VOTE_REST_ON_ALU=1: with the vote on a stream of its own only its kernel runs there (part 1); the list-mode remainder
// of the big read-ends and the emit kernel (part 2) go to the ALU stream in front of the chunk's SW, so that the next chunk's
// vote starts 0.5 ms earlier.  Measured 43.3 ms per step against 41.7 (the vote kernel itself slows down by as much as it
// gains: 4.95 ms per launch against 4.5), so it is off by default.
static bool pm_fused (const pemap_dev * d);
static bool pm_vote_rest_on_alu (const pemap_dev * d)
{
  return d->rest_on_alu;        // set per run (run_slice)
}

// part 0: the whole stage; 1: the vote kernel only; 2: what follows it; 3: of that, the list-mode remainder only; 4: the emit kernel only
static void launch_vote (pemap_dev * d, const RunCtx & c, bool split, int slot, PmChunkCtr * cc, hipEvent_t * ev, hipStream_t st, int part = 0)
{
  const bool set2 = split && slot;
  const PmHits & H = set2 ? d->hits2 : d->hits;
  uint32_t *tasks_s = set2 ? d->d_tasks_s2 : d->d_tasks_s, *tasks_m = set2 ? d->d_tasks_m2 : d->d_tasks_m;
  const int n_ends = c.b.n_ends;
  PmCounters *ctr = &cc->c;
  const int phase_limit = d->kn.seed_phase;  // timing probe, 0 unless built with -DPEMAP_TIMING_PROBES
  if (part != 1 && part != 4)
    {
      // The remainder appends to the chunk's task lists and uses the one spill scratch: a second launch for the same chunk doubles the
      // appended tasks past the lists' ends (the fault PEMAP_REST_STREAM3=1 produced in round 3).  Refused here, reported by run_slice.
      const uint64_t id = (d->run_serial << 24) | (uint64_t) (cc - d->d_chunk_ctr);
      if (d->rest_id[slot & 1] == id)
        {
          d->rest_twice = true;
          return;
        }
      d->rest_id[slot & 1] = id;
    }
  if (part < 2)
    {
      hipLaunchKernelGGL (pm_nop_kernel, dim3 (1), dim3 (1), 0, st);
      hipEventRecord (ev[2], st);
    }
  if (part == 4)
    ;
  else if (split)
    {
      PmLists L = d->lists[slot];
      L.n_big = &cc->n_big;
      if (pm_fused (d))
        {
          // (the fused seed kernel's second tier has taken most of the first tier's list: the monolithic kernel gets what it left)
          L.big_list += d->lists_cap;
          L.n_big = &cc->n_big2;
        }
      L.positions = &cc->positions;
      // PEMAP_VOTE_WAVES=n: at most n one-wave workgroups per CU, each striding over the ends.  Default 1024 = one wave per end: the
      // dispatcher then places vote waves wherever the look-up and SW waves of the other stream leave room (measured 71.6 ms per
      // step against 74.9 with 12 persistent waves per CU)
      const int vw = d->kn.vote_waves;
      const int vprio = d->kn.vote_prio;
      int vgrid = vw * d->n_cus;
      if (vgrid > n_ends)
        vgrid = n_ends;
#define PM_VT(SM) hipLaunchKernelGGL (HIP_KERNEL_NAME (pm_vote_wave_kernel < SM >), dim3 (vgrid), dim3 (64), 0, st, c.ix, c.b, c.prm, H, L, vprio)
      // (the grid never exceeds the blocks d_seed_scratch holds a spill area for; the kernel checks it against its capacity too)
      const int bgrid = d->big_grid < d->scratch_blocks ? d->big_grid : d->scratch_blocks;
#define PM_SEEDL(SM) hipLaunchKernelGGL (HIP_KERNEL_NAME (pm_seed_kernel < SM >), dim3 (bgrid), dim3 (PM_SEED_THREADS), 0, st, c.ix, c.b, \
                                         c.prm, H, tasks_s, tasks_m, ctr, d->d_seed_scratch, d->scratch_blocks, 0, L.big_list, L.n_big)
      switch (seg_template (c.L))
        {
        // ev[2]..ev[3] = the vote kernel alone; the list-mode remainder and the emit kernel end at ev[10]
        case 7: if (part < 2) { PM_VT (7); hipEventRecord (ev[3], st); } if (part != 1) PM_SEEDL (7); break;
        case 10: if (part < 2) { PM_VT (10); hipEventRecord (ev[3], st); } if (part != 1) PM_SEEDL (10); break;
        case 13: if (part < 2) { PM_VT (13); hipEventRecord (ev[3], st); } if (part != 1) PM_SEEDL (13); break;
        case 16: if (part < 2) { PM_VT (16); hipEventRecord (ev[3], st); } if (part != 1) PM_SEEDL (16); break;
        default: if (part < 2) { PM_VT (19); hipEventRecord (ev[3], st); } if (part != 1) PM_SEEDL (19); break;
        }
#undef PM_VT
#undef PM_SEEDL
      if (part != 1)
        hipEventRecord (d->ev_lists_free[slot], st);    // the slot's lists are consumed
    }
  else
    {
      int sgrid = d->seed_grid < n_ends ? d->seed_grid : n_ends;
      if (sgrid > d->scratch_blocks)
        sgrid = d->scratch_blocks;
#define PM_SEED(SM) hipLaunchKernelGGL (HIP_KERNEL_NAME (pm_seed_kernel < SM >), dim3 (sgrid), dim3 (PM_SEED_THREADS), 0, st, c.ix, c.b, c.prm, \
                                        H, tasks_s, tasks_m, ctr, d->d_seed_scratch, d->scratch_blocks, phase_limit, (const uint32_t *) nullptr, \
                                        (const unsigned *) nullptr)
      switch (seg_template (c.L))
        {
        case 7: PM_SEED (7); break;
        case 10: PM_SEED (10); break;
        case 13: PM_SEED (13); break;
        case 16: PM_SEED (16); break;
        default: PM_SEED (19); break;
        }
#undef PM_SEED
    }
  if (part == 1 || part == 3)
    return;
  hipLaunchKernelGGL (pm_emit_kernel, dim3 ((n_ends + 63) / 64), dim3 (64), 0, st, c.ix, c.b, H, tasks_s, tasks_m, ctr);
  if (!split)
    hipEventRecord (ev[3], st);
  hipEventRecord (ev[10], st);
}

// ---- the ALU stream's work for one chunk: (the seed stage unless it ran on the memory stream,) SW, selection, traceback.
template < int W, int LPA > static void launch_chunk (pemap_dev * d, const RunCtx & c, uint32_t * m1, uint32_t * m2, int *mt, bool split, int slot,
                                             PmChunkCtr * cc, hipEvent_t * ev)
{
  // the arrays the walk reads alternate between two sets in the split pipeline
  const int swprio = d->kn.sw_prio;
  const bool set2 = split && slot;
  const PmHits & H = set2 ? d->hits2 : d->hits;
  uint32_t *wins = set2 ? d->d_wins2 : d->d_wins;
  uint32_t *dirbuf = set2 ? d->d_dirbuf2 : d->d_dirbuf;
  uint32_t *tasks_s = set2 ? d->d_tasks_s2 : d->d_tasks_s, *tasks_m = set2 ? d->d_tasks_m2 : d->d_tasks_m;
  uint32_t *dump_slab = dirbuf + (c.dump_slab - d->d_dirbuf);
  const int n_ends = c.b.n_ends;
  PmCounters *ctr = &cc->c;
  if (split && pm_fused (d))
    {
      // the seed stage ran on the look-up's stream (enqueue_lookup); PEMAP_VOTE_REST_ON_ALU=1 leaves the big read-ends' remainder
      // and the emit kernel to this stream
      if (pm_vote_rest_on_alu (d) && !d->serial_split)
        launch_vote (d, c, split, slot, cc, ev, d->stream, 4);
    }
  else if (!(split && d->vote_on_mem))
    launch_vote (d, c, split, slot, cc, ev, d->stream);
  else if (d->vote_stream == 3 && pm_vote_rest_on_alu (d))
    launch_vote (d, c, split, slot, cc, ev, d->stream, 2);
  if (pm_gapless_on (d))
    {
      uint32_t *tasks_dp = tasks_s + d->cap_ends;
      uint32_t *tasks_band = d->kn.band ? tasks_s + 2 * (size_t) d->cap_ends : nullptr;
      int ggrid = (n_ends + PM_GL_PER_BLOCK - 1) / PM_GL_PER_BLOCK;
      // The rule's kernel and the NEXT chunk's seed kernel are launched at the same moment (both wait for this chunk's seed kernel), and
      // whichever gets its waves onto the SIMDs first keeps them: the seed kernel's persistent waves need 168 registers each, six of
      // this kernel's fill a SIMD's 512.  For reads of up to 160 bases 24 workgroups per CU are the measured optimum (the seed
      // kernel's launch is short of its 7 waves per CU for 0.4 ms at most); for longer reads the seed kernel's launches are twice
      // as long, its waves came up late on many CUs, and the step was 69 ms with 12 or more against 54 with 6 (2 x 245 bases:
      // profiles/r03_ab_sweeps.txt; round 2's 256-thread workgroups had hidden this: they rarely found room at all)
      const int gbp = d->kn.gapless_blocks_per_cu > 0 ? d->kn.gapless_blocks_per_cu : (seg_template (c.L) <= 10 ? 24 : 6);
      if (ggrid > d->n_cus * gbp)
        ggrid = d->n_cus * gbp;
      hipLaunchKernelGGL (pm_gapless_kernel, dim3 (ggrid), dim3 (PM_GL_BLOCK), 0, d->stream, c.ix, c.b, c.prm, H, tasks_s, &ctr->n_tasks_s, tasks_dp,
                          &ctr->n_tasks_dp, pm_gapless_max_x (d), tasks_band, &ctr->n_band[0]);
      if (tasks_band)
        {
          // the problems the rule left open whose best diagonal has few mismatches: the DP restricted to a band of 32 diagonals,
          // four lanes per problem (pemap_band.hip.h); what is left for pm_sw_kernel are mostly the reads with a real indel
          int bgrid = (n_ends + 15) / 16;
          if (bgrid > d->n_cus * d->kn.band_waves_per_cu)
            bgrid = d->n_cus * d->kn.band_waves_per_cu;
          hipLaunchKernelGGL (HIP_KERNEL_NAME (pm_band_kernel < true >), dim3 (bgrid), dim3 (64), 0, d->stream, c.ix, c.b, c.prm, H, tasks_band,
                              &ctr->n_band[0], ctr, dirbuf, slab_dwords_for (d, c.L), &ctr->band_next[0]);
        }
      hipLaunchKernelGGL (HIP_KERNEL_NAME (pm_sw_kernel < W, LPA, true >), dim3 (d->sw_grid), dim3 (64), 0, d->stream, c.ix, c.b, c.prm, H,
                          tasks_dp, &ctr->n_tasks_dp, ctr, dirbuf, dump_slab, c.tstride, c.L, swprio, &ctr->sw_next[0]);
    }
  else
    hipLaunchKernelGGL (HIP_KERNEL_NAME (pm_sw_kernel < W, LPA, true >), dim3 (d->sw_grid), dim3 (64), 0, d->stream, c.ix, c.b, c.prm, H,
                        tasks_s, &ctr->n_tasks_s, ctr, dirbuf, dump_slab, c.tstride, c.L, swprio, &ctr->sw_next[0]);
  hipEventRecord (ev[4], d->stream);
  if (pm_gapless_on (d))
    {
      // the same rule on the problems of the multi-hit ends (scores only; a winner it decided is not scored again);
      // sw_next[3] counts what is left to the DP
      uint32_t *tasks_mdp = tasks_m + (size_t) d->cap_ends * PM_MAX_HITS;
      uint32_t *tasks_mband = d->kn.band ? tasks_m + 2 * (size_t) d->cap_ends * PM_MAX_HITS : nullptr;
      hipLaunchKernelGGL (pm_gapless_kernel, dim3 (d->n_cus * 16), dim3 (PM_GL_BLOCK), 0, d->stream, c.ix, c.b, c.prm, H, tasks_m, &ctr->n_tasks_m, tasks_mdp,
                          &ctr->sw_next[3], pm_gapless_max_x (d), tasks_mband, &ctr->n_band[1]);
      if (tasks_mband)
        hipLaunchKernelGGL (HIP_KERNEL_NAME (pm_band_kernel < false >), dim3 (d->n_cus * d->kn.band_waves_per_cu), dim3 (64), 0, d->stream, c.ix, c.b, c.prm, H,
                            tasks_mband, &ctr->n_band[1], ctr, dirbuf, slab_dwords_for (d, c.L), &ctr->band_next[1]);
      hipLaunchKernelGGL (HIP_KERNEL_NAME (pm_sw_kernel < W, LPA, false >), dim3 (d->sw_grid), dim3 (64), 0, d->stream, c.ix, c.b, c.prm, H,
                          tasks_mdp, &ctr->sw_next[3], ctr, dirbuf, dump_slab, c.tstride, c.L, swprio, &ctr->sw_next[1]);
    }
  else
    hipLaunchKernelGGL (HIP_KERNEL_NAME (pm_sw_kernel < W, LPA, false >), dim3 (d->sw_grid), dim3 (64), 0, d->stream, c.ix, c.b, c.prm, H,
                        tasks_m, &ctr->n_tasks_m, ctr, dirbuf, dump_slab, c.tstride, c.L, swprio, &ctr->sw_next[1]);
  hipEventRecord (ev[5], d->stream);
  hipLaunchKernelGGL (pm_select_kernel, dim3 ((c.b.n + 63) / 64), dim3 (64), 0, d->stream, c.b, c.prm, H, d->d_redo, wins, ctr,
                      m1, m2, mt);
  hipEventRecord (ev[6], d->stream);
  hipLaunchKernelGGL (HIP_KERNEL_NAME (pm_sw_kernel < W, LPA, true >), dim3 (d->sw_grid), dim3 (64), 0, d->stream, c.ix, c.b, c.prm, H,
                      d->d_redo, &ctr->n_redo, ctr, dirbuf, dump_slab, c.tstride, c.L, swprio, &ctr->sw_next[2]);
  hipEventRecord (ev[7], d->stream);
  hipEventRecord (ev[9], d->stream);
  // PEMAP_WALK_BLOCKS_PER_CU (default 4, swept 1..16): resident 256-lane blocks of the walk per CU; few enough walkers that
  // their direction lines stay in L2 between steps
  const int wbp = d->kn.walk_blocks_per_cu;
  int wgrid = (n_ends + 63) / 64;
  if (wgrid > d->n_cus * wbp * 4)
    wgrid = d->n_cus * wbp * 4;         // (waves: the knob counts blocks of four)
  // (the walk on the look-up stream, beside the next chunk's vote and SW, was tried: 107 ms per step against 103; it stays on the ALU
  // stream)
  hipStream_t ws = d->stream;
  unsigned long long *path = set2 ? d->d_path2 : d->d_path;
  uint16_t *nsteps = set2 ? d->d_nsteps2 : d->d_nsteps;
  hipLaunchKernelGGL (HIP_KERNEL_NAME (pm_walk_kernel < W, LPA >), dim3 (wgrid), dim3 (64), 0, ws, c.b, H, wins, ctr, d->d_cur,
                      dirbuf, c.tstride, pile_of (d), d->d_ins_log, d->ins_cap, path, d->path_words, nsteps);
  {
    // one wave per winning alignment applies the recorded steps to the pileup
    const int pbp = d->kn.pile_blocks_per_cu;
    int pgrid = n_ends;
    if (pgrid > d->n_cus * pbp * 4)
      pgrid = d->n_cus * pbp * 4;       // (waves: the knob counts blocks of four)
    hipLaunchKernelGGL (pm_pile_kernel, dim3 (pgrid), dim3 (64), 0, ws, c.b, H, wins, ctr, pile_of (d), path, d->path_words, nsteps);
  }
  hipEventRecord (ev[8], ws);
  if (split)
    hipEventRecord (d->ev_walk_done[slot], ws);
}

// wait for the run in flight and fold its chunks' counters and kernel times into the run's totals
static int absorb_run (pemap_dev * d)
{
  HIPCHK (d, hipStreamSynchronize (d->stream2));
  if (d->stream3)
    HIPCHK (d, hipStreamSynchronize (d->stream3));
  HIPCHK (d, hipStreamSynchronize (d->stream));
  const int nch = d->run_chunks;
#ifdef PEMAP_TIMING_PROBES
  {
    // the fused seed kernel's phase probes (pemap_seed4.hip.h): cycles summed over waves, printed per run
    unsigned long long pr[32], z[32] = { 0ull };
    if (hipMemcpyFromSymbol (pr, HIP_SYMBOL (pm_s4_probe), sizeof pr) == hipSuccess)
      {
        unsigned long long tot = 0;
        for (int i = 0; i < 16; i++)
          tot += pr[i];
        if (tot)
          {
            static const char *nm[16] = { "lines0+wait", "decode0", "lines1", "decode1", "records-end", "stage next", "segmask", "candidates", "relevant", "pairs", "rank+walk", "out", "rec:headers", "rec:scan+3", "-", "rec:tails" };
            fprintf (stderr, "[pm_s4_probe]");
            for (int i = 0; i < 16; i++)
              fprintf (stderr, " %s %.1f%%", nm[i], 100.0 * (double) pr[i] / (double) tot);
            fprintf (stderr, " | total %.3f G wave-cycles | segments %llu, dropped for a too-many bucket %llu, of them by the k-mer's own bucket %llu\n", (double) tot / 1e9, pr[16], pr[17], pr[18]);
            fprintf (stderr, "[pm_s4_probe] ends decoded %llu; passed over: the list's room %llu, positions next to candidates %llu\n", pr[20], pr[21], pr[23]);
          }
        (void) hipMemcpyToSymbol (HIP_SYMBOL (pm_s4_probe), z, sizeof z);
      }
  }
#endif
  std::vector < PmChunkCtr > hc (nch);
  HIPCHK (d, hipMemcpy (hc.data (), d->d_chunk_ctr, sizeof (PmChunkCtr) * nch, hipMemcpyDeviceToHost));
  HIPCHK (d, hipMemcpy (&d->last_cur, d->d_cur, sizeof (PmInsCursor), hipMemcpyDeviceToHost));
  PmCounters & t = d->last_ctr;
  for (int k = 0; k < nch; k++)
    {
      const PmCounters & c = hc[k].c;
      t.n_tasks_s += c.n_tasks_s;
      t.n_tasks_m += c.n_tasks_m;
      t.n_tasks_dp += pm_gapless_on (d) ? c.n_tasks_dp : c.n_tasks_s;
      t.sw_next[3] += pm_gapless_on (d) ? c.sw_next[3] : c.n_tasks_m;
      t.n_band[0] += c.n_band[0];
      t.n_band[1] += c.n_band[1];
      t.cells_band += c.cells_band;
      t.n_slots += c.n_slots;
      t.n_redo += c.n_redo;
      t.n_wins += c.n_wins;
      t.positions += c.positions + hc[k].positions;
      t.cells_score += c.cells_score;
      t.cells_dirs += c.cells_dirs;
      t.pile_incs += c.pile_incs;
      t.n_ins += c.n_ins;
      d->last_big += hc[k].n_big;
      d->last_big2 += hc[k].n_big2;
      hipEvent_t *ev = &d->evs[(size_t) k * PM_NEV];
      float ms = 0.f;
      if (d->run_split && hipEventElapsedTime (&ms, ev[0], ev[1]) == hipSuccess)
        {
          d->last_ms[0] += ms;
          d->last_ms[6] += ms;
        }
      if (hipEventElapsedTime (&ms, ev[2], ev[10]) == hipSuccess)
        d->last_ms[0] += ms;
      if (hipEventElapsedTime (&ms, ev[2], ev[3]) == hipSuccess)
        d->last_ms[7] += ms;
      for (int i = 1; i < 6; i++)
        if (hipEventElapsedTime (&ms, (i == 1) ? ev[10] : ev[2 + i], (i == 4) ? ev[9] : ev[3 + i]) == hipSuccess)
          d->last_ms[i] += ms;
    }
  if (d->last_cur.ins_overflow)
    return fail (d, "insertion log overflow (%u bytes): lower PEMAP_CHUNK_PAIRS or map smaller slices", d->ins_cap);
  if (d->last_cur.ins_bytes > d->ins_cap / 2)
    return drain_ins (d);
  return 0;
}

static int ensure_pipeline (pemap_dev * d, int chunk_ends)
{
  if (!d->stream2)
    {
      // (a CU mask and stream priorities for the look-up stream were tried in rounds 1 and 2 and gave nothing)
      HIPCHK (d, hipStreamCreateWithFlags (&d->stream2, hipStreamNonBlocking));
      for (int i = 0; i < 2; i++)
        {
          HIPCHK (d, hipEventCreateWithFlags (&d->ev_lists_ready[i], hipEventDisableTiming));
          HIPCHK (d, hipEventCreateWithFlags (&d->ev_lists_free[i], hipEventDisableTiming));
          HIPCHK (d, hipEventCreateWithFlags (&d->ev_walk_done[i], hipEventDisableTiming));
        }
      TRY (dev_alloc (d, &d->d_chunk_ctr, (size_t) PM_MAX_CHUNKS));
      d->evs.resize ((size_t) PM_MAX_CHUNKS * PM_NEV);
      for (size_t i = 0; i < d->evs.size (); i++)
        HIPCHK (d, hipEventCreate (&d->evs[i]));
    }
  if (chunk_ends > d->lists_cap || (chunk_ends > 0 && !pm_fused (d) && !d->lists_arrays))
    {
      if (chunk_ends < d->lists_cap)
        chunk_ends = d->lists_cap;
      // (the look-up kernels of a pending run may still be writing the old arrays)
      if (d->run_pending)
        {
          TRY (absorb_run (d));
          d->run_pending = false;
        }
      for (int i = 0; i < 2; i++)
        {
          hipFree (d->lists[i].hdr);
          hipFree (d->lists[i].key);
          hipFree (d->lists[i].seg);
          hipFree (d->lists[i].big_list);
          // (the fused seed kernel keeps the lists in LDS: only the big-end list is needed then)
          const size_t list_ends = pm_fused (d) ? 0 : (size_t) chunk_ends;
          d->lists[i].hdr = nullptr;
          d->lists[i].key = nullptr;
          d->lists[i].seg = nullptr;
          d->lists[i].big_list = nullptr;
          TRY (dev_alloc (d, &d->lists[i].hdr, list_ends));
          TRY (dev_alloc (d, &d->lists[i].key, list_ends * 2 * PM_SEED_CAP));
          TRY (dev_alloc (d, &d->lists[i].seg, list_ends * 2 * PM_SEED_CAP));
          // (first half: the ends the fused seed kernel passes over; second half: what its second tier leaves of them)
          TRY (dev_alloc (d, &d->lists[i].big_list, 2 * (size_t) chunk_ends));
        }
      d->lists_arrays = !pm_fused (d);
      d->lists_cap = chunk_ends;
    }
  return 0;
}

// chunk size run_slice cuts a run of n pairs of reads up to L bases into
static int chunk_pairs_for (const pemap_dev * d, int n, int L, bool split)
{
  const int per = d->paired ? 2 : 1;
  // one direction slab per read-end must fit the budget; the pipeline wants several chunks per run
  size_t slab_bytes = slab_dwords_for (d, L) * 4;
  long max_ends = (long) (dir_budget_bytes (d) / slab_bytes);
  if (max_ends > 20000000)
    max_ends = 20000000;        // task ids are end * 200 + hit in 32 bits
  int chunk = (int) (max_ends / per);
  const int want = d->kn.chunk_pairs;
  if (split && want > 0 && chunk > want)
    chunk = want;
  if (chunk < 1)
    chunk = 1;
  if (chunk > n)
    chunk = n;
  if ((n + chunk - 1) / chunk > PM_MAX_CHUNKS)
    chunk = (n + PM_MAX_CHUNKS - 1) / PM_MAX_CHUNKS;
  return chunk;
}

// copy_evs (may be NULL): one event per chunk, recorded behind the host-to-device copy of that chunk's rows; the first stream
// that touches the rows waits for it
static int run_slice (pemap_dev * d, int first, int n, int sync, const hipEvent_t * copy_evs = nullptr)
{
  HIPCHK (d, hipSetDevice (d->device));
  if (!d->index_ready)
    return fail (d, "run: no index loaded");
  if (d->n_staged <= 0)
    return fail (d, "run: no reads staged");
  if (first < 0 || n <= 0 || first + n > d->n_staged)
    return fail (d, "run: slice [%d, %d) outside the %d staged reads", first, first + n, d->n_staged);
  if (d->staged_paired != d->paired)
    return fail (d, "run: reads were staged in %s mode", d->staged_paired ? "paired" : "single");
  const int L = d->max_len_staged;
  const int per = d->paired ? 2 : 1;
  // PEMAP_PIPELINE: 1 (default) look-up kernel on a second stream beside vote/SW/walk of the previous chunk;
  // 0 monolithic seed kernel, one stream; 2 split kernels on one stream (diagnostic).
  const bool split = d->kn.pipeline != 0 && !d->kn.seed_phase;
  d->serial_split = d->kn.pipeline == 2;
  // With the fused seed kernel the big read-ends' remainder and the emit kernel run on the ALU stream for
  // reads of up to 160 bases (where the seed kernel is the longer side: 29.8 ms per step against 31.4), behind the seed kernel for
  // longer ones (2 x 245: 56.9 ms against 58.6)
  d->rest_on_alu = pm_fused (d) && seg_template (L) <= 10;
  // Where the vote runs -- 1: behind its look-ups on the memory stream (the fused kernel's remainder); 2: on a third stream of its own,
  // beside the SW / walk of the previous chunk and the look-ups of the next; 0: on the ALU stream.
  // 2 with the look-up replicas (measured 41.5 ms per step against 44.7 with the vote on the ALU stream: with the
  // cheap look-ups and the gapless rule no stream is saturated any more, and the vote of chunk k+1 fills the gaps), 0 without
  // (it was slower beside the look-ups of the reference's layout).
  { const int vm = (split && pm_fused (d)) ? 1 : (d->n_rep == 8 ? 2 : 0);
    d->vote_on_mem = split && !d->serial_split && vm != 0;
    d->vote_stream = (d->vote_on_mem && vm == 2) ? 3 : 2; }
  if (d->vote_on_mem && d->vote_stream == 3 && !d->stream3)
    {
      HIPCHK (d, hipStreamCreateWithFlags (&d->stream3, hipStreamNonBlocking));
      for (int i = 0; i < 2; i++)
        HIPCHK (d, hipEventCreateWithFlags (&d->ev_lookup_done[i], hipEventDisableTiming));
    }
  const int chunk = chunk_pairs_for (d, n, L, split);
  const int nch = (n + chunk - 1) / chunk;
  // Asynchronous runs queue up behind each other: the chunks of this run continue the pipeline of the pending ones (same
  // slots, same event chain), so that look-ups of this run's first chunk overlap the previous run's last.  The pending runs
  // are absorbed first only when the per-chunk bookkeeping would overflow or the geometry changes.
  int k0 = 0;
  if (d->run_pending)
    {
      // (a smaller chunk than the pending runs' fits their arrays: the tail of a batch continues the pipeline too)
      const bool same = d->run_split == split && d->run_L == L && split && !d->serial_split && chunk * per <= d->cap_ends
        && chunk * per <= d->lists_cap && (size_t) (chunk * per) < d->dir_slabs;
      if (same && d->run_chunks + nch <= PM_MAX_CHUNKS)
        k0 = d->run_chunks;
      else
        {
          TRY (absorb_run (d));
          d->run_pending = false;
        }
    }
  TRY (ensure_work (d, chunk * per, split));
  TRY (ensure_pipeline (d, chunk * per));
  RunCtx c;
  c.ix.pos_index = d->d_pos_index;
  c.ix.mers = d->d_mers;
  c.ix.genome = d->d_genome;
  c.ix.contig_starts = d->d_contig_starts;
  c.ix.n_mers = d->n_mers;
  c.ix.gsize = d->gsize;
  c.ix.n_contigs = d->n_contigs;
  c.ix.idepth = d->idepth;
  c.ix.rep = d->d_rep;
  c.ix.multi = d->d_multi;
  c.ix.multi_base = d->multi_base;
  c.ix.n_rep = d->n_rep;
  c.prm.min_dist = d->min_dist;
  c.prm.max_dist = d->max_dist;
  c.prm.min_align = d->min_align;
  c.prm.bisulfite = d->bisulfite;
  c.L = L;
  c.tstride = tstride_for (d, L);
  // (the last slab of the allocation, whatever this run's chunk size: chunks of earlier, larger runs may still be in flight)
  c.dump_slab = d->d_dirbuf + (d->dir_slabs - 1) * slab_dwords_for (d, L);
  if (k0 == 0)
    {
      memset (&d->last_ctr, 0, sizeof (d->last_ctr));
      memset (d->last_ms, 0, sizeof (d->last_ms));
      d->last_big = d->last_big2 = 0;
      d->run_ends = 0;
    }
  d->run_serial++;
  d->run_first = first;
  d->run_n = n;
  d->run_ends += (uint64_t) n * per;
  d->run_chunks = k0 + nch;
  d->run_split = split;
  d->run_chunk_pairs = chunk;
  d->run_L = L;
  // fresh per-chunk counters: zeroed on the stream that touches them first (the look-up stream in the split pipeline, so
  // that a queued run's first look-ups do not wait for the previous run's ALU work)
  HIPCHK (d, hipMemsetAsync (d->d_chunk_ctr + k0, 0, sizeof (PmChunkCtr) * nch, (split && !d->serial_split) ? d->stream2 : d->stream));
  auto batch_of = [&] (int k, PmBatch & bb, int &f, int &m)
  {
    const int off = k * chunk;
    m = (n - off < chunk) ? n - off : chunk;
    f = first + off;
    bb.reads1 = d->d_reads1 + (size_t) f * d->stride;
    bb.reads2 = d->paired ? d->d_reads2 + (size_t) f * d->stride : nullptr;
    bb.len1 = d->d_len1 + f;
    bb.len2 = d->paired ? d->d_len2 + f : nullptr;
    bb.n = m;
    bb.stride = d->stride;
    bb.paired = d->paired;
    bb.n_ends = m * per;
  };
  auto enqueue_lookup = [&] (int k) -> int
  {
    int f, m;
    RunCtx cl = c;
    batch_of (k, cl.b, f, m);
    const int g = k0 + k, slot = g & 1;
    if (copy_evs)
      HIPCHK (d, hipStreamWaitEvent (d->serial_split ? d->stream : d->stream2, copy_evs[k], 0));
    // the slot's lists must have been consumed by the vote of chunk g-2
    if (g >= 2)
      HIPCHK (d, hipStreamWaitEvent (d->serial_split ? d->stream : d->stream2, d->ev_lists_free[slot], 0));
    if (pm_fused (d))
      {
        // the fused kernel writes the hit arrays that the SW / walk of chunk g-2 used; the list-mode remainder of the big read-ends
        // and the emit kernel follow it on the same stream
        hipStream_t fs = d->serial_split ? d->stream : d->stream2;
        if (g >= 2 && !d->serial_split)
          HIPCHK (d, hipStreamWaitEvent (fs, d->ev_walk_done[slot], 0));
        launch_lookup (d, cl, slot, d->d_chunk_ctr + g, &d->evs[(size_t) g * PM_NEV], true);
        // the monolithic kernel for what both tiers passed over: 256-thread workgroups with 31 KB of LDS, which find no room beside the
        // next chunk's persistent seed waves -- here, between two seed launches, they do (and mostly find nothing to do).  The emit
        // kernel goes to the ALU stream in front of the chunk's DP (rest_on_alu), or follows here
        launch_vote (d, cl, true, slot, d->d_chunk_ctr + g, &d->evs[(size_t) g * PM_NEV], fs, (pm_vote_rest_on_alu (d) && !d->serial_split) ? 3 : 2);
        HIPCHK (d, hipEventRecord (d->ev_lists_ready[slot], fs));
        return 0;
      }
    launch_lookup (d, cl, slot, d->d_chunk_ctr + g, &d->evs[(size_t) g * PM_NEV], true);
    if (d->vote_on_mem)
      {
        // the vote fills the array set that the SW / walk of chunk g-2 used
        hipStream_t vs = d->vote_stream == 3 ? d->stream3 : d->stream2;
        if (vs != d->stream2)
          {
            HIPCHK (d, hipEventRecord (d->ev_lookup_done[slot], d->stream2));
            HIPCHK (d, hipStreamWaitEvent (vs, d->ev_lookup_done[slot], 0));
          }
        if (g >= 2)
          HIPCHK (d, hipStreamWaitEvent (vs, d->ev_walk_done[slot], 0));
        launch_vote (d, cl, true, slot, d->d_chunk_ctr + g, &d->evs[(size_t) g * PM_NEV], vs, (d->vote_stream == 3 && pm_vote_rest_on_alu (d)) ? 1 : 0);
        HIPCHK (d, hipEventRecord (d->ev_lists_ready[slot], vs));
        return 0;
      }
    HIPCHK (d, hipEventRecord (d->ev_lists_ready[slot], d->serial_split ? d->stream : d->stream2));
    return 0;
  };
  // memory stream order: lookup(0), lookup(1), then per chunk k: walk(k), lookup(k+2) -- the look-ups stay one chunk ahead
  if (split && !d->serial_split)
    for (int k = 0; k < 2 && k < nch; k++)
      TRY (enqueue_lookup (k));
  for (int k = 0; k < nch; k++)
    {
      int f, m;
      batch_of (k, c.b, f, m);
      const int g = k0 + k, slot = g & 1;
      PmChunkCtr *cc = d->d_chunk_ctr + g;
      hipEvent_t *ev = &d->evs[(size_t) g * PM_NEV];
      if (copy_evs && !(split && !d->serial_split))
        HIPCHK (d, hipStreamWaitEvent (d->stream, copy_evs[k], 0));
      if (split)
        {
          if (d->serial_split)
            TRY (enqueue_lookup (k));
          HIPCHK (d, hipStreamWaitEvent (d->stream, d->ev_lists_ready[slot], 0));
        }
      uint32_t *m1 = d->d_m1 + f, *m2 = d->paired ? d->d_m2 + f : nullptr;
      int *mt = d->d_mtype + f;
      {
        int lanes, w;
        pick_geom (d, L, &lanes, &w);
#define PM_CH(WW, LL) launch_chunk < WW, LL > (d, c, m1, m2, mt, split, slot, cc, ev)
        if (lanes == 8)
          {
            if (w == 13)
              PM_CH (13, 8);
            else
              PM_CH (19, 8);
          }
        else
          switch (w)
            {
            case 10: PM_CH (10, 16); break;
            case 13: PM_CH (13, 16); break;
            case 16: PM_CH (16, 16); break;
            default: PM_CH (19, 16); break;
            }
#undef PM_CH
      }
      HIPCHK (d, hipGetLastError ());
      if (split && !d->serial_split && k + 2 < nch)
        TRY (enqueue_lookup (k + 2));
    }
  if (d->rest_twice)
    return fail (d, "internal: the seed-stage remainder of a chunk was enqueued twice");
  d->run_pending = true;
  if (sync)
    {
      TRY (absorb_run (d));
      d->run_pending = false;
    }
  return 0;
}

extern "C" int pemap_dev_run (pemap_dev * d, int sync)
{
  return run_slice (d, 0, d->n_staged, sync);
}

extern "C" int pemap_dev_run_slice (pemap_dev * d, int first, int n, int sync)
{
  return run_slice (d, first, n, sync);
}

extern "C" int pemap_dev_sync (pemap_dev * d)
{
  HIPCHK (d, hipSetDevice (d->device));
  if (d->run_pending)
    {
      TRY (absorb_run (d));
      d->run_pending = false;
    }
  HIPCHK (d, hipStreamSynchronize (d->stream));
  return 0;
}

// drain the device insertion log into the host copy and reset the cursor (stream must be idle)
static int drain_ins (pemap_dev * d)
{
  HIPCHK (d, hipStreamSynchronize (d->stream));
  PmInsCursor cur;
  HIPCHK (d, hipMemcpy (&cur, d->d_cur, sizeof (cur), hipMemcpyDeviceToHost));
  unsigned nb = cur.ins_bytes;
  if (nb > d->ins_cap)
    nb = d->ins_cap;
  if (nb)
    {
      size_t at = d->h_ins.size ();
      d->h_ins.resize (at + nb);
      HIPCHK (d, hipMemcpy (d->h_ins.data () + at, d->d_ins_log, nb, hipMemcpyDeviceToHost));
    }
  HIPCHK (d, hipMemset (d->d_cur, 0, sizeof (PmInsCursor)));
  d->last_cur.ins_bytes = 0;
  return 0;
}

extern "C" int pemap_dev_collect (pemap_dev * d, uint32_t * m1, uint32_t * m2, int *mapping_type)
{
  TRY (pemap_dev_sync (d));
  const int n = d->run_n, first = d->run_first;
  if (n <= 0)
    return fail (d, "collect: nothing was run");
  HIPCHK (d, hipMemcpy (m1, d->d_m1 + first, (size_t) n * sizeof (uint32_t), hipMemcpyDeviceToHost));
  if (d->paired && m2)
    HIPCHK (d, hipMemcpy (m2, d->d_m2 + first, (size_t) n * sizeof (uint32_t), hipMemcpyDeviceToHost));
  HIPCHK (d, hipMemcpy (mapping_type, d->d_mtype + first, (size_t) n * sizeof (int), hipMemcpyDeviceToHost));
  TRY (drain_ins (d));
  fold_summary (d, first, n, m1, m2, mapping_type);
  return 0;
}

// ---- batches in flight -------------------------------------------------------------------------------------
// The reference hands a filled batch to a worker thread and goes on reading (pthread_create at pemapper.c:684; the batch's
// mutex is released when the worker is done, 1307).  submit / wait are that seam: submit queues the batch's host-to-device
// copies (their own stream) and its kernels (the object's pipeline, continued from the batch before) and returns; wait blocks
// until the batch's m1 / m2 / mapping_type are in the caller's buffers.  With two or three batches in flight the copies of
// batch k + 1 and the results of batch k - 1 move while batch k is computed, and the pipeline never drains between calls.

// result fold of one finished batch, pemapper.c:1238-1265
static void fold_summary (pemap_dev * d, int first, int n, const uint32_t * m1, const uint32_t * m2, const int *mapping_type)
{
  long *S = d->summary;
  for (int j = 0; j < n; j++)
    {
      uint32_t a = m1[j], b = (d->paired && m2) ? m2[j] : 0;
      int la = d->h_len1[first + j], lb = d->paired ? d->h_len2[first + j] : 0;
      S[4 + mapping_type[j]]++;
      if (a)
        {
          S[0]++;
          S[1] += la;
          if (b)
            {
              S[0]++;
              S[1] += lb;
              long test = (long) (uint32_t) (a - b);    // unsigned difference widened to long, pemapper.c:1250
              if (test < (long) d->max_dist * 4)
                {
                  S[2] += test;
                  S[3]++;
                }
            }
        }
      else if (b)
        {
          S[0]++;
          S[1] += lb;
        }
    }
}

// is [p, p + bytes) inside a range pinned by the caller?
bool pm_host_pin_lookup (const void *p, size_t bytes)
{
  std::lock_guard < std::mutex > lk (g_pin_mu);
  const char *c = (const char *) p;
  for (size_t i = 0; i < g_pinned.size (); i++)
    if (c >= g_pinned[i].base && c + bytes <= g_pinned[i].base + g_pinned[i].bytes)
      return true;
  return false;
}

// register a host range for DMA on behalf of pemap_dev_pin_host.  Registered ranges that share pages with the new one are
// replaced by the union of all of them (a copy must lie inside one registration); the calling object's copy stream is drained
// before a registration is dropped, copies out of it may be queued.
bool pm_host_pin_range (const void *p, size_t bytes, hipStream_t copy_stream)
{
  std::lock_guard < std::mutex > lk (g_pin_mu);
  const size_t page = 4096;
  char *lo = (char *) ((uintptr_t) p & ~(uintptr_t) (page - 1));
  char *hi = (char *) (((uintptr_t) p + bytes + page - 1) & ~(uintptr_t) (page - 1));
  int users = 1;
  for (size_t i = 0; i < g_pinned.size (); i++)
    if (lo >= g_pinned[i].base && hi <= g_pinned[i].base + g_pinned[i].bytes)
      {
        g_pinned[i].users++;
        return true;
      }
  for (size_t i = 0; i < g_pinned.size ();)
    if (lo < g_pinned[i].base + g_pinned[i].bytes && g_pinned[i].base < hi)
      {
        if (g_pinned[i].base < lo)
          lo = g_pinned[i].base;
        if (g_pinned[i].base + g_pinned[i].bytes > hi)
          hi = g_pinned[i].base + g_pinned[i].bytes;
        users += g_pinned[i].users;
        if (copy_stream)
          (void) hipStreamSynchronize (copy_stream);
        (void) hipHostUnregister (g_pinned[i].base);
        (void) hipGetLastError ();
        g_pinned.erase (g_pinned.begin () + i);
      }
    else
      i++;
  if (hipHostRegister (lo, (size_t) (hi - lo), hipHostRegisterDefault) != hipSuccess)
    {
      (void) hipGetLastError ();
      return false;
    }
  PmPinned e;
  e.base = lo;
  e.bytes = (size_t) (hi - lo);
  e.users = users;
  g_pinned.push_back (e);
  return true;
}

// host copy into a pinned staging buffer, on a few threads when it is large (one core moves ~10 GB/s: 330 MB per million pairs)
void pm_par_memcpy (char *dst, const char *src, size_t bytes)
{
  const size_t piece = (size_t) 4 << 20;
  if (bytes < 2 * piece)
    {
      memcpy (dst, src, bytes);
      return;
    }
  const int nt = 4;
  std::thread th[nt];
  const size_t per = ((bytes / nt) + 63) & ~(size_t) 63;
  for (int t = 0; t < nt; t++)
    {
      const size_t o = per * t, m = o >= bytes ? 0 : (bytes - o < per ? bytes - o : per);
      th[t] = std::thread ([=] { if (m) memcpy (dst + o, src + o, m); });
    }
  for (int t = 0; t < nt; t++)
    th[t].join ();
}

extern "C" int pemap_dev_pin_host (pemap_dev * d, const void *host_ptr, uint64_t n_bytes)
{
  HIPCHK (d, hipSetDevice (d->device));
  if (!host_ptr || !n_bytes)
    return fail (d, "pin_host: empty range");
  if (!pm_host_pin_range (host_ptr, (size_t) n_bytes, d->stream_h2d))
    return fail (d, "pin_host: hipHostRegister of %llu bytes failed", (unsigned long long) n_bytes);
  return 0;
}

// undo one pm_host_pin_range of the range that holds host_ptr: 0 done, 1 not a pinned range, 2 the runtime refused
int pm_host_unpin (const void *host_ptr)
{
  std::lock_guard < std::mutex > lk (g_pin_mu);
  const char *c = (const char *) host_ptr;
  for (size_t i = 0; i < g_pinned.size (); i++)
    if (c >= g_pinned[i].base && c < g_pinned[i].base + g_pinned[i].bytes && g_pinned[i].users > 0)
      {
        if (--g_pinned[i].users == 0)
          {
            const hipError_t e = hipHostUnregister (g_pinned[i].base);
            g_pinned.erase (g_pinned.begin () + i);
            if (e != hipSuccess)
              return 2;
          }
        return 0;
      }
  return 1;
}

extern "C" int pemap_dev_unpin_host (pemap_dev * d, const void *host_ptr)
{
  HIPCHK (d, hipSetDevice (d->device));
  // copies out of the range may still be queued on this object's copy stream
  {
    std::lock_guard < std::mutex > lk (d->mu);
    if (d->stream_h2d)
      HIPCHK (d, hipStreamSynchronize (d->stream_h2d));
  }
  const int rc = pm_host_unpin (host_ptr);
  if (rc == 1)
    return fail (d, "unpin_host: %p is not inside a range pinned through pemap_dev_pin_host", host_ptr);
  if (rc == 2)
    return fail (d, "unpin_host: hipHostUnregister failed");
  return 0;
}

// wait for the batch in `slot` and deliver it (mu held through lk; released while the host blocks on the event)
static int ring_finish (pemap_dev * d, int slot, std::unique_lock < std::mutex > &lk)
{
  PmRingSlot & r = d->ring[slot];
  if (!r.active)
    return 0;
  const unsigned long long seq = r.seq;
  hipEvent_t ev = r.ev_done;
  lk.unlock ();
  hipError_t e = hipEventSynchronize (ev);
  lk.lock ();
  if (e != hipSuccess)
    return fail (d, "wait_batch: %s", hipGetErrorString (e));
  if (!r.active || r.seq != seq)
    return 0;                   // another thread delivered it meanwhile
  const int n = r.n;
  memcpy (r.m1, r.h_res, (size_t) n * 4);
  if (d->paired && r.m2)
    memcpy (r.m2, r.h_res + d->ring_cap, (size_t) n * 4);
  memcpy (r.mt, r.h_res + 2 * (size_t) d->ring_cap, (size_t) n * 4);
  fold_summary (d, r.first, n, r.m1, r.m2, r.mt);
  r.active = false;
  return 0;
}

static int ring_finish_all (pemap_dev * d, std::unique_lock < std::mutex > &lk)
{
  // oldest first, so that the summary folds in submission order
  for (;;)
    {
      int pick = -1;
      for (int i = 0; i < PM_RING; i++)
        if (d->ring[i].active && (pick < 0 || d->ring[i].seq < d->ring[pick].seq))
          pick = i;
      if (pick < 0)
        return 0;
      TRY (ring_finish (d, pick, lk));
    }
}

// the staged arrays go back to holding one resident read set (stage_reads, synth_reads): no batch may be in flight
static int ring_leave (pemap_dev * d)
{
  std::unique_lock < std::mutex > sub (d->submit_mu);
  std::unique_lock < std::mutex > lk (d->mu);
  TRY (ring_finish_all (d, lk));
  d->ring_cap = 0;
  return 0;
}

static int ring_setup (pemap_dev * d, int n, int stride, std::unique_lock < std::mutex > &lk)
{
  if (!d->stream_h2d)
    {
      HIPCHK (d, hipStreamCreateWithFlags (&d->stream_h2d, hipStreamNonBlocking));
    }
  if (d->ring_cap >= n && d->stride == stride && d->staged_paired == d->paired && (!d->paired || d->d_reads2))
    return 0;
  // a different geometry: everything in flight is delivered and accounted first
  TRY (ring_finish_all (d, lk));
  if (d->run_pending)
    {
      TRY (absorb_run (d));
      d->run_pending = false;
    }
  HIPCHK (d, hipDeviceSynchronize ());
  int cap = n > d->ring_cap ? n : d->ring_cap;
  if (cap < 1024)
    cap = 1024;
  if ((long) cap * PM_RING > 2000000000L)
    return fail (d, "submit_batch: %d reads per batch are too many", n);
  TRY (ensure_reads (d, cap * PM_RING, stride, d->paired));
  for (int i = 0; i < PM_RING; i++)
    {
      PmRingSlot & r = d->ring[i];
      if (r.h_len)
        hipHostFree (r.h_len);
      if (r.h_res)
        hipHostFree (r.h_res);
      r.h_len = nullptr;
      r.h_res = nullptr;
      if (r.h_rows)
        hipHostFree (r.h_rows);
      r.h_rows = nullptr;
      r.h_rows_bytes = 0;
      HIPCHK (d, hipHostMalloc ((void **) &r.h_len, (size_t) cap * 2 * sizeof (int), hipHostMallocDefault));
      HIPCHK (d, hipHostMalloc ((void **) &r.h_res, (size_t) cap * 3 * sizeof (uint32_t), hipHostMallocDefault));
      if (!r.ev_done)
        HIPCHK (d, hipEventCreateWithFlags (&r.ev_done, hipEventDisableTiming));
    }
  d->ring_cap = cap;
  d->n_staged = cap * PM_RING;
  d->staged_paired = d->paired;
  d->max_len_staged = 0;
  d->min_len_staged = 1 << 30;
  d->h_len1.assign ((size_t) cap * PM_RING, 0);
  d->h_len2.assign (d->paired ? (size_t) cap * PM_RING : 0, 0);
  return 0;
}

extern "C" int pemap_dev_submit_batch (pemap_dev * d, const char *reads1, const int *len1, const char *reads2, const int *len2,
                                       int n, int stride, uint32_t * m1, uint32_t * m2, int *mapping_type, uint64_t * ticket)
{
  std::unique_lock < std::mutex > sub (d->submit_mu);   // ring_seq, the slot and `first` below stay this call's own while mu is dropped
  std::unique_lock < std::mutex > lk (d->mu);
  HIPCHK (d, hipSetDevice (d->device));
  if (!d->index_ready)
    return fail (d, "submit_batch: no index loaded");
  if (n <= 0)
    return fail (d, "submit_batch: n = %d", n);
  if (!reads1 || !len1 || !m1 || !mapping_type || !ticket)
    return fail (d, "submit_batch: a required pointer is NULL");
  if (d->paired && (!reads2 || !len2 || !m2))
    return fail (d, "submit_batch: paired mode needs reads2 / len2 / m2");
  if (stride < PEMAP_MIN_READ)
    return fail (d, "submit_batch: stride %d", stride);
  int mx = 0, mn = 1 << 30;
  TRY (check_lengths (d, len1, n, &mx, &mn));
  if (d->paired)
    TRY (check_lengths (d, len2, n, &mx, &mn));
  if (mx > stride)
    return fail (d, "submit_batch: a read is longer than the row stride %d", stride);
  TRY (ring_setup (d, n, stride, lk));
  const int slot = (int) (d->ring_seq % PM_RING);
  PmRingSlot & r = d->ring[slot];
  TRY (ring_finish (d, slot, lk));      // the batch that used the slot PM_RING submissions ago
  const int first = slot * d->ring_cap;
  // the kernels' geometry follows the longest read seen since the ring was set up (any upper bound gives the same results;
  // a bound that never shrinks keeps consecutive batches in one pipeline)
  if (mx > d->max_len_staged)
    d->max_len_staged = mx;
  if (mn < d->min_len_staged)
    d->min_len_staged = mn;
  memcpy (&d->h_len1[first], len1, (size_t) n * sizeof (int));
  memcpy (r.h_len, len1, (size_t) n * sizeof (int));
  HIPCHK (d, hipMemcpyAsync (d->d_len1 + first, r.h_len, (size_t) n * sizeof (int), hipMemcpyHostToDevice, d->stream_h2d));
  if (d->paired)
    {
      memcpy (&d->h_len2[first], len2, (size_t) n * sizeof (int));
      memcpy (r.h_len + d->ring_cap, len2, (size_t) n * sizeof (int));
      HIPCHK (d, hipMemcpyAsync (d->d_len2 + first, r.h_len + d->ring_cap, (size_t) n * sizeof (int), hipMemcpyHostToDevice, d->stream_h2d));
    }
  // the rows move by DMA straight out of the caller's buffers when the caller pinned them (pemap_dev_pin_host); otherwise they
  // are copied, slice by slice, into the slot's pinned staging buffer first (the DMA of slice k runs beside the host copy of k + 1)
  const bool direct1 = pm_host_pin_lookup (reads1, (size_t) n * stride);
  const bool direct2 = d->paired ? pm_host_pin_lookup (reads2, (size_t) n * stride) : true;
  if (!direct1 || !direct2)
    {
      const size_t need = (size_t) d->ring_cap * stride * 2;
      if (r.h_rows_bytes < need)
        {
          if (r.h_rows)
            hipHostFree (r.h_rows);
          r.h_rows = nullptr;
          r.h_rows_bytes = 0;
          HIPCHK (d, hipHostMalloc ((void **) &r.h_rows, need, hipHostMallocDefault));
          r.h_rows_bytes = need;
        }
    }
  if (!d->stream2)
    TRY (ensure_pipeline (d, 0));       // the pipeline's streams must exist before the first copy event is waited on
  // the copies are cut like the kernels' chunks, one event each: chunk k's look-ups start when its rows have landed
  const bool split = d->kn.pipeline != 0 && !d->kn.seed_phase;
  const int slice = chunk_pairs_for (d, n, d->max_len_staged, split);
  const int n_slices = (n + slice - 1) / slice;
  while ((int) r.ev_copy.size () < n_slices)
    {
      hipEvent_t e;
      HIPCHK (d, hipEventCreateWithFlags (&e, hipEventDisableTiming));
      r.ev_copy.push_back (e);
    }
  for (int k = 0, off = 0; off < n; off += slice, k++)
    {
      const int m = n - off < slice ? n - off : slice;
      const char *src1 = reads1 + (size_t) off * stride, *src2 = d->paired ? reads2 + (size_t) off * stride : nullptr;
      if (!direct1)
        {
          char *stg = r.h_rows + (size_t) off * stride;
          pm_par_memcpy (stg, src1, (size_t) m * stride);
          src1 = stg;
        }
      if (d->paired && !direct2)
        {
          char *stg = r.h_rows + ((size_t) d->ring_cap + off) * stride;
          pm_par_memcpy (stg, src2, (size_t) m * stride);
          src2 = stg;
        }
      HIPCHK (d, hipMemcpyAsync (d->d_reads1 + (size_t) (first + off) * stride, src1, (size_t) m * stride, hipMemcpyHostToDevice, d->stream_h2d));
      if (d->paired)
        HIPCHK (d, hipMemcpyAsync (d->d_reads2 + (size_t) (first + off) * stride, src2, (size_t) m * stride, hipMemcpyHostToDevice, d->stream_h2d));
      HIPCHK (d, hipEventRecord (r.ev_copy[k], d->stream_h2d));
    }
  TRY (run_slice (d, first, n, 0, r.ev_copy.data ()));
  // results: on the ALU stream itself, behind the batch's last kernel (every other stream's work for the batch precedes it).
  // Not on a stream of their own: HIP multiplexes streams onto a few hardware queues, and a copy stream parked on "batch k is
  // done" held up whichever pipeline stream shared its queue -- 3.5 ms per batch (measured: 45.2 ms per step against 41.7).
  hipStream_t rs = d->stream;
  HIPCHK (d, hipMemcpyAsync (r.h_res, d->d_m1 + first, (size_t) n * 4, hipMemcpyDeviceToHost, rs));
  if (d->paired)
    HIPCHK (d, hipMemcpyAsync (r.h_res + d->ring_cap, d->d_m2 + first, (size_t) n * 4, hipMemcpyDeviceToHost, rs));
  HIPCHK (d, hipMemcpyAsync (r.h_res + 2 * (size_t) d->ring_cap, d->d_mtype + first, (size_t) n * 4, hipMemcpyDeviceToHost, rs));
  HIPCHK (d, hipEventRecord (r.ev_done, rs));
  // the slot becomes a batch in flight only now that its event is recorded: an error above leaves it free, and nothing stale is
  // ever "delivered" into the caller's buffers or folded into the summary
  r.active = true;
  r.seq = d->ring_seq;
  r.n = n;
  r.first = first;
  r.m1 = m1;
  r.m2 = m2;
  r.mt = mapping_type;
  *ticket = d->ring_seq++;
  return 0;
}

extern "C" int pemap_dev_wait_batch (pemap_dev * d, uint64_t ticket)
{
  std::unique_lock < std::mutex > lk (d->mu);
  HIPCHK (d, hipSetDevice (d->device));
  if (ticket >= d->ring_seq)
    return fail (d, "wait_batch: ticket %llu was never handed out", (unsigned long long) ticket);
  const int slot = (int) (ticket % PM_RING);
  if (!d->ring[slot].active || d->ring[slot].seq != ticket)
    return 0;                   // delivered already (by a later submit that needed the slot, or by another wait)
  return ring_finish (d, slot, lk);
}

extern "C" int pemap_dev_map_batch (pemap_dev * d, const char *reads1, const int *len1, const char *reads2, const int *len2,
                                    int n, int stride, uint32_t * m1, uint32_t * m2, int *mapping_type)
{
  // submit + wait.  Called from one thread this fills and drains the pipeline once per call; several host threads calling it
  // on the same object (the reference's worker threads, one batch each) overlap like explicit submit / wait pairs do.
  uint64_t t = 0;
  TRY (pemap_dev_submit_batch (d, reads1, len1, reads2, len2, n, stride, m1, m2, mapping_type, &t));
  TRY (pemap_dev_wait_batch (d, t));
  std::unique_lock < std::mutex > lk (d->mu);
  bool idle = true;
  for (int i = 0; i < PM_RING; i++)
    idle = idle && !d->ring[i].active;
  if (idle && d->run_pending)
    {
      // nothing else in flight: account the run now (counters, kernel times, insertion log), so that run_stats describes this call
      TRY (absorb_run (d));
      d->run_pending = false;
    }
  return 0;
}

extern "C" int pemap_dev_summary (pemap_dev * d, long *out13)
{
  memcpy (out13, d->summary, sizeof (d->summary));
  return 0;
}

extern "C" int pemap_dev_run_stats (pemap_dev * d, uint64_t * s, float *t)
{
  TRY (pemap_dev_sync (d));     // a run still in flight is accounted first
  const PmCounters & c = d->last_ctr;
  if (s)
    {
      s[0] = d->run_ends;
      s[1] = c.positions;
      s[2] = (uint64_t) c.n_tasks_s + c.n_tasks_m;
      s[3] = (uint64_t) c.n_tasks_dp + c.n_band[0] + c.n_redo;
      s[4] = c.cells_score;
      s[5] = c.cells_dirs;
      s[6] = c.pile_incs;
      s[7] = c.n_ins;
      s[8] = c.n_wins;
      s[9] = c.n_redo;
      s[10] = d->last_big;
      s[11] = (uint64_t) d->run_chunks;
      s[12] = ((uint64_t) c.n_tasks_s - c.n_tasks_dp - c.n_band[0]) + ((uint64_t) c.n_tasks_m - c.sw_next[3] - c.n_band[1]);
      s[13] = (uint64_t) c.n_band[0] + c.n_band[1];
      s[14] = c.cells_band;
      s[15] = d->last_big2;
    }
  if (t)
    memcpy (t, d->last_ms, sizeof (d->last_ms));
  return 0;
}

extern "C" int pemap_dev_debug_hits (pemap_dev * d, int *n_hits, uint32_t * spot, uint8_t * orient, uint32_t * win_start,
                                     int *win_len, double *score, int *start_k, int *start_i)
{
  HIPCHK (d, hipSetDevice (d->device));
  HIPCHK (d, hipStreamSynchronize (d->stream));
  const int n_ends = d->paired ? 2 * d->run_n : d->run_n;
  const size_t nh = (size_t) n_ends * PM_MAX_HITS;
  if (n_hits)
    HIPCHK (d, hipMemcpy (n_hits, d->hits.n_hits, (size_t) n_ends * sizeof (int), hipMemcpyDeviceToHost));
  if (spot)
    HIPCHK (d, hipMemcpy (spot, d->hits.spot, nh * 4, hipMemcpyDeviceToHost));
  if (orient)
    HIPCHK (d, hipMemcpy (orient, d->hits.orient, nh, hipMemcpyDeviceToHost));
  if (win_start)
    HIPCHK (d, hipMemcpy (win_start, d->hits.gpos, nh * 4, hipMemcpyDeviceToHost));
  if (score)
    HIPCHK (d, hipMemcpy (score, d->hits.score, nh * 8, hipMemcpyDeviceToHost));
  if (win_len || start_i)
    {
      std::vector < int16_t > t (nh);
      if (win_len)
        {
          HIPCHK (d, hipMemcpy (t.data (), d->hits.nn, nh * 2, hipMemcpyDeviceToHost));
          for (size_t i = 0; i < nh; i++)
            win_len[i] = t[i];
        }
      if (start_i)
        {
          HIPCHK (d, hipMemcpy (t.data (), d->hits.sti, nh * 2, hipMemcpyDeviceToHost));
          for (size_t i = 0; i < nh; i++)
            start_i[i] = t[i];
        }
    }
  if (start_k)
    {
      std::vector < uint8_t > t (nh);
      HIPCHK (d, hipMemcpy (t.data (), d->hits.stk, nh, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < nh; i++)
        start_k[i] = t[i] & 3;  // (bit 2 = decided by the gapless rule)
    }
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
extern "C" int pemap_dev_reset_pileup (pemap_dev * d)
{
  HIPCHK (d, hipSetDevice (d->device));
  if (!d->d_counts)
    return fail (d, "reset_pileup: no index");
  HIPCHK (d, hipStreamSynchronize (d->stream));
  HIPCHK (d, hipMemset (d->d_counts, 0, 6 * d->pile_plane_words * sizeof (uint32_t)));
  HIPCHK (d, hipMemset (d->d_cur, 0, sizeof (PmInsCursor)));
  d->h_ins.clear ();
  memset (d->summary, 0, sizeof (d->summary));
  return 0;
}

extern "C" int pemap_dev_fetch_pileup (pemap_dev * d, uint16_t * counts, pemap_ins_cb cb, void *user)
{
  HIPCHK (d, hipSetDevice (d->device));
  if (!d->d_counts)
    return fail (d, "fetch_pileup: no index");
  TRY (pemap_dev_sync (d));
  TRY (drain_ins (d));
  if (counts)
    {
      const uint64_t chunk = 48ull << 20;       // positions per round (6 counters each)
      uint16_t *d_tmp = nullptr;
      TRY (dev_alloc (d, &d_tmp, (size_t) (d->gsize < chunk ? d->gsize : chunk) * 6));
      for (uint64_t o = 0; o < d->gsize; o += chunk)
        {
          const uint64_t m = d->gsize - o < chunk ? d->gsize - o : chunk;
          hipLaunchKernelGGL (pile_to_u16_kernel, dim3 ((unsigned) ((m * 6 + 255) / 256)), dim3 (256), 0, d->stream, pile_of (d), o, m, d_tmp);
          HIPCHK (d, hipStreamSynchronize (d->stream));
          HIPCHK (d, hipMemcpy (counts + o * 6, d_tmp, m * 6 * sizeof (uint16_t), hipMemcpyDeviceToHost));
        }
      hipFree (d_tmp);
    }
  if (cb)
    {
      size_t at = 0;
      const std::vector < uint8_t > &L = d->h_ins;
      char buf[512];
      while (at + 8 <= L.size ())
        {
          uint32_t pos, len;
          memcpy (&pos, &L[at], 4);
          memcpy (&len, &L[at + 4], 4);
          if (len == 0 || len > 300 || at + 8 + len > L.size ())
            return fail (d, "fetch_pileup: corrupt insertion log at byte %zu", at);
          memcpy (buf, &L[at + 8], len);
          buf[len] = 0;
          cb (user, pos, buf, (int) len);
          at += 8 + ((len + 3u) & ~3u);
        }
    }
  return 0;
}

extern "C" int pemap_dev_fetch_records (pemap_dev * d, uint64_t first, uint64_t count, void *out, uint64_t out_capacity,
                                        uint64_t * n_records)
{
  HIPCHK (d, hipSetDevice (d->device));
  if (!d->d_counts)
    return fail (d, "fetch_records: no index");
  if (first + count > d->gsize)
    return fail (d, "fetch_records: range beyond the genome");
  if (count == 0)
    {
      *n_records = 0;
      return 0;
    }
  if (count > (1ull << 31))
    return fail (d, "fetch_records: at most 2^31 sites per call");
  HIPCHK (d, hipStreamSynchronize (d->stream));
  const uint64_t n_tiles = (count + PR_BLOCK - 1) / PR_BLOCK;
  uint32_t *d_tc = nullptr;
  uint64_t *d_to = nullptr, *d_total = nullptr;
  TRY (dev_alloc (d, &d_tc, n_tiles));
  TRY (dev_alloc (d, &d_to, n_tiles));
  TRY (dev_alloc (d, &d_total, 1));
  hipLaunchKernelGGL (pile_count_kernel, dim3 ((unsigned) n_tiles), dim3 (PR_BLOCK), 0, d->stream, pile_of (d), first, count, d_tc);
  hipLaunchKernelGGL (ix_scan_tiles_kernel, dim3 (1), dim3 (1024), 0, d->stream, d_tc, d_to, n_tiles, d_total);
  uint64_t total = 0;
  HIPCHK (d, hipMemcpyAsync (&total, d_total, 8, hipMemcpyDeviceToHost, d->stream));
  HIPCHK (d, hipStreamSynchronize (d->stream));
  *n_records = total;
  int rc = 0;
  if (out && total)
    {
      if (total > out_capacity)
        rc = fail (d, "fetch_records: %llu records do not fit the caller's %llu", (unsigned long long) total,
                   (unsigned long long) out_capacity);
      else
        {
          PileRec *d_out = nullptr;
          rc = dev_alloc (d, &d_out, (size_t) total);
          if (!rc)
            {
              hipLaunchKernelGGL (pile_emit_kernel, dim3 ((unsigned) n_tiles), dim3 (PR_BLOCK), 0, d->stream, pile_of (d), first, count, d_to,
                                  d_out, total);
              if (hipStreamSynchronize (d->stream) != hipSuccess
                  || hipMemcpy (out, d_out, total * sizeof (PileRec), hipMemcpyDeviceToHost) != hipSuccess)
                rc = fail (d, "fetch_records: copy failed");
              hipFree (d_out);
            }
        }
    }
  hipFree (d_tc);
  hipFree (d_to);
  hipFree (d_total);
  return rc;
}

// ------------------------------------------------------------------------------------------------------------
extern "C" int pemap_dev_synth_genome (pemap_dev * d, uint64_t seed, uint64_t genome_size, int n_contigs, double repeat_frac,
                                       void **d_genome_out, uint32_t * contig_len)
{
  HIPCHK (d, hipSetDevice (d->device));
  if (n_contigs < 1 || genome_size < (uint64_t) n_contigs * 1000)
    return fail (d, "synth_genome: bad sizes");
  // hg38-like relative contig lengths (chr1..22, X, Y, M in Mbp) when 25 contigs are asked for, equal split otherwise
  static const double hg[25] = { 248.9, 242.2, 198.3, 190.2, 181.5, 170.8, 159.3, 145.1, 138.4, 133.8, 135.1, 133.3, 114.4, 107.0,
    102.0, 90.3, 83.3, 80.4, 58.6, 64.4, 46.7, 50.8, 156.0, 57.2, 0.0166
  };
  std::vector < uint64_t > real (n_contigs + 1);
  double tot = 0;
  for (int c = 0; c < n_contigs; c++)
    tot += (n_contigs == 25) ? hg[c] : 1.0;
  real[0] = 0;
  uint64_t used = 0;
  for (int c = 0; c < n_contigs; c++)
    {
      uint64_t ln = (uint64_t) ((double) genome_size * ((n_contigs == 25) ? hg[c] : 1.0) / tot);
      if (ln < 64)
        ln = 64;
      if (c == n_contigs - 1)
        ln = genome_size - used;
      contig_len[c] = (uint32_t) ln;
      used += ln;
      real[c + 1] = used;
    }
  if (used != genome_size)
    return fail (d, "synth_genome: contig split failed");
  uint8_t *g = nullptr;
  TRY (dev_alloc (d, &g, genome_size + 512));
  uint64_t *d_real = nullptr;
  TRY (dev_alloc (d, &d_real, (size_t) n_contigs + 1));
  HIPCHK (d, hipMemcpy (d_real, real.data (), (n_contigs + 1) * 8, hipMemcpyHostToDevice));
  unsigned thr = (unsigned) (repeat_frac * 16777216.0);
  hipLaunchKernelGGL (sy_genome_kernel, dim3 ((unsigned) ((genome_size + 255) / 256)), dim3 (256), 0, d->stream, seed, g, genome_size, d_real,
                      n_contigs, thr);
  HIPCHK (d, hipStreamSynchronize (d->stream));
  hipFree (d_real);
  *d_genome_out = g;
  return 0;
}

extern "C" int pemap_dev_free (pemap_dev * d, void *d_ptr)
{
  HIPCHK (d, hipSetDevice (d->device));
  HIPCHK (d, hipFree (d_ptr));
  return 0;
}

static int synth_reads (pemap_dev * d, uint64_t seed, int n, int read_len, int paired, double sub_rate, double indel_rate, double one_indel_frac,
                        uint64_t first_read);

extern "C" int pemap_dev_synth_reads (pemap_dev * d, uint64_t seed, int n, int read_len, int paired, double sub_rate,
                                      double indel_rate, uint64_t first_read)
{
  return synth_reads (d, seed, n, read_len, paired, sub_rate, indel_rate, 0.0, first_read);
}

extern "C" int pemap_dev_synth_reads_indel (pemap_dev * d, uint64_t seed, int n, int read_len, int paired, double sub_rate,
                                            double indel_read_frac, uint64_t first_read)
{
  return synth_reads (d, seed, n, read_len, paired, sub_rate, 0.0, indel_read_frac, first_read);
}

static int synth_reads (pemap_dev * d, uint64_t seed, int n, int read_len, int paired, double sub_rate, double indel_rate, double one_indel_frac,
                        uint64_t first_read)
{
  HIPCHK (d, hipSetDevice (d->device));
  if (!d->index_ready)
    return fail (d, "synth_reads: needs the resident genome");
  if (read_len < PEMAP_MIN_READ || read_len > PEMAP_MAX_READ || n <= 0)
    return fail (d, "synth_reads: bad n/read_len");
  if (d->gsize < 2000)
    return fail (d, "synth_reads: genome too small");
  int stride = (read_len + 15) & ~15;
  TRY (ring_leave (d));
  TRY (pemap_dev_sync (d));
  d->paired = paired ? 1 : 0;
  TRY (ensure_reads (d, n, stride, paired));
  int ends = paired ? 2 * n : n;
  hipLaunchKernelGGL (sy_reads_kernel, dim3 ((ends + 255) / 256), dim3 (256), 0, d->stream, seed, d->d_genome, d->gsize, n, read_len, paired,
                      (unsigned) (sub_rate * 16777216.0), (unsigned) (indel_rate * 16777216.0), first_read, d->d_reads1, d->d_len1,
                      d->d_reads2, d->d_len2, stride, (unsigned) (one_indel_frac * 16777216.0));
  HIPCHK (d, hipStreamSynchronize (d->stream));
  d->h_len1.assign (n, read_len);
  if (paired)
    d->h_len2.assign (n, read_len);
  else
    d->h_len2.clear ();
  d->n_staged = n;
  d->staged_paired = paired ? 1 : 0;
  d->max_len_staged = d->min_len_staged = read_len;
  return 0;
}

extern "C" int pemap_dev_staged_reads (pemap_dev * d, char *reads1, int *len1, char *reads2, int *len2, int stride)
{
  HIPCHK (d, hipSetDevice (d->device));
  if (d->n_staged <= 0)
    return fail (d, "staged_reads: nothing staged");
  if (stride != d->stride)
    return fail (d, "staged_reads: the staged row stride is %d", d->stride);
  HIPCHK (d, hipStreamSynchronize (d->stream));
  HIPCHK (d, hipMemcpy (reads1, d->d_reads1, (size_t) d->n_staged * stride, hipMemcpyDeviceToHost));
  HIPCHK (d, hipMemcpy (len1, d->d_len1, (size_t) d->n_staged * 4, hipMemcpyDeviceToHost));
  if (d->staged_paired && reads2)
    {
      HIPCHK (d, hipMemcpy (reads2, d->d_reads2, (size_t) d->n_staged * stride, hipMemcpyDeviceToHost));
      HIPCHK (d, hipMemcpy (len2, d->d_len2, (size_t) d->n_staged * 4, hipMemcpyDeviceToHost));
    }
  return 0;
}

extern "C" int pemap_dev_staged_info (pemap_dev * d, int *n, int *stride, int *paired)
{
  *n = d->n_staged;
  *stride = d->stride;
  *paired = d->staged_paired;
  return 0;
}
