/*
  bwtm_experimental.h -- entry points that are NOT part of the product library.

  libbwtm.so (the C ABI of include/bwtm.h) does not export them.  They exist in bwt-merge_amd/libbwtm_experimental.so only, a second
  build of the same sources with -DBWTM_EXPERIMENTAL, which the test-suite loads for the tests of these features:

    * the sliced frontier search (bwtm_fslice_*): the dense multi-GPU form of buildRA (fmi.cpp:272-334).  Exact on 1 - 8 parts against
      the oracle; its exchange needs peer-mapped buffers and a gather folded into the step kernel before it is a product path, and no
      machine with more than one GPU was available to measure it;
    * the merge over PARTITIONED records (bwtm_x_index_upload_window / bwtm_x_index_window, bwtm_x_ra_create_range, bwtm_fslice_set_cuts,
      bwtm_fslice_nodes_*, bwtm_fslice_gather_cut, bwtm_fslice_input_buffers): nothing replicated -- every GPU holds one window of each
      input and of the bitvector, nodes and elements travel to the GPU that owns their position, and the product's range entry points
      finish the merge from the windows.  Exact on 1 - 16 parts against the oracle, and against the product merge at full size
      (DESIGN.md section 6.3); drivers: bwt-merge_amd/experimental.py (contexts of one GPU), experimental_dist.py (one process per GPU),
      csrc/host/multi_gpu.h (bwt_merge_experimental -P).  Never run on more than one GPU;
    * the two-plane search view (bwtm_tune("search_view", 1 | 2)): a denser copy of the rank structure for the frontier search.  Exact;
      14 % fewer HBM reads and no time saved, hence not in the product;
    * bwtm_x_device_scan: a test hook of the library's device scan.
*/
#ifndef BWTM_EXPERIMENTAL_H
#define BWTM_EXPERIMENTAL_H

#include "bwtm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* --- sliced frontier search: the dense multi-GPU form of buildRA -----------------------
   bwtm_search() on a block of b's sequences per GPU thins the sorted frontier: every GPU still streams most cache lines of both
   rank structures.  Here every GPU advances a CONTIGUOUS SLICE of the sorted frontier instead -- a slice touches a contiguous
   range of both structures -- and assembles its next slice from its peers' outputs.  Per LF step, on every GPU:
       bwtm_fslice_gather(fs, views of all GPUs, first, last);  <barrier>  bwtm_fslice_advance(fs);  bwtm_fslice_export(fs, &view);  <barrier>
   with [first, last) = this GPU's share of the sum N of all views' totals; the search ends when N = 0; bwtm_fslice_finish() then
   completes the GPU's bits in its rank array, and the rank arrays are combined as for bwtm_search() (all-reduce / bwtm_ra_or_from).
   A view holds raw device pointers into the exporting GPU's memory (hipMalloc): contexts of one device can always read them,
   other devices need peer access. */
#define BWTM_X_MAX_PARTS 16
typedef struct bwtm_fslice bwtm_fslice;
typedef struct
{
  const void* lo; const void* hi;            /* coordinates of the GPU's output elements (hi: NULL below 2^32 positions) */
  const void* prefix; const void* phys;      /* its segment tables: 5 * blocks + 1 entries each */
  uint64_t blocks;
  uint64_t totals[5];                        /* elements per class (the symbols 1..5; the seed holds its sequences in class 0) */
  uint64_t below[5][BWTM_X_MAX_PARTS + 1];   /* after bwtm_fslice_set_cuts(): elements of class c whose B coordinate lies below cut k */
  const void* dense_lo; const void* dense_hi; /* after bwtm_fslice_set_cuts(): the outputs in logical order (class, block): the send buffer of the exchange */
  uint64_t class_first[6];                   /* where every class begins in it */
} bwtm_fslice_view;
/* `capacity` = the largest slice this GPU will be given (ceil(sequences / parts) + 1 is enough: the frontier only shrinks). */
int bwtm_fslice_create(const bwtm_index* a, const bwtm_index* b, bwtm_ra* ra, uint64_t capacity, int parts, bwtm_fslice** out);
void bwtm_fslice_free(bwtm_fslice* fs);
/* The outputs of "step -1": the chains of the sequences [seq_first, seq_first + count) of b at their roots (fmi.cpp:286). */
int bwtm_fslice_seed(bwtm_fslice* fs, uint64_t seq_first, uint64_t count);
int bwtm_fslice_export(bwtm_fslice* fs, bwtm_fslice_view* view);                 /* synchronizes */
int bwtm_fslice_gather(bwtm_fslice* fs, const bwtm_fslice_view* views, int parts, uint64_t first, uint64_t last);   /* synchronizes */
int bwtm_fslice_advance(bwtm_fslice* fs);
int bwtm_fslice_finish(bwtm_fslice* fs);

/* PARTITIONED RECORDS (the search half, a functional prototype).  The sliced search above cuts the frontier into
   equal shares, so a GPU's slice wanders over the whole of A and B and every GPU must hold all records.  With FIXED cuts every GPU keeps
   one window of each index and the frontier's elements travel to the GPU that owns their positions:
     - a cut is a pair (I_k, R_k) = (suffixes of a below w_k, suffixes of b below w_k) for some string w_k (k-mers: ranks by backward
       search, e.g. bwtm_rank_batch); I_0 = R_0 = 0; both ranks are monotone along the merged order, so an element with R_g <= r < R_g+1
       has I_g <= i <= I_g+1;
     - GPU g holds bwtm_x_index_window(a, I_g, I_g+1) and bwtm_x_index_window(b, R_g, R_g+1): the records of those positions only,
       addressed by absolute record numbers (1 / parts of the records per GPU; the super tables are whole, they are small);
     - bwtm_fslice_set_cuts(fs, R_0 .. R_parts) makes every export count, per class, its output elements below every cut
       (bwtm_fslice_view.below), and bwtm_fslice_gather_cut(fs, views, parts, g) pulls exactly GPU g's elements from all peers;
     - the roots of the sequences (b coordinate = sequence number) are seeded on the GPU that owns them: bwtm_fslice_seed with the
       sequences [R_g, R_g+1) that exist.
   The loop is the sliced search's with bwtm_fslice_gather_cut in place of bwtm_fslice_gather.  Every GPU sets the bits of its own
   output range [I_g + R_g, I_g+1 + R_g+1) only: no bitvector exchange is needed afterwards (the prototype keeps whole-length bitvectors).
   A window handle is only valid as an argument of bwtm_fslice_create, bwtm_ra_create and -- with a rank array finalized for an output range that
   the windows cover, margins of two encoder segments included -- bwtm_interleave_range; every other entry point refuses it. */
int bwtm_x_index_window(const bwtm_index* whole, uint64_t pos_first, uint64_t pos_last, bwtm_index** out);
/* The same window transcoded from its OWN share of the native bytes (the sharded transcode of section 6.3): `data` = the whole 64-byte blocks
   [b0, b1) of the index's native stream, first_position = the position block b0 begins at, counts_before[c] = occurrences of symbol c before
   it (both are what the samples of a native file hold per block: bwt.cpp:489-511), bases / sequences / C = the header of the WHOLE index.
   The handle serves the records that lie wholly inside the bytes: positions [first_position rounded up to 128, end rounded down to 128) -- or
   to the end of the index when the bytes hold its last block.  No GPU ever holds the whole index this way. */
int bwtm_x_index_upload_window(const uint8_t* data, uint64_t nbytes, uint64_t first_position, const uint64_t counts_before[6],
                               uint64_t bases, uint64_t sequences, const uint64_t C[7], bwtm_index** out);
/* The rank array of one part: the bits of the output positions [pos_first, pos_last) plus one 65 536-position tile on either side (the
   boundary segments and the halo chunk of an output range), addressed by absolute positions like a whole one.  It serves bwtm_fslice_*,
   bwtm_ra_range_counts / _finalize_range / bwtm_interleave_range for ranges inside it, and bwtm_x_ra_or_range; every entry point that walks
   the whole bitvector refuses it.  bwtm_x_ra_or_range: dst |= src on the words of [pos_first, pos_last) (both must hold them; what crosses a
   boundary between two GPUs: at most one segment). */
int bwtm_x_ra_create_range(const bwtm_index* a, const bwtm_index* b, uint64_t pos_first, uint64_t pos_last, bwtm_ra** out);
int bwtm_x_ra_or_range(bwtm_ra* dst, const bwtm_ra* src, uint64_t pos_first, uint64_t pos_last);
/* The same boundary exchange between processes: the words of [pos_first, pos_last) to / from a caller's device buffer (words the handle does
   not hold read as zero and are not written). */
int bwtm_x_ra_read_words(const bwtm_ra* ra, uint64_t pos_first, uint64_t pos_last, void* device_out);
int bwtm_x_ra_or_words(bwtm_ra* ra, uint64_t pos_first, uint64_t pos_last, const void* device_in);
uint64_t bwtm_x_ra_bytes(const bwtm_ra* ra);                        /* bytes of bitvector the handle holds */
uint64_t bwtm_x_index_record_bytes(const bwtm_index* index);       /* bytes of records the handle holds (a window: its share) */
int bwtm_fslice_set_cuts(bwtm_fslice* fs, const uint64_t* r_cuts, int parts);      /* r_cuts[0 .. parts]: R_0 = 0 <= R_1 <= ... ; R_parts is ignored (= everything) */
int bwtm_fslice_gather_cut(bwtm_fslice* fs, const bwtm_fslice_view* views, int parts, int part);   /* synchronizes */
/* The same step between processes (one per GPU): the caller moves the elements itself -- an all-to-all-v over the views' dense send buffers,
   bwt-merge_amd/experimental_dist.py -- straight into the slice's input buffers, and says how many arrived. */
int bwtm_fslice_input_buffers(bwtm_fslice* fs, void** lo, void** hi, uint64_t* capacity);      /* hi: NULL below 2^32 positions */
int bwtm_fslice_set_input(bwtm_fslice* fs, uint64_t count);

/* The first levels on trie NODES over partitioned records (the node phase of bwtm_search, fmi.cpp:286-323, with the nodes routed like the
   elements): a node (sp, count, r) lives on the GPU that owns sp; its children are exported class by class with their counts below every
   cut, and every GPU assembles its next level from all peers' children.  With cuts at k-mer boundaries no node crosses a cut (a node that
   does is reported as an error).  Per level, on every GPU:
       bwtm_fslice_nodes_step(fs, &nview);   <barrier>   bwtm_fslice_nodes_gather(fs, nviews of all GPUs, parts, g);   <barrier>
   while the level (the sum of the views' totals) is small; then bwtm_fslice_nodes_expand(fs) turns the GPU's nodes into its elements --
   the outputs of a step, picked up by bwtm_fslice_export / bwtm_fslice_gather_cut -- and the element steps go on from there.  The bits of
   the node levels are set by bwtm_fslice_nodes_step itself. */
typedef struct
{
  const void* sp; const void* r; const void* count;   /* the GPU's children of this level: sorted by sp, class after class */
  uint64_t class_first[6];                            /* where every class begins */
  uint64_t below[5][BWTM_X_MAX_PARTS + 1];            /* children of class c with sp below cut k */
} bwtm_fslice_nodes_view;
int bwtm_fslice_nodes_begin(bwtm_fslice* fs, uint64_t seq_first, uint64_t count, uint64_t node_capacity);   /* the root "$" of these sequences (count = 0: no node); after bwtm_fslice_set_cuts */
int bwtm_fslice_nodes_step(bwtm_fslice* fs, bwtm_fslice_nodes_view* view);                                   /* synchronizes */
int bwtm_fslice_nodes_gather(bwtm_fslice* fs, const bwtm_fslice_nodes_view* views, int parts, int part);    /* synchronizes */
/* Between processes: the caller moves the children itself (three all-to-alls per class: experimental_dist.py) into the node buffers. */
int bwtm_fslice_nodes_input_buffers(bwtm_fslice* fs, void** sp, void** r, void** count, uint64_t* capacity);
int bwtm_fslice_nodes_set_input(bwtm_fslice* fs, uint64_t nodes);
int bwtm_fslice_nodes_expand(bwtm_fslice* fs);                                                               /* then bwtm_fslice_export */

/* Test hook of the library's device scan (every table of the path -- segment prefixes, node offsets, block and sample tables -- goes
   through it): exclusive scan of `narrays` arrays of n items each, laid end to end in host memory; op 0 = sum, 1 = max; runs in the
   calling thread's current context. */
int bwtm_x_device_scan(const uint64_t* in, uint64_t* out, uint64_t n, uint64_t narrays, int op);


#ifdef __cplusplus
}
#endif

#endif /* BWTM_EXPERIMENTAL_H */
