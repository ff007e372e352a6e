/*
  bwtm_experimental.h -- entry points that are NOT part of the product library.

  libbwtm.so (the C ABI of include/bwtm.h) does not export them.  They exist in bwt-merge_amd/libbwtm_experimental.so only, a second
  build of the same sources with -DBWTM_EXPERIMENTAL, which the test-suite loads for the tests of these features:

    * the sliced frontier search (bwtm_fslice_*): the dense multi-GPU form of buildRA (fmi.cpp:272-334).  Exact on 1 - 8 parts against
      the oracle; its exchange needs peer-mapped buffers and a gather folded into the step kernel before it is a product path, and no
      machine with more than one GPU was available to measure it;
    * the two-plane search view (bwtm_tune("search_view", 1 | 2)): a denser copy of the rank structure for the frontier search.  Exact;
      14 % fewer HBM reads and no time saved, hence not in the product.
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
typedef struct bwtm_fslice bwtm_fslice;
typedef struct
{
  const void* lo; const void* hi;            /* coordinates of the GPU's output elements (hi: NULL below 2^32 positions) */
  const void* prefix; const void* phys;      /* its segment tables: 5 * blocks + 1 entries each */
  uint64_t blocks;
  uint64_t totals[5];                        /* elements per class (the symbols 1..5; the seed holds its sequences in class 0) */
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


#ifdef __cplusplus
}
#endif

#endif /* BWTM_EXPERIMENTAL_H */
