/*
  bwtm_kernels.hip.h -- hand-written gfx950 (CDNA4, wave64) kernels of the rank-array /
  interleave path.  Included by bwtm_api.hip only.

  Kernel map (reference code each one replaces):
    k_block_len        BWT::build scan of the run stream            bwt.cpp:487-502
    k_build_sup/recs   native blocks -> device records ("transcode at upload", BWT::load)
    k_block_cum        samples[c] at the block starts               bwt.cpp:489-511
    k_sym_*            plain symbols -> device records (input tooling)
    k_ingest_*         reads -> BWT symbols of a leaf (suffix radix sort; SURVEY 8(f1), no counterpart in the reference)
    k_frontier_*       buildRA + BWT::inverse_select + BWT::rank, level-synchronous form (product)
                                                                    fmi.cpp:272-334, bwt.cpp:318-341, 445-464
    k_bound_suffix_min, k_tile_build_frontier
                       mergeRA / RLArray / RankArray: the sorted emits of every step -> bitvector tiles
                                                                    fmi.cpp:139-257, support.h:396-638
    k_lf_walk_binned, k_lf_walk_quad, k_part_*, k_tile_build
                       the same two stages in per-chain form (small shards, long sequences, > 40-bit coordinates)
    k_cut_search_seg, k_pull_tables, k_node_cut_search, k_gather_nodes, k_frontier_step<.., PULL>
                       the same search with the records PARTITIONED over several GPUs (one part each): the thread fan-out of
                       fmi.cpp:351-358 as position ranges instead of sequence blocks (kernels/partition.hip.h, api/pmerge.hip.h)
    kernels/diagnostics.hip.h (only with -DBWTM_DIAGNOSTICS): timing-only and A/B variants, not in the product
    kernels/search_view.hip.h, k_frontier_step<.., VIEW>, k_frontier_gather (only with -DBWTM_EXPERIMENTAL): the two-plane search
                       view and the sliced frontier search -- exact, tested, measured, not in the product (include/bwtm_experimental.h)
    k_chunk_popc       RA finalize (prefix counts of the interleaving bitvector)
    k_interleave_*     mergeBWT                                     bwt.cpp:215-282
    k_enc_*            RunBuffer + Run::write, block starts of BWT::build
                                                                    utils.h:121-142, support.h:256-282, bwt.cpp:496
    k_fold_*           byte-offset carry of Run::write across segments (array.size() % 64, support.h:267)
    k_rank_batch, k_inverse_select_batch, k_extract, k_find_batch
                       BWT::rank / inverse_select / extract, FMI::find  bwt.cpp:318-464, fmi.h:195-221
*/
#pragma once

#include <hip/hip_runtime.h>
#include "bwtm_device.h"
#include "bwtm_bitmerge.h"
#ifdef BWTM_EXPERIMENTAL
#include "bwtm_view.h"
#endif

namespace bwtm
{

constexpr int WAVE = 64;
constexpr int BLOCK_THREADS = 256;

// Read-only view of a device index for kernels.
struct IndexView
{
  const uint4* recs;     // 4 x uint4 per record
  const u64*   sup;      // SUP_STRIDE u64 per super
  u64 n;                 // positions
  u64 m;                 // sequences
  u64 nrecs;
  u64 C[8];              // C[c] = number of symbols smaller than c
#ifdef BWTM_EXPERIMENTAL
  const uint4* view;     // the search view (4 x uint4 per 160 positions; null unless built: bwtm_view.h)
  const u64* vsup;
  u64 nview;
#endif
};

// Streaming stores: data that is written once and not read again by the kernel that writes it (the next frontier's coordinates, emits, records
// of a transcode / interleave, encoded bytes) goes past the L2 with the non-temporal hint, which leaves the cache to the lines that ARE re-used
// (the records neighbouring elements share).  Round 5: -2.8 % on k_frontier_step from its two stores alone (profiles/r05_nt_stores_ab.txt).
typedef unsigned int bwtm_v4u __attribute__((ext_vector_type(4)));
__device__ inline void nt_store(uint4* p, const uint4& v) { __builtin_nontemporal_store(bwtm_v4u{v.x, v.y, v.z, v.w}, (bwtm_v4u*)p); }
__device__ inline void nt_store(uint2* p, const uint2& v) { __builtin_nontemporal_store((unsigned long long)v.x | ((unsigned long long)v.y << 32), (unsigned long long*)p); }
__device__ inline void nt_store(unsigned short* p, unsigned short v) { __builtin_nontemporal_store(v, p); }
__device__ inline void nt_store(u64* p, u64 v) { __builtin_nontemporal_store(v, p); }
__device__ inline void nt_store(u32* p, u32 v) { __builtin_nontemporal_store(v, p); }

#include "kernels/common.hip.h"
#include "kernels/transcode.hip.h"
#include "kernels/queries.hip.h"
#include "kernels/search_walk.hip.h"
#ifdef BWTM_DIAGNOSTICS
#include "kernels/diagnostics.hip.h"
#endif
#ifdef BWTM_EXPERIMENTAL
#include "kernels/search_view.hip.h"
#endif
#include "kernels/search_frontier.hip.h"
#include "kernels/partition.hip.h"
#ifdef BWTM_EXPERIMENTAL
#include "kernels/search_partition.hip.h"
#endif
#include "kernels/search_range.hip.h"
#include "kernels/interleave.hip.h"
#include "kernels/encoder.hip.h"
#include "kernels/ingest.hip.h"

} // namespace bwtm
