/*
  kernels/interleave.hip.h -- rank-array finalize and mergeBWT.
  Part of bwtm_kernels.hip.h (included there, inside namespace bwtm); gfx950 only.
*/
#pragma once

//------------------------------------------------------------------------------
// K2: rank-array finalize.  A chunk is 64 output records = 8192 bits = 128 words; one wave
// per chunk counts the set bits.  (An exclusive scan of the counts follows.)

constexpr int CHUNK_WORDS = 128;

__global__ void __launch_bounds__(BLOCK_THREADS) k_chunk_popc(const u64* bits, u64 nchunks, u64* cnt)
{
  u64 chunk = ((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 6;
  if(chunk >= nchunks) { return; }
  const u64* w = bits + chunk * CHUNK_WORDS + 2 * lane_id();
  u64 v = (u64)__builtin_popcountll(w[0]) + (u64)__builtin_popcountll(w[1]);
  v = wave_sum(v);
  if(lane_id() == 0) { cnt[chunk] = v; }
}

// dst |= src over the words of two interleaving bitvectors (shards of one rank array searched into separate buffers of the same
// device; the bits of different shards are disjoint).
__global__ void __launch_bounds__(BLOCK_THREADS) k_bits_or(u64* dst, const u64* src, u64 nwords)
{
  u64 k = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(k < nwords) { u64 v = src[k]; if(v != 0) { dst[k] |= v; } }
}

// RA[i] for every B position (tests / facade): one wave per chunk, one lane per record.
__global__ void __launch_bounds__(BLOCK_THREADS) k_ra_extract(const u64* bits, const u64* chunk_base, u64 nchunks, u64 nb, u64* ra)
{
  u64 chunk = ((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 6;
  if(chunk >= nchunks) { return; }
  u64 q = chunk * 64 + lane_id();
  u64 m0 = bits[2 * q], m1 = bits[2 * q + 1];
  u64 mine = (u64)__builtin_popcountll(m0) + (u64)__builtin_popcountll(m1);
  u64 incl = wave_incl_sum(mine);
  u64 i = chunk_base[chunk] + incl - mine;
  u64 base = q << REC_SHIFT;
  while(m0) { u32 t = (u32)__builtin_ctzll(m0); m0 &= m0 - 1; if(i < nb) { ra[i] = base + t - i; } i++; }
  while(m1) { u32 t = (u32)__builtin_ctzll(m1); m1 &= m1 - 1; if(i < nb) { ra[i] = base + 64 + t - i; } i++; }
}

// The rank array as maximal (rank, count) runs, the form RankArray hands to mergeBWT (support.h:576-638,
// bwt.cpp:194-213): equal ranks are consecutive 1 bits of the interleaving bitvector, so a run is a maximal
// block of ones; its rank is the number of zeros before it and its count the number of ones up to the next run.
// Pass 1 counts the run starts per chunk; after a scan pass 2 writes (rank, B offset) per run; counts = differences.
__device__ inline u64 run_starts(const u64* bits, u64 w, u64 word)
{
  const u64 carry = (w == 0 ? 0ull : bits[w - 1] >> 63);
  return word & ~((word << 1) | carry);
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_ra_run_count(const u64* bits, u64 nchunks, u64* cnt)
{
  u64 chunk = ((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 6;
  if(chunk >= nchunks) { return; }
  const u64 w = chunk * CHUNK_WORDS + 2 * lane_id();
  u64 v = (u64)__builtin_popcountll(run_starts(bits, w, bits[w])) + (u64)__builtin_popcountll(run_starts(bits, w + 1, bits[w + 1]));
  v = wave_sum(v);
  if(lane_id() == 0) { cnt[chunk] = v; }
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_ra_run_write(const u64* bits, const u64* chunk_base, const u64* run_base, u64 nchunks,
  u64* ranks, u64* boff, u64 capacity)
{
  u64 chunk = ((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 6;
  if(chunk >= nchunks) { return; }
  const u64 w = chunk * CHUNK_WORDS + 2 * lane_id();
  const u64 m0 = bits[w], m1 = bits[w + 1];
  u64 s0 = run_starts(bits, w, m0), s1 = run_starts(bits, w + 1, m1);
  const u64 ones = (u64)__builtin_popcountll(m0) + (u64)__builtin_popcountll(m1);
  const u64 starts = (u64)__builtin_popcountll(s0) + (u64)__builtin_popcountll(s1);
  const u64 ones_incl = wave_incl_sum(ones), starts_incl = wave_incl_sum(starts);
  u64 b = chunk_base[chunk] + ones_incl - ones;            // ones before word w
  u64 k = run_base[chunk] + starts_incl - starts;          // runs before word w
  while(s0) { u32 t = (u32)__builtin_ctzll(s0); s0 &= s0 - 1; u64 before = b + (u64)__builtin_popcountll(m0 & ((1ull << t) - 1)); if(k < capacity) { ranks[k] = (w << 6) + t - before; boff[k] = before; } k++; }
  b += (u64)__builtin_popcountll(m0);
  while(s1) { u32 t = (u32)__builtin_ctzll(s1); s1 &= s1 - 1; u64 before = b + (u64)__builtin_popcountll(m1 & ((1ull << t) - 1)); if(k < capacity) { ranks[k] = ((w + 1) << 6) + t - before; boff[k] = before; } k++; }
}

// counts[k] = boff[k + 1] - boff[k] for k < count (boff has nboff entries; the run after the last one starts at nb).
__global__ void __launch_bounds__(BLOCK_THREADS) k_ra_run_diff(const u64* boff, u64 nboff, u64 count, u64 nb, u64* counts)
{
  u64 k = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(k >= count) { return; }
  counts[k] = (k + 1 < nboff ? boff[k + 1] : nb) - boff[k];
}

//------------------------------------------------------------------------------
// K3: interleave (mergeBWT, bwt.cpp:215-282).  Output position p takes the next symbol of B
// when bit p of the interleaving bitvector is set and the next symbol of A otherwise, so
// an output record needs b_off = rank1(bits, 128 q) and a_off = 128 q - b_off, and its
// header is rank_A(a_off) + rank_B(b_off).  One lane per output record.

// Number of set bits before output record q, given the chunk bases.
__device__ inline u64 bits_before_record(const u64* bits, const u64* chunk_base, u64 q)
{
  u64 chunk = q >> 6;
  u64 b = chunk_base[chunk];
  for(u64 w = chunk * CHUNK_WORDS; w < 2 * q; w++) { b += (u64)__builtin_popcountll(bits[w]); }
  return b;
}

// Super table of the output: absolute counts at the start of every super.
__global__ void __launch_bounds__(BLOCK_THREADS) k_interleave_sup(IndexView A, IndexView B, const u64* bits, const u64* chunk_base,
  u64 n_out, u64* sup, u64 nsup)
{
  u64 s = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(s >= nsup) { return; }
  u64 q = s << SUPER_REC_SHIFT;
  u64 b_off = bits_before_record(bits, chunk_base, q);
  u64 a_off = (q << REC_SHIFT) - b_off;
  if(a_off > A.n) { a_off = A.n; }
  if(b_off > B.n) { b_off = B.n; }
  u64 ra[6], rb[6];
  index_ranks(A, a_off, ra); index_ranks(B, b_off, rb);
  for(int c = 0; c < SUP_STRIDE; c++) { sup[s * SUP_STRIDE + c] = (c >= 1 && c < 6 ? ra[c] + rb[c] : 0); }
  (void)n_out;
}

// The launch covers the chunks [chunk_first, chunk_end) and writes the records [q_lo, q_hi) among them (an output-range
// slice also wants the one record before its range: the encoder looks at the symbol there).  recs_out is indexed by the
// GLOBAL record number: a slice passes its buffer's address minus the offset of its first record.
__global__ void __launch_bounds__(BLOCK_THREADS) k_interleave(IndexView A, IndexView B, const u64* bits, const u64* chunk_base,
  u64 chunk_first, u64 chunk_end, u64 q_lo, u64 q_hi, const u64* sup_out, uint4* recs_out)
{
  u64 chunk = chunk_first + (((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 6);
  if(chunk >= chunk_end) { return; }
  u64 q = chunk * 64 + lane_id();
  u64 m0 = bits[2 * q], m1 = bits[2 * q + 1];
  u64 mine = (u64)__builtin_popcountll(m0) + (u64)__builtin_popcountll(m1);
  u64 incl = wave_incl_sum(mine);
  if(q < q_lo || q >= q_hi) { return; }
  u64 b_off = chunk_base[chunk] + incl - mine;
  u64 a_off = (q << REC_SHIFT) - b_off;

  // Header: counts of symbols 1..5 before output position 128 q.
  u64 ra[6], rb[6];
  index_ranks(A, (a_off > A.n ? A.n : a_off), ra);
  index_ranks(B, (b_off > B.n ? B.n : b_off), rb);
  const u64* s = sup_out + (q >> SUPER_REC_SHIFT) * SUP_STRIDE;
  u32 rel[6]; u32 h[4];
  for(int c = 1; c < 6; c++) { rel[c] = (u32)(ra[c] + rb[c] - s[c]); }
  pack_header(rel, h);

  // Planes: two halves of 64 positions.
  u64 a0, a1, a2, b0, b1, b2, lo0, lo1, lo2, hi0, hi1, hi2;
  load_window(A, a_off, a0, a1, a2); load_window(B, b_off, b0, b1, b2);
  deposit64(m0, a0, a1, a2, b0, b1, b2, lo0, lo1, lo2);
  u64 nb0 = (u64)__builtin_popcountll(m0);
  load_window(A, a_off + 64 - nb0, a0, a1, a2); load_window(B, b_off + nb0, b0, b1, b2);
  deposit64(m1, a0, a1, a2, b0, b1, b2, hi0, hi1, hi2);

  uint4* dst = recs_out + 4 * q;
  dst[0] = make_uint4((u32)lo0, (u32)lo1, (u32)lo2, h[0]);
  dst[1] = make_uint4((u32)(lo0 >> 32), (u32)(lo1 >> 32), (u32)(lo2 >> 32), h[1]);
  dst[2] = make_uint4((u32)hi0, (u32)hi1, (u32)hi2, h[2]);
  dst[3] = make_uint4((u32)(hi0 >> 32), (u32)(hi1 >> 32), (u32)(hi2 >> 32), h[3]);
}
