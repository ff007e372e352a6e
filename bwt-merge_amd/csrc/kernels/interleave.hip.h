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

// Output-range finalize: p[k] += v for k < n; out[s] = rel[(s << 12) - c0] for the supers that start at a chunk in [c0, c1) (a super
// starts at record s << 18 = chunk s << 12), zero for the others.
__global__ void __launch_bounds__(BLOCK_THREADS) k_add_offset(u64* p, u64 n, u64 v)
{
  u64 k = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(k < n) { p[k] += v; }
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_super_local(const u64* rel, u64 c0, u64 c1, u64* out, u64 nsup)
{
  u64 s = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(s >= nsup) { return; }
  const u64 c = s << (SUPER_REC_SHIFT - 6);
  out[s] = (c >= c0 && c < c1 ? rel[c - c0] : 0);
}

// dst |= src over the words of two interleaving bitvectors (shards of one rank array searched into separate buffers of the same
// device; the bits of different shards are disjoint).
__global__ void __launch_bounds__(BLOCK_THREADS) k_bits_or(u64* dst, const u64* src, u64 nwords)
{
  u64 k = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(k < nwords) { u64 v = src[k]; if(v != 0) { dst[k] |= v; } }
}

// Cross-check of two searches: set bits of `part` and the words in which `part` has a bit that `whole` lacks.
__global__ void __launch_bounds__(BLOCK_THREADS) k_bits_subset(const u64* part, const u64* whole, u64 nwords, unsigned long long* out)
{
  u64 k = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  u64 p = (k < nwords ? part[k] : 0), w = (k < nwords ? whole[k] : 0);
  const u64 ones = wave_sum((u64)__builtin_popcountll(p));
  const u64 bad = __builtin_popcountll(__ballot((p & ~w) != 0));
  if(lane_id() == 0)
  {
    if(ones != 0) { atomicAdd(out, (unsigned long long)ones); }
    if(bad != 0) { atomicAdd(out + 1, (unsigned long long)bad); }
  }
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
// super_boff (output-range form, bwtm_ra_finalize_range): the set bits before every super are given -- the bitvector of a GPU that
// took part in a reduce-scatter is complete only inside its own range.
__global__ void __launch_bounds__(BLOCK_THREADS) k_interleave_sup(IndexView A, IndexView B, const u64* bits, const u64* chunk_base,
  u64 n_out, u64* sup, u64 nsup, const u64* super_boff)
{
  u64 s = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(s >= nsup) { return; }
  u64 q = s << SUPER_REC_SHIFT;
  u64 b_off = (super_boff ? super_boff[s] : bits_before_record(bits, chunk_base, q));
  u64 a_off = (q << REC_SHIFT) - b_off;
  if(a_off > A.n) { a_off = A.n; }
  if(b_off > B.n) { b_off = B.n; }
  u64 ra[6], rb[6];
  index_ranks(A, a_off, ra); index_ranks(B, b_off, rb);
  for(int c = 0; c < SUP_STRIDE; c++) { sup[s * SUP_STRIDE + c] = (c >= 1 && c < 6 ? ra[c] + rb[c] : 0); }
  (void)n_out;
}

// One workgroup of 256 threads per chunk of 64 output records; FOUR lanes per record, lane k of a quad owns the 32 positions of
// word k and writes the 16 bytes {plane0, plane1, plane2, header word k}: every wave store is 1 KiB of consecutive bytes.
// A lane needs the number of B symbols before its word (= ones of the bitvector before it: chunk base + a workgroup scan of the
// words' popcounts) -- that gives its cursors into A and B -- and, for the header, the number of each symbol before its record.
// The latter comes from the OUTPUT itself: the lanes count the symbols of their finished words and a second workgroup scan turns
// the counts into prefixes; only the counts at the chunk start are rank queries (two per 64 records instead of two per record).
// The launch covers the chunks [chunk_first, chunk_end) and writes the records [q_lo, q_hi) among them (an output-range
// slice also wants the one record before its range: the encoder looks at the symbol there).  recs_out is indexed by the
// GLOBAL record number: a slice passes its buffer's address minus the offset of its first record.
// The source symbols of a chunk are two contiguous ranges, at most 8192 positions of A and of B: their plane words are
// staged in LDS with coalesced 16-byte loads (257 words cover 8192 positions at any alignment), and every lane then takes
// the 32-bit windows at its two cursors from there.
constexpr u32 IL_WORDS = 2 * CHUNK_WORDS + 8;            // 264 staged words per plane

// Staging.  A chunk of 8192 output positions takes na of them from A and nb = 8192 - na from B, so the 16-byte words it needs from both
// sources together are 256 (+ up to 6 for alignment and the look-ahead word of the 32-bit windows): thread t takes word t of the
// COMBINED list (A's words first, then B's) and the first threads a second one.  (The first version staged 264 words of EACH source
// whatever the split -- 1.86 x the algorithmic read traffic in the PMC counters, and twice the LDS writes.)  The loads are
// unconditional, from clamped indexes, and issued before anything is written to LDS: written as a loop with a predicated load per
// source, the compiler put a full wait after each of them -- three memory round trips in a row at the start of every workgroup.
struct StagedLoad { uint4 v0, v1; u32 k0, k1; bool z0, z1, b0, b1, have1; };

__device__ inline void stage_pick(const IndexView& A, const IndexView& B, u64 wa0, u64 wb0, u32 na_words, u32 j, uint4& v, bool& z, bool& from_b, u32& k)
{
  from_b = (j >= na_words);
  k = (from_b ? j - na_words : j);
  const IndexView& x = (from_b ? B : A);
  const u64 last = 4 * x.nrecs;                                       // > 0: an index has at least one record
  const u64 w = (from_b ? wb0 : wa0) + k;
  z = (w >= last);
  v = (from_b ? B.recs : A.recs)[z ? last - 1 : w];
}

__device__ inline StagedLoad stage_issue(const IndexView& A, const IndexView& B, u64 wa0, u64 wb0, u32 na_words, u32 total_words)
{
  StagedLoad s;
  const u32 j0 = (threadIdx.x < total_words ? threadIdx.x : total_words - 1);
  const u32 j1 = (threadIdx.x + BLOCK_THREADS < total_words ? threadIdx.x + BLOCK_THREADS : total_words - 1);
  s.have1 = (threadIdx.x + BLOCK_THREADS < total_words);
  stage_pick(A, B, wa0, wb0, na_words, j0, s.v0, s.z0, s.b0, s.k0);
  stage_pick(A, B, wa0, wb0, na_words, j1, s.v1, s.z1, s.b1, s.k1);
  return s;
}

__device__ inline void stage_store(const StagedLoad& s, u32 total_words, u32 (*planes_a)[IL_WORDS], u32 (*planes_b)[IL_WORDS])
{
  if(threadIdx.x < total_words)
  {
    u32 (*p)[IL_WORDS] = (s.b0 ? planes_b : planes_a);
    p[0][s.k0] = (s.z0 ? 0u : s.v0.x); p[1][s.k0] = (s.z0 ? 0u : s.v0.y); p[2][s.k0] = (s.z0 ? 0u : s.v0.z);
  }
  if(s.have1)
  {
    u32 (*p)[IL_WORDS] = (s.b1 ? planes_b : planes_a);
    p[0][s.k1] = (s.z1 ? 0u : s.v1.x); p[1][s.k1] = (s.z1 ? 0u : s.v1.y); p[2][s.k1] = (s.z1 ? 0u : s.v1.z);
  }
}

__device__ inline void window32(const u32 (*planes)[IL_WORDS], u64 first_word, u64 pos, u32& p0, u32& p1, u32& p2)
{
  const u32 k = (u32)((pos >> 5) - first_word), sh = (u32)(pos & 31);
  p0 = (u32)((((u64)planes[0][k + 1] << 32) | planes[0][k]) >> sh);
  p1 = (u32)((((u64)planes[1][k + 1] << 32) | planes[1][k]) >> sh);
  p2 = (u32)((((u64)planes[2][k + 1] << 32) | planes[2][k]) >> sh);
}

// Symbol counts at the start of every chunk of the launch, relative to the output's super table: rel[(c - 1) * stride + (chunk - chunk_first)]
// = rank_A(a_off, c) + rank_B(b_off, c) - sup_out[super of the chunk][c].  One lane per chunk, two rank queries each -- 2.4 million of them at
// config 2, 0.1 ms.  Until round 4 lane 0 of every k_interleave workgroup ran these queries itself: a chain of dependent loads in front of
// the workgroup's second barrier, and ~300 instructions that the other 63 lanes of its wave executed masked off.
__global__ void __launch_bounds__(BLOCK_THREADS) k_interleave_base(IndexView A, IndexView B, const u64* chunk_base, u64 chunk_first, u64 chunk_end,
  const u64* sup_out, u32* rel, u64 stride)
{
  const u64 chunk = chunk_first + (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(chunk >= chunk_end) { return; }
  const u64 b_chunk = chunk_base[chunk];
  const u64 a_chunk = (chunk << (REC_SHIFT + 6)) - b_chunk;
  u64 ra[6], rb[6];
  index_ranks(A, (a_chunk > A.n ? A.n : a_chunk), ra);
  index_ranks(B, (b_chunk > B.n ? B.n : b_chunk), rb);
  const u64* sp = sup_out + ((chunk << 6) >> SUPER_REC_SHIFT) * SUP_STRIDE;
#pragma unroll
  for(u32 c = 1; c < 6; c++) { rel[(u64)(c - 1) * stride + (chunk - chunk_first)] = (u32)(ra[c] + rb[c] - sp[c]); }
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_interleave(IndexView A, IndexView B, const u64* bits, const u64* chunk_base,
  u64 chunk_first, u64 chunk_end, u64 q_lo, u64 q_hi, const u32* base_rel, u64 base_stride, uint4* recs_out)
{
  __shared__ u32 wave_tot[4][BLOCK_THREADS / WAVE];
  __shared__ u32 planes_a[3][IL_WORDS], planes_b[3][IL_WORDS];
  const u64 chunk = chunk_first + blockIdx.x;
  if(chunk >= chunk_end) { return; }
  const u32 t = threadIdx.x, lane = lane_id(), wave = t >> 6;
  const u64 w = chunk * (2 * CHUNK_WORDS) + t;                     // 32-bit word of the bitvector = 32-position word of the output
  const u32 m = ((const u32*)bits)[w];
  const u32 ones = (u32)__builtin_popcount(m);
  // counts at the chunk start, relative to the super table entry of the output (one lane; consumed after the barrier below)
  const u64 b_chunk = chunk_base[chunk];
  const u64 a_chunk = (chunk << (REC_SHIFT + 6)) - b_chunk;
  const u64 wa0 = a_chunk >> 5, wb0 = b_chunk >> 5;
  // positions the chunk takes from B = ones of its 8192 bits (chunk_base has an entry behind the last chunk); words that hold
  // them and the A positions, each with the look-ahead word a 32-bit window may touch
  const u32 nb_pos = (u32)(chunk_base[chunk + 1] - b_chunk), na_pos = (u32)(64 * REC_POS) - nb_pos;
  const u32 na_words = (u32)(((a_chunk + na_pos + 31) >> 5) - wa0) + 1, nb_words = (u32)(((b_chunk + nb_pos + 31) >> 5) - wb0) + 1;
  const u32 total_words = na_words + nb_words;                     // <= 256 + 6
  const StagedLoad st = stage_issue(A, B, wa0, wb0, na_words, total_words);
  __builtin_amdgcn_sched_barrier(0);                              // both loads leave before the first LDS write
  stage_store(st, total_words, planes_a, planes_b);
  // counts at the chunk start (k_interleave_base): workgroup-uniform addresses, i.e. scalar loads
  const u32* br = base_rel + blockIdx.x;
  const u32 br1 = br[0], br2 = br[base_stride], br3 = br[2 * base_stride], br4 = br[3 * base_stride], br5 = br[4 * base_stride];
  // cursors: B symbols before this word
  const u32 ones_incl = wave_incl_sum32(ones);
  if(lane == WAVE - 1) { wave_tot[0][wave] = ones_incl; }
  __syncthreads();
  u64 b_off = b_chunk + ones_incl - ones;
  for(u32 k = 0; k < wave; k++) { b_off += wave_tot[0][k]; }
  const u64 a_off = (w << 5) - b_off;

  u32 a0, a1, a2, b0, b1, b2;
  window32(planes_a, wa0, a_off, a0, a1, a2); window32(planes_b, wb0, b_off, b0, b1, b2);
  // the bit merge: displacement planes from one bit-sliced prefix count of the mask word, then pulls stated at the destination
  // (bwtm_bitmerge.h; the two parallel-suffix expands it replaces were ~230 of the kernel's 500 instructions per word)
  const MergeMasks mm = merge_masks(m);
#ifdef BWTM_SLACK_INTERLEAVE
  { u32 slack = m; valu_slack<BWTM_SLACK_INTERLEAVE>(slack); }
#endif
  const u32 o0 = bit_merge32(a0, b0, mm);
  const u32 o1 = bit_merge32(a1, b1, mm);
  const u32 o2 = bit_merge32(a2, b2, mm);

  // symbol counts of this word -> prefixes over the chunk (16-bit fields: a chunk has 8192 positions, and a symbol that fills
  // it completely would need 8192 + ... < 65536)
  const u32 c12 = (u32)__builtin_popcount(plane_match(o0, o1, o2, 1)) | ((u32)__builtin_popcount(plane_match(o0, o1, o2, 2)) << 16);
  const u32 c34 = (u32)__builtin_popcount(plane_match(o0, o1, o2, 3)) | ((u32)__builtin_popcount(plane_match(o0, o1, o2, 4)) << 16);
  const u32 c5 = (u32)__builtin_popcount(plane_match(o0, o1, o2, 5));
  const u32 incl12 = wave_incl_sum32(c12), incl34 = wave_incl_sum32(c34), incl5 = wave_incl_sum32(c5);
  if(lane == WAVE - 1) { wave_tot[1][wave] = incl12; wave_tot[2][wave] = incl34; wave_tot[3][wave] = incl5; }
  __syncthreads();
  u32 before12 = incl12 - c12, before34 = incl34 - c34, before5 = incl5 - c5;
  for(u32 k = 0; k < wave; k++) { before12 += wave_tot[1][k]; before34 += wave_tot[2][k]; before5 += wave_tot[3][k]; }
  // the header belongs to the record: take the prefixes of the quad's first lane (quad broadcast of lane 0: DPP quad_perm [0,0,0,0])
  const u32 rec12 = (u32)__builtin_amdgcn_update_dpp(0, (int)before12, 0x00, 0xF, 0xF, false);
  const u32 rec34 = (u32)__builtin_amdgcn_update_dpp(0, (int)before34, 0x00, 0xF, 0xF, false);
  const u32 rec5 = (u32)__builtin_amdgcn_update_dpp(0, (int)before5, 0x00, 0xF, 0xF, false);
  const u32 f1 = br1 + (rec12 & 0xFFFF), f2 = br2 + (rec12 >> 16), f3 = br3 + (rec34 & 0xFFFF), f4 = br4 + (rec34 >> 16), f5 = br5 + rec5;
  // Word k of the 128-bit header (five 25-bit fields at bit 0, 25, 50, 75, 100: pack_header, bwtm_device.h) holds the top of field k + 1
  // from bit 7 k of that field on and the bottom of field k + 2 at bit 25 - 7 k; lane k of the quad computes only that word.
  const u32 k = lane & 3;
  const u32 fx = (k == 0 ? f1 : (k == 1 ? f2 : (k == 2 ? f3 : f4))), fy = (k == 0 ? f2 : (k == 1 ? f3 : (k == 2 ? f4 : f5)));
  const u32 hk = (fx >> (7 * k)) | (fy << (25 - 7 * k));
  const u64 q = w >> 2;
  if(q >= q_lo && q < q_hi) { recs_out[w] = make_uint4(o0, o1, o2, hk); }
}
