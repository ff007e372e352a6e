/*
  kernels/search_range.hip.h -- the first levels of the search in the reference's own form: nodes of B's reverse trie.
  Part of bwtm_kernels.hip.h (included there, inside namespace bwtm); gfx950 only.

  buildRA (fmi.cpp:272-334) pops nodes (a_pos, b_range): all sequences of B that share the suffix spelled so far occupy ONE
  range [sp, ep] of B and ONE position r of A, emit the run (r, ep - sp + 1) (fmi.cpp:290) and have at most five children
  (fmi.cpp:304-322; symbol 0 ends its chains).  While the trie has far fewer nodes than B has sequences -- the first ~log4(m_B)
  levels of a read collection, every level of a collection of repeated reads -- a level costs three rank-all-symbols queries
  per NODE instead of two record reads per SEQUENCE.  The frontier search (search_frontier.hip.h) keeps one element per sequence;
  it takes over, from nodes expanded into elements, when the nodes of a level approach the number of sequences.

  A level is kept sorted by sp (the order of the frontier): the children of a level sorted by (symbol, parent) are sorted by sp,
  because LF is monotone for a fixed symbol and the symbols own increasing ranges of B.

    k_range_step<false>   per node: ranks of B at sp and ep + 1, child flags (symbol-major, for one exclusive scan), the node's
                          run of bits [sp + r, ep + r] into the interleaving bitvector (short runs in place, long ones as pieces)
    k_range_step<true>    per node: ranks again + ranks of A at r, children written at the scanned positions
    k_range_emit          the long runs, one workgroup per piece
    k_range_expand        nodes -> one frontier element per sequence (the layout k_frontier_init produces)
*/
#pragma once

constexpr u32 RANGE_INLINE_WORDS = 4;          // runs that touch at most this many 32-bit words are set by the node's own lane
constexpr u32 RANGE_PIECE_WORDS = 2048;        // longer runs: pieces of at most this many words, one workgroup each
constexpr u32 RANGE_EXPAND_INLINE = 48;        // nodes of at most this many sequences are expanded by their own lane
constexpr u32 RANGE_EXPAND_PIECE = 1024;

// A long run of bits (or of elements) cut into pieces: [first, first + count) in units of bits (elements); `aux` = r for k_range_expand.
struct RangePiece { u64 first; u64 count; u64 aux; u64 dst; };

__device__ inline void range_set_bits(u32* bits, u64 p0, u64 p1)      // sets [p0, p1), p1 > p0, both inside a few words
{
  for(u64 w = p0 >> 5; w <= ((p1 - 1) >> 5); w++)
  {
    const u64 lo = (w << 5 > p0 ? w << 5 : p0), hi = (((w + 1) << 5) < p1 ? (w + 1) << 5 : p1);
    const u32 n = (u32)(hi - lo), sh = (u32)(lo & 31);
    const u32 mask = (n == 32 ? 0xFFFFFFFFu : ((1u << n) - 1u) << sh);
    atomicOr(bits + w, mask);
  }
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_range_init(u64* sp, u64* r, u64* cnt, u64 seq_first, u64 count, u64 m_a)
{
  if(blockIdx.x == 0 && threadIdx.x == 0) { sp[0] = seq_first; r[0] = m_a; cnt[0] = count; }      // fmi.cpp:286: the root "$"
}

template<bool WRITE>
__global__ void __launch_bounds__(BLOCK_THREADS) k_range_step(IndexView A, IndexView B, const u64* sp, const u64* r, const u64* cnt, u64 N,
  u64* flags, const u64* prefix, u64* sp_next, u64* r_next, u64* cnt_next, u32* bits, RangePiece* pieces, u32* npieces, u32 piece_cap)
{
  const u64 u = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(u >= N) { return; }
  const u64 s = sp[u], k = cnt[u], rr = r[u];
  u64 rs[6], re[6];
  index_ranks(B, s, rs);
  index_ranks(B, s + k, re);                                        // ep + 1 <= n_B: the records cover position n
  if(!WRITE)
  {
#pragma unroll
    for(u32 c = 1; c < 6; c++) { flags[(u64)(c - 1) * N + u] = (re[c] > rs[c] ? 1 : 0); }
    if(u == 0) { flags[5 * N] = 0; }
    // fmi.cpp:290: the record (a_pos, |b_range|) = bits [sp + r, ep + r] of the interleaving bitvector (paper.tex:166)
    const u64 p0 = s + rr, p1 = p0 + k;
    if(((p1 - 1) >> 5) - (p0 >> 5) < RANGE_INLINE_WORDS) { range_set_bits(bits, p0, p1); }
    else
    {
      const u64 per = (u64)RANGE_PIECE_WORDS * 32;
      const u64 first_cut = ((p0 / per) + 1) * per;                 // pieces end at multiples of `per`: no word is shared by two pieces of one run
      const u32 n = (u32)(p1 <= first_cut ? 1 : 1 + (p1 - first_cut + per - 1) / per);
      const u32 at = atomicAdd(npieces, n);
      u64 b = p0;
      for(u32 q = 0; q < n; q++)
      {
        u64 e = (b / per + 1) * per; if(e > p1) { e = p1; }
        if(at + q < piece_cap) { RangePiece pc; pc.first = b; pc.count = e - b; pc.aux = 0; pc.dst = 0; pieces[at + q] = pc; }
        else { range_set_bits(bits, b, e); }                        // the list is sized for every case; exact anyway
        b = e;
      }
    }
  }
  else
  {
    u64 ra[6];
    index_ranks(A, rr, ra);
#pragma unroll
    for(u32 c = 1; c < 6; c++)
    {
      if(re[c] > rs[c])
      {
        const u64 d = prefix[(u64)(c - 1) * N + u];
        sp_next[d] = B.C[c] + rs[c];                                // LF(range, c), utils.h:350-355
        cnt_next[d] = re[c] - rs[c];
        r_next[d] = A.C[c] + ra[c];                                 // LF(a_pos, c), utils.h:343-348
      }
    }
  }
}

// One workgroup per piece (grid-stride): every word of the piece's bit range.
__global__ void __launch_bounds__(BLOCK_THREADS) k_range_emit(const RangePiece* pieces, const u32* npieces, u32 piece_cap, u32* bits)
{
  u32 n = *npieces; if(n > piece_cap) { n = piece_cap; }
  for(u32 q = blockIdx.x; q < n; q += gridDim.x)
  {
    const u64 p0 = pieces[q].first, p1 = p0 + pieces[q].count;
    const u64 w0 = p0 >> 5, w1 = (p1 - 1) >> 5;
    for(u64 w = w0 + threadIdx.x; w <= w1; w += BLOCK_THREADS)
    {
      const u64 lo = (w << 5 > p0 ? w << 5 : p0), hi = (((w + 1) << 5) < p1 ? (w + 1) << 5 : p1);
      const u32 nb = (u32)(hi - lo), sh = (u32)(lo & 31);
      atomicOr(bits + w, (nb == 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u) << sh));
    }
  }
}

// Nodes -> elements.  offset[u] = exclusive scan of cnt: node u becomes elements offset[u] .. offset[u] + cnt[u] - 1 with
// coordinates (sp + j, r).  The same physical layout as k_frontier_init (contiguous, sorted by i).
__device__ inline void range_store_element(uint2* lo, unsigned short* hi, u64 g, u64 i, u64 r)
{
  lo[g] = make_uint2((u32)i, (u32)r);
  if(hi) { hi[g] = (unsigned short)(((i >> 32) & 0xFF) | (((r >> 32) & 0xFF) << 8)); }
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_range_expand(const u64* sp, const u64* r, const u64* cnt, const u64* offset, u64 N,
  uint2* lo, unsigned short* hi, RangePiece* pieces, u32* npieces, u32 piece_cap)
{
  const u64 u = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(u >= N) { return; }
  const u64 s = sp[u], k = cnt[u], rr = r[u], g0 = offset[u];
  if(k <= RANGE_EXPAND_INLINE) { for(u64 j = 0; j < k; j++) { range_store_element(lo, hi, g0 + j, s + j, rr); } return; }
  const u32 n = (u32)((k + RANGE_EXPAND_PIECE - 1) / RANGE_EXPAND_PIECE);
  const u32 at = atomicAdd(npieces, n);
  for(u32 q = 0; q < n; q++)
  {
    const u64 b = (u64)q * RANGE_EXPAND_PIECE;
    const u64 c = (k - b < (u64)RANGE_EXPAND_PIECE ? k - b : (u64)RANGE_EXPAND_PIECE);
    if(at + q < piece_cap) { RangePiece pc; pc.first = s + b; pc.count = c; pc.aux = rr; pc.dst = g0 + b; pieces[at + q] = pc; }
    else { for(u64 j = 0; j < c; j++) { range_store_element(lo, hi, g0 + b + j, s + b + j, rr); } }
  }
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_range_expand_pieces(const RangePiece* pieces, const u32* npieces, u32 piece_cap, uint2* lo, unsigned short* hi)
{
  u32 n = *npieces; if(n > piece_cap) { n = piece_cap; }
  for(u32 q = blockIdx.x; q < n; q += gridDim.x)
  {
    const RangePiece pc = pieces[q];
    for(u64 j = threadIdx.x; j < pc.count; j += BLOCK_THREADS) { range_store_element(lo, hi, pc.dst + j, pc.first + j, pc.aux); }
  }
}

// The segment tables of a frontier whose coordinates are already in place (k_range_expand): what k_frontier_init writes for them.
__global__ void __launch_bounds__(BLOCK_THREADS) k_frontier_init_tables(u64* seg_len, u64* seg_phys, u64 nb_max, u64 count)
{
  u64 g = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(g < 5 * nb_max)
  {
    u64 cls = g / nb_max, b = g % nb_max;
    u64 begin = b * FR_BLOCK;
    seg_len[g] = (cls == 0 && begin < count ? (count - begin < (u64)FR_BLOCK ? count - begin : (u64)FR_BLOCK) : 0);
    seg_phys[g] = begin;
  }
  if(g == 5 * nb_max) { seg_len[g] = 0; }
}
