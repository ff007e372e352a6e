/*
  bwtm_view.h -- record layout and pure (host + device) helpers of the two-plane SEARCH VIEW.  EXPERIMENTAL: compiled into the library
  only with -DBWTM_EXPERIMENTAL (libbwtm_experimental.so); the product build does not contain it.  A measured negative result
  (14 % fewer HBM reads, no time saved), kept exact and tested so that it can be measured again.
*/
#ifndef BWTM_VIEW_H
#define BWTM_VIEW_H

#include "bwtm_device.h"

namespace bwtm
{

//------------------------------------------------------------------------------
// Search view: a denser copy of the rank structure that only the level-synchronous search reads.
// A full LF step streams every record of both indexes once and is bound by those bytes alone, and
// almost every position holds one of A, C, G, T: two bit-planes hold 160 positions in the bytes three planes need for 107.
//
//     view record q covers positions [160 q, 160 q + 160), 16 x u32
//       v[0..4]   bit 0 of (symbol - 1) for the 160 positions     (A, C, G, T -> 0, 1, 2, 3)
//       v[5..9]   bit 1
//       v[10..13] header: five 25-bit fields, field c - 1 = #c in [view super start, 160 q); bit 125 = OVERFLOW
//       v[14..15] up to seven exceptions in ascending order: byte k = position of exception k in the record (0xFF = none),
//                 byte 7 = their kinds (bit k: 0 = endmarker, 1 = N).  Exceptions are stored as 'A' in the planes.
//     view super  vsup[8 s + c] = #c in [0, s * VIEW_SUPER_POS), one line per 2^17 records (counts inside a super fit 25 bits)
//
// A record with more than seven exceptions sets OVERFLOW; its elements take the ordinary 64-byte record of the same
// position instead (one more dependent access, for the 0.2 % of the records of a read collection that need it).

constexpr u32 VIEW_POS        = 160;
constexpr u32 VIEW_WORDS      = 5;                  // words per plane
constexpr int VIEW_SUPER_SHIFT = 17;                // view records per view super
constexpr u64 VIEW_SUPER_POS  = (u64)VIEW_POS << VIEW_SUPER_SHIFT;     // 20 971 520 positions < 2^25
constexpr u32 VIEW_EXC_SLOTS  = 7;
constexpr u32 VIEW_EXC_EMPTY  = 0xFFu;
constexpr u32 VIEW_OVERFLOW_BIT = 29;               // bit 125 of the header = bit 29 of v[13]

BWTM_HD u64 num_view_records(u64 n) { return n / VIEW_POS + 1; }
BWTM_HD u64 num_view_supers(u64 n)  { return (n / VIEW_POS >> VIEW_SUPER_SHIFT) + 1; }

// What the exceptions of a record say about in-record position j: how many of them lie below j (they sit in the planes as 'A'),
// how many of those are N, and the symbol of the exception AT j (0 or 5; 6 = position j is no exception).  The positions are
// ascending, so the slots below j are a prefix.
BWTM_HD void view_exceptions(u32 e_lo, u32 e_hi, u32 j, u32& below, u32& below_n, u32& at)
{
  const u32 p0 = e_lo & 0xFF, p1 = (e_lo >> 8) & 0xFF, p2 = (e_lo >> 16) & 0xFF, p3 = e_lo >> 24;
  const u32 p4 = e_hi & 0xFF, p5 = (e_hi >> 8) & 0xFF, p6 = (e_hi >> 16) & 0xFF, kinds = e_hi >> 24;
  below = (p0 < j) + (p1 < j) + (p2 < j) + (p3 < j) + (p4 < j) + (p5 < j) + (p6 < j);
  below_n = (u32)__builtin_popcount(kinds & ((1u << below) - 1u));
  const bool hit = (p0 == j) | (p1 == j) | (p2 == j) | (p3 == j) | (p4 == j) | (p5 == j) | (p6 == j);      // then it is slot `below`
  at = (hit ? (((kinds >> below) & 1u) ? 5u : 0u) : 6u);
}

// Symbol at in-record position j (0..159) of a view record without OVERFLOW; `at` from view_exceptions.
BWTM_HD u32 view_symbol(const u32* v, u32 j, u32 at)
{
  const u32 w = j >> 5, t = j & 31;
  const u32 a0 = (w == 0 ? v[0] : (w == 1 ? v[1] : (w == 2 ? v[2] : (w == 3 ? v[3] : v[4]))));
  const u32 a1 = (w == 0 ? v[5] : (w == 1 ? v[6] : (w == 2 ? v[7] : (w == 3 ? v[8] : v[9]))));
  const u32 c = 1 + ((a0 >> t) & 1u) + 2 * ((a1 >> t) & 1u);
  return (at == 6 ? c : at);
}

// Occurrences of c (1..5) among the first j positions (0..160) of a view record without OVERFLOW; below / below_n from view_exceptions.
BWTM_HD u32 view_count(const u32* v, u32 c, u32 j, u32 below, u32 below_n)
{
  if(c == 5) { return below_n; }
  const u32 x0 = ((c - 1) & 1) ? 0u : ~0u, x1 = ((c - 1) & 2) ? 0u : ~0u;          // plane ^ x = bits that match the code
  const u32 w = j >> 5, part = (1u << (j & 31)) - 1u;
  u32 total = 0;
#pragma unroll
  for(u32 k = 0; k < VIEW_WORDS; k++)
  {
    const u32 m = (v[k] ^ x0) & (v[VIEW_WORDS + k] ^ x1);
    const u32 mask = (k < w ? ~0u : (k == w ? part : 0u));
    total += (u32)__builtin_popcount(m & mask);
  }
  return total - (c == 1 ? below : 0u);
}

// Relative count field of symbol c (1..5) of a view record (the same packing as rec_header, in v[10..13]).
BWTM_HD u32 view_header(const u32* v, u32 c)
{
  u64 lo = (u64)v[10] | ((u64)v[11] << 32);
  u64 hi = (u64)v[12] | ((u64)v[13] << 32);
  u32 sh = FIELD_BITS * (c - 1);
  u64 x;
  if(sh < 64) { x = lo >> sh; if(sh + FIELD_BITS > 64) { x |= hi << (64 - sh); } }
  else { x = hi >> (sh - 64); }
  return (u32)x & FIELD_MASK;
}

BWTM_HD bool view_overflow(const u32* v) { return ((v[13] >> VIEW_OVERFLOW_BIT) & 1u) != 0; }

} // namespace bwtm

#endif // BWTM_VIEW_H
