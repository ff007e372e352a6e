/*
  bwtm_device.h -- device-resident data layout and the pure (host + device) helpers shared by
  all kernels of the rank-array / interleave path.

  Device rank structure ("records").  The on-disk native format (64-byte run-length blocks,
  support.h:221-286) is what the library reads and writes, but LF() on it needs a
  position -> block predecessor query plus a sequential decode (bwt.cpp:318-341).  On the
  GPU each rank query should cost exactly ONE dependent HBM access, so the index is
  transcoded on the device into fixed-span, self-contained 64-byte records:

      record q covers sequence positions [128 q, 128 q + 128)
      32-bit word  w[4 k + 0..2]  = bit-planes 0..2 of positions 32 k .. 32 k + 31   (k = 0..3)
                   w[4 k + 3]     = word k of the 128-bit header
      header       five 25-bit fields: field c - 1 = #c in [super start, 128 q)   (c = 1..5)
      super table  sup[8 s + c]   = #c in [0, s * 2^25)  as u64   (one 64-byte line per super;
                   a few hundred lines, L2 resident)

  so  rank(i, c) = sup[8 (i >> 25) + c] + header(c) + popcount(match(c) & below(i & 127)),
  and BWT[i] is read from the same 64 bytes.  Records exist for positions 0 .. n inclusive
  (rank(n, c) is needed: utils.h:345-348 is called with i == size()).
*/
#ifndef BWTM_DEVICE_H
#define BWTM_DEVICE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define BWTM_HD __host__ __device__ inline
#else
#define BWTM_HD inline
#endif

namespace bwtm
{

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t  u8;

constexpr int REC_SHIFT       = 7;                  // 128 positions per record
constexpr u64 REC_POS         = 1ull << REC_SHIFT;
constexpr int REC_WORDS       = 16;                 // 64 bytes
constexpr int SUPER_SHIFT     = 25;                 // positions per super block
constexpr int SUPER_REC_SHIFT = SUPER_SHIFT - REC_SHIFT;
constexpr int SUP_STRIDE      = 8;                  // u64 per super entry (64-byte line)
constexpr u32 FIELD_BITS      = 25;
constexpr u32 FIELD_MASK      = (1u << FIELD_BITS) - 1;

constexpr u64 SIGMA     = 6;                        // support.h:228
constexpr u64 RLE_BLOCK = 64;                       // support.h:227
constexpr u64 MAX_RUN   = 42;                       // support.h:229

BWTM_HD u64 num_records(u64 n) { return (n >> REC_SHIFT) + 1; }
BWTM_HD u64 num_supers(u64 n)  { return (n >> SUPER_SHIFT) + 1; }

//------------------------------------------------------------------------------
// Record helpers.  `w` is the record's 16 words (in registers after unrolling).

// Bits of one 32-position word whose symbol equals c.
BWTM_HD u32 plane_match(u32 p0, u32 p1, u32 p2, u32 c)
{
  u32 m0 = (c & 1) ? p0 : ~p0;
  u32 m1 = (c & 2) ? p1 : ~p1;
  u32 m2 = (c & 4) ? p2 : ~p2;
  return m0 & m1 & m2;
}

// Mask of the bits of word k (positions 32 k ..) that lie below in-record position j (0..128).
BWTM_HD u32 below_mask(u32 j, u32 k)
{
  int d = (int)j - 32 * (int)k;
  return (d <= 0 ? 0u : (d >= 32 ? 0xFFFFFFFFu : ((1u << d) - 1u)));
}

// Number of occurrences of c among the first j positions of the record.
BWTM_HD u32 rec_count(const u32* w, u32 c, u32 j)
{
  u32 total = 0;
#pragma unroll
  for(int k = 0; k < 4; k++)
  {
    total += (u32)__builtin_popcount(plane_match(w[4 * k], w[4 * k + 1], w[4 * k + 2], c) & below_mask(j, (u32)k));
  }
  return total;
}

// Symbol at in-record position j (0..127).
BWTM_HD u32 rec_symbol(const u32* w, u32 j)
{
  bool upper = (j & 64) != 0;
  u32 t = j & 63;
  u64 p0 = (upper ? ((u64)w[8]  | ((u64)w[12] << 32)) : ((u64)w[0] | ((u64)w[4] << 32)));
  u64 p1 = (upper ? ((u64)w[9]  | ((u64)w[13] << 32)) : ((u64)w[1] | ((u64)w[5] << 32)));
  u64 p2 = (upper ? ((u64)w[10] | ((u64)w[14] << 32)) : ((u64)w[2] | ((u64)w[6] << 32)));
  return (u32)((p0 >> t) & 1) | ((u32)((p1 >> t) & 1) << 1) | ((u32)((p2 >> t) & 1) << 2);
}

// Relative count field of symbol c (1..5).
BWTM_HD u32 rec_header(const u32* w, u32 c)
{
  u64 lo = (u64)w[3] | ((u64)w[7] << 32);
  u64 hi = (u64)w[11] | ((u64)w[15] << 32);
  u32 sh = FIELD_BITS * (c - 1);                    // 0, 25, 50, 75, 100
  u64 v;
  if(sh < 64) { v = lo >> sh; if(sh + FIELD_BITS > 64) { v |= hi << (64 - sh); } }
  else { v = hi >> (sh - 64); }
  return (u32)v & FIELD_MASK;
}

// Packs the five relative counts into the four header words.
BWTM_HD void pack_header(const u32 rel[6], u32 h[4])
{
  u64 lo = 0, hi = 0;
  for(u32 c = 1; c <= 5; c++)
  {
    u64 v = rel[c] & FIELD_MASK;
    u32 sh = FIELD_BITS * (c - 1);
    if(sh < 64) { lo |= v << sh; if(sh + FIELD_BITS > 64) { hi |= v >> (64 - sh); } }
    else { hi |= v << (sh - 64); }
  }
  h[0] = (u32)lo; h[1] = (u32)(lo >> 32); h[2] = (u32)hi; h[3] = (u32)(hi >> 32);
}

// Bits [from, from + count) of a 128-bit field held as two 64-bit halves.
BWTM_HD void range_mask128(u32 from, u32 count, u64& lo, u64& hi)
{
  u32 to = from + count;                            // <= 128
  u64 lo_to   = (to >= 64 ? ~0ull : ((1ull << to) - 1));
  u64 lo_from = (from >= 64 ? ~0ull : ((1ull << from) - 1));
  lo = lo_to & ~lo_from;
  u64 hi_to   = (to <= 64 ? 0ull : (to >= 128 ? ~0ull : ((1ull << (to - 64)) - 1)));
  u64 hi_from = (from <= 64 ? 0ull : ((1ull << (from - 64)) - 1));
  hi = hi_to & ~hi_from;
}

//------------------------------------------------------------------------------
// Native run codec (support.h:221-286), decode side.  Reads one run at data[pos...].

//------------------------------------------------------------------------------
// Interleave helpers (mergeBWT, bwt.cpp:215-282).

// Bit deposit ("expand", the inverse of compress): bit k of x goes to the position of the k-th
// set bit of m.  The move masks depend on m only, so they are computed once and shared by the
// three planes.  32-bit words: 5 rounds of a 5-step parallel suffix.
struct ExpandMasks { u32 mv[5]; u32 m; };

BWTM_HD ExpandMasks expand_masks(u32 m)
{
  ExpandMasks e; e.m = m;
  u32 mk = ~m << 1;                           // counts the zeros to the right
#pragma unroll
  for(int i = 0; i < 5; i++)
  {
    u32 mp = mk ^ (mk << 1);
    mp ^= mp << 2; mp ^= mp << 4; mp ^= mp << 8; mp ^= mp << 16;
    u32 mv = mp & m;                          // bits that move by 1 << i
    e.mv[i] = mv;
    m = (m ^ mv) | (mv >> (1u << i));
    mk &= ~mp;
  }
  return e;
}

BWTM_HD u32 expand32(u32 x, const ExpandMasks& e)
{
#pragma unroll
  for(int i = 4; i >= 0; i--)
  {
    u32 t = x << (1u << i);
    x = (x & ~e.mv[i]) | (t & e.mv[i]);
  }
  return x & e.m;
}

// Deposits the next symbols of A (where the mask bit is 0) and B (where it is 1) into 64
// output positions.  a* / b* are 64-bit windows of the source planes.
BWTM_HD void deposit64(u64 mask, u64 a0, u64 a1, u64 a2, u64 b0, u64 b1, u64 b2, u64& o0, u64& o1, u64& o2)
{
  u32 m = (u32)mask;
  ExpandMasks eb = expand_masks(m), ea = expand_masks(~m);
  u32 l0 = expand32((u32)b0, eb) | expand32((u32)a0, ea);
  u32 l1 = expand32((u32)b1, eb) | expand32((u32)a1, ea);
  u32 l2 = expand32((u32)b2, eb) | expand32((u32)a2, ea);
  u32 nb = (u32)__builtin_popcount(m), na = 32 - nb;      // a 64-bit shift by 32 is fine (nb, na <= 32)
  b0 >>= nb; b1 >>= nb; b2 >>= nb; a0 >>= na; a1 >>= na; a2 >>= na;
  m = (u32)(mask >> 32);
  eb = expand_masks(m); ea = expand_masks(~m);
  u32 h0 = expand32((u32)b0, eb) | expand32((u32)a0, ea);
  u32 h1 = expand32((u32)b1, eb) | expand32((u32)a1, ea);
  u32 h2 = expand32((u32)b2, eb) | expand32((u32)a2, ea);
  o0 = (u64)l0 | ((u64)h0 << 32); o1 = (u64)l1 | ((u64)h1 << 32); o2 = (u64)l2 | ((u64)h2 << 32);
}

BWTM_HD void run_decode(const u8* data, u64& pos, u32& sym, u64& len)
{
  u32 code = data[pos]; pos++;
  sym = code % 6; len = code / 6 + 1;               // Run::decodeBasic, support.h:236-239
  if(len >= MAX_RUN)                                // support.h:248 + ByteCode::read, 172-184
  {
    u32 shift = 0, v;
    do { v = data[pos]; pos++; len += (u64)(v & 0x7F) << shift; shift += 7; } while(v & 0x80);
  }
}

//------------------------------------------------------------------------------
// Native run codec, encode side: Run::write for a run of length >= 1 appended at byte
// offset `offset` (only offset % 64 matters).  long_run_bytes() returns how many bytes
// Run::write emits; long_run_write() emits them.  Both follow support.h:256-282 step by step.

BWTM_HD u32 bit_length64(u64 x) { return (x == 0 ? 1u : 64u - (u32)__builtin_clzll(x)); }   // utils.h:146-151
BWTM_HD u32 varint_bytes(u64 x) { return (bit_length64(x) + 6) / 7; }                        // support.h:203-212

BWTM_HD u64 long_run_bytes(u64 offset, u64 length)
{
  // Closed form of the loop below for the runs that dominate in practice, 42 <= length < 42 + 128: head byte + one
  // extension byte, unless the head is the last byte of its block (basic length 41, the rest opens the next block).
  if(length >= MAX_RUN && length < MAX_RUN + 128)
  {
    if((offset % RLE_BLOCK) != RLE_BLOCK - 1) { return 2; }
    return (length - (MAX_RUN - 1) < MAX_RUN ? 2 : 3);
  }
  u64 bytes = 0;
  while(length > 0)
  {
    if(length < MAX_RUN) { bytes++; break; }
    u64 remaining = RLE_BLOCK - ((offset + bytes) % RLE_BLOCK);
    u64 basic = (remaining > 1 ? MAX_RUN : MAX_RUN - 1);
    bytes++; length -= basic; remaining--;
    if(remaining > 0)
    {
      u64 ext = length;
      if(bit_length64(length) > 7 * remaining) { ext = (~0ull) >> (64 - 7 * remaining); }
      bytes += varint_bytes(ext); length -= ext;
    }
  }
  return bytes;
}

// Writes the encoding of a run appended at byte offset `offset` of the stream; returns the number of bytes
// written.  The bytes go to out[offset - origin ...] (origin = stream offset of out[0]: a staging buffer
// holds a window of the stream).  Every piece of the run that opens a 64-byte block is reported with the sequence
// position it starts at (`run_start` = position of the run): on_block(block, position); position - 1 are the set bits
// of block_boundaries, bwt.cpp:496.
template<class F>
BWTM_HD u64 long_run_write_cb(u8* out, u64 offset, u32 sym, u64 length, u64 run_start, u64 origin, F&& on_block)
{
  u64 start = offset;
  while(length > 0)
  {
    if((offset % RLE_BLOCK) == 0) { on_block(offset / RLE_BLOCK, run_start); }
    if(length < MAX_RUN) { out[offset - origin] = (u8)(sym + 6 * (length - 1)); offset++; break; }
    u64 remaining = RLE_BLOCK - (offset % RLE_BLOCK);
    u64 basic = (remaining > 1 ? MAX_RUN : MAX_RUN - 1);
    out[offset - origin] = (u8)(sym + 6 * (basic - 1)); offset++; length -= basic; run_start += basic; remaining--;
    if(remaining > 0)
    {
      u64 ext = length;
      if(bit_length64(length) > 7 * remaining) { ext = (~0ull) >> (64 - 7 * remaining); }
      length -= ext; run_start += ext;
      while(ext > 0x7F) { out[offset - origin] = (u8)((ext & 0x7F) | 0x80); offset++; ext >>= 7; }
      out[offset - origin] = (u8)ext; offset++;
    }
  }
  return offset - start;
}

// The same with the block starts stored into an array (or dropped).
BWTM_HD u64 long_run_write(u8* out, u64 offset, u32 sym, u64 length, u64* block_start = nullptr, u64 run_start = 0, u64 origin = 0)
{
  return long_run_write_cb(out, offset, sym, length, run_start, origin, [&](u64 block, u64 position) { if(block_start) { block_start[block] = position; } });
}

} // namespace bwtm

#endif // BWTM_DEVICE_H
