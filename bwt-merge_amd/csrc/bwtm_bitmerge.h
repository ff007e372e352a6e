/*
  bwtm_bitmerge.h -- the bit merge of mergeBWT (bwt.cpp:215-282) on 32 output positions: position j takes the next bit of B where the
  mask bit m_j is set and the next bit of A otherwise,
        out_j = m_j ? b[r_j] : a[z_j],      r_j = #ones of m below j,  z_j = j - r_j = #zeros of m below j.
  Pure host + device helper (tests/host_shim.cpp checks it bit by bit against the definition).

  Round 4.  The first formulation was out = expand(b, m) | expand(a, ~m) with the classic parallel-suffix expand: five rounds per
  mask in which the mask itself is compressed so that the next round's move bits are stated in the coordinates the data has reached
  (2 x ~80 instructions for the masks, 6 x 11 for the deposits: ~230 of k_interleave's 500 per word, and the kernel is issue-bound
  for two thirds of its time).  Here the moves are stated at the DESTINATION instead:

    * out_j = b[j - z_j] is a gather with a displacement d_j = z_j that is non-decreasing in j and grows by at most one per position
      (the same holds for a with d_j = r_j).  For such a displacement the gather decomposes into five "pull" rounds from the most
      significant bit of d down, each with a mask in FINAL coordinates:   y_j <- D_i[j] ? y_{j - 2^i} : y_j   (D_i = bit i of d_j).
      Invariant: after the rounds 4 .. i, y_j = x[j - (d_j with the bits below i cleared)].  Proof of a round: when D_i[j] is set,
      d_j = H 2^(i+1) + 2^i + c with c < 2^i, and d_{j - 2^i} lies in [d_j - 2^i, d_j] = [H 2^(i+1) + c, H 2^(i+1) + 2^i + c], whose members
      all have the same bits above i as d_j -- so what position j - 2^i holds after the earlier rounds was pulled over exactly the
      distance j still has to go.  No mask is ever compressed: 2 instructions per round and plane (shift, bit-field insert).
    * The displacements' bit planes come from ONE bit-sliced prefix sum: P = inclusive prefix popcount of m by a Kogge-Stone scan
      over the 32 positions with bit-sliced adders (level k adds k-bit numbers at distance 2^(k-1): 3 k instructions with three-input
      boolean ops for sum and carry, 45 in all), R_i = P_i << 1 (exclusive), and Z = J - R by a bit-sliced subtraction from the
      constant planes of j (9 instructions).
  About 125 instructions instead of 230, and half the live mask registers.
*/
#ifndef BWTM_BITMERGE_H
#define BWTM_BITMERGE_H

#include "bwtm_device.h"

namespace bwtm
{

// Bit planes of the two displacements of a 32-bit mask: R[i] = bit i of r_j (ones below j), Z[i] = bit i of z_j (zeros below j).
struct MergeMasks { u32 R[5], Z[5], m; };

BWTM_HD u32 bm_maj(u32 a, u32 b, u32 c) { return (a & b) | (c & (a | b)); }

BWTM_HD MergeMasks merge_masks(u32 m)
{
  MergeMasks e; e.m = m;
  // inclusive prefix popcount P_j = #ones of m in [0, j], bit-sliced in p0 .. p4 (bit 5 is set only for P_31 = 32, which no r_j needs)
  u32 t0, t1, t2, t3, c;
  // level 1: 1-bit + 1-bit at distance 1
  t0 = m << 1;
  u32 p0 = m ^ t0, p1 = m & t0;
  // level 2: 2-bit + 2-bit at distance 2
  t0 = p0 << 2; t1 = p1 << 2;
  c = p0 & t0; p0 ^= t0;
  u32 p2 = bm_maj(p1, t1, c); p1 = p1 ^ t1 ^ c;
  // level 3: 3-bit + 3-bit at distance 4
  t0 = p0 << 4; t1 = p1 << 4; t2 = p2 << 4;
  c = p0 & t0; p0 ^= t0;
  u32 c1 = bm_maj(p1, t1, c); p1 = p1 ^ t1 ^ c;
  u32 p3 = bm_maj(p2, t2, c1); p2 = p2 ^ t2 ^ c1;
  // level 4: 4-bit + 4-bit at distance 8
  t0 = p0 << 8; t1 = p1 << 8; t2 = p2 << 8; t3 = p3 << 8;
  c = p0 & t0; p0 ^= t0;
  c1 = bm_maj(p1, t1, c); p1 = p1 ^ t1 ^ c;
  u32 c2 = bm_maj(p2, t2, c1); p2 = p2 ^ t2 ^ c1;
  u32 p4 = bm_maj(p3, t3, c2); p3 = p3 ^ t3 ^ c2;
  // level 5: 5-bit + 5-bit at distance 16 (the carry out of bit 4 is bit 5: dropped)
  t0 = p0 << 16; t1 = p1 << 16; t2 = p2 << 16; t3 = p3 << 16; const u32 t4 = p4 << 16;
  c = p0 & t0; p0 ^= t0;
  c1 = bm_maj(p1, t1, c); p1 = p1 ^ t1 ^ c;
  c2 = bm_maj(p2, t2, c1); p2 = p2 ^ t2 ^ c1;
  const u32 c3 = bm_maj(p3, t3, c2); p3 = p3 ^ t3 ^ c2;
  p4 = p4 ^ t4 ^ c3;
  // exclusive: r_j = P_{j-1}
  e.R[0] = p0 << 1; e.R[1] = p1 << 1; e.R[2] = p2 << 1; e.R[3] = p3 << 1; e.R[4] = p4 << 1;
  // z_j = j - r_j: bit-sliced subtraction from the planes of j (borrow = ~x & y | ~(x ^ y) & borrow_in)
  const u32 J0 = 0xAAAAAAAAu, J1 = 0xCCCCCCCCu, J2 = 0xF0F0F0F0u, J3 = 0xFF00FF00u, J4 = 0xFFFF0000u;
  u32 bw;
  e.Z[0] = J0 ^ e.R[0]; bw = ~J0 & e.R[0];
  e.Z[1] = J1 ^ e.R[1] ^ bw; bw = (~J1 & (e.R[1] | bw)) | (e.R[1] & bw);
  e.Z[2] = J2 ^ e.R[2] ^ bw; bw = (~J2 & (e.R[2] | bw)) | (e.R[2] & bw);
  e.Z[3] = J3 ^ e.R[3] ^ bw; bw = (~J3 & (e.R[3] | bw)) | (e.R[3] & bw);
  e.Z[4] = J4 ^ e.R[4] ^ bw;
  return e;
}

// y_j = x[j - d_j] for a displacement with bit planes D[0..4] that is non-decreasing in j with steps of at most one.
BWTM_HD u32 pull32(u32 x, const u32 D[5])
{
#pragma unroll
  for(int i = 4; i >= 0; i--)
  {
    const u32 t = x << (1u << i);
    x = (x & ~D[i]) | (t & D[i]);
  }
  return x;
}

// One plane of the merge: b's next bits where m is set, a's next bits elsewhere.
BWTM_HD u32 bit_merge32(u32 a, u32 b, const MergeMasks& e)
{
  const u32 ya = pull32(a, e.R), yb = pull32(b, e.Z);
  return (yb & e.m) | (ya & ~e.m);
}

} // namespace bwtm

#endif // BWTM_BITMERGE_H
