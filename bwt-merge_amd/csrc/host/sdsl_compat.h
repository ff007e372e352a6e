/*
  sdsl_compat.h -- the few SDSL container encodings the native file format embeds
  (int_vector<8/64/1/0>, sd_vector<> with its two select_support_mcl members), written from the
  published description of sdsl-lite.

  UNVERIFIED AGAINST REAL SDSL OUTPUT: sdsl-lite is not available in this environment, so files
  written here round-trip through this reader (tests) but byte compatibility with SDSL-built
  files is unpinned.  Only serialization is provided; queries use plain arrays (bwt.h).
  The payload the hot path computes (header, BWT::data, the sample VALUES, C) is pinned against the
  oracle; what is unpinned is the bit layout of sd_vector / select_support_mcl around those values.
  The select supports follow sdsl-lite's two construction paths as recalled: vectors below 100 000
  bits take init_slow (every superblock, the last partial one included, is long or mini by its span),
  larger ones init_fast (the last partial superblock is always stored long, with the width of the
  vector's last position, and its entry in the superblock array is left 0).
*/
#ifndef BWTM_HOST_SDSL_COMPAT_H
#define BWTM_HOST_SDSL_COMPAT_H

#include <istream>
#include <ostream>
#include "utils.h"

namespace bwtmerge
{
namespace sdsl_compat
{

inline size_type hi(size_type x) { return (x == 0 ? 0 : 63 - (size_type)__builtin_clzll(x)); }   // sdsl::bits::hi

template<class T> void write_member(const T& v, std::ostream& out) { out.write((const char*)&v, sizeof(T)); }
template<class T> void read_member(T& v, std::istream& in) { in.read((char*)&v, sizeof(T)); }

// Bit-packed vector of fixed-width integers (int_vector<0> when `dynamic_width`).
struct PackedVector
{
  size_type width = 64, count = 0;
  std::vector<std::uint64_t> words;

  PackedVector() {}
  PackedVector(size_type n, size_type w) : width(w == 0 ? 1 : w), count(n), words((n * (w == 0 ? 1 : w) + 63) / 64 + 1, 0) {}

  void set(size_type i, size_type v)
  {
    size_type bit = i * width, word = bit / 64, off = bit % 64;
    if(width < 64) { v &= ((size_type)1 << width) - 1; }
    words[word] |= v << off;
    if(off + width > 64) { words[word + 1] |= v >> (64 - off); }
  }
  size_type get(size_type i) const
  {
    size_type bit = i * width, word = bit / 64, off = bit % 64;
    size_type v = words[word] >> off;
    if(off + width > 64) { v |= words[word + 1] << (64 - off); }
    if(width < 64) { v &= ((size_type)1 << width) - 1; }
    return v;
  }
  size_type bit_size() const { return count * width; }

  // int_vector<w>::serialize: u64 size in bits, [u8 width when dynamic], ceil(bits / 64) words.
  void serialize(std::ostream& out, bool dynamic_width) const
  {
    size_type bits = bit_size();
    write_member(bits, out);
    if(dynamic_width) { std::uint8_t w = (std::uint8_t)width; write_member(w, out); }
    out.write((const char*)words.data(), ((bits + 63) / 64) * 8);
  }
  void load(std::istream& in, bool dynamic_width, size_type fixed_width)
  {
    size_type bits = 0; read_member(bits, in);
    width = fixed_width;
    if(dynamic_width) { std::uint8_t w = 0; read_member(w, in); width = (w == 0 ? 1 : w); }
    count = bits / width;
    words.assign((bits + 63) / 64 + 1, 0);
    in.read((char*)words.data(), ((bits + 63) / 64) * 8);
  }
};

// select_support_mcl<b, 1> over a bit vector: serialization only.
struct SelectMCL
{
  static void serialize(std::ostream& out, const PackedVector& bits, bool select_ones)
  {
    const size_type SUPER = 4096;
    size_type n = bits.bit_size();
    // positions of the selected bits, word by word (width-1 vector: bit i of the vector is bit i % 64 of word i / 64)
    std::vector<size_type> args;
    for(size_type w = 0; w * 64 < n; w++)
    {
      std::uint64_t word = (select_ones ? bits.words[w] : ~bits.words[w]);
      if((w + 1) * 64 > n) { word &= (~(std::uint64_t)0) >> (64 - (n - w * 64)); }
      while(word != 0) { args.push_back(w * 64 + (size_type)__builtin_ctzll(word)); word &= word - 1; }
    }
    size_type arg_cnt = args.size();
    write_member(arg_cnt, out);
    if(arg_cnt == 0) { return; }
    size_type capacity = ((n + 63) / 64) * 64;
    size_type logn = hi(capacity) + 1, logn4 = logn * logn * logn * logn;
    size_type sb = (arg_cnt + SUPER - 1) / SUPER;
    const bool fast = (n >= 100000);                                   // select_support_mcl's constructor: init_slow below, init_fast from there on
    const bool tail_long = (fast && arg_cnt % SUPER != 0);             // init_fast: "handle last block: append long superblock"
    PackedVector superblock(sb, logn);
    std::vector<bool> is_long(sb, false);
    bool any_long = false;
    for(size_type s = 0; s < sb; s++)
    {
      size_type first = args[s * SUPER], last = args[std::min(arg_cnt, (s + 1) * SUPER) - 1];
      if(tail_long && s + 1 == sb) { is_long[s] = true; any_long = true; continue; }   // its superblock entry stays 0
      superblock.set(s, first);
      if(last - first > logn4) { is_long[s] = true; any_long = true; }
    }
    superblock.serialize(out, true);
    PackedVector mini_or_long(any_long ? sb : 0, 1);
    if(any_long) { for(size_type s = 0; s < sb; s++) { if(!is_long[s]) { mini_or_long.set(s, 1); } } }
    mini_or_long.serialize(out, false);
    for(size_type s = 0; s < sb; s++)
    {
      size_type begin = s * SUPER, end = std::min(arg_cnt, (s + 1) * SUPER);
      if(is_long[s])
      {
        PackedVector block(SUPER, (tail_long && s + 1 == sb ? hi(n - 1) : hi(args[end - 1])) + 1);
        for(size_type k = begin; k < end; k++) { block.set(k - begin, args[k]); }
        block.serialize(out, true);
      }
      else
      {
        PackedVector block(64, hi(args[end - 1] - args[begin]) + 1);
        for(size_type k = begin; k < end; k += 64) { block.set((k - begin) / 64, args[k] - args[begin]); }
        block.serialize(out, true);
      }
    }
  }

  // Skips a serialized select support (the facade rebuilds nothing from it).
  static void skip(std::istream& in)
  {
    size_type arg_cnt = 0; read_member(arg_cnt, in);
    if(arg_cnt == 0) { return; }
    size_type sb = (arg_cnt + 4095) / 4096;
    PackedVector v; v.load(in, true, 0);
    PackedVector mini_or_long; mini_or_long.load(in, false, 1);
    for(size_type s = 0; s < sb; s++) { v.load(in, true, 0); }
  }
};

// sd_vector<>: Elias-Fano over `ones` (strictly increasing) in a universe of `size` bits.
struct SDVector
{
  static void serialize(std::ostream& out, size_type size, const std::vector<size_type>& ones)
  {
    size_type m = ones.size();
    size_type logm = hi(m) + 1, logn = hi(size) + 1;
    if(logm == logn) { logm--; }
    std::uint8_t wl = (std::uint8_t)(logn - logm);
    PackedVector low(m, wl), high(m + ((size_type)1 << logm), 1);
    for(size_type k = 0; k < m; k++)
    {
      low.set(k, ones[k]);
      high.set((ones[k] >> wl) + k, 1);
    }
    write_member(size, out);
    write_member(wl, out);
    low.serialize(out, true);
    high.serialize(out, false);
    SelectMCL::serialize(out, high, true);
    SelectMCL::serialize(out, high, false);
  }

  static void load(std::istream& in, size_type& size, std::vector<size_type>& ones)
  {
    std::uint8_t wl = 0;
    read_member(size, in); read_member(wl, in);
    PackedVector low, high;
    low.load(in, true, 0); high.load(in, false, 1);
    SelectMCL::skip(in); SelectMCL::skip(in);
    ones.clear(); ones.reserve(low.count);
    size_type k = 0, nbits = high.bit_size();
    for(size_type w = 0; w * 64 < nbits && k < low.count; w++)
    {
      std::uint64_t word = high.words[w];
      while(word != 0 && k < low.count)
      {
        const size_type pos = w * 64 + (size_type)__builtin_ctzll(word); word &= word - 1;
        if(pos >= nbits) { break; }
        ones.push_back(((pos - k) << wl) | low.get(k)); k++;
      }
    }
  }
};

} // namespace sdsl_compat
} // namespace bwtmerge

#endif // BWTM_HOST_SDSL_COMPAT_H
