/*
  bwt.h -- host facade: class BWT with the reference's public surface (bwt.h:41-189).  The
  run-length data and its samples live on the host as plain arrays (what block_boundaries and
  samples[c] encode, bwt.h:172-178); scalar queries run on them.  The interleaving constructor
  BWT(a, b, ra) (bwt.h:73, bwt.cpp:286-314) runs on the GPU through the C ABI (include/bwtm.h).
*/
#ifndef BWTM_HOST_BWT_H
#define BWTM_HOST_BWT_H

#include <array>
#include <cstdlib>
#include <bwtm.h>
#include "formats.h"

namespace bwtmerge
{

// Converts a C-ABI failure into the reference's behaviour: message on std::cerr, exit.
inline void gpuCheck(int rc, const char* where)
{
  if(rc != BWTM_OK)
  {
    std::cerr << where << ": " << bwtm_last_error() << std::endl;
    std::exit(EXIT_FAILURE);
  }
}

class BWT;

// Rank array of inserting b into a: a handle on the device bitvector (replaces the temp-file
// backed RankArray of support.h:576-638).
struct RankArray
{
  RankArray() : handle(nullptr), a(nullptr), b(nullptr) {}
  ~RankArray() { clear(); }
  RankArray(const RankArray&) = delete;
  RankArray& operator=(const RankArray&) = delete;
  void clear()
  {
    if(handle) { bwtm_ra_free(handle); handle = nullptr; }
    if(a) { bwtm_index_free(a); a = nullptr; }
    if(b) { bwtm_index_free(b); b = nullptr; }
  }
  bwtm_ra*    handle;
  bwtm_index* a;       // device copies of the inputs the array was built for
  bwtm_index* b;
};

class BWT
{
public:
  typedef bwtmerge::size_type size_type;
  const static size_type SAMPLE_RATE = Run::BLOCK_SIZE;
  const static size_type SIGMA       = Run::SIGMA;

  typedef std::array<size_type, SIGMA>  ranks_type;
  typedef std::array<range_type, SIGMA> rank_ranges_type;

  BWT() {}

  // Interleaves a and b according to ra; all inputs are destroyed (bwt.h:69-73).
  BWT(BWT& a, BWT& b, RankArray& ra)
  {
    a.destroy(); b.destroy();
    gpuCheck(bwtm_ra_finalize(ra.handle), "BWT::BWT()");
    bwtm_index* merged = nullptr;
    gpuCheck(bwtm_interleave(ra.a, ra.b, ra.handle, &merged), "BWT::BWT()");
    gpuCheck(bwtm_index_encode(merged), "BWT::BWT()");
    this->header.sequences = a.sequences() + b.sequences();
    this->header.bases = a.size() + b.size();
    this->header.setOrder(a.header.order());
    this->download(merged);
    bwtm_index_free(merged);
    a.data.clear(); b.data.clear(); ra.clear();
  }

  void swap(BWT& other)
  {
    std::swap(header, other.header); data.swap(other.data); block_end.swap(other.block_end);
    for(size_type c = 0; c < SIGMA; c++) { cumulative[c].swap(other.cumulative[c]); }
  }

  size_type size() const { return header.bases; }
  size_type sequences() const { return header.sequences; }
  size_type bytes() const { return data.size(); }
  size_type blocks() const { return block_end.size(); }
  size_type count(comp_type c) const { return cumulative[c].empty() ? 0 : cumulative[c].back(); }

  // Number of occurrences of c in [0, i).
  size_type rank(size_type i, comp_type c) const
  {
    if(c >= SIGMA) { return 0; }
    if(i > size()) { i = size(); }
    Cursor cur = seek(i);
    size_type result = cumulative[c][cur.block];
    while(cur.seq_pos < i)
    {
      range_type run = Run::read(data, cur.rle_pos);
      size_type take = std::min(run.second, i - cur.seq_pos);
      if(run.first == c) { result += take; }
      cur.seq_pos += run.second;
    }
    return result;
  }

  // rank(i, c) for c = 1 .. SIGMA - 1.
  void ranks(size_type i, ranks_type& results) const
  {
    if(i > size()) { i = size(); }
    Cursor cur = seek(i);
    for(size_type c = 1; c < SIGMA; c++) { results[c] = cumulative[c][cur.block]; }
    while(cur.seq_pos < i)
    {
      range_type run = Run::read(data, cur.rle_pos);
      results[run.first] += std::min(run.second, i - cur.seq_pos);
      cur.seq_pos += run.second;
    }
  }

  // (rank(range.first, c), rank(range.second + 1, c)) for c = 1 .. SIGMA - 1.
  void ranks(range_type range, rank_ranges_type& results) const
  {
    ranks_type sp, ep;
    ranks(std::min(range.first, size()), sp); ranks(std::min(range.second + 1, size()), ep);
    for(size_type c = 1; c < SIGMA; c++) { results[c] = range_type(sp[c], ep[c]); }
  }

  // (rank(i, BWT[i]), BWT[i])
  range_type inverse_select(size_type i) const
  {
    if(i >= size()) { return range_type(0, 0); }
    Cursor cur = seek(i);
    size_type local[SIGMA] = {};
    while(true)
    {
      range_type run = Run::read(data, cur.rle_pos);
      if(cur.seq_pos + run.second > i)
      {
        return range_type(cumulative[run.first][cur.block] + local[run.first] + (i - cur.seq_pos), run.first);
      }
      local[run.first] += run.second; cur.seq_pos += run.second;
    }
  }

  // Position of the i-th occurrence (1-based) of c; size() if there is none.
  size_type select(size_type i, comp_type c) const
  {
    if(c >= SIGMA || i == 0) { return 0; }
    if(i > count(c)) { return size(); }
    size_type block = (size_type)(std::lower_bound(cumulative[c].begin(), cumulative[c].end(), i) - cumulative[c].begin()) - 1;
    size_type seen = cumulative[c][block], rle_pos = block * SAMPLE_RATE, seq_pos = block_start(block);
    while(true)
    {
      range_type run = Run::read(data, rle_pos);
      if(run.first == c)
      {
        if(seen + run.second >= i) { return seq_pos + (i - seen - 1); }
        seen += run.second;
      }
      seq_pos += run.second;
    }
  }

  comp_type operator[](size_type i) const
  {
    if(i >= size()) { return 0; }
    Cursor cur = seek(i);
    while(true)
    {
      range_type run = Run::read(data, cur.rle_pos);
      cur.seq_pos += run.second;
      if(cur.seq_pos > i) { return (comp_type)run.first; }
    }
  }

  template<class ByteVector>
  void extract(range_type range, ByteVector& buffer) const
  {
    if(Range::empty(range) || range.second >= size()) { return; }
    buffer.resize(Range::length(range));
    Cursor cur = seek(range.first);
    size_type out = 0, want = Range::length(range);
    while(out < want)
    {
      range_type run = Run::read(data, cur.rle_pos);
      size_type begin = std::max(cur.seq_pos, range.first), end = std::min(cur.seq_pos + run.second, range.second + 1);
      for(size_type k = begin; k < end; k++) { buffer[out++] = (comp_type)run.first; }
      cur.seq_pos += run.second;
    }
  }

  void characterCounts(std::vector<size_type>& counts) const
  {
    counts.assign(SIGMA, 0);
    for(size_type rle_pos = 0; rle_pos < bytes(); ) { range_type run = Run::read(data, rle_pos); counts[run.first] += run.second; }
  }

  // FNV-1a over the decoded sequence, one byte per position.
  size_type hash() const
  {
    size_type h = FNV_OFFSET_BASIS;
    for(size_type rle_pos = 0; rle_pos < bytes(); )
    {
      range_type run = Run::read(data, rle_pos);
      for(size_type k = 0; k < run.second; k++) { h = fnv1a_hash((byte_type)run.first, h); }
    }
    return h;
  }

  // Builds the samples from the data (BWT::build) and fills the header from the counts.
  void buildFromData(AlphabeticOrder order = AO_DEFAULT)
  {
    block_end.clear();
    for(size_type c = 0; c < SIGMA; c++) { cumulative[c].assign(1, 0); }
    size_type seq_pos = 0, rle_pos = 0, totals[SIGMA] = {};
    while(rle_pos < bytes())
    {
      range_type run = Run::read(data, rle_pos);
      seq_pos += run.second; totals[run.first] += run.second;
      if(rle_pos >= bytes() || rle_pos % SAMPLE_RATE == 0)
      {
        block_end.push_back(seq_pos - 1);
        for(size_type c = 0; c < SIGMA; c++) { cumulative[c].push_back(totals[c]); }
      }
    }
    header.sequences = totals[0]; header.bases = seq_pos; header.setOrder(order);
  }

  void destroy()
  {
    block_end.clear(); block_end.shrink_to_fit();
    for(size_type c = 0; c < SIGMA; c++) { cumulative[c].clear(); cumulative[c].shrink_to_fit(); }
  }

  // Uploads the data to the device and returns the handle (the caller frees it).
  bwtm_index* upload(const std::vector<size_type>& C) const
  {
    bwtm_index* ix = nullptr;
    uint64_t c_array[BWTM_SIGMA + 1];
    for(size_type c = 0; c <= SIGMA; c++) { c_array[c] = C[c]; }
    gpuCheck(bwtm_index_upload(data.data(), data.size(), sequences(), size(), c_array, &ix), "BWT::upload()");
    return ix;
  }

  // Copies data and samples of an encoded device index into this object.
  void download(bwtm_index* ix)
  {
    data.bytes.resize(bwtm_index_bytes(ix));
    gpuCheck(bwtm_index_download_data(ix, data.bytes.data(), data.bytes.size()), "BWT::download()");
    size_type nblocks = bwtm_index_blocks(ix);
    block_end.resize(nblocks);
    std::vector<uint64_t> cum(SIGMA * (nblocks + 1));
    gpuCheck(bwtm_index_download_samples(ix, block_end.data(), cum.data()), "BWT::download()");
    for(size_type c = 0; c < SIGMA; c++) { cumulative[c].assign(cum.begin() + c * (nblocks + 1), cum.begin() + (c + 1) * (nblocks + 1)); }
  }

  // Native format (reference bwt.cpp:111-148; layout SURVEY.md Appendix B).
  void serialize(std::ostream& out) const
  {
    header.serialize(out);
    size_type nbytes = data.size();
    sdsl_compat::write_member(nbytes, out);
    out.write((const char*)data.data(), nbytes);
    size_type padded = data.blocks() * BlockArray::BLOCK_SIZE;
    std::vector<char> zeros(std::min(padded - nbytes, (size_type)1 << 20), 0);
    for(size_type left = padded - nbytes; left > 0; ) { size_type n = std::min(left, (size_type)zeros.size()); out.write(zeros.data(), n); left -= n; }
    size_type nblocks = blocks();
    for(size_type c = 0; c < SIGMA; c++)
    {
      // element k = (count of c in block k) zero bits followed by a one bit (support.h:290-294)
      std::vector<size_type> ones(nblocks);
      for(size_type k = 0; k < nblocks; k++) { ones[k] = cumulative[c][k + 1] + k; }
      sdsl_compat::SDVector::serialize(out, count(c) + nblocks, ones);
      sdsl_compat::write_member(nblocks, out);                     // CumulativeArray::m_size
    }
    sdsl_compat::SDVector::serialize(out, size(), block_end);
  }

  void load(std::istream& in)
  {
    header.load(in);
    if(!header.check()) { std::cerr << "BWT::load(): Invalid header!" << std::endl; std::exit(EXIT_FAILURE); }
    size_type nbytes = 0; sdsl_compat::read_member(nbytes, in);
    data.bytes.resize(nbytes);
    in.read((char*)data.bytes.data(), nbytes);
    in.ignore(data.blocks() * BlockArray::BLOCK_SIZE - nbytes);
    for(size_type c = 0; c < SIGMA; c++)
    {
      size_type universe = 0, m_size = 0; std::vector<size_type> ones;
      sdsl_compat::SDVector::load(in, universe, ones);
      sdsl_compat::read_member(m_size, in);
      cumulative[c].assign(ones.size() + 1, 0);
      for(size_type k = 0; k < ones.size(); k++) { cumulative[c][k + 1] = ones[k] - k; }
    }
    size_type universe = 0;
    sdsl_compat::SDVector::load(in, universe, block_end);
  }

  NativeHeader           header;
  BlockArray             data;
  std::vector<size_type> block_end;            // last sequence position of each block (block_boundaries)
  std::vector<size_type> cumulative[SIGMA];    // cumulative[c][k] = #c in blocks [0, k) (samples[c])

private:
  struct Cursor { size_type block, rle_pos, seq_pos; };

  size_type block_start(size_type block) const { return (block == 0 ? 0 : block_end[block - 1] + 1); }

  // The block that holds position i (or the one after the last when i == size()).
  Cursor seek(size_type i) const
  {
    Cursor cur;
    cur.block = (size_type)(std::lower_bound(block_end.begin(), block_end.end(), i) - block_end.begin());
    cur.rle_pos = cur.block * SAMPLE_RATE; cur.seq_pos = block_start(cur.block);
    return cur;
  }
};

} // namespace bwtmerge

#endif // BWTM_HOST_BWT_H
