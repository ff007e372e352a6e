/*
  bwt.h -- host facade: class BWT with the reference's public surface (bwt.h:41-189).  The
  run-length data and its samples live on the host as flat page-locked arrays (what block_boundaries
  and samples[c] encode, bwt.h:172-178); scalar queries run on them.  The interleaving constructor
  BWT(a, b, ra) (bwt.h:73, bwt.cpp:286-314) runs on the GPU through the C ABI (include/bwtm.h).

  A BWT may also own a DEVICE COPY (the rank structure of the same sequence): the result of a merge
  stays on the device as the next merge's first input (bwt_merge.cpp:167-173), and with
  MergeParameters::lazy_host the host form is produced only when something asks for it
  (materialize(): encode + one download), so `bwt_merge a b c d out` uploads every input once and
  downloads once.
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

// The library's pool takes the addresses of its large blocks from reserved ranges that are consumed and never reused (DESIGN.md section 2;
// INTEGRATION.md "How many merges a process can run").  When they are used up -- after the order of two hundred merges of 2 x 50 Gbase in one
// process -- large blocks come from hipMalloc, which is correct but can stall for seconds: said ONCE on stderr, at the merge that notices it.
inline void warnIfPoolExhausted(const char* where)
{
  static bool warned = false;
  bwtm_pool_info info;
  if(warned || bwtm_pool_stats(&info) != BWTM_OK || !info.address_space_exhausted) { return; }
  warned = true;
  std::cerr << where << ": the device memory pool has used up its address ranges (" << info.address_bytes_reserved << " bytes reserved, "
            << info.hipmalloc_fallbacks << " large blocks served by hipMalloc so far): further merges of this process stay correct but may be slow; "
            << "restart the process to get the mapped pool back" << std::endl;
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
    a = nullptr; b = nullptr;
  }
  bwtm_ra*    handle;
  bwtm_index* a;       // device copies of the inputs the array was built for (owned by the two BWT objects)
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
  ~BWT() { dropDevice(); }
  // A copy shares nothing with the original: the host form is copied, the device copy is not.
  BWT(const BWT& other) : header(other.header), data(other.hostData()), block_end(other.block_end), cum_flat(other.cum_flat), cum_stride(other.cum_stride),
    sample_width(other.sample_width), fields(other.fields), anchors(other.anchors) {}
  BWT(BWT&& other) noexcept { swap(other); }
  BWT& operator=(const BWT& other) { if(this != &other) { BWT copy(other); swap(copy); } return *this; }
  BWT& operator=(BWT&& other) noexcept { if(this != &other) { BWT moved(std::move(other)); swap(moved); } return *this; }

  // Interleaves a and b according to ra; all inputs are destroyed (bwt.h:69-73).
  BWT(BWT& a, BWT& b, RankArray& ra)
  {
    gpuCheck(bwtm_ra_finalize(ra.handle), "BWT::BWT()");
    bwtm_index* merged = nullptr;
    gpuCheck(bwtm_interleave(ra.a, ra.b, ra.handle, &merged), "BWT::BWT()");
    this->header.sequences = a.sequences() + b.sequences();
    this->header.bases = a.size() + b.size();
    this->header.setOrder(a.header.order());
    a.clear(); b.clear(); ra.clear();                      // the device copies of the inputs go before the encoder needs memory
    this->adopt(merged);
    this->materialize();
  }

  void swap(BWT& other)
  {
    std::swap(header, other.header); data.swap(other.data); block_end.swap(other.block_end);
    cum_flat.swap(other.cum_flat); std::swap(cum_stride, other.cum_stride);
    std::swap(sample_width, other.sample_width); fields.swap(other.fields); anchors.swap(other.anchors);
    std::swap(device, other.device); std::swap(pending, other.pending); std::swap(host_current, other.host_current);
  }

  size_type size() const { return header.bases; }
  size_type sequences() const { return header.sequences; }
  size_type bytes() const { materialize(); return data.size(); }
  size_type blocks() const { materialize(); return cum_stride == 0 ? 0 : cum_stride - 1; }
  size_type count(comp_type c) const { materialize(); return cum_stride == 0 ? 0 : cum(c, cum_stride - 1); }

  // The samples exist in one of two forms.  FULL (sample_width == 8): block_end[] and cum_flat[6][blocks + 1], the values
  // block_boundaries and samples[c] hold.  COMPACT (2 or 4; what a merge downloads: 12 or 24 instead of 56 bytes per block):
  // FIELDS [6][blocks] = positions in block k and its occurrences of 1..5, ANCHORS [6][blocks / 64] = start position and
  // counts before every 64th block (include/bwtm.h).  The accessors below hide the difference; expandSamples() converts.
  size_type field(size_type c, size_type k) const
  {
    const size_type at = c * (cum_stride - 1) + k;
    if(sample_width == 1) { return (size_type)((const std::uint8_t*)fields.data())[at]; }
    return (sample_width == 2 ? (size_type)fields[at] : (size_type)((const std::uint32_t*)fields.data())[at]);
  }
  // start position (c == 0) / occurrences of c (1..5) before block k, k <= blocks, compact form
  size_type compactAt(size_type c, size_type k) const
  {
    const size_type nb = cum_stride - 1, nanch = (nb + 63) / 64;
    size_type g = k / 64, from = g * 64;
    if(g >= nanch) { g = nanch - 1; from = g * 64; }          // k == blocks on a group boundary: walk the last group
    size_type v = anchors[c * nanch + g];
    for(size_type j = from; j < k; j++) { v += field(c, j); }
    return v;
  }
  size_type blockStart(size_type k) const
  {
    if(sample_width == 8) { return (k == 0 ? 0 : block_end[k - 1] + 1); }
    return (cum_stride <= 1 ? 0 : compactAt(0, k));
  }
  size_type blockEnd(size_type k) const { return (sample_width == 8 ? block_end[k] : blockStart(k + 1) - 1); }

  // samples[c].sum(k): occurrences of c in blocks [0, k)  (CumulativeArray::sum, support.h:338-343)
  size_type cum(size_type c, size_type k) const
  {
    if(sample_width == 8) { return cum_flat[c * cum_stride + k]; }
    if(cum_stride <= 1) { return 0; }
    if(c != 0) { return compactAt(c, k); }
    size_type rest = 0;
    for(size_type s = 1; s < SIGMA; s++) { rest += compactAt(s, k); }
    return compactAt(0, k) - rest;
  }
  // The six sample arrays as vectors (tests, serialization).
  std::vector<size_type> cumulative(size_type c) const
  {
    materialize();
    std::vector<size_type> row(cum_stride);
    for(size_type k = 0; k < cum_stride; k++) { row[k] = cum(c, k); }
    return row;
  }
  std::vector<size_type> blockEnds() const
  {
    materialize();
    std::vector<size_type> ends(blocks());
    for(size_type k = 0; k < ends.size(); k++) { ends[k] = blockEnd(k); }
    return ends;
  }

  // Compact -> full form (one pass; the queries below work on either).
  void expandSamples() const
  {
    materialize();
    if(sample_width == 8) { return; }
    const size_type nb = cum_stride - 1, nanch = (nb + 63) / 64;
    HostArray<size_type> ends(nb), flat(SIGMA * cum_stride);
    size_type run[SIGMA] = {};
    for(size_type k = 0; k <= nb; k++)
    {
      if(k < nb && k % 64 == 0) { for(size_type c = 0; c < SIGMA; c++) { run[c] = anchors[c * nanch + k / 64]; } }
      size_type rest = 0;
      for(size_type c = 1; c < SIGMA; c++) { flat[c * cum_stride + k] = run[c]; rest += run[c]; }
      flat[k] = run[0] - rest;                                            // row 0: endmarkers = start - the five
      if(k > 0) { ends[k - 1] = run[0] - 1; }
      if(k < nb) { for(size_type c = 0; c < SIGMA; c++) { run[c] += field(c, k); } }
    }
    if(nb == 0) { for(size_type c = 0; c < SIGMA; c++) { flat[c * cum_stride] = 0; } }
    block_end.swap(ends); cum_flat.swap(flat);
    fields.clear(); fields.shrink_to_fit(); anchors.clear(); anchors.shrink_to_fit();
    sample_width = 8;
  }

  // Number of occurrences of c in [0, i).
  size_type rank(size_type i, comp_type c) const
  {
    if(c >= SIGMA) { return 0; }
    if(i > size()) { i = size(); }
    Cursor cur = seek(i);
    size_type result = cum(c, cur.block);
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
    for(size_type c = 1; c < SIGMA; c++) { results[c] = cum(c, cur.block); }
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
        return range_type(cum(run.first, cur.block) + local[run.first] + (i - cur.seq_pos), run.first);
      }
      local[run.first] += run.second; cur.seq_pos += run.second;
    }
  }

  // Position of the i-th occurrence (1-based) of c; size() if there is none.
  size_type select(size_type i, comp_type c) const
  {
    if(c >= SIGMA || i == 0) { return 0; }
    if(i > count(c)) { return size(); }
    // the last block k with cum(c, k) < i
    size_type lo = 0, hi = cum_stride - 1;                   // cum(c, lo) < i <= cum(c, hi)
    while(hi - lo > 1) { size_type mid = (lo + hi) / 2; if(cum(c, mid) < i) { lo = mid; } else { hi = mid; } }
    size_type block = lo;
    size_type seen = cum(c, block), rle_pos = block * SAMPLE_RATE, seq_pos = blockStart(block);
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
    materialize();
    counts.assign(SIGMA, 0);
    for(size_type rle_pos = 0; rle_pos < data.size(); ) { range_type run = Run::read(data, rle_pos); counts[run.first] += run.second; }
  }

  // FNV-1a over the decoded sequence, one byte per position.
  size_type hash() const
  {
    materialize();
    size_type h = FNV_OFFSET_BASIS;
    for(size_type rle_pos = 0; rle_pos < data.size(); )
    {
      range_type run = Run::read(data, rle_pos);
      for(size_type k = 0; k < run.second; k++) { h = fnv1a_hash((byte_type)run.first, h); }
    }
    return h;
  }

  // Builds the samples from the data (BWT::build) and fills the header from the counts.
  void buildFromData(AlphabeticOrder order = AO_DEFAULT)
  {
    dropDevice(); host_current = true; sample_width = 8;
    std::vector<size_type> ends;
    std::vector<size_type> rows[SIGMA];
    for(size_type c = 0; c < SIGMA; c++) { rows[c].assign(1, 0); }
    size_type seq_pos = 0, rle_pos = 0, totals[SIGMA] = {};
    while(rle_pos < data.size())
    {
      range_type run = Run::read(data, rle_pos);
      seq_pos += run.second; totals[run.first] += run.second;
      if(rle_pos >= data.size() || rle_pos % SAMPLE_RATE == 0)
      {
        ends.push_back(seq_pos - 1);
        for(size_type c = 0; c < SIGMA; c++) { rows[c].push_back(totals[c]); }
      }
    }
    block_end.assign(ends.begin(), ends.end());
    cum_stride = ends.size() + 1;
    cum_flat.resizeUninitialized(SIGMA * cum_stride);
    for(size_type c = 0; c < SIGMA; c++) { std::copy(rows[c].begin(), rows[c].end(), cum_flat.begin() + c * cum_stride); }
    header.sequences = totals[0]; header.bases = seq_pos; header.setOrder(order);
  }

  // Drops the samples (the reference's BWT::destroy, bwt.cpp:514-521).
  void destroy()
  {
    block_end.clear(); block_end.shrink_to_fit();
    cum_flat.clear(); cum_flat.shrink_to_fit(); cum_stride = 0;
    fields.clear(); fields.shrink_to_fit(); anchors.clear(); anchors.shrink_to_fit();
    sample_width = 8;
  }

  // Drops everything: host bytes, samples, the device copy.
  void clear() { destroy(); data.clear(); dropDevice(); host_current = true; }

  //--------------------------------------------------------------------------
  // Device side.

  // The device copy of this BWT (uploaded on first use and kept: a chained merge, bwt_merge.cpp:167-173, or a
  // verification after a merge finds it there).
  bwtm_index* onDevice(const std::vector<size_type>& C) const
  {
    if(!device && pending)
    {
      bwtm_upload* u = pending; pending = nullptr;
      gpuCheck(bwtm_upload_finish(u, &device), "BWT::onDevice()");           // announced by prefetchDevice(): the bytes are (nearly) there
      gpuCheck(bwtm_index_drop_native(device), "BWT::onDevice()");
    }
    if(!device)
    {
      uint64_t c_array[BWTM_SIGMA + 1];
      for(size_type c = 0; c <= SIGMA; c++) { c_array[c] = C[c]; }
      gpuCheck(bwtm_index_upload(data.data(), data.size(), sequences(), size(), c_array, &device), "BWT::onDevice()");
      gpuCheck(bwtm_index_drop_native(device), "BWT::onDevice()");          // the host holds the bytes
    }
    return device;
  }
  // Announces that this BWT will be needed on the device soon: its bytes start travelling now (copy stream) and the call
  // returns at once -- bwt_merge announces input k + 1 before it merges input k, so the copy runs under that merge's search.
  void prefetchDevice(const std::vector<size_type>& C) const
  {
    if(device || pending) { return; }
    materialize();
    uint64_t c_array[BWTM_SIGMA + 1];
    for(size_type c = 0; c <= SIGMA; c++) { c_array[c] = C[c]; }
    bwtm_host_input in = { data.data(), data.size(), sequences(), size(), c_array };      // the library keeps its own copy of C
    gpuCheck(bwtm_upload_begin(&in, &pending), "BWT::prefetchDevice()");
  }
  // Hands the announced upload to the caller (bwtm_merge_host_pipelined consumes it); nullptr if there is none.
  bwtm_upload* releasePending() { bwtm_upload* u = pending; pending = nullptr; return u; }
  // Hands the device copy to the caller (who frees or consumes it); nullptr if there is none.
  bwtm_index* releaseDevice() { bwtm_index* d = device; device = nullptr; return d; }
  bool deviceResident() const { return device != nullptr; }
  void dropDevice() const
  {
    if(pending) { bwtm_upload_free(pending); pending = nullptr; }           // waits for the queued copies: they read this object's bytes
    if(device) { bwtm_index_free(device); device = nullptr; }
  }

  // Takes over a merged device index; the host form is produced by materialize() when somebody asks for it.
  void adopt(bwtm_index* merged)
  {
    dropDevice();
    device = merged; host_current = false;
    data.clear(); destroy();
  }

  // Takes over the host form a bwtm_merge_host call has just written into this object's arrays.
  void adoptHost(bwtm_index* kept, size_type nblocks, int width = 8)
  {
    dropDevice();
    device = kept; host_current = true; cum_stride = nblocks + 1; sample_width = width;
  }

  // Makes the host form current: encodes on the device, downloads data and samples (once).
  void materialize() const
  {
    if(host_current) { return; }
    gpuCheck(bwtm_index_encode(device), "BWT::materialize()");
    data.bytes.resizeUninitialized(bwtm_index_bytes(device));
    gpuCheck(bwtm_index_download_data(device, data.bytes.data(), data.bytes.size()), "BWT::materialize()");
    size_type nblocks = bwtm_index_blocks(device);
    cum_stride = nblocks + 1;
    int width = 8;
    gpuCheck(bwtm_index_samples_width(device, &width), "BWT::materialize()");
    sample_width = width;
    if(width == 8)
    {
      block_end.resizeUninitialized(nblocks);
      cum_flat.resizeUninitialized(SIGMA * cum_stride);
      gpuCheck(bwtm_index_download_samples(device, block_end.data(), cum_flat.data()), "BWT::materialize()");
    }
    else
    {
      anchors.resizeUninitialized(SIGMA * ((nblocks + 63) / 64));
      fields.resizeUninitialized((SIGMA * nblocks * (size_type)width + 1) / 2);
      gpuCheck(bwtm_index_download_samples_compact(device, width, fields.data(), anchors.data()), "BWT::materialize()");
    }
    gpuCheck(bwtm_index_drop_native(device), "BWT::materialize()");         // keep only the rank structure on the device
    host_current = true;
  }

  const BlockArray& hostData() const { materialize(); return data; }

  // Native format (reference bwt.cpp:111-148; layout SURVEY.md Appendix B).
  void serialize(std::ostream& out) const
  {
    materialize();
    header.serialize(out);
    size_type nbytes = data.size();
    sdsl_compat::write_member(nbytes, out);
    out.write((const char*)data.data(), nbytes);
    size_type padded = data.blocks() * BlockArray::BLOCK_SIZE;
    std::vector<char> zeros(std::min(padded - nbytes, (size_type)1 << 20), 0);
    for(size_type left = padded - nbytes; left > 0; ) { size_type n = std::min(left, (size_type)zeros.size()); out.write(zeros.data(), n); left -= n; }
    expandSamples();
    size_type nblocks = blocks();
    for(size_type c = 0; c < SIGMA; c++)
    {
      // element k = (count of c in block k) zero bits followed by a one bit (support.h:290-294)
      std::vector<size_type> ones(nblocks);
      for(size_type k = 0; k < nblocks; k++) { ones[k] = cum(c, k + 1) + k; }
      sdsl_compat::SDVector::serialize(out, count(c) + nblocks, ones);
      sdsl_compat::write_member(nblocks, out);                     // CumulativeArray::m_size
    }
    sdsl_compat::SDVector::serialize(out, size(), std::vector<size_type>(block_end.begin(), block_end.end()));
  }

  void load(std::istream& in)
  {
    dropDevice(); host_current = true; sample_width = 8;
    header.load(in);
    if(!header.check()) { std::cerr << "BWT::load(): Invalid header!" << std::endl; std::exit(EXIT_FAILURE); }
    size_type nbytes = 0; sdsl_compat::read_member(nbytes, in);
    data.bytes.resizeUninitialized(nbytes);
    in.read((char*)data.bytes.data(), nbytes);
    in.ignore(data.blocks() * BlockArray::BLOCK_SIZE - nbytes);
    std::vector<size_type> rows[SIGMA];
    for(size_type c = 0; c < SIGMA; c++)
    {
      size_type universe = 0, m_size = 0; std::vector<size_type> ones;
      sdsl_compat::SDVector::load(in, universe, ones);
      sdsl_compat::read_member(m_size, in);
      rows[c].assign(ones.size() + 1, 0);
      for(size_type k = 0; k < ones.size(); k++) { rows[c][k + 1] = ones[k] - k; }
    }
    cum_stride = rows[0].size();
    cum_flat.resizeUninitialized(SIGMA * cum_stride);
    for(size_type c = 0; c < SIGMA; c++) { std::copy(rows[c].begin(), rows[c].end(), cum_flat.begin() + c * cum_stride); }
    size_type universe = 0; std::vector<size_type> ends;
    sdsl_compat::SDVector::load(in, universe, ends);
    block_end.assign(ends.begin(), ends.end());
  }

  NativeHeader                   header;
  mutable BlockArray             data;                 // valid after materialize() (always, unless the BWT came out of a lazy merge)
  mutable HostArray<size_type>   block_end;            // last sequence position of each block (block_boundaries)
  mutable HostArray<size_type>   cum_flat;             // [SIGMA][cum_stride]: cum(c, k) = #c in blocks [0, k) (samples[c])
  mutable size_type              cum_stride = 0;       // blocks + 1
  mutable int                    sample_width = 8;     // 8: block_end + cum_flat; 1 / 2 / 4: 8- / 16- / 32-bit fields + anchors
  mutable HostArray<std::uint16_t> fields;             // raw storage of the fields, 16 or 32 bits each (field())
  mutable HostArray<size_type>   anchors;

private:
  mutable bwtm_index* device = nullptr;                // device copy (rank structure only), owned
  mutable bwtm_upload* pending = nullptr;              // an announced upload of `data` (prefetchDevice), owned
  mutable bool        host_current = true;             // false: only the device holds the BWT (lazy result of a merge)

  struct Cursor { size_type block, rle_pos, seq_pos; };

  // The block that holds position i (or the one after the last when i == size()).
  Cursor seek(size_type i) const
  {
    materialize();
    Cursor cur;
    if(sample_width == 8)
    {
      cur.block = (size_type)(std::lower_bound(block_end.begin(), block_end.end(), i) - block_end.begin());
      cur.seq_pos = (cur.block == 0 ? 0 : block_end[cur.block - 1] + 1);
    }
    else
    {
      // the group whose anchor is the last one <= i, then a walk over at most 64 fields
      const size_type nb = cum_stride - 1, nanch = (nb + 63) / 64;
      size_type g = (nanch == 0 ? 0 : (size_type)(std::upper_bound(anchors.begin(), anchors.begin() + nanch, i) - anchors.begin()));
      if(g > 0) { g--; }
      size_type k = g * 64, start = (nanch == 0 ? 0 : anchors[g]);
      while(k < nb && start + field(0, k) <= i) { start += field(0, k); k++; }
      cur.block = k; cur.seq_pos = start;
    }
    cur.rle_pos = cur.block * SAMPLE_RATE;
    return cur;
  }
};

} // namespace bwtmerge

#endif // BWTM_HOST_BWT_H
