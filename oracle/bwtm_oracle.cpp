/*
  bwtm_oracle.cpp -- CPU ORACLE for the rank-array / interleave path of bwt-merge.

  THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
  and bench.py's cpu_baseline leg may load it.  The product path (bwt-merge_amd/) never
  links, imports or executes anything under oracle/.

  What it is: a plain C++17 restatement (no SDSL, no GPU) of the reference algorithm for the
  path FMI::FMI(a, b, params) -> BWT::BWT(a, b, ra), each function citing the reference
  file:line it follows (paths relative to the reference checkout), plus an independent
  brute-force ground truth (suffix-sorted multi-string BWT of the concatenated collection).

  Pinning status: the reference itself is UNBUILDABLE in this image (utils.h:37 includes
  <sdsl/wavelet_trees.hpp>; SDSL is neither vendored nor installed), and it ships no tests
  or golden vectors.  The oracle is therefore pinned against
    (1) the known-answer vectors recorded from the reference's own code in SURVEY.md
        Appendix C (tests/golden/reference_vectors.json), and
    (2) the identity merge(A,B) == BWT(A || B) checked against the brute-force builder.
  The serialized bytes of SDSL containers are NOT pinned ("parity unpinned" at the SDSL
  byte level); everything the hot path computes (data bytes, header, counts, samples, C)
  is.

  SDSL containers are replaced by what they mathematically are:
    sd_vector block_boundaries + rank/select  -> sorted vector of block-end positions
    CumulativeArray (sd_vector prefix sums)   -> plain prefix-sum vectors
    BlockArray (8 MiB mmap blocks)            -> std::vector<uint8_t>
    int_vector_buffer<8> temp files           -> in-memory byte vectors (same byte stream)
*/

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stack>
#include <thread>
#include <utility>
#include <vector>

namespace orc
{

typedef std::uint64_t u64;
typedef std::uint8_t  u8;
typedef std::pair<u64, u64> range_t;   // closed range [first, second]; utils.h:71

static const u64 SIGMA        = 6;     // support.h:228
static const u64 RLE_BLOCK    = 64;    // support.h:227
static const u64 MAX_RUN      = 42;    // support.h:229  (256 / 6)
static const u64 SHORT_RANGE  = 256;   // fmi.h:91
static const u64 RA_CHUNK     = 1u << 20;  // bwt.cpp:156 (RABuffer::BUFFER_SIZE, runs)

//------------------------------------------------------------------------------
// Range helpers (utils.h:73-99)

static inline u64  range_length(range_t r) { return r.second + 1 - r.first; }
static inline bool range_empty(range_t r)  { return r.first + 1 > r.second + 1; }

//------------------------------------------------------------------------------
// RunBuffer (utils.h:121-142): streaming maximal-run coalescer.

struct RunBuffer
{
  u64 value = 0, length = 0;
  range_t run = range_t(0, 0);

  // Returns true when a previous non-empty run has just been completed into `run`.
  bool add(u64 v, u64 n = 1)
  {
    if(v == value) { length += n; return false; }
    flush();
    value = v; length = n;
    return run.second > 0;
  }
  bool add(range_t r) { return add(r.first, r.second); }
  void flush() { run.first = value; run.second = length; }
};

//------------------------------------------------------------------------------
// bit_length (utils.h:146-151; sdsl::bits::hi(x) + 1, with hi(0) == 0)

static inline u64 bit_length(u64 x)
{
  return (x == 0 ? 1 : 64 - (u64)__builtin_clzll(x));
}

//------------------------------------------------------------------------------
// ByteCode: 7+1 bit LSB-first varint (support.h:160-213)

static inline u64 bytecode_read(const u8* a, u64& i)
{
  u64 shift = 0, res = a[i] & 0x7F;
  while(a[i] & 0x80) { i++; shift += 7; res += ((u64)(a[i] & 0x7F)) << shift; }
  i++;
  return res;
}

static inline void bytecode_write(std::vector<u8>& a, u64 v)
{
  while(v > 0x7F) { a.push_back((u8)((v & 0x7F) | 0x80)); v >>= 7; }
  a.push_back((u8)v);
}

//------------------------------------------------------------------------------
// Run codec (support.h:221-286)

static inline range_t run_read(const u8* a, u64& i)
{
  u8 code = a[i]; i++;
  range_t run(code % SIGMA, code / SIGMA + 1);        // decodeBasic, support.h:236-239
  if(run.second >= MAX_RUN) { run.second += bytecode_read(a, i); }   // support.h:248
  return run;
}

// `virtual_size` lets tests emulate an array that already holds that many bytes.
static inline void run_write(std::vector<u8>& a, u64 comp, u64 length, u64 virtual_size = 0)
{
  while(length > 0)                                   // support.h:259
  {
    if(length < MAX_RUN) { a.push_back((u8)(comp + SIGMA * (length - 1))); return; }
    u64 remaining = RLE_BLOCK - ((a.size() + virtual_size) % RLE_BLOCK);   // support.h:267
    u64 basic = (remaining > 1 ? MAX_RUN : MAX_RUN - 1);                   // support.h:268
    a.push_back((u8)(comp + SIGMA * (basic - 1))); length -= basic;
    remaining--;
    if(remaining > 0)                                 // support.h:272-280
    {
      u64 ext = length;
      if(bit_length(length) > 7 * remaining) { ext = (~(u64)0) >> (64 - 7 * remaining); }
      bytecode_write(a, ext); length -= ext;
    }
  }
}

//------------------------------------------------------------------------------
// FNV-1a (utils.h:155-161)

static const u64 FNV_OFFSET_BASIS = 0xcbf29ce484222325ULL;
static const u64 FNV_PRIME        = 0x100000001b3ULL;
static inline u64 fnv1a(u8 b, u64 seed) { return (seed ^ b) * FNV_PRIME; }

//------------------------------------------------------------------------------
// getBounds (utils.cpp:169-187)

static std::vector<range_t> get_bounds(range_t range, u64 blocks)
{
  if(range_empty(range)) { return std::vector<range_t>(); }
  blocks = std::max(std::min(blocks, range_length(range)), (u64)1);
  std::vector<range_t> bounds(blocks);
  u64 start = range.first;
  for(u64 b = 0; b < blocks; b++)
  {
    bounds[b].first = start;
    if(start <= range.second)
    {
      start += std::max((u64)1, (range.second + 1 - start) / (blocks - b));
    }
    bounds[b].second = start - 1;
  }
  return bounds;
}

//------------------------------------------------------------------------------
// BWT: run-length encoded sequence over 0..5 in 64-byte blocks + per-block samples
// (bwt.h:41-189, bwt.cpp).

struct BWT
{
  u64 sequences = 0, bases = 0;          // NativeHeader fields, formats.h:44-49
  std::vector<u8> data;                  // BlockArray
  std::vector<u64> block_end;            // block_boundaries: last sequence position of each block
  std::vector<u64> cum[SIGMA];           // samples[c]: cum[c][k] = #c in blocks [0, k)
  std::vector<u64> rank_hint;            // rank_hint[j] = block_rank(j << hint_shift): what makes block_rank O(1), as sd_vector's rank_1 is (see block_rank)
  u64 hint_shift = 6;                    // cells of 64 positions for ordinary streams; wider when runs are huge (a table of at most ~4 cells per block)

  u64 size() const { return bases; }
  u64 bytes() const { return data.size(); }
  u64 blocks() const { return block_end.size(); }

  // block_rank(i): number of block-end marks in [0, i)   (sd_vector rank_1).
  // Round 5: answered from a table of the ranks at every 2^hint_shift-th position (64 for ordinary streams: a full block written by
  // Run::write encodes at least 64 positions, so a cell holds a block end or two) plus a search inside the cell, the way SDSL's Elias-Fano
  // rank answers in constant time.  Rounds 1 - 4 ran a binary search over ALL block ends here -- ~20 dependent cache misses per rank query
  // that the reference does not pay, i.e. a port slower than what it restates.  Same values.
  u64 block_rank(u64 i) const
  {
    const u64 cell = std::min(i >> hint_shift, (u64)rank_hint.size() - 2);
    const u64 lo = rank_hint[cell], hi = rank_hint[cell + 1];            // the answer lies in [lo, hi] when i is inside the cell; a few entries
    if(hi - lo <= 8) { u64 k = lo; while(k < block_end.size() && block_end[k] < i) { k++; } return k; }
    return (u64)(std::lower_bound(block_end.begin() + lo, block_end.end(), i) - block_end.begin());
  }
  // block_select(k): position of the k-th mark, 1-based   (sd_vector select_1)
  u64 block_select(u64 k) const { return block_end[k - 1]; }
  // CumulativeArray::sum(k), support.h:338-343
  u64 sample_sum(u64 c, u64 k) const { return cum[c][std::min(k, blocks())]; }
  u64 count(u64 c) const { return cum[c].empty() ? 0 : cum[c].back(); }

  // bwt.cpp:476-512
  void build()
  {
    u64 nblocks = (bytes() + RLE_BLOCK - 1) / RLE_BLOCK;
    block_end.clear(); block_end.reserve(nblocks);
    for(u64 c = 0; c < SIGMA; c++) { cum[c].clear(); cum[c].reserve(nblocks + 1); cum[c].push_back(0); }
    u64 seq_pos = 0, rle_pos = 0;
    u64 cumulative[SIGMA] = {};
    while(rle_pos < bytes())
    {
      range_t run = run_read(data.data(), rle_pos);
      seq_pos += run.second; cumulative[run.first] += run.second;
      if(rle_pos >= bytes() || rle_pos % RLE_BLOCK == 0)
      {
        block_end.push_back(seq_pos - 1);
        for(u64 c = 0; c < SIGMA; c++) { cum[c].push_back(cumulative[c]); }
      }
    }
    build_hint(seq_pos);
  }

  void build_hint(u64 positions)
  {
    hint_shift = 6;
    while((positions >> hint_shift) > 4 * (u64)block_end.size() + 1024) { hint_shift++; }     // streams of huge runs: not one cell per 64 positions
    rank_hint.assign((positions >> hint_shift) + 3, 0);
    u64 k = 0;
    for(u64 j = 0; j < rank_hint.size(); j++)
    {
      while(k < block_end.size() && block_end[k] < (j << hint_shift)) { k++; }
      rank_hint[j] = k;
    }
  }

  // bwt.cpp:514-521
  void destroy()
  {
    block_end.clear(); block_end.shrink_to_fit();
    rank_hint.clear(); rank_hint.shrink_to_fit();
    for(u64 c = 0; c < SIGMA; c++) { cum[c].clear(); cum[c].shrink_to_fit(); }
  }

  void locate(u64 i, u64& block, u64& rle_pos, u64& seq_pos) const
  {
    block = block_rank(i);
    rle_pos = block * RLE_BLOCK;
    seq_pos = (block > 0 ? block_select(block) + 1 : 0);
  }

  // bwt.cpp:318-341
  u64 rank(u64 i, u64 c) const
  {
    if(c >= SIGMA) { return 0; }
    if(i > size()) { i = size(); }
    u64 block, rle_pos, seq_pos; locate(i, block, rle_pos, seq_pos);
    u64 res = sample_sum(c, block);
    while(seq_pos < i)
    {
      range_t run = run_read(data.data(), rle_pos);
      seq_pos += run.second;
      if(run.first == c)
      {
        res += run.second;
        if(seq_pos > i) { res -= seq_pos - i; }
      }
    }
    return res;
  }

  // bwt.cpp:343-361
  void ranks(u64 i, std::array<u64, SIGMA>& results) const
  {
    if(i > size()) { i = size(); }
    u64 block, rle_pos, seq_pos; locate(i, block, rle_pos, seq_pos);
    for(u64 c = 1; c < SIGMA; c++) { results[c] = sample_sum(c, block); }
    u64 prev = 0;
    while(seq_pos < i)
    {
      range_t run = run_read(data.data(), rle_pos);
      seq_pos += run.second;
      results[run.first] += run.second; prev = run.first;
    }
    results[prev] -= seq_pos - i;
  }

  // bwt.cpp:363-403
  void ranks(range_t range, std::array<range_t, SIGMA>& results) const
  {
    range.first = std::min(range.first, size() - 1);
    range.second = std::min(range.second, size() - 1);
    for(u64 c = 1; c < SIGMA; c++) { results[c] = range_t(0, 0); }
    u64 block, rle_pos, seq_pos; locate(range.first, block, rle_pos, seq_pos);

    range_t run(0, 0);
    while(seq_pos < range.first)
    {
      run = run_read(data.data(), rle_pos);
      seq_pos += run.second;
      results[run.first].first += run.second;
      results[run.first].second += run.second;
    }
    results[run.first].first -= seq_pos - range.first;

    while(seq_pos <= range.second)
    {
      run = run_read(data.data(), rle_pos);
      seq_pos += run.second;
      results[run.first].second += run.second;
    }
    results[run.first].second -= (seq_pos - 1) - range.second;

    for(u64 c = 1; c < SIGMA; c++)
    {
      if(results[c].second > results[c].first)
      {
        u64 temp = sample_sum(c, block);
        results[c].first += temp; results[c].second += temp;
      }
    }
  }

  // bwt.cpp:445-464: (rank of BWT[i] among equal symbols before i, BWT[i])
  range_t inverse_select(u64 i) const
  {
    range_t run(0, 0);
    if(i >= size()) { return run; }
    u64 block, rle_pos, seq_pos; locate(i, block, rle_pos, seq_pos);
    u64 local[SIGMA] = {};
    while(seq_pos <= i)
    {
      run = run_read(data.data(), rle_pos);
      seq_pos += run.second;
      local[run.first] += run.second;
    }
    return range_t(sample_sum(run.first, block) + local[run.first] - (seq_pos - i), run.first);
  }

  // bwt.cpp:405-427
  u64 select(u64 i, u64 c) const
  {
    if(c >= SIGMA || i == 0) { return 0; }
    if(i > count(c)) { return size(); }
    // CumulativeArray::inverse(i - 1): the block holding item i - 1 (0-based) of symbol c.
    u64 block = (u64)(std::upper_bound(cum[c].begin() + 1, cum[c].end(), i - 1) - (cum[c].begin() + 1));
    u64 seen = sample_sum(c, block);
    u64 rle_pos = block * RLE_BLOCK;
    u64 seq_pos = (block > 0 ? block_select(block) + 1 : 0);
    while(true)
    {
      range_t run = run_read(data.data(), rle_pos);
      seq_pos += run.second - 1;
      if(run.first == c)
      {
        seen += run.second;
        if(seen >= i) { return seq_pos + i - seen; }
      }
      seq_pos++;
    }
  }

  // bwt.cpp:429-443
  u64 at(u64 i) const
  {
    if(i >= size()) { return 0; }
    u64 block, rle_pos, seq_pos; locate(i, block, rle_pos, seq_pos);
    while(true)
    {
      range_t run = run_read(data.data(), rle_pos);
      seq_pos += run.second;
      if(seq_pos > i) { return run.first; }
    }
  }

  // bwt.cpp:525-536
  void character_counts(u64* counts) const
  {
    for(u64 c = 0; c < SIGMA; c++) { counts[c] = 0; }
    u64 rle_pos = 0;
    while(rle_pos < bytes())
    {
      range_t run = run_read(data.data(), rle_pos);
      counts[run.first] += run.second;
    }
  }

  // bwt.cpp:538-549
  u64 hash() const
  {
    u64 res = FNV_OFFSET_BASIS, rle_pos = 0;
    while(rle_pos < bytes())
    {
      range_t run = run_read(data.data(), rle_pos);
      for(u64 k = 0; k < run.second; k++) { res = fnv1a((u8)run.first, res); }
    }
    return res;
  }
};

//------------------------------------------------------------------------------
// FMI = BWT + C array (fmi.h:86-230; generic LF utils.h:335-355). The char maps of
// Alphabet are irrelevant to the path; equality of alphabets is checked by the caller.

struct FMI
{
  BWT bwt;
  u64 C[SIGMA + 1] = {};

  u64 size() const { return bwt.size(); }
  u64 sequences() const { return bwt.sequences; }

  void set_C_from_counts()
  {
    C[0] = 0;
    for(u64 c = 0; c < SIGMA; c++) { C[c + 1] = C[c] + bwt.count(c); }   // support.cpp:90
  }

  // fmi.h:147-150 / utils.h:335-341: (LF(i), BWT[i])
  range_t LF(u64 i) const
  {
    range_t t = bwt.inverse_select(i);
    return range_t(t.first + C[t.second], t.second);
  }
  // fmi.h:152-155 / utils.h:343-348
  u64 LF(u64 i, u64 c) const { return C[c] + bwt.rank(i, c); }
  // fmi.h:157-160 / utils.h:350-355
  range_t LF(range_t r, u64 c) const { return range_t(LF(r.first, c), LF(r.second + 1, c) - 1); }
  // fmi.h:165-169
  void LF(u64 i, std::array<u64, SIGMA>& results) const
  {
    bwt.ranks(i, results);
    for(u64 c = 1; c < SIGMA; c++) { results[c] += C[c]; }
  }
  // fmi.h:174-181
  void LF(range_t range, std::array<u64, SIGMA>& sp, std::array<u64, SIGMA>& ep) const
  {
    bwt.ranks(range.first, sp); bwt.ranks(range.second + 1, ep);
    for(u64 c = 1; c < SIGMA; c++) { sp[c] += C[c]; ep[c] += C[c] - 1; }
  }
  // fmi.h:186-193
  void LF(range_t range, std::array<range_t, SIGMA>& results) const
  {
    bwt.ranks(range, results);
    for(u64 c = 1; c < SIGMA; c++) { results[c].first += C[c]; results[c].second += C[c] - 1; }
  }
  // fmi.h:195-209 (pattern given as comp values)
  range_t find(const u8* pattern, u64 length) const
  {
    if(length == 0) { return range_t(0, size() - 1); }
    u64 k = length - 1;
    range_t range(C[pattern[k]], C[pattern[k] + 1] - 1);    // charRange, utils.h:318-323
    while(!range_empty(range) && k > 0)
    {
      k--;
      range = LF(range, pattern[k]);
    }
    return range;
  }
};

//------------------------------------------------------------------------------
// RLArray: sorted multiset of ranks as varint (delta value, run length) pairs
// (support.h:396-517) and its iterator (support.h:528-572).

struct RLArray
{
  std::vector<u8> data;
  u64 run_count = 0, value_count = 0;

  bool empty() const { return run_count == 0; }
  u64  bytes() const { return data.size(); }
  void clear() { data.clear(); data.shrink_to_fit(); run_count = value_count = 0; }
  void swap(RLArray& o) { data.swap(o.data); std::swap(run_count, o.run_count); std::swap(value_count, o.value_count); }

  void add_run(range_t run, u64& prev)      // support.h:511-516
  {
    bytecode_write(data, run.first - prev); prev = run.first;
    bytecode_write(data, run.second);
    run_count++; value_count += run.second;
  }
};

struct RLIterator
{
  const RLArray* array = nullptr;
  u64 pos = 0, ptr = 0;
  range_t run = range_t(0, 0);

  RLIterator() {}
  explicit RLIterator(const RLArray& a) : array(&a) { read(); }
  bool end() const { return pos >= array->run_count; }
  void next() { pos++; read(); }
  void read()                                  // support.h:563-568
  {
    if(end()) { run.first = ~(u64)0; run.second = ~(u64)0; return; }
    run.first += bytecode_read(array->data.data(), ptr);
    run.second = bytecode_read(array->data.data(), ptr);
  }
};

// support.h:415-429: sort the records, coalesce equal ranks, encode.
static void rlarray_from_records(RLArray& out, std::vector<range_t>& source)
{
  out.clear();
  if(source.empty()) { return; }
  std::sort(source.begin(), source.end());
  u64 prev = 0;
  RunBuffer rb;
  for(u64 k = 0; k < source.size(); k++)
  {
    if(rb.add(source[k])) { out.add_run(rb.run, prev); }
  }
  rb.flush(); out.add_run(rb.run, prev);
}

// support.h:434-453: two-way merge; `a` wins ties; inputs are cleared.
static void rlarray_merge(RLArray& out, RLArray& a, RLArray& b)
{
  RLArray result;
  if(a.empty()) { result.swap(b); a.clear(); out.swap(result); return; }
  if(b.empty()) { result.swap(a); b.clear(); out.swap(result); return; }
  RLIterator ai(a), bi(b);
  u64 prev = 0;
  RunBuffer rb;
  // the merged array is at most as long as both inputs together plus the re-coded first delta of one of them: no regrowth while it
  // is written (the reference writes into a BlockArray, which never copies; a std::vector grown by doubling copied every byte ~twice)
  result.data.reserve(a.bytes() + b.bytes() + 16);
  while(!ai.end() || !bi.end())
  {
    range_t temp;
    if(ai.run.first <= bi.run.first) { temp = ai.run; ai.next(); }
    else { temp = bi.run; bi.next(); }
    if(rb.add(temp)) { result.add_run(rb.run, prev); }
  }
  rb.flush(); result.add_run(rb.run, prev);
  a.clear(); b.clear();
  out.swap(result);
}

//------------------------------------------------------------------------------
// RankArray: k-way heap merge over the spilled arrays (support.h:576-638,
// support.cpp:528-574). The temp files of the reference are in-memory arrays here.

struct RankArray
{
  std::vector<RLArray>    files;
  std::vector<RLIterator> iterators;

  u64 size() const { return files.size(); }
  static u64 left(u64 i) { return 2 * i + 1; }
  static u64 right(u64 i) { return 2 * i + 2; }

  u64 smaller(u64 i, u64 j) const
  {
    return (iterators[j].run.first < iterators[i].run.first ? j : i);
  }
  void down(u64 i)                             // support.h:617-627
  {
    while(left(i) < size())
    {
      u64 next = smaller(i, left(i));
      if(right(i) < size()) { next = smaller(next, right(i)); }
      if(next == i) { return; }
      std::swap(iterators[i], iterators[next]);
      i = next;
    }
  }
  void open()                                  // support.cpp:538-552, 562-574
  {
    iterators.clear();
    for(u64 k = 0; k < size(); k++) { iterators.push_back(RLIterator(files[k])); }
    if(size() <= 1) { return; }
    u64 i = (size() - 2) / 2;
    while(true) { down(i); if(i == 0) { break; } i--; }
  }
  bool end() const { return iterators.empty() || iterators[0].end(); }
  range_t top() const { return iterators[0].run; }
  void next() { iterators[0].next(); down(0); }
};

//------------------------------------------------------------------------------
// MergeParameters (fmi.h:45-80)

struct MergeParameters
{
  u64 run_buffer_size    = 8u << 20;        // records
  u64 thread_buffer_size = 256u << 20;      // bytes
  u64 merge_buffers      = 6;
  u64 threads            = 1;
  u64 sequence_blocks    = 4;

  void sanitize(u64 hw_threads)             // fmi.cpp:462-468
  {
    threads = std::max(std::min(threads, hw_threads), (u64)1);
    sequence_blocks = std::max(sequence_blocks, (u64)1);
    threads = std::min(threads, sequence_blocks);
    if(merge_buffers == 0) { merge_buffers = 1; }
  }
};

//------------------------------------------------------------------------------
// MergeBuffer + mergeRA (fmi.cpp:139-257)

struct MergeBuffer
{
  MergeParameters parameters;
  std::mutex buffer_lock;
  std::vector<RLArray> merge_buffers;
  std::mutex ra_lock;
  RankArray ra;

  explicit MergeBuffer(const MergeParameters& p) : parameters(p), merge_buffers(p.merge_buffers) {}

  void write(RLArray& buffer)                // fmi.cpp:164-200
  {
    if(buffer.empty()) { return; }
    RLArray spilled; spilled.swap(buffer);
    std::lock_guard<std::mutex> lock(ra_lock);
    ra.files.push_back(RLArray());
    ra.files.back().swap(spilled);
  }

  void flush()                               // fmi.cpp:202-217
  {
    for(u64 i = 1; i < merge_buffers.size(); i++)
    {
      rlarray_merge(merge_buffers[i], merge_buffers[i], merge_buffers[i - 1]);
    }
    write(merge_buffers[merge_buffers.size() - 1]);
  }
};

struct SearchStats
{
  std::atomic<u64> single{0}, shortr{0}, longr{0};
  // Where the threads of the search phase spend their time (nanoseconds, summed over threads):
  //   dfs            inside buildRA's loop over trie nodes, without the calls below
  //   sort_encode    RLArray(run_buffer): sort + coalesce + encode (fmi.cpp:224)
  //   merge_thread   the two-way merge of that array into the thread buffer (fmi.cpp:225)
  //   lock_wait      waiting for buffer_lock (fmi.cpp:240)
  //   merge_global   two-way merges with global merge buffers taken over from others (fmi.cpp:252)
  //   write          MergeBuffer::write (the reference's temp file; an in-memory array here)
  //   flush          MergeBuffer::flush after the threads have joined (one thread, wall clock)
  //   threads_wall   sum over threads of (finish - start): dfs + ... + write + idle at the end is the difference to threads x wall
  std::atomic<u64> t_dfs{0}, t_sort_encode{0}, t_merge_thread{0}, t_lock_wait{0}, t_merge_global{0}, t_write{0}, t_flush{0}, t_threads_wall{0};
};

static inline u64 now_ns() { return (u64)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void merge_ra(MergeBuffer& mb, RLArray& thread_buffer, std::vector<range_t>& run_buffer, bool force, SearchStats* stats = nullptr)
{
  u64 t0 = (stats ? now_ns() : 0), t1 = 0;
#define ORC_LAP(field) if(stats) { t1 = now_ns(); stats->field += t1 - t0; t0 = t1; }
  RLArray temp;                                                   // fmi.cpp:224-226
  rlarray_from_records(temp, run_buffer); run_buffer.clear();
  ORC_LAP(t_sort_encode)
  rlarray_merge(thread_buffer, thread_buffer, temp);
  ORC_LAP(t_merge_thread)
  if(!force && thread_buffer.bytes() < mb.parameters.thread_buffer_size) { return; }

  for(u64 i = 0; i < mb.merge_buffers.size(); i++)                // fmi.cpp:236-254
  {
    bool done = false;
    {
      std::unique_lock<std::mutex> lock(mb.buffer_lock);
      ORC_LAP(t_lock_wait)
      if(mb.merge_buffers[i].empty()) { thread_buffer.swap(mb.merge_buffers[i]); done = true; }
      else { temp.swap(mb.merge_buffers[i]); }
    }
    if(done) { return; }
    rlarray_merge(thread_buffer, thread_buffer, temp);
    ORC_LAP(t_merge_global)
  }
  mb.write(thread_buffer);                                        // fmi.cpp:256
  ORC_LAP(t_write)
#undef ORC_LAP
}

//------------------------------------------------------------------------------
// buildRA (fmi.cpp:261-334): DFS over the reverse trie of B's sequences in one block.

struct MergePosition
{
  u64 a_pos; range_t b_range;
  MergePosition(u64 a, range_t b) : a_pos(a), b_range(b) {}
};


static void build_ra(std::atomic<u64>& tail, const std::vector<range_t>& blocks,
  const FMI& a, const FMI& b, MergeBuffer& mb, SearchStats* stats)
{
  const u64 t_start = (stats ? now_ns() : 0);
  u64 in_merge = 0;                                               // nanoseconds this thread spent inside merge_ra
  auto timed_merge = [&](RLArray& tb, std::vector<range_t>& rbuf, bool force)
  {
    const u64 t0 = (stats ? now_ns() : 0);
    merge_ra(mb, tb, rbuf, force, stats);
    if(stats) { in_merge += now_ns() - t0; }
  };
  while(true)
  {
    u64 block = tail++;                                           // utils.cpp:204-209
    if(block >= blocks.size())
    {
      if(stats) { const u64 wall = now_ns() - t_start; stats->t_threads_wall += wall; stats->t_dfs += wall - in_merge; }
      return;
    }
    range_t sequence_range = blocks[block];

    RLArray thread_buffer;
    std::vector<range_t> run_buffer; run_buffer.reserve(std::min(mb.parameters.run_buffer_size, (u64)1 << 24));
    std::stack<MergePosition> positions;
    std::array<u64, SIGMA> a_pos, b_sp, b_ep;
    std::array<range_t, SIGMA> b_range;
    u64 n_single = 0, n_short = 0, n_long = 0;

    positions.push(MergePosition(a.sequences(), sequence_range)); // fmi.cpp:286
    while(!positions.empty())
    {
      MergePosition curr = positions.top(); positions.pop();
      run_buffer.push_back(range_t(curr.a_pos, range_length(curr.b_range)));   // fmi.cpp:290
      if(run_buffer.size() >= mb.parameters.run_buffer_size) { timed_merge(thread_buffer, run_buffer, false); }

      if(range_length(curr.b_range) == 1)                         // fmi.cpp:296-303
      {
        n_single++;
        range_t pred = b.LF(curr.b_range.first);
        if(pred.second != 0)
        {
          positions.push(MergePosition(a.LF(curr.a_pos, pred.second), range_t(pred.first, pred.first)));
        }
      }
      else if(range_length(curr.b_range) <= SHORT_RANGE)          // fmi.cpp:304-314
      {
        n_short++;
        b.LF(curr.b_range, b_range);
        for(u64 c = 1; c < SIGMA; c++)
        {
          if(!range_empty(b_range[c])) { positions.push(MergePosition(a.LF(curr.a_pos, c), b_range[c])); }
        }
      }
      else                                                        // fmi.cpp:315-322
      {
        n_long++;
        a.LF(curr.a_pos, a_pos); b.LF(curr.b_range, b_sp, b_ep);
        for(u64 c = 1; c < SIGMA; c++)
        {
          if(b_sp[c] <= b_ep[c]) { positions.push(MergePosition(a_pos[c], range_t(b_sp[c], b_ep[c]))); }
        }
      }
    }
    timed_merge(thread_buffer, run_buffer, true);                 // fmi.cpp:325
    if(stats) { stats->single += n_single; stats->shortr += n_short; stats->longr += n_long; }
  }
}

// Search phase of FMI::FMI(a, b, params): fmi.cpp:351-358.
static void search(const FMI& a, const FMI& b, const MergeParameters& p, MergeBuffer& mb, SearchStats* stats)
{
  if(b.sequences() == 0) { return; }
  std::vector<range_t> blocks = get_bounds(range_t(0, b.sequences() - 1), p.sequence_blocks);   // utils.cpp:189-197
  u64 thread_count = std::max(std::min(p.threads, (u64)blocks.size()), (u64)1);
  std::atomic<u64> tail(0);
  std::vector<std::thread> threads;
  for(u64 t = 0; t < thread_count; t++)
  {
    threads.emplace_back(build_ra, std::ref(tail), std::cref(blocks), std::cref(a), std::cref(b), std::ref(mb), stats);
  }
  for(auto& t : threads) { t.join(); }
  const u64 t0 = (stats ? now_ns() : 0);
  mb.flush();
  if(stats) { stats->t_flush += now_ns() - t0; }
}

//------------------------------------------------------------------------------
// Interleave (bwt.cpp:152-314): producer thread merges the spilled arrays into maximal
// (rank, count) runs and hands them over in 1 Mi-run chunks; the consumer interleaves.

struct RABuffer
{
  std::mutex mtx;
  std::condition_variable full, empty;
  bool finished = false;
  std::vector<range_t> buffer;

  void get(std::vector<range_t>& out, bool& last)
  {
    std::unique_lock<std::mutex> lock(mtx);
    full.wait(lock, [this]() { return !buffer.empty(); });
    out.swap(buffer); last = finished;
    empty.notify_one();
  }
  void add(std::vector<range_t>& in, bool last)
  {
    std::unique_lock<std::mutex> lock(mtx);
    empty.wait(lock, [this]() { return buffer.empty(); });
    buffer.swap(in); finished = last;
    full.notify_one();
  }
};

static void merge_ra_producer(RankArray& ra, RABuffer& out)       // bwt.cpp:194-213
{
  std::vector<range_t> chunk; chunk.reserve(RA_CHUNK);
  RunBuffer rb;
  for(ra.open(); !ra.end(); ra.next())
  {
    if(rb.add(ra.top()))
    {
      chunk.push_back(rb.run);
      if(chunk.size() >= RA_CHUNK) { out.add(chunk, ra.end()); }
    }
  }
  rb.flush(); chunk.push_back(rb.run);
  if(chunk.size() > 0) { out.add(chunk, ra.end()); }
}

static void merge_bwt(const BWT& a, const BWT& b, BWT& result, u64* counts, RABuffer& ra_buffer)   // bwt.cpp:215-282
{
  std::vector<range_t> in; in.reserve(RA_CHUNK);
  RunBuffer out;
  bool ra_finished = false;
  u64 a_rle = 0, b_rle = 0, a_seq = 0;
  const u8* ad = a.data.data(); const u8* bd = b.data.data();
  range_t a_run = (a.bytes() > 0 ? run_read(ad, a_rle) : range_t(0, 0));
  range_t b_run = (b.bytes() > 0 ? run_read(bd, b_rle) : range_t(0, 0));

  auto emit = [&]() { run_write(result.data, out.run.first, out.run.second); counts[out.run.first] += out.run.second; };

  while(!ra_finished)
  {
    ra_buffer.get(in, ra_finished);
    for(u64 k = 0; k < in.size(); k++)
    {
      range_t curr = in[k];
      while(a_seq < curr.first)                                   // bwt.cpp:234-247
      {
        u64 len = std::min(curr.first - a_seq, a_run.second);
        if(out.add(a_run.first, len)) { emit(); }
        a_run.second -= len; a_seq += len;
        if(a_run.second == 0 && a_rle < a.bytes()) { a_run = run_read(ad, a_rle); }
      }
      while(curr.second > 0)                                      // bwt.cpp:248-261
      {
        u64 len = std::min(curr.second, b_run.second);
        if(out.add(b_run.first, len)) { emit(); }
        b_run.second -= len; curr.second -= len;
        if(b_run.second == 0 && b_rle < b.bytes()) { b_run = run_read(bd, b_rle); }
      }
    }
    in.clear();
  }
  while(a_run.second > 0)                                         // bwt.cpp:266-276
  {
    if(out.add(a_run)) { emit(); }
    if(a_rle < a.bytes()) { a_run = run_read(ad, a_rle); } else { a_run.second = 0; }
  }
  out.flush(); emit();                                            // bwt.cpp:278-281
}

// BWT::BWT(a, b, ra): bwt.cpp:286-314.
static void interleave(BWT& a, BWT& b, RankArray& ra, BWT& result)
{
  a.destroy(); b.destroy();
  RABuffer ra_buffer;
  u64 counts[SIGMA] = {};
  result.data.clear();
  // Degenerate inputs (an empty side) are outside the reference's contract (Run::read on an
  // empty array is undefined there); merge_bwt() below starts from an empty run instead.
  std::thread producer(merge_ra_producer, std::ref(ra), std::ref(ra_buffer));
  merge_bwt(a, b, result, counts, ra_buffer);
  producer.join();
  result.sequences = a.sequences + b.sequences;
  result.bases = a.bases + b.bases;
  result.build();
}

// FMI::FMI(a, b, params): fmi.cpp:336-369. Consumes a and b.
static void merge(FMI& a, FMI& b, MergeParameters p, FMI& out, SearchStats* stats, double* phase_seconds)
{
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = now();
  MergeBuffer mb(p);
  search(a, b, p, mb, stats);
  double t1 = now();
  interleave(a.bwt, b.bwt, mb.ra, out.bwt);
  double t2 = now();
  for(u64 c = 0; c <= SIGMA; c++) { out.C[c] = a.C[c] + b.C[c]; }   // fmi.cpp:367-368
  if(phase_seconds) { phase_seconds[0] = t1 - t0; phase_seconds[1] = t2 - t1; }
}

//------------------------------------------------------------------------------
// Encoding a plain symbol string the way every reader of the reference does
// (PlainData::read, formats.cpp:133-161): maximal runs -> Run::write.

static void encode_symbols(const u8* symbols, u64 n, BWT& out)
{
  out.data.clear();
  RunBuffer rb;
  u64 counts[SIGMA] = {};
  for(u64 k = 0; k < n; k++)
  {
    if(rb.add(symbols[k])) { run_write(out.data, rb.run.first, rb.run.second); counts[rb.run.first] += rb.run.second; }
  }
  rb.flush(); run_write(out.data, rb.run.first, rb.run.second); counts[rb.run.first] += rb.run.second;
  out.sequences = counts[0];                                      // bwt.cpp:468-474
  out.bases = n;
  out.build();
}

// The same from (symbol, length) pairs: adjacent pairs of one symbol coalesce in the RunBuffer exactly like repeated
// symbols do (utils.h:121-142), so strings of billions of positions can be stated in a few megabytes.
static void encode_runs(const u64* symbols, const u64* lengths, u64 nruns, BWT& out)
{
  out.data.clear();
  RunBuffer rb;
  u64 counts[SIGMA] = {};
  u64 n = 0;
  for(u64 k = 0; k < nruns; k++)
  {
    if(lengths[k] == 0) { continue; }
    n += lengths[k];
    if(rb.add(symbols[k], lengths[k])) { run_write(out.data, rb.run.first, rb.run.second); counts[rb.run.first] += rb.run.second; }
  }
  rb.flush();
  if(rb.run.second > 0) { run_write(out.data, rb.run.first, rb.run.second); counts[rb.run.first] += rb.run.second; }
  out.sequences = counts[0];
  out.bases = n;
  out.build();
}

static void decode_symbols(const BWT& bwt, u8* symbols)
{
  u64 rle_pos = 0, k = 0;
  while(rle_pos < bwt.bytes())
  {
    range_t run = run_read(bwt.data.data(), rle_pos);
    std::memset(symbols + k, (int)run.first, run.second); k += run.second;
  }
}

//------------------------------------------------------------------------------
// Brute-force ground truth: multi-string BWT by suffix sorting. `text` holds the sequences
// as comp values 1..5, each followed by a 0 terminator. Suffixes are compared symbol by
// symbol in comp order; a terminator ends the comparison and ties are broken by sequence
// index.

static void naive_bwt(const u8* text, u64 n, std::vector<u8>& bwt_out)
{
  // packed[p]: the next 21 symbols of the suffix at p (3 bits each, first symbol in the
  // top bits), zero after the terminator. rem[p]: symbols up to and including the terminator.
  std::vector<u64> packed(n + 1, 0);
  std::vector<std::uint32_t> rem(n + 1, 0), seq(n, 0);
  for(u64 p = n; p-- > 0; )
  {
    if(text[p] == 0) { packed[p] = 0; rem[p] = 1; }
    else { packed[p] = ((u64)text[p] << 60) | (packed[p + 1] >> 3); rem[p] = rem[p + 1] + 1; }
  }
  std::uint32_t s = 0;
  for(u64 p = 0; p < n; p++) { seq[p] = s; if(text[p] == 0) { s++; } }

  std::vector<u64> sa(n);
  for(u64 p = 0; p < n; p++) { sa[p] = p; }
  std::sort(sa.begin(), sa.end(), [&](u64 x, u64 y)
  {
    for(u64 k = 0; ; k += 21)
    {
      u64 wx = (k < rem[x] ? packed[x + k] : 0), wy = (k < rem[y] ? packed[y + k] : 0);
      if(wx != wy) { return wx < wy; }
      if(k + 21 >= rem[x] || k + 21 >= rem[y]) { return seq[x] < seq[y]; }
    }
  });
  bwt_out.resize(n);
  for(u64 k = 0; k < n; k++)
  {
    u64 p = sa[k];
    bwt_out[k] = (p == 0 || text[p - 1] == 0 ? 0 : text[p - 1]);
  }
}

//------------------------------------------------------------------------------
// Synthetic reads (SURVEY.md 8(d) distribution; counter-based so that the GPU tooling can
// generate the same reads in parallel): base t of read j draws z1, z2 from splitmix64 at
// counters 2*(j*stride + t) + 1 and + 2; N (comp 5) if z1 % 256 == 0, else 1 + z2 % 4.

static inline u64 splitmix64_at(u64 seed, u64 counter)
{
  u64 z = seed + counter * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

static inline u8 synthetic_base(u64 seed, u64 read, u64 t)
{
  u64 idx = read * 256 + t;       // stride 256 >= any read length used
  u64 z1 = splitmix64_at(seed, 2 * idx + 1);
  if(z1 % 256 == 0) { return 5; }
  return (u8)(1 + splitmix64_at(seed, 2 * idx + 2) % 4);
}

} // namespace orc

//==============================================================================
// C entry points (loaded with ctypes by tests/, smoke() and bench.py's cpu_baseline).

using namespace orc;

extern "C"
{

// --- codec known-answer hooks -------------------------------------------------

// Appends Run::write(comp, length) to an array that already holds `prefill` bytes; returns
// the number of bytes produced.
u64 orc_run_write(u64 prefill, u64 comp, u64 length, u8* out, u64 out_capacity)
{
  std::vector<u8> buf; run_write(buf, comp, length, prefill);
  u64 n = std::min((u64)buf.size(), out_capacity);
  std::memcpy(out, buf.data(), n);
  return buf.size();
}

// Decodes runs from `data`; returns the number of runs, fills comp/len arrays up to capacity.
u64 orc_run_decode(const u8* data, u64 nbytes, u64* comps, u64* lens, u64 capacity)
{
  u64 i = 0, k = 0;
  while(i < nbytes)
  {
    range_t run = run_read(data, i);
    if(k < capacity) { comps[k] = run.first; lens[k] = run.second; }
    k++;
  }
  return k;
}

u64 orc_bytecode_write(u64 value, u8* out, u64 out_capacity)
{
  std::vector<u8> buf; bytecode_write(buf, value);
  std::memcpy(out, buf.data(), std::min((u64)buf.size(), out_capacity));
  return buf.size();
}

u64 orc_bytecode_read(const u8* data, u64* consumed)
{
  u64 i = 0; u64 v = bytecode_read(data, i); *consumed = i; return v;
}

u64 orc_get_bounds(u64 first, u64 last, u64 blocks, u64* out_first, u64* out_last, u64 capacity)
{
  std::vector<range_t> b = get_bounds(range_t(first, last), blocks);
  for(u64 k = 0; k < b.size() && k < capacity; k++) { out_first[k] = b[k].first; out_last[k] = b[k].second; }
  return b.size();
}

u64 orc_fnv1a_bytes(const u8* data, u64 n)
{
  u64 h = FNV_OFFSET_BASIS;
  for(u64 k = 0; k < n; k++) { h = fnv1a(data[k], h); }
  return h;
}

// RunBuffer trace: feeds (value, length) pairs, writes completed runs; returns run count.
u64 orc_runbuffer(const u64* values, const u64* lengths, u64 n, u64* out_values, u64* out_lengths)
{
  RunBuffer rb; u64 k = 0;
  for(u64 i = 0; i < n; i++)
  {
    if(rb.add(values[i], lengths[i])) { out_values[k] = rb.run.first; out_lengths[k] = rb.run.second; k++; }
  }
  rb.flush(); out_values[k] = rb.run.first; out_lengths[k] = rb.run.second; k++;
  return k;
}

// --- synthetic reads ------------------------------------------------------------

// Writes `nreads` reads of `readlen` symbols each, every read followed by a 0 terminator.
// `first_read` is the index of the first read within the set (for sharded generation).
void orc_generate_reads(u64 seed, u64 first_read, u64 nreads, u64 readlen, u8* out)
{
  for(u64 j = 0; j < nreads; j++)
  {
    u8* dst = out + j * (readlen + 1);
    for(u64 t = 0; t < readlen; t++) { dst[t] = synthetic_base(seed, first_read + j, t); }
    dst[readlen] = 0;
  }
}

// --- FMI handles ------------------------------------------------------------------

void* orc_fmi_from_text(const u8* text, u64 n)
{
  std::vector<u8> symbols; naive_bwt(text, n, symbols);
  FMI* f = new FMI();
  encode_symbols(symbols.data(), n, f->bwt);
  f->set_C_from_counts();
  return f;
}

void* orc_fmi_from_symbols(const u8* symbols, u64 n)
{
  FMI* f = new FMI();
  encode_symbols(symbols, n, f->bwt);
  f->set_C_from_counts();
  return f;
}

void* orc_fmi_from_runs(const u64* symbols, const u64* lengths, u64 nruns)
{
  FMI* f = new FMI();
  encode_runs(symbols, lengths, nruns, f->bwt);
  f->set_C_from_counts();
  return f;
}

void* orc_fmi_from_native(const u8* data, u64 nbytes, u64 sequences, u64 bases)
{
  FMI* f = new FMI();
  f->bwt.data.assign(data, data + nbytes);
  f->bwt.sequences = sequences; f->bwt.bases = bases;
  f->bwt.build();
  f->set_C_from_counts();
  return f;
}

void* orc_fmi_clone(const void* h) { return new FMI(*(const FMI*)h); }
void  orc_fmi_free(void* h) { delete (FMI*)h; }

u64 orc_fmi_bases(const void* h)     { return ((const FMI*)h)->bwt.bases; }
u64 orc_fmi_sequences(const void* h) { return ((const FMI*)h)->bwt.sequences; }
u64 orc_fmi_bytes(const void* h)     { return ((const FMI*)h)->bwt.bytes(); }
u64 orc_fmi_blocks(const void* h)    { return ((const FMI*)h)->bwt.blocks(); }
u64 orc_fmi_hash(const void* h)      { return ((const FMI*)h)->bwt.hash(); }
void orc_fmi_C(const void* h, u64* out) { for(u64 c = 0; c <= SIGMA; c++) { out[c] = ((const FMI*)h)->C[c]; } }
void orc_fmi_data(const void* h, u8* out) { const FMI* f = (const FMI*)h; std::memcpy(out, f->bwt.data.data(), f->bwt.bytes()); }
void orc_fmi_symbols(const void* h, u8* out) { decode_symbols(((const FMI*)h)->bwt, out); }
void orc_fmi_character_counts(const void* h, u64* out) { ((const FMI*)h)->bwt.character_counts(out); }

// Samples as plain arrays: block_end[blocks], cum[6][blocks + 1] (row-major).
void orc_fmi_samples(const void* h, u64* block_end, u64* cum)
{
  const FMI* f = (const FMI*)h; u64 nb = f->bwt.blocks();
  std::memcpy(block_end, f->bwt.block_end.data(), nb * sizeof(u64));
  for(u64 c = 0; c < SIGMA; c++) { std::memcpy(cum + c * (nb + 1), f->bwt.cum[c].data(), (nb + 1) * sizeof(u64)); }
}

u64  orc_rank(const void* h, u64 i, u64 c) { return ((const FMI*)h)->bwt.rank(i, c); }
u64  orc_select(const void* h, u64 i, u64 c) { return ((const FMI*)h)->bwt.select(i, c); }
u64  orc_at(const void* h, u64 i) { return ((const FMI*)h)->bwt.at(i); }
void orc_inverse_select(const void* h, u64 i, u64* rank, u64* comp)
{
  range_t r = ((const FMI*)h)->bwt.inverse_select(i); *rank = r.first; *comp = r.second;
}
void orc_ranks_at(const void* h, u64 i, u64* out)
{
  std::array<u64, SIGMA> r; r.fill(0); ((const FMI*)h)->bwt.ranks(i, r);
  for(u64 c = 0; c < SIGMA; c++) { out[c] = r[c]; }
}
void orc_ranks_range(const void* h, u64 sp, u64 ep, u64* out_first, u64* out_second)
{
  std::array<range_t, SIGMA> r; ((const FMI*)h)->bwt.ranks(range_t(sp, ep), r);
  for(u64 c = 1; c < SIGMA; c++) { out_first[c] = r[c].first; out_second[c] = r[c].second; }
}
void orc_LF(const void* h, u64 i, u64* next, u64* comp)
{
  range_t r = ((const FMI*)h)->LF(i); *next = r.first; *comp = r.second;
}
u64 orc_LF_c(const void* h, u64 i, u64 c) { return ((const FMI*)h)->LF(i, c); }
void orc_find(const void* h, const u8* pattern, u64 length, u64* sp, u64* ep)
{
  range_t r = ((const FMI*)h)->find(pattern, length); *sp = r.first; *ep = r.second;
}

// --- the hot path -------------------------------------------------------------------

struct OrcParams { u64 run_buffer_size, thread_buffer_size, merge_buffers, threads, sequence_blocks; };

static MergeParameters to_params(const OrcParams* p)
{
  MergeParameters mp;
  if(p)
  {
    if(p->run_buffer_size)    { mp.run_buffer_size = p->run_buffer_size; }
    if(p->thread_buffer_size) { mp.thread_buffer_size = p->thread_buffer_size; }
    if(p->merge_buffers)      { mp.merge_buffers = p->merge_buffers; }
    if(p->threads)            { mp.threads = p->threads; }
    if(p->sequence_blocks)    { mp.sequence_blocks = p->sequence_blocks; } else { mp.sequence_blocks = 4 * mp.threads; }
  }
  mp.sanitize(std::max(1u, std::thread::hardware_concurrency()));
  return mp;
}

// Search phase only: the rank array as maximal (rank, count) runs in rank order.
// Returns the number of runs; fills up to `capacity`. stats[3] = node counts per branch.
u64 orc_search(const void* ha, const void* hb, const OrcParams* params, u64* ranks, u64* counts, u64 capacity, u64* stats)
{
  const FMI* a = (const FMI*)ha; const FMI* b = (const FMI*)hb;
  MergeParameters p = to_params(params);
  MergeBuffer mb(p);
  SearchStats st;
  search(*a, *b, p, mb, &st);
  if(stats) { stats[0] = st.single; stats[1] = st.shortr; stats[2] = st.longr; }
  u64 k = 0;
  RunBuffer rb; bool any = false;
  for(mb.ra.open(); !mb.ra.end(); mb.ra.next())
  {
    any = true;
    if(rb.add(mb.ra.top())) { if(k < capacity) { ranks[k] = rb.run.first; counts[k] = rb.run.second; } k++; }
  }
  if(any) { rb.flush(); if(k < capacity) { ranks[k] = rb.run.first; counts[k] = rb.run.second; } k++; }
  return k;
}

// Full merge: FMI::FMI(a, b, params). Consumes the contents of a and b (the handles must
// still be freed). seconds[0] = search, seconds[1] = interleave + sample build.
void* orc_merge(void* ha, void* hb, const OrcParams* params, double* seconds)
{
  FMI* a = (FMI*)ha; FMI* b = (FMI*)hb;
  FMI* out = new FMI();
  merge(*a, *b, to_params(params), *out, nullptr, seconds);
  return out;
}

// The same with the time accounting of the search phase: timing[0..7] = dfs, sort_encode, merge_thread, lock_wait, merge_global, write, flush,
// threads_wall in seconds (summed over threads except flush), timing[8] = threads used, timing[9] = sequence blocks.
void* orc_merge_timed(void* ha, void* hb, const OrcParams* params, double* seconds, double* timing)
{
  FMI* a = (FMI*)ha; FMI* b = (FMI*)hb;
  FMI* out = new FMI();
  SearchStats st;
  MergeParameters p = to_params(params);
  merge(*a, *b, p, *out, &st, seconds);
  const std::atomic<u64>* src[8] = {&st.t_dfs, &st.t_sort_encode, &st.t_merge_thread, &st.t_lock_wait, &st.t_merge_global, &st.t_write, &st.t_flush, &st.t_threads_wall};
  for(int k = 0; k < 8; k++) { timing[k] = (double)src[k]->load() * 1e-9; }
  timing[8] = (double)std::min(p.threads, p.sequence_blocks); timing[9] = (double)p.sequence_blocks;
  return out;
}

// Interleave two symbol strings given an explicit rank array (one value per B position,
// non-decreasing): the definition the GPU interleave kernel is checked against.
void orc_interleave_symbols(const u8* a, u64 na, const u8* b, u64 nb, const u64* ra, u8* out)
{
  u64 ai = 0, k = 0;
  for(u64 i = 0; i < nb; i++)
  {
    while(ai < ra[i] && ai < na) { out[k++] = a[ai++]; }
    out[k++] = b[i];
  }
  while(ai < na) { out[k++] = a[ai++]; }
}

} // extern "C"
