/*
  fmi.h -- host facade: FMI, MergeParameters and the format-dispatching load / serialize
  (reference fmi.h:38-230, fmi.cpp:336-495).  The merging constructor runs the whole hot path
  on the GPU through the C ABI; everything else is small host code.
*/
#ifndef BWTM_HOST_FMI_H
#define BWTM_HOST_FMI_H

#include <thread>
#include "bwt.h"

namespace bwtmerge
{

class FMI;
void serialize(const FMI& fmi, const std::string& filename, const std::string& format);
void load(FMI& fmi, const std::string& filename, const std::string& format);

/*
  The reference's knobs.  Buffer sizes, merge buffers and the temp directory have nothing to
  configure on the device (no buffer hierarchy, no temp files); they are accepted, printed and
  otherwise ignored.  sequence_blocks is honoured by the two-step API (buildRA: one bwtm_search()
  call per block); the merging constructor searches all sequences in one call, which is what the
  device wants (it partitions the work itself).
  lazy_host (an addition): the merged FMI is left on the device and its host form is produced when
  something asks for it; the tool sets it for every merge but the last of a chain.
*/
struct MergeParameters
{
  typedef range_type run_type;

  const static size_type RUN_BUFFER_SIZE = 8 * MEGABYTE;
  const static size_type THREAD_BUFFER_SIZE = 256 * MEGABYTE;
  const static size_type MERGE_BUFFERS = 6;
  const static size_type BLOCKS_PER_THREAD = 4;

  MergeParameters() :
    run_buffer_size(RUN_BUFFER_SIZE), thread_buffer_size(THREAD_BUFFER_SIZE), merge_buffers(MERGE_BUFFERS),
    threads(Parallel::max_threads), sequence_blocks(threads * BLOCKS_PER_THREAD), temp_dir("."), lazy_host(false) {}

  void sanitize()
  {
    threads = Range::bound(threads, 1, Parallel::max_threads);
    sequence_blocks = std::max(sequence_blocks, (size_type)1);
    threads = std::min(threads, sequence_blocks);
  }

  static double defaultRB() { return inMegabytes(RUN_BUFFER_SIZE * sizeof(run_type)); }
  static double defaultTB() { return inMegabytes(THREAD_BUFFER_SIZE); }
  static size_type defaultMB() { return MERGE_BUFFERS; }
  static size_type defaultT()  { return Parallel::max_threads; }
  static size_type defaultSB() { return BLOCKS_PER_THREAD; }

  void setRB(size_type mb) { run_buffer_size = mb * MEGABYTE / sizeof(run_type); }
  void setTB(size_type mb) { thread_buffer_size = mb * MEGABYTE; }
  void setMB(size_type n)  { merge_buffers = n; }
  void setT(size_type n)   { threads = n; }
  void setSB(size_type n)  { sequence_blocks = n; }
  void setTemp(const std::string& directory)
  {
    if(directory.empty()) { temp_dir = "."; }
    else { temp_dir = (directory.back() == '/' ? directory.substr(0, directory.length() - 1) : directory); }
  }

  size_type run_buffer_size, thread_buffer_size;
  size_type merge_buffers;
  size_type threads, sequence_blocks;
  std::string temp_dir;
  bool lazy_host;
};

inline std::ostream& operator<<(std::ostream& out, const MergeParameters& p)
{
  out << "Run buffers:      " << inMegabytes(p.run_buffer_size * sizeof(MergeParameters::run_type)) << " MB" << std::endl;
  out << "Thread buffers:   " << inMegabytes(p.thread_buffer_size) << " MB" << std::endl;
  out << "Merge buffers:    " << p.merge_buffers << std::endl;
  out << "Threads:          " << p.threads << std::endl;
  out << "Sequence blocks:  " << p.sequence_blocks << std::endl;
  out << "Temp directory:   " << p.temp_dir << std::endl;
  return out;
}

class FMI
{
public:
  typedef BWT::size_type size_type;
  const static size_type SHORT_RANGE = 256;

  FMI() {}

  // Merges a and b, destroying them (reference fmi.h:107-110).
  FMI(FMI& a, FMI& b, MergeParameters parameters = MergeParameters());

  void swap(FMI& other) { bwt.swap(other.bwt); std::swap(alpha, other.alpha); }

  size_type size() const { return bwt.size(); }
  size_type sequences() const { return bwt.sequences(); }
  range_type charRange(comp_type comp) const { return range_type(alpha.C[comp], alpha.C[comp + 1] - 1); }

  // (LF(i), BWT[i])
  range_type LF(size_type i) const
  {
    range_type t = bwt.inverse_select(i);
    return range_type(t.first + alpha.C[t.second], t.second);
  }
  size_type LF(size_type i, comp_type comp) const { return alpha.C[comp] + bwt.rank(i, comp); }
  range_type LF(range_type range, comp_type comp) const { return range_type(LF(range.first, comp), LF(range.second + 1, comp) - 1); }
  void LF(size_type i, BWT::ranks_type& results) const
  {
    bwt.ranks(i, results);
    for(size_type c = 1; c < alpha.sigma; c++) { results[c] += alpha.C[c]; }
  }
  void LF(range_type range, BWT::ranks_type& sp, BWT::ranks_type& ep) const
  {
    bwt.ranks(range.first, sp); bwt.ranks(range.second + 1, ep);
    for(size_type c = 1; c < alpha.sigma; c++) { sp[c] += alpha.C[c]; ep[c] += alpha.C[c] - 1; }
  }
  void LF(range_type range, BWT::rank_ranges_type& results) const
  {
    bwt.ranks(range, results);
    for(size_type c = 1; c < alpha.sigma; c++) { results[c].first += alpha.C[c]; results[c].second += alpha.C[c] - 1; }
  }

  // Backward search; the pattern is given in characters.
  template<class Iterator>
  range_type find(Iterator begin, Iterator end) const
  {
    if(begin == end) { return range_type(0, size() - 1); }
    --end;
    range_type range = charRange(alpha.char2comp[(byte_type)*end]);
    while(!Range::empty(range) && end != begin)
    {
      --end;
      range = LF(range, alpha.char2comp[(byte_type)*end]);
    }
    return range;
  }
  template<class Container> range_type find(const Container& pattern) const { return find(pattern.begin(), pattern.end()); }

  template<class Format> void serialize(const std::string& filename) const;
  template<class Format> void load(const std::string& filename);

  BWT      bwt;
  Alphabet alpha;
};

// Search phase (buildRA, reference fmi.cpp:272-334) as a free function: makes sure a and b are on the device and
// fills a device rank array; parameters.sequence_blocks splits the sequences of b.
inline void buildRA(const FMI& a, const FMI& b, const MergeParameters& parameters, RankArray& ra)
{
  ra.clear();
  ra.a = a.bwt.onDevice(a.alpha.C);
  ra.b = b.bwt.onDevice(b.alpha.C);
  gpuCheck(bwtm_ra_create(ra.a, ra.b, &ra.handle), "buildRA()");
  if(b.sequences() == 0) { return; }
  for(range_type block : getBounds(range_type(0, b.sequences() - 1), parameters.sequence_blocks))
  {
    gpuCheck(bwtm_search(ra.a, ra.b, block.first, block.second, ra.handle), "buildRA()");
  }
}

// The output buffers of bwtm_merge_host are the result's own page-locked arrays.
inline void* mergeSink(void* user, int what, uint64_t nbytes)
{
  BWT* bwt = (BWT*)user;
  switch(what)
  {
  case BWTM_BUF_DATA:      bwt->data.bytes.resizeUninitialized(std::max<uint64_t>(nbytes, 1)); bwt->data.bytes.resizeUninitialized(nbytes); return bwt->data.bytes.data();
  case BWTM_BUF_BLOCK_END: bwt->block_end.resizeUninitialized(std::max<uint64_t>(nbytes / sizeof(size_type), 1)); bwt->block_end.resizeUninitialized(nbytes / sizeof(size_type)); return bwt->block_end.data();
  case BWTM_BUF_CUM:       bwt->cum_flat.resizeUninitialized(nbytes / sizeof(size_type)); return bwt->cum_flat.data();
  case BWTM_BUF_ANCHORS:   bwt->anchors.resizeUninitialized(std::max<uint64_t>(nbytes / sizeof(size_type), 1)); bwt->anchors.resizeUninitialized(nbytes / sizeof(size_type)); return bwt->anchors.data();
  case BWTM_BUF_FIELDS:    bwt->fields.resizeUninitialized(std::max<uint64_t>((nbytes + 1) / 2, 1)); bwt->fields.resizeUninitialized((nbytes + 1) / 2); return bwt->fields.data();
  }
  return nullptr;
}

inline FMI::FMI(FMI& a, FMI& b, MergeParameters parameters)
{
  if(a.alpha != b.alpha)
  {
    std::cerr << "FMI::FMI(): Cannot merge BWTs with different alphabets" << std::endl;
    std::exit(EXIT_FAILURE);
  }
#ifdef VERBOSE_STATUS_INFO
  // the reference's progress lines on stderr (fmi.cpp:344-364, bwt.cpp:300-313), with the device's phases behind them
  std::cerr << "bwt_merge: " << a.sequences() << " sequences of total length " << a.size() << std::endl;
  std::cerr << "bwt_merge: Adding " << b.sequences() << " sequences of total length " << b.size() << std::endl;
  std::cerr << "bwt_merge: Memory usage before merging: " << inGigabytes(memoryUsage()) << " GB" << std::endl;
  double verbose_start = readTimer();
#endif
  Alphabet merged = a.alpha;
  for(size_type c = 0; c <= merged.sigma; c++) { merged.C[c] += b.alpha.C[c]; }
  this->bwt.header.sequences = a.sequences() + b.sequences();
  this->bwt.header.bases = a.size() + b.size();
  this->bwt.header.setOrder(a.bwt.header.order());
  if(parameters.lazy_host)
  {
    // Device only: the result is the next merge's first input; nothing is encoded or downloaded now.
    bwtm_index* A = a.bwt.onDevice(a.alpha.C); bwtm_index* B = b.bwt.onDevice(b.alpha.C);
    bwtm_ra* ra = nullptr; bwtm_index* M = nullptr;
    gpuCheck(bwtm_ra_create(A, B, &ra), "FMI::FMI()");
    if(b.sequences() > 0) { gpuCheck(bwtm_search(A, B, 0, b.sequences() - 1, ra), "FMI::FMI()"); }
    gpuCheck(bwtm_ra_finalize(ra), "FMI::FMI()");
#ifdef VERBOSE_STATUS_INFO
    double verbose_mid = readTimer();
    std::cerr << "bwt_merge: RA built in " << (verbose_mid - verbose_start) << " seconds" << std::endl;
#endif
    gpuCheck(bwtm_interleave(A, B, ra, &M), "FMI::FMI()");
    bwtm_ra_free(ra);
    a.bwt.clear(); b.bwt.clear();
    this->bwt.adopt(M);
#ifdef VERBOSE_STATUS_INFO
    gpuCheck(bwtm_synchronize(), "FMI::FMI()");
    std::cerr << "bwt_merge: BWTs merged in " << (readTimer() - verbose_mid) << " seconds (the result stays on the device)" << std::endl;
#endif
  }
  else
  {
    // Host to host in one pipelined call (bwtm_merge_host): b is uploaded from its page-locked bytes, a likewise unless it
    // is already on the device (the result of the previous merge); data and samples land in this object's arrays.
    std::vector<uint64_t> cb(b.alpha.C.begin(), b.alpha.C.end()), ca(a.alpha.C.begin(), a.alpha.C.end());
    const BlockArray& bdata = b.bwt.hostData();
    bwtm_host_input hb = { bdata.data(), bdata.size(), b.sequences(), b.size(), cb.data() };
    bwtm_host_output out;
    bwtm_index* kept = nullptr;
    bwtm_upload* b_pending = (b.bwt.deviceResident() ? nullptr : b.bwt.releasePending());   // announced while the previous merge ran
    b.bwt.dropDevice();
    if(a.bwt.deviceResident() && b_pending)
    {
      gpuCheck(bwtm_merge_host_pipelined(a.bwt.releaseDevice(), nullptr, nullptr, b_pending, nullptr, nullptr, mergeSink, &this->bwt, BWTM_SAMPLES_COMPACT, &out, &kept),
        "FMI::FMI()");
    }
    else if(a.bwt.deviceResident())
    {
      gpuCheck(bwtm_merge_host_chained(a.bwt.releaseDevice(), &hb, mergeSink, &this->bwt, BWTM_SAMPLES_COMPACT, &out, &kept), "FMI::FMI()");
    }
    else
    {
      const BlockArray& adata = a.bwt.hostData();
      bwtm_host_input ha = { adata.data(), adata.size(), a.sequences(), a.size(), ca.data() };
      if(b_pending) { gpuCheck(bwtm_merge_host_pipelined(nullptr, &ha, nullptr, b_pending, nullptr, nullptr, mergeSink, &this->bwt, BWTM_SAMPLES_COMPACT, &out, &kept), "FMI::FMI()"); }
      else { gpuCheck(bwtm_merge_host(&ha, &hb, mergeSink, &this->bwt, BWTM_SAMPLES_COMPACT, &out, &kept), "FMI::FMI()"); }
    }
    this->bwt.adoptHost(kept, out.blocks, out.sample_width);
    a.bwt.clear(); b.bwt.clear();
#ifdef VERBOSE_STATUS_INFO
    std::cerr << "bwt_merge: RA built in " << (out.ms_upload + out.ms_search) / 1000.0 << " seconds" << std::endl;
    std::cerr << "bwt_merge: BWTs merged in " << (out.ms_interleave + out.ms_encode_download) / 1000.0 << " seconds" << std::endl;
    std::cerr << "bwt_merge: rank/select built in " << out.ms_samples / 1000.0 << " seconds" << std::endl;
#endif
  }
  this->alpha = merged;
  warnIfPoolExhausted("FMI::FMI()");
}

//------------------------------------------------------------------------------
// Formats.

template<>
inline void FMI::serialize<NativeFormat>(const std::string& filename) const
{
  std::ofstream out(filename.c_str(), std::ios_base::binary);
  if(!out) { std::cerr << "FMI::serialize(): Cannot open output file " << filename << std::endl; return; }
  bwt.serialize(out);
  // Alphabet: char2comp int_vector<8>, comp2char int_vector<8>, C int_vector<64>, u64 sigma.
  sdsl_compat::PackedVector c2c(alpha.char2comp.size(), 8), c2ch(alpha.comp2char.size(), 8), cc(alpha.C.size(), 64);
  for(size_type k = 0; k < alpha.char2comp.size(); k++) { c2c.set(k, alpha.char2comp[k]); }
  for(size_type k = 0; k < alpha.comp2char.size(); k++) { c2ch.set(k, alpha.comp2char[k]); }
  for(size_type k = 0; k < alpha.C.size(); k++) { cc.set(k, alpha.C[k]); }
  c2c.serialize(out, false); c2ch.serialize(out, false); cc.serialize(out, false);
  sdsl_compat::write_member(alpha.sigma, out);
}

template<>
inline void FMI::load<NativeFormat>(const std::string& filename)
{
  std::ifstream in(filename.c_str(), std::ios_base::binary);
  if(!in) { std::cerr << "FMI::load(): Cannot open input file " << filename << std::endl; std::exit(EXIT_FAILURE); }
  bwt.load(in);
  sdsl_compat::PackedVector c2c, c2ch, cc;
  c2c.load(in, false, 8); c2ch.load(in, false, 8); cc.load(in, false, 64);
  alpha.char2comp.resize(c2c.count); alpha.comp2char.resize(c2ch.count); alpha.C.resize(cc.count);
  for(size_type k = 0; k < c2c.count; k++) { alpha.char2comp[k] = (byte_type)c2c.get(k); }
  for(size_type k = 0; k < c2ch.count; k++) { alpha.comp2char[k] = (byte_type)c2ch.get(k); }
  for(size_type k = 0; k < cc.count; k++) { alpha.C[k] = cc.get(k); }
  sdsl_compat::read_member(alpha.sigma, in);
}

// Foreign formats (reference FMI::load / serialize<Format>, fmi.h:114-134; BWT::load<Format>, bwt.h:90-104).
// The file's items become maximal runs first (RunBuffer) and are mapped to comp values afterwards, like the
// reference does (formats.cpp:147-156: 'a' next to 'A' stays two runs); comp values outside the alphabet map to 0
// (the identity alphabet of formats.cpp:228, support.cpp:100-104).
template<class Format>
inline void FMI::load(const std::string& filename)
{
  std::ifstream in(filename.c_str(), std::ios_base::binary);
  if(!in) { std::cerr << "BWT::load(): Cannot open input file " << filename << std::endl; std::exit(EXIT_FAILURE); }
  const Alphabet order_alpha = createAlphabet(Format::order());
  bwt.data.clear();
  RunBuffer run_buffer;
  auto emit_run = [&](range_type run)
  {
    size_type comp = (Format::characters ? order_alpha.char2comp[run.first & 0xFF] : (run.first < BWT::SIGMA ? run.first : 0));
    Run::write(bwt.data, (comp_type)comp, run.second);
  };
  Format::decode(in, [&](size_type value, size_type length) { if(run_buffer.add(value, length)) { emit_run(run_buffer.run); } });
  run_buffer.flush(); emit_run(run_buffer.run);
  bwt.buildFromData(Format::order());
  std::vector<size_type> counts(BWT::SIGMA);
  for(size_type c = 0; c < BWT::SIGMA; c++) { counts[c] = bwt.count(c); }
  alpha = Alphabet(counts, order_alpha.char2comp, order_alpha.comp2char);
  bwt.header.setOrder(identifyAlphabet(alpha));
}

template<class Format>
inline void FMI::serialize(const std::string& filename) const
{
  if(!compatible(alpha, Format::order()))
  {
    std::cerr << "FMI::serialize(): Warning: " << Format::name() << " is not compatible with "
              << alphabetName(identifyAlphabet(alpha)) << " alphabets!" << std::endl;
  }
  std::ofstream out(filename.c_str(), std::ios_base::binary);
  if(!out) { std::cerr << "BWT::serialize(): Cannot open output file " << filename << std::endl; return; }
  const Alphabet order_alpha = createAlphabet(Format::order());
  Format::encode(out, bwt.header, [&](auto&& f)
  {
    const BlockArray& data = bwt.hostData();
    for(size_type rle_pos = 0; rle_pos < data.size(); )
    {
      range_type run = Run::read(data, rle_pos);
      f((size_type)(Format::characters ? order_alpha.comp2char[run.first] : run.first), run.second);
    }
  });
}

inline void serialize(const FMI& fmi, const std::string& filename, const std::string& format)
{
  if(!withFormat(format, [&](auto f) { fmi.serialize<decltype(f)>(filename); }))
  {
    std::cerr << "serialize(): Invalid BWT format: " << format << std::endl; std::exit(EXIT_FAILURE);
  }
}

inline void load(FMI& fmi, const std::string& filename, const std::string& format)
{
  if(!withFormat(format, [&](auto f) { fmi.load<decltype(f)>(filename); }))
  {
    std::cerr << "load(): Invalid BWT format: " << format << std::endl; std::exit(EXIT_FAILURE);
  }
}

// Bytes the native serialization takes (stands in for sdsl::size_in_bytes in the size report).
inline size_type sizeInBytes(const FMI& fmi)
{
  struct Counter : std::streambuf { size_type n = 0; std::streamsize xsputn(const char*, std::streamsize k) override { n += k; return k; } int overflow(int c) override { n++; return c; } };
  // Exact for data, header and alphabet; the Elias-Fano samples are estimated from their parameters
  // (2 + log2(n / m) bits per block), serializing them only to count bytes would take seconds.
  size_type blocks = fmi.bwt.blocks();
  size_type total = 24 + 8 + fmi.bwt.hostData().blocks() * BlockArray::BLOCK_SIZE + 256 + 6 + 7 * 8 + 8 + 3 * 8;
  for(size_type c = 0; c <= BWT::SIGMA; c++)
  {
    size_type universe = (c < BWT::SIGMA ? fmi.bwt.count((comp_type)c) + blocks : fmi.size());
    size_type per = 2 + (blocks > 0 && universe > blocks ? bit_length(universe / blocks) : 1);
    total += blocks * per / 8 + 64;
  }
  return total;
}

} // namespace bwtmerge

#endif // BWTM_HOST_FMI_H
