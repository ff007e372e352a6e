/*
  host_cpu_test -- the parts of the C++ facade that need no GPU: Run / ByteCode / RunBuffer / getBounds, BWT queries on the full
  and on the COMPACT form of the samples (the form a merge downloads: 8- / 16- / 32-bit per-block fields + anchors every 64 blocks),
  expandSamples(), native serialization round trip.  Exit code 0 = every check passed.
*/
#include <cstdio>
#include <random>
#include "fmi.h"

using namespace bwtmerge;
size_type Parallel::max_threads = 2;

static int failures = 0;
#define CHECK(cond) do { if(!(cond)) { std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); failures++; } } while(0)

// Installs the compact form of `full`'s samples into `target` (what bwtm_index_download_samples_compact delivers).
static void installCompact(const BWT& full, BWT& target, int width)
{
  const size_type nb = full.blocks(), nanch = (nb + 63) / 64;
  target.cum_stride = nb + 1; target.sample_width = width;
  target.anchors.resize(BWT::SIGMA * nanch);
  target.fields.resize((BWT::SIGMA * nb * (size_type)width + 1) / 2);
  for(size_type k = 0; k < nb; k++)
  {
    size_type start = (k == 0 ? 0 : full.block_end[k - 1] + 1), length = full.block_end[k] + 1 - start;
    for(size_type c = 0; c < BWT::SIGMA; c++)
    {
      size_type value = (c == 0 ? length : full.cum(c, k + 1) - full.cum(c, k));
      if(width == 1) { ((std::uint8_t*)target.fields.data())[c * nb + k] = (std::uint8_t)value; }
      else if(width == 2) { target.fields[c * nb + k] = (std::uint16_t)value; }
      else { ((std::uint32_t*)target.fields.data())[c * nb + k] = (std::uint32_t)value; }
      if(k % 64 == 0) { target.anchors[c * nanch + k / 64] = (c == 0 ? start : full.cum(c, k)); }
    }
  }
  target.block_end.clear(); target.cum_flat.clear();
}

int main()
{
  // Run::write block rule against the reference's vectors: len 170 at offset 62, len 16426 at offset 61.
  {
    BlockArray a; for(int k = 0; k < 62; k++) { a.push_back(0); }
    Run::write(a, 3, 170);
    CHECK(a.size() == 65 && a[62] == 0xf9 && a[63] == 0x7f && a[64] == 0x03);
    BlockArray b; for(int k = 0; k < 61; k++) { b.push_back(0); }
    Run::write(b, 3, 16426);
    CHECK(b.size() == 65 && b[61] == 0xf9 && b[62] == 0xff && b[63] == 0x7f && b[64] == 0x03);
    size_type pos = 61; range_type r1 = Run::read(b, pos), r2 = Run::read(b, pos);
    CHECK(r1 == range_type(3, 16425) && r2 == range_type(3, 1) && pos == 65);
  }
  CHECK(getBounds(range_type(0, 9), 4) == std::vector<range_type>({range_type(0, 1), range_type(2, 3), range_type(4, 6), range_type(7, 9)}));

  std::mt19937_64 rng(5);
  for(int variant = 0; variant < 4; variant++)
  {
    // a run-structured string: short runs (variant 0), runs up to 70000 (variant 1: needs 32-bit fields), tiny (variant 2)
    FMI full;
    std::vector<byte_type> symbols;
    size_type nruns = (variant == 2 ? 5 : 30000);
    comp_type previous = 6;
    for(size_type k = 0; k < nruns; k++)
    {
      comp_type c = (comp_type)(rng() % 6); if(c == previous) { c = (comp_type)((c + 1) % 6); } previous = c;
      size_type choices0[] = {1, 1, 2, 3, 41, 42, 170}, choices1[] = {1, 2, 50, 70000, 3};
      size_type len = (variant == 1 ? choices1[rng() % 5] : (variant == 3 ? choices0[rng() % 4] : choices0[rng() % 7]));   // variant 3: runs of 1 .. 3, 8-bit fields
      Run::write(full.bwt.data, c, len);
      if(symbols.size() < 400000) { for(size_type j = 0; j < len && symbols.size() < 400000; j++) { symbols.push_back(c); } }
    }
    full.bwt.buildFromData();
    const size_type n = full.bwt.size();
    int width = (variant == 1 ? 4 : (variant == 3 ? 1 : 2));
    FMI compact; compact.bwt.header = full.bwt.header; compact.bwt.data = full.bwt.data;
    installCompact(full.bwt, compact.bwt, width);
    CHECK(compact.bwt.blocks() == full.bwt.blocks() && compact.bwt.blockEnds() == full.bwt.blockEnds());
    for(size_type c = 0; c < 6; c++) { CHECK(compact.bwt.cumulative(c) == full.bwt.cumulative(c)); CHECK(compact.bwt.count((comp_type)c) == full.bwt.count((comp_type)c)); }
    // queries on both forms against each other and against the plain prefix of the string
    size_type seen[6] = {};
    for(size_type i = 0; i <= std::min<size_type>(n, symbols.size()); i++)
    {
      if(i % 97 == 0 || i + 1 >= symbols.size())
      {
        for(comp_type c = 0; c < 6; c++) { CHECK(full.bwt.rank(i, c) == seen[c]); CHECK(compact.bwt.rank(i, c) == seen[c]); }
        if(i < n && i < symbols.size())
        {
          CHECK(compact.bwt[i] == symbols[i] && compact.bwt.inverse_select(i) == full.bwt.inverse_select(i));
          CHECK(compact.bwt.select(seen[symbols[i]] + 1, symbols[i]) == i);
        }
      }
      if(i < symbols.size()) { seen[symbols[i]]++; }
    }
    for(size_type t = 0; t < 2000; t++)
    {
      size_type i = rng() % (n + 1); comp_type c = (comp_type)(rng() % 6);
      CHECK(compact.bwt.rank(i, c) == full.bwt.rank(i, c));
      size_type cnt = full.bwt.count(c);
      if(cnt > 0) { size_type k = 1 + rng() % cnt; CHECK(compact.bwt.select(k, c) == full.bwt.select(k, c)); }
    }
    // expansion gives back the full arrays; serialization of either form gives the same file
    FMI expanded = compact; expanded.bwt.expandSamples();
    CHECK(expanded.bwt.sample_width == 8 && expanded.bwt.block_end == full.bwt.block_end && expanded.bwt.cum_flat == full.bwt.cum_flat);
    std::vector<size_type> counts(6); for(size_type c = 0; c < 6; c++) { counts[c] = full.bwt.count((comp_type)c); }
    full.alpha = Alphabet(counts); compact.alpha = full.alpha;
    std::string f1 = "/tmp/bwtm_host_cpu_test_full.native", f2 = "/tmp/bwtm_host_cpu_test_compact.native";
    serialize(full, f1, "native"); serialize(compact, f2, "native");
    std::ifstream i1(f1, std::ios_base::binary), i2(f2, std::ios_base::binary);
    std::vector<char> b1((std::istreambuf_iterator<char>(i1)), std::istreambuf_iterator<char>()), b2((std::istreambuf_iterator<char>(i2)), std::istreambuf_iterator<char>());
    CHECK(!b1.empty() && b1 == b2);
    FMI back; load(back, f2, "native");
    CHECK(back.bwt.data.bytes == full.bwt.data.bytes && back.bwt.block_end == full.bwt.block_end && back.bwt.cum_flat == full.bwt.cum_flat && back.alpha.C == full.alpha.C);
    std::remove(f1.c_str()); std::remove(f2.c_str());
  }

  if(failures == 0) { std::printf("host_cpu_test: all checks passed\n"); return 0; }
  std::fprintf(stderr, "host_cpu_test: %d checks failed\n", failures);
  return 1;
}
