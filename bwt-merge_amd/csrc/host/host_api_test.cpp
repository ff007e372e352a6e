/*
  host_api_test -- exercises the C++ facade (FMI / BWT / RankArray / RunBuffer / formats) on the GPU.
  Usage: host_api_test a.plain b.plain expected_data.bin workdir
  `expected_data.bin` holds the native data bytes the oracle computed for merge(a, b).
  Exit code 0 = every check passed.
*/
#include <cstdio>
#include <thread>
#include "multi_gpu.h"

using namespace bwtmerge;
size_type Parallel::max_threads = 4;

static int failures = 0;
#define CHECK(cond) do { if(!(cond)) { std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); failures++; } } while(0)

static std::vector<byte_type> readFile(const std::string& name)
{
  std::ifstream in(name.c_str(), std::ios_base::binary);
  return std::vector<byte_type>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}

static std::vector<byte_type> symbolsOf(const FMI& fmi)
{
  std::vector<byte_type> out; fmi.bwt.extract(range_type(0, fmi.size() - 1), out); return out;
}

int main(int argc, char** argv)
{
  if(argc < 5) { std::fprintf(stderr, "usage: host_api_test a.plain b.plain expected_data.bin workdir\n"); return 2; }
  std::string work = argv[4];
  gpuCheck(bwtm_init(0), "host_api_test");

  // RunBuffer contract (utils.h:110-142 of the reference).
  {
    RunBuffer rb; std::vector<range_type> runs;
    size_type values[] = {0, 0, 3, 3, 3, 1, 3};
    for(size_type v : values) { if(rb.add(v)) { runs.push_back(rb.run); } }
    rb.flush(); runs.push_back(rb.run);
    CHECK(runs.size() == 4 && runs[0] == range_type(0, 2) && runs[1] == range_type(3, 3) && runs[2] == range_type(1, 1) && runs[3] == range_type(3, 1));
  }

  FMI a, b;
  load(a, argv[1], "plain_default"); load(b, argv[2], "plain_default");
  FMI a2 = a, b2 = b, a3 = a, b3 = b, b4 = b, b5 = b;
  FMI a6 = a, b6 = b, a7 = a, b7 = b, a8 = a, b8 = b, a9 = a, b9 = b, a10 = a, b10 = b;
  FMI a11 = a, b11 = b, a12 = a, b12 = b, a13 = a, b13 = b;                 // (the merges over partitioned records)
  std::vector<byte_type> sa = symbolsOf(a), sb = symbolsOf(b);
  size_type na = a.size(), nb = b.size();

  // Scalar queries against a plain scan.
  {
    size_type seen[6] = {};
    for(size_type i = 0; i <= na; i++)
    {
      if(i % 37 == 0 || i == na)
      {
        for(comp_type c = 0; c < 6; c++) { CHECK(a.bwt.rank(i, c) == seen[c]); }
        BWT::ranks_type r; a.bwt.ranks(i, r);
        for(size_type c = 1; c < 6; c++) { CHECK(r[c] == seen[c]); }
        if(i < na)
        {
          CHECK(a.bwt[i] == sa[i]);
          range_type is = a.bwt.inverse_select(i);
          CHECK(is.second == sa[i] && is.first == seen[sa[i]]);
          CHECK(a.LF(i).first == a.alpha.C[sa[i]] + seen[sa[i]]);
          if(seen[sa[i]] + 1 <= a.bwt.count(sa[i])) { CHECK(a.bwt.select(seen[sa[i]] + 1, sa[i]) == i); }
        }
      }
      if(i < na) { seen[sa[i]]++; }
    }
    std::vector<size_type> counts; a.bwt.characterCounts(counts);
    for(size_type c = 0; c < 6; c++) { CHECK(counts[c] == seen[c]); CHECK(a.alpha.C[c + 1] - a.alpha.C[c] == seen[c]); }
    size_type h = FNV_OFFSET_BASIS; for(byte_type s : sa) { h = fnv1a_hash(s, h); }
    CHECK(a.bwt.hash() == h);
  }

  // Pattern counts before the merge (what bwt_merge -v compares).
  std::vector<std::string> patterns = {"A", "ACG", "GATTACA", "TTTT", "N", "CGCG", ""};
  std::vector<size_type> before;
  for(const std::string& p : patterns)
  {
    range_type ra = a.find(p), rb = b.find(p);
    before.push_back((Range::empty(ra) ? 0 : Range::length(ra)) + (Range::empty(rb) ? 0 : Range::length(rb)));
  }

  // The merging constructor == FMI::FMI(a, b, parameters).
  FMI merged(a, b, MergeParameters());
  CHECK(a.bwt.bytes() == 0 && b.bwt.bytes() == 0 && !a.bwt.deviceResident() && !b.bwt.deviceResident());   // inputs are consumed
  CHECK(merged.bwt.deviceResident());                                      // the result is also there as the next merge's input
  CHECK(merged.size() == na + nb);
  std::vector<byte_type> expected = readFile(argv[3]);
  CHECK(merged.bwt.data.bytes == expected);
  for(size_type k = 0; k < patterns.size(); k++)
  {
    range_type r = merged.find(patterns[k]);
    CHECK((Range::empty(r) ? 0 : Range::length(r)) == before[k]);
  }
  {
    FMI rebuilt; rebuilt.bwt.data = merged.bwt.data; rebuilt.bwt.buildFromData();
    CHECK(merged.bwt.sample_width == 1);                                     // the merge downloaded the compact samples (reads: 8-bit fields)
    CHECK(rebuilt.bwt.blockEnds() == merged.bwt.blockEnds());
    for(size_type c = 0; c < 6; c++) { CHECK(rebuilt.bwt.cumulative(c) == merged.bwt.cumulative(c)); }
    // queries on the compact form == queries on the full form
    FMI expanded = merged; expanded.bwt.expandSamples();
    CHECK(expanded.bwt.sample_width == 8 && expanded.bwt.block_end == rebuilt.bwt.block_end && expanded.bwt.cum_flat == rebuilt.bwt.cum_flat);
    for(size_type i = 0; i <= merged.size(); i += 41)
    {
      for(comp_type c = 0; c < 6; c++) { CHECK(merged.bwt.rank(i, c) == expanded.bwt.rank(i, c)); }
      if(i < merged.size()) { CHECK(merged.bwt.inverse_select(i) == expanded.bwt.inverse_select(i) && merged.bwt[i] == expanded.bwt[i]); }
    }
    for(comp_type c = 0; c < 6; c++)
    {
      for(size_type k = 1; k <= merged.bwt.count(c); k += 97) { CHECK(merged.bwt.select(k, c) == expanded.bwt.select(k, c)); }
    }
    CHECK(rebuilt.bwt.header.sequences == merged.sequences() && rebuilt.bwt.header.bases == merged.size());
  }

  // The two-step API: buildRA over several sequence blocks + BWT(a, b, ra).
  {
    MergeParameters p; p.setSB(5);
    RankArray ra; buildRA(a2, b2, p, ra);
    BWT interleaved(a2.bwt, b2.bwt, ra);
    CHECK(interleaved.data.bytes == expected);
  }

  // Chaining (bwt_merge.cpp:167-173): a lazy merge leaves its result on the device; as the first input of the next merge it is
  // not uploaded again, and the outcome equals the one computed from host copies only.
  {
    MergeParameters lazy; lazy.lazy_host = true;
    FMI on_device(a3, b3, lazy);
    CHECK(on_device.bwt.deviceResident() && on_device.bwt.data.bytes.empty() && on_device.size() == na + nb);
    FMI host_copy = merged;                                               // a copy has no device side
    CHECK(!host_copy.bwt.deviceResident());
    FMI chained(on_device, b4, MergeParameters());                        // first input taken from the device
    FMI from_host(host_copy, b5, MergeParameters());                      // both inputs uploaded from the host
    CHECK(chained.size() == na + 2 * nb && chained.bwt.data.bytes == from_host.bwt.data.bytes);
    CHECK(chained.bwt.blockEnds() == from_host.bwt.blockEnds());
    for(size_type c = 0; c < 6; c++) { CHECK(chained.bwt.cumulative(c) == from_host.bwt.cumulative(c)); }
    CHECK(chained.alpha.C == from_host.alpha.C);
    FMI again = on_device;                                                // consumed above: empty
    CHECK(again.bwt.bytes() == 0);
  }

  // One host thread per GPU (multi_gpu.h), sequence blocks.  On this box the "GPUs" are contexts of GPU 0: 1, 2, 3 and 4 threads, each
  // searching its block of b's sequences and producing its range of the output; same bytes and samples as the single call.
  {
    std::vector<std::vector<int>> device_lists = { {0}, {0, 0}, {0, 0, 0}, {0, 0, 0, 0} };
    FMI* as[4] = { &a6, &a7, &a8, &a9 }; FMI* bs[4] = { &b6, &b7, &b8, &b9 };
    for(size_type k = 0; k < 4; k++)
    {
      FMI sharded; MultiGPUTimes times;
      const size_type input_bytes = as[k]->bwt.bytes() + bs[k]->bwt.bytes();
      mergeMultiGPU(*as[k], *bs[k], device_lists[k], sharded, &times, MultiGPUMode::SequenceBlocks);
      CHECK(sharded.bwt.data.bytes == expected);
      CHECK(sharded.bwt.blockEnds() == merged.bwt.blockEnds());
      for(size_type c = 0; c < 6; c++) { CHECK(sharded.bwt.cumulative(c) == merged.bwt.cumulative(c)); }
      CHECK(sharded.alpha.C == merged.alpha.C && sharded.size() == merged.size() && sharded.sequences() == merged.sequences());
      // sharded upload: every context receives 1 / G of the native bytes from the host (parts are rounded up to 256 bytes)
      const size_type G = device_lists[k].size();
      CHECK(times.host_bytes_gpu0 <= input_bytes / G + 512 && times.host_bytes_gpu0 + 512 * G >= input_bytes / G);
      CHECK(as[k]->bwt.bytes() == 0 && times.total > 0);
    }
  }

#ifdef BWTM_EXPERIMENTAL
  // The same with the sliced frontier search: three contexts, each advancing a slice of the sorted frontier.
  {
    FMI sliced; MultiGPUTimes times;
    mergeMultiGPU(a10, b10, std::vector<int>({0, 0, 0}), sliced, &times, MultiGPUMode::Sliced);
    CHECK(sliced.bwt.data.bytes == expected);
    CHECK(sliced.bwt.blockEnds() == merged.bwt.blockEnds());
    for(size_type c = 0; c < 6; c++) { CHECK(sliced.bwt.cumulative(c) == merged.bwt.cumulative(c)); }
  }
#endif
  // Partitioned records (what several GPUs run by default): contexts standing in for 1, 3 and 5 GPUs, each with windows transcoded from its
  // share of the bytes, the parts' step kernels reading each other's output buffers.
  {
    FMI* pa[3] = { &a11, &a12, &a13 }; FMI* pb[3] = { &b11, &b12, &b13 };
    const size_type counts[3] = { 1, 3, 5 };
    for(size_type k = 0; k < 3; k++)
    {
      const size_type input_bytes = pa[k]->bwt.bytes() + pb[k]->bwt.bytes();
      FMI part; MultiGPUTimes times;
      mergeMultiGPU(*pa[k], *pb[k], std::vector<int>(counts[k], 0), part, &times, (k == 2 ? MultiGPUMode::Auto : MultiGPUMode::Partitioned));
      CHECK(part.bwt.data.bytes == expected);
      CHECK(part.bwt.blockEnds() == merged.bwt.blockEnds());
      for(size_type c = 0; c < 6; c++) { CHECK(part.bwt.cumulative(c) == merged.bwt.cumulative(c)); }
      CHECK(part.alpha.C == merged.alpha.C && part.size() == merged.size() && part.sequences() == merged.sequences());
      CHECK(times.host_bytes_gpu0 <= input_bytes && times.total > 0);     // a part uploads the blocks of its windows, never more than the inputs
    }
  }

  // Native file round trip.
  {
    std::string name = work + "/merged.native";
    serialize(merged, name, "native");
    FMI back; load(back, name, "native");
    CHECK(back.bwt.data.bytes == merged.bwt.data.bytes);
    CHECK(back.bwt.blockEnds() == merged.bwt.blockEnds());
    for(size_type c = 0; c < 6; c++) { CHECK(back.bwt.cumulative(c) == merged.bwt.cumulative(c)); }
    CHECK(back.alpha == merged.alpha && back.alpha.C == merged.alpha.C);
    CHECK(back.bwt.header.sequences == merged.sequences() && back.bwt.header.bases == merged.size() && back.bwt.header.check());
    CHECK(back.bwt.hash() == merged.bwt.hash());
    std::vector<byte_type> raw = readFile(name);
    CHECK(raw.size() >= 24 + 8 + BlockArray::BLOCK_SIZE);               // whole 8 MiB blocks (reference support.cpp:302-306)
    std::uint32_t tag; std::memcpy(&tag, raw.data(), 4); CHECK(tag == 0x54574221);
  }

  if(failures == 0) { std::printf("host_api_test: all checks passed\n"); return 0; }
  std::fprintf(stderr, "host_api_test: %d checks failed\n", failures);
  return 1;
}
