/*
  bwt_ingest -- builds the BWT of a read collection on the GPU.  The reference has no such tool: its
  inputs come from RopeBWT / SGA (README.md:5,20).  Input: text, one read per line over ACGTN (other characters map to N,
  like the default alphabet, support.cpp:66-81); reads keep their order, equal suffixes are ordered by read -- the BWT a
  chain of bwt_merge runs over the single reads would produce.  Output: any format of formats.h (default native).
*/
#include <unistd.h>

#include "fmi.h"

using namespace bwtmerge;

size_type Parallel::max_threads = std::max(1u, std::thread::hardware_concurrency());

static void printUsage()
{
  std::cerr << "Usage: bwt_ingest [options] reads output" << std::endl << std::endl;
  std::cerr << "Options:" << std::endl;
  std::cerr << "  -g N           Use GPU N (default: 0)" << std::endl;
  std::cerr << "  -l N           Sort N reads per leaf (default: 524288)" << std::endl;
  std::cerr << "  -o format      Write the output in the given format (default: native)" << std::endl << std::endl;
  printFormats(std::cerr);
}

int main(int argc, char** argv)
{
  if(argc < 2) { printUsage(); std::exit(EXIT_SUCCESS); }

  std::cout << "BWT ingest" << std::endl << std::endl;

  int device = 0;
  size_type leaf_reads = (size_type)1 << 19;
  std::string output_tag = NativeFormat::tag();
  for(int c = 0; (c = getopt(argc, argv, "g:l:o:")) != -1; )
  {
    switch(c)
    {
    case 'g': device = std::stoi(optarg); break;
    case 'l': leaf_reads = std::max<size_type>(1, std::stoul(optarg)); break;
    case 'o':
      output_tag = optarg;
      if(!formatExists(output_tag)) { std::cerr << "bwt_ingest: Invalid output format: " << output_tag << std::endl; std::exit(EXIT_FAILURE); }
      break;
    default: std::exit(EXIT_FAILURE);
    }
  }
  if(optind + 1 >= argc) { std::cerr << "bwt_ingest: Output file not specified" << std::endl; std::exit(EXIT_FAILURE); }
  std::string input_name = argv[optind], output_name = argv[optind + 1];
  std::cout << "Input:   " << input_name << " (reads, one per line)" << std::endl;
  std::cout << "Output:  " << output_name << " (" << output_tag << ")" << std::endl << std::endl;

  double start = readTimer();
  gpuCheck(bwtm_init(device), "bwt_ingest");
  std::ifstream in(input_name.c_str(), std::ios_base::binary);
  if(!in) { std::cerr << "bwt_ingest: Cannot open input file " << input_name << std::endl; std::exit(EXIT_FAILURE); }

  bwtm_builder* builder = nullptr;
  gpuCheck(bwtm_builder_create(leaf_reads, &builder), "bwt_ingest");
  const Alphabet alpha;                                 // $ACGTN
  std::vector<std::string> batch;
  HostArray<byte_type> rows; HostArray<std::uint32_t> lengths;
  size_type reads = 0, symbols = 0;
  auto flush = [&]()
  {
    if(batch.empty()) { return; }
    size_type width = 1;
    for(const std::string& r : batch) { width = std::max<size_type>(width, r.size()); }
    rows.assign(batch.size() * width, 0); lengths.resizeUninitialized(batch.size());
    for(size_type k = 0; k < batch.size(); k++)
    {
      lengths[k] = (std::uint32_t)batch[k].size();
      byte_type* row = rows.data() + k * width;
      for(size_type j = 0; j < batch[k].size(); j++)
      {
        byte_type comp = alpha.char2comp[(byte_type)batch[k][j]];
        row[j] = (comp == 0 ? 5 : comp);                 // an endmarker character inside a read is not a symbol
      }
    }
    gpuCheck(bwtm_builder_add(builder, rows.data(), batch.size(), (std::uint32_t)width, width, lengths.data(), 0), "bwt_ingest");
    batch.clear();
  };
  for(std::string line; std::getline(in, line); )
  {
    if(!line.empty() && line.back() == '\r') { line.pop_back(); }
    reads++; symbols += line.size();
    batch.push_back(std::move(line));
    if(batch.size() >= leaf_reads) { flush(); }
  }
  flush();
  std::cout << "Read " << reads << " reads of total length " << symbols << std::endl << std::endl;

  bwtm_index* built = nullptr;
  gpuCheck(bwtm_builder_finish(builder, &built), "bwt_ingest");
  FMI fmi;
  std::vector<uint64_t> C(BWTM_SIGMA + 1, 0);
  bwtm_index_C(built, C.data());
  std::vector<size_type> counts(BWT::SIGMA);
  for(size_type c = 0; c < BWT::SIGMA; c++) { counts[c] = C[c + 1] - C[c]; }
  fmi.alpha = Alphabet(counts);
  fmi.bwt.header.sequences = bwtm_index_sequences(built);
  fmi.bwt.header.bases = bwtm_index_bases(built);
  fmi.bwt.header.setOrder(AO_DEFAULT);
  fmi.bwt.adopt(built);
  size_type size = fmi.size();
  printSize("FMI", sizeInBytes(fmi), fmi.size());
  std::cout << std::endl;
  serialize(fmi, output_name, output_tag);
  double seconds = readTimer() - start;

  std::cout << "BWT built in " << seconds << " seconds (" << (inMegabytes(size) / seconds) << " MB/s)" << std::endl << std::endl;
  std::cout << "Memory usage: " << inGigabytes(memoryUsage()) << " GB" << std::endl << std::endl;
  return 0;
}
