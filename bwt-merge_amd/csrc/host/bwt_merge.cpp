/*
  bwt_merge -- merges BWT files, with the hot path on an MI355X.  Same command line and the
  same stdout lines as the reference tool (bwt_merge.cpp:47-299); all formats of formats.h.
*/
#include <atomic>
#include <sstream>
#include <unistd.h>

#include "multi_gpu.h"

using namespace bwtmerge;

size_type Parallel::max_threads = std::max(1u, std::thread::hardware_concurrency());

static void printUsage()
{
  std::cerr << "Usage: bwt_merge [options] input1 input2 [input3 ...] output" << std::endl << std::endl;
  std::cerr << "Options:" << std::endl;
  std::cerr << "  -b N          Set thread buffer size to N megabytes / thread (default: " << MergeParameters::defaultTB() << ")" << std::endl;
  std::cerr << "  -m N          Set the number of merge buffers to N (default: " << MergeParameters::defaultMB() << ")" << std::endl;
  std::cerr << "  -r N          Set run buffer size to N megabytes / thread (default: " << MergeParameters::defaultRB() << ")" << std::endl;
  std::cerr << "  -s N          Set the number of sequence blocks to N (default: " << MergeParameters::defaultSB() << " / thread; the device partitions the search itself)" << std::endl;
  std::cerr << "  -t N          Use N parallel threads (default: " << MergeParameters::defaultT() << " on this system)" << std::endl << std::endl;
  std::cerr << "  -d directory  Use the given directory for temporary files (default: .)" << std::endl;
  std::cerr << "  -v filename   Verify by querying with patterns from the given file" << std::endl << std::endl;
  std::cerr << "  -i formats    Read the inputs in the given formats (default: native)" << std::endl;
  std::cerr << "                Multiple comma-separated formats can be provided." << std::endl;
  std::cerr << "  -o format     Write the output in the given format (default: native)" << std::endl;
  std::cerr << "  -g N[,M,...]  Use GPU N (default: 0), or one host thread per listed GPU (at most 16).  With several GPUs the records are" << std::endl;
  std::cerr << "                PARTITIONED: every GPU holds one window of each input and of the bitvector, transcoded from its share" << std::endl;
  std::cerr << "                of the bytes, and produces its range of the output; the frontier's elements travel between the GPUs" << std::endl;
  std::cerr << "                (the buffer options have no effect on the device)" << std::endl;
  std::cerr << "  -B            With several GPUs: sequence blocks instead (replicated records, every GPU searches a block of the" << std::endl;
  std::cerr << "                increment's sequences, one reduce-scatter of the rank-array bitvector by output range)" << std::endl;
#ifdef BWTM_EXPERIMENTAL
  std::cerr << "  -S            With several GPUs: sliced search (replicated records, every GPU advances a contiguous slice of the" << std::endl;
  std::cerr << "                sorted frontier; experimental build only)" << std::endl;
#endif
  std::cerr << std::endl;
  printFormats(std::cerr);
}

static size_type readRows(const std::string& filename, std::vector<std::string>& rows)
{
  std::ifstream in(filename.c_str(), std::ios_base::binary);
  if(!in) { std::cerr << "readRows(): Cannot open input file " << filename << std::endl; return 0; }
  size_type chars = 0;
  for(std::string line; std::getline(in, line); ) { if(!line.empty()) { rows.push_back(line); chars += line.length(); } }
  return chars;
}

// Adds the number of occurrences of every pattern to `results` (reference verifyFMI / queryFMI,
// bwt_merge.cpp:240-285).  The backward searches run on the GPU in one batch.
static void verifyFMI(const FMI& fmi, const std::string& name, const std::vector<std::string>& patterns, std::vector<size_type>& results)
{
  size_type chars = 0;
  for(const std::string& p : patterns) { chars += p.length(); }
  printSize(name, sizeInBytes(fmi), fmi.size());
  if(chars > 0)
  {
    double start = readTimer();
    std::vector<byte_type> text; text.reserve(chars);
    std::vector<uint64_t> offsets(patterns.size() + 1, 0), sp(patterns.size()), ep(patterns.size());
    for(size_type k = 0; k < patterns.size(); k++)
    {
      for(char ch : patterns[k]) { text.push_back(fmi.alpha.char2comp[(byte_type)ch]); }
      offsets[k + 1] = text.size();
    }
    bwtm_index* ix = fmi.bwt.onDevice(fmi.alpha.C);                 // stays there: the merge that follows uses the same copy
    gpuCheck(bwtm_find_batch(ix, text.data(), offsets.data(), patterns.size(), sp.data(), ep.data()), "verifyFMI()");
    size_type found = 0, matches = 0;
    for(size_type k = 0; k < patterns.size(); k++)
    {
      range_type range(sp[k], ep[k]);
      results[k] += Range::length(range);
      if(!Range::empty(range)) { found++; matches += Range::length(range); }
    }
    printTime(name, found, matches, chars, readTimer() - start);
  }
  std::cout << std::endl;
}

static MultiGPUMode multi_gpu_mode = MultiGPUMode::Auto;      // several GPUs: partitioned records; -B sequence blocks, -S sliced search (multi_gpu.h)
static const char* modeName(MultiGPUMode m)
{
  return (m == MultiGPUMode::SequenceBlocks ? " (sequence blocks)" : (m == MultiGPUMode::Sliced ? " (sliced search)" : " (partitioned records)"));
}

static void merge(FMI& index, FMI& increment, const MergeParameters& parameters, const std::vector<int>& devices)
{
  double increment_mb = inMegabytes(increment.size());
  double start = readTimer();
  if(devices.size() > 1)
  {
    FMI temp; MultiGPUTimes times;
    mergeMultiGPU(index, increment, devices, temp, &times, multi_gpu_mode);      // one host thread per GPU, result assembled on the host
    index.swap(temp);
#ifdef VERBOSE_STATUS_INFO
    // the phases of the sharded merge as GPU 0's thread saw them (stderr, like the reference's status lines): what a SCALE session reads
    std::cerr << "mergeMultiGPU(): " << devices.size() << " GPUs" << modeName(multi_gpu_mode) << ": upload " << times.upload << " s, search " << times.search
              << " s, exchange " << times.exchange << " s, interleave + encode " << times.interleave_encode << " s, download " << times.download
              << " s, total " << times.total << " s; exchanged " << times.exchange_bytes << " bytes per GPU; host bytes to GPU 0 " << times.host_bytes_gpu0 << std::endl;
#endif
  }
  else
  {
    FMI temp(index, increment, parameters);
    index.swap(temp);
  }
  double seconds = readTimer() - start;
  std::cout << "BWTs merged in " << seconds << " seconds (" << (increment_mb / seconds) << " MB/s)" << std::endl << std::endl;
}

int main(int argc, char** argv)
{
  if(argc < 2) { printUsage(); std::exit(EXIT_SUCCESS); }

  double start = readTimer();
  std::cout << "BWT-merge" << std::endl << std::endl;

  int c = 0;
  std::vector<int> devices;
  bool verify = false;
  MergeParameters parameters;
  std::string pattern_name, output_format;
  std::vector<std::string> input_formats;
  while((c = getopt(argc, argv,
#ifdef BWTM_EXPERIMENTAL
    "b:m:r:s:t:d:v:i:o:g:BPS"
#else
    "b:m:r:s:t:d:v:i:o:g:BP"
#endif
    )) != -1)
  {
    switch(c)
    {
    case 'b': parameters.setTB(std::stoul(optarg)); break;
    case 'm': parameters.setMB(std::stoul(optarg)); break;
    case 'r': parameters.setRB(std::stoul(optarg)); break;
    case 's': parameters.setSB(std::stoul(optarg)); break;
    case 't': parameters.setT(std::stoul(optarg)); break;
    case 'd': parameters.setTemp(optarg); break;
    case 'g':
      {
        std::istringstream ss(optarg);
        for(std::string token; std::getline(ss, token, ','); ) { devices.push_back(std::stoi(token)); }
      }
      break;
    case 'B': multi_gpu_mode = MultiGPUMode::SequenceBlocks; break;
    case 'P': multi_gpu_mode = MultiGPUMode::Partitioned; break;      // the default, without the fall-back to sequence blocks
#ifdef BWTM_EXPERIMENTAL
    case 'S': multi_gpu_mode = MultiGPUMode::Sliced; break;
#endif
    case 'v': pattern_name = optarg; verify = true; break;
    case 'i':
      {
        std::istringstream ss(optarg);
        for(std::string token; std::getline(ss, token, ','); ) { input_formats.push_back(token); }
        for(const std::string& f : input_formats)
        {
          if(!formatExists(f)) { std::cerr << "bwt_merge: Invalid input format: " << f << std::endl; std::exit(EXIT_FAILURE); }
        }
      }
      break;
    case 'o':
      output_format = optarg;
      if(!formatExists(output_format)) { std::cerr << "bwt_merge: Invalid output format: " << output_format << std::endl; std::exit(EXIT_FAILURE); }
      break;
    default: std::exit(EXIT_FAILURE);
    }
  }

  int inputs = (argc - 1) - optind;
  if(inputs < 2) { std::cerr << "bwt_merge: Output file not specified" << std::endl; std::exit(EXIT_FAILURE); }
  if(input_formats.empty()) { input_formats.assign(inputs, NativeFormat::tag()); }
  if(input_formats.size() == 1) { input_formats.assign(inputs, input_formats[0]); }
  if(input_formats.size() != (size_type)inputs)
  {
    std::cerr << "bwt_merge: Specified " << input_formats.size() << " formats for " << inputs << " inputs" << std::endl;
    std::exit(EXIT_FAILURE);
  }
  if(output_format.empty()) { output_format = NativeFormat::tag(); }
  parameters.sanitize();
  Parallel::max_threads = parameters.threads;

  for(int i = optind; i < argc - 1; i++) { std::cout << "Input:            " << argv[i] << " (" << input_formats[i - optind] << ")" << std::endl; }
  std::cout << "Output:           " << argv[argc - 1] << " (" << output_format << ")" << std::endl;
  if(verify) { std::cout << "Patterns:         " << pattern_name << std::endl; }
  std::cout << std::endl << parameters << std::endl;

  if(devices.empty()) { devices.push_back(0); }
  gpuCheck(bwtm_init(devices[0]), "bwt_merge");

  std::vector<std::string> patterns;
  std::vector<size_type> pre_results, post_results;
  if(verify)
  {
    size_type chars = readRows(pattern_name, patterns);
    pre_results.assign(patterns.size(), 0); post_results.assign(patterns.size(), 0);
    std::cout << "Read " << patterns.size() << " patterns of total length " << chars << std::endl << std::endl;
  }

  FMI index; load(index, argv[optind], input_formats[0]);
  verifyFMI(index, "Input", patterns, pre_results);

  // A chain is software-pipelined: input k + 1 is loaded and ANNOUNCED to the device (BWT::prefetchDevice: its bytes start
  // travelling on the copy stream) before input k is merged, so the transfer runs under that merge's search.
  size_type bytes_added = 0;
  FMI increment; load(increment, argv[optind + 1], input_formats[1]);
  for(int input = 1; input < inputs; input++)
  {
    bytes_added += increment.size();
    verifyFMI(increment, "Input", patterns, pre_results);
    FMI next;
    const bool prefetch = (input + 1 < inputs && devices.size() == 1);
    if(prefetch)
    {
      // one GPU: the next input is read early so that its bytes can travel under this merge's search; this merge's own inputs go
      // first on the link (the first merge of a chain uploads both of them)
      load(next, argv[optind + input + 1], input_formats[input + 1]);
      index.bwt.onDevice(index.alpha.C); increment.bwt.onDevice(increment.alpha.C);
      next.bwt.prefetchDevice(next.alpha.C);
    }
    // Intermediate results of a chain are only ever the next merge's first input: they stay on the device.  The last
    // merge produces the host-resident FMI inside its timer, like the reference's.
    MergeParameters p = parameters; p.lazy_host = (input + 1 < inputs);
    merge(index, increment, p, devices);
    // several GPUs: nothing overlaps with the load, so the next input is read only now (one input less in host memory during the merge)
    if(input + 1 < inputs && !prefetch) { load(next, argv[optind + input + 1], input_formats[input + 1]); }
    increment.swap(next);
  }

  serialize(index, argv[argc - 1], output_format);
  verifyFMI(index, "Output", patterns, post_results);

  if(verify)
  {
    size_type errors = 0;
    for(size_type i = 0; i < patterns.size(); i++) { if(pre_results[i] != post_results[i]) { errors++; } }
    if(errors > 0) { std::cout << "Verification failed for " << errors << " patterns" << std::endl; }
    else { std::cout << "Verification successful" << std::endl; }
    std::cout << std::endl;
  }

  double seconds = readTimer() - start;
  std::cout << "Total time:       " << seconds << " seconds (" << (inMegabytes(bytes_added) / seconds) << " MB/s)" << std::endl;
  std::cout << "Peak memory:      " << inGigabytes(memoryUsage()) << " GB" << std::endl << std::endl;
  return 0;
}
