/*
  bwt_convert -- converts a BWT file between the formats of formats.h.  Same command line and the same
  stdout lines as the reference tool (bwt_convert.cpp:36-125).  Pure host code: the codecs are serial
  byte work next to the hot path; no GPU is needed.
*/
#include <unistd.h>

#include "fmi.h"

using namespace bwtmerge;

size_type Parallel::max_threads = std::max(1u, std::thread::hardware_concurrency());

static void printUsage()
{
  std::cerr << "Usage: bwt_convert [options] input output" << std::endl << std::endl;
  std::cerr << "Options:" << std::endl;
  std::cerr << "  -i format      Read the input in the given format (default: sga)" << std::endl;
  std::cerr << "  -o format      Write the output in the given format (default: native)" << std::endl << std::endl;
  printFormats(std::cerr);
}

int main(int argc, char** argv)
{
  if(argc < 2) { printUsage(); std::exit(EXIT_SUCCESS); }

  std::cout << "BWT converter" << std::endl << std::endl;

  std::string input_tag = SGAFormat::tag(), output_tag = NativeFormat::tag();
  for(int c = 0; (c = getopt(argc, argv, "i:o:")) != -1; )
  {
    std::string* target = (c == 'i' ? &input_tag : (c == 'o' ? &output_tag : nullptr));
    if(target == nullptr) { std::exit(EXIT_FAILURE); }
    *target = optarg;
    if(!formatExists(*target))
    {
      std::cerr << "bwt_convert: Invalid " << (c == 'i' ? "input" : "output") << " format: " << *target << std::endl;
      std::exit(EXIT_FAILURE);
    }
  }
  if(optind + 1 >= argc) { std::cerr << "bwt_convert: Output file not specified" << std::endl; std::exit(EXIT_FAILURE); }
  std::string input_name = argv[optind], output_name = argv[optind + 1];

  std::cout << "Input:   " << input_name << " (" << input_tag << ")" << std::endl;
  std::cout << "Output:  " << output_name << " (" << output_tag << ")" << std::endl << std::endl;

  double start = readTimer();
  FMI fmi; load(fmi, input_name, input_tag);
  size_type size = fmi.size();
  printSize("FMI", sizeInBytes(fmi), fmi.size());
  std::cout << std::endl;
  serialize(fmi, output_name, output_tag);
  double seconds = readTimer() - start;

  std::cout << "BWT converted in " << seconds << " seconds (" << (inMegabytes(size) / seconds) << " MB/s)" << std::endl << std::endl;
  std::cout << "Memory usage: " << inGigabytes(memoryUsage()) << " GB" << std::endl << std::endl;
  return 0;
}
