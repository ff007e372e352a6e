/*
  utils.h -- host facade, utilities.  Same names and semantics as the reference's utils.h
  (types utils.h:44-47, Range 73-99, RunBuffer 121-142, bit_length 146-151, FNV-1a 155-176,
  getBounds utils.cpp:169-187, timers utils.cpp:79-96), written for this repository: no SDSL.
*/
#ifndef BWTM_HOST_UTILS_H
#define BWTM_HOST_UTILS_H

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <iostream>
#include <string>
#include <utility>
#include <vector>

#include <sys/resource.h>

namespace bwtmerge
{

typedef std::uint64_t size_type;
typedef std::uint8_t  char_type;
typedef std::uint8_t  comp_type;
typedef std::uint8_t  byte_type;

const size_type KILOBYTE = 1024;
const size_type MEGABYTE = KILOBYTE * KILOBYTE;
const size_type GIGABYTE = KILOBYTE * MEGABYTE;

typedef std::pair<size_type, size_type> range_type;    // closed range; empty when first > second

struct Range
{
  static size_type length(range_type r) { return r.second + 1 - r.first; }
  static bool empty(range_type r) { return r.first + 1 > r.second + 1; }
  static size_type bound(size_type v, size_type low, size_type high) { return std::max(std::min(v, high), low); }
  static range_type empty_range() { return range_type(1, 0); }
};

template<class A, class B>
std::ostream& operator<<(std::ostream& out, const std::pair<A, B>& p) { return out << "(" << p.first << ", " << p.second << ")"; }

/*
  Turns a stream of values / runs into maximal runs (public API of the reference, kept):
    RunBuffer buffer;
    while(...) { if(buffer.add(...)) { use(buffer.run); } }
    buffer.flush(); use(buffer.run);
*/
struct RunBuffer
{
  RunBuffer() : value(0), length(0), run(0, 0) {}

  bool add(size_type v, size_type n = 1)
  {
    if(v == value) { length += n; return false; }
    flush();
    value = v; length = n;
    return run.second > 0;
  }
  bool add(range_type r) { return add(r.first, r.second); }
  void flush() { run = range_type(value, length); }

  size_type  value, length;
  range_type run;
};

inline size_type bit_length(size_type v) { return (v == 0 ? 1 : 64 - (size_type)__builtin_clzll(v)); }

const size_type FNV_OFFSET_BASIS = 0xcbf29ce484222325ULL;
const size_type FNV_PRIME        = 0x100000001b3ULL;
inline size_type fnv1a_hash(byte_type b, size_type seed) { return (seed ^ b) * FNV_PRIME; }

inline double inMegabytes(size_type bytes) { return bytes / (double)MEGABYTE; }
inline double inGigabytes(size_type bytes) { return bytes / (double)GIGABYTE; }
inline double inBPC(size_type bytes, size_type size) { return (8.0 * bytes) / size; }

const size_type DEFAULT_INDENT = 18;

inline void printHeader(const std::string& header, size_type indent = DEFAULT_INDENT)
{
  std::string padding;
  if(header.length() + 1 < indent) { padding = std::string(indent - 1 - header.length(), ' '); }
  std::cout << header << ":" << padding;
}

inline void printSize(const std::string& header, size_type bytes, size_type data_size, size_type indent = DEFAULT_INDENT)
{
  printHeader(header, indent);
  std::cout << inMegabytes(bytes) << " MB (" << inBPC(bytes, data_size) << " bpc)" << std::endl;
}

inline void printTime(const std::string& header, size_type found, size_type matches, size_type bytes, double seconds, size_type indent = DEFAULT_INDENT)
{
  printHeader(header, indent);
  std::cout << "Found " << found << " patterns with " << matches << " occ in " << seconds << " seconds ("
            << (inMegabytes(bytes) / seconds) << " MB/s)" << std::endl;
}

inline double readTimer()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline size_type memoryUsage()
{
  rusage usage; getrusage(RUSAGE_SELF, &usage);
  return KILOBYTE * (size_type)usage.ru_maxrss;
}

struct Parallel { static size_type max_threads; };

// Near-equal closed sub-ranges; at most `blocks` of them, at least one.
inline std::vector<range_type> getBounds(range_type range, size_type blocks)
{
  std::vector<range_type> bounds;
  if(Range::empty(range)) { return bounds; }
  blocks = Range::bound(blocks, 1, Range::length(range));
  size_type start = range.first;
  for(size_type block = 0; block < blocks; block++)
  {
    size_type first = start;
    if(start <= range.second) { start += std::max((size_type)1, (range.second + 1 - start) / (blocks - block)); }
    bounds.push_back(range_type(first, start - 1));
  }
  return bounds;
}

} // namespace bwtmerge

#endif // BWTM_HOST_UTILS_H
