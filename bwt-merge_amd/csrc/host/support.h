/*
  support.h -- host facade: Alphabet, BlockArray, ByteCode, Run (reference support.h:41-286).
  The RA buffer hierarchy of the reference (RLArray, RankArray over temp files) has no
  counterpart here: the rank array lives on the device (fmi.h, RankArray below wraps the handle).
*/
#ifndef BWTM_HOST_SUPPORT_H
#define BWTM_HOST_SUPPORT_H

#include <cstdlib>
#include <cstring>
#include <bwtm.h>
#include "utils.h"

namespace bwtmerge
{

// A flat array in PAGE-LOCKED host memory (bwtm_host_alloc) when a GPU is usable, in plain memory otherwise (the
// converters run without one): transfers from / to it run at PCIe speed and overlap with kernels.  The part of
// std::vector's interface the facade uses.
template<class T>
class HostArray
{
public:
  typedef T value_type;

  HostArray() : ptr(nullptr), count(0), capacity(0), pinned(false) {}
  explicit HostArray(size_type n) : HostArray() { resize(n); }
  HostArray(const HostArray& other) : HostArray() { assign(other.ptr, other.ptr + other.count); }
  HostArray(HostArray&& other) noexcept : HostArray() { swap(other); }
  HostArray& operator=(const HostArray& other) { if(this != &other) { assign(other.ptr, other.ptr + other.count); } return *this; }
  HostArray& operator=(HostArray&& other) noexcept { if(this != &other) { release(); swap(other); } return *this; }
  ~HostArray() { release(); }

  size_type size() const { return count; }
  bool empty() const { return count == 0; }
  T* data() { return ptr; }
  const T* data() const { return ptr; }
  T& operator[](size_type i) { return ptr[i]; }
  const T& operator[](size_type i) const { return ptr[i]; }
  T& back() { return ptr[count - 1]; }
  const T& back() const { return ptr[count - 1]; }
  T* begin() { return ptr; }
  T* end() { return ptr + count; }
  const T* begin() const { return ptr; }
  const T* end() const { return ptr + count; }

  void reserve(size_type n)
  {
    if(n <= capacity) { return; }
    bool new_pinned = false;
    T* fresh = allocate(n, new_pinned);
    if(count > 0) { std::memcpy((void*)fresh, (const void*)ptr, count * sizeof(T)); }
    deallocate(ptr, pinned);
    ptr = fresh; capacity = n; pinned = new_pinned;
  }
  void resize(size_type n, T value = T())
  {
    if(n > capacity) { reserve(n); }
    for(size_type k = count; k < n; k++) { ptr[k] = value; }
    count = n;
  }
  // Sets the size without initializing new elements (buffers that a download fills).
  void resizeUninitialized(size_type n) { if(n > capacity) { reserve(n); } count = n; }
  void push_back(T v)
  {
    if(count == capacity) { reserve(capacity < 1024 ? 4096 : 2 * capacity); }
    ptr[count++] = v;
  }
  template<class It> void assign(It first, It last)
  {
    size_type n = (size_type)(last - first);
    resizeUninitialized(n);
    for(size_type k = 0; k < n; k++, ++first) { ptr[k] = *first; }
  }
  void assign(size_type n, T value) { count = 0; resize(n, value); }
  void clear() { count = 0; }
  void shrink_to_fit() { if(count == 0) { release(); } }
  void swap(HostArray& other) { std::swap(ptr, other.ptr); std::swap(count, other.count); std::swap(capacity, other.capacity); std::swap(pinned, other.pinned); }

  bool operator==(const HostArray& other) const { return count == other.count && (count == 0 || std::memcmp(ptr, other.ptr, count * sizeof(T)) == 0); }
  bool operator!=(const HostArray& other) const { return !(*this == other); }
  bool operator==(const std::vector<T>& other) const { return count == other.size() && (count == 0 || std::memcmp(ptr, other.data(), count * sizeof(T)) == 0); }

private:
  static T* allocate(size_type n, bool& is_pinned)
  {
    static bool pinned_available = true;                 // one failed attempt (no GPU) switches to plain memory for good
    void* p = nullptr;
    if(pinned_available && n * sizeof(T) >= PINNED_MIN)
    {
      if(bwtm_host_alloc(n * sizeof(T), &p) == BWTM_OK) { is_pinned = true; return (T*)p; }
      pinned_available = false;
    }
    p = std::malloc(n * sizeof(T) > 0 ? n * sizeof(T) : 1);
    if(!p) { std::cerr << "HostArray: cannot allocate " << n * sizeof(T) << " bytes" << std::endl; std::exit(EXIT_FAILURE); }
    is_pinned = false;
    return (T*)p;
  }
  static void deallocate(T* p, bool is_pinned) { if(!p) { return; } if(is_pinned) { bwtm_host_free(p); } else { std::free(p); } }
  void release() { deallocate(ptr, pinned); ptr = nullptr; count = 0; capacity = 0; pinned = false; }

  const static size_type PINNED_MIN = 1 << 20;           // small arrays are not worth a pinning call

  T* ptr; size_type count, capacity; bool pinned;
};

// char <-> comp maps and the C array.  Default order $ACGTN = 0..5; every other byte maps to N.
class Alphabet
{
public:
  const static size_type MAX_SIGMA = 256;

  Alphabet() : char2comp(MAX_SIGMA, 5), comp2char({'$', 'A', 'C', 'G', 'T', 'N'}), C(7, 0), sigma(6)
  {
    char2comp[0] = 0; char2comp['$'] = 0;
    const char* upper = "ACGT"; const char* lower = "acgt";
    for(size_type k = 0; k < 4; k++) { char2comp[(byte_type)upper[k]] = k + 1; char2comp[(byte_type)lower[k]] = k + 1; }
  }

  // counts[c] = occurrences of comp value c
  explicit Alphabet(const std::vector<size_type>& counts) : Alphabet()
  {
    for(size_type c = 0; c < counts.size() && c < sigma; c++) { C[c + 1] = C[c] + counts[c]; }
  }

  // counts + explicit maps (reference support.cpp:93-105): what FMI::load<Format> builds for a foreign format
  Alphabet(const std::vector<size_type>& counts, const std::vector<byte_type>& _char2comp, const std::vector<byte_type>& _comp2char) :
    char2comp(_char2comp), comp2char(_comp2char), C(_comp2char.size() + 1, 0), sigma(_comp2char.size())
  {
    for(size_type c = 0; c < counts.size() && c < sigma; c++) { C[c + 1] = C[c] + counts[c]; }
  }

  bool sorted() const
  {
    for(size_type c = 1; c < sigma; c++) { if(comp2char[c - 1] >= comp2char[c]) { return false; } }
    return true;
  }

  // Equality ignores C (reference support.cpp:192-205).
  bool operator==(const Alphabet& another) const
  {
    return sigma == another.sigma && char2comp == another.char2comp && comp2char == another.comp2char;
  }
  bool operator!=(const Alphabet& another) const { return !(*this == another); }

  std::vector<byte_type> char2comp, comp2char;
  std::vector<size_type> C;
  size_type              sigma;
};

// Byte array of the BWT.  The reference keeps 8 MiB mmap blocks (support.h:90-150); here the
// bytes are contiguous (they are uploaded in one piece), and only serialization pads to blocks.
class BlockArray
{
public:
  typedef byte_type value_type;
  const static size_type BLOCK_SIZE = 8 * MEGABYTE;

  size_type size() const { return bytes.size(); }
  bool empty() const { return bytes.empty(); }
  size_type blocks() const { return (bytes.size() + BLOCK_SIZE - 1) / BLOCK_SIZE; }
  void clear() { bytes.clear(); bytes.shrink_to_fit(); }
  value_type operator[](size_type i) const { return bytes[i]; }
  value_type& operator[](size_type i) { return bytes[i]; }
  void push_back(value_type v) { bytes.push_back(v); }
  void swap(BlockArray& other) { bytes.swap(other.bytes); }
  const value_type* data() const { return bytes.data(); }

  HostArray<value_type> bytes;
};

// 7 data bits per byte, least significant group first, high bit = "continues".
struct ByteCode
{
  template<class ByteArray>
  static size_type read(const ByteArray& array, size_type& i)
  {
    size_type shift = 0, value = array[i] & 0x7F;
    while(array[i] & 0x80) { i++; shift += 7; value += ((size_type)(array[i] & 0x7F)) << shift; }
    i++;
    return value;
  }

  template<class ByteArray>
  static void write(ByteArray& array, size_type value)
  {
    for(; value > 0x7F; value >>= 7) { array.push_back((byte_type)((value & 0x7F) | 0x80)); }
    array.push_back((byte_type)value);
  }
};

// A BWT run: one byte comp + 6 * (length - 1) for length < 42; otherwise that byte with
// length 42 followed by the varint of the rest.  No run crosses a 64-byte boundary.
struct Run
{
  const static size_type BLOCK_SIZE = 64;
  const static size_type SIGMA      = 6;
  const static size_type MAX_RUN    = 256 / SIGMA;

  static byte_type encodeBasic(comp_type comp, size_type length) { return (byte_type)(comp + SIGMA * (length - 1)); }
  static range_type decodeBasic(byte_type code) { return range_type(code % SIGMA, code / SIGMA + 1); }

  template<class ByteArray>
  static range_type read(const ByteArray& array, size_type& i)
  {
    range_type run = decodeBasic(array[i]); i++;
    if(run.second >= MAX_RUN) { run.second += ByteCode::read(array, i); }
    return run;
  }

  template<class ByteArray>
  static void write(ByteArray& array, comp_type comp, size_type length)
  {
    while(length > 0)
    {
      if(length < MAX_RUN) { array.push_back(encodeBasic(comp, length)); return; }
      size_type room = BLOCK_SIZE - (array.size() % BLOCK_SIZE);       // bytes left in the block, >= 1
      size_type head = (room > 1 ? MAX_RUN : MAX_RUN - 1);               // a lone last byte cannot announce an extension
      array.push_back(encodeBasic(comp, head)); length -= head; room--;
      if(room > 0)
      {
        size_type extension = length;
        if(bit_length(length) > 7 * room) { extension = (~(size_type)0) >> (64 - 7 * room); }
        ByteCode::write(array, extension); length -= extension;
      }
    }
  }

  template<class ByteArray>
  static void write(ByteArray& array, range_type run) { write(array, (comp_type)run.first, run.second); }
};

} // namespace bwtmerge

#endif // BWTM_HOST_SUPPORT_H
