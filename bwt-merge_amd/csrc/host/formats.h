/*
  formats.h -- host facade: NativeHeader (reference formats.h:44-62) and the BWT file formats of the
  reference (formats.h:64-156, formats.cpp:100-445): "native" (the reference's own format, SURVEY.md
  Appendix B) and the six foreign ones, which are serial CPU codecs next to the hot path:

    plain_default / plain_sorted   one character per base
    rfm                            SDSL int_vector<8> of comp values, sorted alphabet
    sdsl                           SDSL int_vector<8> of characters, sorted alphabet
    ropebwt                        4-byte tag, then one byte per run: (length << 3) | comp, length <= 31
    sga                            30-byte header, then one byte per run: (comp << 5) | length, length <= 31

  Every foreign format is a pair of functions here: decode(in, emit) calls emit(value, length) for the
  items of the file in order (value = a character or a comp value, see FormatTraits::characters), and
  encode(out, header, for_each_run) writes the runs it is handed.  fmi.h turns items into the native
  run stream (RunBuffer + Run::write) and back.
*/
#ifndef BWTM_HOST_FORMATS_H
#define BWTM_HOST_FORMATS_H

#include <fstream>
#include "support.h"
#include "sdsl_compat.h"

namespace bwtmerge
{

enum AlphabeticOrder { AO_DEFAULT = 0, AO_SORTED = 1, AO_ANY = 254, AO_UNKNOWN = 255 };

inline AlphabeticOrder identifyAlphabet(const Alphabet& alpha)
{
  if(alpha.sorted()) { return AO_SORTED; }
  return (alpha == Alphabet() ? AO_DEFAULT : AO_UNKNOWN);
}

struct NativeHeader
{
  std::uint32_t tag;
  std::uint32_t flags;
  std::uint64_t sequences;
  std::uint64_t bases;

  const static std::uint32_t DEFAULT_TAG = 0x54574221;
  const static std::uint32_t ALPHABET_MASK = 0xFF;

  NativeHeader() : tag(DEFAULT_TAG), flags(0), sequences(0), bases(0) {}
  bool check() const { return tag == DEFAULT_TAG; }
  AlphabeticOrder order() const { return static_cast<AlphabeticOrder>(flags & ALPHABET_MASK); }
  void setOrder(AlphabeticOrder ao) { flags = (flags & ~ALPHABET_MASK) | (static_cast<std::uint32_t>(ao) & ALPHABET_MASK); }

  void serialize(std::ostream& out) const
  {
    sdsl_compat::write_member(tag, out); sdsl_compat::write_member(flags, out);
    sdsl_compat::write_member(sequences, out); sdsl_compat::write_member(bases, out);
  }
  void load(std::istream& in)
  {
    sdsl_compat::read_member(tag, in); sdsl_compat::read_member(flags, in);
    sdsl_compat::read_member(sequences, in); sdsl_compat::read_member(bases, in);
  }
};

// Alphabets of the two alphabetic orders (formats.cpp:33-52): sorted = $ACGNT.
inline Alphabet createAlphabet(AlphabeticOrder order)
{
  Alphabet alpha;
  if(order == AO_SORTED)
  {
    std::swap(alpha.comp2char[4], alpha.comp2char[5]);
    std::swap(alpha.char2comp[(byte_type)'N'], alpha.char2comp[(byte_type)'T']);
    std::swap(alpha.char2comp[(byte_type)'n'], alpha.char2comp[(byte_type)'t']);
  }
  return alpha;
}

inline std::string alphabetName(AlphabeticOrder order)
{
  switch(order)
  {
    case AO_DEFAULT: return "default";
    case AO_SORTED:  return "sorted";
    case AO_ANY:     return "any";
    default:         return "unknown";
  }
}

// formats.cpp:83-98
inline bool compatible(const Alphabet& alpha, AlphabeticOrder order)
{
  switch(order)
  {
    case AO_DEFAULT: return alpha == Alphabet();
    case AO_SORTED:  return alpha.sorted();
    case AO_ANY:     return true;
    default:         return false;
  }
}

struct RopeHeader
{
  std::uint32_t tag;
  const static std::uint32_t DEFAULT_TAG = 0x06454C52;
  const static size_type SIZE = 4;
  RopeHeader() : tag(DEFAULT_TAG) {}
  void serialize(std::ostream& out) const { sdsl_compat::write_member(tag, out); }
  void load(std::istream& in) { sdsl_compat::read_member(tag, in); }
  bool check() const { return tag == DEFAULT_TAG; }
};

struct SGAHeader
{
  std::uint16_t tag;
  std::uint64_t sequences, bases, bytes;
  std::uint32_t flags;
  const static std::uint16_t DEFAULT_TAG = 0xCACA;
  const static std::uint32_t DEFAULT_FLAGS = 0;
  SGAHeader() : tag(DEFAULT_TAG), sequences(0), bases(0), bytes(0), flags(DEFAULT_FLAGS) {}
  void serialize(std::ostream& out) const
  {
    sdsl_compat::write_member(tag, out); sdsl_compat::write_member(sequences, out); sdsl_compat::write_member(bases, out);
    sdsl_compat::write_member(bytes, out); sdsl_compat::write_member(flags, out);
  }
  void load(std::istream& in)
  {
    sdsl_compat::read_member(tag, in); sdsl_compat::read_member(sequences, in); sdsl_compat::read_member(bases, in);
    sdsl_compat::read_member(bytes, in); sdsl_compat::read_member(flags, in);
  }
  bool check() const { return tag == DEFAULT_TAG && flags == DEFAULT_FLAGS; }
};

inline size_type remainingBytes(std::istream& in)
{
  std::streamoff here = in.tellg();
  in.seekg(0, std::ios_base::end);
  std::streamoff end = in.tellg();
  in.seekg(here);
  return (size_type)(end - here);
}

namespace codec
{

const size_type CHUNK = MEGABYTE;

// `bytes` bytes of the stream, one item each.
template<class Emit>
void decodeBytes(std::istream& in, size_type bytes, Emit&& emit)
{
  std::vector<char> buffer(CHUNK);
  for(size_type done = 0; done < bytes && in; )
  {
    size_type want = std::min(CHUNK, bytes - done);
    in.read(buffer.data(), want);
    size_type got = (size_type)in.gcount();
    for(size_type k = 0; k < got; k++) { emit((size_type)(byte_type)buffer[k], (size_type)1); }
    done += got;
    if(got < want) { break; }
  }
}

// One byte per run; `split` maps a byte to (comp, length).
template<class Split, class Emit>
void decodeRunBytes(std::istream& in, size_type bytes, Split&& split, Emit&& emit)
{
  decodeBytes(in, bytes, [&](size_type code, size_type) { range_type run = split((byte_type)code); emit(run.first, run.second); });
}

struct ByteWriter
{
  explicit ByteWriter(std::ostream& stream) : out(stream) { buffer.reserve(CHUNK); }
  ~ByteWriter() { flush(); }
  void put(byte_type b) { buffer.push_back((char)b); if(buffer.size() >= CHUNK) { flush(); } }
  void fill(byte_type b, size_type n)
  {
    while(n > 0)
    {
      size_type take = std::min(n, CHUNK - buffer.size());
      buffer.insert(buffer.end(), take, (char)b); n -= take;
      if(buffer.size() >= CHUNK) { flush(); }
    }
  }
  void flush() { out.write(buffer.data(), buffer.size()); written += buffer.size(); buffer.clear(); }
  std::ostream& out; std::vector<char> buffer; size_type written = 0;
};

} // namespace codec

/*
  The formats.  Each type has: tag(), name(), order(), characters (items are characters and are mapped
  through the alphabet after the runs are formed, formats.cpp:147-156; otherwise they are comp values),
  decode(in, emit) and encode(out, header, runs) where runs(f) calls f(value, length) for every run
  (value already mapped to what the file stores).
*/
struct NativeFormat
{
  static const char* tag() { return "native"; }
  static const char* name() { return "Native format"; }
  static AlphabeticOrder order() { return AO_ANY; }
};

template<AlphabeticOrder ORDER>
struct PlainFormat
{
  static AlphabeticOrder order() { return ORDER; }
  const static bool characters = true;
  template<class Emit> static void decode(std::istream& in, Emit&& emit) { codec::decodeBytes(in, remainingBytes(in), emit); }
  template<class Runs> static void encode(std::ostream& out, const NativeHeader&, Runs&& runs)
  {
    codec::ByteWriter w(out);
    runs([&](size_type value, size_type length) { w.fill((byte_type)value, length); });
  }
};
struct PlainFormatD : PlainFormat<AO_DEFAULT> { static const char* tag() { return "plain_default"; } static const char* name() { return "Plain format (default alphabet)"; } };
struct PlainFormatS : PlainFormat<AO_SORTED>  { static const char* tag() { return "plain_sorted"; }  static const char* name() { return "Plain format (sorted alphabet)"; } };

// SDSL int_vector<8>: 64-bit length in bits, then the bytes padded to a multiple of 8 (utils.h:374-407).
template<bool CHARACTERS>
struct IntVector8Format
{
  static AlphabeticOrder order() { return AO_SORTED; }
  const static bool characters = CHARACTERS;
  template<class Emit> static void decode(std::istream& in, Emit&& emit)
  {
    size_type bits = 0; sdsl_compat::read_member(bits, in);
    codec::decodeBytes(in, bits / 8, emit);
  }
  template<class Runs> static void encode(std::ostream& out, const NativeHeader& header, Runs&& runs)
  {
    size_type bits = header.bases * 8; sdsl_compat::write_member(bits, out);
    codec::ByteWriter w(out);
    runs([&](size_type value, size_type length) { w.fill((byte_type)value, length); });
    if(header.bases % 8 != 0) { w.fill(0, 8 - header.bases % 8); }
  }
};
struct RFMFormat  : IntVector8Format<false> { static const char* tag() { return "rfm"; }  static const char* name() { return "RFM format"; } };
struct SDSLFormat : IntVector8Format<true>  { static const char* tag() { return "sdsl"; } static const char* name() { return "SDSL format"; } };

// One byte per run of at most 31 (formats.cpp:237-300); RopeBWT: (length << 3) | comp, SGA: (comp << 5) | length.
struct RopeFormat
{
  static const char* tag() { return "ropebwt"; }
  static const char* name() { return "RopeBWT format"; }
  static AlphabeticOrder order() { return AO_DEFAULT; }
  const static bool characters = false;
  const static size_type MAX_RUN = 31;
  template<class Emit> static void decode(std::istream& in, Emit&& emit)
  {
    RopeHeader header; header.load(in);
    if(!header.check()) { std::cerr << "RopeFormat::load(): Invalid header!" << std::endl; std::exit(EXIT_FAILURE); }
    codec::decodeRunBytes(in, remainingBytes(in), [](byte_type code) { return range_type(code & 0x07, code >> 3); }, emit);
  }
  template<class Runs> static void encode(std::ostream& out, const NativeHeader&, Runs&& runs)
  {
    RopeHeader().serialize(out);
    codec::ByteWriter w(out);
    runs([&](size_type comp, size_type length)
    {
      for(; length > MAX_RUN; length -= MAX_RUN) { w.put((byte_type)((MAX_RUN << 3) | comp)); }
      w.put((byte_type)((length << 3) | comp));
    });
  }
};

struct SGAFormat
{
  static const char* tag() { return "sga"; }
  static const char* name() { return "SGA format"; }
  static AlphabeticOrder order() { return AO_DEFAULT; }
  const static bool characters = false;
  const static size_type MAX_RUN = 31;
  template<class Emit> static void decode(std::istream& in, Emit&& emit)
  {
    SGAHeader header; header.load(in);
    if(!header.check()) { std::cerr << "SGAFormat::load(): Invalid header!" << std::endl; std::exit(EXIT_FAILURE); }
    codec::decodeRunBytes(in, header.bytes, [](byte_type code) { return range_type(code >> 5, code & 0x1F); }, emit);
  }
  template<class Runs> static void encode(std::ostream& out, const NativeHeader& info, Runs&& runs)
  {
    SGAHeader header; header.sequences = info.sequences; header.bases = info.bases;
    runs([&](size_type, size_type length) { header.bytes += (length + MAX_RUN - 1) / MAX_RUN; });   // formats.cpp:302-322
    header.serialize(out);
    codec::ByteWriter w(out);
    runs([&](size_type comp, size_type length)
    {
      for(; length > MAX_RUN; length -= MAX_RUN) { w.put((byte_type)((comp << 5) | MAX_RUN)); }
      w.put((byte_type)((comp << 5) | length));
    });
  }
};

// Calls f(Format()) for the format with the given tag; false if there is none.
template<class F>
bool withFormat(const std::string& tag, F&& f)
{
  if(tag == NativeFormat::tag())      { f(NativeFormat()); }
  else if(tag == PlainFormatD::tag()) { f(PlainFormatD()); }
  else if(tag == PlainFormatS::tag()) { f(PlainFormatS()); }
  else if(tag == RFMFormat::tag())    { f(RFMFormat()); }
  else if(tag == SDSLFormat::tag())   { f(SDSLFormat()); }
  else if(tag == RopeFormat::tag())   { f(RopeFormat()); }
  else if(tag == SGAFormat::tag())    { f(SGAFormat()); }
  else { return false; }
  return true;
}

inline bool formatExists(const std::string& format) { return withFormat(format, [](auto) {}); }

template<class Format>
void printFormat(std::ostream& out)
{
  std::string tag = Format::tag();
  out << "  " << tag << (tag.length() < 15 ? std::string(15 - tag.length(), ' ') : std::string()) << Format::name() << std::endl;
}

// formats.cpp:462-481
inline void printFormats(std::ostream& out)
{
  out << "Formats supporting any alphabetic order:" << std::endl;
  printFormat<NativeFormat>(out);
  out << std::endl;
  out << "Formats using the default alphabet:" << std::endl;
  printFormat<PlainFormatD>(out); printFormat<RopeFormat>(out); printFormat<SGAFormat>(out);
  out << std::endl;
  out << "Formats using sorted alphabet:" << std::endl;
  printFormat<PlainFormatS>(out); printFormat<RFMFormat>(out); printFormat<SDSLFormat>(out);
  out << std::endl;
}

} // namespace bwtmerge

#endif // BWTM_HOST_FORMATS_H
