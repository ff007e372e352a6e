/*
  formats.h -- host facade: NativeHeader (reference formats.h:44-62) and the two formats this
  repository reads and writes: "native" (the reference's own format) and
  "plain_default" (one character per base, default alphabet; formats.cpp:126-235).  The other
  foreign formats (rfm, sdsl, ropebwt, sga, plain_sorted) are serial CPU codecs outside the hot
  path and are not provided.
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

struct NativeFormat  { static const char* tag() { return "native"; }        static const char* name() { return "Native format"; } };
struct PlainFormatD  { static const char* tag() { return "plain_default"; } static const char* name() { return "Plain format (default alphabet)"; } };

inline bool formatExists(const std::string& format) { return format == NativeFormat::tag() || format == PlainFormatD::tag(); }

inline void printFormats(std::ostream& out)
{
  out << "Formats supporting any alphabetic order:" << std::endl;
  out << "  " << NativeFormat::tag() << std::string(15 - std::string(NativeFormat::tag()).length(), ' ') << NativeFormat::name() << std::endl << std::endl;
  out << "Formats using the default alphabet:" << std::endl;
  out << "  " << PlainFormatD::tag() << std::string(15 - std::string(PlainFormatD::tag()).length(), ' ') << PlainFormatD::name() << std::endl << std::endl;
}

} // namespace bwtmerge

#endif // BWTM_HOST_FORMATS_H
