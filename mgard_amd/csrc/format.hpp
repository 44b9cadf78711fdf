// Self-describing container of MGARD-X compressed data: hand-written proto3 encoder/decoder
// for `mgard.pb.Header`, CRC-32 and the preamble, byte-compatible with the reference
//   "MGARD" (5 B) | u64 LE header_size | u32 LE crc32(header) | proto3 Header
// (src/mgard-x/Metadata/Metadata.cpp:241-247, 249-462; include/format.hpp:28-34; schema
// src/mgard.proto). No protobuf dependency. proto3 rules that matter for byte equality with
// the stock runtime: fields in field-number order, zero-valued scalars omitted, sub-messages
// that were touched are written even when empty, repeated numerics packed.
//
// Quirk mirrored from Metadata.cpp:252-272: `file_format_version` is only touched (written as
// an empty message) and `mgard_version` carries the FILE format version (1.0.0).
#pragma once
#include <cstdint>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

namespace mgh {
namespace fmt {

// enum values of src/mgard.proto
enum : uint64_t { DD_NOOP = 0, DD_MAX_DIMENSION = 1, DD_BLOCK = 2, DD_VARIABLE = 3 };
enum : uint64_t { HIER_POW2P1 = 0, HIER_MULTIDIM = 1, HIER_ONE_DIM = 2, HIER_HYBRID = 3 };
enum : uint64_t {
  COMP_NOOP = 0, COMP_CPU_HUFFMAN_ZLIB = 1, COMP_CPU_HUFFMAN_ZSTD = 2, COMP_X_HUFFMAN = 3,
  COMP_X_HUFFMAN_LZ4 = 4, COMP_X_HUFFMAN_ZSTD = 5
};
enum : uint64_t { DEV_CPU = 0, DEV_X_SERIAL = 1, DEV_X_OPENMP = 2, DEV_X_CUDA = 3, DEV_X_HIP = 4, DEV_X_SYCL = 5 };

struct Header {
  uint64_t version[3] = {1, 0, 0};       // what lands in mgard_version (see the quirk above)
  uint64_t file_version[3] = {0, 0, 0};  // parsed only
  bool is_double = false;
  std::vector<uint64_t> shape;
  bool uniform = true;
  std::vector<std::vector<double>> coords;  // when !uniform
  bool rel = false;                          // ErrorControl.mode
  double tol = 0, s = 0, norm = 0;
  uint64_t dd_method = DD_NOOP, dd_dim = 0, dd_size = 0;
  uint64_t hierarchy = HIER_MULTIDIM, l_target = 0;
  bool quantized = true;  // COEFFICIENTWISE_LINEAR / INT64_T
  bool big_endian = false;
  bool reorder = false;
  uint64_t compressor = COMP_X_HUFFMAN, huff_dict_size = 0, huff_block_size = 0;
  uint64_t backend = DEV_X_HIP;
};

// ---- CRC-32 (zlib / IEEE 802.3, reflected 0xEDB88320) ------------------------------------
inline uint32_t crc32(const uint8_t *p, size_t n) {
  // (function-local static object: initialised once, safely from several threads)
  struct Table {
    uint32_t t[256];
    Table() {
      for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        t[i] = c;
      }
    }
  };
  static const Table tab;
  const uint32_t *table = tab.t;
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

// ---- proto3 wire writer -------------------------------------------------------------------
struct Writer {
  std::vector<uint8_t> b;
  void varint(uint64_t v) {
    while (v >= 0x80) {
      b.push_back((uint8_t)(v | 0x80));
      v >>= 7;
    }
    b.push_back((uint8_t)v);
  }
  void tag(int field, int wire) { varint(((uint64_t)field << 3) | (uint64_t)wire); }
  void u64(int field, uint64_t v) {  // scalar: omitted when zero
    if (!v) return;
    tag(field, 0);
    varint(v);
  }
  void f64(int field, double v) {  // omitted when the bit pattern is +0.0
    uint64_t bits;
    std::memcpy(&bits, &v, 8);
    if (!bits) return;
    tag(field, 1);
    for (int i = 0; i < 8; i++) b.push_back((uint8_t)(bits >> (8 * i)));
  }
  void bytes(int field, const std::vector<uint8_t> &m) {  // sub-message: always written
    tag(field, 2);
    varint(m.size());
    b.insert(b.end(), m.begin(), m.end());
  }
};

inline std::vector<uint8_t> encode_header(const Header &h) {
  Writer out;
  {  // mgard_version = 2, file_format_version = 3
    Writer v;
    v.u64(1, h.version[0]);
    v.u64(2, h.version[1]);
    v.u64(3, h.version[2]);
    out.bytes(2, v.b);
    out.bytes(3, {});
  }
  {  // domain = 4
    Writer d, topo;
    topo.u64(1, h.shape.size());
    if (!h.shape.empty()) {
      Writer packed;
      for (uint64_t n : h.shape) packed.varint(n);
      topo.bytes(2, packed.b);
    }
    d.bytes(2, topo.b);
    if (!h.uniform) {
      d.u64(3, 1);  // EXPLICIT_CUBE
      Writer geo, packed;
      for (const auto &c : h.coords)
        for (double x : c) {
          uint64_t bits;
          std::memcpy(&bits, &x, 8);
          for (int i = 0; i < 8; i++) packed.b.push_back((uint8_t)(bits >> (8 * i)));
        }
      if (!packed.b.empty()) geo.bytes(2, packed.b);
      d.bytes(4, geo.b);
    }
    out.bytes(4, d.b);
  }
  {  // dataset = 5
    Writer d;
    d.u64(1, h.is_double ? 1 : 0);
    d.u64(2, 1);
    out.bytes(5, d.b);
  }
  {  // error_control = 6
    Writer e;
    e.u64(1, h.rel ? 1 : 0);
    e.u64(2, h.s == std::numeric_limits<double>::infinity() ? 0 : 1);
    e.f64(3, h.s);
    if (h.rel) e.f64(4, h.norm);
    e.f64(5, h.tol);
    out.bytes(6, e.b);
  }
  {  // domain_decomposition = 7
    Writer d;
    d.u64(1, h.dd_method);
    d.u64(2, h.dd_dim);
    d.u64(3, h.dd_size);
    out.bytes(7, d.b);
  }
  {  // function_decomposition = 8
    Writer f;
    f.u64(2, h.hierarchy);
    f.u64(3, h.l_target);
    out.bytes(8, f.b);
  }
  {  // quantization = 9
    Writer q;
    if (h.quantized) {
      q.u64(1, 1);  // COEFFICIENTWISE_LINEAR
      q.u64(3, 3);  // INT64_T
      q.u64(4, h.big_endian ? 1 : 0);
    }
    out.bytes(9, q.b);
  }
  out.bytes(10, {});  // bitplane_encoding: NOOP
  {                   // encoding = 11
    Writer e;
    e.u64(1, h.reorder ? 1 : 0);
    e.u64(2, h.compressor);
    e.u64(3, h.huff_dict_size);
    e.u64(4, h.huff_block_size);
    out.bytes(11, e.b);
  }
  {  // device = 12
    Writer d;
    d.u64(1, h.backend);
    out.bytes(12, d.b);
  }
  return out.b;
}

constexpr size_t kPreambleSize = 5 + 8 + 4;

inline std::vector<uint8_t> serialize_metadata(const Header &h) {
  const std::vector<uint8_t> body = encode_header(h);
  std::vector<uint8_t> out = {0x4d, 0x47, 0x41, 0x52, 0x44};
  const uint64_t n = body.size();
  for (int i = 0; i < 8; i++) out.push_back((uint8_t)(n >> (8 * i)));
  const uint32_t c = crc32(body.data(), body.size());
  for (int i = 0; i < 4; i++) out.push_back((uint8_t)(c >> (8 * i)));
  out.insert(out.end(), body.begin(), body.end());
  return out;
}

// ---- proto3 wire reader -------------------------------------------------------------------
struct Reader {
  const uint8_t *p, *end;
  Reader(const uint8_t *b, size_t n) : p(b), end(b + n) {}
  bool done() const { return p >= end; }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      if (p >= end) throw std::runtime_error("header: truncated varint");
      const uint8_t c = *p++;
      v |= (uint64_t)(c & 0x7F) << shift;
      if (!(c & 0x80)) return v;
    }
    throw std::runtime_error("header: varint too long");
  }
  double f64() {
    if (end - p < 8) throw std::runtime_error("header: truncated double");
    uint64_t bits = 0;
    for (int i = 0; i < 8; i++) bits |= (uint64_t)p[i] << (8 * i);
    p += 8;
    double v;
    std::memcpy(&v, &bits, 8);
    return v;
  }
  Reader sub() {
    const uint64_t n = varint();
    if ((uint64_t)(end - p) < n) throw std::runtime_error("header: truncated sub-message");
    Reader r(p, (size_t)n);
    p += n;
    return r;
  }
  void skip(int wire) {
    if (wire == 0) (void)varint();
    else if (wire == 1) { if (end - p < 8) throw std::runtime_error("header: truncated"); p += 8; }
    else if (wire == 2) (void)sub();
    else if (wire == 5) { if (end - p < 4) throw std::runtime_error("header: truncated"); p += 4; }
    else throw std::runtime_error("header: unsupported wire type");
  }
};

inline void decode_header(const uint8_t *b, size_t n, Header &h) {
  h = Header();
  h.version[0] = h.version[1] = h.version[2] = 0;
  h.quantized = false;
  h.hierarchy = HIER_POW2P1;
  h.compressor = COMP_NOOP;
  h.backend = DEV_CPU;
  uint64_t dim = 0;
  std::vector<double> flat_coords;
  bool explicit_cube = false;
  uint64_t norm_kind = 0;
  double s_field = 0;
  Reader r(b, n);
  while (!r.done()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wire = (int)(key & 7);
    if (wire != 2) { r.skip(wire); continue; }
    Reader m = r.sub();
    auto each = [&](auto &&fn) {
      while (!m.done()) {
        const uint64_t k = m.varint();
        fn((int)(k >> 3), (int)(k & 7));
      }
    };
    switch (field) {
    case 2: case 3: {
      uint64_t *v = field == 2 ? h.version : h.file_version;
      each([&](int f, int w) { if (w == 0 && f >= 1 && f <= 3) v[f - 1] = m.varint(); else m.skip(w); });
      break;
    }
    case 4:
      each([&](int f, int w) {
        if (f == 2 && w == 2) {
          Reader t = m.sub();
          while (!t.done()) {
            const uint64_t k = t.varint();
            const int tf = (int)(k >> 3), tw = (int)(k & 7);
            if (tf == 1 && tw == 0) dim = t.varint();
            else if (tf == 2 && tw == 2) { Reader pk = t.sub(); while (!pk.done()) h.shape.push_back(pk.varint()); }
            else if (tf == 2 && tw == 0) h.shape.push_back(t.varint());
            else t.skip(tw);
          }
        } else if (f == 3 && w == 0) {
          explicit_cube = m.varint() == 1;
        } else if (f == 4 && w == 2) {
          Reader g = m.sub();
          while (!g.done()) {
            const uint64_t k = g.varint();
            const int gf = (int)(k >> 3), gw = (int)(k & 7);
            if (gf == 2 && gw == 2) { Reader pk = g.sub(); while (!pk.done()) flat_coords.push_back(pk.f64()); }
            else if (gf == 2 && gw == 1) flat_coords.push_back(g.f64());
            else g.skip(gw);
          }
        } else m.skip(w);
      });
      break;
    case 5:
      each([&](int f, int w) { if (f == 1 && w == 0) h.is_double = m.varint() == 1; else m.skip(w); });
      break;
    case 6:
      each([&](int f, int w) {
        if (f == 1 && w == 0) h.rel = m.varint() == 1;
        else if (f == 2 && w == 0) norm_kind = m.varint();
        else if (f == 3 && w == 1) s_field = m.f64();
        else if (f == 4 && w == 1) h.norm = m.f64();
        else if (f == 5 && w == 1) h.tol = m.f64();
        else m.skip(w);
      });
      break;
    case 7:
      each([&](int f, int w) {
        if (f == 1 && w == 0) h.dd_method = m.varint();
        else if (f == 2 && w == 0) h.dd_dim = m.varint();
        else if (f == 3 && w == 0) h.dd_size = m.varint();
        else m.skip(w);
      });
      break;
    case 8:
      each([&](int f, int w) {
        if (f == 2 && w == 0) h.hierarchy = m.varint();
        else if (f == 3 && w == 0) h.l_target = m.varint();
        else m.skip(w);
      });
      break;
    case 9:
      each([&](int f, int w) {
        if (f == 1 && w == 0) h.quantized = m.varint() != 0;
        else if (f == 4 && w == 0) h.big_endian = m.varint() != 0;
        else m.skip(w);
      });
      break;
    case 11:
      each([&](int f, int w) {
        if (f == 1 && w == 0) h.reorder = m.varint() == 1;
        else if (f == 2 && w == 0) h.compressor = m.varint();
        else if (f == 3 && w == 0) h.huff_dict_size = m.varint();
        else if (f == 4 && w == 0) h.huff_block_size = m.varint();
        else m.skip(w);
      });
      break;
    case 12:
      each([&](int f, int w) { if (f == 1 && w == 0) h.backend = m.varint(); else m.skip(w); });
      break;
    default:
      break;
    }
  }
  if (dim != h.shape.size()) throw std::runtime_error("header: grid shape does not match given dimension");
  // Metadata.cpp:575-582: L_INFINITY means s = inf whatever the field says
  h.s = norm_kind == 0 ? std::numeric_limits<double>::infinity() : s_field;
  h.uniform = !explicit_cube;
  if (explicit_cube) {
    uint64_t total = 0;
    for (uint64_t n2 : h.shape) total += n2;
    if (total != flat_coords.size())
      throw std::runtime_error("header: mismatch between number of node coordinates and grid shape");
    size_t off = 0;
    for (uint64_t n2 : h.shape) {
      h.coords.emplace_back(flat_coords.begin() + off, flat_coords.begin() + off + n2);
      off += n2;
    }
  }
}

// Parse preamble + header; returns the metadata size (offset of the first subdomain record).
inline size_t parse_metadata(const uint8_t *b, size_t n, Header &h) {
  static const uint8_t sig[5] = {0x4d, 0x47, 0x41, 0x52, 0x44};
  if (n < kPreambleSize || std::memcmp(b, sig, 5) != 0) throw std::runtime_error("signature mismatch");
  uint64_t hs = 0;
  for (int i = 0; i < 8; i++) hs |= (uint64_t)b[5 + i] << (8 * i);
  uint32_t crc = 0;
  for (int i = 0; i < 4; i++) crc |= (uint32_t)b[13 + i] << (8 * i);
  if (hs > n - kPreambleSize) throw std::runtime_error("header: truncated");
  if (crc32(b + kPreambleSize, (size_t)hs) != crc) throw std::runtime_error("header CRC32 mismatch");
  decode_header(b + kPreambleSize, (size_t)hs, h);
  return kPreambleSize + (size_t)hs;
}

} // namespace fmt
} // namespace mgh
