// Lossless coder for the stride-8 latent coordinates: replaces the reference's round trip through an
// ASCII PLY file and the external MPEG G-PCC `tmc3` binary (model/model.py:388-486) inside the timed region.
// Host code (the set has ~13 k points per vox10 block): breadth-free depth-first octree over the Morton-sorted cells,
// occupancy bits coded with an adaptive binary range coder (LZMA-style rc: 32-bit range, 11-bit probabilities,
// shift-5 adaptation) under the context (depth, child position, occupied siblings already coded in this node).
// Stream: u32 n | u8 depth | range-coder bytes.  Not the G-PCC syntax (that binary is absent here); lossless.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "pcc_common.h"

namespace {

constexpr int kProbBits = 11, kMoveBits = 5;
constexpr uint32_t kTop = 1u << 24;

struct RcEnc {
  std::vector<uint8_t>& out;
  uint64_t low = 0;
  uint32_t range = 0xFFFFFFFFu;
  uint8_t cache = 0;
  uint64_t cache_size = 1;
  explicit RcEnc(std::vector<uint8_t>& o) : out(o) {}
  void shift_low() {
    if ((uint32_t)low < 0xFF000000u || (int)(low >> 32) != 0) {
      uint8_t temp = cache;
      do {
        out.push_back((uint8_t)(temp + (uint8_t)(low >> 32)));
        temp = 0xFF;
      } while (--cache_size != 0);
      cache = (uint8_t)((uint32_t)low >> 24);
    }
    cache_size++;
    low = (uint32_t)low << 8;
  }
  void encode(uint16_t& p, int bit) {
    const uint32_t bound = (range >> kProbBits) * p;
    if (!bit) { range = bound; p += ((1u << kProbBits) - p) >> kMoveBits; }
    else { low += bound; range -= bound; p -= p >> kMoveBits; }
    while (range < kTop) { range <<= 8; shift_low(); }
  }
  void flush() { for (int i = 0; i < 5; ++i) shift_low(); }
};

struct RcDec {
  const uint8_t* p; const uint8_t* end;
  uint32_t range = 0xFFFFFFFFu, code = 0;
  RcDec(const uint8_t* b, const uint8_t* e) : p(b), end(e) { for (int i = 0; i < 5; ++i) code = (code << 8) | next(); }
  uint8_t next() { return p < end ? *p++ : 0; }
  int decode(uint16_t& pr) {
    const uint32_t bound = (range >> kProbBits) * pr;
    int bit;
    if (code < bound) { range = bound; pr += ((1u << kProbBits) - pr) >> kMoveBits; bit = 0; }
    else { code -= bound; range -= bound; pr -= pr >> kMoveBits; bit = 1; }
    while (range < kTop) { range <<= 8; code = (code << 8) | next(); }
    return bit;
  }
};

inline int ctx_of(int level, int child, int occupied_so_far) { return (level * 8 + child) * 8 + std::min(occupied_so_far, 7); }

uint64_t morton3(uint32_t x, uint32_t y, uint32_t z, int depth) {   // x most significant within each bit triple
  uint64_t m = 0;
  for (int b = 0; b < depth; ++b)
    m |= ((uint64_t)((x >> b) & 1) << (3 * b + 2)) | ((uint64_t)((y >> b) & 1) << (3 * b + 1)) | ((uint64_t)((z >> b) & 1) << (3 * b));
  return m;
}

void enc_node(const uint64_t* m, int64_t lo, int64_t hi, int level, int depth, std::vector<uint16_t>& probs, RcEnc& rc) {
  if (level == depth) return;
  const int shift = 3 * (depth - 1 - level);
  int64_t start[9];
  int64_t p = lo;
  for (int c = 0; c < 8; ++c) {
    start[c] = p;
    while (p < hi && (int)((m[p] >> shift) & 7) == c) ++p;
  }
  start[8] = hi;
  int occ = 0;
  for (int c = 0; c < 8; ++c) {
    const int bit = start[c + 1] > start[c];
    rc.encode(probs[ctx_of(level, c, occ)], bit);
    occ += bit;
  }
  for (int c = 0; c < 8; ++c)
    if (start[c + 1] > start[c]) enc_node(m, start[c], start[c + 1], level + 1, depth, probs, rc);
}

bool dec_node(uint64_t prefix, int level, int depth, std::vector<uint16_t>& probs, RcDec& rc, std::vector<uint64_t>& out,
              int64_t cap) {
  if (level == depth) {
    if ((int64_t)out.size() >= cap) return false;
    out.push_back(prefix);
    return true;
  }
  int bits[8], occ = 0;
  for (int c = 0; c < 8; ++c) {
    bits[c] = rc.decode(probs[ctx_of(level, c, occ)]);
    occ += bits[c];
  }
  if (occ == 0) return false;   // an internal node always has a child: corrupt stream
  for (int c = 0; c < 8; ++c)
    if (bits[c] && !dec_node((prefix << 3) | (uint64_t)c, level + 1, depth, probs, rc, out, cap)) return false;
  return true;
}

}  // namespace

extern "C" int64_t pcc_octree_max_bytes(int64_t n, int32_t depth) { return 16 + n * (int64_t)depth * 2 + 64; }

// h_cells: [n,3] int32 (x,y,z) cell coordinates in [0, 2^depth), unique.  Output order of the decoder: Morton order.
extern "C" int pcc_octree_encode_host(const int32_t* h_cells, int64_t n, int32_t depth, uint8_t* h_out, int64_t cap,
                                      int64_t* h_nbytes) {
  PCC_REQUIRE(h_out && h_nbytes && (n == 0 || h_cells) && depth >= 1 && depth <= 16, "pcc_octree_encode_host: bad arguments");
  std::vector<uint64_t> m((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    const int32_t x = h_cells[3 * i], y = h_cells[3 * i + 1], z = h_cells[3 * i + 2];
    PCC_REQUIRE(x >= 0 && y >= 0 && z >= 0 && x < (1 << depth) && y < (1 << depth) && z < (1 << depth),
                "pcc_octree_encode_host: cell (%d,%d,%d) outside the 2^%d cube", x, y, z, depth);
    m[(size_t)i] = morton3((uint32_t)x, (uint32_t)y, (uint32_t)z, depth);
  }
  std::sort(m.begin(), m.end());
  PCC_REQUIRE(std::adjacent_find(m.begin(), m.end()) == m.end(), "pcc_octree_encode_host: duplicate cells");
  std::vector<uint8_t> out;
  out.reserve((size_t)(n * 2 + 64));
  const uint32_t n32 = (uint32_t)n;
  for (int i = 0; i < 4; ++i) out.push_back((uint8_t)(n32 >> (8 * i)));
  out.push_back((uint8_t)depth);
  if (n > 0) {
    std::vector<uint16_t> probs((size_t)depth * 64, (uint16_t)(1u << (kProbBits - 1)));
    RcEnc rc(out);
    enc_node(m.data(), 0, n, 0, depth, probs, rc);
    rc.flush();
  }
  PCC_REQUIRE((int64_t)out.size() <= cap, "pcc_octree_encode_host: output capacity too small");
  memcpy(h_out, out.data(), out.size());
  *h_nbytes = (int64_t)out.size();
  return PCC_OK;
}

extern "C" int pcc_octree_decode_host(const uint8_t* h_data, int64_t nbytes, int32_t* h_cells, int64_t cap_points,
                                      int64_t* h_n, int32_t* h_depth) {
  PCC_REQUIRE(h_data && h_n && nbytes >= 5, "pcc_octree_decode_host: truncated stream");
  uint32_t n = 0;
  for (int i = 0; i < 4; ++i) n |= (uint32_t)h_data[i] << (8 * i);
  const int depth = h_data[4];
  PCC_REQUIRE(depth >= 1 && depth <= 16, "pcc_octree_decode_host: bad depth %d", depth);
  if (h_depth) *h_depth = depth;
  *h_n = n;
  if (!h_cells) return PCC_OK;            // size query
  PCC_REQUIRE((int64_t)n <= cap_points, "pcc_octree_decode_host: output capacity too small");
  if (n == 0) return PCC_OK;
  std::vector<uint16_t> probs((size_t)depth * 64, (uint16_t)(1u << (kProbBits - 1)));
  RcDec rc(h_data + 5, h_data + nbytes);
  std::vector<uint64_t> out;
  out.reserve(n);
  PCC_REQUIRE(dec_node(0, 0, depth, probs, rc, out, (int64_t)n) && out.size() == n, "pcc_octree_decode_host: corrupt stream");
  for (size_t i = 0; i < out.size(); ++i) {
    uint32_t x = 0, y = 0, z = 0;
    for (int b = 0; b < depth; ++b) {
      x |= (uint32_t)((out[i] >> (3 * b + 2)) & 1) << b;
      y |= (uint32_t)((out[i] >> (3 * b + 1)) & 1) << b;
      z |= (uint32_t)((out[i] >> (3 * b)) & 1) << b;
    }
    h_cells[3 * i] = (int32_t)x; h_cells[3 * i + 1] = (int32_t)y; h_cells[3 * i + 2] = (int32_t)z;
  }
  return PCC_OK;
}
