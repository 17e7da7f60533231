// Fuzz harness for the host-side parsers of untrusted bytes: the latent-coordinate octree coder
// (pcc_octree_decode_host; reference counterpart: the tmc3 subprocess output read back by model/model.py:443-486, no validation) and
// the single-stream rANS decoder (pcc_rans_decode_host; reference: compressai's C++ decoder behind model/entropy_models.py:438-484).
// Built by tests/fuzz/Makefile with -fsanitize=address,undefined on the HOST side (CPU only; no HIP call is made), run by
// tests/test_cpu_fuzz.py.  Every case must either decode or return a PCC_E* status; a sanitizer report aborts the run.
//   fuzz_host [cases = 12000] [seed = 1]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#include "pcc_hip.h"

static uint64_t g_s = 88172645463325252ull;
static uint64_t rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return g_s; }
static int rint_(int lo, int hi) { return lo + (int)(rnd() % (uint64_t)(hi - lo + 1)); }

static long g_ok = 0, g_rejected = 0, g_roundtrips = 0;

#define MUST(c, ...) do { if (!(c)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); std::exit(2); } } while (0)

// ---- mutations -----------------------------------------------------------------------------------------------------------
static std::vector<uint8_t> mutate(const std::vector<uint8_t>& in, int kind) {
  std::vector<uint8_t> v = in;
  switch (kind) {
    case 0: v.resize((size_t)rint_(0, (int)v.size())); break;                                   // truncation
    case 1: for (int i = rint_(1, 8); i > 0 && !v.empty(); --i) v[(size_t)(rnd() % v.size())] ^= (uint8_t)(1u << rint_(0, 7)); break;   // bit flips
    case 2: if (v.size() >= 4) { const uint32_t lie = (rnd() & 1) ? (uint32_t)rnd() : (uint32_t)rint_(0, 1 << 20); memcpy(v.data(), &lie, 4); } break;   // length-lying header
    case 3: for (int i = rint_(1, 64); i > 0; --i) v.push_back((uint8_t)rnd()); break;            // trailing garbage
    case 4: for (auto& b : v) if ((rnd() & 15) == 0) b = (uint8_t)rnd(); break;                  // scattered garbage
    default: { const size_t n = (size_t)rint_(0, 64); v.assign(n, 0); for (auto& b : v) b = (uint8_t)rnd(); }     // pure noise
  }
  return v;
}

// ---- octree --------------------------------------------------------------------------------------------------------------
static void octree_case() {
  const int depth = rint_(1, 7);
  const int side = 1 << depth;
  const int n = rint_(0, std::min(2000, side * side * side));
  std::set<uint32_t> seen;
  std::vector<int32_t> cells;
  while ((int)seen.size() < n) {
    const int x = rint_(0, side - 1), y = rint_(0, side - 1), z = rint_(0, side - 1);
    if (seen.insert(((uint32_t)x << 20) | ((uint32_t)y << 10) | (uint32_t)z).second) { cells.push_back(x); cells.push_back(y); cells.push_back(z); }
  }
  const int64_t cap = pcc_octree_max_bytes(n, depth);
  std::vector<uint8_t> buf((size_t)cap);
  int64_t nb = 0;
  MUST(pcc_octree_encode_host(cells.data(), n, depth, buf.data(), cap, &nb) == PCC_OK, "octree encode: %s", pcc_last_error());
  buf.resize((size_t)nb);
  auto decode = [&](const std::vector<uint8_t>& s, std::vector<int32_t>& out, int64_t& cnt) -> int {
    // exactly the calling sequence of container.decode_points: size query, bound on the claimed count, decode
    std::vector<uint8_t> exact(s);                       // heap copy of the exact size: ASan sees any read past the end
    int64_t cn = 0; int32_t cd = 0;
    int rc = pcc_octree_decode_host(exact.data(), (int64_t)exact.size(), nullptr, 0, &cn, &cd);
    if (rc != PCC_OK) return rc;
    int64_t lattice = 1; for (int i = 0; i < std::min(cd, 20); ++i) lattice *= 8;
    if (cn < 0 || cn > std::min<int64_t>(lattice, 4096 * (int64_t)std::max<size_t>(exact.size(), 1) + 4096)) return PCC_EINVAL;
    out.assign((size_t)std::max<int64_t>(cn, 1) * 3, 0);
    rc = pcc_octree_decode_host(exact.data(), (int64_t)exact.size(), out.data(), cn, &cnt, &cd);
    return rc;
  };
  std::vector<int32_t> out; int64_t cnt = 0;
  MUST(decode(buf, out, cnt) == PCC_OK && cnt == n, "octree round trip: rc / count");
  std::set<uint32_t> back;
  for (int64_t i = 0; i < cnt; ++i) back.insert(((uint32_t)out[3 * i] << 20) | ((uint32_t)out[3 * i + 1] << 10) | (uint32_t)out[3 * i + 2]);
  MUST(back == seen, "octree round trip: cells differ");
  ++g_roundtrips;
  for (int k = 0; k < 6; ++k) {
    const std::vector<uint8_t> bad = mutate(buf, k);
    const int rc = decode(bad, out, cnt);
    MUST(rc == PCC_OK || rc == PCC_EINVAL || rc == PCC_EWS, "octree: unexpected status %d", rc);
    (rc == PCC_OK ? g_ok : g_rejected)++;
  }
}

// ---- single-stream rANS ----------------------------------------------------------------------------------------------------
static void rans_case() {
  const int rows = rint_(1, 6), stride = 66;
  std::vector<int32_t> cdf((size_t)rows * stride, 0), sizes(rows), offsets(rows);
  for (int r = 0; r < rows; ++r) {
    const int nsym = rint_(2, 63);                       // pmf length incl. the escape bin
    std::vector<float> pmf(nsym);
    float tot = 0.f;
    for (auto& p : pmf) { p = (float)(rnd() % 1000 + 1); if ((rnd() & 7) == 0) p *= 50.f; tot += p; }
    for (auto& p : pmf) p /= tot;
    std::vector<int32_t> q(nsym + 1);
    MUST(pcc_pmf_to_quantized_cdf(pmf.data(), nsym, 16, q.data()) == PCC_OK, "cdf: %s", pcc_last_error());
    memcpy(&cdf[(size_t)r * stride], q.data(), sizeof(int32_t) * (nsym + 1));
    sizes[r] = nsym + 1;
    offsets[r] = -rint_(0, nsym);
  }
  const int n = rint_(0, 3000);
  std::vector<int32_t> sym(n), idx(n);
  for (int i = 0; i < n; ++i) {
    idx[i] = rint_(0, rows - 1);
    const int span = sizes[idx[i]] - 2;                  // regular values: offset .. offset + span - 1
    sym[i] = offsets[idx[i]] + rint_(-3, span + 2);      // a few out-of-range values take the bypass path
    if ((rnd() & 255) == 0) sym[i] += (rnd() & 1) ? 100000 : -100000;
  }
  const int64_t cap = pcc_rans_max_bytes(n) + 64 * (int64_t)n;      // (bypass payload of far-out values)
  std::vector<uint8_t> buf((size_t)cap);
  int64_t nb = 0;
  const int erc = pcc_rans_encode_host(sym.data(), idx.data(), n, cdf.data(), stride, sizes.data(), offsets.data(), buf.data(), cap, &nb);
  MUST(erc == PCC_OK, "rans encode: %s", pcc_last_error());
  buf.resize((size_t)nb);
  std::vector<int32_t> out((size_t)std::max(n, 1));
  {
    std::vector<uint8_t> exact(buf);
    MUST(pcc_rans_decode_host(exact.data(), (int64_t)exact.size(), idx.data(), n, cdf.data(), stride, sizes.data(), offsets.data(), out.data()) == PCC_OK,
         "rans decode of a clean stream: %s", pcc_last_error());
    for (int i = 0; i < n; ++i) MUST(out[i] == sym[i], "rans round trip: symbol %d: %d != %d", i, out[i], sym[i]);
    ++g_roundtrips;
  }
  for (int k = 0; k < 6; ++k) {
    std::vector<uint8_t> bad = mutate(buf, k);
    std::vector<uint8_t> exact(bad);
    const int rc = pcc_rans_decode_host(exact.empty() ? (const uint8_t*)"" : exact.data(), (int64_t)exact.size(), idx.data(), n, cdf.data(), stride,
                                        sizes.data(), offsets.data(), out.data());
    MUST(rc == PCC_OK || rc == PCC_EINVAL, "rans: unexpected status %d", rc);
    (rc == PCC_OK ? g_ok : g_rejected)++;
  }
}

int main(int argc, char** argv) {
  const long cases = argc > 1 ? std::atol(argv[1]) : 12000;
  if (argc > 2) g_s ^= (uint64_t)std::atoll(argv[2]) * 0x9E3779B97F4A7C15ull;
  long done = 0;
  while (done < cases) {
    octree_case(); done += 6;
    rans_case(); done += 6;
  }
  std::printf("FUZZ OK: %ld mutated cases (%ld decoded to something, %ld rejected), %ld clean round trips\n", g_ok + g_rejected, g_ok, g_rejected, g_roundtrips);
  return 0;
}
