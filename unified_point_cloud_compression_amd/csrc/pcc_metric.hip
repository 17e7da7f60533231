// Nearest-neighbour association for the rate/distortion report: what the reference does with two
// Open3D KD-trees and a Python loop per point at metrics/metric.py:36-43.  Exact search on integer voxel coordinates.
//
// B is given sorted by x (canonical key order is).  A query walks outwards from its own x in both directions and
// stops in a direction once (x_b - x_q)^2 exceeds the best squared distance found: everything beyond is farther.
// Queries in canonical order make neighbouring lanes walk nearly the same rows (cache-resident).
#include "pcc_common.h"

__global__ void __launch_bounds__(256) k_nn_sorted_x(const int* __restrict__ a, long long n_a,
                                                     const int* __restrict__ b, int n_b, long long* __restrict__ d2_out,
                                                     int* __restrict__ nn_out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_a) return;
  const int qx = a[3 * i], qy = a[3 * i + 1], qz = a[3 * i + 2];
  int lo = 0, hi = n_b;                                   // first row with x_b >= qx
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (b[3 * (long long)mid] < qx) lo = mid + 1; else hi = mid;
  }
  long long best = 0x7FFFFFFFFFFFFFFFll;
  int arg = -1;
  int up = lo, dn = lo - 1;
  bool more_up = up < n_b, more_dn = dn >= 0;
  while (more_up || more_dn) {
    if (more_up) {
      const long long dx = (long long)b[3 * (long long)up] - qx;
      if (dx * dx > best) more_up = false;
      else {
        const long long dy = (long long)b[3 * (long long)up + 1] - qy, dz = (long long)b[3 * (long long)up + 2] - qz;
        const long long d = dx * dx + dy * dy + dz * dz;
        if (d < best || (d == best && up < arg)) { best = d; arg = up; }
        more_up = ++up < n_b;
      }
    }
    if (more_dn) {
      const long long dx = (long long)b[3 * (long long)dn] - qx;
      if (dx * dx > best) more_dn = false;
      else {
        const long long dy = (long long)b[3 * (long long)dn + 1] - qy, dz = (long long)b[3 * (long long)dn + 2] - qz;
        const long long d = dx * dx + dy * dy + dz * dz;
        if (d < best || (d == best && dn < arg)) { best = d; arg = dn; }
        more_dn = --dn >= 0;
      }
    }
  }
  d2_out[i] = best;
  nn_out[i] = arg;
}

extern "C" int pcc_nn_sorted_x(const int32_t* a_xyz, int64_t n_a, const int32_t* b_xyz, int64_t n_b, int64_t* d2,
                               int32_t* nn, void* stream) {
  if (n_a <= 0) return PCC_OK;
  PCC_REQUIRE(a_xyz && b_xyz && d2 && nn, "pcc_nn_sorted_x: NULL array");
  PCC_REQUIRE(n_b >= 1 && n_b < (1ll << 31), "pcc_nn_sorted_x: B must hold 1 .. 2^31-1 points");
  k_nn_sorted_x<<<(unsigned)pcc_cdiv(n_a, 256), 256, 0, (hipStream_t)stream>>>(a_xyz, n_a, b_xyz, (int)n_b,
                                                                              (long long*)d2, nn);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}
