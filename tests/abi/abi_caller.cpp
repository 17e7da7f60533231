// A caller of libpcc_hip that is not Python: plain C++ host code over include/pcc_hip.h (extern "C", raw device
// pointers, sizes, a HIP stream).  It canonicalises a coordinate list (pack -> radix sort -> unique) and runs a 1x1
// convolution (the MFMA path) and checks both against host loops.  Built and run by tests/test_gpu_abi_caller.py.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pcc_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define PCC_CALL(x) do { int rc_ = (x); if (rc_ != PCC_OK) { std::printf("%s failed rc=%d: %s\n", #x, rc_, pcc_last_error()); return 3; } } while (0)

int main() {
  if (pcc_version() < 100) { std::printf("bad version\n"); return 1; }
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  // ---- coordinates: n rows (b,x,y,z) with duplicates and negative values ---------------------------------------------
  const int64_t n = 50000;
  std::vector<int32_t> C(n * 4);
  std::srand(7);
  for (int64_t i = 0; i < n; ++i) { C[4 * i] = std::rand() % 2; for (int c = 1; c < 4; ++c) C[4 * i + c] = std::rand() % 40 - 9; }
  int32_t* d_C; int64_t *d_keys, *d_sorted, *d_uniq, *d_count;
  HIP_OK(hipMalloc(&d_C, n * 16)); HIP_OK(hipMalloc(&d_keys, n * 8)); HIP_OK(hipMalloc(&d_sorted, n * 8));
  HIP_OK(hipMalloc(&d_uniq, n * 8)); HIP_OK(hipMalloc(&d_count, 8));
  HIP_OK(hipMemcpy(d_C, C.data(), n * 16, hipMemcpyHostToDevice));
  const size_t ws_bytes = std::max(pcc_sort_ws_bytes(n), pcc_unique_ws_bytes(n));
  void* d_ws; HIP_OK(hipMalloc(&d_ws, ws_bytes));
  PCC_CALL(pcc_keys_pack_i32(d_C, n, d_keys, stream));
  PCC_CALL(pcc_sort_keys(d_keys, n, ~0ull, d_sorted, nullptr, d_ws, ws_bytes, stream));
  PCC_CALL(pcc_unique_sorted(d_sorted, n, d_uniq, nullptr, d_count, d_ws, ws_bytes, stream));
  HIP_OK(hipStreamSynchronize(stream));
  int64_t count = 0;
  HIP_OK(hipMemcpy(&count, d_count, 8, hipMemcpyDeviceToHost));
  std::vector<int64_t> uniq(count);
  HIP_OK(hipMemcpy(uniq.data(), d_uniq, count * 8, hipMemcpyDeviceToHost));
  std::vector<int64_t> ref(n);
  for (int64_t i = 0; i < n; ++i)
    ref[i] = ((int64_t)C[4 * i] << 48) | ((int64_t)(C[4 * i + 1] + 32768) << 32) | ((int64_t)(C[4 * i + 2] + 32768) << 16) | (int64_t)(C[4 * i + 3] + 32768);
  std::sort(ref.begin(), ref.end());
  ref.erase(std::unique(ref.begin(), ref.end()), ref.end());
  if ((int64_t)ref.size() != count || !std::equal(ref.begin(), ref.end(), uniq.begin())) { std::printf("canonical set mismatch (%lld vs %zu)\n", (long long)count, ref.size()); return 4; }
  // ---- 1x1 convolution 32 -> 32 with bias and ReLU on the MFMA path ------------------------------------------------------
  const int64_t m = 5000; const int cin = 32, cout = 32;
  std::vector<float> X(m * cin), W(cin * cout), B(cout), Y(m * cout);
  for (auto& v : X) v = (std::rand() % 2001 - 1000) / 500.f;
  for (auto& v : W) v = (std::rand() % 2001 - 1000) / 4000.f;
  for (auto& v : B) v = (std::rand() % 2001 - 1000) / 1000.f;
  const int64_t pe = pcc_conv_packed_elems(1, cin, cout);
  if (pe <= 0) { std::printf("shape unsupported\n"); return 5; }
  float *d_X, *d_W, *d_B, *d_P, *d_Y;
  HIP_OK(hipMalloc(&d_X, X.size() * 4)); HIP_OK(hipMalloc(&d_W, W.size() * 4)); HIP_OK(hipMalloc(&d_B, B.size() * 4));
  HIP_OK(hipMalloc(&d_P, pe * 4)); HIP_OK(hipMalloc(&d_Y, Y.size() * 4));
  HIP_OK(hipMemcpy(d_X, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_W, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_B, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  if (pcc_conv_pack_weights(d_W, 1, cin, cout, d_P, pe - 1, stream) != PCC_EWS) { std::printf("undersized pack buffer was not refused\n"); return 6; }
  PCC_CALL(pcc_conv_pack_weights(d_W, 1, cin, cout, d_P, pe, stream));
  const size_t cws = pcc_conv_ws_bytes(m, 1, cin, cout);
  void* d_cws; HIP_OK(hipMalloc(&d_cws, cws));
  PCC_CALL(pcc_conv_fwd(d_X, m, cin, d_P, d_B, 1, cout, nullptr, nullptr, nullptr, m, d_Y, PCC_ACT_RELU, 0.f, d_cws, cws, PCC_ARITH_H3, nullptr, stream));
  HIP_OK(hipStreamSynchronize(stream));
  HIP_OK(hipMemcpy(Y.data(), d_Y, Y.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int64_t i = 0; i < m; ++i)
    for (int o = 0; o < cout; ++o) {
      double acc = B[o];
      for (int c = 0; c < cin; ++c) acc += (double)X[i * cin + c] * W[c * cout + o];
      const double want = acc > 0 ? acc : 0;
      worst = std::max(worst, std::fabs(want - Y[i * cout + o]) / (1.0 + std::fabs(want)));
    }
  if (worst > 1e-4) { std::printf("conv mismatch %.3e\n", worst); return 7; }
  std::printf("ABI OK: %lld unique keys, conv max rel err %.2e\n", (long long)count, worst);
  return 0;
}
