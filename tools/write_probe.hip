// What write bandwidth can a kernel reach on this chip, and how much of it does the dense-product store pattern get?
// Stand-alone probe (hipcc --offload-arch=gfx950 -O3 tools/write_probe.hip -o /tmp/write_probe && /tmp/write_probe):
// writes the 5.1 GB per-pair buffer of the level-2 composite convolution (58 051 x 21 952 fp32) with
//   0  linear 16-byte stores (grid-stride): the ceiling
//   1  k_gemm_h2's epilogue pattern: 128 x 128 tiles, 4-byte stores, a wave instruction = two 128-byte row segments
//   2  128 x 128 tiles, 16-byte stores: a wave instruction = two 512-byte row segments
//   3  32 x 512 tiles, 4-byte stores (2 KB contiguous per row and tile)
//   4  pattern 1 with the tile order of k_gemm_h2 (8 row tiles per column block, contiguous work ranges per XCD)
// No arithmetic, no loads: pure store streams, HIP-event timed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_linear(float4* p, size_t n4) {
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = v;
}

__device__ inline void tile_of(int wid, int gy, int mode, int& tile, int& colblock) {
  if (mode == 4 && gy > 8) {
    const int g = wid / (8 * gy), rem = wid - g * 8 * gy;
    colblock = rem >> 3; tile = g * 8 + (rem & 7);
  } else { tile = wid / gy; colblock = wid - tile * gy; }
}

template <int MODE>
__global__ void __launch_bounds__(256, 3) k_tile(float* out, long long rows, unsigned ncol) {
  extern __shared__ float dyn_lds[];     // (dynamic LDS only limits how many workgroups share a CU)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (rows < 0) out[0] = dyn_lds[tid];
  int wid = blockIdx.x;
  if (MODE == 4) { const int cpx = gridDim.x >> 3; wid = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3); }
  if (MODE == 3) {                       // 32 rows x 512 columns
    const int gy = (ncol + 511) / 512;
    const int tile = wid / gy, cb = wid - tile * gy;
    const long long r0 = (long long)tile * 32;
    if (r0 >= rows) return;
    for (int e = 0; e < 64; ++e) {       // 8 rows per wave, 8 column segments of 64 lanes
      const int r = w * 8 + (e >> 3);
      const unsigned c = cb * 512u + (e & 7) * 64u + lane;
      if (r0 + r < rows && c < ncol) out[(size_t)(r0 + r) * ncol + c] = (float)e;
    }
    return;
  }
  const int gy = (ncol + 127) / 128;
  int tile, cb;
  tile_of(wid, gy, MODE, tile, cb);
  const long long r0 = (long long)tile * 128;
  if (r0 >= rows) return;
  const int wm = w >> 1, wn = w & 1, half = lane >> 5, r31 = lane & 31;
  if (MODE == 2) {                       // 16-byte stores: lane -> 4 consecutive columns, 32 lanes = 512 B of a row, two rows per instruction
    for (int e = 0; e < 16; ++e) {
      const int r = w * 32 + e * 2 + half;
      const unsigned c = cb * 128u + r31 * 4u;
      if (r0 + r < rows && c + 3 < ncol) *reinterpret_cast<float4*>(out + (size_t)(r0 + r) * ncol + c) = make_float4(1.f, 2.f, 3.f, (float)e);
    }
    return;
  }
  for (int i = 0; i < 2; ++i)
    for (int e = 0; e < 16; ++e)
      for (int j = 0; j < 2; ++j) {
        const int r = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        const unsigned c = cb * 128u + wn * 64 + j * 32 + r31;
        if (r0 + r < rows && c < ncol) out[(size_t)(r0 + r) * ncol + c] = (float)e;
      }
}

int main() {
  const long long rows = 58051;
  const unsigned ncol = 21952;
  const size_t n = (size_t)rows * ncol;
  float* d;
  CK(hipMalloc(&d, n * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)k_tile<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  CK(hipFuncSetAttribute((const void*)k_tile<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  CK(hipFuncSetAttribute((const void*)k_tile<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  CK(hipFuncSetAttribute((const void*)k_tile<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  const char* names[5] = {"linear 16-byte stores", "128x128 tiles, 4-byte stores (k_gemm_h2 epilogue)", "128x128 tiles, 16-byte stores",
                          "32x512 tiles, 4-byte stores", "128x128 tiles, 4-byte stores, k_gemm_h2 tile order"};
  for (int occ = 3; occ >= 1; --occ)
  for (int mode = 0; mode < 5; ++mode) {
    // occ workgroups (4 waves each) per CU: a wave holds at most 63 memory operations in flight (vmcnt), so the bytes a CU
    // keeps in flight scale with the waves that are storing AND with the bytes per store instruction
    const size_t lds = occ == 3 ? 0 : (occ == 2 ? 70 * 1024 : 100 * 1024);
    if (mode == 0 && occ != 3) continue;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0, 0));
      const unsigned t128 = (unsigned)((rows + 127) / 128), gy = (ncol + 127) / 128;
      unsigned g128 = t128 * gy;
      if (mode == 0) k_linear<<<256 * 16, 256>>>((float4*)d, n / 4);
      else if (mode == 1) k_tile<1><<<g128, 256, lds>>>(d, rows, ncol);
      else if (mode == 2) k_tile<2><<<g128, 256, lds>>>(d, rows, ncol);
      else if (mode == 3) k_tile<3><<<(unsigned)((rows + 31) / 32) * ((ncol + 511) / 512), 256, lds>>>(d, rows, ncol);
      else { g128 = (((t128 + 7) / 8 * 8) * gy + 7) / 8 * 8; k_tile<4><<<g128, 256, lds>>>(d, rows, ncol); }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    printf("%d workgroups/CU  %-56s %7.3f ms  %6.2f TB/s\n", occ, names[mode], best, n * 4 / (best * 1e-3) / 1e12);
  }
  return 0;
}
