// How many kernels does hipMemsetAsync launch for a given size / value width?  (rocprofv3 --kernel-trace --stats -- ./memset_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
int main() {
  char* p; hipMalloc(&p, 256 << 20);
  hipStream_t s; hipStreamCreate(&s);
  const size_t sizes[] = {128u << 20, (128u << 20) + 8, 8000000, 8000008, 1234568, 296, 65536 * 8, 65536 * 8 + 8, 4096 * 8 + 8, 1024};
  for (size_t sz : sizes) {
    for (int r = 0; r < 3; ++r) hipMemsetAsync(p, 0, sz, s);
    hipStreamSynchronize(s);
    printf("size %zu done\n", sz);
  }
  // same with an offset base
  for (int r = 0; r < 3; ++r) hipMemsetAsync(p + 8, 0, 8000000, s);
  hipStreamSynchronize(s);
  return 0;
}
