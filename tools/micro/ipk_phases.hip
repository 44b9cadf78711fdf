// Ablation of the LDS-staged Thomas solves at the top level of 512^3 (coarse box 257^3):
// build with -DMGH_IPK_PHASES=<bits> (1 stream in, 2 solve, 4 stream out) and compare.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMGH_IPK_PHASES=7 \
//     -I mgard_amd/csrc -o tools/micro/ipk_phases_7 tools/micro/ipk_phases.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "kernels_ipk.hpp"
using namespace mgh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? atoi(argv[1]) : 257;
  const size_t N = (size_t)n * n * n;
  float *x, *tt, *add;
  CK(hipMalloc(&x, N * 4)); CK(hipMalloc(&add, N * 4)); CK(hipMalloc(&tt, (4 * n + 1) * 4));
  std::vector<float> h(N, 1.0f), t(4 * n + 1);
  for (uint32_t i = 0; i < n; i++) { t[i] = 0.25f; t[n + i] = 0.15f; t[2 * n + i] = 0.6f; t[3 * n + i] = 1.0f / 0.6f; }
  t[4 * n] = getenv("IEEE_ONLY") ? 0.0f : 1.0f;
  CK(hipMemcpy(x, h.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(add, h.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(tt, t.data(), (4 * n + 1) * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void *)k_ipk_lds_contig<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void *)k_ipk_lds_strided<float, 48>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void *)k_ipk_lds_strided<float, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void *)k_ipk_lds_strided<float, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t np = n * n, magic = (uint32_t)((((uint64_t)1 << 32) + n - 1) / n);
  float *x0; CK(hipMalloc(&x0, N * 4));
  {  // smooth data + noise in the normal range, restored before every launch (the solves are in place)
    for (size_t i = 0; i < N; i++) h[i] = 1.0f + 0.5f * (float)((i * 2654435761u) % 1000) / 1000.0f;
    CK(hipMemcpy(x0, h.data(), N * 4, hipMemcpyHostToDevice));
  }
  auto timeit = [&](const char *name, auto fn) {
    float tot = 0;
    for (int i = 0; i < 13; i++) {
      hipMemcpy(x, x0, N * 4, hipMemcpyDeviceToDevice);
      hipEventRecord(e0);
      fn();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (i >= 3) tot += ms;
    }
    printf("phases=%d n=%u %-22s %7.1f us\n", MGH_IPK_PHASES, n, name, tot / 10 * 1000);
    return 0;
  };
  for (uint32_t P : {64u, 48u, 32u, 16u})
    timeit(("contig P=" + std::to_string(P)).c_str(), [&] {
      k_ipk_lds_contig<float><<<(np + P - 1) / P, 256, P * n * 4>>>(np, n, 0, magic, P, x, tt, nullptr, 1, 0); });
  if (MGH_IPK_PHASES == 2)  // solve only: one-wave workgroups (where do the solver waves land?)
    for (uint32_t P : {64u, 48u, 32u})
      timeit(("contig 1-wave P=" + std::to_string(P)).c_str(), [&] {
        k_ipk_lds_contig<float><<<(np + P - 1) / P, 64, P * n * 4>>>(np, n, 0, magic, P, x, tt, nullptr, 1, 0); });
  timeit("strided<48> c", [&] { k_ipk_lds_strided<float, 48><<<((np + 47) / 48 + 7) / 8 * 8, 256, 48 * n * 4>>>(n, n, (size_t)n * n, n, n, x, tt, nullptr, 1); });
  timeit("strided<64> c", [&] { k_ipk_lds_strided<float, 64><<<((np + 63) / 64 + 7) / 8 * 8, 256, 64 * n * 4>>>(n, n, (size_t)n * n, n, n, x, tt, nullptr, 1); });
  timeit("strided<32> c", [&] { k_ipk_lds_strided<float, 32><<<((np + 31) / 32 + 7) / 8 * 8, 256, 32 * n * 4>>>(n, n, (size_t)n * n, n, n, x, tt, nullptr, 1); });
  timeit("strided<48> r add", [&] { k_ipk_lds_strided<float, 48><<<((np + 47) / 48 + 7) / 8 * 8, 256, 48 * n * 4>>>(1, n * n, (size_t)n * n, (size_t)n * n, n, x, tt, add, 1); });
  timeit("strided<64> r add", [&] { k_ipk_lds_strided<float, 64><<<((np + 63) / 64 + 7) / 8 * 8, 256, 64 * n * 4>>>(1, n * n, (size_t)n * n, (size_t)n * n, n, x, tt, add, 1); });
  CK(hipDeviceSynchronize());
  return 0;
}
