// k_ipk_dma (kernels_ipk_dma.hpp) against k_ipk_stream (kernels_ipk_stream.hpp): bit-exactness and
// time on a cube of n^3 coarse nodes (strided directions: 1 = c, 0 = r with AddND).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I mgard_amd/csrc \
//     -o tools/micro/ipk_dma tools/micro/ipk_dma.hip
//   tools/micro/ipk_dma [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "kernels_ipk.hpp"
#include "kernels_ipk_stream.hpp"
#include "kernels_ipk_dma.hpp"
using namespace mgh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(64) k_empty(float *x) { if (x == nullptr) x[0] = 1; }
__global__ void __launch_bounds__(64) k_touch(float *x, uint32_t nb) { x[(size_t)blockIdx.x * 64 + threadIdx.x] += 1.0f; }

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? atoi(argv[1]) : 257;
  const size_t N = (size_t)n * n * n;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float *x, *x0, *tt, *add, *add0;
  CK(hipMalloc(&x, N * 4)); CK(hipMalloc(&x0, N * 4));
  CK(hipMalloc(&add, N * 4)); CK(hipMalloc(&add0, N * 4)); CK(hipMalloc(&tt, 4 * n * 4));
  std::vector<float> h(N), t(4 * n);
  {  // Thomas tables of a uniform grid (Hierarchy.hpp:124-155 in float)
    const float hh = 1.0f / (float)(n - 1);
    std::vector<float> ha(n + 1, 0.f), hb(n + 1, 0.f), am(n + 1, 0.f), bm(n + 1, 0.f);
    hb[0] = 2 * hh / 6;
    for (uint32_t i = 1; i + 1 < n; i++) { float a = hh / 6, w = a / hb[i - 1]; hb[i] = 2 * (hh + hh) / 6 - w * a; ha[i] = a; }
    { float a = hh / 6, w = a / hb[n - 2]; hb[n - 1] = 2 * hh / 6 - w * a; ha[n - 1] = a; }
    for (uint32_t i = 0; i < n; i++) am[i] = ha[i];
    bm[0] = 1;
    for (uint32_t i = 0; i < n; i++) bm[i + 1] = hb[i];
    for (uint32_t i = 0; i < n; i++) { t[i] = am[i] / bm[i]; t[n + i] = am[i + 1]; t[2 * n + i] = bm[i + 1]; t[3 * n + i] = 1.0f / bm[i + 1]; }
  }
  for (size_t i = 0; i < N; i++) {
    const uint32_t r = (uint32_t)((i * 2654435761u) >> 8);
    float v = 1e-5f * (1.0f + 0.5f * (float)(r % 1000) / 1000.0f) * ((r & 4096) ? -1.f : 1.f);
    if (r % 97 == 0) v = 0.0f;
    h[i] = v;
  }
  CK(hipMemcpy(x0, h.data(), N * 4, hipMemcpyHostToDevice));
  for (size_t i = 0; i < N; i++) h[i] = 1.0f + (float)(i % 777) * 1e-3f;
  CK(hipMemcpy(add0, h.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(tt, t.data(), 4 * n * 4, hipMemcpyHostToDevice));
  const uint32_t np = n * n;
  std::vector<float> hr(N), hx(N);
  auto timeit = [&](const char *name, bool with_add, auto fn, bool check) {
    float tot = 0, best = 1e9;
    for (int i = 0; i < 9; i++) {
      hipMemcpy(x, x0, N * 4, hipMemcpyDeviceToDevice);
      if (with_add) hipMemcpy(add, add0, N * 4, hipMemcpyDeviceToDevice);
      hipEventRecord(e0);
      fn();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (i >= 3) tot += ms;
      if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    size_t bad = 0;
    if (check) {
      CK(hipMemcpy(hx.data(), with_add ? add : x, N * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < N; i++) bad += memcmp(&hx[i], &hr[i], 4) != 0;
    }
    printf("n=%u %-44s avg %7.1f us  best %7.1f us%s\n", n, name, tot / 6 * 1000, best * 1000,
           check ? (bad ? "  MISMATCH" : "  bit-exact") : "");
    if (bad) printf("   %zu of %zu differ\n", bad, N);
  };
  auto set_ref = [&](bool with_add) { CK(hipMemcpy(hr.data(), with_add ? add : x, N * 4, hipMemcpyDeviceToHost)); };
  auto old_stream = [&](int dir, uint32_t W, auto kern, int U, int KR, const char *tag) {
    const uint32_t nb = n / U, parked = nb - KR;
    const size_t lds = (size_t)W * parked * U * 4;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const unsigned blocks = ((np + W - 1) / W + 7) / 8 * 8;
    char name[128];
    snprintf(name, sizeof name, "k_ipk_stream %s dir=%d W=%u lds=%zuK", tag, dir, W, lds / 1024);
    timeit(name, dir == 0, [&] {
      if (dir == 1) kern<<<blocks, 64, lds>>>(np, n, (size_t)n * n, 1, n, n, W, 0, x, tt, nullptr, 1);
      if (dir == 0) kern<<<blocks, 64, lds>>>(np, np, 0, 1, (size_t)n * n, n, W, 0, x, tt, add, 1);
    }, false);
  };
  auto dma = [&](int dir, auto kern, int U, int KR, const char *tag) {
    if ((int)(n / U) < KR) return;
    const uint32_t nl = n - KR * U;
    const size_t lds = (size_t)nl * 64 * 4;
    if (lds > 160 * 1024) return;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const unsigned blocks = ((np + 63) / 64 + 7) / 8 * 8;
    char name[128];
    snprintf(name, sizeof name, "k_ipk_dma %s dir=%d KR=%d lds=%.1fK", tag, dir, KR, lds / 1024.0);
    timeit(name, dir == 0, [&] {
      if (dir == 1) kern<<<blocks, 64, lds>>>(np, n, (size_t)n * n, 1, n, n, x, tt, nullptr);
      if (dir == 0) kern<<<blocks, 64, lds>>>(np, np, 0, 1, (size_t)n * n, n, x, tt, add);
    }, true);
#ifdef MGH_IPK_DBG
    unsigned long long hd[8][8];
    CK(hipMemcpyFromSymbol(hd, HIP_SYMBOL(g_ipk_dbg), sizeof hd));
    for (int t = 0; t < 8; t++) {
      printf("   tile %4d (10 ns ticks since its start):", t * 128 + 5);
      for (int k = 1; k < 8; k++) printf(" %6lld", (long long)(hd[t][k] - hd[t][0]));
      printf("   start %lld\n", (long long)(hd[t][0] - hd[0][0]));
    }
#endif
  };
  timeit("empty kernel, 1040 x 64, 28 KB LDS", false, [&] { k_empty<<<1040, 64, 28 * 1024>>>(x); }, false);
  timeit("empty kernel, 1040 x 64", false, [&] { k_empty<<<1040, 64>>>(x); }, false);
  timeit("touch kernel, 1040 x 64 (256 KB rmw)", false, [&] { k_touch<<<1040, 64>>>(x, 0); }, false);
  for (int dir : {1, 0}) {
    old_stream(dir, 60, k_ipk_stream<float, 16, 8, 1, false, false>, 16, 8, "pd1");
    set_ref(dir == 0);
    old_stream(dir, 64, k_ipk_stream<float, 16, 8, 1, false, false>, 16, 8, "pd1");
    old_stream(dir, 60, k_ipk_stream<float, 16, 8, 2, false, false>, 16, 8, "pd2");
    if (dir == 1) {
      dma(dir, k_ipk_dma<float, 16, 8, 0>, 16, 8, "");
      dma(dir, k_ipk_dma<float, 16, 9, 0>, 16, 9, "");
      dma(dir, k_ipk_dma<float, 16, 10, 0>, 16, 10, "");
      dma(dir, k_ipk_dma<float, 16, 6, 0>, 16, 6, "");
    } else {
      dma(dir, k_ipk_dma<float, 16, 8, 1>, 16, 8, "");
      dma(dir, k_ipk_dma<float, 16, 9, 1>, 16, 9, "");
      dma(dir, k_ipk_dma<float, 16, 10, 1>, 16, 10, "");
      dma(dir, k_ipk_dma<float, 16, 6, 1>, 16, 6, "");
    }
  }
  CK(hipDeviceSynchronize());
  return 0;
}
