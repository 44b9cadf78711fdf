// Streaming Thomas solves (kernels_ipk_stream.hpp) against the LDS-staged ones (kernels_ipk.hpp):
// bit-exactness and time on a cube of n^3 coarse nodes, plus the latency of the dependent chains.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I mgard_amd/csrc \
//     -o tools/micro/ipk_stream tools/micro/ipk_stream.hip
//   tools/micro/ipk_stream [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "kernels_ipk.hpp"
#include "kernels_ipk_stream.hpp"
using namespace mgh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- latency of the dependent chains (registers only) ---------------------------------------
template <int MODE>
__global__ void __launch_bounds__(64) k_chain(int steps, float a0, float w, float b, float y, float *out) {
  float prev = a0 + threadIdx.x * 1e-3f;
  for (int i = 0; i < steps; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (MODE == 0) prev = a0 - prev * w;                          // forward step
      if (MODE == 1) prev = (a0 - w * prev) / b;                    // backward step, IEEE
      if (MODE == 2) prev = div_markstein<float>(a0 - w * prev, b, y);  // backward step, FMA
    }
  }
  if (prev == 12345.f) out[0] = prev;
}

// streaming baselines: in-place scale of N floats, 16 bytes per lane / 4 bytes per lane
__global__ void __launch_bounds__(256) k_scale4(float4 *x, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 v = x[i]; v.x *= 2; v.y *= 2; v.z *= 2; v.w *= 2; x[i] = v;
  }
}
__global__ void __launch_bounds__(256) k_scale1(float *x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] *= 2;
}
__global__ void __launch_bounds__(256) k_scale_add(float *x, float *a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] += x[i];
}

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? atoi(argv[1]) : 257;
  const size_t N = (size_t)n * n * n;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float *dummy; CK(hipMalloc(&dummy, 64));
  if (argc <= 2) {
    for (int wps : {1, 2, 4}) {
      const int steps = 2000;
      auto run = [&](const char *nm, auto k) {
        float best = 1e9;
        for (int it = 0; it < 4; it++) {
          hipEventRecord(e0);
          k<<<1024 * wps, 64>>>(steps, 1.0f, 0.25f, 0.6f, 1.0f / 0.6f, dummy);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
        }
        printf("chain %-10s %d wave(s)/SIMD: %.2f ns per step\n", nm, wps, best * 1e6 / (steps * 16));
      };
      run("fwd", k_chain<0>);
      run("bwd-ieee", k_chain<1>);
      run("bwd-fma", k_chain<2>);
    }
  }

  // ---- data, tables ---------------------------------------------------------------------------
  float *x, *x0, *ref, *tt, *add, *add0;
  CK(hipMalloc(&x, N * 4)); CK(hipMalloc(&x0, N * 4)); CK(hipMalloc(&ref, N * 4));
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
  CK(hipFuncSetAttribute((const void *)k_ipk_lds_contig<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void *)k_ipk_lds_strided<float, 48>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const uint32_t np = n * n, magic = (uint32_t)((((uint64_t)1 << 32) + n - 1) / n);
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

  // one variant of the streaming kernel; dir: 2 = f (contiguous), 1 = c, 0 = r (+ add)
  auto stream = [&](int dir, uint32_t W, uint32_t lds_batches_max, auto kern, int U, int KR, const char *tag) {
    const uint32_t nb = n / U;
    if ((int)nb < KR) return;
    const uint32_t parked = nb - KR;  // batches parked outside the registers
    const uint32_t lb = std::min(parked, lds_batches_max);
    const uint32_t n_glob = (parked - lb) * U;
    const size_t lds = (size_t)W * lb * U * 4;
    if (lds > 160 * 1024) return;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const unsigned blocks = ((np + W - 1) / W + 7) / 8 * 8;
    char name[128];
    snprintf(name, sizeof name, "stream %s dir=%d W=%u U=%d KR=%d lds=%zuK n_glob=%u", tag, dir, W, U, KR, lds / 1024, n_glob);
    timeit(name, dir == 0, [&] {
      if (dir == 2) kern<<<blocks, 64, lds>>>(np, np, 0, n, 1, n, W, n_glob, x, tt, nullptr, 1);
      if (dir == 1) kern<<<blocks, 64, lds>>>(np, n, (size_t)n * n, 1, n, n, W, n_glob, x, tt, nullptr, 1);
      if (dir == 0) kern<<<blocks, 64, lds>>>(np, np, 0, 1, (size_t)n * n, n, W, n_glob, x, tt, add, 1);
    }, true);
  };

  for (int blocks : {1024, 2048, 4096, 16384}) {
    char nm[64];
    snprintf(nm, sizeof nm, "baseline scale float4 blocks=%d", blocks);
    timeit(nm, false, [&] { k_scale4<<<blocks, 256>>>((float4 *)x, N / 4); }, false);
    snprintf(nm, sizeof nm, "baseline scale float blocks=%d", blocks);
    timeit(nm, false, [&] { k_scale1<<<blocks, 256>>>(x, N); }, false);
    snprintf(nm, sizeof nm, "baseline add float blocks=%d", blocks);
    timeit(nm, true, [&] { k_scale_add<<<blocks, 256>>>(x, add, N); }, false);
  }
  for (int dir : {2, 1, 0}) {
    // reference: the LDS-staged kernels
    if (dir == 2)
      timeit("lds contig P=48", false, [&] { k_ipk_lds_contig<float><<<(np + 47) / 48, 256, 48 * (n + (n % 2 ? 0 : 1)) * 4>>>(np, n, n % 2 ? 0 : 1, magic, 48, x, tt, nullptr, 1, 0); }, false);
    if (dir == 1)
      timeit("lds strided<48> c", false, [&] { k_ipk_lds_strided<float, 48><<<((np + 47) / 48 + 7) / 8 * 8, 256, 48 * n * 4>>>(n, n, (size_t)n * n, n, n, x, tt, nullptr, 1); }, false);
    if (dir == 0)
      timeit("lds strided<48> r add", true, [&] { k_ipk_lds_strided<float, 48><<<((np + 47) / 48 + 7) / 8 * 8, 256, 48 * n * 4>>>(1, n * n, (size_t)n * n, (size_t)n * n, n, x, tt, add, 1); }, false);
    set_ref(dir == 0);
#define VARIANTS(U, KR, PD)                                                                     \
    for (uint32_t W : {60u, 48u})                                                               \
      for (uint32_t lbm : {1000u, 4u}) {                                                         \
        if (dir == 2) {                                                                         \
          stream(dir, W, lbm, k_ipk_stream<float, U, KR, PD, true, false>, U, KR, "ieee pd" #PD);          \
          stream(dir, W, lbm, k_ipk_stream<float, U, KR, PD, true, true>, U, KR, "fma  pd" #PD);           \
        } else {                                                                                \
          stream(dir, W, lbm, k_ipk_stream<float, U, KR, PD, false, false>, U, KR, "ieee pd" #PD);         \
          stream(dir, W, lbm, k_ipk_stream<float, U, KR, PD, false, true>, U, KR, "fma  pd" #PD);          \
        }                                                                                       \
      }
    VARIANTS(16, 8, 4)
    VARIANTS(16, 8, 2)
    VARIANTS(16, 8, 1)
    VARIANTS(16, 4, 4)
  }
  CK(hipDeviceSynchronize());
  return 0;
}
