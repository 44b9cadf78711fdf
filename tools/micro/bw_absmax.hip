// Microbenchmark: the norm reduction (k_absmax) against a plain read-only stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../mgard_amd/csrc/kernels_v1.hpp"

template <int U>
__global__ void __launch_bounds__(256) k_read(const float *__restrict__ in, unsigned long long *out, size_t n) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
  float m = 0;
  const float4 *i4 = (const float4 *)in;
  size_t i = tid;
  for (; i + (U - 1) * nth < n / 4; i += U * nth) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = i4[i + u * nth];
#pragma unroll
    for (int u = 0; u < U; u++) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
  }
  for (; i < n / 4; i += nth) { float4 v = i4[i]; m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)))); }
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__float_as_uint(m));
}

template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) f();
  hipEventRecord(a);
  for (int i = 0; i < 10; i++) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}

int main() {
  const size_t n = (size_t)512 * 512 * 512;
  float *in; unsigned long long *out;
  hipMalloc(&in, n * 4); hipMalloc(&out, 8);
  std::vector<float> h(n);
  for (size_t i = 0; i < n; i++) h[i] = (float)((i * 2654435761u) & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int blocks : {1024, 2048, 4096, 8192, 32768, 131072}) {
    float t0 = timeit([&] { mgh::k_absmax<float><<<blocks, 256>>>(in, n, out); });
    float t1 = timeit([&] { k_read<1><<<blocks, 256>>>(in, out, n); });
    float t2 = timeit([&] { k_read<2><<<blocks, 256>>>(in, out, n); });
    float t4 = timeit([&] { k_read<4><<<blocks, 256>>>(in, out, n); });
    printf("blocks %6d: k_absmax %.3f ms (%.2f TB/s) | read U1 %.3f (%.2f) U2 %.3f (%.2f) U4 %.3f (%.2f)\n", blocks, t0, 4.0 * n / t0 / 1e9,
           t1, 4.0 * n / t1 / 1e9, t2, 4.0 * n / t2 / 1e9, t4, 4.0 * n / t4 / 1e9);
  }
  return 0;
}
