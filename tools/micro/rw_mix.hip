// What does a PURE stream with the traffic mix of the top-level pass reach on this device?
// 512^3 floats in (512 MiB), 512^3 int64 out (1 GiB) [+ 256 MiB of float side outputs], every
// element touched once, no arithmetic to speak of. Variants: 4 B loads + 8 B stores (what the tile
// kernel issues), plain / nontemporal; 16 B loads + 2 x 16 B stores; segment = contiguous run a
// wave writes (the tile kernel: 256 B).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/rw_mix tools/micro/rw_mix.hip && tools/micro/rw_mix
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
template <bool NT> __global__ void __launch_bounds__(256) k_4_8(const float *__restrict__ in, int64_t *__restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int64_t v = (int64_t)(int)in[i];
    if (NT) __builtin_nontemporal_store(v, &out[i]); else out[i] = v;
  }
}
template <bool NT> __global__ void __launch_bounds__(256) k_16_32(const float4 *__restrict__ in, longlong2 *__restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = in[i];
    longlong2 a = {(int64_t)(int)v.x, (int64_t)(int)v.y}, b = {(int64_t)(int)v.z, (int64_t)(int)v.w};
    if (NT) {
      __builtin_nontemporal_store(a.x, &out[2 * i].x); __builtin_nontemporal_store(a.y, &out[2 * i].y);
      __builtin_nontemporal_store(b.x, &out[2 * i + 1].x); __builtin_nontemporal_store(b.y, &out[2 * i + 1].y);
    } else { out[2 * i] = a; out[2 * i + 1] = b; }
  }
}
// the tile kernel's mix: + two float side arrays of n/8 elements each... (coarse + load vector)
template <bool NT> __global__ void __launch_bounds__(256) k_mix(const float *__restrict__ in, int64_t *__restrict__ out, float *__restrict__ s1, float *__restrict__ s2, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float x = in[i];
    const int64_t v = (int64_t)(int)x;
    if (NT) __builtin_nontemporal_store(v, &out[i]); else out[i] = v;
    if ((i & 7) == 0) { s1[i >> 3] = x; s2[i >> 3] = x + 1.f; }
  }
}
__global__ void __launch_bounds__(256) k_read(const float4 *__restrict__ in, float *out, size_t n4) {
  float m = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { float4 v = in[i]; m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w))); }
  if (m == 12345.f) out[0] = 1;
}
template <bool NT> __global__ void __launch_bounds__(256) k_write(int64_t *__restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store((int64_t)i, &out[i]); else out[i] = (int64_t)i;
  }
}
template <typename F> static float timeit(F f, int reps = 20) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) f();
  hipEventRecord(a);
  for (int i = 0; i < reps; i++) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1000.f;
}
int main() {
  const size_t n = (size_t)512 * 512 * 512;
  float *in, *s1, *s2; int64_t *out;
  if (hipMalloc(&in, n * 4) != hipSuccess || hipMalloc(&out, n * 8) != hipSuccess) return 1;
  hipMalloc(&s1, n / 2); hipMalloc(&s2, n / 2);
  hipMemset(in, 0, n * 4);
  for (int g : {2048, 8192, 32768}) {
    printf("grid %5d: ", g);
    float t;
    t = timeit([&] { k_4_8<false><<<g, 256>>>(in, out, n); }); printf("4/8 plain %6.1f us (%.2f TB/s)  ", t, 12.0 * n / t / 1e6);
    t = timeit([&] { k_4_8<true><<<g, 256>>>(in, out, n); }); printf("4/8 nt %6.1f us (%.2f)  ", t, 12.0 * n / t / 1e6);
    t = timeit([&] { k_16_32<false><<<g, 256>>>((const float4 *)in, (longlong2 *)out, n / 4); }); printf("16/32 plain %6.1f us (%.2f)  ", t, 12.0 * n / t / 1e6);
    t = timeit([&] { k_16_32<true><<<g, 256>>>((const float4 *)in, (longlong2 *)out, n / 4); }); printf("16/32 nt %6.1f us (%.2f)\n", t, 12.0 * n / t / 1e6);
    printf("            ");
    t = timeit([&] { k_mix<false><<<g, 256>>>(in, out, s1, s2, n); }); printf("mix plain %6.1f us (%.2f TB/s)  ", t, 13.0 * n / t / 1e6);
    t = timeit([&] { k_mix<true><<<g, 256>>>(in, out, s1, s2, n); }); printf("mix nt %6.1f us (%.2f)  ", t, 13.0 * n / t / 1e6);
    t = timeit([&] { k_read<<<g, 256>>>((const float4 *)in, s1, n / 4); }); printf("read %6.1f us (%.2f)  ", t, 4.0 * n / t / 1e6);
    t = timeit([&] { k_write<false><<<g, 256>>>(out, n); }); printf("write plain %6.1f us (%.2f)  ", t, 8.0 * n / t / 1e6);
    t = timeit([&] { k_write<true><<<g, 256>>>(out, n); }); printf("write nt %6.1f us (%.2f)\n", t, 8.0 * n / t / 1e6);
  }
  return 0;
}
