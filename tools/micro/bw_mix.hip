// Microbenchmark: what does a 4-byte-read / 8-byte-write stream (the quantizer's traffic mix)
// reach on this GPU, as a function of store width and cache policy?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_conv(const float *__restrict__ in, int64_t *__restrict__ out, size_t n) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
  if (MODE == 0) {  // 4-B load, 8-B store
    for (size_t i = tid; i < n; i += nth) out[i] = (int64_t)(int)in[i];
  } else if (MODE == 1) {  // 8-B load, 16-B store
    const float2 *i2 = (const float2 *)in; longlong2 *o2 = (longlong2 *)out;
    for (size_t i = tid; i < n / 2; i += nth) { float2 v = i2[i]; longlong2 r; r.x = (int)v.x; r.y = (int)v.y; o2[i] = r; }
  } else if (MODE == 2) {  // 16-B load, 2 x 16-B store
    const float4 *i4 = (const float4 *)in; longlong2 *o2 = (longlong2 *)out;
    for (size_t i = tid; i < n / 4; i += nth) { float4 v = i4[i]; longlong2 a, b; a.x = (int)v.x; a.y = (int)v.y; b.x = (int)v.z; b.y = (int)v.w; o2[2 * i] = a; o2[2 * i + 1] = b; }
  } else if (MODE == 3) {  // 8-B load, 16-B nontemporal store
    const float2 *i2 = (const float2 *)in; longlong2 *o2 = (longlong2 *)out;
    for (size_t i = tid; i < n / 2; i += nth) { float2 v = i2[i]; __builtin_nontemporal_store((int64_t)(int)v.x, &o2[i].x); __builtin_nontemporal_store((int64_t)(int)v.y, &o2[i].y); }
  } else if (MODE == 4) {  // 4-B load, 8-B store, 7 strided streams like the reordered layout
    const size_t seg = n / 8;
    for (size_t i = tid; i < seg; i += nth) {
#pragma unroll
      for (int k = 0; k < 8; k++) out[k * seg + i] = (int64_t)(int)in[k * seg + i];
    }
  } else if (MODE == 5) {  // read only
    float m = 0;
    const float4 *i4 = (const float4 *)in;
    for (size_t i = tid; i < n / 4; i += nth) { float4 v = i4[i]; m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w))); }
    if (m == 12345.f) out[0] = 1;
  } else if (MODE == 6) {  // write only 16 B
    longlong2 *o2 = (longlong2 *)out; longlong2 r; r.x = 1; r.y = 2;
    for (size_t i = tid; i < n / 2; i += nth) o2[i] = r;
  }
}

template <int MODE> float run(const float *in, int64_t *out, size_t n, int blocks) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) k_conv<MODE><<<blocks, 256>>>(in, out, n);
  hipEventRecord(a);
  for (int i = 0; i < 10; i++) k_conv<MODE><<<blocks, 256>>>(in, out, n);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}

int main() {
  const size_t n = (size_t)512 * 512 * 512;
  float *in; int64_t *out;
  CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 8));
  CK(hipMemset(in, 0, n * 4));
  const char *names[] = {"ld4 st8", "ld8 st16", "ld16 2xst16", "ld8 st16 nt", "8 streams ld4 st8", "read only ld16", "write only st16"};
  const double bytes[] = {12, 12, 12, 12, 12, 4, 8};
  for (int blocks : {2048, 8192, 32768, 131072}) {
    float t[7] = {run<0>(in, out, n, blocks), run<1>(in, out, n, blocks), run<2>(in, out, n, blocks), run<3>(in, out, n, blocks),
                  run<4>(in, out, n, blocks), run<5>(in, out, n, blocks), run<6>(in, out, n, blocks)};
    for (int m = 0; m < 7; m++) printf("blocks %6d  %-20s %.3f ms  %.2f TB/s\n", blocks, names[m], t[m], bytes[m] * n / t[m] / 1e9);
  }
  return 0;
}
