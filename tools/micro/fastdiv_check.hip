// Exhaustive check of the division used in the backward Thomas sweep against the IEEE
// quotient: for each divisor b of a list, EVERY float numerator a with 2^-80 <= |a| <= 2^80.
//   y = RN(1/b);  q0 = RN(a y);  r0 = fma(-b, q0, a);  q1 = fma(r0, y, q0);
//   r1 = fma(-b, q1, a);  q2 = fma(r1, y, q1)             (Markstein: q2 = RN(a / b))
// Also samples doubles (2^36 per divisor) for the f64 variant.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/micro/fastdiv_check tools/micro/fastdiv_check.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

__device__ __forceinline__ float fdiv(float a, float b, float y) {
  float q = a * y;
  float r = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, y, q);
}
__device__ __forceinline__ double fdiv(double a, double b, double y) {
  double q = a * y;
  double r = __builtin_fma(-b, q, a);
  q = __builtin_fma(r, y, q);
  r = __builtin_fma(-b, q, a);
  return __builtin_fma(r, y, q);
}

__global__ void k32(const float *bs, int nb, unsigned long long *bad, unsigned long long *tested) {
  const float b = bs[blockIdx.y], y = 1.0f / b;
  unsigned long long nbad = 0, nt = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32);
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const float a = __uint_as_float((unsigned)i);
    const float m = fabsf(a);
    if (!(m >= 0x1p-80f && m <= 0x1p80f)) continue;
    nt++;
    if (__float_as_uint(fdiv(a, b, y)) != __float_as_uint(a / b)) nbad++;
  }
  atomicAdd(bad, nbad);
  atomicAdd(tested, nt);
}

__device__ unsigned long long mix(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
__global__ void k64(const double *bs, int nb, unsigned long long per_b, unsigned long long *bad,
                    unsigned long long *tested) {
  const double b = bs[blockIdx.y], y = 1.0 / b;
  unsigned long long nbad = 0, nt = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_b;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    // random sign / mantissa, exponent in [-900, 900]; every 4th sample close to a multiple of b
    unsigned long long h = mix(i * 0x9e3779b97f4a7c15ull + blockIdx.y);
    const long long e = (long long)(mix(h) % 1801) - 900;
    unsigned long long bits = (h & 0x800fffffffffffffull) | ((unsigned long long)(e + 1023) << 52);
    double a = __longlong_as_double((long long)bits);
    if ((i & 3) == 0) a = b * (double)(long long)(h >> 40) * (1.0 + ((h & 1) ? 0x1p-52 : -0x1p-53));
    const double m = fabs(a);
    if (!(m >= 0x1p-900 && m <= 0x1p900)) continue;
    nt++;
    if (__double_as_longlong(fdiv(a, b, y)) != __double_as_longlong(a / b)) nbad++;
  }
  atomicAdd(bad, nbad);
  atomicAdd(tested, nt);
}

int main() {
  std::vector<float> bs;
  // the bm table of a uniform 257-node level (normalised and unnormalised coordinates)
  for (float h : {1.0f / 256.0f, 1.0f, 1.0f / 512.0f, 2.0f}) {
    float hb = 2 * h / 6;
    bs.push_back(hb);
    for (int i = 1; i < 40; i++) { const float a = h / 6, w = a / hb; hb = 2 * (h + h) / 6 - w * a; bs.push_back(hb); }
  }
  std::mt19937_64 g(7);
  for (int i = 0; i < 64; i++) {  // random mantissas, exponents in [-20, 20]
    unsigned u = (unsigned)(g() & 0x7fffff) | ((unsigned)(127 - 20 + (int)(g() % 41)) << 23);
    float f; memcpy(&f, &u, 4); bs.push_back(f);
  }
  for (unsigned man : {0u, 1u, 0x7fffffu, 0x7ffffeu, 0x400000u, 0x3fffffu, 0x555555u, 0x2aaaaau})
    for (int e : {-20, -1, 0, 1, 20}) { unsigned u = man | ((unsigned)(127 + e) << 23); float f; memcpy(&f, &u, 4); bs.push_back(f); }
  const int nb = (int)bs.size();
  float *dbs; unsigned long long *cnt;
  hipMalloc(&dbs, nb * 4); hipMalloc(&cnt, 16); hipMemset(cnt, 0, 16);
  hipMemcpy(dbs, bs.data(), nb * 4, hipMemcpyHostToDevice);
  k32<<<dim3(2048, nb), 256>>>(dbs, nb, cnt, cnt + 1);
  unsigned long long h[2];
  hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
  printf("f32: %d divisors, %llu quotients compared, %llu differ from IEEE\n", nb, h[1], h[0]);

  std::vector<double> bd;
  for (float f : bs) bd.push_back((double)f * (1.0 + 0x1p-30));
  for (int i = 0; i < 64; i++) {
    unsigned long long u = (g() & 0xfffffffffffffull) | ((unsigned long long)(1023 - 100 + (int)(g() % 201)) << 52);
    double d; memcpy(&d, &u, 8); bd.push_back(d);
  }
  for (unsigned long long man : {0ull, 1ull, 0xfffffffffffffull, 0xffffffffffffeull, 0x8000000000000ull, 0x7ffffffffffffull})
    for (int e : {-100, 0, 100}) { unsigned long long u = man | ((unsigned long long)(1023 + e) << 52); double d; memcpy(&d, &u, 8); bd.push_back(d); }
  const int nd = (int)bd.size();
  double *dbd; hipMalloc(&dbd, nd * 8); hipMemcpy(dbd, bd.data(), nd * 8, hipMemcpyHostToDevice);
  hipMemset(cnt, 0, 16);
  k64<<<dim3(2048, nd), 256>>>(dbd, nd, 1ull << 32, cnt, cnt + 1);
  hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
  printf("f64: %d divisors, %llu quotients compared (sampled), %llu differ from IEEE\n", nd, h[1], h[0]);
  return (int)(h[0] != 0);
}
