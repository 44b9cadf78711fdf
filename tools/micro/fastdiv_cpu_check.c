// CPU check of the FMA quotient of the backward Thomas sweep (mgard_amd/csrc/kernels_ipk.hpp:
// div_markstein) against the IEEE quotient a / b, inside the windows the library uses:
//   float : divisor in [2^-40, 2^40],   numerator in [2^-80, 2^80]
//   double: divisor in [2^-450, 2^450], numerator in [2^-500, 2^500]
// Random pairs plus adversarial ones: divisors with all-ones / all-zero / nearly such significands,
// numerators at, just below and just above b * (q + ulp/2) and b * q for random quotients q (the
// rounding midpoints and the exact cases). fmaf / fma of glibc are correctly rounded.
//   gcc -O2 -fopenmp -ffp-contract=off -mfma -o fastdiv_cpu_check tools/micro/fastdiv_cpu_check.c -lm
//   ./fastdiv_cpu_check [log2 of the number of pairs per type, default 31]
#include <math.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <omp.h>
static inline float fdivf(float a, float b, float y) {
  float q = a * y;
  float r = fmaf(-b, q, a);
  q = fmaf(r, y, q);
  r = fmaf(-b, q, a);
  return fmaf(r, y, q);
}
static inline double fdivd(double a, double b, double y) {
  double q = a * y;
  double r = fma(-b, q, a);
  q = fma(r, y, q);
  r = fma(-b, q, a);
  return fma(r, y, q);
}
static inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
int main(int argc, char **argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 31;
  const long nblk = 1L << (lg > 19 ? lg - 19 : 0);
  unsigned long long bad32 = 0, bad64 = 0, n32 = 0, n64 = 0;
  #pragma omp parallel for reduction(+:bad32,bad64,n32,n64) schedule(dynamic)
  for (long blk = 0; blk < nblk; blk++) {
    for (long k = 0; k < (1 << 19); k++) {
      uint64_t h = mix((uint64_t)blk * (1 << 19) + k + 12345);
      // divisor: random mantissa (every 8th: all ones / one / near power of two), exponent in [-40, 40]
      uint32_t mb = (uint32_t)(h & 0x7fffff);
      if ((k & 7) == 1) mb = 0x7fffff; if ((k & 7) == 2) mb = 0; if ((k & 7) == 3) mb = 0x7ffffe; if ((k & 7) == 4) mb = 1;
      int eb = (int)((h >> 24) % 81) - 40;
      uint32_t bb = ((uint32_t)(eb + 127) << 23) | mb;
      float b; memcpy(&b, &bb, 4);
      float y = 1.0f / b;
      uint64_t g = mix(h);
      int ea = (int)((g >> 40) % 161) - 80;
      uint32_t ab = ((uint32_t)(g & 1) << 31) | ((uint32_t)(ea + 127) << 23) | (uint32_t)((g >> 8) & 0x7fffff);
      float a; memcpy(&a, &ab, 4);
      if ((k & 3) == 0) {  // numerator close to b * (random 24-bit quotient): near-exact and near-midpoint cases
        float q; uint32_t qb = ((uint32_t)(127 + (int)((g >> 50) % 41) - 20) << 23) | (uint32_t)((g >> 16) & 0x7fffff);
        memcpy(&q, &qb, 4);
        double t = (double)b * (double)q;          // exact product (48 bits)
        double mid = t + 0.5 * (double)b * ldexp(1.0, ilogbf(q) - 23);  // b * (q + ulp/2)
        a = (float)((g & 2) ? t : mid);
        if (g & 4) a = nextafterf(a, (g & 8) ? INFINITY : -INFINITY);
        float m = fabsf(a);
        if (!(m >= 0x1p-80f && m <= 0x1p80f)) continue;
      }
      n32++;
      float want = a / b, got = fdivf(a, b, y);
      if (memcmp(&want, &got, 4)) bad32++;
      // double
      uint64_t mbd = mix(g) & 0xfffffffffffffull;
      if ((k & 7) == 1) mbd = 0xfffffffffffffull; if ((k & 7) == 2) mbd = 0;
      int ebd = (int)((h >> 30) % 901) - 450;
      uint64_t bbd = ((uint64_t)(ebd + 1023) << 52) | mbd; double bd; memcpy(&bd, &bbd, 8);
      int ead = (int)((g >> 20) % 1001) - 500;
      uint64_t abd = ((uint64_t)(ead + 1023) << 52) | (mix(g + 7) & 0xfffffffffffffull); double ad; memcpy(&ad, &abd, 8);
      double yd = 1.0 / bd;
      n64++;
      double wd = ad / bd, gd = fdivd(ad, bd, yd);
      if (memcmp(&wd, &gd, 8)) bad64++;
    }
  }
  printf("f32: %llu mismatches of %llu; f64: %llu of %llu\n", bad32, n32, bad64, n64);
  return (bad32 || bad64) ? 1 : 0;
}
