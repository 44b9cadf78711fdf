// VALU issue rate and dependent latency on gfx950, in shader cycles (s_memtime) and wall time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/micro/valu_rate tools/micro/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: dependent mul+sub chain; 1: 8 independent mul+sub chains; 2: dependent fma chain;
// 3: 8 independent fma chains; 4: 8 independent v_pk_mul/v_pk_add chains (float2)
template <int MODE>
__global__ void __launch_bounds__(256) k(int steps, float a0, float w, unsigned long long *cyc, float *out) {
  float p[8];
  for (int k = 0; k < 8; k++) p[k] = a0 + threadIdx.x * 1e-3f + k;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 q[8];
  for (int k = 0; k < 8; k++) q[k] = f2{p[k], p[k] + 1};
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < steps; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (MODE == 0) p[0] = a0 - p[0] * w;
      if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 8; k++) p[k] = a0 - p[k] * w;
      }
      if (MODE == 2) p[0] = __builtin_fmaf(p[0], w, a0);
      if (MODE == 3) {
#pragma unroll
        for (int k = 0; k < 8; k++) p[k] = __builtin_fmaf(p[k], w, a0);
      }
      if (MODE == 4) {
#pragma unroll
        for (int k = 0; k < 8; k++) q[k] = f2{a0, a0} - q[k] * f2{w, w};
      }
    }
  }
  const unsigned long long t1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int k = 0; k < 8; k++) s += p[k] + q[k].x + q[k].y;
  if (s == 12345.f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

int main() {
  unsigned long long *cyc; float *out;
  CK(hipMalloc(&cyc, 16)); CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int steps = 20000;
  auto run = [&](const char *nm, auto kern, int ops_per_step, int wps, int bs = 64) {
    float best = 1e9; unsigned long long h[2] = {0, 0};
    for (int it = 0; it < 3; it++) {
      hipEventRecord(e0);
      kern<<<1024 * wps * 64 / bs, bs>>>(steps, 1.0f, 0.25f, cyc, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
    const double nops = (double)steps * 8 * ops_per_step;
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, bs, 0);
    printf("bs=%3d occ/CU=%2d %-28s %d wave/SIMD: %6.2f ns/op/wave (events), %6.2f s_memtime ticks/op, %6.2f ns/op (wall_clock64 @100MHz)\n",
           bs, occ, nm, wps, best * 1e6 / nops, (double)h[0] / nops, (double)h[1] * 10.0 / nops);
  };
  for (int wps : {1, 2, 4}) for (int bs : {256, 128}) { run("dep mul+sub", k<0>, 2, wps, bs); run("8 indep mul+sub", k<1>, 16, wps, bs); }
  for (int wps : {1, 2, 4, 8}) {
    run("dep mul+sub", k<0>, 2, wps);
    run("8 indep mul+sub", k<1>, 16, wps);
    run("dep fma", k<2>, 1, wps);
    run("8 indep fma", k<3>, 8, wps);
    run("8 indep pk_mul+pk_sub", k<4>, 16, wps);
  }
  return 0;
}
