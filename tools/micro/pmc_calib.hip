// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths of the fused level
// kernel (the guide: "calibrate on a known byte count in your own access pattern"): 4-byte-per-lane
// contiguous loads, 8-byte-per-lane stores, plain and nontemporal, over a KNOWN number of bytes
// (512^3 floats in = 512 MiB, 512^3 int64 out = 1024 MiB; every launch touches everything once).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/pmc_calib tools/micro/pmc_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out/f -- tools/micro/pmc_calib   (then WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void __launch_bounds__(256) calib_ld4_st8_plain(const float *__restrict__ in, int64_t *__restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (int64_t)(int)in[i];
}
__global__ void __launch_bounds__(256) calib_ld4_st8_nt(const float *__restrict__ in, int64_t *__restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store((int64_t)(int)in[i], &out[i]);
}
__global__ void __launch_bounds__(256) calib_ld16_readonly(const float4 *__restrict__ in, int64_t *__restrict__ out, size_t n4) {
  float m = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { float4 v = in[i]; m = fmaxf(m, v.x + v.y + v.z + v.w); }
  if (m == 12345.f) out[0] = 1;
}
int main() {
  const size_t n = (size_t)512 * 512 * 512;
  float *in; int64_t *out;
  if (hipMalloc(&in, n * 4) != hipSuccess || hipMalloc(&out, n * 8) != hipSuccess) return 1;
  hipMemset(in, 0, n * 4);
  for (int it = 0; it < 4; it++) {
    calib_ld4_st8_plain<<<8192, 256>>>(in, out, n);
    calib_ld4_st8_nt<<<8192, 256>>>(in, out, n);
    calib_ld16_readonly<<<8192, 256>>>((const float4 *)in, out, n / 4);
  }
  hipDeviceSynchronize();
  printf("done: each launch reads %zu KiB%s\n", n * 4 / 1024, " and (first two kernels) writes twice that");
  return 0;
}
