// Microbenchmark (round 5, verdict item 2b): what does a grid-wide barrier inside ONE persistent
// kernel cost on this GPU, against the launch-to-launch cost of dependent kernels on one stream?
// The question behind it: levels <= 7 of the 512^3 step are 13 dependent launches of 7-24 us each;
// would one cooperative kernel with barriers between the phases be cheaper?
//   barrier: sense-reversing counter in global memory (one atomic per workgroup, lane 0 spins on a
//   generation word with agent-scope loads), measured with `work` = 0 and with a small dependent
//   phase between the barriers (every workgroup writes a line, reads its neighbour's: what a phase
//   boundary of the level kernels needs to be visible).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/grid_barrier tools/micro/grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned *count, unsigned *gen, unsigned nblocks) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    if (atomicAdd(count, 1u) == nblocks - 1) {
      __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence();
      __hip_atomic_store(gen, g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) k_persistent(unsigned *count, unsigned *gen, float *buf, int rounds, int work) {
  const unsigned nb = gridDim.x;
  float acc = 0;
  for (int r = 0; r < rounds; r++) {
    if (work) {
      buf[(size_t)blockIdx.x * 256 + threadIdx.x] = acc + r;
      grid_barrier(count, gen, nb);
      acc += buf[(size_t)((blockIdx.x + 1) % nb) * 256 + threadIdx.x];
    } else {
      grid_barrier(count, gen, nb);
    }
  }
  if (acc == 12345.f) buf[0] = acc;
}

__global__ void __launch_bounds__(256) k_phase(float *buf, int r, int work) {
  if (work) {
    const unsigned nb = gridDim.x;
    float v = buf[(size_t)((blockIdx.x + 1) % nb) * 256 + threadIdx.x];
    buf[(size_t)blockIdx.x * 256 + threadIdx.x + (size_t)nb * 256] = v + r;
  }
}

int main() {
  unsigned *sync;
  float *buf;
  CK(hipMalloc(&sync, 256));
  CK(hipMalloc(&buf, (size_t)8192 * 256 * 2 * sizeof(float)));
  CK(hipMemset(buf, 0, (size_t)8192 * 256 * 2 * sizeof(float)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int rounds = 200;
  for (int work = 0; work < 2; work++)
    for (unsigned nb : {64u, 256u, 512u, 1024u}) {
      // co-residency: 256 CUs x up to 8 workgroups of 256 threads
      float best = 1e9f;
      for (int rep = 0; rep < 5; rep++) {
        CK(hipMemset(sync, 0, 256));
        CK(hipEventRecord(e0));
        k_persistent<<<nb, 256>>>(sync, sync + 32, buf, rounds, work);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      float bestl = 1e9f;
      for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < rounds; r++) k_phase<<<nb, 256>>>(buf, r, work);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        bestl = ms < bestl ? ms : bestl;
      }
      printf("work %d  %4u workgroups: grid barrier %.2f us each (persistent kernel, %d rounds), dependent launches %.2f us each\n",
             work, nb, best * 1e3f / rounds, rounds, bestl * 1e3f / rounds);
    }
  return 0;
}
