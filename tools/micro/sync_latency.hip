// How long after the end of a stream's work does the host know? hipStreamSynchronize against polling
// a word of pinned host memory that the last operation of the stream writes (a one-thread kernel
// storing to host memory / hipStreamWriteValue32). The work: a ~1 ms kernel + a 32 KB device-to-host
// copy, as between quantizer and code construction in mgh_compress. Dev tool:
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/sync_latency tools/micro/sync_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(long long cycles, unsigned *out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (out) out[0] = 1;
}
__global__ void flag(volatile unsigned *host_word, unsigned v) { *host_word = v; __threadfence_system(); }
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  unsigned *dbuf; CK(hipMalloc(&dbuf, 32768));
  unsigned *pin; CK(hipHostMalloc(&pin, 32768 + 64, hipHostMallocDefault));
  volatile unsigned *word = pin + 8192;
  hipEvent_t ev; CK(hipEventCreate(&ev));
  const long long cyc = 100000;  // 100 MHz wall clock: 1 ms
  auto now = [] { return std::chrono::steady_clock::now(); };
  for (int mode = 0; mode < 4; mode++) {
    // total wall time of the sequence, launch to knowledge
    std::vector<double> tot;
    for (int it = 0; it < 50; it++) {
      *word = 0;
      const auto t0 = now();
      spin<<<1, 64, 0, st>>>(cyc, dbuf);
      CK(hipMemcpyAsync(pin, dbuf, 32768, hipMemcpyDeviceToHost, st));
      if (mode == 1) flag<<<1, 1, 0, st>>>(word, 1u);
      if (mode == 2) { if (hipStreamWriteValue32(st, (void *)word, 1u, 0) != hipSuccess) { (void)hipGetLastError(); break; } }
      if (mode == 3) CK(hipEventRecord(ev, st));
      if (mode == 0) CK(hipStreamSynchronize(st));
      else if (mode == 3) { while (hipEventQuery(ev) == hipErrorNotReady) {} }
      else { while (*word == 0) { __builtin_ia32_pause(); } }
      tot.push_back(std::chrono::duration<double, std::micro>(now() - t0).count());
      CK(hipStreamSynchronize(st));
    }
    if (tot.empty()) continue;
    std::sort(tot.begin(), tot.end());
    const char *names[4] = {"hipStreamSynchronize", "poll a word written by a one-thread kernel", "poll a word written by hipStreamWriteValue32", "spin on hipEventQuery"};
    printf("%-48s median %8.1f us  min %8.1f us (1000 us of it is the kernel)\n", names[mode], tot[tot.size() / 2], tot[0]);
  }
  return 0;
}
