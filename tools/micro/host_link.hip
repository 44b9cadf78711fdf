// What the host <-> device link of the box gives, piece by piece (round 6, the host-buffer path of
// mgh_compress / mgh_decompress): pinned DMA rates by transfer size, both directions at once, the
// host-side copy pageable -> pinned by thread count, first touch of fresh pages, registration of
// caller memory, a kernel reading pinned memory itself.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/host_link tools/micro/host_link.hip -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      std::printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(1);                                                                 \
    }                                                                               \
  } while (0)

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void par(int nt, size_t bytes, const std::function<void(size_t, size_t)> &f) {
  std::vector<std::thread> th;
  const size_t part = (bytes / nt + 4095) / 4096 * 4096;
  for (int t = 0; t < nt; t++) {
    const size_t lo = std::min(bytes, t * part), hi = std::min(bytes, (t + 1) * part);
    th.emplace_back([=, &f] { if (hi > lo) f(lo, hi); });
  }
  for (auto &x : th) x.join();
}

__global__ void k_read_host(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n, float *out) {
  float m = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = src[i];
    dst[i] = v;
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  if (m > 1e30f) *out = m;
}
__global__ void k_write_host(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

int main(int argc, char **argv) {
  const size_t B = (argc > 1 ? (size_t)atol(argv[1]) : 512) << 20;
  std::printf("cpus online %ld, page %ld\n", sysconf(_SC_NPROCESSORS_ONLN), sysconf(_SC_PAGESIZE));
  void *dev, *dev2, *pin, *pin2;
  CK(hipMalloc(&dev, B));
  CK(hipMalloc(&dev2, B));
  CK(hipHostMalloc(&pin, B, hipHostMallocDefault));
  CK(hipHostMalloc(&pin2, B, hipHostMallocDefault));
  std::memset(pin, 1, B);
  std::memset(pin2, 1, B);
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));

  // 1. pinned DMA, one copy of the whole buffer and in chunks
  for (int dir = 0; dir < 2; dir++) {
    for (size_t chunk : {B, (size_t)64 << 20, (size_t)16 << 20, (size_t)4 << 20, (size_t)1 << 20}) {
      double best = 1e9;
      for (int rep = 0; rep < 4; rep++) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (size_t off = 0; off < B; off += chunk) {
          if (dir == 0) CK(hipMemcpyAsync((char *)dev + off, (char *)pin + off, std::min(chunk, B - off), hipMemcpyHostToDevice, s1));
          else CK(hipMemcpyAsync((char *)pin + off, (char *)dev + off, std::min(chunk, B - off), hipMemcpyDeviceToHost, s1));
        }
        CK(hipStreamSynchronize(s1));
        best = std::min(best, now() - t0);
      }
      std::printf("pinned %s chunk %4zu MB: %.2f ms  %.1f GB/s\n", dir ? "D2H" : "H2D", chunk >> 20, best * 1e3, B / best / 1e9);
    }
  }
  // 2. both directions at once
  {
    double best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
      CK(hipDeviceSynchronize());
      const double t0 = now();
      CK(hipMemcpyAsync(dev, pin, B, hipMemcpyHostToDevice, s1));
      CK(hipMemcpyAsync(pin2, dev2, B, hipMemcpyDeviceToHost, s2));
      CK(hipStreamSynchronize(s1));
      CK(hipStreamSynchronize(s2));
      best = std::min(best, now() - t0);
    }
    std::printf("pinned H2D + D2H at once: %.2f ms  %.1f GB/s each way\n", best * 1e3, B / best / 1e9);
  }
  // 3. kernel reads / writes pinned memory itself
  {
    float *flag;
    CK(hipMalloc(&flag, 4));
    for (int grid : {256, 1024, 4096}) {
      double best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        k_read_host<<<grid, 256, 0, s1>>>((const float4 *)pin, (float4 *)dev, B / 16, flag);
        CK(hipStreamSynchronize(s1));
        best = std::min(best, now() - t0);
      }
      std::printf("kernel reads pinned host (grid %d): %.2f ms  %.1f GB/s\n", grid, best * 1e3, B / best / 1e9);
      best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        k_write_host<<<grid, 256, 0, s1>>>((const float4 *)dev, (float4 *)pin, B / 16);
        CK(hipStreamSynchronize(s1));
        best = std::min(best, now() - t0);
      }
      std::printf("kernel writes pinned host (grid %d): %.2f ms  %.1f GB/s\n", grid, best * 1e3, B / best / 1e9);
    }
  }
  // 4. host copy pageable -> pinned and pinned -> pageable by thread count
  void *page = std::malloc(B);
  {
    double t0 = now();
    std::memset(page, 2, B);
    std::printf("first touch of %zu MB malloc (memset, 1 thread): %.1f ms\n", B >> 20, (now() - t0) * 1e3);
    for (int nt : {1, 2, 4, 8, 16, 32, 64}) {
      double b1 = 1e9, b2 = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        par(nt, B, [&](size_t lo, size_t hi) { std::memcpy((char *)pin + lo, (char *)page + lo, hi - lo); });
        b1 = std::min(b1, now() - t0);
        t0 = now();
        par(nt, B, [&](size_t lo, size_t hi) { std::memcpy((char *)page + lo, (char *)pin + lo, hi - lo); });
        b2 = std::min(b2, now() - t0);
      }
      std::printf("memcpy %2d threads: pageable->pinned %.1f GB/s, pinned->pageable %.1f GB/s\n", nt, B / b1 / 1e9, B / b2 / 1e9);
    }
  }
  // 5. first touch of fresh pages: plain, parallel, MAP_POPULATE, huge pages
  {
    for (int nt : {1, 8, 32}) {
      void *p = mmap(nullptr, B, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      double t0 = now();
      par(nt, B, [&](size_t lo, size_t hi) { for (size_t o = lo; o < hi; o += 4096) ((volatile char *)p)[o] = 0; });
      std::printf("fresh mmap, touch with %2d threads: %.1f ms\n", nt, (now() - t0) * 1e3);
      munmap(p, B);
    }
    {
      double t0 = now();
      void *p = mmap(nullptr, B, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
      std::printf("fresh mmap MAP_POPULATE: %.1f ms\n", (now() - t0) * 1e3);
      munmap(p, B);
    }
    {
      void *p = mmap(nullptr, B, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      double t0 = now();
      int r = madvise(p, B, MADV_HUGEPAGE);
      par(8, B, [&](size_t lo, size_t hi) { for (size_t o = lo; o < hi; o += 4096) ((volatile char *)p)[o] = 0; });
      std::printf("fresh mmap MADV_HUGEPAGE (rc %d), touch with 8 threads: %.1f ms\n", r, (now() - t0) * 1e3);
      munmap(p, B);
    }
    {
      double t0 = now();
      void *p = std::malloc(B);
      CK(hipMemcpy(p, dev, B, hipMemcpyDeviceToHost));
      std::printf("hipMemcpy D2H into a fresh malloc: %.1f ms\n", (now() - t0) * 1e3);
      t0 = now();
      CK(hipMemcpy(p, dev, B, hipMemcpyDeviceToHost));
      std::printf("hipMemcpy D2H into the same (touched) malloc: %.1f ms\n", (now() - t0) * 1e3);
      t0 = now();
      CK(hipMemcpy(dev, p, B, hipMemcpyHostToDevice));
      std::printf("hipMemcpy H2D from pageable: %.1f ms\n", (now() - t0) * 1e3);
      std::free(p);
    }
  }
  // 5b. what mgh_decompress does for an output it allocates: posix_memalign + MADV_HUGEPAGE + 8 touching threads + free
  for (int rep = 0; rep < 3; rep++) {
    double t0 = now();
    void *p = nullptr;
    int rc = posix_memalign(&p, (size_t)2 << 20, B);
    int r = madvise(p, B, MADV_HUGEPAGE);
    const double ta = now() - t0;
    t0 = now();
    par(8, B, [&](size_t lo, size_t hi) { for (size_t o = lo; o < hi; o += 4096) ((volatile char *)p)[o] = 0; });
    const double tt = now() - t0;
    t0 = now();
    CK(hipMemcpy(p, dev, B, hipMemcpyDeviceToHost));
    const double tc = now() - t0;
    t0 = now();
    std::free(p);
    std::printf("posix_memalign(rc %d) + madvise(rc %d) %.2f ms, touch 8 threads %.1f ms, hipMemcpy D2H %.1f ms, free %.1f ms\n", rc, r,
                ta * 1e3, tt * 1e3, tc * 1e3, (now() - t0) * 1e3);
  }
  // 6. registration of caller memory
  {
    for (int rep = 0; rep < 2; rep++) {
      double t0 = now();
      hipError_t e = hipHostRegister(page, B, hipHostRegisterDefault);
      const double tr = now() - t0;
      if (e != hipSuccess) {
        std::printf("hipHostRegister failed: %s\n", hipGetErrorString(e));
        break;
      }
      t0 = now();
      CK(hipMemcpyAsync(dev, page, B, hipMemcpyHostToDevice, s1));
      CK(hipStreamSynchronize(s1));
      const double tc = now() - t0;
      t0 = now();
      CK(hipHostUnregister(page));
      std::printf("hipHostRegister %.1f ms, H2D out of it %.2f ms (%.1f GB/s), unregister %.1f ms\n", tr * 1e3, tc * 1e3,
                  B / tc / 1e9, (now() - t0) * 1e3);
    }
  }
  // 7. pipelined staging: K persistent threads copy chunk c into ring slot, main thread queues the DMA
  for (size_t chunk : {(size_t)4 << 20, (size_t)8 << 20, (size_t)16 << 20, (size_t)32 << 20}) {
    for (int nt : {4, 8, 16}) {
      constexpr int kSlots = 4;
      hipEvent_t ev[kSlots];
      for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      const size_t nch = (B + chunk - 1) / chunk;
      double best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        // threads split every chunk (a barrier per chunk through atomics)
        std::atomic<size_t> filled{0};  // chunks completely copied into their slot
        std::atomic<size_t> freed{kSlots};  // chunks whose slot may be overwritten: c < freed
        std::vector<std::atomic<int>> parts(nch);
        for (auto &p : parts) p.store(0);
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
          th.emplace_back([&, t] {
            for (size_t c = 0; c < nch; c++) {
              while (freed.load(std::memory_order_acquire) <= c) {}
              const size_t nb = std::min(chunk, B - c * chunk);
              const size_t part = (nb / nt + 4095) / 4096 * 4096;
              const size_t lo = std::min(nb, t * part), hi = std::min(nb, (t + 1) * part);
              if (hi > lo) std::memcpy((char *)pin + (c % kSlots) * chunk + lo, (char *)page + c * chunk + lo, hi - lo);
              if (parts[c].fetch_add(1, std::memory_order_acq_rel) + 1 == nt) filled.store(c + 1, std::memory_order_release);
            }
          });
        size_t waited = 0;
        for (size_t c = 0; c < nch; c++) {
          while (filled.load(std::memory_order_acquire) <= c) {}
          const size_t nb = std::min(chunk, B - c * chunk);
          CK(hipMemcpyAsync((char *)dev + c * chunk, (char *)pin + (c % kSlots) * chunk, nb, hipMemcpyHostToDevice, s1));
          CK(hipEventRecord(ev[c % kSlots], s1));
          // free the slot of the oldest DMA still counted as busy
          while (waited + kSlots - 1 <= c) {
            CK(hipEventSynchronize(ev[waited % kSlots]));
            waited++;
            freed.store(waited + kSlots, std::memory_order_release);
          }
        }
        CK(hipStreamSynchronize(s1));
        for (auto &x : th) x.join();
        best = std::min(best, now() - t0);
      }
      std::printf("staged H2D ring: chunk %2zu MB, %2d threads: %.2f ms  %.1f GB/s\n", chunk >> 20, nt, best * 1e3, B / best / 1e9);
      for (auto &e : ev) CK(hipEventDestroy(e));
    }
  }
  // check the last staging moved the right bytes
  {
    std::vector<char> back(1 << 20);
    CK(hipMemcpy(back.data(), (char *)dev + B - back.size(), back.size(), hipMemcpyDeviceToHost));
    std::printf("staged copy tail %s\n", std::memcmp(back.data(), (char *)page + B - back.size(), back.size()) ? "DIFFERS" : "ok");
  }
  return 0;
}
