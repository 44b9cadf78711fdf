// The usage pattern of the reference's examples/mgard-x/HighLevelAPIs/Example.cpp and
// HighLevelAPIsGPUBuffer/Example.cu (host buffers, device buffers, pre-allocated and
// library-allocated outputs) written against `namespace mgard_x` exactly as there -- the only
// change to the calls is the include line -- plus the checks a test needs: the round-trip error
// against the requested bound, the pinned-memory trio and the status codes of what this library
// does not provide.
#include "compress_x_hip.hpp"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#define REQUIRE(x) do { if (!(x)) { std::printf("FAILED: %s (line %d)\n", #x, __LINE__); return 1; } } while (0)

int main() {
  mgard_x::SIZE n1 = 40;
  mgard_x::SIZE n2 = 50;
  mgard_x::SIZE n3 = 60;
  const size_t n = n1 * n2 * n3;

  // prepare
  double *in_array_cpu = new double[n];
  for (size_t i = 0; i < n; i++) {
    const double x = (double)(i % n3) / n3, y = (double)((i / n3) % n2) / n2, z = (double)(i / (n2 * n3)) / n1;
    in_array_cpu[i] = std::sin(6.0 * x) * std::cos(4.0 * y) + 0.5 * std::sin(9.0 * z);
  }
  double maxabs = 0;
  for (size_t i = 0; i < n; i++) maxabs = std::max(maxabs, std::fabs(in_array_cpu[i]));
  void *compressed_array_cpu = NULL;
  size_t compressed_size;
  std::vector<mgard_x::SIZE> shape{n1, n2, n3};
  double tol = 0.01, s = std::numeric_limits<double>::infinity();

  // ---- HighLevelAPIs/Example.cpp: host in, library-allocated host out ----
  mgard_x::Config config;
  config.lossless = mgard_x::lossless_type::Huffman_Zstd;
  config.dev_type = mgard_x::device_type::HIP;
  REQUIRE(mgard_x::compress(3, mgard_x::data_type::Double, shape, tol, s,
                            mgard_x::error_bound_type::REL, in_array_cpu,
                            compressed_array_cpu, compressed_size, config, false) ==
          mgard_x::compress_status_type::Success);
  REQUIRE(compressed_array_cpu != NULL && compressed_size > 0 && compressed_size < n * sizeof(double));
  void *decompressed_array_cpu = NULL;
  std::vector<mgard_x::SIZE> got_shape;
  mgard_x::data_type got_type;
  REQUIRE(mgard_x::decompress(compressed_array_cpu, compressed_size, decompressed_array_cpu, got_shape,
                              got_type, config, false) == mgard_x::compress_status_type::Success);
  REQUIRE(got_shape == shape && got_type == mgard_x::data_type::Double);
  double err = 0;
  for (size_t i = 0; i < n; i++) err = std::max(err, std::fabs(((double *)decompressed_array_cpu)[i] - in_array_cpu[i]));
  REQUIRE(err <= tol * maxabs);
  // a backend this library does not have
  config.dev_type = mgard_x::device_type::SERIAL;
  void *dummy = NULL;
  REQUIRE(mgard_x::decompress(compressed_array_cpu, compressed_size, dummy, config, false) ==
          mgard_x::compress_status_type::BackendNotAvailableFailure);
  config.dev_type = mgard_x::device_type::HIP;

  // ---- HighLevelAPIsGPUBuffer/Example.cu: device buffers, pre-allocated outputs ----
  double *in_array_gpu = nullptr;
  REQUIRE(hipMalloc((void **)&in_array_gpu, n * sizeof(double)) == hipSuccess);
  REQUIRE(hipMemcpy(in_array_gpu, in_array_cpu, n * sizeof(double), hipMemcpyDefault) == hipSuccess);
  void *compressed_array_gpu = nullptr;
  REQUIRE(hipMalloc((void **)&compressed_array_gpu, n * sizeof(double) + 1e6) == hipSuccess);
  void *decompressed_array_gpu = nullptr;
  REQUIRE(hipMalloc((void **)&decompressed_array_gpu, n * sizeof(double)) == hipSuccess);
  config.lossless = mgard_x::lossless_type::Huffman;
  size_t compressed_size_gpu = n * sizeof(double) + 1e6;
  REQUIRE(mgard_x::compress(3, mgard_x::data_type::Double, shape, tol, s,
                            mgard_x::error_bound_type::REL, in_array_gpu, compressed_array_gpu,
                            compressed_size_gpu, config, true) == mgard_x::compress_status_type::Success);
  REQUIRE(mgard_x::decompress(compressed_array_gpu, compressed_size_gpu, decompressed_array_gpu, config,
                              true) == mgard_x::compress_status_type::Success);
  std::vector<double> back(n);
  REQUIRE(hipMemcpy(back.data(), decompressed_array_gpu, n * sizeof(double), hipMemcpyDefault) == hipSuccess);
  err = 0;
  for (size_t i = 0; i < n; i++) err = std::max(err, std::fabs(back[i] - in_array_cpu[i]));
  REQUIRE(err <= tol * maxabs);
  // output too small
  size_t tiny = 64;
  REQUIRE(mgard_x::compress(3, mgard_x::data_type::Double, shape, tol, s, mgard_x::error_bound_type::REL,
                            in_array_gpu, compressed_array_gpu, tiny, config, true) ==
          mgard_x::compress_status_type::OutputTooLargeFailure);

  // ---- pinned host memory (compress_x.hpp:166-178) ----
  REQUIRE(!mgard_x::check_memory_pinned(in_array_cpu, config));
  mgard_x::pin_memory(in_array_cpu, n * sizeof(double), config);
  REQUIRE(mgard_x::check_memory_pinned(in_array_cpu, config));
  void *c2 = NULL;
  size_t c2_size = 0;
  REQUIRE(mgard_x::compress(3, mgard_x::data_type::Double, shape, tol, s, mgard_x::error_bound_type::ABS,
                            in_array_cpu, c2, c2_size, false) == mgard_x::compress_status_type::Success);
  mgard_x::unpin_memory(in_array_cpu, config);
  REQUIRE(!mgard_x::check_memory_pinned(in_array_cpu, config));
  REQUIRE(mgard_x::release_cache(config) == mgard_x::compress_status_type::Success);

  std::free(compressed_array_cpu);
  std::free(decompressed_array_cpu);
  std::free(c2);
  (void)hipFree(in_array_gpu);
  (void)hipFree(compressed_array_gpu);
  (void)hipFree(decompressed_array_gpu);
  delete[] in_array_cpu;
  std::cout << "OK\n";
  return 0;
}
