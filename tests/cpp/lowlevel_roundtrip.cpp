// C++ consumer of include/mgard_hip.hpp, written the way a MGARD-X low-level API user drives
// mgard_x::Hierarchy / mgard_x::Compressor (reference doc/MGARD-X.md "low-level APIs",
// include/mgard-x/CompressionLowLevel/Compressor.hpp:193-272): stage by stage, then the fused
// entry points, checking that both give the same integers and that the round trip honours the
// error bound. Exit code 0 = pass.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "mgard_hip.hpp"

static void *dalloc(size_t n) {
  void *p = nullptr;
  return hipMalloc(&p, n) == hipSuccess ? p : nullptr;
}
static void dfree(void *p) { (void)hipFree(p); }

template <typename T> int run(std::vector<mgard_hip::SIZE> shape, double tol, T s) {
  using namespace mgard_hip;
  constexpr DIM D = 3;
  SIZE n = 1;
  for (auto x : shape) n *= x;
  std::vector<T> u(n);
  for (SIZE i = 0; i < n; i++) {
    const double x = (double)(i % shape[2]) / (shape[2] - 1), y = (double)((i / shape[2]) % shape[1]) / (shape[1] - 1);
    u[i] = (T)(std::sin(6.28318 * 3 * x) * std::cos(6.28318 * 2 * y) + 1e-3 * ((i * 2654435761u) % 1000) / 1000.0);
  }
  Config config;
  Hierarchy<D, T> hierarchy(shape, config);
  Compressor<D, T> compressor(hierarchy, config, DeviceAllocator{dalloc, dfree});
  T *d_data = (T *)dalloc(n * sizeof(T)), *d_orig = (T *)dalloc(n * sizeof(T));
  (void)hipMemcpy(d_orig, u.data(), n * sizeof(T), hipMemcpyHostToDevice);
  (void)hipMemcpy(d_data, d_orig, n * sizeof(T), hipMemcpyDeviceToDevice);

  // --- staged, like Compressor::Compress lines 216-218
  T norm = 1;
  compressor.CalculateNorm(d_data, error_bound_type::REL, s, norm);
  compressor.Decompose(d_data);
  compressor.Quantize(d_data, error_bound_type::REL, (T)tol, s, norm);
  ATOMIC_IDX count = 0;
  (void)hipMemcpy(&count, compressor.outlier_count_device(), sizeof(count), hipMemcpyDeviceToHost);
  std::vector<QUANTIZED_INT> q1(n);
  (void)hipMemcpy(q1.data(), compressor.quantized_array(), n * sizeof(QUANTIZED_INT), hipMemcpyDeviceToHost);
  if (count > compressor.outlier_capacity()) { std::printf("outlier overflow\n"); return 1; }

  // --- fused
  T norm2 = 0;
  compressor.DecomposeQuantize(d_orig, error_bound_type::REL, (T)tol, s, norm2);
  ATOMIC_IDX count2 = 0;
  (void)hipMemcpy(&count2, compressor.outlier_count_device(), sizeof(count2), hipMemcpyDeviceToHost);
  std::vector<QUANTIZED_INT> q2(n);
  (void)hipMemcpy(q2.data(), compressor.quantized_array(), n * sizeof(QUANTIZED_INT), hipMemcpyDeviceToHost);
  if (norm2 != norm || count2 != count || q1 != q2) {
    std::printf("fused path differs from staged path (norm %g/%g, outliers %llu/%llu)\n", (double)norm,
                (double)norm2, (unsigned long long)count, (unsigned long long)count2);
    return 1;
  }

  // --- decompress: Compressor::Decompress lines 256-257
  compressor.DequantizeRecompose(d_data, error_bound_type::REL, (T)tol, s, norm, count);
  std::vector<T> back(n);
  (void)hipMemcpy(back.data(), d_data, n * sizeof(T), hipMemcpyDeviceToHost);

  // --- the 16-bit symbol extension: same values narrowed, bit-identical reconstruction
  if (compressor.SupportsSym16()) {
    uint16_t *d_sym = nullptr;
    if (hipMalloc(&d_sym, n * sizeof(uint16_t)) != hipSuccess) return 1;
    T norm3 = 0;
    compressor.DecomposeQuantizeSym16(d_orig, error_bound_type::REL, (T)tol, s, norm3, d_sym);
    ATOMIC_IDX count3 = 0;
    (void)hipMemcpy(&count3, compressor.outlier_count_device(), sizeof(count3), hipMemcpyDeviceToHost);
    std::vector<uint16_t> sym(n);
    (void)hipMemcpy(sym.data(), d_sym, n * sizeof(uint16_t), hipMemcpyDeviceToHost);
    bool same = norm3 == norm && count3 == count;
    for (SIZE i = 0; same && i < n; i++) same = (QUANTIZED_INT)sym[i] == q2[i];
    compressor.DequantizeRecomposeSym16(d_data, error_bound_type::REL, (T)tol, s, norm, d_sym, count3);
    std::vector<T> back16(n);
    (void)hipMemcpy(back16.data(), d_data, n * sizeof(T), hipMemcpyDeviceToHost);
    (void)hipFree(d_sym);
    if (!same || std::memcmp(back16.data(), back.data(), n * sizeof(T)) != 0) {
      std::printf("16-bit symbol path differs from the int64 path\n");
      return 1;
    }
  }
  double err = 0;
  for (SIZE i = 0; i < n; i++) err = std::fmax(err, std::fabs((double)back[i] - (double)u[i]));
  std::printf("shape %llux%llux%llu %s: l_target %llu, norm %g, outliers %llu, L-inf error %.3e <= %.3e\n",
              (unsigned long long)shape[0], (unsigned long long)shape[1], (unsigned long long)shape[2],
              sizeof(T) == 4 ? "f32" : "f64", (unsigned long long)hierarchy.l_target(), (double)norm,
              (unsigned long long)count, err, tol * norm);
  dfree(d_data);
  dfree(d_orig);
  return err <= tol * norm ? 0 : 1;
}

int main() {
  int rc = 0;
  try {
    rc |= run<float>({65, 70, 129}, 1e-3, std::numeric_limits<float>::infinity());
    rc |= run<double>({34, 33, 65}, 1e-4, std::numeric_limits<double>::infinity());
    // invalid shape: every dimension needs >= 3 nodes (Hierarchy.hpp:742-756)
    bool threw = false;
    try {
      mgard_hip::Hierarchy<3, float> bad({2, 5, 5});
    } catch (const std::runtime_error &) {
      threw = true;
    }
    if (!threw) { std::printf("invalid shape accepted\n"); rc = 1; }
  } catch (const std::exception &e) {
    std::printf("exception: %s\n", e.what());
    return 2;
  }
  std::printf(rc == 0 ? "OK\n" : "FAILED\n");
  return rc;
}
