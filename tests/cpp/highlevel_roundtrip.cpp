// A consumer of the high-level API written the way one writes against mgard_x::compress /
// decompress (reference examples/mgard-x/HighLevelAPIs), with the namespace changed to
// mgard_hip: host buffers, library-allocated outputs, shape/type recovered from the stream.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

#include "compress_hip.hpp"

int main() {
  const mgard_hip::SIZE n1 = 70, n2 = 65, n3 = 129;
  std::vector<mgard_hip::SIZE> shape{n1, n2, n3};
  std::vector<double> in(n1 * n2 * n3);
  for (size_t i = 0; i < n1; i++)
    for (size_t j = 0; j < n2; j++)
      for (size_t k = 0; k < n3; k++)
        in[(i * n2 + j) * n3 + k] = std::sin(0.05 * i) * std::cos(0.07 * j) + 0.3 * std::sin(0.04 * k);
  const double tol = 1e-4, s = std::numeric_limits<double>::infinity();
  mgard_hip::HighLevelConfig config;
  config.lossless = mgard_hip::lossless_type::Huffman_Zstd;
  void *compressed = nullptr;
  size_t compressed_size = 0;
  auto st = mgard_hip::compress(3, mgard_hip::data_type::Double, shape, tol, s,
                                mgard_hip::error_bound_type::ABS, in.data(), compressed, compressed_size,
                                config, false);
  if (st != mgard_hip::compress_status_type::Success) {
    std::printf("compress failed: %s\n", mgh_last_error());
    return 1;
  }
  void *out = nullptr;
  std::vector<mgard_hip::SIZE> shape2;
  mgard_hip::data_type dtype2;
  st = mgard_hip::decompress(compressed, compressed_size, out, shape2, dtype2, config, false);
  if (st != mgard_hip::compress_status_type::Success) {
    std::printf("decompress failed: %s\n", mgh_last_error());
    return 1;
  }
  if (shape2 != shape || dtype2 != mgard_hip::data_type::Double) {
    std::printf("shape / type mismatch\n");
    return 1;
  }
  double err = 0;
  const double *o = static_cast<const double *>(out);
  for (size_t i = 0; i < in.size(); i++) err = std::fmax(err, std::fabs(o[i] - in[i]));
  std::printf("ratio %.1f  max error %.3e (tol %.1e)\n", (double)(in.size() * 8) / compressed_size, err, tol);
  // a pre-allocated output that is too small is reported, not overrun
  std::vector<unsigned char> small(256);
  void *sp = small.data();
  size_t ssz = small.size();
  st = mgard_hip::compress(3, mgard_hip::data_type::Double, shape, tol, s, mgard_hip::error_bound_type::ABS,
                           in.data(), sp, ssz, config, true);
  const bool too_large_ok = st == mgard_hip::compress_status_type::OutputTooLargeFailure;
  std::free(compressed);
  std::free(out);
  mgard_hip::release_cache();
  if (err > tol || !too_large_ok) return 1;
  std::printf("OK\n");
  return 0;
}
