// A consumer of the high-level API written the way one writes against mgard_x::compress /
// decompress (reference examples/mgard-x/HighLevelAPIs), with the namespace changed to
// mgard_hip: host buffers, library-allocated outputs, shape/type recovered from the stream.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

#include <hip/hip_runtime.h>

#include "compress_hip.hpp"

static void *dalloc(size_t n) {
  void *p = nullptr;
  return hipMalloc(&p, n) == hipSuccess ? p : nullptr;
}
static void dfree(void *p) { (void)hipFree(p); }

// The low-level stage chain of Compressor::Compress / Decompress (Compressor.hpp:193-272)
// including the lossless stages, on one device-resident array.
static int lowlevel_with_lossless() {
  using namespace mgard_hip;
  std::vector<SIZE> shape{33, 40, 65};
  const SIZE n = 33 * 40 * 65;
  std::vector<float> u(n);
  for (SIZE i = 0; i < 33; i++)
    for (SIZE j = 0; j < 40; j++)
      for (SIZE k = 0; k < 65; k++)
        u[(i * 40 + j) * 65 + k] = std::sin(0.1f * i) * std::cos(0.09f * j) + 0.3f * std::sin(0.06f * k);
  HighLevelConfig config;
  Hierarchy<3, float> hierarchy(shape, config);
  Compressor<3, float> compressor(hierarchy, config, DeviceAllocator{dalloc, dfree});
  LosslessCompressor<3, float> lossless(compressor, config);
  float *d = (float *)dalloc(n * sizeof(float));
  (void)hipMemcpy(d, u.data(), n * sizeof(float), hipMemcpyHostToDevice);
  const float tol = 1e-3f, s = std::numeric_limits<float>::infinity();
  float norm = 0;
  compressor.CalculateNorm(d, error_bound_type::REL, s, norm);
  compressor.Decompose(d);
  compressor.Quantize(d, error_bound_type::REL, tol, s, norm);
  ATOMIC_IDX count = 0;
  (void)hipMemcpy(&count, compressor.outlier_count_device(), sizeof(count), hipMemcpyDeviceToHost);
  const Byte *rec = nullptr;
  SIZE rec_size = 0;
  lossless.LosslessCompress(n, count, rec, rec_size);
  std::vector<Byte> stored(rec, rec + rec_size);  // what a caller would write out
  // ... and back
  (void)hipMemset(compressor.quantized_array(), 0xff, n * sizeof(QUANTIZED_INT));
  const ATOMIC_IDX *oi = nullptr;
  const QUANTIZED_INT *ov = nullptr;
  SIZE ocount = 0;
  lossless.LosslessDecompress(stored.data(), stored.size(), n, oi, ov, ocount);
  if (ocount != count) return 1;
  // the decoded outlier list lives in the lossless context: dequantize + recompose with it
  int rc = mgh_dequantize_recompose(hierarchy.handle(), compressor.quantized_array(), (int)error_bound_type::REL,
                                    tol, s, norm, config.huff_dict_size, 1, oi, ov, ocount, d, nullptr);
  if (rc != MGH_SUCCESS) return 1;
  std::vector<float> back(n);
  (void)hipMemcpy(back.data(), d, n * sizeof(float), hipMemcpyDeviceToHost);
  double err = 0;
  for (SIZE i = 0; i < n; i++) err = std::fmax(err, std::fabs((double)back[i] - (double)u[i]));
  std::printf("low-level + lossless: record %llu bytes for %llu, error %.3e <= %.3e\n",
              (unsigned long long)rec_size, (unsigned long long)(n * 4), err, (double)tol * norm);
  dfree(d);
  return err <= (double)tol * norm ? 0 : 1;
}

int main() {
  if (lowlevel_with_lossless() != 0) {
    std::printf("low-level + lossless failed: %s\n", mgh_last_error());
    return 1;
  }
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
