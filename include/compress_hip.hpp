// compress_hip.hpp -- C++ mirror of the reference's HIGH-LEVEL API (include/compress_x.hpp:31-178)
// on top of the C ABI of mgard_hip_compress.h. Header only. Same names, argument order,
// ownership rules and status codes as mgard_x::compress / decompress, so that a caller switches
// by changing the namespace:
//
//   mgard_hip::Config config;                       // defaults of Config.cpp:14-43
//   config.lossless = mgard_hip::lossless_type::Huffman_Zstd;
//   void *compressed = nullptr; size_t compressed_size = 0;
//   mgard_hip::compress(3, mgard_hip::data_type::Float, {512, 512, 512}, 1e-3,
//                       std::numeric_limits<double>::infinity(), mgard_hip::error_bound_type::REL,
//                       data, compressed, compressed_size, config, false);
//   void *out = nullptr;
//   mgard_hip::decompress(compressed, compressed_size, out, config, false);
//
// Buffers may be host or device memory; outputs that are not pre-allocated are malloc'ed (host
// input) or hipMalloc'ed (device input) by the library and belong to the caller
// (CompressionHighLevel.hpp:147-162).
#ifndef COMPRESS_HIP_HPP
#define COMPRESS_HIP_HPP

#include <cstddef>
#include <cstdint>
#include <limits>
#include <vector>

#include "mgard_hip.hpp"
#include "mgard_hip_compress.h"

namespace mgard_hip {

using Byte = unsigned char;                                                 // DataTypes.h:126
enum class domain_decomposition_type : uint8_t { MaxDim, Block, Variable };  // Types.h:50
enum class lossless_type : uint8_t { Huffman, Huffman_LZ4, Huffman_Zstd, CPU_Lossless };

// mgard_x::Config (Config/Config.h:10-42): the fields the high-level path reads
struct HighLevelConfig : Config {
  domain_decomposition_type domain_decomposition = domain_decomposition_type::MaxDim;
  int domain_decomposition_dim = 0;
  std::vector<SIZE> domain_decomposition_sizes;
  SIZE block_size = 256;
  SIZE huff_block_size = 1024 * 20;
  lossless_type lossless = lossless_type::Huffman;
  int zstd_compress_level = 3;
  SIZE max_memory_footprint = std::numeric_limits<SIZE>::max();
  bool auto_pin_host_buffers = false;  // (reference: true; see mgh_config_default in highlevel.hip)
};

namespace detail {
inline mgh_config to_c(const HighLevelConfig &c) {
  mgh_config m;
  mgh_config_default(&m);
  m.dev_id = c.dev_id;
  m.domain_decomposition = (int)c.domain_decomposition;
  m.domain_decomposition_dim = c.domain_decomposition_dim;
  m.domain_decomposition_sizes = c.domain_decomposition_sizes.empty() ? nullptr : c.domain_decomposition_sizes.data();
  m.num_domain_decomposition_sizes = c.domain_decomposition_sizes.size();
  m.block_size = c.block_size;
  m.estimate_outlier_ratio = c.estimate_outlier_ratio;
  m.huff_dict_size = c.huff_dict_size;
  m.huff_block_size = c.huff_block_size;
  m.lossless = (int)c.lossless;
  m.zstd_compress_level = c.zstd_compress_level;
  m.normalize_coordinates = c.normalize_coordinates ? 1 : 0;
  m.max_larget_level = c.max_larget_level;
  m.max_memory_footprint = c.max_memory_footprint;
  m.auto_pin_host_buffers = c.auto_pin_host_buffers ? 1 : 0;
  return m;
}
inline compress_status_type status(int rc) {
  if (rc == MGH_ERR_OUTPUT_TOO_LARGE) return compress_status_type::OutputTooLargeFailure;
  return to_status(rc);
}
} // namespace detail

// compress (compress_x.hpp:56-76): non-uniform grid, explicit config
inline compress_status_type compress(DIM D, data_type dtype, std::vector<SIZE> shape, double tol, double s,
                                     error_bound_type mode, const void *original_data,
                                     void *&compressed_data, size_t &compressed_size,
                                     std::vector<const Byte *> coords, HighLevelConfig config,
                                     bool output_pre_allocated) {
  if (shape.size() != D) return compress_status_type::Failure;
  const mgh_config c = detail::to_c(config);
  std::vector<const void *> cp(coords.begin(), coords.end());
  return detail::status(mgh_compress(D, (int)dtype, shape.data(), tol, s, (int)mode, original_data,
                                     &compressed_data, &compressed_size, cp.empty() ? nullptr : cp.data(),
                                     &c, output_pre_allocated ? 1 : 0));
}
// (compress_x.hpp:44-54)
inline compress_status_type compress(DIM D, data_type dtype, std::vector<SIZE> shape, double tol, double s,
                                     error_bound_type mode, const void *original_data,
                                     void *&compressed_data, size_t &compressed_size,
                                     HighLevelConfig config, bool output_pre_allocated) {
  return compress(D, dtype, shape, tol, s, mode, original_data, compressed_data, compressed_size,
                  std::vector<const Byte *>(), config, output_pre_allocated);
}
// (compress_x.hpp:31-42)
inline compress_status_type compress(DIM D, data_type dtype, std::vector<SIZE> shape, double tol, double s,
                                     error_bound_type mode, const void *original_data,
                                     void *&compressed_data, size_t &compressed_size,
                                     bool output_pre_allocated) {
  return compress(D, dtype, shape, tol, s, mode, original_data, compressed_data, compressed_size,
                  HighLevelConfig(), output_pre_allocated);
}
// (compress_x.hpp:78-100)
inline compress_status_type compress(DIM D, data_type dtype, std::vector<SIZE> shape, double tol, double s,
                                     error_bound_type mode, const void *original_data,
                                     void *&compressed_data, size_t &compressed_size,
                                     std::vector<const Byte *> coords, bool output_pre_allocated) {
  return compress(D, dtype, shape, tol, s, mode, original_data, compressed_data, compressed_size, coords,
                  HighLevelConfig(), output_pre_allocated);
}

// decompress (compress_x.hpp:109-154)
inline compress_status_type decompress(const void *compressed_data, size_t compressed_size,
                                       void *&decompressed_data, HighLevelConfig config,
                                       bool output_pre_allocated) {
  const mgh_config c = detail::to_c(config);
  return detail::status(mgh_decompress(compressed_data, compressed_size, &decompressed_data, &c,
                                       output_pre_allocated ? 1 : 0));
}
inline compress_status_type decompress(const void *compressed_data, size_t compressed_size,
                                       void *&decompressed_data, bool output_pre_allocated) {
  return decompress(compressed_data, compressed_size, decompressed_data, HighLevelConfig(),
                    output_pre_allocated);
}
inline compress_status_type decompress(const void *compressed_data, size_t compressed_size,
                                       void *&decompressed_data, std::vector<SIZE> &shape,
                                       data_type &dtype, HighLevelConfig config, bool output_pre_allocated) {
  int D = 0, dt = 0;
  uint64_t shp[MGH_MAX_DIM];
  int rc = mgh_infer_shape(compressed_data, compressed_size, &D, shp);
  if (rc == MGH_SUCCESS) rc = mgh_infer_data_type(compressed_data, compressed_size, &dt);
  if (rc != MGH_SUCCESS) return detail::status(rc);
  shape.assign(shp, shp + D);
  dtype = dt == MGH_DOUBLE ? data_type::Double : data_type::Float;
  return decompress(compressed_data, compressed_size, decompressed_data, config, output_pre_allocated);
}
inline compress_status_type decompress(const void *compressed_data, size_t compressed_size,
                                       void *&decompressed_data, std::vector<SIZE> &shape,
                                       data_type &dtype, bool output_pre_allocated) {
  return decompress(compressed_data, compressed_size, decompressed_data, shape, dtype, HighLevelConfig(),
                    output_pre_allocated);
}

// The remaining stages of the reference's LOW-LEVEL Compressor (Compressor.h:39-78):
// LosslessCompress + Serialize, Deserialize + LosslessDecompress, and Compress / Decompress that
// chain all stages (Compressor.hpp:193-272). The serialized record is the subdomain payload of
// the container (Huffman.hpp:163-239); it lives in host memory owned by this object.
template <DIM D, typename T> class LosslessCompressor {
public:
  explicit LosslessCompressor(Compressor<D, T> &c, HighLevelConfig config = HighLevelConfig())
      : c_(&c), config_(config) {
    check(mgh_lossless_create(&ctx_, config.dev_id), "LosslessCompressor");
  }
  ~LosslessCompressor() { mgh_lossless_destroy(ctx_); }
  LosslessCompressor(const LosslessCompressor &) = delete;
  LosslessCompressor &operator=(const LosslessCompressor &) = delete;

  // LosslessCompress + Serialize: quantized_array() and the outlier list of the Compressor ->
  // record bytes (valid until the next call)
  void LosslessCompress(SIZE n, SIZE outlier_count, const Byte *&data, SIZE &size, void *queue = nullptr) {
    const uint8_t *p = nullptr;
    uint64_t sz = 0;
    check(mgh_lossless_compress(ctx_, c_->quantized_array(), n, config_.huff_dict_size,
                                config_.huff_block_size, (int)config_.lossless,
                                config_.zstd_compress_level, c_->outlier_indexes(), c_->outliers(),
                                outlier_count, &p, &sz, queue),
          "LosslessCompress");
    data = p;
    size = sz;
  }
  // Deserialize + LosslessDecompress: record bytes -> quantized_array(); the outlier list comes
  // back in device buffers owned by this object
  void LosslessDecompress(const Byte *data, SIZE size, SIZE n, const ATOMIC_IDX *&outlier_idx,
                          const QUANTIZED_INT *&outliers, SIZE &outlier_count, void *queue = nullptr) {
    uint64_t cnt = 0;
    check(mgh_lossless_decompress(ctx_, data, size, (int)config_.lossless, c_->quantized_array(), n,
                                  &outlier_idx, &outliers, &cnt, queue),
          "LosslessDecompress");
    outlier_count = cnt;
  }

private:
  Compressor<D, T> *c_;
  HighLevelConfig config_;
  mgh_lossless_ctx *ctx_ = nullptr;
};

// release_cache (compress_x.hpp:159)
inline compress_status_type release_cache(HighLevelConfig = HighLevelConfig()) {
  mgh_release_cache();
  return compress_status_type::Success;
}

} // namespace mgard_hip

#endif // COMPRESS_HIP_HPP
