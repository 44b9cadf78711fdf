// compress_x_hip.hpp -- the reference's high-level API under its OWN names: everything a caller
// of `#include "mgard/compress_x.hpp"` (reference include/compress_x.hpp:31-178,
// include/mgard-x/Config/Config.h:10-42, include/mgard-x/Utilities/Types.h:19-65,
// include/mgard-x/RuntimeX/DataTypes.h:106-134) uses lives here in `namespace mgard_x` with the
// same spelling, argument order, defaults, ownership rules and status codes, implemented on the
// C ABI of mgard_hip_compress.h. A program written against the reference switches by changing that
// one include line and linking libmgard_hip.so (tests/cpp/highlevel_api_example.cpp is a consumer
// written against this header with the call shapes of the reference's
// examples/mgard-x/HighLevelAPIs/Example.cpp -- same calls, its own data, sizes and checks).
//
// What differs, because this library is one backend and one path:
//   * dev_type AUTO, HIP and CUDA select the HIP device `dev_id`; SERIAL / OPENMP / SYCL return
//     BackendNotAvailableFailure (there is no CPU implementation in the product);
//   * decomposition must be MultiDim, compressor MGARD, lossless Huffman or Huffman_Zstd,
//     reorder 0 or 1 -- anything else returns Failure instead of silently doing something different;
//   * the fields that only steer the reference's runtime (log_level, prefetch, lz4_block_size,
//     total_num_bitplanes, mdr_*, compress_with_dryrun,
//     num_local_refactoring_level, auto_cache_release, cpu_mode) are accepted and ignored;
//   * adjust_shape is honoured (the array is viewed with the balanced shape of
//     ShapeAdjustment.hpp:43-77 before it is compressed; uniform grids);
//   * auto_pin_host_buffers defaults to FALSE (reference Config.cpp:33: true): registering and
//     unregistering the caller's buffer around every call tripped an intermittent GPU memory
//     fault in the ROCm 7.0 runtime; pageable buffers go through pinned
//     bounce buffers instead, and pin_memory() below pins a buffer once for callers who want
//     direct DMA.
// Do not include this header together with mgard_hip.hpp / compress_hip.hpp in one translation
// unit that also says `using namespace` for both namespaces: the enum names are the same.
#ifndef COMPRESS_X_HIP_HPP
#define COMPRESS_X_HIP_HPP

#include <cstddef>
#include <cstdint>
#include <limits>
#include <vector>

#include "mgard_hip_compress.h"

namespace mgard_x {

using SIZE = uint64_t;            // RuntimeX/DataTypes.h:124
using DIM = uint8_t;              // :134
using Byte = unsigned char;       // :126
using QUANTIZED_INT = int64_t;    // :128

enum class device_type : uint8_t { AUTO, SERIAL, OPENMP, CUDA, HIP, SYCL, NONE };  // DataTypes.h:106-114
enum class cpu_parallelization_mode : uint8_t { INTRA_BLOCK, INTER_BLOCK };
enum class decomposition_type : uint8_t { MultiDim, SingleDim, Hybrid };           // Types.h:19
enum class error_bound_type : uint8_t { REL, ABS };
enum class lossless_type : uint8_t { Huffman, Huffman_LZ4, Huffman_Zstd, CPU_Lossless };
enum class data_type : uint8_t { Float, Double };
enum class domain_decomposition_type : uint8_t { MaxDim, Block, Variable };
enum class compress_status_type : uint8_t {
  Success,
  Failure,
  OutputTooLargeFailure,
  NotSupportHigherNumberOfDimensionsFailure,
  NotSupportDataTypeFailure,
  BackendNotAvailableFailure
};
enum class compressor_type : uint8_t { MGARD, ZFP };

namespace log {  // RuntimeX/Utilities/Log.h: bit flags of Config::log_level
constexpr int ERR = 1, WARN = 2, INFO = 4, DBG = 8, TIME = 16;
}

// Config/Config.h:10-42, defaults of src/mgard-x/Config/Config.cpp:14-43
struct Config {
  device_type dev_type = device_type::AUTO;
  int dev_id = 0;
  compressor_type compressor = compressor_type::MGARD;
  domain_decomposition_type domain_decomposition = domain_decomposition_type::MaxDim;
  decomposition_type decomposition = decomposition_type::MultiDim;
  double estimate_outlier_ratio = 1.0;
  SIZE huff_dict_size = 8192;
  SIZE huff_block_size = 1024 * 20;
  SIZE lz4_block_size = 1 << 15;
  int zstd_compress_level = 3;
  bool normalize_coordinates = true;
  lossless_type lossless = lossless_type::Huffman;
  int reorder = 0;
  int log_level = log::ERR;
  bool prefetch = false;
  bool auto_pin_host_buffers = false;  // (reference: true -- see the header comment)
  // (not a field of the reference's Config) decompress a non-uniform grid with the coordinates
  // rounded through (float) first, as stock MGARD-X does (CompressionHighLevel.hpp:455-462)
  bool mirror_reference_coord_cast = false;
  SIZE max_larget_level = std::numeric_limits<SIZE>::max();
  SIZE max_memory_footprint = std::numeric_limits<SIZE>::max();
  SIZE total_num_bitplanes = 32;
  SIZE block_size = 256;
  SIZE domain_decomposition_dim = 0;
  std::vector<SIZE> domain_decomposition_sizes;
  bool mdr_adaptive_resolution = false;
  bool adjust_shape = false;
  bool compress_with_dryrun = false;
  int num_local_refactoring_level = 1;
  bool auto_cache_release = false;
  cpu_parallelization_mode cpu_mode = cpu_parallelization_mode::INTER_BLOCK;
  void apply() {}
};

namespace detail {
inline compress_status_type status(int rc) {
  switch (rc) {
  case MGH_SUCCESS: return compress_status_type::Success;
  case MGH_ERR_OUTPUT_TOO_LARGE: return compress_status_type::OutputTooLargeFailure;
  case MGH_ERR_UNSUPPORTED_DIMENSION: return compress_status_type::NotSupportHigherNumberOfDimensionsFailure;
  case MGH_ERR_UNSUPPORTED_DTYPE: return compress_status_type::NotSupportDataTypeFailure;
  case MGH_ERR_NO_DEVICE: return compress_status_type::BackendNotAvailableFailure;
  default: return compress_status_type::Failure;
  }
}
// Success if this library can honour the configuration
inline compress_status_type check(const Config &c) {
  if (c.dev_type != device_type::AUTO && c.dev_type != device_type::HIP && c.dev_type != device_type::CUDA)
    return compress_status_type::BackendNotAvailableFailure;
  if (c.decomposition != decomposition_type::MultiDim || c.compressor != compressor_type::MGARD ||
      (c.reorder != 0 && c.reorder != 1))
    return compress_status_type::Failure;
  if (c.lossless != lossless_type::Huffman && c.lossless != lossless_type::Huffman_Zstd)
    return compress_status_type::Failure;
  return compress_status_type::Success;
}
// Config::adjust_shape (CompressionHighLevel.hpp:62-65, ShapeAdjustment.hpp:43-77): before
// compressing, the array is VIEWED with a more balanced shape of the same size -- the largest
// extent is taken apart into its prime factors, which are multiplied, largest first, onto
// whichever extent is the smallest at that moment; with a Variable decomposition the leading
// (time) dimension is adjusted per subdomain and put back together. The header records the
// adjusted shape, which is what decompress hands back.
inline void adjust_shape(std::vector<SIZE> &shape, const Config &c) {
  if (shape.empty()) return;
  SIZE steps = 1;
  const bool variable = c.domain_decomposition == domain_decomposition_type::Variable &&
                        c.domain_decomposition_dim == 0 && !c.domain_decomposition_sizes.empty() &&
                        c.domain_decomposition_sizes[0] > 0;
  if (variable) {
    steps = shape[0] / c.domain_decomposition_sizes[0];
    shape[0] = c.domain_decomposition_sizes[0];
  }
  size_t big = 0;
  for (size_t d = 1; d < shape.size(); d++)
    if (shape[d] > shape[big]) big = d;  // (the first of equal extents)
  SIZE rest = shape[big];
  std::vector<SIZE> primes;
  for (SIZE z = 2; z * z <= rest;) {
    if (rest % z == 0) {
      primes.push_back(z);
      rest /= z;
    } else {
      z++;
    }
  }
  if (rest > 1) primes.push_back(rest);
  shape[big] = 1;
  for (size_t k = primes.size(); k-- > 0;) {
    size_t small = 0;
    for (size_t d = 1; d < shape.size(); d++)
      if (shape[d] < shape[small]) small = d;  // (the first of equal extents)
    shape[small] *= primes[k];
  }
  if (variable) shape[0] *= steps;
}
inline mgh_config to_c(const Config &c) {
  mgh_config m;
  mgh_config_default(&m);
  m.dev_id = c.dev_id;
  m.domain_decomposition = (int)c.domain_decomposition;
  m.domain_decomposition_dim = (int)c.domain_decomposition_dim;
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
  m.reorder = c.reorder;
  m.mirror_reference_coord_cast = c.mirror_reference_coord_cast ? 1 : 0;
  return m;
}
} // namespace detail

// ---- compress (compress_x.hpp:31-100) -----------------------------------------------------------
inline enum compress_status_type
compress(DIM D, data_type dtype, std::vector<SIZE> shape, double tol, double s,
         enum error_bound_type mode, const void *original_data, void *&compressed_data,
         size_t &compressed_size, std::vector<const Byte *> coords, Config config,
         bool output_pre_allocated) {
  const compress_status_type ok = detail::check(config);
  if (ok != compress_status_type::Success) return ok;
  if (shape.size() != D) return compress_status_type::Failure;
  if (config.adjust_shape) {
    // The re-view only makes sense on a uniform grid: coords[d] are sized by the ORIGINAL extents
    // (the reference re-views anyway and reads past them, CompressionHighLevel.hpp:62-65,85).
    if (!coords.empty()) return compress_status_type::Failure;
    const bool variable = config.domain_decomposition == domain_decomposition_type::Variable;
    if (variable) {
      // (ShapeAdjustment.hpp:47-52 takes sizes[0] as THE subdomain extent: equal time steps only)
      if (config.domain_decomposition_dim != 0 || config.domain_decomposition_sizes.empty() ||
          config.domain_decomposition_sizes[0] == 0)
        return compress_status_type::Failure;
      SIZE sum = 0;
      for (SIZE z : config.domain_decomposition_sizes) {
        if (z != config.domain_decomposition_sizes[0]) return compress_status_type::Failure;
        sum += z;
      }
      if (sum != shape[0]) return compress_status_type::Failure;
    }
    const size_t steps = config.domain_decomposition_sizes.size();
    detail::adjust_shape(shape, config);
    // the subdomain sizes follow the adjusted leading extent, so that they still add up to it
    if (variable) config.domain_decomposition_sizes.assign(steps, shape[0] / (SIZE)steps);
  }
  const mgh_config c = detail::to_c(config);
  std::vector<const void *> cp(coords.begin(), coords.end());
  return detail::status(mgh_compress(D, (int)dtype, shape.data(), tol, s, (int)mode, original_data,
                                     &compressed_data, &compressed_size, cp.empty() ? nullptr : cp.data(),
                                     &c, output_pre_allocated ? 1 : 0));
}
inline enum compress_status_type
compress(DIM D, data_type dtype, std::vector<SIZE> shape, double tol, double s,
         enum error_bound_type mode, const void *original_data, void *&compressed_data,
         size_t &compressed_size, Config config, bool output_pre_allocated) {
  return compress(D, dtype, shape, tol, s, mode, original_data, compressed_data, compressed_size,
                  std::vector<const Byte *>(), config, output_pre_allocated);
}
inline enum compress_status_type
compress(DIM D, data_type dtype, std::vector<SIZE> shape, double tol, double s,
         enum error_bound_type mode, const void *original_data, void *&compressed_data,
         size_t &compressed_size, bool output_pre_allocated) {
  return compress(D, dtype, shape, tol, s, mode, original_data, compressed_data, compressed_size,
                  Config(), output_pre_allocated);
}
inline enum compress_status_type
compress(DIM D, data_type dtype, std::vector<SIZE> shape, double tol, double s,
         enum error_bound_type mode, const void *original_data, void *&compressed_data,
         size_t &compressed_size, std::vector<const Byte *> coords, bool output_pre_allocated) {
  return compress(D, dtype, shape, tol, s, mode, original_data, compressed_data, compressed_size,
                  coords, Config(), output_pre_allocated);
}

// ---- decompress (compress_x.hpp:109-154) --------------------------------------------------------
inline enum compress_status_type decompress(const void *compressed_data, size_t compressed_size,
                                            void *&decompressed_data, Config config,
                                            bool output_pre_allocated) {
  const compress_status_type ok = detail::check(config);
  if (ok != compress_status_type::Success) return ok;
  const mgh_config c = detail::to_c(config);
  return detail::status(mgh_decompress(compressed_data, compressed_size, &decompressed_data, &c,
                                       output_pre_allocated ? 1 : 0));
}
inline enum compress_status_type decompress(const void *compressed_data, size_t compressed_size,
                                            void *&decompressed_data, bool output_pre_allocated) {
  return decompress(compressed_data, compressed_size, decompressed_data, Config(), output_pre_allocated);
}
inline enum compress_status_type
decompress(const void *compressed_data, size_t compressed_size, void *&decompressed_data,
           std::vector<mgard_x::SIZE> &shape, data_type &dtype, Config config,
           bool output_pre_allocated) {
  int D = 0, dt = 0;
  uint64_t shp[MGH_MAX_DIM];
  int rc = mgh_infer_shape(compressed_data, compressed_size, &D, shp);
  if (rc == MGH_SUCCESS) rc = mgh_infer_data_type(compressed_data, compressed_size, &dt);
  if (rc != MGH_SUCCESS) return detail::status(rc);
  shape.assign(shp, shp + D);
  dtype = dt == MGH_DOUBLE ? data_type::Double : data_type::Float;
  return decompress(compressed_data, compressed_size, decompressed_data, config, output_pre_allocated);
}
inline enum compress_status_type
decompress(const void *compressed_data, size_t compressed_size, void *&decompressed_data,
           std::vector<mgard_x::SIZE> &shape, data_type &dtype, bool output_pre_allocated) {
  return decompress(compressed_data, compressed_size, decompressed_data, shape, dtype, Config(),
                    output_pre_allocated);
}

// ---- cache and pinned memory (compress_x.hpp:159-178) -------------------------------------------
inline enum compress_status_type release_cache(Config config) {
  (void)config;
  mgh_release_cache();
  return compress_status_type::Success;
}
inline void pin_memory(void *ptr, SIZE num_bytes, Config config) {
  (void)config;
  (void)mgh_pin_memory(ptr, (size_t)num_bytes);
}
inline bool check_memory_pinned(void *ptr, Config config) {
  (void)config;
  return mgh_check_memory_pinned(ptr) != 0;
}
inline void unpin_memory(void *ptr, Config config) {
  (void)config;
  (void)mgh_unpin_memory(ptr);
}

} // namespace mgard_x

#endif // COMPRESS_X_HIP_HPP
