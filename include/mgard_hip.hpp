// mgard_hip.hpp -- C++ host-side mirror of the reference's LOW-LEVEL MGARD-X interface for
// the hot path, on top of the C ABI (mgard_hip.h). Header only.
//
// It keeps the reference's names, argument meaning and error behaviour so that code written
// against mgard_x::Hierarchy / mgard_x::Compressor
// (include/mgard-x/Hierarchy/Hierarchy.h:17-100,
//  include/mgard-x/CompressionLowLevel/Compressor.h:28-89) reads the same:
//
//   mgard_hip::Hierarchy<D, T> hierarchy(shape, config);          // or (shape, coords, config)
//   mgard_hip::Compressor<D, T> compressor(hierarchy, config);
//   compressor.CalculateNorm(d_data, ebtype, s, norm, stream);
//   compressor.Decompose(d_data, stream);                          // in place
//   compressor.Quantize(d_data, ebtype, tol, s, norm, stream);     // -> quantized_array()
//   ...LosslessCompress / Serialize stay with MGARD-X (not part of this library)...
//   compressor.Dequantize(d_data, ebtype, tol, s, norm, stream);
//   compressor.Recompose(d_data, stream);
//
// Differences, all deliberate: device buffers are plain T* (dense, last dimension fastest)
// instead of mgard_x::Array; queues are hipStream_t passed as void*; the outlier buffers are
// owned by the Compressor (in the reference they live in the Huffman workspace); fatal errors
// throw std::runtime_error with the C ABI's message instead of calling exit(-1).
#ifndef MGARD_HIP_HPP
#define MGARD_HIP_HPP

#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "mgard_hip.h"

namespace mgard_hip {

using SIZE = uint64_t;           // mgard_x::SIZE  (RuntimeX/DataTypes.h:124)
using DIM = uint8_t;             // mgard_x::DIM
using QUANTIZED_INT = int64_t;   // mgard_x::QUANTIZED_INT (RuntimeX/DataTypes.h:128)
using ATOMIC_IDX = uint64_t;     // mgard_x::ATOMIC_IDX

enum class error_bound_type : uint8_t { REL, ABS };  // Utilities/Types.h:32
enum class data_type : uint8_t { Float, Double };    // Utilities/Types.h:41
enum class compress_status_type : uint8_t {          // Utilities/Types.h:56-63
  Success,
  Failure,
  OutputTooLargeFailure,
  NotSupportHigherNumberOfDimensionsFailure,
  NotSupportDataTypeFailure,
  BackendNotAvailableFailure
};

// The subset of mgard_x::Config (Config/Config.h:10-42, defaults Config.cpp:14-43) that
// this path reads.
struct Config {
  int dev_id = 0;
  SIZE huff_dict_size = 8192;
  bool normalize_coordinates = true;
  bool prep_huffman = true;  // lossless != lossless_type::CPU_Lossless
  SIZE max_larget_level = std::numeric_limits<SIZE>::max();
  double estimate_outlier_ratio = 1.0;
};

inline compress_status_type to_status(int rc) {
  switch (rc) {
  case MGH_SUCCESS: return compress_status_type::Success;
  case MGH_ERR_UNSUPPORTED_DIMENSION:
    return compress_status_type::NotSupportHigherNumberOfDimensionsFailure;
  case MGH_ERR_UNSUPPORTED_DTYPE: return compress_status_type::NotSupportDataTypeFailure;
  case MGH_ERR_NO_DEVICE: return compress_status_type::BackendNotAvailableFailure;
  default: return compress_status_type::Failure;
  }
}

inline void check(int rc, const char *what) {
  if (rc < 0) throw std::runtime_error(std::string(what) + ": " + mgh_last_error());
}

template <DIM D, typename T> class Hierarchy {
  static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value,
                "float or double");

public:
  // Hierarchy(shape, config): uniform grid (Hierarchy.hpp:712)
  Hierarchy(std::vector<SIZE> shape, Config config = Config()) { init(shape, nullptr, config); }
  // Hierarchy(shape, coords, config): host coordinate arrays (Hierarchy.hpp:741)
  Hierarchy(std::vector<SIZE> shape, std::vector<T *> coords, Config config = Config()) {
    std::vector<const void *> c(coords.begin(), coords.end());
    init(shape, c.data(), config);
  }
  ~Hierarchy() { mgh_hierarchy_destroy(h_); }
  Hierarchy(const Hierarchy &) = delete;
  Hierarchy &operator=(const Hierarchy &) = delete;

  SIZE l_target() const { return (SIZE)mgh_l_target(h_); }
  SIZE total_num_elems() const { return mgh_total_num_elems(h_); }
  std::vector<SIZE> level_shape(SIZE level) const {
    std::vector<SIZE> s(D);
    check(mgh_level_shape(h_, (int)level, s.data()), "level_shape");
    return s;
  }
  SIZE level_shape(SIZE level, DIM dim) const { return dim >= D ? 1 : level_shape(level)[dim]; }
  mgh_hierarchy *handle() const { return h_; }
  // Leading dimensions of the arrays the stages read (which = MGH_LD_IN) / write (MGH_LD_OUT):
  // Array::ld(d) of a pitched mgard_x::Array (Array.hpp:70-84); an empty vector = dense.
  void set_ld(int which, const std::vector<SIZE> &ld) {
    if (!ld.empty() && ld.size() != D) throw std::runtime_error("set_ld: one entry per dimension");
    check(mgh_set_ld(h_, which, ld.empty() ? nullptr : ld.data()), "set_ld");
  }

private:
  void init(const std::vector<SIZE> &shape, const void *const *coords, const Config &config) {
    if (shape.size() != D) throw std::runtime_error("Number of dimensions mismatch.");
    check(mgh_hierarchy_create(&h_, D, shape.data(),
                               std::is_same<T, double>::value ? MGH_DOUBLE : MGH_FLOAT, coords,
                               config.normalize_coordinates ? 1 : 0, config.max_larget_level,
                               config.dev_id),
          "mgard_hip::Hierarchy");
  }
  mgh_hierarchy *h_ = nullptr;
};

// Device allocation hooks (the library itself never allocates user-visible buffers):
// supply hipMalloc/hipFree-compatible functions, e.g. from your allocator.
struct DeviceAllocator {
  void *(*alloc)(size_t) = nullptr;
  void (*release)(void *) = nullptr;
};

template <DIM D, typename T> class Compressor {
public:
  Compressor(Hierarchy<D, T> &hierarchy, Config config, DeviceAllocator a)
      : hierarchy_(&hierarchy), config_(config), a_(a) {
    const SIZE n = hierarchy.total_num_elems();
    outlier_cap_ = (SIZE)(config.estimate_outlier_ratio * (double)n);
    if (outlier_cap_ < 1) outlier_cap_ = 1;
    quantized_ = static_cast<QUANTIZED_INT *>(a_.alloc(n * sizeof(QUANTIZED_INT)));
    outlier_count_ = static_cast<ATOMIC_IDX *>(a_.alloc(sizeof(ATOMIC_IDX)));
    outlier_idx_ = static_cast<ATOMIC_IDX *>(a_.alloc(outlier_cap_ * sizeof(ATOMIC_IDX)));
    outliers_ = static_cast<QUANTIZED_INT *>(a_.alloc(outlier_cap_ * sizeof(QUANTIZED_INT)));
    if (!quantized_ || !outlier_count_ || !outlier_idx_ || !outliers_)
      throw std::runtime_error("mgard_hip::Compressor: device allocation failed");
  }
  ~Compressor() {
    a_.release(quantized_);
    a_.release(outlier_count_);
    a_.release(outlier_idx_);
    a_.release(outliers_);
  }
  Compressor(const Compressor &) = delete;
  Compressor &operator=(const Compressor &) = delete;

  // Compressor::CalculateNorm (Compressor.hpp:125-133): REL only; returns through `norm`
  void CalculateNorm(const T *original_data, error_bound_type ebtype, T s, T &norm,
                     void *queue = nullptr) {
    if (ebtype != error_bound_type::REL) return;
    double n = 0;
    check(mgh_norm(hierarchy_->handle(), original_data, (double)s, &n, queue), "CalculateNorm");
    norm = (T)n;
  }
  // Compressor::Decompose (:135-139): in place
  void Decompose(T *original_data, void *queue = nullptr) {
    check(mgh_decompose(hierarchy_->handle(), original_data, original_data, queue), "Decompose");
  }
  // Compressor::Quantize (:141-147): result in quantized_array() + outlier list. Like
  // LinearQuantizer::Quantize it is up to the caller to grow the outlier buffers and retry
  // when outlier_count() > outlier_capacity() (LinearQuantization.hpp:621-676).
  void Quantize(const T *original_data, error_bound_type ebtype, T tol, T s, T norm,
                void *queue = nullptr) {
    check(mgh_quantize(hierarchy_->handle(), original_data, (int)ebtype, (double)tol, (double)s,
                       (double)norm, config_.huff_dict_size, config_.prep_huffman ? 1 : 0,
                       quantized_, outlier_count_, outlier_idx_, outliers_, outlier_cap_, queue),
          "Quantize");
  }
  // Compressor::Dequantize (:170-177); outlier_count as produced by Quantize
  void Dequantize(T *decompressed_data, error_bound_type ebtype, T tol, T s, T norm,
                  SIZE outlier_count, void *queue = nullptr) {
    check(mgh_dequantize(hierarchy_->handle(), quantized_, (int)ebtype, (double)tol, (double)s,
                         (double)norm, config_.huff_dict_size, config_.prep_huffman ? 1 : 0,
                         outlier_idx_, outliers_, outlier_count, decompressed_data, queue),
          "Dequantize");
  }
  // Compressor::Recompose (:163-168): in place
  void Recompose(T *decompressed_data, void *queue = nullptr) {
    check(mgh_recompose(hierarchy_->handle(), decompressed_data, decompressed_data, queue),
          "Recompose");
  }
  // Lines 216-218 of Compressor::Compress in one call (norm + decompose + quantize, the float
  // coefficients are never materialised). norm is returned like CalculateNorm does.
  void DecomposeQuantize(const T *original_data, error_bound_type ebtype, T tol, T s, T &norm,
                         void *queue = nullptr) {
    double n = ebtype == error_bound_type::REL ? 0.0 : 1.0;
    check(mgh_decompose_quantize(hierarchy_->handle(), original_data, (int)ebtype, (double)tol,
                                 (double)s, n, &n, config_.huff_dict_size,
                                 config_.prep_huffman ? 1 : 0, quantized_, outlier_count_,
                                 outlier_idx_, outliers_, outlier_cap_, nullptr, queue),
          "DecomposeQuantize");
    norm = (T)n;
  }
  // Lines 256-257 of Compressor::Decompress in one call
  void DequantizeRecompose(T *decompressed_data, error_bound_type ebtype, T tol, T s, T norm,
                           SIZE outlier_count, void *queue = nullptr) {
    check(mgh_dequantize_recompose(hierarchy_->handle(), quantized_, (int)ebtype, (double)tol,
                                   (double)s, (double)norm, config_.huff_dict_size,
                                   config_.prep_huffman ? 1 : 0, outlier_idx_, outliers_,
                                   outlier_count, decompressed_data, queue),
          "DequantizeRecompose");
  }

  // Extension (mgh_*_sym16): the same two calls with the quantized values as 16-bit dictionary
  // symbols in a caller-provided device buffer of hierarchy.total_num_elems() uint16_t
  // (prep_huffman semantics; only where SupportsSym16(), i.e. on the fused 3-D path).
  bool SupportsSym16() const { return mgh_sym16_supported(hierarchy_->handle()) != 0; }
  void DecomposeQuantizeSym16(const T *original_data, error_bound_type ebtype, T tol, T s, T &norm,
                              uint16_t *symbols, void *queue = nullptr) {
    double n = ebtype == error_bound_type::REL ? 0.0 : 1.0;
    check(mgh_decompose_quantize_sym16(hierarchy_->handle(), original_data, (int)ebtype, (double)tol,
                                       (double)s, n, &n, config_.huff_dict_size, symbols,
                                       outlier_count_, outlier_idx_, outliers_, outlier_cap_, queue),
          "DecomposeQuantizeSym16");
    norm = (T)n;
  }
  void DequantizeRecomposeSym16(T *decompressed_data, error_bound_type ebtype, T tol, T s, T norm,
                                const uint16_t *symbols, SIZE outlier_count, void *queue = nullptr) {
    check(mgh_dequantize_recompose_sym16(hierarchy_->handle(), symbols, (int)ebtype, (double)tol,
                                         (double)s, (double)norm, config_.huff_dict_size,
                                         outlier_idx_, outliers_, outlier_count, decompressed_data,
                                         queue),
          "DequantizeRecomposeSym16");
  }

  QUANTIZED_INT *quantized_array() { return quantized_; }
  ATOMIC_IDX *outlier_count_device() { return outlier_count_; }
  ATOMIC_IDX *outlier_indexes() { return outlier_idx_; }
  QUANTIZED_INT *outliers() { return outliers_; }
  SIZE outlier_capacity() const { return outlier_cap_; }

private:
  Hierarchy<D, T> *hierarchy_;
  Config config_;
  DeviceAllocator a_;
  QUANTIZED_INT *quantized_ = nullptr;
  ATOMIC_IDX *outlier_count_ = nullptr, *outlier_idx_ = nullptr;
  QUANTIZED_INT *outliers_ = nullptr;
  SIZE outlier_cap_ = 0;
};

} // namespace mgard_hip

#endif // MGARD_HIP_HPP
