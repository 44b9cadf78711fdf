/*
 * mgard_hip.h -- C ABI of the MI355X-native MGARD-X hot path
 * (multilevel decomposition chain + level-wise linear quantizer).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Every entry point names the reference (CODARcode/MGARD v1.6.0) interface it
 * replaces; paths are relative to the reference checkout.
 *
 * Conventions
 *  - All data pointers are DEVICE pointers on the hierarchy's device unless the
 *    parameter name starts with h_ (host).
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream). Calls
 *    are asynchronous with respect to the host unless stated otherwise.
 *  - Arrays are dense row-major with the LAST dimension fastest, shape given at
 *    hierarchy creation (mgard_x convention, shape[D-1] = fastest).
 *  - Return value: MGH_SUCCESS (0) or a negative mgh_status. No exceptions
 *    cross this boundary. mgh_last_error() gives a human-readable message.
 */
#ifndef MGARD_HIP_H
#define MGARD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum mgh_status {
  MGH_SUCCESS = 0,
  MGH_ERR_INVALID_ARGUMENT = -1,
  MGH_ERR_UNSUPPORTED_DIMENSION = -2, /* cf. NotSupportHigherNumberOfDimensionsFailure */
  MGH_ERR_UNSUPPORTED_DTYPE = -3,     /* cf. NotSupportDataTypeFailure */
  MGH_ERR_DEVICE = -4,                /* HIP runtime error (message in mgh_last_error) */
  MGH_ERR_OUT_OF_MEMORY = -5,
  MGH_ERR_NO_DEVICE = -6 /* cf. BackendNotAvailableFailure */
} mgh_status;

/* mgard_x::data_type (include/mgard-x/Utilities/Types.h:41) */
typedef enum mgh_dtype { MGH_FLOAT = 0, MGH_DOUBLE = 1 } mgh_dtype;
/* mgard_x::error_bound_type (include/mgard-x/Utilities/Types.h:32) */
typedef enum mgh_error_bound { MGH_REL = 0, MGH_ABS = 1 } mgh_error_bound;

#define MGH_MAX_DIM 5

/* Opaque handle: level shapes + per-level spacing tables on host and device,
 * plus the device workspace the kernels need. Replaces
 * mgard_x::Hierarchy<D,T,HIP> (include/mgard-x/Hierarchy/Hierarchy.hpp:193-418,
 * ctor :712-757) together with the workspace half of
 * mgard_x::DataRefactor<D,T,HIP> (include/mgard-x/DataRefactoring/DataRefactor.hpp:19-71). */
typedef struct mgh_hierarchy mgh_hierarchy;

const char *mgh_last_error(void);
/* Number of visible HIP devices (0 if none / no driver). */
int mgh_device_count(void);

/* h_coords: NULL for a uniform grid (Hierarchy(shape, config), Hierarchy.hpp:712),
 * else D host arrays of `dtype` with shape[d] strictly increasing coordinates
 * (Hierarchy(shape, coords, config), Hierarchy.hpp:741). normalize_coordinates
 * mirrors Config::normalize_coordinates (Config/Config.h:22), max_level mirrors
 * Config::max_larget_level (pass UINT64_MAX for "no limit"). */
int mgh_hierarchy_create(mgh_hierarchy **out, int D, const uint64_t *shape, int dtype,
                         const void *const *h_coords, int normalize_coordinates,
                         uint64_t max_level, int device);
void mgh_hierarchy_destroy(mgh_hierarchy *h);

int mgh_l_target(const mgh_hierarchy *h);                       /* Hierarchy::l_target() */
int mgh_level_shape(const mgh_hierarchy *h, int level, uint64_t *out_shape); /* ::level_shape(l) */
uint64_t mgh_total_num_elems(const mgh_hierarchy *h);           /* ::total_num_elems() */
/* Device bytes held by the handle (tables + workspace). */
size_t mgh_device_bytes(const mgh_hierarchy *h);
/* Device address of the norm (one value of the hierarchy's type) the last fused call with a REL
 * bound and no host read-back (h_norm_out == NULL) computed and used; valid until the next call on
 * the handle. Lets a caller fetch the norm with its own asynchronous copy behind later work instead
 * of paying a synchronisation inside mgh_decompose_quantize*. NULL for a NULL handle. */
const void *mgh_norm_device_ptr(const mgh_hierarchy *h);

/* Host copies of the per-level tables, for inspection/tests. kind: 0=dist
 * (Hierarchy::dist), 1=ratio (::ratio), 2=am, 3=bm (n+1 entries, ::am/::bm),
 * 4=level_marks (int32, only level == l_target). Returns the number of entries
 * written to h_out (capacity `cap` entries), or a negative status. */
int64_t mgh_hierarchy_table(const mgh_hierarchy *h, int kind, int level, int dim, void *h_out,
                            uint64_t cap);

/* Norm used for REL error bounds. Replaces Compressor::CalculateNorm ->
 * norm_calculator (include/mgard-x/CompressionLowLevel/NormCalculator.hpp:12-80).
 * s = +inf: max|x|; otherwise sqrt(sum x^2 / N) (normalize_coordinates) or
 * sqrt(sum x^2); 0 -> epsilon. SYNCHRONOUS (returns the value to the host). */
int mgh_norm(mgh_hierarchy *h, const void *d_data, double s, double *h_norm_out, void *stream);

/* Multilevel decomposition. Replaces Compressor::Decompose ->
 * DataRefactor::Decompose -> multi_dimension::decompose
 * (include/mgard-x/DataRefactoring/MultiDimension/DataRefactoring.hpp:25-177).
 * d_coeff receives the coefficients in MGARD-X's in-place reordered layout
 * (coarse corner first, level by level). d_coeff may equal d_data (the
 * reference's in-place behaviour; costs one extra copy here). */
int mgh_decompose(mgh_hierarchy *h, const void *d_data, void *d_coeff, void *stream);

/* Inverse. Replaces Compressor::Recompose -> multi_dimension::recompose
 * (DataRefactoring.hpp:179-317). d_data may equal d_coeff. */
int mgh_recompose(mgh_hierarchy *h, const void *d_coeff, void *d_data, void *stream);

/* Level-wise linear quantizer. Replaces Compressor::Quantize ->
 * LinearQuantizer::Quantize (include/mgard-x/Quantization/LinearQuantization.hpp
 * :564-683; quantizers :495-545; kernel :21-301). Output keeps the reordered N-D
 * layout (config.reorder == 0). With prep_huffman != 0 values are shifted by
 * dict_size/2 and values outside [0, dict_size) go to the outlier list
 * (linear index, shifted value) while 0 is stored in d_quantized (:208-241).
 * d_outlier_count is a single uint64 on the device, zeroed by this call; if it
 * ends up larger than outlier_capacity only the first outlier_capacity entries
 * were stored and the caller must retry with bigger buffers (:621-676).
 * Outlier order is unspecified (atomic order), as in the reference. */
int mgh_quantize(mgh_hierarchy *h, const void *d_coeff, int error_bound_type, double tol,
                 double s, double norm, uint64_t dict_size, int prep_huffman,
                 int64_t *d_quantized, uint64_t *d_outlier_count, uint64_t *d_outlier_idx,
                 int64_t *d_outlier_val, uint64_t outlier_capacity, void *stream);

/* Replaces Compressor::Dequantize -> LinearQuantizer::Dequantize
 * (LinearQuantization.hpp:685-780, OutlierRestore :304-350). d_quantized is
 * modified (outliers scattered back), like the reference. */
int mgh_dequantize(mgh_hierarchy *h, int64_t *d_quantized, int error_bound_type, double tol,
                   double s, double norm, uint64_t dict_size, int prep_huffman,
                   const uint64_t *d_outlier_idx, const int64_t *d_outlier_val,
                   uint64_t outlier_count, void *d_coeff, void *stream);

/* The fused hot path: [norm] + decompose + quantize in one call, what
 * Compressor::Compress runs before the lossless stage
 * (include/mgard-x/CompressionLowLevel/Compressor.hpp:193-237, lines 216-218).
 * Coefficients are quantized as they are produced; the float coefficient array
 * is not materialised unless d_coeff_opt != NULL. For REL bounds pass
 * norm <= 0 to have the norm computed here (returned through h_norm_out if not
 * NULL; this makes the call synchronise once), or a positive norm to use it.
 * Results are identical to mgh_decompose followed by mgh_quantize. */
int mgh_decompose_quantize(mgh_hierarchy *h, const void *d_data, int error_bound_type,
                           double tol, double s, double norm, double *h_norm_out,
                           uint64_t dict_size, int prep_huffman, int64_t *d_quantized,
                           uint64_t *d_outlier_count, uint64_t *d_outlier_idx,
                           int64_t *d_outlier_val, uint64_t outlier_capacity,
                           void *d_coeff_opt, void *stream);

/* The same with the quantized values delivered as 16-bit DICTIONARY SYMBOLS (what the lossless
 * stage consumes: q + dict_size/2 in [0, dict_size), out-of-dictionary values as symbol 0 plus an
 * entry of the outlier list, LinearQuantization.hpp:208-241) -- the values of
 * mgh_decompose_quantize(..., prep_huffman = 1, ...) narrowed to uint16_t, a quarter of the
 * output bytes. An extension for callers that feed the Huffman stage directly (mgh_compress does);
 * the reference always materialises the int64 array (QUANTIZED_INT, RuntimeX/DataTypes.h:128).
 * dict_size <= 65536. Only where the fused kernels run (3-D, at least one level), else
 * MGH_ERR_UNSUPPORTED_DIMENSION: use mgh_decompose_quantize there. */
int mgh_decompose_quantize_sym16(mgh_hierarchy *h, const void *d_data, int error_bound_type,
                                 double tol, double s, double norm, double *h_norm_out,
                                 uint64_t dict_size, uint16_t *d_symbols,
                                 uint64_t *d_outlier_count, uint64_t *d_outlier_idx,
                                 int64_t *d_outlier_val, uint64_t outlier_capacity, void *stream);

/* Inverse of mgh_decompose_quantize_sym16: dequantize + recompose from 16-bit symbols and the
 * outlier list (found by index through a hash table: symbol 0 at an outlier's position).
 * Same result as mgh_dequantize_recompose on the widened values. mgh_sym16_supported tells
 * whether this hierarchy runs BOTH *_sym16 calls (1: the fused 3-D path) or not (0); the
 * compression-side call alone also runs on the fused 4-D path and returns
 * MGH_ERR_UNSUPPORTED_DIMENSION elsewhere. */
int mgh_dequantize_recompose_sym16(mgh_hierarchy *h, const uint16_t *d_symbols, int error_bound_type,
                                   double tol, double s, double norm, uint64_t dict_size,
                                   const uint64_t *d_outlier_idx, const int64_t *d_outlier_val,
                                   uint64_t outlier_count, void *d_data_out, void *stream);
int mgh_sym16_supported(const mgh_hierarchy *h);

/* Leading dimensions of the caller's arrays of the hierarchy's data type (mgard_x::Array::ld,
 * RuntimeX/DataStructures/Array.hpp:70-84: the reference's HIP backend allocates its arrays
 * with hipMallocPitch by default, which pads the FASTEST dimension; SubArray.hpp:136-139
 * carries one ld per dimension). `ld`: D entries, ld[d] >= shape[d] for d >= 1 (ld[0] is not
 * used); element (i0, .., i_{D-1}) lies at ((i0 * ld[1] + i1) * ld[2] + ...) + i_{D-1}.
 * NULL = dense again. which = MGH_LD_IN: every T array an entry point READS (data of
 * mgh_norm*, mgh_decompose, mgh_decompose_quantize*; coefficients of mgh_recompose,
 * mgh_quantize); MGH_LD_OUT: every T array it WRITES (coefficients of mgh_decompose,
 * mgh_dequantize, d_coeff_opt; data of mgh_recompose, mgh_dequantize_recompose*). The
 * quantized integers, symbols and outlier indices are always dense (Compressor.hpp:48-53
 * allocates them unpitched: they are linearised for the lossless stage). The setting stays
 * until it is changed. The fused 3-D kernels read / write the pitched array in place; the
 * other paths (D != 3, thin shapes, stand-alone stages) go through a dense copy inside the
 * hierarchy. mgh_norm_stream_add takes parts of a dense array only. */
#define MGH_LD_IN 0
#define MGH_LD_OUT 1
int mgh_set_ld(mgh_hierarchy *h, int which, const uint64_t *ld);

/* Norm of an input that is still ARRIVING (host -> device in slabs): _begin once, _add
 * for every part that has landed (any partition of the array; `cold` != 0: the part is
 * read with nontemporal loads, for parts the level pass will not find in the cache
 * anyway), then ONE call of mgh_decompose_quantize / mgh_decompose_quantize_sym16 with
 * a REL bound and norm = 0 on the same hierarchy, which takes the accumulated value
 * instead of reducing the array again. max|x| is exact in any order; the L2 sum has
 * the order dependence of its last bits that the one-pass reduction has too. All calls
 * ASYNCHRONOUS, in stream order. Fused 3-D / 4-D path only
 * (MGH_ERR_UNSUPPORTED_DIMENSION elsewhere). No reference counterpart: the reference
 * computes the norm after the whole subdomain has arrived (Compressor.hpp:158-176). */
int mgh_norm_stream_begin(mgh_hierarchy *h, void *stream);
int mgh_norm_stream_add(mgh_hierarchy *h, const void *d_part, uint64_t count, double s, int cold,
                        void *stream);

/* Norm that stays on the device: writes one value of the hierarchy's dtype to
 * d_norm_out (max|x| for s = +inf, else the L2 norm of this array as
 * norm_calculator defines it). ASYNCHRONOUS. For a decomposed domain the caller
 * reduces the per-subdomain values itself (MAX for s = +inf; RCCL all-reduce). */
int mgh_norm_device(mgh_hierarchy *h, const void *d_data, double s, void *d_norm_out,
                    void *stream);

/* Fused hot path for one subdomain of a decomposed domain, fully asynchronous:
 * d_norm = GLOBAL norm (dtype of the hierarchy, on the device). The bound applied
 * is the reference's per-subdomain bound, calc_local_abs_tol
 * (include/mgard-x/CompressionHighLevel/ErrorToleranceCalculator.hpp:134-155,
 * applied at CompressionHighLevel.hpp:122-144): REL, s=inf: (T)(tol*norm) as an
 * ABS bound; REL, finite s: sqrt((tol*norm)^2 / num_subdomains); ABS: tol resp.
 * sqrt(tol^2 / num_subdomains). 3-D only. */
int mgh_decompose_quantize_dn(mgh_hierarchy *h, const void *d_data, int error_bound_type,
                              double tol, double s, const void *d_norm, uint64_t num_subdomains,
                              uint64_t dict_size, int prep_huffman, int64_t *d_quantized,
                              uint64_t *d_outlier_count, uint64_t *d_outlier_idx,
                              int64_t *d_outlier_val, uint64_t outlier_capacity, void *stream);

/* Inverse of the above: dequantize + recompose
 * (Compressor::Decompress, Compressor.hpp:239-272, lines 256-257). */
int mgh_dequantize_recompose(mgh_hierarchy *h, int64_t *d_quantized, int error_bound_type,
                             double tol, double s, double norm, uint64_t dict_size,
                             int prep_huffman, const uint64_t *d_outlier_idx,
                             const int64_t *d_outlier_val, uint64_t outlier_count, void *d_data,
                             void *stream);

/* OutlierRestore (LinearQuantization.hpp:304-350) on its own: d_q[idx[i]] = val[i]; indices
 * outside [0, n) are ignored. */
int mgh_outlier_restore(int64_t *d_q, uint64_t n, const uint64_t *d_outlier_idx,
                        const int64_t *d_outlier_val, uint64_t outlier_count, void *stream);

/* config.reorder == 1 of the reference ("level linearised" quantized output,
 * Quantization/LinearQuantization.hpp:46-146 calc_level_offset, :588-605 slot of a level): the
 * quantized array with the entries of level 0 first, then the coefficients of level 1 in the
 * natural row-major order of the level-1 grid, and so on. A permutation of the reordered N-D
 * array mgh_quantize / mgh_decompose_quantize write; inverse != 0 undoes it. d_in != d_out.
 * d_outlier_idx (optional, forward only): outlier indices are rewritten in place to positions in
 * the linearised array, as the reference records them in this mode (:226-232); their number is
 * outlier_count, or *d_outlier_count (device) capped at outlier_capacity when that is given. */
int mgh_level_linearize(mgh_hierarchy *h, const int64_t *d_in, int64_t *d_out, int inverse,
                        uint64_t *d_outlier_idx, const uint64_t *d_outlier_count, uint64_t outlier_count,
                        uint64_t outlier_capacity, void *stream);

/* Per-kernel timing hook used by bench.py for the roofline line: when enabled,
 * every kernel launched through this handle is bracketed by HIP events on the
 * launch stream; mgh_profile_read() synchronises and returns accumulated
 * milliseconds and launch counts per kernel name since the last reset.
 * (Reference counterpart: DeviceRuntime::TimingAllKernels,
 * src/mgard-x/RuntimeX/DeviceAdapters/DeviceAdapterSerial.cpp:18-19.) */
int mgh_profile_enable(mgh_hierarchy *h, int enable);
/* Restrict the event bracketing to launches of one kernel name (NULL = all), so
 * that a timed region can carry events on its dominant kernel only. */
int mgh_profile_filter(mgh_hierarchy *h, const char *kernel_name_or_null);
/* Writes up to cap entries; returns number of distinct kernels. names[i] points
 * to a static string. */
int mgh_profile_read(mgh_hierarchy *h, const char **names, double *total_ms, uint64_t *launches,
                     int cap, int reset);

/* Measurement aid for the roofline line (no reference counterpart): a PURE stream with the
 * read/write mix of the top-level pass of the hot path -- n elements of `dtype` read once,
 * n int64 written once with streaming stores, two side arrays of n/8 elements written -- and
 * nothing else, timed with HIP events on `stream` over `reps` launches after 2 warm-up
 * launches. *ms_out = average launch time. It tells how fast this device moves the pass's
 * bytes when no arithmetic, halo or tiling is involved (about 4.0 ... 4.9 TB/s, not the 8 TB/s
 * pin rate). d_in: n elements; d_out: n int64; d_side: n/4 elements of `dtype`. */
int mgh_stream_calibrate(int dtype, const void *d_in, int64_t *d_out, void *d_side, uint64_t n,
                         int reps, double *ms_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MGARD_HIP_H */
