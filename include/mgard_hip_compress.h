/* mgard_hip_compress.h -- C ABI of the HIGH-LEVEL path: whole-array compress / decompress with
 * domain decomposition, the lossless stage and the self-describing container, on top of the
 * low-level entry points of mgard_hip.h. Same shared library (libmgard_hip.so).
 *
 * What each entry point replaces in the reference (MGARD-X):
 *   mgh_compress / mgh_decompress   mgard_x::compress / decompress
 *                                   (include/compress_x.hpp:31-100, 109-154;
 *                                    CompressionHighLevel.hpp:47-330, 478-640)
 *   mgh_config                      mgard_x::Config (Config/Config.h:10-42, defaults Config.cpp:14-43)
 *   mgh_metadata_* / mgh_infer_*    Metadata<..>::Serialize / Deserialize, infer_shape,
 *                                   infer_data_type (Metadata/Metadata.hpp:226-262)
 *   mgh_huffman_*                   ComposedLosslessCompressor::Compress/Serialize and
 *                                   Deserialize/Decompress (Lossless/Lossless.hpp:70-118)
 *
 * Output container (byte-compatible with the reference, see mgard_amd/csrc/format.hpp):
 *   "MGARD" | u64 LE header_size | u32 LE crc32 | proto3 Header | per subdomain:
 *   [u64 compressed_size][payload]   (GPUPipelines.hpp:189-193), payload = the Huffman record
 *   of Huffman.hpp:163-239 (optionally [u64 size][zstd frame] around it, Zstd.hpp:69-90), or the
 *   raw subdomain when compression would not shrink it (GPUPipelines.hpp:136-155).
 *
 * Conventions as in mgard_hip.h. `original_data`, `compressed_data`, `decompressed_data` may be
 * host or device pointers (detected like MemoryManager::IsDevicePointer); when the output is not
 * pre-allocated the library allocates it in the same memory space as the input
 * (CompressionHighLevel.hpp:149-162): host outputs with malloc (release with free), device
 * outputs with hipMalloc (release with hipFree / mgh_free_device).
 */
#ifndef MGARD_HIP_COMPRESS_H
#define MGARD_HIP_COMPRESS_H

#include <stddef.h>
#include <stdint.h>

#include "mgard_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MGH_ERR_OUTPUT_TOO_LARGE (-7) /* cf. compress_status_type::OutputTooLargeFailure */
#define MGH_ERR_FORMAT (-8)           /* malformed / unsupported compressed stream */

typedef enum mgh_domain_decomposition { /* domain_decomposition_type, Utilities/Types.h:50 */
  MGH_DD_MAXDIM = 0,
  MGH_DD_BLOCK = 1,
  MGH_DD_VARIABLE = 2
} mgh_domain_decomposition;

typedef enum mgh_lossless { /* lossless_type, Utilities/Types.h:33-38 */
  MGH_LOSSLESS_HUFFMAN = 0,
  MGH_LOSSLESS_HUFFMAN_LZ4 = 1, /* not supported: MGH_ERR_INVALID_ARGUMENT */
  MGH_LOSSLESS_HUFFMAN_ZSTD = 2,
  MGH_LOSSLESS_CPU = 3 /* not supported */
} mgh_lossless;

/* The fields of mgard_x::Config this path reads; mgh_config_default() fills the reference's
 * defaults (Config.cpp:14-43). */
typedef struct mgh_config {
  int dev_id;
  int domain_decomposition;      /* mgh_domain_decomposition */
  int domain_decomposition_dim;  /* Variable */
  const uint64_t *domain_decomposition_sizes; /* Variable: extent of every subdomain */
  uint64_t num_domain_decomposition_sizes;
  uint64_t block_size;           /* Block */
  double estimate_outlier_ratio;
  uint64_t huff_dict_size;
  uint64_t huff_block_size;
  int lossless;                  /* mgh_lossless */
  int zstd_compress_level;
  int normalize_coordinates;
  uint64_t max_larget_level;
  uint64_t max_memory_footprint; /* bytes of device memory the call may plan with */
  int auto_pin_host_buffers; /* default 0 here (reference: 1), see mgh_config_default */
  int reorder;               /* 0 (default): quantized integers in the N-D layout; 1: level by level
                                (Config::reorder, LinearQuantization.hpp:46-146) -- recorded in the header */
  int mirror_reference_coord_cast; /* decompression of a NON-uniform grid: 1 = coordinates through (float)
                                first, as the reference does even for double data
                                (CompressionHighLevel.hpp:455-462); 0 (default) = at full precision */
} mgh_config;

void mgh_config_default(mgh_config *config);

/* mgard_x::compress. coords: NULL (uniform) or D host arrays of the data type. If
 * output_pre_allocated, *compressed_size carries the capacity in and the size out. */
int mgh_compress(int D, int dtype, const uint64_t *shape, double tol, double s,
                 int error_bound_type, const void *original_data, void **compressed_data,
                 size_t *compressed_size, const void *const *coords, const mgh_config *config,
                 int output_pre_allocated);

/* mgard_x::decompress. Shape and type come from the header (query them first with
 * mgh_infer_shape / mgh_infer_data_type to pre-allocate). */
int mgh_decompress(const void *compressed_data, size_t compressed_size,
                   void **decompressed_data, const mgh_config *config,
                   int output_pre_allocated);

/* mgh_decompress into a buffer of the caller whose size and type the library CHECKS (against the
 * header it reads anyway, before anything is written): out_bytes must be exactly the bytes of the
 * array in the stream, out_dtype its type (MGH_ERR_INVALID_ARGUMENT otherwise). The reference's
 * pre-allocated form (compress_x.hpp:115-154) trusts the caller; this is the form for bindings that
 * hand over buffers of known size (extension). Host or device memory like mgh_decompress. */
int mgh_decompress_into(const void *compressed_data, size_t compressed_size, void *out, size_t out_bytes,
                        int out_dtype, const mgh_config *config);

/* One process, several devices (the reference's MGARD_ENABLE_MULTI_DEVICE switch is dead code,
 * include/mgard-x/RuntimeX/RuntimeX.h:53; its multi-GPU example runs one rank per GPU:
 * examples/mgard-x/CompressXgcData/TestXGCAbsoluteError.cpp:36-252). mgh_compress_multi takes a
 * HOST buffer (container in host memory) or a volume resident on ONE device (GPUPipelines.hpp:69-207
 * takes device pointers: the slabs of other devices travel there once, device to device --
 * hipMemcpyPeerAsync over xGMI --, the slabs of the source device are compressed where they are,
 * and the container comes back in device memory of the source device); a pre-allocated output
 * must be of the same kind as the input. mgh_decompress_multi: host buffers only. The domain
 * is cut into slabs of the slowest dimension, slab id runs on device dev_ids[id % num_dev] (one
 * host thread, stream set and cache per device; an id may be listed more than once); a REL
 * bound uses the norm of the WHOLE domain (slab norms combined on the host,
 * ErrorToleranceCalculator.hpp:69-89) and every slab the ABS bound of :134-155. The result is an
 * ordinary MGARD-X container with a MaxDim decomposition of dimension 0: mgh_decompress,
 * mgh_decompress_multi and stock MGARD-X read it; mgh_decompress_multi shares the subdomains of
 * any container decomposed that way out over the devices and hands everything else to
 * mgh_decompress on dev_ids[0]. config->dev_id is ignored. */
int mgh_compress_multi(int num_dev, const int *dev_ids, int D, int dtype, const uint64_t *shape,
                       double tol, double s, int error_bound_type, const void *original_data,
                       void **compressed_data, size_t *compressed_size, const void *const *coords,
                       const mgh_config *config, int output_pre_allocated);
int mgh_decompress_multi(int num_dev, const int *dev_ids, const void *compressed_data,
                         size_t compressed_size, void **decompressed_data,
                         const mgh_config *config, int output_pre_allocated);

/* One RANK per GPU over RCCL (the reference's own multi-GPU pattern: one MPI rank per device, each
 * compressing its block -- examples/mgard-x/CompressXgcData/TestXGCAbsoluteError.cpp:36-252). The
 * ranks' slabs of the SLOWEST dimension, in rank order, form one domain: every rank passes the
 * shape of ITS slab (dimensions 1.. equal on all ranks; all ranks hold the same number of planes,
 * the last one may hold fewer, at least 3) resident on config->dev_id. nccl_comm: the caller's
 * ncclComm_t of `nranks` ranks (passed as void *, so this header needs no rccl.h). Collectives, on
 * a stream of the library: ncclAllGather of the slab shapes and of the record sizes, ncclAllReduce
 * of ONE double for the norm a REL bound refers to (MAX for s = inf, SUM of squares otherwise:
 * ErrorToleranceCalculator.hpp:69-131; every slab then runs with the ABS bound of :134-155),
 * ncclSend / ncclRecv of the records to `root`, which writes the container mgh_compress would write
 * for a MaxDim decomposition of dimension 0 (GPUPipelines.hpp:189-193) -- in DEVICE memory of the
 * root (hipMalloc'ed unless pre-allocated; release with mgh_free_device). Other ranks get
 * *compressed_size = 0 and may pass compressed_data = NULL. coords: NULL, or D host arrays with the
 * rank's own slice of dimension 0 and the full arrays of the other dimensions.
 * mgh_decompress_dist is the mirror: the root holds the container (device memory), every rank gets
 * its slab back in d_local_out (device memory it allocated: mgh_infer_shape on the root tells the
 * global shape). Every rank must make the call; a rank that fails its argument checks returns before
 * the first collective, a failure later leaves the others inside RCCL until its time-out.
 * RCCL is resolved at first use: symbols already in the process (an application that links RCCL),
 * else librccl.so.1; mgh_dist_use_library(path) names the library the communicator was created with
 * when that is another one (e.g. the copy a Python framework bundles). With nranks == 1 the calls
 * run their first collective and then ARE mgh_compress / mgh_decompress (same bytes). */
int mgh_dist_use_library(const char *librccl_path);
int mgh_compress_dist(void *nccl_comm, int rank, int nranks, int root, int D, int dtype,
                      const uint64_t *local_shape, double tol, double s, int error_bound_type,
                      const void *d_local_data, void **compressed_data, size_t *compressed_size,
                      const void *const *coords, const mgh_config *config, int output_pre_allocated);
int mgh_decompress_dist(void *nccl_comm, int rank, int nranks, int root, const void *compressed_data,
                        size_t compressed_size, void *d_local_out, const mgh_config *config);

/* infer_shape / infer_data_type (Metadata.hpp:244-247). compressed_data: host or device. */
int mgh_infer_shape(const void *compressed_data, size_t compressed_size, int *D_out,
                    uint64_t *shape_out /* [MGH_MAX_DIM] */);
int mgh_infer_data_type(const void *compressed_data, size_t compressed_size, int *dtype_out);

/* Stream contract of mgh_compress / mgh_decompress: the calls return when the result is
 * complete (they synchronise their own pipeline streams before returning). The pipeline streams
 * are created with default (blocking) flags like the reference's queues
 * (include/mgard-x/RuntimeX/DeviceAdapters/DeviceAdapterHip.h:514), i.e. they are ordered against
 * the NULL stream: a device-resident input that earlier work on the NULL stream is still
 * producing is complete before the first stage reads it. Work the caller has in flight on other
 * NON-BLOCKING streams is not waited for -- synchronise those before the call. */
/* pin_memory / check_memory_pinned / unpin_memory (compress_x.hpp:166-178): page-lock a host
 * buffer so that the pipeline's transfers out of / into it are asynchronous DMA. */
int mgh_pin_memory(void *ptr, size_t num_bytes);
int mgh_check_memory_pinned(const void *ptr); /* 1 = pinned, 0 = not */
int mgh_unpin_memory(void *ptr);
void mgh_free_device(void *p);
/* release_cache (compress_x.hpp:159): drops this thread's cached hierarchies and buffers. */
void mgh_release_cache(void);
/* Synchronous copy between any two of host / device memory (hipMemcpyDefault); lets a host
 * language without HIP bindings read the buffers the contexts own. */
int mgh_memcpy(void *dst, const void *src, size_t bytes);

/* ---- header (metadata) on its own: host only, no device needed -------------------------- */
typedef struct mgh_header_info {
  uint64_t version[3];
  int dtype;
  int D;
  uint64_t shape[MGH_MAX_DIM];
  int uniform;
  const double *coords[MGH_MAX_DIM]; /* !uniform: serialize reads them; parse points them into
                                        the caller's coords_storage */
  int error_bound_type; /* mgh_error_bound */
  double tol, s, norm;
  int domain_decomposed;
  int dd_method; /* mgh_domain_decomposition */
  uint64_t dd_dim, dd_size;
  uint64_t l_target;
  int reorder;
  int lossless; /* mgh_lossless */
  uint64_t huff_dict_size, huff_block_size;
} mgh_header_info;

/* Returns the number of bytes written (or needed when out == NULL), negative on error. */
int64_t mgh_metadata_serialize(const mgh_header_info *info, uint8_t *out, uint64_t capacity);
/* Parses preamble + header; *metadata_size_out = offset of the first subdomain record. */
int mgh_metadata_parse(const uint8_t *data, uint64_t size, mgh_header_info *info,
                       double *coords_storage, uint64_t coords_capacity,
                       uint64_t *metadata_size_out);

/* ---- lossless stage on its own (device pointers) ----------------------------------------- */
/* Host only: the canonical code built from a histogram, as it goes into the payload --
 * code[s] = (length << 56) | value (0: unused symbol), first[64] / entry[64] / keys[dict] with
 * the meaning Decode.hpp:52-106 gives them (reference GetCodebook.hpp:23-147 produces the
 * same three tables on the device). */
int mgh_huffman_codebook(const uint32_t *freq, uint64_t dict_size, uint64_t *code_out,
                         uint64_t *first_out /* [64] */, uint64_t *entry_out /* [64] */,
                         uint64_t *keys_out /* [dict_size] */);

typedef struct mgh_lossless_ctx mgh_lossless_ctx;
int mgh_lossless_create(mgh_lossless_ctx **out, int dev_id);
void mgh_lossless_destroy(mgh_lossless_ctx *ctx);
/* Quantized symbols (int64 in [0, dict_size)) + outlier list -> serialized payload in a host
 * buffer owned by the context (valid until the next call); *size_out bytes. */
int mgh_lossless_compress(mgh_lossless_ctx *ctx, const int64_t *d_quantized, uint64_t n,
                          uint64_t dict_size, uint64_t chunk_size, int lossless, int zstd_level,
                          const uint64_t *d_outlier_idx, const int64_t *d_outlier_val,
                          uint64_t outlier_count, const uint8_t **h_payload_out,
                          uint64_t *size_out, void *stream);
/* The same with the record written into DEVICE memory of the caller (d_record_out, any byte
 * alignment, `capacity` bytes): what the subdomain pipeline of mgh_compress does with every record
 * of a device-resident container -- the encoder stores its code units straight into the record
 * (GPUPipelines.hpp:189-193 lays the records out back to back, so a record starts wherever the
 * previous one ended). Huffman only (a Zstd frame is assembled on the host).
 * MGH_ERR_OUTPUT_TOO_LARGE when the record does not fit. */
int mgh_lossless_compress_device(mgh_lossless_ctx *ctx, const int64_t *d_quantized, uint64_t n,
                                 uint64_t dict_size, uint64_t chunk_size,
                                 const uint64_t *d_outlier_idx, const int64_t *d_outlier_val,
                                 uint64_t outlier_count, void *d_record_out, uint64_t capacity,
                                 uint64_t *size_out, void *stream);
/* Inverse: payload (host, or device memory at any byte alignment) -> d_quantized [n], outlier
 * list in device buffers owned by the
 * context (*d_outlier_idx_out / *d_outlier_val_out, *outlier_count_out entries). */
int mgh_lossless_decompress(mgh_lossless_ctx *ctx, const uint8_t *h_payload, uint64_t size,
                            int lossless, int64_t *d_quantized, uint64_t n,
                            const uint64_t **d_outlier_idx_out, const int64_t **d_outlier_val_out,
                            uint64_t *outlier_count_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MGARD_HIP_COMPRESS_H */
