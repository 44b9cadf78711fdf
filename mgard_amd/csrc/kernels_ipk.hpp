// LDS-staged Thomas solves (IPK) for gfx950.
//
// One wavefront owns up to 64 pencils: the whole pencils are brought into LDS
// with coalesced 256-byte row reads, each lane then runs the forward and the
// backward substitution of ITS pencil entirely out of LDS (one HBM read and one
// HBM write per element instead of two of each), and the result goes back with
// coalesced writes -- optionally accumulated (+/-) into the coarse nodal array
// (AddND / SubtractND fused).  Arithmetic and order are those of
// tridiag_forward2 / tridiag_backward2 (reference
// include/mgard-x/DataRefactoring/MultiDimension/Correction/IPKFunctor.h:127,147)
// with the am/bm indexing of IterativeProcessingKernel3D.hpp:108-124,223-262;
// results are bit-identical to kernels_v1.hpp:k_ipk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mgh {

// The two sweeps of one lane's pencil, stored in LDS at s[i * stride].
// tt: [0,n) forward multiplier am[i]/bm[i]; [n,2n) backward am[i+1]; [2n,3n) bm[i+1]
// (all wave-uniform -> scalar loads).
template <typename T>
__device__ __forceinline__ void thomas_lds(T *s, uint32_t stride, uint32_t n,
                                           const T *__restrict__ tt) {
  T prev = 0;
  uint32_t i = 0;
  constexpr int U = 8;
  for (; i + U <= n; i += U) {
    T a[U];
#pragma unroll
    for (int u = 0; u < U; u++) a[u] = s[(i + u) * stride];
#pragma unroll
    for (int u = 0; u < U; u++) {
      a[u] = a[u] - prev * tt[i + u];
      prev = a[u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) s[(i + u) * stride] = a[u];
  }
  for (; i < n; i++) {
    T a = s[i * stride];
    a = a - prev * tt[i];
    s[i * stride] = a;
    prev = a;
  }
  prev = 0;
  const T *am = tt + n, *bm = tt + 2 * n;
  int64_t k = (int64_t)n - 1;
  for (; k >= U - 1; k -= U) {
    T a[U];
#pragma unroll
    for (int u = 0; u < U; u++) a[u] = s[(k - u) * stride];
#pragma unroll
    for (int u = 0; u < U; u++) {
      a[u] = (a[u] - am[k - u] * prev) / bm[k - u];
      prev = a[u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) s[(k - u) * stride] = a[u];
  }
  for (; k >= 0; k--) {
    T a = s[k * stride];
    a = (a - am[k] * prev) / bm[k];
    s[k * stride] = a;
    prev = a;
  }
}

// Pencils contiguous in memory (solve along the fastest dim): x is an
// [npencil][n] matrix, so a tile of P pencils is one contiguous chunk. All four
// waves of the block stream the chunk into LDS (8 loads in flight per lane),
// wave 0 runs the P sweeps, all waves stream the result out. LDS rows are padded
// by `pad` (0 for odd n, 1 for even n) so that lane t walking row t is
// bank-conflict free; row = e / n is computed as umulhi(e, magic).
template <typename T>
__global__ void __launch_bounds__(256)
k_ipk_lds_contig(uint32_t npencil, uint32_t n, uint32_t pad, uint32_t magic, uint32_t P,
                 T *__restrict__ x, const T *__restrict__ tt, T *__restrict__ add_to, int sign) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *sm = reinterpret_cast<T *>(smem_raw);
  const uint32_t tid = threadIdx.x;
  const uint32_t p0 = blockIdx.x * P;
  const uint32_t cnt = min(P, npencil - p0);
  const size_t base = (size_t)p0 * n;
  const uint32_t total = cnt * n;
  const T *g = x + base;
  constexpr int U = 8;
  uint32_t e = tid;
  for (; e + (U - 1) * 256 < total; e += U * 256) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = g[e + u * 256];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t ee = e + u * 256;
      sm[ee + (pad ? __umulhi(ee, magic) : 0u)] = v[u];
    }
  }
  for (; e < total; e += 256) sm[e + (pad ? __umulhi(e, magic) : 0u)] = g[e];
  __syncthreads();
  if (tid < cnt) thomas_lds<T>(sm + tid * (n + pad), 1, n, tt);
  __syncthreads();
  T *o = (add_to ? add_to : x) + base;
  e = tid;
  if (add_to) {
    for (; e + (U - 1) * 256 < total; e += U * 256) {
      T v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = o[e + u * 256];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t ee = e + u * 256;
        const T d = sm[ee + (pad ? __umulhi(ee, magic) : 0u)];
        o[ee] = sign > 0 ? v[u] + d : v[u] - d;
      }
    }
    for (; e < total; e += 256) {
      const T d = sm[e + (pad ? __umulhi(e, magic) : 0u)];
      o[e] = sign > 0 ? o[e] + d : o[e] - d;
    }
  } else {
    for (; e < total; e += 256) o[e] = sm[e + (pad ? __umulhi(e, magic) : 0u)];
  }
}

// Strided pencils (solve along a slower dim): the tile is W consecutive
// elements of the fastest dim (every row access is a coalesced segment) by the
// whole pencil length; the pencil of column c sits in LDS at sm[i * W + c]
// (conflict free). The four waves split the rows for the streaming phases,
// wave 0 runs the sweeps. Pencil id p in [0, n_outer * n_inner):
// base = (p / n_inner) * outer_stride + p % n_inner; consecutive positions are
// `stride` elements apart. W = 64, or 32 when 64 whole pencils do not fit in LDS.
template <typename T, int W>
__global__ void __launch_bounds__(256)
k_ipk_lds_strided(uint32_t n_outer, uint32_t n_inner, size_t outer_stride, size_t stride,
                  uint32_t n, T *__restrict__ x, const T *__restrict__ tt,
                  T *__restrict__ add_to, int sign) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *sm = reinterpret_cast<T *>(smem_raw);
  constexpr int RW = 256 / W;  // rows covered per pass by the block (W = 48: 240 threads busy)
  const uint32_t col = threadIdx.x % W;
  const uint32_t r0 = threadIdx.x / W;
  const uint32_t p = blockIdx.x * W + col;
  const bool live = r0 < (uint32_t)RW && p < n_outer * n_inner;
  const size_t base = live ? (size_t)(p / n_inner) * outer_stride + (p % n_inner) : 0;
  constexpr int U = 8;
  if (live) {
    uint32_t i = r0;
    for (; i + (U - 1) * RW < n; i += U * RW) {
      T v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = x[base + (size_t)(i + u * RW) * stride];
#pragma unroll
      for (int u = 0; u < U; u++) sm[(i + u * RW) * W + col] = v[u];
    }
    for (; i < n; i += RW) sm[i * W + col] = x[base + (size_t)i * stride];
  }
  __syncthreads();
  if (threadIdx.x < W && live) thomas_lds<T>(sm + col, W, n, tt);
  __syncthreads();
  if (live) {
    T *o = (add_to ? add_to : x) + base;
    uint32_t i = r0;
    if (add_to) {
      for (; i + (U - 1) * RW < n; i += U * RW) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = o[(size_t)(i + u * RW) * stride];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const T d = sm[(i + u * RW) * W + col];
          o[(size_t)(i + u * RW) * stride] = sign > 0 ? v[u] + d : v[u] - d;
        }
      }
      for (; i < n; i += RW) {
        const T d = sm[i * W + col];
        T *q = o + (size_t)i * stride;
        *q = sign > 0 ? *q + d : *q - d;
      }
    } else {
      for (; i < n; i += RW) o[(size_t)i * stride] = sm[i * W + col];
    }
  }
}

// f-solve and c-solve of one coarse r-plane in ONE workgroup (both are independent per plane):
// the plane sits in LDS with an odd row pitch, rows are solved by one lane each (lane t walks
// row t: conflict-free because the pitch is odd), then columns (lane t walks column t). One
// launch and one pass over HBM instead of two for every level whose coarse plane fits in LDS
// -- on those levels a launch is mostly dispatch latency.
template <typename T>
__global__ void __launch_bounds__(256)
k_ipk_plane_fc(uint32_t m1, uint32_t m2, uint32_t pitch, T *__restrict__ x,
               const T *__restrict__ tt_f, const T *__restrict__ tt_c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *sm = reinterpret_cast<T *>(smem_raw);
  T *g = x + (size_t)blockIdx.x * m1 * m2;
  const uint32_t total = m1 * m2;
  for (uint32_t e = threadIdx.x; e < total; e += 256) sm[(e / m2) * pitch + e % m2] = g[e];
  __syncthreads();
  for (uint32_t r = threadIdx.x; r < m1; r += 256) thomas_lds<T>(sm + r * pitch, 1, m2, tt_f);
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < m2; c += 256) thomas_lds<T>(sm + c, pitch, m1, tt_c);
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < total; e += 256) g[e] = sm[(e / m2) * pitch + e % m2];
}

} // namespace mgh
