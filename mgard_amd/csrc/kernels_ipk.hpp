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

// Developer ablation builds (tools/micro/ipk_phases.hip): bit 0 = stream in, bit 1 = solve,
// bit 2 = stream out. The library is always built with all three.
#ifndef MGH_IPK_PHASES
#define MGH_IPK_PHASES 7
#endif

namespace mgh {

// 16 bytes of T with the alignment of T only: the tiles start at arbitrary elements, and
// gfx950 serves global_load/store_dwordx4 at any 4-byte aligned address.
template <typename T> struct VecU;
template <> struct VecU<float> {
  typedef float aligned_type __attribute__((ext_vector_type(4)));
  typedef aligned_type type __attribute__((aligned(4)));
  static constexpr int N = 4;
};
template <> struct VecU<double> {
  typedef double aligned_type __attribute__((ext_vector_type(2)));
  typedef aligned_type type __attribute__((aligned(8)));
  static constexpr int N = 2;
};

// U consecutive table entries starting at t (wave-uniform address) through the VECTOR memory
// path: `zero` is a VGPR holding 0 that the compiler cannot see through, so the loads become
// global_load_dwordx4 (own in-order counter, one instruction per 16 bytes) instead of one
// s_load_dword per entry sharing lgkmcnt with the LDS traffic.
template <typename T, int U, bool VMEM>
__device__ __forceinline__ void table_load(const T *t, uint32_t zero, T (&out)[U]) {
  if constexpr (VMEM) {
    using VU = typename VecU<T>::type;
    using VA = typename VecU<T>::aligned_type;
    constexpr int VN = VecU<T>::N;
    static_assert(U % VN == 0, "batch must be whole vectors");
#pragma unroll
    for (int q = 0; q < U / VN; q++) {
      const VA v = *reinterpret_cast<const VU *>(t + zero + q * VN);
#pragma unroll
      for (int k = 0; k < VN; k++) out[q * VN + k] = v[k];
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; u++) out[u] = t[u];  // wave-uniform: scalar loads
  }
}

// U forward steps from element i: x[j] -= x[j-1] * w[j]. All LDS values and table entries of
// the batch are requested first; the solver wave then only issues the dependent chain (a wave
// issues at most one instruction every 4 cycles, so the instruction count of these loops IS the
// duration of a solve -- big batches amortise the one exposed LDS + table latency per batch).
template <typename T, int U, bool VMEM>
__device__ __forceinline__ void thomas_fwd(T *s, uint32_t stride, uint32_t i, const T *tt,
                                           uint32_t zero, T &prev) {
  T a[U], w[U];
  table_load<T, U, VMEM>(tt + i, zero, w);
#pragma unroll
  for (int u = 0; u < U; u++) a[u] = s[(i + u) * stride];
#pragma unroll
  for (int u = 0; u < U; u++) {
    a[u] = a[u] - prev * w[u];
    prev = a[u];
  }
#pragma unroll
  for (int u = 0; u < U; u++) s[(i + u) * stride] = a[u];
}

// U backward steps from element k downwards: x[j] = (x[j] - am[j] * x[j+1]) / bm[j].
template <typename T, int U, bool VMEM>
__device__ __forceinline__ void thomas_bwd(T *s, uint32_t stride, uint32_t k, const T *am,
                                           const T *bm, uint32_t zero, T &prev) {
  T a[U], ca[U], cb[U];  // ca[j] = am[k - (U-1) + j]
  table_load<T, U, VMEM>(am + (k - (U - 1)), zero, ca);
  table_load<T, U, VMEM>(bm + (k - (U - 1)), zero, cb);
#pragma unroll
  for (int u = 0; u < U; u++) a[u] = s[(k - u) * stride];
#pragma unroll
  for (int u = 0; u < U; u++) {
    a[u] = (a[u] - ca[U - 1 - u] * prev) / cb[U - 1 - u];
    prev = a[u];
  }
#pragma unroll
  for (int u = 0; u < U; u++) s[(k - u) * stride] = a[u];
}

// The two sweeps of one lane's pencil, stored in LDS at s[i * stride].
// tt: [0,n) forward multiplier am[i]/bm[i]; [n,2n) backward am[i+1]; [2n,3n) bm[i+1], all
// wave-uniform. BIG (the tiled kernels, long pencils): batches of 32 (16 for double) with the
// tables through the vector memory path; otherwise (the single-workgroup tail kernel, n <= 17,
// where every table is touched once and the compiler may hoist scalar loads freely): batches of
// 8 and scalar table loads.
template <typename T, bool BIG = true>
__device__ __forceinline__ void thomas_lds(T *s, uint32_t stride, uint32_t n,
                                           const T *__restrict__ tt) {
  constexpr int UB = !BIG ? 8 : (sizeof(T) == 4 ? 32 : 16), US = 8;
  T prev = 0;
  uint32_t i = 0;
  uint32_t zero = 0;
  if constexpr (BIG) asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
  for (; i + UB <= n; i += UB) thomas_fwd<T, UB, BIG>(s, stride, i, tt, zero, prev);
  if constexpr (UB != US)
    for (; i + US <= n; i += US) thomas_fwd<T, US, BIG>(s, stride, i, tt, zero, prev);
  for (; i < n; i++) {
    T a = s[i * stride];
    a = a - prev * tt[i];
    s[i * stride] = a;
    prev = a;
  }
  prev = 0;
  const T *am = tt + n, *bm = tt + 2 * n;
  int64_t k = (int64_t)n - 1;
  for (; k >= UB - 1; k -= UB) thomas_bwd<T, UB, BIG>(s, stride, (uint32_t)k, am, bm, zero, prev);
  if constexpr (UB != US)
    for (; k >= US - 1; k -= US) thomas_bwd<T, US, BIG>(s, stride, (uint32_t)k, am, bm, zero, prev);
  for (; k >= 0; k--) {
    T a = s[k * stride];
    a = (a - am[k] * prev) / bm[k];
    s[k * stride] = a;
    prev = a;
  }
}

__device__ __forceinline__ bool same_bits(float a, float b) { return __float_as_uint(a) == __float_as_uint(b); }
__device__ __forceinline__ bool same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }

// The two sweeps of a pencil cut into NCH chunks, one lane per chunk, every chunk VERIFIED
// against the sequential sweep (the idea of kernels_ipk_spec.hpp inside a tile that sits in
// LDS): with one lane per pencil a 64-pencil tile keeps one wave busy for 2 n dependent steps
// (the backward step is a division) while the workgroup's other waves wait at a barrier.
// Here lane (c, j) sweeps chunk c of pencil j: it starts the recurrence from state 0 K elements
// in front of its chunk -- the multipliers are below 0.5, the state it arrives with at its first
// element has in practice met the sequential run -- and the value it computed for the element
// in front of its chunk must equal, BIT FOR BIT, the last value of chunk c - 1 (exact by
// induction: chunk 0 starts where the sequential sweep starts; two runs of one floating-point
// recurrence that agree in one value agree in all later ones). The function returns true if any
// lane of the workgroup saw a mismatch: the tile's content is then undefined and the caller
// reloads it and runs the one-lane sweep (thomas_lds). Results are therefore those of
// thomas_lds in every bit, by construction. The sweeps are in place: all warm-ups (reads only)
// run before a barrier, the chunks' own sweeps (in place) behind it.
// Must be called by ALL threads of the workgroup (barriers); c < NCH and j < 64 for live lanes;
// edge: 2 * NCH * 64 values of LDS.
template <typename T, int NCH>
__device__ __forceinline__ bool thomas_chunked(T *s, uint32_t stride, uint32_t n,
                                               const T *__restrict__ tt, uint32_t K, uint32_t c,
                                               uint32_t j, bool live, T *edge) {
  const uint32_t CS = (n + NCH - 1) / NCH;
  const uint32_t start = min(n, c * CS), end = min(n, start + CS);
  const T *am = tt + n, *bm = tt + 2 * n;
  uint32_t zero = 0;
  asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
  T *spec = edge + c * 64 + j, *last = edge + (NCH + c) * 64 + j;
  T prev = 0;
  // ---- forward: x[i] -= x[i-1] * w[i]. Warm-up (reads only), barrier, the chunk in place.
  if (live) {
    uint32_t i = start > K ? start - K : 0;
    for (; i + 8 <= start; i += 8) {
      T a[8], w[8];
      table_load<T, 8, true>(tt + i, zero, w);
#pragma unroll
      for (int u = 0; u < 8; u++) a[u] = s[(i + u) * stride];
#pragma unroll
      for (int u = 0; u < 8; u++) prev = a[u] - prev * w[u];
    }
    for (; i < start; i++) prev = s[i * stride] - prev * tt[i];
    *spec = prev;
  }
  __syncthreads();
  if (live) {
    uint32_t i = start;
    for (; i + 8 <= end; i += 8) thomas_fwd<T, 8, true>(s, stride, i, tt, zero, prev);
    for (; i < end; i++) {
      T a = s[i * stride];
      a = a - prev * tt[i];
      s[i * stride] = a;
      prev = a;
    }
    *last = prev;
  }
  __syncthreads();
  bool bad = live && c >= 1 && !same_bits(*spec, edge[(NCH + c - 1) * 64 + j]);
  // ---- backward: x[k] = (x[k] - am[k] * x[k+1]) / bm[k]
  prev = 0;
  if (live) {
    int64_t k = (int64_t)min((uint64_t)n, (uint64_t)end + K) - 1;
    for (; k - 7 >= (int64_t)end; k -= 8) {
      T a[8], ca[8], cb[8];
      table_load<T, 8, true>(am + (k - 7), zero, ca);
      table_load<T, 8, true>(bm + (k - 7), zero, cb);
#pragma unroll
      for (int u = 0; u < 8; u++) a[u] = s[(k - u) * stride];
#pragma unroll
      for (int u = 0; u < 8; u++) prev = (a[u] - ca[7 - u] * prev) / cb[7 - u];
    }
    for (; k >= (int64_t)end; k--) prev = (s[k * stride] - am[k] * prev) / bm[k];
  }
  if (__syncthreads_or(bad)) return true;  // (forward results of all chunks final, nobody has written yet)
  if (live) {
    *spec = prev;
    int64_t k = (int64_t)end - 1;
    for (; k - 7 >= (int64_t)start; k -= 8) thomas_bwd<T, 8, true>(s, stride, (uint32_t)k, am, bm, zero, prev);
    for (; k >= (int64_t)start; k--) {
      T a = s[k * stride];
      a = (a - am[k] * prev) / bm[k];
      s[k * stride] = a;
      prev = a;
    }
    *last = prev;  // (the value at `start`)
  }
  __syncthreads();
  bad = live && c + 1 < NCH && !same_bits(*spec, edge[(NCH + c + 1) * 64 + j]);
  return __syncthreads_or(bad);
}

// Pencils contiguous in memory (solve along the fastest dim): x is an
// [npencil][n] matrix, so a tile of P pencils is one contiguous chunk. All four
// waves of the block stream the chunk into LDS with 16-byte loads, UV of them in
// flight per lane (the streaming phases are latency-bound: a tile is resident in LDS
// for load + solve + store, and LDS capacity limits the tiles per CU), wave 0 runs
// the P sweeps, all waves stream the result out. LDS rows are padded by `pad` (0 for
// odd n, 1 for even n) so that lane t walking row t is bank-conflict free;
// row = e / n is computed as umulhi(e, magic).
// CH and K > 0: the solve of a tile is shared by the four waves (thomas_chunked, warm-up K, P <= 64).
template <typename T, bool CH = false>
__global__ void __launch_bounds__(256, CH ? 3 : 1)
k_ipk_lds_contig(uint32_t npencil, uint32_t n, uint32_t pad, uint32_t magic, uint32_t P,
                 T *__restrict__ x, const T *__restrict__ tt, T *__restrict__ add_to, int sign,
                 uint32_t K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *sm = reinterpret_cast<T *>(smem_raw);
  __shared__ T edge[CH ? 2 * 4 * 64 : 1];
  using VU = typename VecU<T>::type;
  using VA = typename VecU<T>::aligned_type;
  constexpr int VN = VecU<T>::N;
  constexpr int UV = 8;
  const uint32_t tid = threadIdx.x;
  const uint32_t p0 = blockIdx.x * P;
  const uint32_t cnt = min(P, npencil - p0);
  const size_t base = (size_t)p0 * n;
  const uint32_t total = cnt * n;
  const uint32_t nvec = total / VN;
  const T *g = x + base;
  auto lds_put = [&](uint32_t e, const VA &v) {
    if (pad) {
#pragma unroll
      for (int k = 0; k < VN; k++) sm[e + k + __umulhi(e + k, magic)] = v[k];
    } else {
      *reinterpret_cast<VA *>(sm + e) = v;
    }
  };
  auto lds_get = [&](uint32_t e) {
    VA v;
    if (pad) {
#pragma unroll
      for (int k = 0; k < VN; k++) v[k] = sm[e + k + __umulhi(e + k, magic)];
    } else {
      v = *reinterpret_cast<const VA *>(sm + e);
    }
    return v;
  };
  // attempt 0 (K > 0 only): chunked solve; attempt 1: one lane per pencil, after a chunked solve
  // that did not verify (x still holds the right-hand sides: the result is written at the end)
  for (int attempt = CH && K > 0 ? 0 : 1; attempt < 2; attempt++) {
    for (uint32_t q = tid; (MGH_IPK_PHASES & 1) && q < nvec; q += UV * 256) {
      VA v[UV];
#pragma unroll
      for (int u = 0; u < UV; u++) {
        const uint32_t qq = min(q + u * 256, nvec - 1);
        v[u] = *reinterpret_cast<const VU *>(g + (size_t)qq * VN);
      }
#pragma unroll
      for (int u = 0; u < UV; u++)
        if (q + u * 256 < nvec) lds_put((q + u * 256) * VN, v[u]);
    }
    for (uint32_t e = nvec * VN + tid; e < total; e += 256)
      sm[e + (pad ? __umulhi(e, magic) : 0u)] = g[e];
    __syncthreads();
    if (CH && attempt == 0) {
      const uint32_t c = __builtin_amdgcn_readfirstlane(tid >> 6), j = tid & 63;
      if (!thomas_chunked<T, 4>(sm + j * (n + pad), 1, n, tt, K, c, j, j < cnt, edge)) break;
      continue;
    }
    // (the chunked variant's second pass is rare: small batches keep its registers out of the budget)
    if ((MGH_IPK_PHASES & 2) && tid < cnt) thomas_lds<T, !CH>(sm + tid * (n + pad), 1, n, tt);
  }
  __syncthreads();
  T *o = (add_to ? add_to : x) + base;
  if (!(MGH_IPK_PHASES & 4)) return;
  if (add_to) {
    for (uint32_t q = tid; q < nvec; q += UV * 256) {
      VA v[UV];
#pragma unroll
      for (int u = 0; u < UV; u++) {
        const uint32_t qq = min(q + u * 256, nvec - 1);
        v[u] = *reinterpret_cast<const VU *>(o + (size_t)qq * VN);
      }
#pragma unroll
      for (int u = 0; u < UV; u++) {
        if (q + u * 256 < nvec) {
          const uint32_t e = (q + u * 256) * VN;
          const VA d = lds_get(e);
          VA r;
#pragma unroll
          for (int k = 0; k < VN; k++) r[k] = sign > 0 ? v[u][k] + d[k] : v[u][k] - d[k];
          *reinterpret_cast<VU *>(o + e) = r;
        }
      }
    }
    for (uint32_t e = nvec * VN + tid; e < total; e += 256) {
      const T d = sm[e + (pad ? __umulhi(e, magic) : 0u)];
      o[e] = sign > 0 ? o[e] + d : o[e] - d;
    }
  } else {
    for (uint32_t q = tid; q < nvec; q += 256) *reinterpret_cast<VU *>(o + (size_t)q * VN) = lds_get(q * VN);
    for (uint32_t e = nvec * VN + tid; e < total; e += 256)
      o[e] = sm[e + (pad ? __umulhi(e, magic) : 0u)];
  }
}

// Strided pencils (solve along a slower dim): the tile is W consecutive
// elements of the fastest dim (every row access is a coalesced segment) by the
// whole pencil length; the pencil of column c sits in LDS at sm[i * W + c]
// (conflict free). A thread moves 16 bytes (VN neighbouring pencils) of a row per
// access, UR rows in flight; wave 0 runs the sweeps. Pencil id p in
// [0, n_outer * n_inner): base = (p / n_inner) * outer_stride + p % n_inner; consecutive
// positions are `stride` elements apart. Where the VN pencils of a thread are not
// neighbours in memory (the tile crosses an outer boundary, or the end of the range)
// the thread falls back to element accesses. Launch with a grid padded to a multiple of 8.
template <typename T, int W>
__global__ void __launch_bounds__(256)
k_ipk_lds_strided(uint32_t n_outer, uint32_t n_inner, size_t outer_stride, size_t stride,
                  uint32_t n, T *__restrict__ x, const T *__restrict__ tt,
                  T *__restrict__ add_to, int sign) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *sm = reinterpret_cast<T *>(smem_raw);
  using VU = typename VecU<T>::type;
  using VA = typename VecU<T>::aligned_type;
  constexpr int VN = VecU<T>::N;
  constexpr int WV = W / VN;     // 16-byte columns of the tile
  constexpr int RW = 256 / WV;   // rows covered per pass by the block
  constexpr int UR = 9;
  static_assert(W % VN == 0, "tile width must be a multiple of the vector width");
  const uint32_t npencil = n_outer * n_inner;
  const uint32_t cv = threadIdx.x % WV;
  const uint32_t r0 = threadIdx.x / WV;
  // workgroups go round-robin to the 8 XCDs (own L2 each): give every XCD a contiguous range of
  // tiles, so that the cache lines two neighbouring tiles share are fetched by one L2
  const uint32_t per = (gridDim.x + 7) / 8;
  const uint32_t bid = (blockIdx.x % 8) * per + blockIdx.x / 8;
  if ((uint64_t)bid * W >= (uint64_t)n_outer * n_inner) return;  // grid is padded to a multiple of 8
  const uint32_t p = bid * W + cv * VN;  // first pencil of this thread
  const bool rows_live = r0 < (uint32_t)RW;
  const uint32_t po = p / n_inner;
  const bool vec = p + VN - 1 < npencil && (p + VN - 1) / n_inner == po;
  size_t base[VN];
  bool live[VN];
#pragma unroll
  for (int k = 0; k < VN; k++) {
    live[k] = p + k < npencil;
    const uint32_t pk = live[k] ? p + k : 0;
    base[k] = (size_t)(pk / n_inner) * outer_stride + (pk % n_inner);
  }
  auto gload = [&](const T *src, uint32_t row) {
    VA v;
    if (vec) {
      v = *reinterpret_cast<const VU *>(src + base[0] + (size_t)row * stride);
    } else {
#pragma unroll
      for (int k = 0; k < VN; k++) v[k] = live[k] ? src[base[k] + (size_t)row * stride] : T(0);
    }
    return v;
  };
  auto gstore = [&](T *dst, uint32_t row, const VA &v) {
    if (vec) {
      *reinterpret_cast<VU *>(dst + base[0] + (size_t)row * stride) = v;
    } else {
#pragma unroll
      for (int k = 0; k < VN; k++)
        if (live[k]) dst[base[k] + (size_t)row * stride] = v[k];
    }
  };
  if ((MGH_IPK_PHASES & 1) && rows_live) {
    for (uint32_t i = r0; i < n; i += UR * RW) {
      VA v[UR];
#pragma unroll
      for (int u = 0; u < UR; u++) v[u] = gload(x, min(i + u * RW, n - 1));
#pragma unroll
      for (int u = 0; u < UR; u++)
        if (i + u * RW < n) *reinterpret_cast<VA *>(sm + (i + u * RW) * W + cv * VN) = v[u];
    }
  }
  __syncthreads();
  if ((MGH_IPK_PHASES & 2) && threadIdx.x < W && bid * W + threadIdx.x < npencil)
    thomas_lds<T>(sm + threadIdx.x, W, n, tt);
  __syncthreads();
  if ((MGH_IPK_PHASES & 4) && rows_live) {
    if (add_to) {
      for (uint32_t i = r0; i < n; i += UR * RW) {
        VA v[UR];
#pragma unroll
        for (int u = 0; u < UR; u++) v[u] = gload(add_to, min(i + u * RW, n - 1));
#pragma unroll
        for (int u = 0; u < UR; u++) {
          if (i + u * RW < n) {
            const VA d = *reinterpret_cast<const VA *>(sm + (i + u * RW) * W + cv * VN);
            VA r;
#pragma unroll
            for (int k = 0; k < VN; k++) r[k] = sign > 0 ? v[u][k] + d[k] : v[u][k] - d[k];
            gstore(add_to, i + u * RW, r);
          }
        }
      }
    } else {
      for (uint32_t i = r0; i < n; i += RW)
        gstore(x, i, *reinterpret_cast<const VA *>(sm + i * W + cv * VN));
    }
  }
}

// f-solve and c-solve of one coarse r-plane in ONE workgroup (both are independent per plane):
// the plane sits in LDS with an odd row pitch, rows are solved by one lane each (lane t walks
// row t: conflict-free because the pitch is odd), then columns (lane t walks column t). One
// launch and one pass over HBM instead of two for every level whose coarse plane fits in LDS
// -- on those levels a launch is mostly dispatch latency. The plane streams in and out with
// 16-byte accesses, UV in flight per lane; magic = ceil(2^32 / m2) gives e / m2 as a umulhi.
template <typename T>
__global__ void __launch_bounds__(256)
k_ipk_plane_fc(uint32_t m1, uint32_t m2, uint32_t pitch, uint32_t magic, T *__restrict__ x,
               const T *__restrict__ tt_f, const T *__restrict__ tt_c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *sm = reinterpret_cast<T *>(smem_raw);
  using VU = typename VecU<T>::type;
  using VA = typename VecU<T>::aligned_type;
  constexpr int VN = VecU<T>::N;
  constexpr int UV = 8;
  T *g = x + (size_t)blockIdx.x * m1 * m2;
  const uint32_t total = m1 * m2;
  const uint32_t nvec = total / VN;
  const uint32_t padw = pitch - m2;  // 0 or 1
  auto at = [&](uint32_t e) { return e + (padw ? __umulhi(e, magic) : 0u); };
  for (uint32_t q = threadIdx.x; q < nvec; q += UV * 256) {
    VA v[UV];
#pragma unroll
    for (int u = 0; u < UV; u++)
      v[u] = *reinterpret_cast<const VU *>(g + (size_t)min(q + u * 256, nvec - 1) * VN);
#pragma unroll
    for (int u = 0; u < UV; u++) {
      if (q + u * 256 < nvec) {
        const uint32_t e = (q + u * 256) * VN;
        if (padw) {
#pragma unroll
          for (int k = 0; k < VN; k++) sm[at(e + k)] = v[u][k];
        } else {
          *reinterpret_cast<VA *>(sm + e) = v[u];
        }
      }
    }
  }
  for (uint32_t e = nvec * VN + threadIdx.x; e < total; e += 256) sm[at(e)] = g[e];
  __syncthreads();
  for (uint32_t r = threadIdx.x; r < m1; r += 256) thomas_lds<T>(sm + r * pitch, 1, m2, tt_f);
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < m2; c += 256) thomas_lds<T>(sm + c, pitch, m1, tt_c);
  __syncthreads();
  for (uint32_t q = threadIdx.x; q < nvec; q += 256) {
    const uint32_t e = q * VN;
    VA v;
    if (padw) {
#pragma unroll
      for (int k = 0; k < VN; k++) v[k] = sm[at(e + k)];
    } else {
      v = *reinterpret_cast<const VA *>(sm + e);
    }
    *reinterpret_cast<VU *>(g + e) = v;
  }
  for (uint32_t e = nvec * VN + threadIdx.x; e < total; e += 256) g[e] = sm[at(e)];
}

} // namespace mgh
