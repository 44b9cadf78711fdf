// Streaming Thomas solves (IPK) for gfx950: every wave is a solver.
//
// A one-wave workgroup owns W <= 64 pencils, one per lane, and there are no
// separate stream-in / stream-out phases and no barriers: the lane loads the
// elements of ITS pencil straight from global memory a batch ahead of the
// dependent chain (strided pencils: a row of W neighbouring pencils is one
// coalesced segment; contiguous pencils: 16-byte pieces of the lane's own row,
// a whole 128-byte line per batch), runs the forward substitution, parks the
// forward results, then runs the backward substitution and stores each batch
// (optionally accumulated +/- into the coarse nodal array: AddND / SubtractND
// fused) as soon as its chain is done.
//
// Where the forward results are parked decides how many pencils a CU can work
// on at once (the chain is pure dependent-operation latency, so residency is
// the throughput): the remainder and the last full batch stay in registers,
// the leading `n_glob` elements go back to global memory in place (they are
// re-read a batch ahead of the backward chain; L2-resident) and only the rest
// occupies LDS at sm[(i - n_glob) * W + lane] (conflict-free). The host picks
// W and n_glob so that all tiles of a level are resident in ONE round.
//
// Arithmetic and order are those of tridiag_forward2 / tridiag_backward2
// (reference include/mgard-x/DataRefactoring/MultiDimension/Correction/IPKFunctor.h:127,147)
// with the am/bm indexing of IterativeProcessingKernel3D.hpp:108-124,223-262;
// results are bit-identical to kernels_v1.hpp:k_ipk.
//
// FD = true: the quotient of the backward step is formed from a tabulated
// correctly rounded reciprocal y = RN(1/b) with Markstein's FMA sequence
//   q0 = RN(a y); r0 = fma(-b, q0, a); q1 = fma(r0, y, q0); r1 = fma(-b, q1, a); q = fma(r1, y, q1)
// which equals the IEEE quotient RN(a / b) whenever no intermediate under- or
// overflows (5 dependent operations instead of the 10 of the IEEE expansion).
// Every numerator is checked against a safe window (or +0); a batch with any
// numerator outside it is redone with the IEEE division, so the result is
// bit-identical in all cases (tools/micro/fastdiv_check.hip: exhaustive check).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_ipk.hpp"

namespace mgh {

template <typename T> __device__ __forceinline__ T fma_t(T a, T b, T c);
template <> __device__ __forceinline__ float fma_t<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> __device__ __forceinline__ double fma_t<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T> __device__ __forceinline__ T div_markstein(T a, T b, T y) {
  T q = a * y;
  T r = fma_t<T>(-b, q, a);
  q = fma_t<T>(r, y, q);
  r = fma_t<T>(-b, q, a);
  return fma_t<T>(r, y, q);
}

// numerator outside the window in which div_markstein is proven equal to a / b
// (+0 is fine as well: every step of the sequence reproduces it)
__device__ __forceinline__ bool fd_unsafe(float a) {
  const float m = __builtin_fabsf(a);
  return !(m >= 0x1p-80f && m <= 0x1p80f) && __float_as_uint(a) != 0u;
}
__device__ __forceinline__ bool fd_unsafe(double a) {
  const double m = __builtin_fabs(a);
  return !(m >= 0x1p-900 && m <= 0x1p900) && __double_as_longlong(a) != 0ll;
}

// Access to the lane's pencil: wave-uniform array pointer + 32-bit lane offset `lo` (BYTES), so
// that every access is `global_load/store v, v_off, s[base:base+1]` and the uniform part of the
// address is scalar arithmetic (arrays of < 2^32 bytes; the host checks). STRIDED: element i at
// x[lo + i * stride]. Contiguous: x[lo + i], moved in 16-byte pieces (the rows start at arbitrary
// elements: 4/8-byte alignment only).
template <typename T, int U, bool CONTIG> struct PencilIO {
  using VU = typename VecU<T>::type;
  using VA = typename VecU<T>::aligned_type;
  static constexpr int VN = VecU<T>::N;
  static __device__ __forceinline__ void load(const T *x, uint32_t lo, size_t stride, uint32_t i, T (&v)[U]) {
    if constexpr (CONTIG) {
      static_assert(U % VN == 0, "batch must be whole vectors");
      const T *r = x + i;
#pragma unroll
      for (int q = 0; q < U / VN; q++) {
        const VA t = *reinterpret_cast<const VU *>(reinterpret_cast<const char *>(r + q * VN) + lo);
#pragma unroll
        for (int k = 0; k < VN; k++) v[q * VN + k] = t[k];
      }
    } else {
      const T *r = x + (size_t)i * stride;
#pragma unroll
      for (int u = 0; u < U; u++) {
        v[u] = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(r) + lo);
        r += stride;
      }
    }
  }
  static __device__ __forceinline__ void store(T *x, uint32_t lo, size_t stride, uint32_t i, const T (&v)[U]) {
    if constexpr (CONTIG) {
      T *r = x + i;
#pragma unroll
      for (int q = 0; q < U / VN; q++) {
        VA t;
#pragma unroll
        for (int k = 0; k < VN; k++) t[k] = v[q * VN + k];
        *reinterpret_cast<VU *>(reinterpret_cast<char *>(r + q * VN) + lo) = t;
      }
    } else {
      T *r = x + (size_t)i * stride;
#pragma unroll
      for (int u = 0; u < U; u++) {
        *reinterpret_cast<T *>(reinterpret_cast<char *>(r) + lo) = v[u];
        r += stride;
      }
    }
  }
};

// Contiguous pencils, wave-cooperative: a tile of W <= 64 consecutive rows (row pitch n elements).
// A lane reading 64 bytes of ITS row per batch makes every load instruction touch 64 different
// cache lines (measured: the contiguous streaming solve 1.5-2 x slower than the strided one on the
// same array). Here the WAVE moves the batch -- U = 16 columns of all its rows -- with QL lanes
// per row and 16 bytes per lane (16 rows = 16 segments of 64 bytes per instruction for float),
// and the rows reach their lanes through a small LDS staging area st[row * (U + 1) + col]
// (odd pitch: conflict-free both ways). All 64 lanes of the wave must call these together.
template <typename T, int U> struct TileIO {
  using VU = typename VecU<T>::type;
  using VA = typename VecU<T>::aligned_type;
  static constexpr int VN = VecU<T>::N;
  static constexpr int QL = U / VN;        // lanes per row
  static constexpr int RP = 64 / QL;       // rows per instruction
  static constexpr int NI = 64 / RP;       // instructions per batch
  static constexpr int PITCH = U + 1;
  static constexpr int stage_elems = 64 * PITCH;
  // x: first element of the tile; rows: rows of the tile that exist; i: first column.
  // issue(): the global loads only -- v holds RAW vectors (instruction m, element k at
  // v[m * VN + k]: 16 bytes of row lane / QL + m * RP) and nothing waits for them;
  // untangle(): raw -> the lane's own row, through the staging area, when the values are needed.
  static __device__ __forceinline__ void issue(const T *x, uint32_t n, uint32_t rows, uint32_t i,
                                               uint32_t lane, T (&v)[U]) {
    const uint32_t q = lane % QL, r0 = lane / QL;
#pragma unroll
    for (int m = 0; m < NI; m++) {
      const uint32_t r = min(r0 + m * RP, rows - 1);
      const VA t = *reinterpret_cast<const VU *>(x + (size_t)r * n + i + q * VN);
#pragma unroll
      for (int k = 0; k < VN; k++) v[m * VN + k] = t[k];
    }
  }
  static __device__ __forceinline__ void untangle(uint32_t lane, T *st, T (&v)[U]) {
    const uint32_t q = lane % QL, r0 = lane / QL;
#pragma unroll
    for (int m = 0; m < NI; m++)
#pragma unroll
      for (int k = 0; k < VN; k++) st[(r0 + m * RP) * PITCH + q * VN + k] = v[m * VN + k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = st[lane * PITCH + u];
    __builtin_amdgcn_wave_barrier();
  }
  static __device__ __forceinline__ void store(T *x, uint32_t n, uint32_t rows, uint32_t i,
                                               uint32_t lane, T *st, const T (&v)[U]) {
    const uint32_t q = lane % QL, r0 = lane / QL;
#pragma unroll
    for (int u = 0; u < U; u++) st[lane * PITCH + u] = v[u];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < NI; m++) {
      const uint32_t r = r0 + m * RP;
      VA t;
#pragma unroll
      for (int k = 0; k < VN; k++) t[k] = st[r * PITCH + q * VN + k];
      if (r < rows) *reinterpret_cast<VU *>(x + (size_t)r * n + i + q * VN) = t;
    }
    __builtin_amdgcn_wave_barrier();
  }
};

// tt: [0,n) forward multiplier am[i]/bm[i]; [n,2n) am[i+1]; [2n,3n) bm[i+1]; FD: [3n,4n) RN(1/bm[i+1]).
// Pencil id p in [0, npencil): base = (p / n_inner) * outer_stride + (p % n_inner) * inner_stride,
// consecutive elements `stride` apart (CONTIG: stride == 1).
// KR = number of trailing full batches whose forward results stay in registers (the host
// guarantees n / U >= KR); the batches before them are parked in LDS, the first n_glob / U of
// them in global memory. PD = depth of the load pipeline in batches (PD divides KR): the loads of
// the register-resident batches and of the first PD others are ALL issued before the first chain
// starts, and a pipeline slot is refilled as soon as its chain is done, so the forward sweep pays
// the memory latency a couple of times instead of once per batch; the values to accumulate into
// (add_to) run through the same slots PD batches ahead of the backward sweep.
template <typename T, int U, int KR, int PD, bool CONTIG, bool FD>
__global__ void __launch_bounds__(64)
k_ipk_stream(uint32_t npencil, uint32_t n_inner, size_t outer_stride, size_t inner_stride,
             size_t stride, uint32_t n, uint32_t W, uint32_t n_glob, T *__restrict__ x,
             const T *__restrict__ tt, T *__restrict__ add_to, int sign) {
  static_assert(KR % PD == 0, "pipeline depth must divide the register-resident batches");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *sm = reinterpret_cast<T *>(smem_raw);
  using IO = PencilIO<T, U, CONTIG>;
  using TIO = TileIO<T, U>;
  const uint32_t lane = threadIdx.x;
  const uint32_t ls = min(lane, W - 1);  // lanes beyond the tile shadow its last pencil
  // workgroups go round-robin to the 8 XCDs (own L2 each): give every XCD a contiguous range of
  // tiles, so that the cache lines neighbouring tiles share are fetched by one L2
  const uint32_t per = (gridDim.x + 7) / 8;
  const uint32_t tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
  if ((uint64_t)tile * W >= npencil) return;  // grid is padded to a multiple of 8
  const uint32_t pid = tile * W + ls;
  const bool live = lane < W && pid < npencil;
  const uint32_t p = min(pid, npencil - 1);
  const uint32_t lo = (uint32_t)(((size_t)(p / n_inner) * outer_stride + (size_t)(p % n_inner) * inner_stride) * sizeof(T));
  T *xo = add_to ? add_to : x;
  T *sl = sm + ls;
  // CONTIG: the wave moves its batches together (TileIO); staging area behind the parked values
  const uint32_t rows = min(W, npencil - tile * W);
  T *st = sm + (size_t)W * ((n / U - KR) * U - n_glob);
  const size_t tile0 = (size_t)tile * W * n;  // first element of the tile (CONTIG: pencil p at p * n)
  // ld(): request a batch (CONTIG: raw, see TileIO); fix(): make it the lane's own values --
  // called where the batch is consumed, so that the loads stay in flight in between
  auto ld = [&](const T *a, uint32_t i, T(&v)[U]) {
    if constexpr (CONTIG) TIO::issue(a + tile0, n, rows, i, lane, v);
    else IO::load(a, lo, stride, i, v);
  };
  auto fix = [&](T(&v)[U]) {
    if constexpr (CONTIG) TIO::untangle(lane, st, v);
  };
  auto sto = [&](T *a, uint32_t i, const T(&v)[U]) {
    if constexpr (CONTIG) TIO::store(a + tile0, n, rows, i, lane, st, v);
    else if (live) IO::store(a, lo, stride, i, v);
  };
  const uint32_t nb = n / U, rem = n - nb * U;
  const uint32_t nbl = nb - KR;  // batches parked in LDS / global memory
  const T *am = tt + n, *bm = tt + 2 * n, *ym = tt + 3 * n;

  auto fwd_chain = [&](uint32_t i, T(&v)[U], T &prev) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      v[u] = v[u] - prev * tt[i + u];
      prev = v[u];
    }
  };
  // backward chain of the batch at i (values in v; o = what to accumulate into), then its store
  auto bwd_batch = [&](uint32_t i, T(&v)[U], const T(&o)[U], T &prev) {
    bool redo = !FD;
    if constexpr (FD) {
      T a[U];
      T pv = prev;
      bool bad = false;
#pragma unroll
      for (int u = U - 1; u >= 0; u--) {
        const T num = v[u] - am[i + u] * pv;
        bad |= fd_unsafe(num);
        a[u] = div_markstein<T>(num, bm[i + u], ym[i + u]);
        pv = a[u];
      }
      redo = __any(bad);
      if (!redo) {
        prev = pv;
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = a[u];
      }
    }
    if (redo) {
#pragma unroll
      for (int u = U - 1; u >= 0; u--) {
        v[u] = (v[u] - am[i + u] * prev) / bm[i + u];
        prev = v[u];
      }
    }
    if (add_to) {
      T r[U];
#pragma unroll
      for (int u = 0; u < U; u++) r[u] = sign > 0 ? o[u] + v[u] : o[u] - v[u];
      sto(xo, i, r);
    } else {
      sto(xo, i, v);
    }
  };

  // ---- forward: x[i] -= x[i-1] * w[i] -------------------------------------------------------
  T prev = 0;
  T R[U];         // remainder
  T buf[PD][U];   // pipeline slots
  T park[KR][U];  // the last KR full batches
#pragma unroll
  for (int j = 0; j < PD; j++)
    if ((uint32_t)j < nbl) ld(x, j * U, buf[j]);
#pragma unroll
  for (int j = 0; j < KR; j++) ld(x, (nbl + j) * U, park[j]);
#pragma unroll
  for (int u = 0; u < U - 1; u++)
    R[u] = (uint32_t)u < rem ? *reinterpret_cast<const T *>(reinterpret_cast<const char *>(x + (size_t)(nb * U + u) * stride) + lo) : (T)0;
  for (uint32_t g = 0; g < nbl; g += PD) {
#pragma unroll
    for (int j = 0; j < PD; j++) {
      const uint32_t b = g + j;
      if (b < nbl) {
        const uint32_t i = b * U;
        fix(buf[j]);
        fwd_chain(i, buf[j], prev);
        if (i < n_glob) {
          sto(x, i, buf[j]);
        } else {
#pragma unroll
          for (int u = 0; u < U; u++) sl[(i - n_glob + u) * W] = buf[j][u];
        }
        if (b + PD < nbl) ld(x, i + PD * U, buf[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < KR; j++) {
    fix(park[j]);
    fwd_chain((nbl + j) * U, park[j], prev);
  }
#pragma unroll
  for (int u = 0; u < U - 1; u++)
    if ((uint32_t)u < rem) {
      R[u] = R[u] - prev * tt[nb * U + u];
      prev = R[u];
    }

  // ---- backward: x[i] = (x[i] - am[i+1] x[i+1]) / bm[i+1] ----------------------------------
  // backward order t = 0 .. nb-1 is batch nb-1-t; its add_to values sit in slot t % PD
  if (add_to) {
#pragma unroll
    for (int t = 0; t < PD; t++) ld(xo, (nb - 1 - t) * U, buf[t]);  // nb >= KR >= PD
  }
  prev = 0;
  {
    T o[U];
    if (add_to) {
#pragma unroll
      for (int u = 0; u < U - 1; u++)
        o[u] = (uint32_t)u < rem ? *reinterpret_cast<const T *>(reinterpret_cast<const char *>(xo + (size_t)(nb * U + u) * stride) + lo) : (T)0;
    }
#pragma unroll
    for (int u = U - 2; u >= 0; u--)
      if ((uint32_t)u < rem) {
        const uint32_t k = nb * U + u;
        R[u] = (R[u] - am[k] * prev) / bm[k];
        prev = R[u];
      }
#pragma unroll
    for (int u = 0; u < U - 1; u++)
      if ((uint32_t)u < rem && live) {
        const T r = add_to ? (sign > 0 ? o[u] + R[u] : o[u] - R[u]) : R[u];
        *reinterpret_cast<T *>(reinterpret_cast<char *>(xo + (size_t)(nb * U + u) * stride) + lo) = r;
      }
  }
  auto load_parked = [&](uint32_t j, T(&dst)[U]) {
    if (j < n_glob) {
      ld(x, j, dst);
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) dst[u] = sl[(j - n_glob + u) * W];
    }
  };
  T v[U];
  // the first parked batch is in flight while the register-resident ones are solved
  if (nbl > 0) load_parked((nbl - 1) * U, v);
#pragma unroll
  for (int t = 0; t < KR; t++) {
    const uint32_t b = nb - 1 - t;
    if (add_to) fix(buf[t % PD]);
    bwd_batch(b * U, park[KR - 1 - t], buf[t % PD], prev);
    if (add_to && (uint32_t)(t + PD) < nb) ld(xo, (b - PD) * U, buf[t % PD]);
  }
  for (uint32_t g = 0; g < nbl; g += PD) {
#pragma unroll
    for (int j = 0; j < PD; j++) {
      const uint32_t t = KR + g + j;  // slot t % PD == j
      if (t < nb) {
        const uint32_t b = nb - 1 - t;
        T nx[U];
        if (b > 0) load_parked((b - 1) * U, nx);
        if (b * U < n_glob) fix(v);  // (this batch came back from global memory: raw)
        if (add_to) fix(buf[j]);
        bwd_batch(b * U, v, buf[j], prev);
        if (add_to && t + PD < nb) ld(xo, (b - PD) * U, buf[j]);
        if (b > 0) {
#pragma unroll
          for (int u = 0; u < U; u++) v[u] = nx[u];
        }
      }
    }
  }
}

} // namespace mgh
