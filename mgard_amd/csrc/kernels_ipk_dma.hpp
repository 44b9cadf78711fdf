// Streaming Thomas solves with an LDS-DMA front end (gfx950), strided pencils.
//
// rocprofv3 SQ counters on k_ipk_stream at the top level of 512^3 (profiles/r04a_sq_counters_chain.json):
// a solver wave spends 48 % of its life parked at s_waitcnt and only 14 % stalled on the dependent
// chain -- with ~one wave per SIMD nothing covers its memory waits, and the forward substitution
// (2 dependent operations per element, 77 ns per batch of 16) outruns a load pipeline that is one
// batch deep: every LDS-parked batch pays a full memory round trip. A deeper register pipeline
// costs the second wave per SIMD (230 -> 290 VGPRs), which a level of 1033 tiles on 1024 SIMDs
// cannot afford.
//
// Here the part of a pencil that is parked in LDS anyway is brought there by LDS-DMA
// (global_load_lds_dword: no destination registers, the row of 64 neighbouring pencils lands at
// sm[i * 64 + lane], exactly the parking layout), ALL of it requested before anything else, the
// register-resident tail of the pencil right behind it: the whole tile is in flight from the
// first microsecond and the forward sweep runs at the rate the memory system delivers. The
// forward results overwrite the raw values in place (LDS) / stay in registers; the backward sweep
// is k_ipk_stream's.
//
// Ordering of the DMA data: the DMAs are issued by inline asm (hipcc does not count them) and
// are followed by KR * U >= 64 ordinary loads; VMEM operations of a wave retire in issue order, so
// `s_waitcnt vmcnt(63)` after the last of those -- at most 63 operations outstanding, all of them
// younger than every DMA -- means every DMA has landed. One wave per workgroup: no barrier.
//
// Arithmetic and order: tridiag_forward2 / tridiag_backward2 (reference
// include/mgard-x/DataRefactoring/MultiDimension/Correction/IPKFunctor.h:127,147, indexing of
// IterativeProcessingKernel3D.hpp:108-124,223-262); bit-identical to kernels_v1.hpp:k_ipk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_ipk.hpp"

namespace mgh {

// Developer build -DMGH_IPK_DBG: shader-clock time stamps of the phases of a few tiles
#ifdef MGH_IPK_DBG
__device__ unsigned long long g_ipk_dbg[8][8];
#define MGH_IPK_T(k) do { if (lane == 0 && (tile % 128) == 5 && tile / 128 < 8) g_ipk_dbg[tile / 128][k] = wall_clock64(); } while (0)
#else
#define MGH_IPK_T(k)
#endif

// One row of the tile: 64 lanes x 4 bytes from (uniform) row + (per-lane) byte offset voff to the
// LDS bytes [lds_addr, lds_addr + 256) of this workgroup. M0 is compiler-reserved: written,
// used and restored inside ONE statement (cdna_hip_programming.md, inline-asm rules).
__device__ __forceinline__ void glds_row_dword(const void *row, uint32_t voff, uint32_t lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(row), "s"(lds_addr)
               : "memory");
}

// Four consecutive rows of the tile in one instruction: lane l moves the 16 bytes at
// row0 + voff (voff = (l / 16) rows + (l % 16) * 16 bytes, built by the caller) to the LDS bytes
// lds_addr + l * 16, i.e. rows i .. i + 3 land back to back in the parking layout. The rows start
// at arbitrary elements: 4-byte alignment only.
__device__ __forceinline__ void glds_rows_x4(const void *row0, uint32_t voff, uint32_t lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(row0), "s"(lds_addr)
               : "memory");
}

// U consecutive entries of a wave-uniform table as ONE scalar load (s_load_dwordx16 for 16 floats).
// Left to itself hipcc loads the entries of an unrolled batch one s_load_dword at a time, each
// followed by s_waitcnt lgkmcnt(0) right in front of its use: two exposed scalar-cache round trips
// per element of the chain (measured: 1.9 us instead of 0.5 us per batch of 16).
template <typename T, int U> struct TableVec { T v[U]; };
template <typename T, int U>
__device__ __forceinline__ TableVec<T, U> table_vec(const T *t) {
  return *reinterpret_cast<const TableVec<T, U> *>(t);
}

// tt: [0,n) forward multiplier am[i]/bm[i]; [n,2n) am[i+1]; [2n,3n) bm[i+1].
// Pencil id p in [0, npencil): base = (p / n_inner) * outer_stride + (p % n_inner) * inner_stride,
// consecutive elements `stride` apart. A tile is 64 consecutive pencil ids (the last tile shadows
// its last pencil). Elements [0, nl) of every pencil are parked in LDS (nl = n - KR * U, any
// value >= 0: the leading nl % U elements are walked one by one), the last KR * U in registers.
// Dynamic LDS: nl * 64 * sizeof(T) bytes.
// ADD: 0 = the result replaces x; +1 / -1 = it is added to / subtracted from add_to (AddND /
// SubtractND fused; compile-time: the unrolled chains exist once, ~30 KB of code instead of 55).
template <typename T, int U, int KR, int ADD, bool X4 = true>
__global__ void __launch_bounds__(64, 2)  // two waves per SIMD: 1033 tiles do not fit 1024 SIMDs
k_ipk_dma(uint32_t npencil, uint32_t n_inner, size_t outer_stride, size_t inner_stride, size_t stride,
          uint32_t n, T *__restrict__ x, const T *__restrict__ tt, T *__restrict__ add_to_arg) {
  constexpr bool add_to = ADD != 0;
  constexpr int sign = ADD;
  static_assert(sizeof(T) == 4, "global_load_lds_dword moves 4 bytes per lane");
  static_assert(KR * U >= 64, "the vmcnt(63) argument needs 64 ordinary loads behind the DMAs");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *sm = reinterpret_cast<T *>(smem_raw);
  const uint32_t lane = threadIdx.x;
  // tiles in contiguous ranges per XCD (workgroups go round-robin to the 8 XCDs)
  const uint32_t per = (gridDim.x + 7) / 8;
  const uint32_t tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
  if ((uint64_t)tile * 64 >= npencil) return;  // grid is padded to a multiple of 8
  const uint32_t pid = tile * 64 + lane;
  const bool live = pid < npencil;
  const uint32_t p = min(pid, npencil - 1);
  const uint32_t lo = (uint32_t)(((size_t)(p / n_inner) * outer_stride + (size_t)(p % n_inner) * inner_stride) * sizeof(T));
  T *xo = add_to ? add_to_arg : x;
  T *sl = sm + lane;
  const uint32_t nl = n - KR * U;     // rows parked in LDS
  const uint32_t nhead = nl % U;      // walked one by one
  const T *am = tt + n, *bm = tt + 2 * n;
  auto gld = [&](const T *a, uint32_t i, T(&v)[U]) {
    const T *r = a + (size_t)i * stride;
#pragma unroll
    for (int u = 0; u < U; u++) {
      v[u] = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(r) + lo);
      r += stride;
    }
  };
  auto gst = [&](T *a, uint32_t i, const T(&v)[U]) {
#ifdef MGH_IPK_NOSTORE
    if (v[0] != (T)123.456f) return;
#endif
    if (!live) return;
    // (laundered: stores into the array the loads came from would otherwise keep all KR * U row
    // addresses of the prologue alive -- 2 SGPRs each, spilled to VGPR lanes, then to scratch)
    asm volatile("" : "+s"(a));
    T *r = a + (size_t)i * stride;
#pragma unroll
    for (int u = 0; u < U; u++) {
      *reinterpret_cast<T *>(reinterpret_cast<char *>(r) + lo) = v[u];
      r += stride;
    }
  };

  MGH_IPK_T(0);
  // ---- everything in flight: DMA of the LDS-parked rows, then the register-resident tail ----
  {
    const uint32_t lds0 = (uint32_t)(uintptr_t)sm;
    // a full tile whose 64 pencils are neighbours in memory moves four rows per instruction
    const uint32_t p0 = tile * 64;
    const bool x4 = X4 && p0 + 63 < npencil && p0 / n_inner == (p0 + 63) / n_inner && inner_stride == 1 && nl >= 4;
    if (x4) {
      const uint32_t lo0 = (uint32_t)(((size_t)(p0 / n_inner) * outer_stride + (size_t)(p0 % n_inner)) * sizeof(T));
      const uint32_t vo = lo0 + (lane / 16) * (uint32_t)(stride * sizeof(T)) + (lane % 16) * 16;
      const T *r = x;
      uint32_t i = 0;
      for (; i + 4 <= nl; i += 4) {
        glds_rows_x4(r, vo, lds0 + i * 64 * (uint32_t)sizeof(T));
        r += 4 * stride;
      }
      if (i < nl)  // the last 1..3 rows: once more the four rows that end the parked part
        glds_rows_x4(x + (size_t)(nl - 4) * stride, vo, lds0 + (nl - 4) * 64 * (uint32_t)sizeof(T));
    } else {
      const T *r = x;
      for (uint32_t i = 0; i < nl; i++) {
        glds_row_dword(r, lo, lds0 + i * 64 * (uint32_t)sizeof(T));
        r += stride;
      }
    }
  }
  MGH_IPK_T(1);
  T park[KR][U];
#pragma unroll
  for (int j = 0; j < KR; j++) gld(x, nl + j * U, park[j]);
  // the values to accumulate into travel one batch ahead of the backward sweep, the first batch
  // from the very start (two batches ahead measured the same: with AddND the backward sweep moves
  // 136 MB at ~6 TB/s -- it is bound by the memory system, not by the latency of these loads)
  T o[U];
  if (add_to) gld(xo, n - U, o);
  MGH_IPK_T(2);
  asm volatile("s_waitcnt vmcnt(63)" ::: "memory");  // every DMA has landed (see the header)
  MGH_IPK_T(3);

  // ---- forward: x[i] -= x[i-1] * w[i] -------------------------------------------------------
  T prev = 0;
  for (uint32_t i = 0; i < nhead; i++) {
    T a = sl[i * 64];
    a = a - prev * tt[i];
    sl[i * 64] = a;
    prev = a;
  }
  for (uint32_t i = nhead; i < nl; i += U) {
    T v[U];
    const TableVec<T, U> w = table_vec<T, U>(tt + i);
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = sl[(i + u) * 64];
#pragma unroll
    for (int u = 0; u < U; u++) {
      v[u] = v[u] - prev * w.v[u];
      prev = v[u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) sl[(i + u) * 64] = v[u];
  }
  MGH_IPK_T(4);
#pragma unroll
  for (int j = 0; j < KR; j++) {
    const uint32_t i = nl + j * U;
    __builtin_amdgcn_sched_barrier(0);  // (keeps the scheduler from piling up table values)
    const TableVec<T, U> w = table_vec<T, U>(tt + i);
#pragma unroll
    for (int u = 0; u < U; u++) {
      park[j][u] = park[j][u] - prev * w.v[u];
      prev = park[j][u];
    }
  }

  // ---- backward: x[i] = (x[i] - am[i+1] x[i+1]) / bm[i+1]; each batch stored (+/- add_to) as
  // soon as its chain is done; the values to accumulate into travel one batch ahead ----------
  MGH_IPK_T(5);
  prev = 0;
  // (requesting the tables of a batch while the batch before it is solved -- 64 scalar registers
  // live -- measured slower: 9.7 vs 8.1 us for the register-resident batches)
  auto bwd = [&](uint32_t i, T(&v)[U], T &pv) {
    // one batch = one scheduling region: the quotients' denominators are wave-uniform, and a
    // scheduler that is free to start all 144 divisions early runs out of registers
    __builtin_amdgcn_sched_barrier(0);
    const TableVec<T, U> a = table_vec<T, U>(am + i), b = table_vec<T, U>(bm + i);
#pragma unroll
    for (int u = U - 1; u >= 0; u--) {
#ifdef MGH_IPK_NODIV
      v[u] = (v[u] - a.v[u] * pv) * b.v[u];
#else
      v[u] = (v[u] - a.v[u] * pv) / b.v[u];
#endif
      pv = v[u];
    }
  };
  auto out = [&](uint32_t i, const T(&v)[U], const T(&ov)[U]) {
    if (add_to) {
      T r[U];
#pragma unroll
      for (int u = 0; u < U; u++) r[u] = sign > 0 ? ov[u] + v[u] : ov[u] - v[u];
      gst(xo, i, r);
    } else {
      gst(xo, i, v);
    }
  };
  T v[U];
  const uint32_t nbl = (nl - nhead) / U;  // full batches in LDS
  if (nbl > 0) {
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = sl[(nl - U + u) * 64];
  }
#pragma unroll
  for (int t = 0; t < KR; t++) {
    const uint32_t i = nl + (KR - 1 - t) * U;
    T on[U];
    if (add_to && (t + 1 < KR || nbl > 0)) gld(xo, i - U, on);
    bwd(i, park[KR - 1 - t], prev);
    out(i, park[KR - 1 - t], o);
    if (add_to) {
#pragma unroll
      for (int u = 0; u < U; u++) o[u] = on[u];
    }
  }
  MGH_IPK_T(6);
  for (uint32_t b = nbl; b-- > 0;) {
    const uint32_t i = nhead + b * U;
    T nx[U], on[U];
    if (b > 0) {
#pragma unroll
      for (int u = 0; u < U; u++) nx[u] = sl[(i - U + u) * 64];
    }
    if (add_to && b > 0) gld(xo, i - U, on);
    bwd(i, v, prev);
    out(i, v, o);
    if (b > 0) {
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = nx[u];
      if (add_to) {
#pragma unroll
        for (int u = 0; u < U; u++) o[u] = on[u];
      }
    }
  }
  MGH_IPK_T(7);
  for (uint32_t i = nhead; i-- > 0;) {
    T a = sl[i * 64];
    a = (a - am[i] * prev) / bm[i];
    prev = a;
    if (live) {
      T *q = reinterpret_cast<T *>(reinterpret_cast<char *>(xo + (size_t)i * stride) + lo);
      *q = add_to ? (sign > 0 ? *q + a : *q - a) : a;
    }
  }
}

} // namespace mgh
