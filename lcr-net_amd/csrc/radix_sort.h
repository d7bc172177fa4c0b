// radix_sort.h — stable LSD radix sort of (u64 key, u32 value) pairs, 11-bit digits, device-side element count and
// device-side pass count (passes >= *num_passes exit immediately, so the host can launch a fixed upper bound without
// synchronising).  Result lands in buffer (num_passes & 1): 0 -> A, 1 -> B.
//
// Per pass: per-tile 2048-bin histograms -> device-wide scan (digit-major) -> stable scatter.  Stability inside a tile:
// every wavefront owns a contiguous slice of the tile and ranks its keys in order with ballot-based digit matching;
// slices are then offset by the per-digit totals of the earlier wavefronts.
#pragma once
#include "common.h"

namespace lcr {

// 11-bit digits: the voxel keys of a KITTI-sized batch are 19-28 bits wide, i.e. 2-3 passes instead of 3-4 with 8-bit digits.
// Every pass is three dependent launches in a latency-bound chain, which matters more than the larger histograms.
constexpr int RX_BITS = 11;
constexpr int RX_D = 1 << RX_BITS;        // digits per pass
constexpr int RX_T = 256;                 // threads per tile
constexpr int RX_I = 8;                   // keys per thread
constexpr int RX_TILE = RX_T * RX_I;      // 2048 keys per tile
constexpr int RX_WSLICE = 64 * RX_I;      // keys per wavefront slice

struct RadixCtl {
  int64_t n;           // elements
  int     num_passes;  // 0..radix_passes(64)
};

inline size_t radix_hist_elems(int64_t n_cap) { return static_cast<size_t>((n_cap + RX_TILE - 1) / RX_TILE) * RX_D + 1; }
__host__ __device__ inline int radix_passes(int key_bits) { return (key_bits + RX_BITS - 1) / RX_BITS; }

int radix_sort_pairs(const RadixCtl* ctl, uint64_t* keysA, uint64_t* keysB, uint32_t* valsA, uint32_t* valsB, int64_t n_cap,
                     int max_passes, int32_t* hist /*radix_hist_elems*/, void* scan_ws, hipStream_t st);

}  // namespace lcr
