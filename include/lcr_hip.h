/* lcr_hip.h — C ABI of liblcr_hip.so: the MI355X (gfx950) implementation of LCR-Net's per-scan hot path.
 *
 * Drop-in boundary.  The reference binds its native ops with pybind11 as module `utils.ext`
 * (utils/extensions/pybind.cpp:7-24) and calls them from experiments/lcrnet/modules/ops/{grid_subsample,radius_search}.py.  This header is
 * what a foreign-function binding (ctypes / cffi / cgo) would bind instead; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`; all memory is caller-owned;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous and stream-ordered, never
 *     synchronises the host and never allocates;
 *   - scratch memory comes from the caller: ask `*_ws_bytes`, pass `ws`/`ws_bytes`;
 *   - stacked ("stack mode") clouds: points f32[N,3] row-major, lengths i64[B] (reference layout,
 *     utils/extensions/cpu/grid_subsampling/grid_subsampling.cpp:20-30);
 *   - return value: 0 ok, -1 bad argument, -2 workspace/output too small, -3 HIP launch error
 *     (text via lcr_last_error()).  Data-dependent conditions are reported through the `status` words.
 */
#ifndef LCR_HIP_H
#define LCR_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LCR_OK 0
#define LCR_EARG (-1)
#define LCR_ESPACE (-2)
#define LCR_EHIP (-3)

/* bits of the device-side status word written by data-dependent stages */
#define LCR_STATUS_KEY_OVERFLOW 1u   /* voxel key needs more than 64 bits together with the cloud id */
#define LCR_STATUS_LEN_MISMATCH 2u   /* sum(lengths) exceeds the capacity passed by the host */

const char* lcr_last_error(void);
int lcr_version(void);

/* ------------------------------------------------------------------------------------------------
 * a-1  grid subsampling — replaces utils.ext.grid_subsampling
 *      (utils/extensions/cpu/grid_subsampling/grid_subsampling.cpp:5-62 → grid_subsampling_cpu.cpp:3-75).
 * Barycentre of every occupied voxel, per cloud; fp32 sums in INPUT order; output in libstdc++
 * std::unordered_map iteration order (bit-exact with the reference).
 *   xyz      f32[n_cap,3]   stacked input clouds (first sum(len) rows are used)
 *   len      i64[B]
 *   out_xyz  f32[n_cap,3]   first sum(out_len) rows are written
 *   out_len  i64[B]
 *   status   u32[1]         OR-ed LCR_STATUS_* bits (must be zeroed by the caller once)
 * ------------------------------------------------------------------------------------------------ */
int lcr_grid_subsample_ws_bytes(int64_t n_cap, int B, size_t* bytes);
int lcr_grid_subsample(const float* xyz, const int64_t* len, int B, int64_t n_cap, float voxel,
                       float* out_xyz, int64_t* out_len, uint32_t* status,
                       void* ws, size_t ws_bytes, void* stream);
/* Same, with a host-side promise that (voxel key bits + cloud id bits) <= key_bits_hint (0 = unknown, 64-bit safe):
 * bounds the number of radix passes launched.  A violated promise sets LCR_STATUS_KEY_OVERFLOW; retry with 0. */
int lcr_grid_subsample_ex(const float* xyz, const int64_t* len, int B, int64_t n_cap, float voxel, int key_bits_hint,
                          float* out_xyz, int64_t* out_len, uint32_t* status,
                          void* ws, size_t ws_bytes, void* stream);
/* HOST helper (no GPU): order[j] = insertion rank of the j-th element that libstdc++'s
 * std::unordered_map<size_t,...> visits after inserting the n distinct keys in the given order — the serial mirror of
 * the device kernel that fixes lcr_grid_subsample's output order (grid_subsampling_cpu.cpp:26,45-47). */
int lcr_hashmap_order_host(const uint64_t* keys_host, int64_t n, int64_t* order_host);

/* ------------------------------------------------------------------------------------------------
 * a-2  radius search — replaces utils.ext.radius_neighbors + the [:, :limit] slice of
 *      modules/ops/radius_search.py:7-27 (radius_neighbors.cpp:5-68 → radius_neighbors_cpu.cpp:3-91).
 * For every query: supports of the SAME cloud with d2 < radius*radius (fp32, d2 = ((dx*dx)+dy*dy)+dz*dz,
 * no FMA), ascending by (d2, index), global support index, padded with sum(slen).
 *   q f32[nq_cap,3], s f32[ns_cap,3], qlen/slen i64[B]
 *   limit > 0 : out_idx64 / out_idx32 are [nq_cap, limit] (either may be NULL)
 *   limit == 0: count only (out_cnt required)
 *   out_cnt   i32[nq_cap] uncapped in-radius count per query (NULL allowed when limit > 0)
 * lcr_support_grid_build + lcr_radius_query split the call so one grid serves several query sets.
 * ------------------------------------------------------------------------------------------------ */
int lcr_support_grid_ws_bytes(int64_t ns_cap, int B, size_t* bytes);
int lcr_support_grid_build(const float* s, const int64_t* slen, int B, int64_t ns_cap, float radius,
                           uint32_t* status, void* grid_ws, size_t grid_ws_bytes, void* stream);
int lcr_radius_query(const float* q, const int64_t* qlen, int B, int64_t nq_cap,
                     const void* grid_ws, int64_t ns_cap /* as passed to the build */, float radius, int limit,
                     int64_t* out_idx64, int32_t* out_idx32, int32_t* out_cnt, void* stream);
int lcr_radius_search_ws_bytes(int64_t nq_cap, int64_t ns_cap, int B, size_t* bytes);
int lcr_radius_search(const float* q, const float* s, const int64_t* qlen, const int64_t* slen, int B,
                      int64_t nq_cap, int64_t ns_cap, float radius, int limit,
                      int64_t* out_idx64, int32_t* out_idx32, int32_t* out_cnt, uint32_t* status,
                      void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LCR_HIP_H */
