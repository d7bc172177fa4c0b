// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// C-ABI driver around the REFERENCE's own C++ (compiled from where it lies under /root/reference by
// oracle/Makefile; objects and the .so go to oracle/_ref/, which is git-ignored).  No reference source
// is copied: this file only declares the two entry points the reference exports in
//   utils/extensions/cpu/grid_subsampling/grid_subsampling_cpu.h:23-37
//   utils/extensions/cpu/radius_neighbors/radius_neighbors_cpu.h:9-16
// via their own headers, and marshals flat arrays into the std::vector arguments they take — the same
// marshalling the torch bindings do in grid_subsampling.cpp:20-59 / radius_neighbors.cpp:29-65.
// The torch/pybind binding files themselves are NOT built (they need <ATen/cuda/CUDAContext.h>, absent on ROCm).
#include <cstdint>
#include <cstring>
#include <vector>

#include "cpu/grid_subsampling/grid_subsampling_cpu.h"
#include "cpu/radius_neighbors/radius_neighbors_cpu.h"

extern "C" {

int64_t ref_grid_subsample(const float* xyz, const int64_t* len, int B, float voxel, float* out_xyz, int64_t* out_len) {
  int64_t n = 0;
  for (int b = 0; b < B; ++b) n += len[b];
  std::vector<PointXYZ> pts(reinterpret_cast<const PointXYZ*>(xyz), reinterpret_cast<const PointXYZ*>(xyz) + n);
  std::vector<long> lens(len, len + B), s_lens;
  std::vector<PointXYZ> s_pts;
  grid_subsampling_cpu(pts, s_pts, lens, s_lens, voxel);
  std::memcpy(out_xyz, s_pts.data(), sizeof(float) * 3 * s_pts.size());
  for (int b = 0; b < B; ++b) out_len[b] = s_lens[b];
  return static_cast<int64_t>(s_pts.size());
}

// Two-call protocol: call with out == nullptr to learn the width (max count), then with a buffer of Nq*width.
int64_t ref_radius_neighbors(const float* q, const float* s, const int64_t* qlen, const int64_t* slen, int B,
                             float radius, int64_t* out, int64_t out_width) {
  static thread_local std::vector<long> cache;
  static thread_local int64_t cache_w = 0;
  int64_t nq = 0, ns = 0;
  for (int b = 0; b < B; ++b) { nq += qlen[b]; ns += slen[b]; }
  if (out == nullptr) {
    std::vector<PointXYZ> qp(reinterpret_cast<const PointXYZ*>(q), reinterpret_cast<const PointXYZ*>(q) + nq);
    std::vector<PointXYZ> sp(reinterpret_cast<const PointXYZ*>(s), reinterpret_cast<const PointXYZ*>(s) + ns);
    std::vector<long> ql(qlen, qlen + B), sl(slen, slen + B);
    cache.clear();
    radius_neighbors_cpu(qp, sp, ql, sl, cache, radius);
    cache_w = nq > 0 ? static_cast<int64_t>(cache.size()) / nq : 0;
    return cache_w;
  }
  if (out_width != cache_w) return -1;
  std::memcpy(out, cache.data(), sizeof(long) * cache.size());
  return cache_w;
}

}  // extern "C"
