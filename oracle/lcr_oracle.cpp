// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of the two native point-cloud ops on LCR-Net's per-scan hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
//
// Follows (semantics, not text):
//   grid subsampling : /root/reference/utils/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:3-75
//                      /root/reference/utils/extensions/cpu/grid_subsampling/grid_subsampling_cpu.h:7-21
//   radius search    : /root/reference/utils/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91
//                      distance / strict '<' / ascending order: extra/nanoflann/nanoflann.hpp:432-440, 249-253, 1280-1289
//                      canonical (d2, idx) tie order: /root/reference/cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:125-208
//
// Parity pin: tests/test_oracle_vs_reference.py compares this file with oracle/_ref/libref_ops.so
// (the reference's own C++ compiled from /root/reference) on the six demo scans, and
// tests/test_oracle_golden.py with the committed golden digests generated from that same library.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off; NO -ffast-math, NO -march=native).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace {

struct VoxelAcc {
  int   count = 0;
  float x = 0.f, y = 0.f, z = 0.f;
};

// float -> size_t the way x86-64 gcc lowers it for in-range values (negative values wrap
// through int64); only reachable when fp32 rounding puts the origin one ulp above the minimum.
inline uint64_t f2u64(float v) { return static_cast<uint64_t>(static_cast<int64_t>(v)); }

// One cloud.  Emits barycentres in std::unordered_map iteration order (that IS the reference's
// output order: grid_subsampling_cpu.cpp:26,45-47).
size_t subsample_one(const float* p, size_t n, float voxel, float* out) {
  if (n == 0) return 0;
  float mn[3] = {p[0], p[1], p[2]}, mx[3] = {p[0], p[1], p[2]};
  for (size_t i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) {
      float v = p[3 * i + d];
      if (v < mn[d]) mn[d] = v;
      if (v > mx[d]) mx[d] = v;
    }
  // origin = floor(min * float(1.0/voxel)) * voxel          (grid_subsampling_cpu.cpp:11)
  const float inv = static_cast<float>(1.0 / static_cast<double>(voxel));
  float org[3];
  for (int d = 0; d < 3; ++d) org[d] = std::floor(mn[d] * inv) * voxel;
  const uint64_t NX = f2u64(std::floor((mx[0] - org[0]) / voxel) + 1.f);   // :13-16
  const uint64_t NY = f2u64(std::floor((mx[1] - org[1]) / voxel) + 1.f);   // :17-20

  std::unordered_map<size_t, VoxelAcc> cells;
  for (size_t i = 0; i < n; ++i) {
    const uint64_t ix = f2u64(std::floor((p[3 * i + 0] - org[0]) / voxel));  // true fp32 division (:32-34)
    const uint64_t iy = f2u64(std::floor((p[3 * i + 1] - org[1]) / voxel));
    const uint64_t iz = f2u64(std::floor((p[3 * i + 2] - org[2]) / voxel));
    const uint64_t key = ix + NX * iy + NX * NY * iz;                        // :35
    VoxelAcc& a = cells[key];      // insert-if-absent then accumulate in INPUT order (.h:17-20)
    a.count += 1;
    a.x += p[3 * i + 0];
    a.y += p[3 * i + 1];
    a.z += p[3 * i + 2];
  }
  size_t m = 0;
  for (auto& kv : cells) {
    const float r = static_cast<float>(1.0 / static_cast<double>(kv.second.count));  // :46
    out[3 * m + 0] = kv.second.x * r;
    out[3 * m + 1] = kv.second.y * r;
    out[3 * m + 2] = kv.second.z * r;
    ++m;
  }
  return m;
}

struct Cand {
  float   d2;
  int64_t idx;
};

}  // namespace

extern "C" {

// Stacked clouds; out_xyz must hold sum(len)*3 floats.  Returns total number of output points.
int64_t oracle_grid_subsample(const float* xyz, const int64_t* len, int B, float voxel,
                              float* out_xyz, int64_t* out_len) {
  int64_t in_off = 0, out_off = 0;
  for (int b = 0; b < B; ++b) {
    size_t m = subsample_one(xyz + 3 * in_off, static_cast<size_t>(len[b]), voxel, out_xyz + 3 * out_off);
    out_len[b] = static_cast<int64_t>(m);
    in_off += len[b];
    out_off += static_cast<int64_t>(m);
  }
  return out_off;
}

// Iteration order of std::unordered_map<size_t,...> after inserting n DISTINCT keys in the given order:
// order[j] = insertion rank (0-based) of the j-th element visited.  This is the libstdc++ behaviour the
// reference's output order depends on (grid_subsampling_cpu.cpp:26,45-47); used to unit-test the product's
// explicit emulation of it.
int64_t oracle_hashmap_order(const uint64_t* keys, int64_t n, int64_t* order) {
  std::unordered_map<size_t, int64_t> m;
  for (int64_t i = 0; i < n; ++i) m.emplace(static_cast<size_t>(keys[i]), i);
  int64_t j = 0;
  for (auto& kv : m) order[j++] = kv.second;
  return j;
}

// Per-row in-radius counts (uncapped).  Returns the maximum count (= the reference's output width).
// d2 = ((0 + dx*dx) + dy*dy) + dz*dz in fp32, strict d2 < r*r, same-cloud supports only.
int64_t oracle_radius_count(const float* q, const float* s, const int64_t* qlen, const int64_t* slen,
                            int B, float radius, int32_t* counts) {
  const float r2 = radius * radius;
  int64_t q0 = 0, s0 = 0, mx = 0;
  for (int b = 0; b < B; ++b) {
    for (int64_t i = q0; i < q0 + qlen[b]; ++i) {
      int32_t c = 0;
      for (int64_t j = s0; j < s0 + slen[b]; ++j) {
        float dx = q[3 * i] - s[3 * j], dy = q[3 * i + 1] - s[3 * j + 1], dz = q[3 * i + 2] - s[3 * j + 2];
        float d2 = 0.f;
        d2 += dx * dx;
        d2 += dy * dy;
        d2 += dz * dz;
        if (d2 < r2) ++c;
      }
      counts[i] = c;
      if (c > mx) mx = c;
    }
    q0 += qlen[b];
    s0 += slen[b];
  }
  return mx;
}

// out[Nq, width]: ascending (d2, idx); padded with Ns_total.  If limit > 0 width must equal limit
// (== reference output sliced [:, :limit], modules/ops/radius_search.py:25-26); if limit <= 0 width
// must be >= the maximum count (reference returns exactly width == max count).
// A uniform grid (cell = radius) only prunes candidates; membership and order are decided by the
// fp32 d2 above, so the result equals the brute-force definition.
int oracle_radius_search(const float* q, const float* s, const int64_t* qlen, const int64_t* slen,
                         int B, float radius, int64_t width, int64_t* out, int32_t* counts /*nullable*/) {
  const float r2 = radius * radius;
  int64_t Ns_total = 0;
  for (int b = 0; b < B; ++b) Ns_total += slen[b];
  int64_t q0 = 0, s0 = 0;
  std::vector<Cand> cand;
  for (int b = 0; b < B; ++b) {
    const int64_t ns = slen[b], nq = qlen[b];
    // grid over the support cloud
    float mn[3] = {0, 0, 0};
    if (ns > 0) {
      for (int d = 0; d < 3; ++d) mn[d] = s[3 * s0 + d];
      for (int64_t j = s0; j < s0 + ns; ++j)
        for (int d = 0; d < 3; ++d) mn[d] = std::min(mn[d], s[3 * j + d]);
    }
    const double cell = std::max(static_cast<double>(radius) * 1.000001, 1e-9);  // >= radius: 27 cells cover the ball
    std::unordered_map<uint64_t, std::vector<int64_t>> grid;
    auto cidx = [&](const float* p, int d) { return static_cast<int64_t>(std::floor((static_cast<double>(p[d]) - mn[d]) / cell)); };
    auto ckey = [](int64_t x, int64_t y, int64_t z) {
      return (static_cast<uint64_t>(x + (1 << 20)) << 42) ^ (static_cast<uint64_t>(y + (1 << 20)) << 21) ^ static_cast<uint64_t>(z + (1 << 20));
    };
    for (int64_t j = s0; j < s0 + ns; ++j) grid[ckey(cidx(s + 3 * j, 0), cidx(s + 3 * j, 1), cidx(s + 3 * j, 2))].push_back(j);
    for (int64_t i = q0; i < q0 + nq; ++i) {
      cand.clear();
      const int64_t cx = cidx(q + 3 * i, 0), cy = cidx(q + 3 * i, 1), cz = cidx(q + 3 * i, 2);
      for (int64_t x = cx - 1; x <= cx + 1; ++x)
        for (int64_t y = cy - 1; y <= cy + 1; ++y)
          for (int64_t z = cz - 1; z <= cz + 1; ++z) {
            auto it = grid.find(ckey(x, y, z));
            if (it == grid.end()) continue;
            for (int64_t j : it->second) {
              float dx = q[3 * i] - s[3 * j], dy = q[3 * i + 1] - s[3 * j + 1], dz = q[3 * i + 2] - s[3 * j + 2];
              float d2 = 0.f;
              d2 += dx * dx;
              d2 += dy * dy;
              d2 += dz * dz;
              if (d2 < r2) cand.push_back({d2, j});
            }
          }
      std::sort(cand.begin(), cand.end(), [](const Cand& a, const Cand& b) { return a.d2 < b.d2 || (a.d2 == b.d2 && a.idx < b.idx); });
      if (counts) counts[i] = static_cast<int32_t>(cand.size());
      for (int64_t c = 0; c < width; ++c) out[i * width + c] = c < static_cast<int64_t>(cand.size()) ? cand[c].idx : Ns_total;
    }
    q0 += nq;
    s0 += ns;
  }
  return 0;
}

}  // extern "C"
