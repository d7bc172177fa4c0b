/* lcr_hip.h — C ABI of liblcr_hip.so: the MI355X (gfx950) implementation of LCR-Net's per-scan hot path.
 *
 * Drop-in boundary.  The reference binds its native ops with pybind11 as module `utils.ext`
 * (utils/extensions/pybind.cpp:7-24) and calls them from experiments/lcrnet/modules/ops/{grid_subsample,radius_search}.py.  This header is
 * what a foreign-function binding (ctypes / cffi / cgo) would bind instead; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`; all memory is caller-owned;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous and stream-ordered, does not
 *     synchronise the host and does not allocate — with these stated exceptions:
 *       lcr_precompute_batch[_rows] ends with ONE synchronisation of `stream` (it returns the per-stage
 *       lengths of the batch on the host) and owns a small per-process pool of side streams, events and
 *       pinned staging words, created on first use and kept until process exit;
 *       lcr_ktimer_read synchronises on the events it logged (measurement harness only);
 *       the opt-in stream-K GEMM keeps one library-owned device scratch buffer per (device, stream), allocated on first use;
 *   - all other scratch memory comes from the caller: ask `*_ws_bytes`, pass `ws`/`ws_bytes`;
 *   - stacked ("stack mode") clouds: points f32[N,3] row-major, lengths i64[B] (reference layout,
 *     utils/extensions/cpu/grid_subsampling/grid_subsampling.cpp:20-30);
 *   - return value: 0 ok, -1 bad argument, -2 workspace/output too small, -3 HIP launch error
 *     (text via lcr_last_error()).  Data-dependent conditions are reported through the `status` words.
 *
 * Shape domain (what the kernels are built for = every shape of the shipped configuration, experiments/lcrnet/config_model.py:33-43 with
 * best-model-mixed.tar, and of BASELINE.json's configs; a call outside it returns LCR_EARG with a message, it never computes something else):
 *   - clouds per call (B): 1..64 for the support grids / radius searches / lcr_precompute_batch (GRID_MAX_B); batches of more scans are
 *     split by the caller (lcr-net_amd/pipeline.py feeds 8 per call, PairPipeline at most 32 pairs = 64 clouds);
 *   - neighbour columns per row (`limit`, H): 1..128 for the KPConv aggregation / C_in = 1 KPConv / max-pool (KP_HMAX; the reference's
 *     calibrated limits are 64..80); the radius search itself accepts any limit >= 1 (balls with more than 512 hits take an exact
 *     storage-free path) and limit <= 0 = count-only;
 *   - KPConv feature width (C_in = C_out = mid channels of a ResidualBlock): 32, 64, 128 or 256 (init_dim 64 -> 32..256), 15 kernel points;
 *     C_in = 1 (encoder1_1) has its own fused kernel with any C_out <= 256;
 *   - GroupNorm: channels % 4 == 0 and % groups == 0 (32 groups in the reference); segment tables (per scan or per pair) of S <= 256
 *     segments with S * groups <= 2048;
 *   - lcr_encoder_forward: exactly the KPEncoder of backbone4.py:11-89 — 4 stages, ConvBlock + 10 ResidualBlocks in that order, any
 *     init_dim whose block widths fall in the KPConv domain above.  Other depths run block by block through the same building-block
 *     entries (lcr-net_amd/modules/kpconv/modules.py does that when `native_encoder.eligible` says no);
 *   - GEMMs: any M; N, K >= 1 (K % 4 == 0 for the vectorised forms, else the generic tile); the split-bf16 form needs K >= 288, K % 32 == 0, N >= 64;
 *   - attention: head dim 32 (d_model 128 / 4 heads in the reference), <= 64 stacked problems per segmented call; dense form any key count,
 *     top-k form <= 4096 keys per cloud (the scores of a row live in LDS: 70 KB of the 160 KB of a gfx950 CU — this library targets gfx950 only);
 *   - retrieval: descriptor width % 4 == 0 (256 in the reference), k <= 128 (50 in the reference), any corpus size that fits HBM;
 *   - pose tail: <= 4000 coarse nodes per cloud, patches of <= 128 points (+ dustbin: 129 x 129 transport problems; point_limit 128 in the
 *     reference), 3 x 3 weighted Procrustes.
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
#define LCR_STATUS_LEN_MISMATCH 2u
#define LCR_GN_REPLICAS 8  /* copies of every GroupNorm statistics table (see lcr_gemm_f32) */   /* sum(lengths) exceeds the capacity passed by the host */

const char* lcr_last_error(void);
/* Opt-in launch timing for measurement harnesses (bench.py's roofline leg): while enabled, lcr_gemm_f32 (kind 0, meta = M,N,K)
 * and lcr_kpconv_aggregate (kind 1, meta = M,Ns,H,C,index bytes) bracket their launch with HIP events on the launch stream —
 * also when they are called from lcr_encoder_forward.  lcr_ktimer_enable(1) clears the log; lcr_ktimer_read synchronises on
 * the logged events and returns the number of records of `kind`. */
void lcr_ktimer_enable(int on);
void lcr_ktimer_sample(int every);   /* time every n-th instrumented launch of a kind only (default 1 = all) */
void lcr_ktimer_kinds(unsigned mask); /* time only the kinds whose bit is set (gemm 0, aggregate 1, radius 2, attention 3, fused 4, sinkhorn 5; default all) */
int lcr_ktimer_read(int kind, int max_records, double* seconds, int64_t* meta);
/* same, plus the kernel's own begin-to-end duration (what a profiler reports; < 0 where a launch site does not record it) */
int lcr_ktimer_read2(int kind, int max_records, double* seconds, double* seconds_kernel, int64_t* meta);
/* One wavefront spinning for `microseconds` on `stream`.  The runtime deals streams to 4 hardware queues; two busy
 * streams on one queue serialise each other.  A short kernel on stream B behind a spin on stream A tells whether A and B share
 * a queue (lcr-net_amd/pipeline.py picks its streams on distinct queues this way).  Test / tuning hooks live in lcr_hip_debug.h. */
int lcr_stream_spin(int microseconds, void* stream);
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
/* Same on rows of `row_floats` (3 .. 64) floats whose first three are x, y, z — a KITTI velodyne scan f32[N,4] (x, y, z, intensity:
 * data/Kitti/downsample_pcd.py:21; dataset_overlap_online.py:245 slices [:, :3] on the host) is consumed as it lies in memory; the
 * output is f32[M,3]. */
int lcr_grid_subsample_rows(const float* xyz, int row_floats, const int64_t* len, int B, int64_t n_cap, float voxel, int key_bits_hint,
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
/* lcr_radius_query with a processing order for the queries (q_order i32[nq]: a permutation of the query rows, e.g. the query set's
 * own cell order from lcr_support_grid_build_ex; NULL = row order).  Results are identical; spatially coherent wavefronts re-use
 * their candidate cells from cache. */
int lcr_radius_query_ordered(const float* q, const int64_t* qlen, int B, int64_t nq_cap,
                             const void* grid_ws, int64_t ns_cap, float radius, int limit,
                             int64_t* out_idx64, int32_t* out_idx32, int32_t* out_cnt, const int32_t* q_order, void* stream);
/* Several searches (<= 12) of one collate in ONE launch: the ten searches of precompute_data_stack_mode (data.py:28-66) run against
 * four grids; alone, the coarse-stage ones are a handful of workgroups on the launch floor.  int32 rows only (limit >= 1), same
 * rows as lcr_radius_query_ordered search by search.  All searches share the cloud count B. */
typedef struct LcrRadiusQuery {
    const float*   q;          /* [nq_cap,3] */
    const int64_t* qlen;       /* [B] device */
    int64_t        nq_cap;
    const void*    grid_ws;    /* a built support grid */
    int64_t        ns_cap;     /* as passed to its build */
    float          radius;
    int            limit;
    int32_t*       out_idx32;  /* [nq_cap, limit] */
    const int32_t* q_order;    /* processing order or NULL */
} LcrRadiusQuery;
#define LCR_RADIUS_QUERY_MULTI_MAX 12   /* searches per lcr_radius_query_multi launch; longer lists: call it on slices */
int lcr_radius_query_multi(const LcrRadiusQuery* list, int n, int B, void* stream);
/* Same build that also writes the cell-sorted processing order (what lcr_support_grid_order returns) into order i32[ns_cap]. */
int lcr_support_grid_build_ex(const float* s, const int64_t* slen, int B, int64_t ns_cap, float radius,
                              uint32_t* status, void* grid_ws, size_t grid_ws_bytes, int32_t* order, void* stream);
/* order[i] = stacked row of the i-th support in cell-sorted order (a spatially coherent processing order for gather kernels). */
int lcr_support_grid_order(const void* grid_ws, int64_t ns_cap, int B, int32_t* order, void* stream);
int lcr_radius_search_ws_bytes(int64_t nq_cap, int64_t ns_cap, int B, size_t* bytes);
int lcr_radius_search(const float* q, const float* s, const int64_t* qlen, const int64_t* slen, int B,
                      int64_t nq_cap, int64_t ns_cap, float radius, int limit,
                      int64_t* out_idx64, int32_t* out_idx32, int32_t* out_cnt, uint32_t* status,
                      void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a-3  the whole per-batch pre-processing — replaces precompute_data_stack_mode (experiments/lcrnet/data.py:10-74: the
 *      subsampling loop :20-29 and the 10 searches :33-69) for a stack of B clouds, as ONE native call: 3 grid subsamples,
 *      4 support grids and up to 10 radius searches issued fork-join over side streams, voxel counts read back at the end.
 * lcr_precompute_layout sizes the two arenas for stage-0 capacity n0 and reports where every result lives in `out`
 * (byte offsets; int32 indices padded with sum(lengths) like the reference's lists; stage-0 points are the input itself).
 * lcr_precompute_batch: points0 f32[n0,3] and lengths0 i64[B] on the device; voxel_size is stage 0's (stage i uses
 * voxel_size * 2^i, radius * 2^i, data.py:28,73); lengths_host i64[num_stages*B] and status_host (LCR_STATUS_* bits) are
 * HOST outputs — the call returns after synchronising `stream`.  LCR_STATUS_KEY_OVERFLOW: retry with key_bits_hint = 0.
 * ------------------------------------------------------------------------------------------------ */
#define LCR_MAX_STAGES 8
typedef struct LcrPrecomputeLayout {
  int     num_stages, B, upsampling;
  int64_t n_raw;                               /* > 0: the input is a stack of RAW scans, voxelised first (stage 0 lives in `out`) */
  int     limits[LCR_MAX_STAGES];
  int64_t cap[LCR_MAX_STAGES];                 /* row capacity of every stage-i array */
  size_t  off_points[LCR_MAX_STAGES];          /* f32[cap,3]            (i >= 1; i == 0 too in raw mode: f32[n_raw,3]) */
  size_t  off_lengths[LCR_MAX_STAGES];         /* i64[B]                (i >= 1; i == 0 too in raw mode) */
  size_t  off_order[LCR_MAX_STAGES];           /* i32[cap]  cell-sorted processing order */
  size_t  off_neighbors[LCR_MAX_STAGES];       /* i32[cap, limits[i]] */
  size_t  off_subsampling[LCR_MAX_STAGES];     /* i32[cap, limits[i]]   rows = stage i+1 points (i < num_stages-1) */
  size_t  off_upsampling[LCR_MAX_STAGES];      /* i32[cap, limits[i+1]] rows = stage i points   (i < num_stages-1, if enabled); upsampling == 2: i32[cap, 1] */
  size_t  out_bytes, ws_bytes;
} LcrPrecomputeLayout;
/* n_raw > 0: raw-scan mode — the input of lcr_precompute_batch is then f32[n_raw,3] raw points + i64[B] raw lengths, voxelised
 * with `raw_voxel` (replaces the offline Open3D step) into stage 0 inside the same call; n0 is the CAPACITY
 * assumed for the voxel count (LCR_STATUS_LEN_MISMATCH in status_host if it was too small: retry with n0 = n_raw).
 * upsampling: 0 = no upsampling lists (descriptor-only deployment), 1 = the reference collate's full rows [n_i, limits[i+1]], 2 = NEAREST-ONLY
 * lists [n_i, 1]: column 0 of the full rows, which is all KPDecoder reads (nearest_upsample, backbone4.py:355-367; modules/kpconv/functional.py:21). */
int lcr_precompute_layout(int64_t n0, int B, int num_stages, const int* limits, int upsampling, int64_t n_raw,
                          LcrPrecomputeLayout* layout);
int lcr_precompute_batch(const float* points0, const int64_t* lengths0, const LcrPrecomputeLayout* layout, float voxel_size,
                         float radius, float raw_voxel, int key_bits_hint, void* out, size_t out_bytes, void* ws, size_t ws_bytes,
                         int64_t* lengths_host, uint32_t* status_host, void* stream);
/* raw-scan mode on rows of `raw_row_floats` floats (x, y, z first; 4 = KITTI velodyne [N,4] as loaded from a .bin / xyzi .npy file and
 * uploaded unsliced): points0 is then f32[n_raw, raw_row_floats].  raw_row_floats = 3 is lcr_precompute_batch. */
int lcr_precompute_batch_rows(const float* points0, int raw_row_floats, const int64_t* lengths0, const LcrPrecomputeLayout* layout,
                              float voxel_size, float radius, float raw_voxel, int key_bits_hint, void* out, size_t out_bytes, void* ws,
                              size_t ws_bytes, int64_t* lengths_host, uint32_t* status_host, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a-4 / a-5 / a-6  KPConv encoder building blocks (fp32).  Index tensors are [M,H] int32 or int64 (idx_is_64),
 * padded with Ns like the reference's neighbour lists; feature tensors are row-major [N,C].
 * ------------------------------------------------------------------------------------------------ */
/* C = A·B on the fp32 matrix cores with fused epilogue: C[m][:] = (A·B)[m][:] / rowdiv[m] + bias, and per-(segment,group)
 * sum / sum-of-squares of C ADDED to stats[LCR_GN_REPLICAS,S,groups,2] (fp64; the caller zeroes the table — one fill per
 * forward pass can cover every layer's table; consumers add the replicas — they
 * only spread same-address atomics) for the GroupNorm that follows.
 * transA: A is stored [K,M]; transB: B is stored [N,K] (nn.Linear weight).  Replaces F.linear + the (15,C,Cout)
 * contraction of KPConv.forward (modules/kpconv/kpconv.py:108-116) and torch.matmul in NetVlad.py:56,68. */
int lcr_gemm_f32(const float* A, const float* B, float* C, int64_t M, int N, int K, int transA, int transB,
                 const float* bias, const float* rowdiv, const int64_t* seg_len, int S, int groups, double* stats,
                 void* stream);
/* The K-deep contractions on the bf16 matrix cores with fp32-faithful operands: an fp32 number is exactly the sum of three bf16 numbers.
 * lcr_split_bf16x3 writes those terms of an array as planes u16[3][n]; lcr_split_bf16x3_tiles writes them for a constant operand (weights
 * [N,K], K % 32 == 0) in the layout the GEMM stages them in — u16[ceil(N/64)][K/32][3][64][32], i.e. ceil(N/64) * (K/32) * 12288 bytes — once;
 * lcr_gemm_f32_bsplit computes C = A[M,K] . B[N,K]^T with A split on the fly and six of the nine cross products (the dropped ones are
 * <= 2^-24 of a product), fp32 accumulation, same epilogue as lcr_gemm_f32.  Not bit-identical to lcr_gemm_f32 (other rounding points). */
int lcr_split_bf16x3(const float* w, int64_t n, uint16_t* planes, void* stream);
int lcr_split_bf16x3_tiles(const float* w, int N, int K, uint16_t* tiles, void* stream);
int lcr_gemm_f32_bsplit(const float* A, const uint16_t* Bs_tiles, float* C, int64_t M, int N, int K, const float* bias, const float* rowdiv,
                        const int64_t* seg_len, int S, int groups, double* stats, void* stream);
/* C = LeakyReLU(GroupNorm(A)) · B^T (+ bias, + statistics of C as above), A being the RAW [M,K] output of the layer whose sums are
 * a_stats[LCR_GN_REPLICAS,S,a_groups,2]: the normalisation happens while A's tiles are staged, the normalised tensor never
 * exists in memory.  Replaces norm_conv + leaky_relu + unary2.mlp of ResidualBlock.forward (modules/kpconv/modules.py:215-217).
 * Needs B stored [N,K] (nn.Linear weight), 32 < N, K <= 256, K % 4 == 0, a segment table (S >= 1) whose segments all hold >= 64
 * rows (the caller's guarantee: a 64-row block then touches at most two segments).  LCR_EARG outside that form. */
int lcr_gemm_f32_anorm(const float* A, const float* B, float* C, int64_t M, int N, int K, const float* bias,
                       const double* a_stats, const float* a_gamma, const float* a_beta, int a_groups, float a_eps,
                       float a_slope, const int64_t* seg_len, int S, int groups, double* stats, void* stream);
/* K-deep problems (A [M,K] x B [K,N], K >= 480) whose 64x64 tiles do not divide evenly over the CUs can run as stream-K (opt-in,
 * environment LCR_GEMM_STREAMK; scratch library-owned, one per stream; hook: lcr_gemm_debug_streamk in lcr_hip_debug.h). */
/* Batched C_z[M,N] = A_z^T · B_z with A_z stored [K_z, M] (per-entry K and element offsets, HOST arrays, count <= 64). */
int lcr_gemm_f32_batched_ta(const float* A, const float* B, float* C, int64_t M, int N, int count, const int* k_host,
                            const int64_t* a_off_host, const int64_t* b_off_host, const int64_t* c_off_host, void* stream);
/* Gather + kernel-point influences + weighted aggregation of KPConv.forward (kpconv.py:91-105): A[m][k*C+c] =
 * sum_h max(0, 1-|s[idx[m,h]]-q[m]-kp[k]|/sigma) * s_feats[idx[m,h]][c];  nn[m] = max(1, #neighbours with s_pos != 0)
 * (kpconv.py:113-116).  kernel_points_host: 15x3 floats in HOST memory.  C in {32,64,128,256}, H <= 128. */
int lcr_kpconv_aggregate(const float* s_feats, const uint8_t* s_pos, const float* q_pts, const float* s_pts,
                         const void* idx, int idx_is_64, int64_t M, int64_t Ns, int H, int C,
                         const float* kernel_points_host, float sigma, float* A, float* nn,
                         const int32_t* order /* optional i32[M]: processing order, e.g. lcr_support_grid_order */, void* stream);
/* Same with flags.  LCR_KP_VALID_FIRST: the caller guarantees that every row of idx holds its valid entries first and the padding (any value
 * outside [0, Ns)) behind them — what a radius search emits (radius_neighbors_cpu.cpp:70-88 pads behind the sorted hits); the kernel then
 * stops reading a row at the first 64-column chunk that contains padding.  Results are identical for such rows. */
#define LCR_KP_VALID_FIRST 1
int lcr_kpconv_aggregate_ex(const float* s_feats, const uint8_t* s_pos, const float* q_pts, const float* s_pts,
                            const void* idx, int idx_is_64, int64_t M, int64_t Ns, int H, int C,
                            const float* kernel_points_host, float sigma, float* A, float* nn,
                            const int32_t* order, int flags, void* stream);
/* Whole KPConv (kpconv.py:79-122) for C_in = C_out = 32 — the two widest query sets of the encoder — in one launch: the
 * aggregate above lives only as 16-query tiles in LDS and is contracted there with W [15*32, 32] (split-K over the wavefronts,
 * weights in registers); out[M,32] = contraction / neighbour count + bias, plus the GroupNorm sums of `out` ADDED to
 * stats[LCR_GN_REPLICAS,S,groups,2] (optional; S <= 64).  Same results as lcr_kpconv_aggregate + lcr_gemm_f32 up to fp32
 * summation order. */
int lcr_kpconv_fused(const float* s_feats, const uint8_t* s_pos, const float* q_pts, const float* s_pts, const void* idx,
                     int idx_is_64, int64_t M, int64_t Ns, int H, int C, const float* kernel_points_host, float sigma,
                     const float* W, const float* bias, float* out, const int64_t* seg_len, int S, int groups, double* stats,
                     const int32_t* order, void* stream);
/* Whole KPConv for C_in = 1 (encoder1_1, backbone4.py:15): out[M,Cout] incl. count normalisation and bias. W: [15,Cout]. */
int lcr_kpconv_cin1(const float* s_feats, const float* q_pts, const float* s_pts, const void* idx, int idx_is_64,
                    int64_t M, int64_t Ns, int H, const float* kernel_points_host, float sigma, const float* W,
                    const float* bias, int Cout, float* out, const int32_t* order, void* stream);
/* maxpool over neighbours, zero shadow row (modules/kpconv/functional.py:54-67). */
int lcr_maxpool(const float* x, const void* idx, int idx_is_64, int64_t M, int64_t Ns, int H, int C, float* out,
                const int32_t* order, void* stream);
/* pos[n] = (sum_c x[n][c] > 0) — the flag behind KPConv's neighbour count. */
int lcr_row_positive(const float* x, int64_t N, int C, uint8_t* pos, void* stream);
/* Segmented GroupNorm statistics (sum, sumsq per segment and group, fp64, [LCR_GN_REPLICAS,S,groups,2]) of x[N,C],
 * ADDED to the caller-zeroed table. */
int lcr_groupnorm_stats(const float* x, int64_t N, int C, int groups, const int64_t* seg_len, int S, double* stats, void* stream);
/* y = act( GN(x; stats,gamma,beta) [+ res | + GN(res; res_stats,res_gamma,res_beta)] ), act = LeakyReLU(slope) if act != 0
 * (modules/kpconv/modules.py:33-50, 78-84, 207-225).  Optional pos[n] = (sum_c y[n][c] > 0). */
int lcr_groupnorm_apply(const float* x, const double* stats, const float* gamma, const float* beta, const float* res,
                        const double* res_stats, const float* res_gamma, const float* res_beta, float* y, int64_t N, int C,
                        int groups, const int64_t* seg_len, int S, float eps, float slope, int act, uint8_t* pos, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a-6  the whole KPEncoder.forward (experiments/lcrnet/backbone4.py:60-89: encoder1_1 ... encoder4_3) as ONE native call: the
 *      host-side sequencer over the building blocks above (same launches, arguments and order as the module tree of
 *      modules/kpconv/modules.py:104-225, so the outputs are bit-identical to calling the blocks one by one).
 * Weights are device pointers into the model's own parameter tensors (nothing is copied or repacked); kernel points are HOST
 * float[15*3] arrays.  A NULL `w` in a unary block means nn.Identity (modules.py:171,176).
 *   points[4] f32[n_i,3]; neighbors[4] i32[n_i, limits[i]]; subsampling[3] i32[n_{i+1}, limits[i]]; order[4] i32[n_i] or NULL;
 *   seg_len[4] i64[S] GroupNorm segments per stage (device); n_host[4], limits[4] on the host; feats0 f32[n_0] (C_in = 1);
 *   seg_min_rows_host[4] (or NULL): rows of the shortest segment per stage as the host knows them, 0 = unknown — with >= 64
 *   the in-block norm_conv pass is folded into unary2's GEMM (lcr_gemm_f32_anorm), otherwise it is a launch of its own;
 *   out_feats[4]: f32[n_0,2d], [n_1,4d], [n_2,8d], [n_3,16d] (d = init_dim) — the four stage outputs (feats_list).
 * ------------------------------------------------------------------------------------------------ */
#define LCR_ENC_BLOCKS 10
typedef struct LcrUnaryW {
  const float *w, *b;          /* nn.Linear weight [cout,cin], bias [cout] */
  const float *gn_w, *gn_b;    /* GroupNorm affine [cout] */
  const uint16_t* w_split;     /* optional: lcr_split_bf16x3_tiles of w; used for K-deep shapes (cin >= 288, cin % 32 == 0, cout >= 64) */
} LcrUnaryW;
typedef struct LcrBlockW {     /* ResidualBlock (modules.py:148-225) */
  int   cin, cout, strided;
  float sigma;
  const float* kernel_points_host;
  const float *kp_w, *kp_b;    /* KPConv weights [15, cout/4, cout/4], bias [cout/4] */
  const float* kp_wt;          /* optional: the same weights as [cout/4, 15 * cout/4] (k-contiguous rows: the K-deep GEMM form); NULL = not provided */
  const uint16_t* kp_wt_split; /* optional: lcr_split_bf16x3_tiles of kp_wt: the contraction then runs on the bf16 matrix cores (lcr_gemm_f32_bsplit; cout/4 >= 64) */
  const float *normconv_w, *normconv_b;
  LcrUnaryW unary1, unary2, shortcut;
} LcrBlockW;
typedef struct LcrEncoderW {
  int   groups, c1_cout;
  float c1_sigma;
  const float* c1_kernel_points_host;
  const float *c1_w, *c1_b;    /* encoder1_1 KPConv [15,1,c1_cout], bias */
  const float *c1_gn_w, *c1_gn_b;
  LcrBlockW blocks[LCR_ENC_BLOCKS];   /* encoder1_2, 2_1, 2_2, 2_3, 3_1, 3_2, 3_3, 4_1, 4_2, 4_3 */
} LcrEncoderW;
/* neighbors / subsampling: any [n, limit] index rows padded with the number of support rows (padding may sit anywhere in a row). */
int lcr_encoder_ws_bytes(const LcrEncoderW* W, const int64_t* n_host, int S, size_t* bytes);
int lcr_encoder_forward(const LcrEncoderW* W, const float* feats0, const float* const* points, const int32_t* const* neighbors,
                        const int32_t* const* subsampling, const int32_t* const* order, const int64_t* const* seg_len, int S,
                        const int64_t* n_host, const int64_t* seg_min_rows_host, const int* limits, float* const* out_feats, void* ws,
                        size_t ws_bytes, void* stream);
/* Same with flags.  LCR_ENC_LISTS_VALID_FIRST: the caller guarantees that every row of neighbors / subsampling holds its valid entries
 * first and the padding behind them — what a radius search emits (the reference's radius_neighbors and lcr_radius_search alike); the
 * KPConv aggregation then stops reading a row at its first chunk with a hole (LCR_KP_VALID_FIRST of lcr_kpconv_aggregate_ex).  With
 * rows that violate the promise the neighbours behind the first hole are silently ignored, hence opt-in (lcr_encoder_forward = flags 0).
 * The Python host sets it for data dictionaries built by its own collate (`lists_valid_first`).  Environment LCR_KP_VALID_FIRST=0
 * ignores the flag (A/B). */
#define LCR_ENC_LISTS_VALID_FIRST 1u
int lcr_encoder_forward_ex(const LcrEncoderW* W, const float* feats0, const float* const* points, const int32_t* const* neighbors,
                           const int32_t* const* subsampling, const int32_t* const* order, const int64_t* const* seg_len, int S,
                           const int64_t* n_host, const int64_t* seg_min_rows_host, const int* limits, float* const* out_feats,
                           unsigned flags, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a-7  global descriptor head: F.normalize -> NetVLADLoupe2 -> GatingContext -> F.normalize
 *      (model_family/LCRNet_GlobalDescrition.py:34-38, modules/netvlad/NetVlad.py:49-87, 165-201), S scans per call.
 * ------------------------------------------------------------------------------------------------ */
typedef struct LcrNetvladWeights {   /* device pointers; reference parameter names `netvlad.*` */
  const float* cluster_weights;      /* (1024, 64) */
  const float* cluster_weights2;     /* (1, 1024, 64) */
  const float* hidden1_weights;      /* (65536, 256) */
  const float *bn1_w, *bn1_b, *bn1_mean, *bn1_var;
  const float *bn2_w, *bn2_b, *bn2_mean, *bn2_var;
  const float* gating_weights;       /* (256, 256) */
  const float *gbn_w, *gbn_b, *gbn_mean, *gbn_var;
} LcrNetvladWeights;
int lcr_netvlad_ws_bytes(int64_t n_rows, int S, size_t* bytes);
int lcr_netvlad_forward(const float* feats /*[sum(seg_len),1024]*/, const int64_t* seg_len_host, int S,
                        const LcrNetvladWeights* weights_host, float* out /*[S,256]*/, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a-8  3D-RoFormer attention block (fp32) — RPEMultiHeadAttention / MultiHeadAttention of
 *      modules/thdroformer/rpetransformer.py:41-108 and vanilla_transformer.py:30-118.
 * ------------------------------------------------------------------------------------------------ */
/* x[N, heads*32] <- x*cos(theta) + rot(x)*sin(theta) in place, theta[N, heads*16] (one angle per adjacent channel pair). */
int lcr_rotary_embed(float* x, const float* theta, int64_t N, int heads, void* stream);
/* out[Nq, heads*32] = softmax(q k^T / sqrt(32)) v per head, fused (no score matrix in memory); head_dim must be 32. */
int lcr_attention_f32(const float* q, const float* k, const float* v, int64_t Nq, int64_t Nk, int heads, int head_dim,
                      float* out, void* stream);
/* P (<= 64) independent attention problems in one launch over stacked q / k / v: problem p attends its q_len[p] query rows to its
 * k_len[p] key rows (HOST arrays; rows are stacked in problem order).  The batch form of RPEMultiHeadAttention /
 * MultiHeadAttention for P registration pairs per call (reference loop: model_family/LCRNet.py:274-321, one pair per forward). */
int lcr_attention_seg_f32(const float* q, const float* k, const float* v, const int64_t* q_len_host, const int64_t* k_len_host, int P,
                          int heads, int head_dim, float* out, void* stream);
/* Top-k sparsified attention — dynamic_attention with k != None (rpetransformer.py:19-39; cfg.GAT.k, None in the shipped configuration):
 * per query and head the kk_host[p] largest scores of problem p are soft-maxed, all others contribute nothing (kk = int(n_queries * k) is
 * the caller's arithmetic; ties at the threshold are taken in index order — torch.topk leaves that order unspecified).  Keys per problem
 * <= 4096. */
int lcr_attention_topk_f32(const float* q, const float* k, const float* v, const int64_t* q_len_host, const int64_t* k_len_host,
                           const int* kk_host, int P, int heads, int head_dim, float* out, void* stream);
/* y = LayerNorm(a + b) (b may be NULL), rows of D <= 1024 features. */
int lcr_add_layernorm(const float* a, const float* b, const float* gamma, const float* beta, int64_t N, int D, float eps,
                      float* y, void* stream);
int lcr_relu_inplace(float* x, int64_t n, void* stream);

/* The whole ThDRoFormer.forward (thdroformer_linear.py:60-97: position embedding, in_proj, [self, cross] x num_layers, out_proj) as ONE native
 * call: the host-side sequencer over the entries above (same launches, arguments and order as lcr-net_amd/modules/thdroformer, so the
 * outputs are bit-identical to the module tree).  Dense attention only (cfg.GAT.k = None, the shipped configuration; the top-k form runs
 * through the module tree).  Weights are device pointers into the model's own parameters (nn.Linear layout [out,in]).
 *   points f32[n,3], feats f32[n,d_in]: the rows of all FIRST clouds of the P pairs stacked, then those of all SECOND clouds
 *   (n = sum lens0 + sum lens1; lens*_host: P entries each, 1 <= P <= 32);
 *   out f32[n,d_out] (same row order), theta_out f32[n,d_model/2] (the rotary angles, thdroformer_linear.py:94-95). */
#define LCR_ROFORMER_MAX_BLOCKS 16
typedef struct LcrLinearW {
  const float *w, *b;          /* nn.Linear weight [out,in], bias [out] */
} LcrLinearW;
typedef struct LcrRoformerLayerW {   /* _TransformerLayer: attention (proj_q/k/v, linear, norm) + output (expand, squeeze, norm) */
  LcrLinearW   q, k, v, lin;
  const float *ln1_w, *ln1_b;
  float        ln1_eps;
  LcrLinearW   expand, squeeze;
  const float *ln2_w, *ln2_b;
  float        ln2_eps;
} LcrRoformerLayerW;
typedef struct LcrRoformerW {
  int        d_in, d_model, d_out, heads, num_blocks;
  int        block_is_self[LCR_ROFORMER_MAX_BLOCKS];   /* 1 = rotary self layer over both clouds, 0 = sequential cross layer */
  LcrLinearW emb1, emb2, in_proj, out_proj;
  LcrRoformerLayerW layers[LCR_ROFORMER_MAX_BLOCKS];
} LcrRoformerW;
int lcr_roformer_ws_bytes(const LcrRoformerW* W, int64_t n_rows, size_t* bytes);
int lcr_roformer_forward(const LcrRoformerW* W, const float* points, const float* feats, const int64_t* lens0_host, const int64_t* lens1_host,
                         int P, float* out, float* theta_out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a-9  descriptor retrieval: exhaustive squared-L2 top-k with the temporal exclusion window — replaces the per-query
 *      faiss IndexIVFFlat(nlist=1) loop of experiments/loop_detection/eval_loop_detection_overlap_dataset.py:183-214.
 * Query row r is global frame q0+r; its database is frames [0, q0+r-exclude).  Rows ascending in (d2, index); short rows
 * are padded with (-1, +inf).  k <= 128.
 * ------------------------------------------------------------------------------------------------ */
int lcr_retrieval_ws_bytes(int64_t Q, int64_t C, size_t* bytes);
int lcr_retrieval_topk(const float* queries /*[Q,D]*/, int64_t Q, int64_t q0, const float* database /*[C,D]*/, int64_t C,
                       int D, int k, int exclude, int32_t* out_idx /*[Q,k]*/, float* out_d2 /*[Q,k]*/,
                       void* ws, size_t ws_bytes, void* stream);

/* Uniform batched GEMM: C_z = op(A_z)·op(B_z), constant strides in floats (multiples of 4), count <= 65535
 * (einsum('bnd,bmd->bnm') of LCRNet.py:237). */
int lcr_gemm_f32_strided_batched(const float* A, const float* B, float* C, int64_t M, int N, int K, int transA, int transB,
                                 int64_t strideA, int64_t strideB, int64_t strideC, int count, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a-10  registration tail (all device-side; replaces op chains with .cpu() hops and Python loops)
 * ------------------------------------------------------------------------------------------------ */
/* out = xyz + off * min(1, max_range/|off|)  (modules/vote/vote.py:166-175) */
int lcr_vote_shift(const float* xyz, const float* offsets, int64_t N, float max_range, float* out, void* stream);
/* Greedy NMS of modules/vote/vote.py:13-70, exact (order-dependent rule replayed in parallel rounds), one cloud per
 * workgroup.  keep u8[n_total], out_len i64[B]; ws: lcr_greedy_nms_ws_bytes(n_total). */
int lcr_greedy_nms_ws_bytes(int64_t n_total, size_t* bytes);
int lcr_greedy_nms(const float* pts, const int64_t* len, int B, int64_t n_total, float radius, uint8_t* keep,
                   int64_t* out_len, void* ws, void* stream);
/* out[m] = mean of pts[idx[m,h]] over the valid (0 <= idx < pad) neighbours (backbone4.py:161-175). */
int lcr_neighbor_mean(const float* pts, const void* idx, int idx_is_64, int64_t M, int H, int64_t pad, float* out, void* stream);
/* point_to_node_partition (modules/ops/pointcloud_partition.py:60-107): knn i64[M,K] padded with N, knn_mask u8[M,K],
 * node_mask u8[M], optional p2n i32[N]; M <= 4000. */
int lcr_point_to_node_ws_bytes(int64_t N, int M, size_t* bytes);
int lcr_point_to_node_partition(const float* points, int64_t N, const float* nodes, int M, int K, int32_t* p2n, int64_t* knn,
                                uint8_t* knn_mask, uint8_t* node_mask, uint32_t* status, void* ws, size_t ws_bytes, void* stream);

/* The same partition for C stacked clouds in one launch sequence (cloud c: points [point_off[c], point_off[c+1]), nodes
 * [node_off[c], node_off[c+1]); HOST offsets, C <= 64): per-cloud results stacked, indices local to their cloud, knn padded with the
 * cloud's own point count.  Workspace: lcr_point_to_node_ws_bytes(total points, total nodes).  The pair model's group path calls
 * this once per group of pairs instead of once per cloud (reference: ops/pointcloud_partition.py:60-107, called per cloud at
 * model_family/LCRNet.py:150-165). */
int lcr_point_to_node_partition_stack(const float* points, const int64_t* point_off, const float* nodes, const int64_t* node_off, int C, int K,
                                      int32_t* p2n_out, int64_t* knn, uint8_t* knn_mask, uint8_t* node_mask, uint32_t* status, void* ws,
                                      size_t ws_bytes, void* stream);
/* S[B,M+1,N+1] = scale*raw with dustbin row/col = alpha[0] and masked rows/cols = -inf_val (learnable_sinkhorn.py:36-45). */
int lcr_build_padded_scores(const float* raw, const uint8_t* row_mask, const uint8_t* col_mask, int64_t B, int M, int N,
                            float scale, const float* alpha, float inf_val, float* S, void* stream);
/* LearnableLogOptimalTransport.forward (learnable_sinkhorn.py:13-66) in place on S; uv_ws: B*(2*(M+N+2)+1) floats. */
int lcr_log_sinkhorn(float* S, const uint8_t* row_mask, const uint8_t* col_mask, int64_t B, int M, int N, int iters,
                     float inf_val, float* uv_ws, void* stream);
/* same with a workspace of lcr_log_sinkhorn_ws_floats(B, M, N) floats: matrices beyond LDS (the ~350 x 330 node level) then run as ONE
 * persistent launch (row slabs per workgroup, one counter hand-off per iteration) instead of two launches per iteration; the last
 * floats of the workspace hold a status word (bit 0: a hand-off timed out — the result is then invalid). */
int lcr_log_sinkhorn_ws_floats(int64_t B, int M, int N, size_t* floats);
/* which form lcr_log_sinkhorn_ex takes for (B, M, N) with a workspace of lcr_log_sinkhorn_ws_floats floats: 0 register-resident
 * (scaled / log domain), 1 LDS-resident, 2 persistent (the ONLY form that writes the status word), 3 two launches per iteration. */
int lcr_log_sinkhorn_form(int64_t B, int M, int N, int* form);
int lcr_log_sinkhorn_ex(float* S, const uint8_t* row_mask, const uint8_t* col_mask, int64_t B, int M, int N, int iters,
                        float inf_val, float* uv_ws, size_t uv_floats, void* stream);
/* Dustbin top-1 matching in the exp domain (superpoint_matching.py:130-162; local_global_registration.py:49-92 with k=1,
 * mutual=False, use_dustbin=True): (b,i,j) kept if it is its row's maximum beating the dustbin column OR its column's
 * maximum beating the dustbin row (and row/col masks, if given).  Phase 1: out_bij == NULL -> *total (device i64);
 * phase 2: same arguments with out_bij i32[total,3], out_score f32[total].  Row-major order. */
int lcr_top1_matching_ws_bytes(int64_t B, int M, int N, size_t* bytes);
int lcr_top1_matching(const float* logS, int64_t B, int M, int N, const uint8_t* row_mask, const uint8_t* col_mask,
                      int64_t* total, int32_t* out_bij, float* out_score, void* ws, size_t ws_bytes, void* stream);
/* Same with the `mutual` switch of LocalGlobalRegistration (local_global_registration.py:84-87): != 0 keeps a pair only if it is both
 * its row's and its column's maximum (each beating its dustbin); 0 = either, the shipped configuration. */
int lcr_top1_matching_ex(const float* logS, int64_t B, int M, int N, const uint8_t* row_mask, const uint8_t* col_mask, int mutual,
                         int64_t* total, int32_t* out_bij, float* out_score, void* ws, size_t ws_bytes, void* stream);
/* The patch score matrices of the dense point matching (model_family/LCRNet.py:236-250 + the padding of learnable_sinkhorn.py:38-49) in one
 * kernel: S[p] f32[(K+1),(K+1)] = scale * gather(feats_a, idx_a[p]) . gather(feats_b, idx_b[p])^T over C channels on the fp32 matrix cores,
 * dustbin row / column = *alpha, entries of masked rows / columns = -inf_val — what lcr_build_padded_scores makes of the batched product of
 * two lcr_gather_rows results, without the gathered copies.  idx_* i64[P,K] with the shadow index (== N*) for a zero row, mask_* u8[P,K];
 * K = 128 (cfg.model.num_points_in_patch), C % 32 == 0, feats 16-byte aligned.  feats_a / feats_b may be the same tensor. */
int lcr_patch_scores(const float* feats_a, int64_t Na, const float* feats_b, int64_t Nb, int C, const int64_t* idx_a, const int64_t* idx_b,
                     const uint8_t* mask_a, const uint8_t* mask_b, int64_t P, int K, float scale, const float* alpha, float inf_val, float* S,
                     void* stream);
/* Dustbin top-K matching for K >= 1 (LocalGlobalRegistration(k=K), local_global_registration.py:56-82; K = 1 in the shipped configuration):
 * (i, j) is kept from the row side if P[i][j] is among the K largest of row i — dustbin column included, equal values in index order — and
 * beats the row's dustbin; from the column side likewise; `mutual` as above.  Two-phase and row-major like lcr_top1_matching. */
int lcr_topk_matching_ws_bytes(int64_t B, int M, int N, size_t* bytes);
int lcr_topk_matching(const float* logS, int64_t B, int M, int N, const uint8_t* row_mask, const uint8_t* col_mask, int K, int mutual,
                      int64_t* total, int32_t* out_bij, float* out_score, void* ws, size_t ws_bytes, void* stream);
/* Every switch of LocalGlobalRegistration.compute_correspondence_matrix (local_global_registration.py:48-93):
 *   use_dustbin = 0 (:62-65, :74-77; LCRNet.py:256-257 strips the dustbin first): the K largest are taken over the M x N interior, the selected
 *     values are scattered into a zero matrix and kept where that exceeds `confidence_threshold` (an unselected entry counts as 0);
 *   global_scores f32[B] or NULL (use_global_score, :236-237): emitted scores of patch pair b are multiplied by global_scores[b].
 * logS is the (M+1) x (N+1) transport output in both forms.  use_dustbin = 1, threshold ignored, NULL = lcr_topk_matching. */
int lcr_topk_matching_ex(const float* logS, int64_t B, int M, int N, const uint8_t* row_mask, const uint8_t* col_mask, int K, int mutual,
                         int use_dustbin, float confidence_threshold, const float* global_scores, int64_t* total, int32_t* out_bij,
                         float* out_score, void* ws, size_t ws_bytes, void* stream);
/* out[n] = [ x[idx[n,0]] (zeros for the shadow index), skip[n] ]  (nearest_upsample + cat, backbone4.py:355-367) */
int lcr_upsample_concat(const float* x, int64_t Nx, int C1, const void* idx, int idx_is_64, int H, const float* skip, int C2,
                        int64_t N, float* out, void* stream);
/* out[r] = src[idx[r]] (zeros where idx == pad): index_select on a zero-padded tensor */
int lcr_gather_rows(const float* src, int64_t pad, int C, const int64_t* idx, int64_t R, float* out, void* stream);
/* weighted_procrustes (modules/registration/procrustes.py:6-73), batched: problem p = correspondences [start[p],start[p+1]);
 * 3x3 SVD on the device (one-sided Jacobi, fp64); T f32[P,4,4]. */
int lcr_procrustes_batched(const float* src, const float* ref, const float* w, const int32_t* start, int P, float eps, float* T, void* stream);
/* counts[p] = #{ |ref - T_p src| < radius } (-1 if the hypothesis came from < min_count correspondences); best = first argmax */
int lcr_inlier_count(const float* T, int P, const float* src, const float* ref, int n, float radius, const int32_t* start,
                     int min_count, int32_t* counts, int32_t* best, void* stream);
/* local_to_global_registration (geotransformer/local_global_registration.py:134-201) for S pairs in one launch sequence:
 * correspondences stacked pair-major; hypothesis h = weighted Procrustes of rows [hyp_start[h], hyp_start[h+1]) (one per patch
 * correspondence); pair s owns hypotheses [seg_hyp_start[s], seg_hyp_start[s+1]).  Per pair: inlier counts of its hypotheses over
 * its own rows (chunks below min_count rows never win), first argmax, then `steps` re-weighted refits; a pair without a valid
 * hypothesis starts from the fit over all its rows.  T_out f32[S,4,4]; optional hyp_out f32[H,4,4], counts_out i32[H], best_out
 * i32[S] (-1: fallback).  No host synchronisation. */
int lcr_lgr_ws_bytes(int64_t n, int H, int S, size_t* bytes);
int lcr_local_global_registration(const float* src, const float* ref, const float* score, int64_t n, const int32_t* hyp_start, int H,
                                  const int32_t* seg_hyp_start, int S, float radius, int min_count, int steps, float* T_out,
                                  float* hyp_out, int32_t* counts_out, int32_t* best_out, void* ws, size_t ws_bytes, void* stream);
/* Same with LocalGlobalRegistration(correspondence_limit=L) (:152-160): a pair with more than L correspondences verifies (inlier counts, the
 * degenerate-branch fit and every refit) on its L highest-scoring ones only — ties at the L-th score in row order — while the hypotheses
 * still come from all rows of their patch correspondence.  L = 0: no limit (= lcr_local_global_registration).  Same workspace size. */
int lcr_local_global_registration_ex(const float* src, const float* ref, const float* score, int64_t n, const int32_t* hyp_start, int H,
                                     const int32_t* seg_hyp_start, int S, float radius, int min_count, int steps, int correspondence_limit,
                                     float* T_out, float* hyp_out, int32_t* counts_out, int32_t* best_out, void* ws, size_t ws_bytes,
                                     void* stream);
/* w_out = score * [ |ref - T src| < radius ], T = T_all[sel ? *sel : 0]  (recompute_correspondence_scores, LGR :127-132) */
int lcr_inlier_weights(const float* T_all, const int32_t* sel, const float* src, const float* ref, const float* score, int n,
                       float radius, float* w_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LCR_HIP_H */
