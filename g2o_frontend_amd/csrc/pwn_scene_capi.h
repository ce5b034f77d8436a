// pwn_scene_capi.h -- C-ABI entry points of the scene-maintenance stage; included by
// pwn_hip_capi.hip (same translation unit: it uses that file's context, cloud handle and helpers).
// Reference: pwn_core/{cloud.cpp:11-186, merger.cpp:15-119, voxelcalculator.cpp:15-73, gaussian3.h, pinholepointprojector.cpp:93-133}.
#pragma once

#include <fstream>
#include <sstream>

namespace {

int scene_alloc(pwn_hip_ctx* ctx, void** p, size_t bytes) {
  if (*p) return PWN_HIP_OK;
  HIPCHK(ctx, hipMalloc(p, bytes), PWN_HIP_ERR_ALLOCATION);
  return PWN_HIP_OK;
}
// Gaussians of a cloud (24 floats + flags per point)
int ensure_gauss(pwn_hip_ctx* ctx, pwn_hip_cloud* c) {
  const size_t cap = (size_t)c->d.capacity;
  if (int rc = scene_alloc(ctx, (void**)&c->sb.G, cap * sizeof(GaussD))) return rc;
  if (!c->sb.Gf) {
    if (int rc = scene_alloc(ctx, (void**)&c->sb.Gf, cap * sizeof(int))) return rc;
    HIPCHK(ctx, hipMemsetAsync(c->sb.Gf, 0, cap * sizeof(int), ctx->stream), PWN_HIP_ERR_COPY);
  }
  return PWN_HIP_OK;
}
// a cloud that receives appended clouds keeps explicit normal information matrices and Stats
int ensure_scene(pwn_hip_ctx* ctx, pwn_hip_cloud* c, bool with_gauss) {
  const size_t cap = (size_t)c->d.capacity;
  if (!c->d.OmN) {
    float* planes = nullptr;
    if (int rc = scene_alloc(ctx, (void**)&planes, cap * 9 * sizeof(float))) return rc;
    if (c->n_host > 0) hipLaunchKernelGGL(k_expand_omega_n, dim3((c->n_host + 255) / 256), dim3(256), 0, ctx->stream, c->d, planes, c->n_host);
    c->d.OmN = planes;
  }
  if (!c->d.St) {
    if (int rc = scene_alloc(ctx, (void**)&c->d.St, cap * 16 * sizeof(float))) return rc;
    c->has_stats = false;
  }
  if (!c->has_stats) {        // default Stats() for whatever the cloud already holds
    if (c->n_host > 0) {
      std::vector<float> st((size_t)c->n_host * 16, 0.f);
      for (int i = 0; i < c->n_host; ++i) { st[(size_t)16 * i] = 1.f; st[(size_t)16 * i + 4] = 1.f; st[(size_t)16 * i + 8] = 1.f; }
      HIPCHK(ctx, hipMemcpyAsync(c->d.St, st.data(), st.size() * 4, hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY);
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_COPY);
    }
    c->has_stats = true;
  }
  if (with_gauss) { if (int rc = ensure_gauss(ctx, c)) return rc; }
  return PWN_HIP_OK;
}
// second set of per-point arrays: stable compactions / reorderings write there, then the two sets are swapped
int ensure_back(pwn_hip_ctx* ctx, pwn_hip_cloud* c) {
  const size_t cap = (size_t)c->d.capacity;
  c->back.capacity = c->d.capacity; c->back.count = c->d.count;
  std::memcpy(c->back.omN, c->d.omN, sizeof(c->d.omN)); c->back.clsThr = c->d.clsThr; c->back.omSym = c->d.omSym;
  if (int rc = scene_alloc(ctx, (void**)&c->back.P3, cap * 3 * sizeof(float))) return rc;
  if (int rc = scene_alloc(ctx, (void**)&c->back.Nc, cap * sizeof(float4))) return rc;
  if (int rc = scene_alloc(ctx, (void**)&c->back.Om, om_floats(c->d) * sizeof(float))) return rc;
  if (c->d.OmN) { if (int rc = scene_alloc(ctx, (void**)&c->back.OmN, cap * 9 * sizeof(float))) return rc; }
  if (c->d.St) { if (int rc = scene_alloc(ctx, (void**)&c->back.St, cap * 16 * sizeof(float))) return rc; }
  if (c->sb.G) {
    if (int rc = scene_alloc(ctx, (void**)&c->sback.G, cap * sizeof(GaussD))) return rc;
    if (int rc = scene_alloc(ctx, (void**)&c->sback.Gf, cap * sizeof(int))) return rc;
  }
  return PWN_HIP_OK;
}
void swap_back(pwn_hip_cloud* c) {
  std::swap(c->d.P3, c->back.P3); std::swap(c->d.Nc, c->back.Nc); std::swap(c->d.Om, c->back.Om);
  if (c->d.OmN && c->back.OmN) std::swap(c->d.OmN, c->back.OmN);
  if (c->d.St && c->back.St) std::swap(c->d.St, c->back.St);
  if (c->sb.G && c->sback.G) { std::swap(c->sb.G, c->sback.G); std::swap(c->sb.Gf, c->sback.Gf); }
}
// context scratch of the scene stage: int arrays of n entries and 64-bit arrays
int scene_scratch(pwn_hip_ctx* ctx, size_t n_int, size_t n_u64) {
  if (n_int > ctx->scene_icap) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
    for (int k = 0; k < 8; ++k) { if (ctx->scene_i[k]) (void)hipFree(ctx->scene_i[k]); ctx->scene_i[k] = nullptr; }
    ctx->scene_icap = 0;
    for (int k = 0; k < 8; ++k) HIPCHK(ctx, hipMalloc((void**)&ctx->scene_i[k], (n_int + 1024) * sizeof(int)), PWN_HIP_ERR_ALLOCATION);
    ctx->scene_icap = n_int;
  }
  if (n_u64 > ctx->scene_kcap) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
    for (int k = 0; k < 3; ++k) { if (ctx->scene_k[k]) (void)hipFree(ctx->scene_k[k]); ctx->scene_k[k] = nullptr; }
    ctx->scene_kcap = 0;
    for (int k = 0; k < 3; ++k) HIPCHK(ctx, hipMalloc((void**)&ctx->scene_k[k], n_u64 * sizeof(unsigned long long)), PWN_HIP_ERR_ALLOCATION);
    ctx->scene_kcap = n_u64;
  }
  if (!ctx->scene_total) HIPCHK(ctx, hipMalloc((void**)&ctx->scene_total, 4 * sizeof(int)), PWN_HIP_ERR_ALLOCATION);
  return PWN_HIP_OK;
}
// out[i] = sum of in[0..i), *total = sum of all; sums: scratch of >= n/1024 + 1 ints
int exclusive_scan(pwn_hip_ctx* ctx, const int* in, int* out, int n, int* sums, int* total) {
  const int nb = (n + 1023) / 1024;
  if (n <= 0) { HIPCHK(ctx, hipMemsetAsync(total, 0, sizeof(int), ctx->stream), PWN_HIP_ERR_COPY); return PWN_HIP_OK; }
  hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(1024), 0, ctx->stream, in, out, n, sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, sums, nb, total);
  hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(1024), 0, ctx->stream, out, n, (const int*)sums);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  return PWN_HIP_OK;
}

}  // namespace

extern "C" {

// The Gaussian half of PinholePointProjector::unProject (pinholepointprojector.cpp:104-123) for a cloud that pwn_hip_convert made
// from the same depth image with the same parameters (incl. Cloud::transformInPlace(sensorOffset) on them, cloud.cpp:180).
int pwn_hip_cloud_gaussians(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float* depth, int rows, int cols, pwn_hip_cloud* cloud,
                            float baseline, float alpha) {
  if (!ctx || !p || !depth || !cloud) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (int rc = check_image(ctx, rows, cols)) return rc;
  if (int rc = ensure_gauss(ctx, cloud)) return rc;
  if (int rc = ensure_desc(ctx, 1)) return rc;
  if (int rc = absorb_copies(ctx)) return rc;      // caller pointers may be the destination of a queued pwn_hip_copy_async
  const size_t N = (size_t)rows * cols;
  const ConvertParams cp = make_convert_params(ctx, p, nullptr, rows, cols, 0);
  Mat4 KRt, iKRt; Mat3 iK;
  projector_matrices(mat3_from(p->K), mat4_identity(), KRt, iKRt, iK);
  const float* d = nullptr;
  if (int rc = stage_depth(ctx, depth, N, &d)) return rc;
  fill_frame(ctx, 0, 0, d, cloud->d, rows);
  HIPCHK(ctx, hipMemcpyAsync(ctx->frames_dev, ctx->frames_host, sizeof(FrameDesc), hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY);
  hipLaunchKernelGGL(k_row_count, dim3(rows, 1), dim3(256), 0, ctx->stream, ctx->frames_dev, cp);
  hipLaunchKernelGGL(k_row_offsets, dim3(1), dim3(1024), 0, ctx->stream, ctx->frames_dev, rows);
  hipLaunchKernelGGL(k_gaussians, dim3(rows, 1), dim3(256), 0, ctx->stream, ctx->frames_dev, cp, iK, baseline * p->K[0], alpha, cloud->sb);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  pwn_hip_cloud* cl[1] = { cloud };
  if (int rc = sync_and_counts(ctx, cl, 1)) return rc;
  cloud->n_gauss = cloud->n_host;
  return PWN_HIP_OK;
}
int pwn_hip_cloud_num_gaussians(pwn_hip_ctx* ctx, const pwn_hip_cloud* c, int* n) {
  if (!c || !n) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  *n = c->sb.G ? c->n_gauss : 0;
  return PWN_HIP_OK;
}
// mean n*3, cov n*9 (column-major 3x3), info_vec n*3, info n*9, flags n (1 = moments valid, 2 = information form valid); host pointers
int pwn_hip_cloud_download_gaussians(pwn_hip_ctx* ctx, const pwn_hip_cloud* c, float* mean, float* cov, float* info_vec, float* info, int* flags) {
  if (!ctx || !c) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const int n = c->sb.G ? c->n_gauss : 0;
  if (n == 0) return PWN_HIP_OK;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  std::vector<GaussD> g(n); std::vector<int> f(n);
  HIPCHK(ctx, hipMemcpy(g.data(), c->sb.G, sizeof(GaussD) * (size_t)n, hipMemcpyDeviceToHost), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipMemcpy(f.data(), c->sb.Gf, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost), PWN_HIP_ERR_COPY);
  for (int i = 0; i < n; ++i) {
    if (mean) std::memcpy(mean + 3 * (size_t)i, g[i].mean, 12);
    if (cov) std::memcpy(cov + 9 * (size_t)i, g[i].cov, 36);
    if (info_vec) std::memcpy(info_vec + 3 * (size_t)i, g[i].infoVec, 12);
    if (info) std::memcpy(info + 9 * (size_t)i, g[i].info, 36);
    if (flags) flags[i] = f[i];
  }
  return PWN_HIP_OK;
}

// Cloud::add (cloud.cpp:145-171): dst gets a copy of src transformed by T appended; src is not modified.
int pwn_hip_cloud_add(pwn_hip_ctx* ctx, pwn_hip_cloud* dst, const pwn_hip_cloud* src, const float T[16]) {
  if (!ctx || !dst || !src || !T) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (dst == src) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "cannot add a cloud to itself");
  cloud_changes(ctx, dst);
  const int k = dst->n_host, n = src->n_host;
  if ((size_t)k + (size_t)n > (size_t)dst->d.capacity) return fail(ctx, PWN_HIP_ERR_CAPACITY, "destination cloud capacity too small for Cloud::add");
  const bool srcGauss = src->sb.G && src->n_gauss > 0;
  if (int rc = ensure_scene(ctx, dst, srcGauss || dst->sb.G)) return rc;
  const Mat4 m = forced(T);
  const int ident = is_identity(m) ? 1 : 0;                                  // cloud.cpp:176
  CloudDev s = src->d;
  if (!src->has_stats) s.St = nullptr;
  const int ng = srcGauss ? std::min(src->n_gauss, n) : 0;
  if (n > 0) hipLaunchKernelGGL(k_cloud_append, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dst->d, dst->sb, s, src->sb, k, n, ng, m, ident);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  const int total = k + n;
  HIPCHK(ctx, hipMemcpyAsync(dst->d.count, &total, sizeof(int), hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  // _gaussians.resize(k + cloud.gaussians().size()) (cloud.cpp:153): entries the source does not provide are default Gaussians
  if (dst->sb.G) {
    const int newg = k + (srcGauss ? src->n_gauss : 0);
    if (k > dst->n_gauss) HIPCHK(ctx, hipMemset(dst->sb.Gf + dst->n_gauss, 0, sizeof(int) * (size_t)(k - dst->n_gauss)), PWN_HIP_ERR_COPY);
    dst->n_gauss = std::min(newg, dst->d.capacity);
  }
  dst->n_host = total; dst->idx_valid = false;
  return PWN_HIP_OK;
}

// Merger::merge (merger.cpp:15-119).  K / min_distance / max_distance: the projector of the merger's DepthImageConverter; T: its pose.
// collapsed (optional, host, size of the cloud before the call) receives _collapsedIndices.
int pwn_hip_merge(pwn_hip_ctx* ctx, pwn_hip_cloud* cloud, const float K[9], const float T[16], float min_distance, float max_distance, int rows, int cols,
                  float distance_threshold, float normal_threshold, float max_point_depth, int* new_size, int* collapsed) {
  if (!ctx || !cloud || !K || !T) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (int rc = check_image(ctx, rows, cols)) return rc;
  if (min_distance < 0.f) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "min_distance must be >= 0");
  cloud_changes(ctx, cloud);
  const int n = cloud->n_host;
  if (n > kMaxCloudPoints) return fail(ctx, PWN_HIP_ERR_CAPACITY, "cloud has more points than the z-buffer index field holds (2^25)");
  if (!cloud->sb.G || cloud->n_gauss < n) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "Merger::merge needs the cloud's Gaussians (pwn_hip_cloud_gaussians)");
  if (n == 0) { if (new_size) *new_size = 0; return PWN_HIP_OK; }
  if (int rc = ensure_back(ctx, cloud)) return rc;
  if (int rc = scene_scratch(ctx, (size_t)n, 0)) return rc;
  int* d_collapsed = ctx->scene_i[0]; int* d_head = ctx->scene_i[1]; int* d_next = ctx->scene_i[2]; int* d_keep = ctx->scene_i[3];
  int* d_offs = ctx->scene_i[4]; int* d_sums = ctx->scene_i[5];
  Mat4 KRt, iKRt; Mat3 iK;
  projector_matrices(mat3_from(K), mat4_from(T), KRt, iKRt, iK);
  unsigned tag = kZTag0;
  if (int rc = take_tags(ctx, 1, &tag)) return rc;
  hipStream_t st = ctx->stream;
  const int nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_project_single, dim3(nb), dim3(256), 0, st, cloud->d, KRt, min_distance, max_distance, rows, cols, ctx->zref_ws, tag);
  HIPCHK(ctx, hipMemsetAsync(d_head, 0xFF, sizeof(int) * (size_t)n, st), PWN_HIP_ERR_COPY);
  hipLaunchKernelGGL(k_merge_classify, dim3(nb), dim3(256), 0, st, cloud->d, n, KRt, min_distance, max_distance, max_point_depth, rows, cols,
                     (const unsigned long long*)ctx->zref_ws, tag, distance_threshold, normal_threshold, d_collapsed, d_head, d_next);
  hipLaunchKernelGGL(k_merge_accumulate, dim3(nb), dim3(256), 0, st, cloud->d, cloud->sb, n, (const int*)d_collapsed, (const int*)d_head, (const int*)d_next);
  hipLaunchKernelGGL(k_merge_keep_flags, dim3(nb), dim3(256), 0, st, (const int*)d_collapsed, n, d_keep);
  if (int rc = exclusive_scan(ctx, d_keep, d_offs, n, d_sums, ctx->scene_total)) return rc;
  hipLaunchKernelGGL(k_merge_compact, dim3(nb), dim3(256), 0, st, cloud->d, cloud->sb, cloud->back, cloud->sback, n, cloud->n_gauss, (const int*)d_keep, (const int*)d_offs);
  hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, cloud->d.count, (const int*)ctx->scene_total);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  int k = 0;
  HIPCHK(ctx, hipMemcpyAsync(&k, ctx->scene_total, sizeof(int), hipMemcpyDeviceToHost, st), PWN_HIP_ERR_COPY);
  if (collapsed) HIPCHK(ctx, hipMemcpyAsync(collapsed, d_collapsed, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(st), PWN_HIP_ERR_LAUNCH);
  // the Gaussian vector is compacted but not resized (merger.cpp:108-112): its tail keeps the old entries
  if (cloud->n_gauss > k) hipLaunchKernelGGL(k_gauss_copy_tail, dim3((cloud->n_gauss - k + 255) / 256), dim3(256), 0, st, cloud->sb, cloud->sback, k, cloud->n_gauss);
  HIPCHK(ctx, hipStreamSynchronize(st), PWN_HIP_ERR_LAUNCH);
  swap_back(cloud);
  cloud->n_host = k; cloud->idx_valid = false;
  ctx->img_valid = false;                 // slot 0 of the z-buffer was used
  if (new_size) *new_size = k;
  return PWN_HIP_OK;
}

// VoxelCalculator::compute (voxelcalculator.cpp:15-73) with the intended ordering of the voxel keys (see pwn_scene_kernels.h).
// kept (optional, host) receives the original indices of the survivors in output order.
int pwn_hip_voxelize(pwn_hip_ctx* ctx, pwn_hip_cloud* cloud, float resolution, int* new_size, int* kept) {
  if (!ctx || !cloud || !(resolution > 0.f)) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "bad argument");
  cloud_changes(ctx, cloud);
  const int n = cloud->n_host;
  if (n == 0) { if (new_size) *new_size = 0; return PWN_HIP_OK; }
  if (int rc = ensure_back(ctx, cloud)) return rc;
  size_t table = 1024; while (table < 2 * (size_t)n) table <<= 1;
  const int nblocksSort = (n + kSortChunk - 1) / kSortChunk;
  if (int rc = scene_scratch(ctx, std::max(table, (size_t)256 * nblocksSort + 1024), std::max(table, (size_t)n))) return rc;
  int* d_first = ctx->scene_i[0]; int* d_slot = ctx->scene_i[1]; int* d_keep = ctx->scene_i[2]; int* d_offs = ctx->scene_i[3]; int* d_sums = ctx->scene_i[4];
  int* d_idxA = ctx->scene_i[5]; int* d_idxB = ctx->scene_i[6]; int* d_hist = ctx->scene_i[7];
  unsigned long long* d_table = ctx->scene_k[0]; unsigned long long* d_keyA = ctx->scene_k[1]; unsigned long long* d_keyB = ctx->scene_k[2];
  hipStream_t st = ctx->stream;
  const float inverseResolution = 1.0f / resolution;
  const int nb = (n + 255) / 256;
  int* d_fault = ctx->scene_total + 1;
  HIPCHK(ctx, hipMemsetAsync(d_table, 0xFF, table * 8, st), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipMemsetAsync(d_first, 0x7F, table * 4, st), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipMemsetAsync(d_fault, 0, sizeof(int), st), PWN_HIP_ERR_COPY);
  hipLaunchKernelGGL(k_voxel_insert, dim3(nb), dim3(256), 0, st, cloud->d, n, inverseResolution, d_table, d_first, (unsigned)(table - 1), d_slot, d_fault);
  hipLaunchKernelGGL(k_voxel_survivors, dim3(nb), dim3(256), 0, st, n, (const int*)d_slot, (const int*)d_first, d_keep);
  if (int rc = exclusive_scan(ctx, d_keep, d_offs, n, d_sums, ctx->scene_total)) return rc;
  hipLaunchKernelGGL(k_voxel_records, dim3(nb), dim3(256), 0, st, cloud->d, n, inverseResolution, (const int*)d_keep, (const int*)d_offs, d_keyA, d_idxA);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  int hostv[2] = { 0, 0 };
  HIPCHK(ctx, hipMemcpyAsync(hostv, ctx->scene_total, 2 * sizeof(int), hipMemcpyDeviceToHost, st), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(st), PWN_HIP_ERR_LAUNCH);
  if (hostv[1] != 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, hostv[1] == 1 ? "voxel index outside +-2^20 (or NaN point)" : "voxel table full");
  const int m = hostv[0];
  // LSD radix sort of the survivors by their 63-bit voxel key (8 passes of 8 bits), stable -> lexicographic (ix, iy, iz) order
  const int sb_ = (m + kSortChunk - 1) / kSortChunk;
  for (int pass = 0; pass < 8 && m > 1; ++pass) {
    const int shift = 8 * pass;
    hipLaunchKernelGGL(k_sort_hist, dim3(sb_), dim3(256), 0, st, (const unsigned long long*)d_keyA, m, shift, d_hist, sb_);
    if (int rc = exclusive_scan(ctx, d_hist, d_hist, 256 * sb_, d_sums, ctx->scene_total + 2)) return rc;
    hipLaunchKernelGGL(k_sort_scatter, dim3(sb_), dim3(64), 0, st, (const unsigned long long*)d_keyA, (const int*)d_idxA, m, shift, (const int*)d_hist, sb_, d_keyB, d_idxB);
    std::swap(d_keyA, d_keyB); std::swap(d_idxA, d_idxB);
  }
  const bool withGauss = cloud->sb.G && cloud->n_gauss == n;          // voxelcalculator.cpp:62-64
  if (m > 0) hipLaunchKernelGGL(k_voxel_gather, dim3((m + 255) / 256), dim3(256), 0, st, cloud->d, cloud->sb, cloud->back, cloud->sback, m, withGauss ? 1 : 0, (const int*)d_idxA);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  HIPCHK(ctx, hipMemcpyAsync(cloud->d.count, &m, sizeof(int), hipMemcpyHostToDevice, st), PWN_HIP_ERR_COPY);
  if (kept && m > 0) HIPCHK(ctx, hipMemcpyAsync(kept, d_idxA, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost, st), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(st), PWN_HIP_ERR_LAUNCH);
  swap_back(cloud);
  cloud->n_host = m; cloud->idx_valid = false;
  cloud->n_gauss = withGauss ? m : 0;
  if (new_size) *new_size = m;
  return PWN_HIP_OK;
}

// Cloud::save (cloud.cpp:84-136).  Text records are the reference's, token for token (operator<< on floats).  Binary records keep the
// reference's object layout on x86-64 (Point 32 B, Normal 32 B, Stats 112 B: vptr, padding, data) with the non-data bytes zeroed.
int pwn_hip_cloud_save(pwn_hip_ctx* ctx, const pwn_hip_cloud* c, const char* filename, const float T[16], int step, int binary) {
  if (!ctx || !c || !filename || !T || step <= 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "bad argument");
  const int n = c->n_host;
  std::vector<float> P, Nm, St;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  if (int rc = cloud_fetch_records(ctx, c->d, n, P, Nm)) return rc;
  if (n > 0) {
    if (c->has_stats && c->d.St) { St.resize((size_t)n * 16); HIPCHK(ctx, hipMemcpy(St.data(), c->d.St, St.size() * 4, hipMemcpyDeviceToHost), PWN_HIP_ERR_COPY); }
  }
  std::ofstream os(filename);
  if (!os) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, std::string("cannot open ") + filename);
  os << "PWNCLOUD " << (size_t)n / (size_t)step << " " << (binary ? true : false) << std::endl;
  float tv[6]; t2v(mat4_from(T), tv);
  os << tv[0] << " " << tv[1] << " " << tv[2] << " " << tv[3] << " " << tv[4] << " " << tv[5] << " " << std::endl;
  for (int i = 0; i < n; i += step) {
    float S[16];                                     // Stats as a row/column-indexed 4x4 (column-major here)
    for (int q = 0; q < 16; ++q) S[q] = 0.f;
    int npts = 0; float ev[3] = { 0.f, 0.f, 0.f };
    if (!St.empty()) {
      const float* s = &St[(size_t)16 * i];
      for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) S[r + 4 * q] = s[r + 3 * q];
      S[12] = s[12]; S[13] = s[13]; S[14] = s[14]; S[15] = 1.0f;
      ev[0] = s[9]; ev[1] = s[10]; ev[2] = s[11]; npts = (int)s[15];
    }
    const float* p = &P[(size_t)4 * i]; const float* nm = &Nm[(size_t)4 * i];
    if (!binary) {
      os << "POINTWITHSTATS ";
      for (int k = 0; k < 3; ++k) os << p[k] << " ";
      for (int k = 0; k < 3; ++k) os << nm[k] << " ";
      for (int r = 0; r < 4; ++r) for (int q = 0; q < 4; ++q) os << S[r + 4 * q] << " ";
      os << std::endl;
    } else {
      char rec[176]; std::memset(rec, 0, sizeof(rec));
      const float pw[4] = { p[0], p[1], p[2], 1.0f }, nw[4] = { nm[0], nm[1], nm[2], 0.0f };
      std::memcpy(rec + 16, pw, 16); std::memcpy(rec + 32 + 16, nw, 16);
      char* st = rec + 64;
      std::memcpy(st + 16, S, 64); std::memcpy(st + 80, &npts, 4); std::memcpy(st + 84, ev, 12);
      // Stats::_curvatureComputed / _curvature as DepthImageConverter::compute leaves them: cached where the stats calculator
      // evaluated curvature() (statscalculatorintegralimage.cpp:72), the Stats() defaults (false, 1.0f) where it skipped the pixel;
      // a cloud without Stats (uploaded) carries curvatures set by setCurvature()
      const bool cached = St.empty() || npts > 0;
      st[96] = cached ? 1 : 0; const float curv = cached ? p[3] : 1.0f; std::memcpy(st + 100, &curv, 4);
      os.write(rec, sizeof(rec));
    }
  }
  if (!os.good()) return fail(ctx, PWN_HIP_ERR_COPY, std::string("write error on ") + filename);
  return PWN_HIP_OK;
}
// Cloud::load (cloud.cpp:25-82): points, normals and Stats; the information matrices are not part of the format (they stay zero) and
// the curvature comes from the stored eigenvalues (binary) or is the Stats default (text: eigenvalues are not stored -> 0/(0+1e-9) = 0).
int pwn_hip_cloud_load(pwn_hip_ctx* ctx, pwn_hip_cloud* c, const char* filename, float T_out[16]) {
  if (!ctx || !c || !filename || !T_out) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  std::ifstream is(filename);
  if (!is) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, std::string("cannot open ") + filename);
  char buf[1024];
  is.getline(buf, 1024);
  std::istringstream ls(buf);
  std::string tag; size_t numPoints = 0; bool binary = false;
  ls >> tag;
  if (tag != "PWNCLOUD") return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "not a PWNCLOUD file");
  ls >> numPoints >> binary;
  if (numPoints > (size_t)c->d.capacity) return fail(ctx, PWN_HIP_ERR_CAPACITY, "cloud capacity smaller than the file's point count");
  cloud_changes(ctx, c);
  is.getline(buf, 1024);
  std::istringstream lst(buf);
  float tv[6] = { 0, 0, 0, 0, 0, 0 };
  lst >> tv[0] >> tv[1] >> tv[2] >> tv[3] >> tv[4] >> tv[5];
  const Mat4 T = v2t(tv); std::memcpy(T_out, T.m, sizeof(T.m));
  const int n = (int)numPoints;
  std::vector<float> P((size_t)n * 4, 0.f), Nm((size_t)n * 4, 0.f), St((size_t)n * 16, 0.f);
  for (int i = 0; i < n; ++i) { St[(size_t)16 * i] = 1.f; St[(size_t)16 * i + 4] = 1.f; St[(size_t)16 * i + 8] = 1.f; }
  size_t k = 0;
  while (k < (size_t)n && is.good()) {
    float S[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    float ev[3] = { 0.f, 0.f, 0.f }; int npts = 0; float curv = 0.f; bool haveCurv = false;
    float* p = &P[4 * k]; float* nm = &Nm[4 * k];
    if (!binary) {
      is.getline(buf, 1024);
      std::istringstream l2(buf);
      std::string s2; l2 >> s2;
      if (s2 != "POINTWITHSTATS") continue;
      for (int i = 0; i < 3 && l2; ++i) l2 >> p[i];
      for (int i = 0; i < 3 && l2; ++i) l2 >> nm[i];
      for (int r = 0; r < 4 && l2; ++r) for (int q = 0; q < 4 && l2; ++q) l2 >> S[r + 4 * q];
    } else {
      char rec[176];
      is.read(rec, sizeof(rec));
      std::memcpy(p, rec + 16, 12); std::memcpy(nm, rec + 32 + 16, 12);
      const char* st = rec + 64;
      std::memcpy(S, st + 16, 64); std::memcpy(&npts, st + 80, 4); std::memcpy(ev, st + 84, 12);
      haveCurv = st[96] != 0; std::memcpy(&curv, st + 100, 4);
    }
    if (!haveCurv) curv = (float)((double)ev[0] / ((double)(ev[0] + ev[1] + ev[2]) + 1e-9));     // stats.h:98-103
    p[3] = curv;
    const int cls = 0; std::memcpy(&nm[3], &cls, 4);
    float* s = &St[16 * k];
    for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) s[r + 3 * q] = S[r + 4 * q];
    s[12] = S[12]; s[13] = S[13]; s[14] = S[14]; s[9] = ev[0]; s[10] = ev[1]; s[11] = ev[2]; s[15] = (float)npts;
    ++k;
  }
  const bool ok = is.good();
  if (int rc = scene_alloc(ctx, (void**)&c->d.St, (size_t)c->d.capacity * 16 * sizeof(float))) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  // Cloud::load leaves both information-matrix vectors empty (cloud.cpp:25-82): zero matrices for every point.  The class of the normal
  // information matrix is derived from normal and curvature on the device, so a loaded cloud says "zero" with explicit planes.
  if (int rc = scene_alloc(ctx, (void**)&c->d.OmN, (size_t)c->d.capacity * 9 * sizeof(float))) return rc;
  if (n > 0) {
    if (int rc = cloud_store_records(ctx, c->d, n, P, Nm)) return rc;
    HIPCHK(ctx, hipMemcpy(c->d.St, St.data(), St.size() * 4, hipMemcpyHostToDevice), PWN_HIP_ERR_COPY);
    HIPCHK(ctx, hipMemset(c->d.Om, 0, om_floats(c->d) * sizeof(float)), PWN_HIP_ERR_COPY);
  }
  HIPCHK(ctx, hipMemset(c->d.OmN, 0, (size_t)c->d.capacity * 9 * sizeof(float)), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipMemcpy(c->d.count, &n, sizeof(int), hipMemcpyHostToDevice), PWN_HIP_ERR_COPY);
  c->n_host = n; c->has_stats = true; c->n_gauss = 0; c->idx_valid = false;
  if (!ok) return fail(ctx, PWN_HIP_ERR_COPY, "read error / truncated PWNCLOUD file");
  return PWN_HIP_OK;
}

}  // extern "C"
