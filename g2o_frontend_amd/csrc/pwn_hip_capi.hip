// pwn_hip_capi.hip -- implementation of include/pwn_hip.h: context, device clouds and the launch
// sequences that replace the reference's DepthImageConverterIntegralImage::compute
// (pwn_core/depthimageconverterintegralimage.cpp:15-55) and Aligner::align (pwn_core/aligner.cpp:49-125).
//
// Host code only orchestrates: every per-pixel / per-point / per-correspondence loop of the reference runs
// in a kernel of pwn_kernels.h, and the Gauss-Newton loop runs without host round trips (the 6x6 solve and
// the SE(3) update are a one-wave kernel).  There is no CPU fallback: without a HIP device every entry point
// returns PWN_HIP_ERR_NO_DEVICE.
#include <cstdlib>
#include "../../include/pwn_hip.h"
#include "../../include/pwn_hip_testing.h"
#include "pwn_kernels.h"
#include "pwn_scene_kernels.h"
#include "pwn_stats.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

using namespace pwnhip;

namespace {

thread_local std::string g_err = "";

struct StageAcc { float ms = 0.f; int launches = 0; };
struct EventRec { std::string stage; hipEvent_t a, b; };

// pwn_hip_convert_scaled_begin / _end: a helper context (streams and workspaces of its own) driven by a helper thread, so that a frame
// is converted next to whatever the caller runs on the context meanwhile
struct AsyncConvert {
  pwn_hip_ctx* helper = nullptr;           // default-priority streams: the tracker's look-ahead (a frame converted beside ONE alignment's short kernels)
  pwn_hip_ctx* helper_hi = nullptr;        // high-priority streams, created on first use: a frame converted beside a BATCH call (pwn_hip_convert_export_begin)
  pwn_hip_ctx* job_ctx = nullptr;          // the one the current job runs on
  std::thread worker;
  std::mutex m;
  std::condition_variable cv;
  bool has_job = false, done = true, quit = false;
  pwn_hip_converter_params p;
  const float* depth = nullptr; int rows = 0, cols = 0, step = 1; float max_depth_cov = 0.f;
  const uint16_t* raw = nullptr; float raw_scale = 0.f;      // pwn_hip_convert_export_begin: a raw uint16 frame instead of `depth`
  void* flat_dst = nullptr; size_t flat_bytes = 0, flat_written = 0;      // ... and the cloud's flat form written behind the conversion
  float job_ms = 0.f;                                        // wall time of the job on the helper thread (conversion + export, waits included)
  pwn_hip_cloud* cloud = nullptr;
  int rc = 0; std::string err;
};

}  // namespace

struct pwn_hip_cloud {
  CloudDev d;
  int n_host = 0;
  bool has_stats = false;
  // scene stage (pwn_scene_capi.h): sensor-noise Gaussians, and a second set of arrays for compactions / reorderings
  SceneBuffers sb = { nullptr, nullptr };
  int n_gauss = 0;                       // Cloud::gaussians().size()
  CloudDev back = {};
  SceneBuffers sback = { nullptr, nullptr };
  // The index image DepthImageConverter::compute produced for this cloud (written here directly instead of into a workspace), with what
  // it was made from.  Projecting a cloud with the very projector it was unprojected with (same K, size, range, identity pose) returns
  // this image: every point falls back on its own pixel (the round trip moves it by < 0.01 pixel and its depth not at all), so batch
  // alignments take it as the current index image and skip that projection.  Anything that changes the points invalidates it.
  int* idximg = nullptr; size_t idx_cap = 0; bool idx_valid = false;
  int idx_rows = 0, idx_cols = 0; float idx_K[9] = { 0 }; float idx_minD = 0.f, idx_maxD = 0.f;
};

struct pwn_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  hipStream_t stream2 = nullptr;           // batch calls deal sub-batches round-robin over `stream`, `stream2` and `extra` (own streams only)
  hipStream_t extra[2] = { nullptr, nullptr };
  hipEvent_t fork_ev = nullptr, join_ev = nullptr, join_extra[2] = { nullptr, nullptr };
  hipStream_t copy_stream = nullptr;       // host frames of a batch call are copied on their own stream, one sub-batch ahead of the kernels
  std::vector<hipEvent_t> sync_events;     // ordering events of that hand-over (copied[k], converted[k]); grown on demand
  hipEvent_t copy_ev = nullptr;            // pwn_hip_copy_async: the next call that reads frames waits for the copies issued so far
  hipEvent_t foreign_ev = nullptr;         // pwn_hip_ctx_wait_stream: marks the caller's stream
  bool copy_pending = false;
  int max_rows = 0, max_cols = 0, max_batch = 0;
  int num_cus = 256;                       // compute units of the device (hipDeviceAttributeMultiprocessorCount)
  size_t N = 0;
  int sub_frames = 64, sub_pairs = 64;
  int concurrency = 4;
  int omega_sym = 1;                       // pwn_hip_ctx_set_omega_storage: storage of the point information matrices of clouds created from now on (default: sym6)
  // convert workspaces (per slot)
  float* depth_ws = nullptr; int* index_ws = nullptr; int* interval_ws = nullptr; float* integral_ws = nullptr; int* rowoff_ws = nullptr;
  uint16_t* raw_ws = nullptr;
  unsigned long long* carry_ws = nullptr; size_t carry_slot = 0; size_t rowoff_slot = 0;   // single-pass integral image: hand-over words, strip offsets
  unsigned convert_epoch = 0; int* fault_dev = nullptr;
  int last_convert_fault = 0;              // fault word of the last converter launch of the CURRENT call (1 = a bounded poll timed out); reset when a call starts
  int spin_limit = kSpinLimit; int dbg_withhold = -1;        // pwn_hip_debug_withhold_carry (test hook)
  int dbg_withhold_once = 0;                                 // the hook switches itself off after the first launch that timed out
  int convert_retries = 0;                                   // conversions repeated after a hand-over time-out (pwn_hip_debug_convert_retries)
  bool in_step_retry = false;
  // align workspaces (per slot)
  unsigned long long* zref_ws = nullptr;        // 64-bit z-buffer of the stand-alone projection and of Merger::merge (one image)
  unsigned* z32ref_ws = nullptr; unsigned* z32cur_ws = nullptr;      // the aligner's 32-bit z-buffers (tag | index), one image per slot
  int* curidx_ws = nullptr; double* partials_ws = nullptr; PairState* state_ws = nullptr;
  int nblocks_max = 0;
  // descriptors (one entry per frame / pair of a batch call; grown on demand)
  int desc_cap = 0;
  FrameDesc* frames_dev = nullptr; PairDesc* pairs_dev = nullptr; RawDesc* raw_dev = nullptr; int* counts_dev = nullptr;
  FrameDesc* frames_host = nullptr; PairDesc* pairs_host = nullptr; RawDesc* raw_host = nullptr; PairState* state_host = nullptr; int* counts_host = nullptr;
  // misc scratch
  MatchAcc* match_dev = nullptr; MatchAcc* match_host = nullptr; int match_cap = 0;
  SolveOut* stats_dev = nullptr; SolveOut* stats_host = nullptr;
  SolveOut* solve_dev = nullptr; int* counters_dev = nullptr; int2* corr_ws = nullptr; int* scratch_count = nullptr;
  float* io_ws = nullptr;   // N*16 floats staging for cloud up/download
  // images of the last single align
  int img_rows = 0, img_cols = 0; bool img_valid = false; unsigned img_ref_tag = kZ32Tag0, img_cur_tag = kZ32Tag0;
  int img_pair = 0;                         // descriptor (pairs_host / pairs_dev entry) of the pair whose images sit in workspace slot 0
  // A single alignment whose current cloud carries its own index image (pwn_hip_cloud::idximg) does not project that cloud at all; its
  // z-buffer is filled in only when pwn_hip_align_images asks for the finder's current images (pwn_hip_match_score reads the depths off the cloud)
  bool img_cur_lazy = false; AlignParams img_ap; int img_cur_capacity = 0;
  int index_shortcut = 1;                   // pwn_hip_debug_set_index_shortcut (test hook): 0 = always project
  const pwn_hip_cloud* img_ref_cloud = nullptr; const pwn_hip_cloud* img_cur_cloud = nullptr;      // its clouds: the depth images are recomputed from their points
  // z-buffer epoch tags are handed out in descending order ACROSS batch calls (a smaller tag wins, so whatever earlier calls left in
  // the buffers reads as empty): the buffers are cleared only when the 12-bit tag space is used up, not once per alignment
  unsigned ztag_next = 0;                   // 64-bit buffer (scene stage)
  unsigned z32tag_next = 0;                 // 32-bit buffers (aligner)
  std::string err;
  bool profiling = false;
  std::map<std::string, StageAcc> stages;
  std::vector<EventRec> pending;
  std::vector<hipEvent_t> event_pool;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  // scene-stage scratch (grown on demand)
  int* scene_i[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr }; size_t scene_icap = 0;
  unsigned long long* scene_k[3] = { nullptr, nullptr, nullptr }; size_t scene_kcap = 0;
  int* scene_total = nullptr;
  // Retired clouds kept for reuse: pwn_hip_cloud_create / _destroy are called once per frame by makeCloud-style callers (the reference
  // returns a `new Cloud` per depth image, pwn_matcher_base.cpp:77-85), and hipMalloc / hipFree of the point arrays cost more than the
  // conversion of a frame.  Bounded by kCloudPoolBytes.
  std::vector<pwn_hip_cloud*> cloud_pool; size_t cloud_pool_bytes = 0;
  AsyncConvert* async = nullptr;           // pwn_hip_convert_scaled_begin: created on first use
  void* flat_hdr_host = nullptr;           // page-locked staging of a flat cloud's 256-byte header (pwn_hip_cloud_export / _import)
  float* records_ws = nullptr; int records_cap = 0;      // result records of a batch on their way to host memory (k_pack_records)
  int* ids_dev = nullptr; int ids_cap = 0;               // the caller's pair ids of those records
  // Projection fault path (z32_settle): a page-locked word the projection kernels raise when a pixel's settle loop gave up; the call is then
  // repeated with the two-pass projection (k_project_robust) and its depth images (allocated on first use)
  void (*enqueued_cb)(void*) = nullptr; void* enqueued_user = nullptr;      // pwn_hip_ctx_set_enqueued_callback
  int* align_fault_host = nullptr;
  unsigned* zdepth_ws = nullptr;
  int settle_guard = kSettleGuard;                       // pwn_hip_debug_set_settle_guard (test hook)
  int last_align_fault = 0;                              // the last alignment call ended with the fault word raised
  int projection_fallbacks = 0;                          // calls repeated with the two-pass projection so far (pwn_hip_debug_projection_fallbacks)
};

namespace {

constexpr size_t kCloudPoolBytes = 1ull << 30;
size_t om_floats(const CloudDev& d) { return (size_t)d.capacity * 3 * (size_t)om_planes(d.omSym); }      // floats of the Om planes (9 or 6 per point)
size_t cloud_core_bytes(const pwn_hip_cloud* c) { return (size_t)c->d.capacity * (3 * sizeof(float) + sizeof(float4)) + om_floats(c->d) * sizeof(float) + c->idx_cap * sizeof(int); }
void cloud_free(pwn_hip_cloud* c) {
  void* p[] = { c->d.P3, c->d.Nc, c->d.Om, c->d.OmN, c->d.St, c->d.count, c->sb.G, c->sb.Gf,
                c->back.P3, c->back.Nc, c->back.Om, c->back.OmN, c->back.St, c->sback.G, c->sback.Gf, c->idximg };
  for (void* q : p) if (q) (void)hipFree(q);
  delete c;
}

// the fused correspondence + linearize pass: the throughput shape, or the latency shape (same sums bit for bit, see k_corr_linearize_lat)
// when all workgroups of the launch find a CU of their own -- its 1024-thread workgroups fit one per CU, a second round costs more than
// the shape saves: one VGA pair is 150 workgroups, two pairs or one 1280x960 pair are not worth it on 256 CUs
// sym: storage of the current clouds' point information matrices (CloudDev::omSym; the same for every pair of the launch)
template <bool SAME_T, bool FULL_H, bool SYM>
void launch_corr_linearize_s(const pwn_hip_ctx* ctx, int nb, int m, hipStream_t st, const PairDesc* pr, const AlignParams& ap, unsigned tag, int usePrevTc, int ownRef) {
  if ((long long)nb * m <= ctx->num_cus) hipLaunchKernelGGL((k_corr_linearize_lat<SAME_T, FULL_H, SYM>), dim3(nb, m), dim3(kLatBlock), 0, st, pr, ap, tag, usePrevTc, ownRef);
  else hipLaunchKernelGGL((k_corr_linearize<SAME_T, FULL_H, SYM>), dim3(nb, m), dim3(kAlignBlock), 0, st, pr, ap, tag, usePrevTc, ownRef);
}
template <bool SAME_T, bool FULL_H>
void launch_corr_linearize(const pwn_hip_ctx* ctx, int sym, int nb, int m, hipStream_t st, const PairDesc* pr, const AlignParams& ap, unsigned tag, int usePrevTc, int ownRef) {
  if (sym) launch_corr_linearize_s<SAME_T, FULL_H, true>(ctx, nb, m, st, pr, ap, tag, usePrevTc, ownRef);
  else launch_corr_linearize_s<SAME_T, FULL_H, false>(ctx, nb, m, st, pr, ap, tag, usePrevTc, ownRef);
}

int fail(pwn_hip_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  g_err = msg;
  return code;
}
// The finder's depth images of the last alignment are recomputed from the clouds' points (the z-buffer keeps indices only): anything
// that changes or frees one of those clouds ends the validity of pwn_hip_align_images / pwn_hip_match_score for that alignment.
void cloud_changes(pwn_hip_ctx* ctx, const pwn_hip_cloud* c) {
  if (ctx && c && (c == ctx->img_ref_cloud || c == ctx->img_cur_cloud)) { ctx->img_valid = false; ctx->img_ref_cloud = nullptr; ctx->img_cur_cloud = nullptr; }
}
#define HIPCHK(ctx, call, code)                                                                               \
  do {                                                                                                        \
    hipError_t e_ = (call);                                                                                   \
    if (e_ != hipSuccess) return fail(ctx, code, std::string(#call) + ": " + hipGetErrorString(e_));           \
  } while (0)

// projection of one cloud of each of the m pairs (which: 0 = reference, 1 = current): four points per thread when the launch is large
constexpr int kProjectPPT = 4;
// zd: the depth images of the m pairs' workspace slots (contiguous, ctx->N words each) when the call runs the two-pass projection, else nullptr
int launch_project(pwn_hip_ctx* ctx, int capacity, int m, hipStream_t st, const PairDesc* pr, const AlignParams& ap, int which, unsigned tag, unsigned* zd = nullptr) {
  if (zd) {
    HIPCHK(ctx, hipMemsetAsync(zd, 0xFF, (size_t)m * ctx->N * sizeof(unsigned), st), PWN_HIP_ERR_COPY);
    for (int pass = 0; pass < 2; ++pass)
      hipLaunchKernelGGL(k_project_robust, dim3((capacity + 255) / 256, m), dim3(256), 0, st, pr, ap, which, tag, pass);
    return PWN_HIP_OK;
  }
  if (m >= 8) hipLaunchKernelGGL((k_project<kProjectPPT>), dim3((capacity + 256 * kProjectPPT - 1) / (256 * kProjectPPT), m), dim3(256), 0, st, pr, ap, which, tag);
  else hipLaunchKernelGGL((k_project<1>), dim3((capacity + 255) / 256, m), dim3(256), 0, st, pr, ap, which, tag);
  return PWN_HIP_OK;
}

bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}
// copy helper: any combination of host/device
hipError_t copy_any(void* dst, const void* src, size_t bytes, hipStream_t s) {
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, s);
}

// device-to-device section copies of the flat cloud: a kernel (hipMemcpyAsync between device buffers is a blit kernel of the runtime's with a
// launch of its own per call: 0.3 ms per 17 MB cloud measured, against ~10 us); sizes and offsets are multiples of 4 bytes.
// The grid is SMALL on purpose (at most kCopyBlocks workgroups, four independent 16-byte accesses in flight per thread: ~2 TB/s alone, a 5 MB
// section in a few microseconds): these copies run beside a batch alignment (export in the look-ahead job, import on a second context), where a
// grid that fills the device (2048 workgroups until round 6) displaced the batch's own workgroups for ~80 us per section -- measured as +2.5 % on the
// sum of the batch's kernel times for 0.25 % more bytes.
constexpr unsigned kCopyBlocks = 96;
template <typename V> __global__ void __launch_bounds__(256) k_copy_words(const V* __restrict__ src, V* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const V a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n; i += stride) dst[i] = src[i];
}
hipError_t copy_section(void* dst, const void* src, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  if (!is_device_ptr(dst) || !is_device_ptr(src) || (bytes & 3) || (((uintptr_t)dst | (uintptr_t)src) & 3)) return copy_any(dst, src, bytes, st);
  if (((((uintptr_t)dst | (uintptr_t)src) | bytes) & 15) == 0) {
    const size_t n = bytes / 16;
    hipLaunchKernelGGL(k_copy_words<uint4>, dim3((unsigned)std::min<size_t>((n + 1023) / 1024, kCopyBlocks)), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, n);
  } else {
    const size_t n = bytes / 4;
    hipLaunchKernelGGL(k_copy_words<unsigned>, dim3((unsigned)std::min<size_t>((n + 1023) / 1024, 2 * kCopyBlocks)), dim3(256), 0, st, (const unsigned*)src, (unsigned*)dst, n);
  }
  return hipGetLastError();
}

hipEvent_t get_event(pwn_hip_ctx* ctx) {
  if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
  // timing-only events: no system-scope fence (no cache write-back / invalidate between the kernels they bracket)
  hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence); return e;
}
struct StageTimer {
  pwn_hip_ctx* ctx; EventRec rec; bool on; hipStream_t st;
  StageTimer(pwn_hip_ctx* c, const char* stage, hipStream_t s = nullptr) : ctx(c), on(c->profiling), st(s ? s : c->stream) {
    if (on) { rec.stage = stage; rec.a = get_event(ctx); rec.b = get_event(ctx); (void)hipEventRecord(rec.a, st); }
  }
  ~StageTimer() { if (on) { (void)hipEventRecord(rec.b, st); ctx->pending.push_back(rec); } }
};
// Two-stream mode: only with the context's own streams and when the workspaces hold two sub-batches.
struct StreamPlan {
  int sub; int ns; hipStream_t s[4];
  bool dual() const { return ns > 1; }
  hipStream_t stream(int k) const { return s[k % ns]; }
  int slot0(int k) const { return (k % ns) * sub; }
};
StreamPlan make_plan(pwn_hip_ctx* ctx, int want_sub, int n) {
  StreamPlan p;
  p.sub = std::max(1, std::min(want_sub, ctx->max_batch));
  // A batch is cut into a multiple of `streams` sub-batches of equal size (not larger than asked for), so that no stream idles while the other
  // works: the short dependent kernels of one sub-batch (projection, 6x6 step) fill the gaps of the other's large ones.  Measured on MI355X
  // (one-submission step, two streams): 64 VGA pairs as 2 x 32 instead of 1 x 64: 11 340 -> 12 170 alignments/s; 32 pairs as 2 x 16: 10 620 ->
  // 11 440; 32 pairs of 1280x960 as 2 x 16: 2 900 -> 3 090 (docs/experiments.md, round 4).  A call is only cut when every part keeps at least 16
  // items (the converter's single-pass kernels start there: kSinglePassMinFrames); smaller calls stay one launch sequence.
  if (ctx->stream == ctx->own_stream && ctx->stream2 && ctx->concurrency >= 2) {
    const int kmax = std::min(std::min(ctx->concurrency, 4), (!ctx->extra[0] ? 2 : (!ctx->extra[1] ? 3 : 4)));
    for (int k = kmax; k >= 2; --k) {      // as many streams as leave every part 16 items
      const int nsub_even = k * ((n + k * p.sub - 1) / (k * p.sub));
      if (nsub_even > 0 && n / nsub_even >= 16) { p.sub = (n + nsub_even - 1) / nsub_even; break; }
    }
  }
  const int nsub = (n + p.sub - 1) / p.sub;
  int ns = 1;
  if (ctx->stream == ctx->own_stream && ctx->stream2)
    ns = std::max(1, std::min(std::min(ctx->concurrency, 4), std::min(nsub, ctx->max_batch / p.sub)));
  if (ns > 2 && (!ctx->extra[0] || (ns > 3 && !ctx->extra[1]))) ns = 2;
  p.ns = ns;
  p.s[0] = ctx->stream; p.s[1] = ctx->stream2; p.s[2] = ctx->extra[0]; p.s[3] = ctx->extra[1];
  return p;
}
int plan_fork(pwn_hip_ctx* ctx, const StreamPlan& p) {      // the other streams start after everything enqueued so far on `stream`
  if (!p.dual()) return PWN_HIP_OK;
  HIPCHK(ctx, hipEventRecord(ctx->fork_ev, p.s[0]), PWN_HIP_ERR_LAUNCH);
  for (int k = 1; k < p.ns; ++k) HIPCHK(ctx, hipStreamWaitEvent(p.s[k], ctx->fork_ev, 0), PWN_HIP_ERR_LAUNCH);
  return PWN_HIP_OK;
}
int plan_join(pwn_hip_ctx* ctx, const StreamPlan& p) {      // `stream` continues after the other streams' work
  if (!p.dual()) return PWN_HIP_OK;
  for (int k = 1; k < p.ns; ++k) {
    hipEvent_t ev = k == 1 ? ctx->join_ev : ctx->join_extra[k - 2];
    HIPCHK(ctx, hipEventRecord(ev, p.s[k]), PWN_HIP_ERR_LAUNCH);
    HIPCHK(ctx, hipStreamWaitEvent(p.s[0], ev, 0), PWN_HIP_ERR_LAUNCH);
  }
  return PWN_HIP_OK;
}
// pwn_hip_copy_async: everything queued on the context's stream from here on runs after the copies issued so far (the other streams of a
// batch call fork from that stream)
int absorb_copies(pwn_hip_ctx* ctx) {
  if (!ctx->copy_pending) return PWN_HIP_OK;
  HIPCHK(ctx, hipEventRecord(ctx->copy_ev, ctx->copy_stream), PWN_HIP_ERR_LAUNCH);
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->copy_ev, 0), PWN_HIP_ERR_LAUNCH);
  ctx->copy_pending = false;
  return PWN_HIP_OK;
}
void collect_stage_times(pwn_hip_ctx* ctx) {
  for (auto& r : ctx->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { auto& s = ctx->stages[r.stage]; s.ms += ms; s.launches += 1; }
    ctx->event_pool.push_back(r.a); ctx->event_pool.push_back(r.b);
  }
  ctx->pending.clear();
}

Mat4 forced(const float* T) { Mat4 m = mat4_from(T); set_last_row(m); return m; }
bool is_identity(const Mat4& m) { const Mat4 I = mat4_identity(); for (int i = 0; i < 16; ++i) if (m.m[i] != I.m[i]) return false; return true; }

ConvertParams make_convert_params(const pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float* T, int rows, int cols, int keep_stats) {
  ConvertParams cp;
  cp.spinLimit = ctx ? ctx->spin_limit : kSpinLimit; cp.dbgWithhold = ctx ? ctx->dbg_withhold : -1;
  cp.rows = rows; cp.cols = cols;
  const Mat3 K = mat3_from(p->K);
  Mat4 KRt; Mat3 iK;
  projector_matrices(K, T ? mat4_from(T) : mat4_identity(), KRt, cp.iKRt, iK);
  // _projectInterval: p = K * (R, R, 0)   (pinholepointprojector.h:269)
  const float R = p->world_radius;
  cp.ivx = dot3seq(K(0,0), R, K(0,1), R, K(0,2), 0.f);
  cp.ivy = dot3seq(K(1,0), R, K(1,1), R, K(1,2), 0.f);
  cp.minD = p->min_distance; cp.maxD = p->max_distance;
  cp.minRadius = p->min_image_radius; cp.maxRadius = p->max_image_radius; cp.minPoints = p->min_points;
  cp.statsCurvThr = p->stats_curvature_threshold;
  cp.pointInfoCurvThr = p->point_info_curvature_threshold; cp.normalInfoCurvThr = p->normal_info_curvature_threshold;
  for (int i = 0; i < 3; ++i) { cp.pFlat[i] = p->point_flat_diag[i]; cp.pNonFlat[i] = p->point_nonflat_diag[i]; }
  cp.offset = forced(p->sensor_offset);
  cp.hasOffset = is_identity(cp.offset) ? 0 : 1;
  cp.keepStats = keep_stats;
  cp.lean = 0;
  cp.omSym = 0;        // set per call from the clouds being written (convert_batch_impl)
  return cp;
}
// class matrices of the normal information matrix, after Cloud::transformInPlace (T * Omega * T^t, informationmatrix.h:111-121)
void make_omega_n_classes(const pwn_hip_converter_params* p, CloudDev& d) {
  const Mat4 m = forced(p->sensor_offset);
  const bool ident = is_identity(m);
  for (int c = 0; c < 2; ++c) {
    const float* dg = c == 0 ? p->normal_flat_diag : p->normal_nonflat_diag;
    float om[9] = { dg[0], 0, 0, 0, dg[1], 0, 0, 0, dg[2] };
    if (!ident) {
      float t1[9];
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t1[3 * i + j] = dot3seq(m(i,0), om[0 + j], m(i,1), om[3 + j], m(i,2), om[6 + j]);
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) om[3 * i + j] = dot3seq(t1[3 * i], m(j,0), t1[3 * i + 1], m(j,1), t1[3 * i + 2], m(j,2));
    }
    for (int k = 0; k < 9; ++k) d.omN[c][k] = om[k];
  }
  d.clsThr = p->normal_info_curvature_threshold;      // what normal_class() needs to tell flat from non-flat
}
// Host staging in the records the rest of the host code was written for: P = (x, y, z, curvature), Nm = (nx, ny, nz, class word), n
// points each; the device keeps 12-byte points and (normal, curvature) records (pwn_kernels.h).
int cloud_fetch_records(pwn_hip_ctx* ctx, const CloudDev& d, int n, std::vector<float>& P, std::vector<float>& Nm) {
  std::vector<float> p3((size_t)n * 3), nc((size_t)n * 4);
  if (n > 0) {
    HIPCHK(ctx, hipMemcpy(p3.data(), d.P3, p3.size() * 4, hipMemcpyDeviceToHost), PWN_HIP_ERR_COPY);
    HIPCHK(ctx, hipMemcpy(nc.data(), d.Nc, nc.size() * 4, hipMemcpyDeviceToHost), PWN_HIP_ERR_COPY);
  }
  P.resize((size_t)n * 4); Nm.resize((size_t)n * 4);
  for (int i = 0; i < n; ++i) {
    P[4 * i] = p3[3 * i]; P[4 * i + 1] = p3[3 * i + 1]; P[4 * i + 2] = p3[3 * i + 2]; P[4 * i + 3] = nc[4 * i + 3];
    Nm[4 * i] = nc[4 * i]; Nm[4 * i + 1] = nc[4 * i + 1]; Nm[4 * i + 2] = nc[4 * i + 2];
    const int cls = normal_class(nc[4 * i], nc[4 * i + 1], nc[4 * i + 2], nc[4 * i + 3], d.clsThr);
    std::memcpy(&Nm[4 * i + 3], &cls, 4);
  }
  return PWN_HIP_OK;
}
int cloud_store_records(pwn_hip_ctx* ctx, const CloudDev& d, int n, const std::vector<float>& P, const std::vector<float>& Nm) {
  if (n <= 0) return PWN_HIP_OK;
  std::vector<float> p3((size_t)n * 3), nc((size_t)n * 4);
  for (int i = 0; i < n; ++i) {
    p3[3 * i] = P[4 * i]; p3[3 * i + 1] = P[4 * i + 1]; p3[3 * i + 2] = P[4 * i + 2];
    nc[4 * i] = Nm[4 * i]; nc[4 * i + 1] = Nm[4 * i + 1]; nc[4 * i + 2] = Nm[4 * i + 2]; nc[4 * i + 3] = P[4 * i + 3];
  }
  HIPCHK(ctx, hipMemcpy(d.P3, p3.data(), p3.size() * 4, hipMemcpyHostToDevice), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipMemcpy(d.Nc, nc.data(), nc.size() * 4, hipMemcpyHostToDevice), PWN_HIP_ERR_COPY);
  return PWN_HIP_OK;
}
AlignParams make_align_params(const pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p) {
  AlignParams ap;
  ap.settleGuard = ctx ? ctx->settle_guard : kSettleGuard;
  ap.rows = p->rows; ap.cols = p->cols;
  ap.K = mat3_from(p->K);
  ap.refOffset = mat4_from(p->reference_sensor_offset);
  ap.minD = p->min_distance; ap.maxD = p->max_distance;
  ap.sqDist = p->inlier_distance_threshold * p->inlier_distance_threshold;
  ap.normalThr = p->inlier_normal_angular_threshold;
  ap.flatThr = p->flat_curvature_threshold;
  ap.minRatio = 1.0f / p->inlier_curvature_ratio_threshold;
  ap.maxRatio = p->inlier_curvature_ratio_threshold;
  ap.maxChi2 = p->inlier_max_chi2;
  ap.robust = p->robust_kernel;
  return ap;
}

int check_image(pwn_hip_ctx* ctx, int rows, int cols) {
  if (rows <= 0 || cols <= 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "image has zero size");
  // both sides are bounded: the hand-over / offset workspaces (rowoff_slot, carry_slot in pwn_hip_ctx_create) are sized for
  // rows, cols <= max(max_rows, max_cols); a wide, short image with rows*cols <= N would overrun them
  const int M = std::max(ctx->max_rows, ctx->max_cols);
  if ((size_t)rows * cols > ctx->N || rows > M || cols > M)
    return fail(ctx, PWN_HIP_ERR_CAPACITY, "image larger than the context was created for");
  return PWN_HIP_OK;
}
// first (largest) of `need` consecutive descending tags of the 64-bit z-buffer (scene stage); clears it when the tag space is used up
int take_tags(pwn_hip_ctx* ctx, unsigned need, unsigned* first) {
  if (need > kZTag0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "more projections than z-buffer epoch tags");
  if (ctx->ztag_next < need) {
    HIPCHK(ctx, hipMemsetAsync(ctx->zref_ws, 0xFF, ctx->N * 8, ctx->stream), PWN_HIP_ERR_COPY);
    ctx->ztag_next = kZTag0;
  }
  *first = ctx->ztag_next;
  ctx->ztag_next -= need;
  return PWN_HIP_OK;
}
// the same for the aligner's 32-bit z-buffers (11-bit tags): clears every slot of both when the tag space is used up
int take_tags32(pwn_hip_ctx* ctx, unsigned need, unsigned* first) {
  if (need > kZ32Tag0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "more projections per alignment than z-buffer epoch tags");
  if (ctx->z32tag_next < need) {
    HIPCHK(ctx, hipMemsetAsync(ctx->z32ref_ws, 0xFF, (size_t)ctx->max_batch * ctx->N * 4, ctx->stream), PWN_HIP_ERR_COPY);
    HIPCHK(ctx, hipMemsetAsync(ctx->z32cur_ws, 0xFF, (size_t)ctx->max_batch * ctx->N * 4, ctx->stream), PWN_HIP_ERR_COPY);
    ctx->z32tag_next = kZ32Tag0;
  }
  *first = ctx->z32tag_next;
  ctx->z32tag_next -= need;
  return PWN_HIP_OK;
}
// the two-pass projection's depth images: one per workspace slot, allocated when a call first needs them
int ensure_zdepth(pwn_hip_ctx* ctx) {
  if (!ctx->zdepth_ws) HIPCHK(ctx, hipMalloc((void**)&ctx->zdepth_ws, (size_t)ctx->max_batch * ctx->N * sizeof(unsigned)), PWN_HIP_ERR_ALLOCATION);
  return PWN_HIP_OK;
}
// after the wait that ends an alignment call: did a projection of it give up on a pixel?  (the word is host memory the kernels store to)
bool take_align_fault(pwn_hip_ctx* ctx) {
  const bool f = ctx->align_fault_host && *(volatile int*)ctx->align_fault_host != 0;
  if (f) *(volatile int*)ctx->align_fault_host = 0;
  ctx->last_align_fault = f ? 1 : 0;
  return f;
}
const char* const kSettleMessage = "projection: a pixel's z-buffer settle loop gave up (too many points of one cloud in one pixel)";
int align_nblocks(int N) { return (N + kAlignBlock * kPixPerThread - 1) / (kAlignBlock * kPixPerThread); }

// descriptor / state arrays for a batch of n items
int ensure_desc(pwn_hip_ctx* ctx, int n) {
  if (n <= ctx->desc_cap) return PWN_HIP_OK;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  void* dev[] = { ctx->frames_dev, ctx->pairs_dev, ctx->raw_dev, ctx->counts_dev, ctx->state_ws };
  for (void* p : dev) if (p) (void)hipFree(p);
  void* host[] = { ctx->frames_host, ctx->pairs_host, ctx->raw_host, ctx->state_host, ctx->counts_host };
  for (void* p : host) if (p) (void)hipHostFree(p);
  ctx->frames_dev = nullptr; ctx->pairs_dev = nullptr; ctx->raw_dev = nullptr; ctx->counts_dev = nullptr; ctx->state_ws = nullptr;
  ctx->frames_host = nullptr; ctx->pairs_host = nullptr; ctx->raw_host = nullptr; ctx->state_host = nullptr; ctx->counts_host = nullptr;
  ctx->desc_cap = 0;
  ctx->img_valid = false;                    // the descriptor of the last alignment's pair goes with the old arrays
  const size_t B = (size_t)std::max(n, 16);
  HIPCHK(ctx, hipMalloc((void**)&ctx->frames_dev, B * sizeof(FrameDesc)), PWN_HIP_ERR_ALLOCATION);
  HIPCHK(ctx, hipMalloc((void**)&ctx->pairs_dev, B * sizeof(PairDesc)), PWN_HIP_ERR_ALLOCATION);
  HIPCHK(ctx, hipMalloc((void**)&ctx->raw_dev, B * sizeof(RawDesc)), PWN_HIP_ERR_ALLOCATION);
  HIPCHK(ctx, hipMalloc((void**)&ctx->counts_dev, (B + 1) * sizeof(int)), PWN_HIP_ERR_ALLOCATION);
  HIPCHK(ctx, hipMalloc((void**)&ctx->state_ws, B * sizeof(PairState)), PWN_HIP_ERR_ALLOCATION);
  HIPCHK(ctx, hipHostMalloc((void**)&ctx->frames_host, B * sizeof(FrameDesc)), PWN_HIP_ERR_ALLOCATION);
  HIPCHK(ctx, hipHostMalloc((void**)&ctx->pairs_host, B * sizeof(PairDesc)), PWN_HIP_ERR_ALLOCATION);
  HIPCHK(ctx, hipHostMalloc((void**)&ctx->raw_host, B * sizeof(RawDesc)), PWN_HIP_ERR_ALLOCATION);
  HIPCHK(ctx, hipHostMalloc((void**)&ctx->state_host, B * sizeof(PairState)), PWN_HIP_ERR_ALLOCATION);
  HIPCHK(ctx, hipHostMalloc((void**)&ctx->counts_host, (B + 1) * sizeof(int)), PWN_HIP_ERR_ALLOCATION);
  if (ctx->match_dev) (void)hipFree(ctx->match_dev);
  if (ctx->match_host) (void)hipHostFree(ctx->match_host);
  ctx->match_dev = nullptr; ctx->match_host = nullptr;
  HIPCHK(ctx, hipMalloc((void**)&ctx->match_dev, B * sizeof(MatchAcc)), PWN_HIP_ERR_ALLOCATION);
  HIPCHK(ctx, hipHostMalloc((void**)&ctx->match_host, B * sizeof(MatchAcc)), PWN_HIP_ERR_ALLOCATION);
  if (ctx->stats_dev) (void)hipFree(ctx->stats_dev);
  if (ctx->stats_host) (void)hipHostFree(ctx->stats_host);
  ctx->stats_dev = nullptr; ctx->stats_host = nullptr;
  HIPCHK(ctx, hipMalloc((void**)&ctx->stats_dev, B * sizeof(SolveOut)), PWN_HIP_ERR_ALLOCATION);
  HIPCHK(ctx, hipHostMalloc((void**)&ctx->stats_host, B * sizeof(SolveOut)), PWN_HIP_ERR_ALLOCATION);
  ctx->desc_cap = (int)B;
  return PWN_HIP_OK;
}

// launch sequence of the converter for the frames [base, base+n) of the uploaded descriptor array
// fault_out: page-locked host word for the launch's time-out flag (latency path only; nullptr = the caller copies ctx->fault_dev itself)
// launches of at least this many frames take the single-pass strip kernel; measured on MI355X at VGA, three kernels vs single pass: 8 frames 0.20 vs
// 0.27 ms, 16 frames 0.38 vs 0.35, 32 frames 0.74 vs 0.58
constexpr int kSinglePassMinFrames = 16;
int launch_convert(pwn_hip_ctx* ctx, const ConvertParams& cp, int base, int n, hipStream_t st, int* fault_out = nullptr) {
  const FrameDesc* fr = ctx->frames_dev + base;
  if (n >= kSinglePassMinFrames) {
    // throughput path: the integral planes are written once; a frame is a chain of strips * bands hand-over steps, so it
    // needs several frames in flight to fill the device
    { StageTimer t(ctx, "unproject", st);          // ordered compaction: valid pixels per (row, strip) and their offsets
      // frames of one call are all raw uint16 or all float (convert_batch_impl); 16-byte loads need 16-byte aligned rows
      const bool raw = ctx->frames_host[base].raw != nullptr;
      bool aligned = cp.cols % (raw ? 8 : 4) == 0;
      for (int i = 0; i < n && aligned; ++i) {
        const FrameDesc& fd = ctx->frames_host[base + i];
        aligned = ((uintptr_t)(raw ? (const void*)fd.raw : (const void*)fd.depth) & 15u) == 0;
      }
      if (aligned && raw) hipLaunchKernelGGL(k_strip_count<true>, dim3((cp.rows + 3) / 4, n), dim3(256), 0, st, fr, cp);
      else if (aligned) hipLaunchKernelGGL(k_strip_count<false>, dim3((cp.rows + 3) / 4, n), dim3(256), 0, st, fr, cp);
      else hipLaunchKernelGGL(k_strip_count_any, dim3(cp.rows, n), dim3(256), 0, st, fr, cp);
      hipLaunchKernelGGL(k_row_offsets, dim3(n), dim3(1024), 0, st, fr, cp.rows * strips_of(cp.cols)); }
    { StageTimer t(ctx, "integral", st);           // unProject + intervals + the three integral-image passes
      const unsigned epoch = ++ctx->convert_epoch;
      if (epoch == 0) return fail(ctx, PWN_HIP_ERR_LAUNCH, "convert epoch wrapped: recreate the context");
      hipLaunchKernelGGL(k_unproject_integral, dim3(8u * (unsigned)((n + 7) / 8) * (unsigned)strips_of(cp.cols)), dim3(kII_Threads), 0, st, fr, cp, n,
                         epoch, ctx->fault_dev); }
  } else {
    // latency path (single frames: tracker, makeCloud): three short, fully parallel kernels
    { StageTimer t(ctx, "unproject", st);          // ordered compaction: per-row counts and offsets
      hipLaunchKernelGGL(k_row_count, dim3(cp.rows, n), dim3(256), 0, st, fr, cp);
      hipLaunchKernelGGL(k_row_offsets, dim3(n), dim3(1024), 0, st, fr, cp.rows); }
    { StageTimer t(ctx, "integral_rows", st);      // unProject + intervals + accumulate + row prefix, one pass over the depth
      const unsigned epoch = ++ctx->convert_epoch;
      if (epoch == 0) return fail(ctx, PWN_HIP_ERR_LAUNCH, "convert epoch wrapped: recreate the context");
      hipLaunchKernelGGL(k_unproject_integral_rows, dim3(8u * (unsigned)((bands_of(cp.rows) + 7) / 8) * (unsigned)strips_of(cp.cols), n), dim3(256), 0, st, fr, cp,
                         epoch, ctx->fault_dev); }
    { StageTimer t(ctx, "integral_cols", st);
      hipLaunchKernelGGL(k_integral_cols, dim3((cp.cols + kIC_Block - 1) / kIC_Block, kIntegralChannels, n), dim3(kIC_Block), 0, st, fr, cp.rows, cp.cols,
                         (const int*)ctx->fault_dev, fault_out); }
  }
  { StageTimer t(ctx, "stats", st);
    const unsigned perFrame = (unsigned)cp.rows * (unsigned)((cp.cols + 255) / 256);
    const unsigned nblk = (n >= 8 ? 8u * (unsigned)((n + 7) / 8) : (unsigned)n) * perFrame;      // see k_stats: XCD-aware placement from 8 frames on
    hipLaunchKernelGGL(k_stats, dim3(nblk), dim3(256), 0, st, fr, cp, n); }
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  return PWN_HIP_OK;
}
void fill_frame(pwn_hip_ctx* ctx, int entry, int slot, const float* depth_dev, const CloudDev& cl, int rows) {
  FrameDesc& f = ctx->frames_host[entry];
  f.depth = depth_dev; f.raw = nullptr; f.raw_scale = 0.f;
  f.index = ctx->index_ws + (size_t)slot * ctx->N;
  f.interval = ctx->interval_ws + (size_t)slot * ctx->N;
  f.integral = ctx->integral_ws + (size_t)slot * ctx->N * kIntegralChannels;
  (void)rows;
  f.rowoff = ctx->rowoff_ws + (size_t)slot * ctx->rowoff_slot;
  f.carry = ctx->carry_ws + (size_t)slot * ctx->carry_slot;
  f.cloud = cl;
  f.count_out = nullptr;
}
int ensure_stats(pwn_hip_ctx* ctx, pwn_hip_cloud* c) {
  if (!c->d.St) HIPCHK(ctx, hipMalloc(&c->d.St, sizeof(float) * 16 * (size_t)c->d.capacity), PWN_HIP_ERR_ALLOCATION);
  return PWN_HIP_OK;
}
// direct: the kernels of the call have written counts and fault flag into counts_host themselves (FrameDesc::count_out, launch_convert's fault_out)
// the two halves of sync_and_counts for callers that wait for the stream themselves: queue the copy back of counts + fault flag; digest them
int counts_enqueue(pwn_hip_ctx* ctx, int n) {
  if (n <= 0) return PWN_HIP_OK;
  hipLaunchKernelGGL(k_gather_counts, dim3((n + 256) / 256), dim3(256), 0, ctx->stream, ctx->frames_dev, n, ctx->counts_dev, ctx->fault_dev);
  HIPCHK(ctx, hipMemcpyAsync(ctx->counts_host, ctx->counts_dev, sizeof(int) * (n + 1), hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
  return PWN_HIP_OK;
}
int counts_apply(pwn_hip_ctx* ctx, pwn_hip_cloud* const* clouds, int n);
int sync_and_counts(pwn_hip_ctx* ctx, pwn_hip_cloud* const* clouds, int n, bool direct = false) {
  // frames_dev[0..n) must describe clouds[0..n)
  if (n > 0 && !direct) { if (int rc = counts_enqueue(ctx, n)) return rc; }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  collect_stage_times(ctx);
  return counts_apply(ctx, clouds, n);
}
int counts_apply(pwn_hip_ctx* ctx, pwn_hip_cloud* const* clouds, int n) {
  if (n > 0 && ctx->counts_host[n] != 0) {
    const int code = ctx->counts_host[n];
    (void)hipMemset(ctx->fault_dev, 0, sizeof(int));
    ctx->last_convert_fault = code;
    if (ctx->dbg_withhold_once) { ctx->dbg_withhold = -1; ctx->spin_limit = kSpinLimit; ctx->dbg_withhold_once = 0; }
    return fail(ctx, PWN_HIP_ERR_LAUNCH, "integral image: strip hand-over timed out (results invalid)");
  }
  ctx->last_convert_fault = 0;
  for (int i = 0; i < n; ++i) {
    if (ctx->counts_host[i] > clouds[i]->d.capacity) return fail(ctx, PWN_HIP_ERR_CAPACITY, "cloud capacity smaller than the number of valid depth pixels");
    clouds[i]->n_host = ctx->counts_host[i];
  }
  return PWN_HIP_OK;
}

// One converter call, cut into the pieces a caller can interleave with other work on the same streams: convert_prepare (descriptors of all
// frames, uploaded on ctx->stream), convert_enqueue (the kernels of frames [base, base + m) on one stream) and convert_finish (counts and
// fault flag back, clouds' host-side sizes).  slot[i] = workspace slot of frame i: frames that are in flight at the same time on different
// streams must not share one; a stream reuses its slots from launch to launch (stream order serialises the reuse).
struct ConvertJob {
  ConvertParams cp;
  int n = 0, rows = 0, cols = 0;
  size_t N = 0;
  bool raw = false, host_input = false, direct = false;
  float depth_scale = 0.f;
  std::vector<const void*> src;          // the caller's frame pointers (host frames are staged by convert_enqueue)
  std::vector<int> slot;
};
template <typename SRC>
int convert_prepare(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const SRC* const* frames, float depth_scale, int n, int rows, int cols,
                    pwn_hip_cloud* const* clouds, int keep_stats, bool want_interval, const std::vector<int>& slot, bool direct, ConvertJob& job) {
  const size_t N = (size_t)rows * cols;
  ctx->last_convert_fault = 0;            // the fault word belongs to this call from here on (a stale 1 would make an unrelated failure look like a time-out)
  ConvertParams cp = make_convert_params(ctx, p, nullptr, rows, cols, keep_stats);
  cp.lean = want_interval ? 0 : 1;      // the interval image leaves the converter only through pwn_hip_convert(..., interval_image)
  if (int rc = ensure_desc(ctx, n)) return rc;
  const bool raw = std::is_same<SRC, uint16_t>::value;
  job.n = n; job.rows = rows; job.cols = cols; job.N = N; job.raw = raw; job.depth_scale = depth_scale; job.slot = slot; job.direct = direct;
  job.src.assign(n, nullptr);
  // every frame gets a descriptor; workspace slots are reused round-robin (stream order serialises the reuse)
  for (int i = 0; i < n; ++i) {
    pwn_hip_cloud* c = clouds[i];
    if (!c || !frames[i]) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null frame or cloud");
    if (i == 0) cp.omSym = c->d.omSym;
    else if (c->d.omSym != cp.omSym) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "clouds of one convert batch must share one omega storage (exact9 / sym6)");
    cloud_changes(ctx, c);
    if (keep_stats) { if (int rc = ensure_stats(ctx, c)) return rc; }
    c->has_stats = keep_stats != 0;
    c->n_gauss = 0;                              // the cloud's Gaussians (if any) belonged to its previous content
    if (c->d.OmN) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH); (void)hipFree(c->d.OmN); c->d.OmN = nullptr; }
    make_omega_n_classes(p, c->d);
    job.src[i] = frames[i];
    const float* depth_dev = nullptr;
    if (!raw) depth_dev = reinterpret_cast<const float*>(frames[i]);         // patched below if it is a host pointer
    fill_frame(ctx, i, slot[i], depth_dev, c->d, rows);
    if (c->idx_cap < N) {
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
      if (c->idximg) (void)hipFree(c->idximg);
      c->idximg = nullptr; c->idx_cap = 0;
      HIPCHK(ctx, hipMalloc((void**)&c->idximg, N * sizeof(int)), PWN_HIP_ERR_ALLOCATION);
      c->idx_cap = N;
    }
    ctx->frames_host[i].index = c->idximg;                    // the index image stays with the cloud
    c->idx_valid = cp.hasOffset == 0;
    c->idx_rows = rows; c->idx_cols = cols; std::memcpy(c->idx_K, p->K, sizeof(c->idx_K)); c->idx_minD = p->min_distance; c->idx_maxD = p->max_distance;
    if (raw) {                                                                // uint16 frames are converted on the fly by the kernels
      ctx->frames_host[i].raw = reinterpret_cast<const uint16_t*>(frames[i]); // patched below if it is a host pointer
      ctx->frames_host[i].raw_scale = depth_scale;
    }
  }
  // host inputs are staged per launch; device inputs are used in place
  job.host_input = n > 0 && !is_device_ptr(frames[0]);
  if (job.host_input) {
    for (int i = 0; i < n; ++i) {
      if (raw) ctx->frames_host[i].raw = ctx->raw_ws + (size_t)slot[i] * ctx->N;
      else ctx->frames_host[i].depth = ctx->depth_ws + (size_t)slot[i] * ctx->N;
    }
  }
  // a call of a few frames (latency path, one launch): counts and fault flag land in page-locked host words straight from the kernels
  if (direct) for (int i = 0; i < n; ++i) ctx->frames_host[i].count_out = ctx->counts_host + i;
  HIPCHK(ctx, hipMemcpyAsync(ctx->frames_dev, ctx->frames_host, sizeof(FrameDesc) * n, hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY);
  job.cp = cp;
  return PWN_HIP_OK;
}
// frames [base, base + m) of the job: their host frames (if any) copied on `cs`, their kernels launched on `st`.  The frames' slots must be
// consecutive (slot[base + i] = slot[base] + i).
int convert_stage_frames(pwn_hip_ctx* ctx, const ConvertJob& job, int base, int m, hipStream_t cs) {
  // frames that follow each other in host memory (a ring buffer, one block for the batch) go in one transfer -- only when the frame
  // fills its staging slot (N == ctx->N), so that the slots are contiguous too
  const size_t fbytes = job.N * (job.raw ? sizeof(uint16_t) : sizeof(float));
  const int s0 = job.slot[base];
  for (int i = 0; i < m;) {
    int run = 1;
    if (job.N == ctx->N)
      while (i + run < m && (const char*)job.src[base + i + run] == (const char*)job.src[base + i] + (size_t)run * fbytes) ++run;
    void* dst = job.raw ? (void*)(ctx->raw_ws + (size_t)(s0 + i) * ctx->N) : (void*)(ctx->depth_ws + (size_t)(s0 + i) * ctx->N);
    HIPCHK(ctx, hipMemcpyAsync(dst, job.src[base + i], fbytes * run, hipMemcpyHostToDevice, cs), PWN_HIP_ERR_COPY);
    i += run;
  }
  return PWN_HIP_OK;
}
int convert_finish(pwn_hip_ctx* ctx, const ConvertJob& job, pwn_hip_cloud* const* clouds) { return sync_and_counts(ctx, clouds, job.n, job.direct); }

template <typename SRC>
int convert_batch_impl(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const SRC* const* frames, float depth_scale, int n,
                       int rows, int cols, pwn_hip_cloud* const* clouds, int keep_stats, bool want_interval = false, bool retried = false) {
  if (!ctx || !p || !frames || !clouds || n < 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (int rc = check_image(ctx, rows, cols)) return rc;
  if (int rc = absorb_copies(ctx)) return rc;
  ctx->stages.clear();
  const StreamPlan plan = make_plan(ctx, ctx->sub_frames, n);
  const int sub = plan.sub;
  std::vector<int> slot((size_t)std::max(n, 0));
  for (int i = 0; i < n; ++i) slot[i] = plan.slot0(i / sub) + i % sub;
  const bool direct = n > 0 && n < kSinglePassMinFrames && n <= sub;
  ConvertJob job;
  if (int rc = convert_prepare<SRC>(ctx, p, frames, depth_scale, n, rows, cols, clouds, keep_stats, want_interval, slot, direct, job)) return rc;
  if (int rc = plan_fork(ctx, plan)) return rc;
  // Host frames travel on the copy stream, ahead of the kernels: the frames of sub-batch k are copied while sub-batches k-1, k-2 ... are
  // being converted (with the copies on the sub-batch's own stream the two streams copy at the same time and then compute at the same
  // time: 7.5 ms per 256 VGA frames against 5).  copied[k] orders convert k after its copies; converted[k] orders the copies into a
  // staging block after the kernels that read its previous content.
  const int nsub = (n + sub - 1) / sub;
  const bool ahead = job.host_input && ctx->copy_stream && plan.dual();
  if (ahead) {
    while ((int)ctx->sync_events.size() < 2 * nsub) {
      hipEvent_t e = nullptr;
      HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming), PWN_HIP_ERR_ALLOCATION);
      ctx->sync_events.push_back(e);
    }
    HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->fork_ev, 0), PWN_HIP_ERR_LAUNCH);      // after everything queued before this call
  }
  for (int base = 0, k = 0; base < n; base += sub, ++k) {
    const int m = std::min(sub, n - base);
    hipStream_t st = plan.stream(k);
    if (job.host_input) {
      hipStream_t cs = ahead ? ctx->copy_stream : st;
      if (ahead && k >= plan.ns) HIPCHK(ctx, hipStreamWaitEvent(cs, ctx->sync_events[2 * (k - plan.ns) + 1], 0), PWN_HIP_ERR_LAUNCH);
      if (int rc = convert_stage_frames(ctx, job, base, m, cs)) return rc;
      if (ahead) {
        HIPCHK(ctx, hipEventRecord(ctx->sync_events[2 * k], cs), PWN_HIP_ERR_LAUNCH);
        HIPCHK(ctx, hipStreamWaitEvent(st, ctx->sync_events[2 * k], 0), PWN_HIP_ERR_LAUNCH);
      }
    }
    if (int rc = launch_convert(ctx, job.cp, base, m, st, direct ? ctx->counts_host + n : nullptr)) return rc;
    if (ahead) HIPCHK(ctx, hipEventRecord(ctx->sync_events[2 * k + 1], st), PWN_HIP_ERR_LAUNCH);
  }
  if (int rc = plan_join(ctx, plan)) return rc;
  const int rc = convert_finish(ctx, job, clouds);
  if (rc != PWN_HIP_OK && ctx->last_convert_fault == 1 && !retried) {
    // A strip waited for its left neighbour longer than the poll bound (~1 s): the neighbour's workgroup was not dispatched in time -- a device
    // shared with another process's long kernels -- and the planes of this call are invalid.  Nothing is left behind (epoch-tagged words), so
    // the call is simply made again, once; a second time-out is reported.
    ++ctx->convert_retries;
    return convert_batch_impl<SRC>(ctx, p, frames, depth_scale, n, rows, cols, clouds, keep_stats, want_interval, true);
  }
  return rc;
}

}  // namespace

// ================================================================================================================
extern "C" {

int pwn_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}
const char* pwn_hip_last_error_string(const pwn_hip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int pwn_hip_host_alloc(void** ptr, size_t bytes) {
  if (!ptr || bytes == 0) return fail(nullptr, PWN_HIP_ERR_INVALID_ARGUMENT, "bad host_alloc argument");
  *ptr = nullptr;
  hipError_t e = hipHostMalloc(ptr, bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); *ptr = nullptr; return fail(nullptr, PWN_HIP_ERR_ALLOCATION, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
  return PWN_HIP_OK;
}
int pwn_hip_host_free(void* ptr) {
  if (!ptr) return PWN_HIP_OK;
  hipError_t e = hipHostFree(ptr);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, PWN_HIP_ERR_INVALID_ARGUMENT, std::string("hipHostFree: ") + hipGetErrorString(e)); }
  return PWN_HIP_OK;
}

int pwn_hip_device_alloc(pwn_hip_ctx* ctx, void** ptr, size_t bytes) {
  if (!ctx || !ptr || bytes == 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "bad device_alloc argument");
  *ptr = nullptr;
  HIPCHK(ctx, hipSetDevice(ctx->device), PWN_HIP_ERR_NO_DEVICE);
  hipError_t e = hipMalloc(ptr, bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); *ptr = nullptr; return fail(ctx, PWN_HIP_ERR_ALLOCATION, std::string("hipMalloc: ") + hipGetErrorString(e)); }
  return PWN_HIP_OK;
}
int pwn_hip_device_free(pwn_hip_ctx* ctx, void* ptr) {
  if (!ptr) return PWN_HIP_OK;
  if (!ctx) {      // the context that allocated it is gone (and with it everything that could still use the buffer): plain hipFree, which
    hipError_t e = hipFree(ptr);      // waits for the device by itself
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, PWN_HIP_ERR_INVALID_ARGUMENT, std::string("hipFree: ") + hipGetErrorString(e)); }
    return PWN_HIP_OK;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device), PWN_HIP_ERR_NO_DEVICE);
  if (int rc = absorb_copies(ctx)) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);       // nothing queued may still read or write it
  HIPCHK(ctx, hipFree(ptr), PWN_HIP_ERR_INVALID_ARGUMENT);
  return PWN_HIP_OK;
}
int pwn_hip_copy(pwn_hip_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || !dst || !src) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "bad copy argument");
  if (bytes == 0) return PWN_HIP_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device), PWN_HIP_ERR_NO_DEVICE);
  if (int rc = absorb_copies(ctx)) return rc;
  HIPCHK(ctx, copy_any(dst, src, bytes, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_COPY);
  return PWN_HIP_OK;
}
int pwn_hip_copy_async(pwn_hip_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || !dst || !src) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "bad copy argument");
  if (bytes == 0) return PWN_HIP_OK;
  if (!ctx->copy_stream || !ctx->copy_ev) return pwn_hip_copy(ctx, dst, src, bytes);
  HIPCHK(ctx, hipSetDevice(ctx->device), PWN_HIP_ERR_NO_DEVICE);
  HIPCHK(ctx, copy_any(dst, src, bytes, ctx->copy_stream), PWN_HIP_ERR_COPY);
  ctx->copy_pending = true;
  return PWN_HIP_OK;
}

void pwn_hip_default_converter_params(pwn_hip_converter_params* p) {
  std::memset(p, 0, sizeof(*p));
  const float K[9] = { 1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.5f, 0.5f, 1.f };   // pinholepointprojector.cpp:6-9
  std::memcpy(p->K, K, sizeof(K));
  p->min_distance = 0.01f; p->max_distance = 6.0f;
  p->world_radius = 0.1f; p->min_image_radius = 10; p->max_image_radius = 30; p->min_points = 50;
  p->stats_curvature_threshold = 0.02f; p->point_info_curvature_threshold = 0.02f; p->normal_info_curvature_threshold = 0.02f;
  p->point_flat_diag[0] = 1000.f; p->point_flat_diag[1] = 1.f; p->point_flat_diag[2] = 1.f;
  for (int i = 0; i < 3; ++i) { p->point_nonflat_diag[i] = 1.f; p->normal_flat_diag[i] = 100.f; p->normal_nonflat_diag[i] = 1.f; }
  const Mat4 I = mat4_identity();
  std::memcpy(p->sensor_offset, I.m, sizeof(I.m));
}
void pwn_hip_default_aligner_params(pwn_hip_aligner_params* p) {
  std::memset(p, 0, sizeof(*p));
  const float K[9] = { 1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.5f, 0.5f, 1.f };
  std::memcpy(p->K, K, sizeof(K));
  p->min_distance = 0.01f; p->max_distance = 6.0f;
  p->rows = 0; p->cols = 0;
  p->inlier_distance_threshold = 0.5f;
  p->inlier_normal_angular_threshold = (float)cos(M_PI / 6);
  p->flat_curvature_threshold = 0.02f; p->inlier_curvature_ratio_threshold = 1.3f;
  p->inlier_max_chi2 = 9e3f; p->robust_kernel = 1; p->outer_iterations = 10; p->inner_iterations = 1;
  const Mat4 I = mat4_identity();
  std::memcpy(p->reference_sensor_offset, I.m, sizeof(I.m));
  std::memcpy(p->current_sensor_offset, I.m, sizeof(I.m));
  std::memcpy(p->initial_guess, I.m, sizeof(I.m));
}

// high_priority: the streams of the look-ahead helper context that works beside BATCH calls (AsyncConvert::helper_hi).  Its one-frame jobs run beside
// streams that hold hundreds of queued launches; on the default priority a stream of the helper can share a hardware queue with one of those and is
// then served when that queue has drained -- at the end of the batch instead of beside it.  (Only for a context whose streams do not wait on other streams' events: a
// high-priority context used for the partition step's imports, which wait for a broadcast, made the whole step 8-13 % slower -- docs/experiments.md, round 6.)
static int ctx_create(pwn_hip_ctx** out, int device, int max_rows, int max_cols, int max_batch, bool high_priority);
int pwn_hip_ctx_create(pwn_hip_ctx** out, int device, int max_rows, int max_cols, int max_batch) { return ctx_create(out, device, max_rows, max_cols, max_batch, false); }
static hipError_t make_stream(hipStream_t* s, bool high_priority) {
  if (high_priority) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least && hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();
  }
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
static int ctx_create(pwn_hip_ctx** out, int device, int max_rows, int max_cols, int max_batch, bool high_priority) {
  if (!out || max_rows <= 0 || max_cols <= 0 || max_batch <= 0) return fail(nullptr, PWN_HIP_ERR_INVALID_ARGUMENT, "bad ctx_create argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return fail(nullptr, PWN_HIP_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)"); }
  if (device < 0 || device >= ndev) return fail(nullptr, PWN_HIP_ERR_NO_DEVICE, "device index out of range");
  HIPCHK(nullptr, hipSetDevice(device), PWN_HIP_ERR_NO_DEVICE);
  pwn_hip_ctx* ctx = new pwn_hip_ctx();
  ctx->device = device; ctx->max_rows = max_rows; ctx->max_cols = max_cols; ctx->max_batch = max_batch;
  ctx->N = (size_t)max_rows * max_cols;
  { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->num_cus = cus; }
  const size_t N = ctx->N, B = (size_t)max_batch;
  ctx->nblocks_max = align_nblocks((int)N);
#define ALLOC(ptr, bytes) do { hipError_t e_ = hipMalloc((void**)&(ptr), (bytes)); if (e_ != hipSuccess) { std::string m = std::string("hipMalloc ") + #ptr + ": " + hipGetErrorString(e_); pwn_hip_ctx_destroy(ctx); return fail(nullptr, PWN_HIP_ERR_ALLOCATION, m); } } while (0)
#define HALLOC(ptr, bytes) do { hipError_t e_ = hipHostMalloc((void**)&(ptr), (bytes)); if (e_ != hipSuccess) { std::string m = std::string("hipHostMalloc ") + #ptr + ": " + hipGetErrorString(e_); pwn_hip_ctx_destroy(ctx); return fail(nullptr, PWN_HIP_ERR_ALLOCATION, m); } } while (0)
  if (make_stream(&ctx->own_stream, high_priority) != hipSuccess) { delete ctx; return fail(nullptr, PWN_HIP_ERR_ALLOCATION, "hipStreamCreate failed"); }
  ctx->stream = ctx->own_stream;
  (void)make_stream(&ctx->stream2, high_priority);
  for (int k = 0; k < 2; ++k) { (void)make_stream(&ctx->extra[k], high_priority); (void)hipEventCreateWithFlags(&ctx->join_extra[k], hipEventDisableTiming); }
  (void)hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming); (void)hipEventCreateWithFlags(&ctx->join_ev, hipEventDisableTiming);
  (void)hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
  (void)hipEventCreateWithFlags(&ctx->copy_ev, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&ctx->foreign_ev, hipEventDisableTiming);
  ALLOC(ctx->depth_ws, B * N * sizeof(float));
  ALLOC(ctx->raw_ws, B * N * sizeof(uint16_t));
  ALLOC(ctx->index_ws, B * N * sizeof(int));
  ALLOC(ctx->interval_ws, B * N * sizeof(int));
  ALLOC(ctx->integral_ws, B * N * kIntegralChannels * sizeof(float));
  {   // any rows x cols image with rows*cols <= N and rows, cols <= M: rows*strips <= N/64 + M, strips*bands <= N/1024 + M/64 + M/16 + 1
    const size_t M = (size_t)std::max(max_rows, max_cols);
    ctx->rowoff_slot = N / 64 + M + 64;
    ctx->carry_slot = (N / ((size_t)kIR_Cols * kIR_Rows) + M / kIR_Cols + M / kIR_Rows + 2) * (size_t)kII_Chains;
  }
  ALLOC(ctx->rowoff_ws, B * ctx->rowoff_slot * sizeof(int));
  ALLOC(ctx->carry_ws, B * ctx->carry_slot * sizeof(unsigned long long));
  ALLOC(ctx->fault_dev, sizeof(int));
  HALLOC(ctx->align_fault_host, sizeof(int));
  HALLOC(ctx->flat_hdr_host, 256);
  *ctx->align_fault_host = 0;
  if (hipMemset(ctx->carry_ws, 0, B * ctx->carry_slot * sizeof(unsigned long long)) != hipSuccess || hipMemset(ctx->fault_dev, 0, sizeof(int)) != hipSuccess) {
    pwn_hip_ctx_destroy(ctx); return fail(nullptr, PWN_HIP_ERR_ALLOCATION, "hipMemset of the hand-over workspace failed"); }
  ALLOC(ctx->zref_ws, N * sizeof(unsigned long long));
  ALLOC(ctx->z32ref_ws, B * N * sizeof(unsigned));
  ALLOC(ctx->z32cur_ws, B * N * sizeof(unsigned));
  ALLOC(ctx->curidx_ws, B * N * sizeof(int));
  ALLOC(ctx->partials_ws, B * (size_t)ctx->nblocks_max * kAccN * sizeof(double));
  ALLOC(ctx->solve_dev, sizeof(SolveOut));
  ALLOC(ctx->counters_dev, 16 * sizeof(int));
  ALLOC(ctx->corr_ws, N * sizeof(int2));
  ALLOC(ctx->scratch_count, sizeof(int));
  ALLOC(ctx->io_ws, N * 16 * sizeof(float));
#undef ALLOC
#undef HALLOC
  (void)hipEventCreateWithFlags(&ctx->t0, hipEventDisableSystemFence); (void)hipEventCreateWithFlags(&ctx->t1, hipEventDisableSystemFence);
  if (int rc = ensure_desc(ctx, std::max(16, max_batch))) { std::string m = ctx->err; pwn_hip_ctx_destroy(ctx); return fail(nullptr, rc, m); }
  *out = ctx;
  return PWN_HIP_OK;
}
int pwn_hip_ctx_destroy(pwn_hip_ctx* ctx) {
  if (!ctx) return PWN_HIP_OK;
  (void)hipSetDevice(ctx->device);
  if (AsyncConvert* a = ctx->async) {       // a conversion still in flight finishes first; its result is dropped
    { std::unique_lock<std::mutex> lk(a->m); a->quit = true; }
    a->cv.notify_all();
    if (a->worker.joinable()) a->worker.join();
    if (a->helper) pwn_hip_ctx_destroy(a->helper);
    if (a->helper_hi) pwn_hip_ctx_destroy(a->helper_hi);
    delete a; ctx->async = nullptr;
  }
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);      // pwn_hip_copy_async transfers still in flight
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  void* dev[] = { ctx->depth_ws, ctx->raw_ws, ctx->index_ws, ctx->interval_ws, ctx->integral_ws, ctx->rowoff_ws, ctx->carry_ws, ctx->fault_dev, ctx->zref_ws, ctx->z32ref_ws, ctx->z32cur_ws, ctx->curidx_ws,
                  ctx->partials_ws, ctx->state_ws, ctx->frames_dev, ctx->pairs_dev, ctx->raw_dev, ctx->counts_dev, ctx->solve_dev, ctx->counters_dev,
                  ctx->corr_ws, ctx->scratch_count, ctx->io_ws };
  for (void* p : dev) if (p) (void)hipFree(p);
  void* host[] = { ctx->frames_host, ctx->pairs_host, ctx->raw_host, ctx->state_host, ctx->counts_host, ctx->match_host };
  for (void* p : host) if (p) (void)hipHostFree(p);
  if (ctx->match_dev) (void)hipFree(ctx->match_dev);
  if (ctx->stats_dev) (void)hipFree(ctx->stats_dev);
  if (ctx->stats_host) (void)hipHostFree(ctx->stats_host);
  collect_stage_times(ctx);
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->t0) (void)hipEventDestroy(ctx->t0);
  if (ctx->t1) (void)hipEventDestroy(ctx->t1);
  for (pwn_hip_cloud* r : ctx->cloud_pool) cloud_free(r);
  ctx->cloud_pool.clear();
  for (int k = 0; k < 8; ++k) if (ctx->scene_i[k]) (void)hipFree(ctx->scene_i[k]);
  for (int k = 0; k < 3; ++k) if (ctx->scene_k[k]) (void)hipFree(ctx->scene_k[k]);
  if (ctx->scene_total) (void)hipFree(ctx->scene_total);
  if (ctx->records_ws) (void)hipFree(ctx->records_ws);
  if (ctx->ids_dev) (void)hipFree(ctx->ids_dev);
  if (ctx->zdepth_ws) (void)hipFree(ctx->zdepth_ws);
  if (ctx->align_fault_host) (void)hipHostFree(ctx->align_fault_host);
  if (ctx->flat_hdr_host) (void)hipHostFree(ctx->flat_hdr_host);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  for (int k = 0; k < 2; ++k) { if (ctx->extra[k]) (void)hipStreamDestroy(ctx->extra[k]); if (ctx->join_extra[k]) (void)hipEventDestroy(ctx->join_extra[k]); }
  if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
  if (ctx->copy_ev) (void)hipEventDestroy(ctx->copy_ev);
  if (ctx->foreign_ev) (void)hipEventDestroy(ctx->foreign_ev);
  for (hipEvent_t e : ctx->sync_events) (void)hipEventDestroy(e);
  ctx->sync_events.clear();
  if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
  if (ctx->join_ev) (void)hipEventDestroy(ctx->join_ev);
  delete ctx;
  return PWN_HIP_OK;
}
int pwn_hip_ctx_set_stream(pwn_hip_ctx* ctx, void* hip_stream) {
  if (!ctx) return fail(nullptr, PWN_HIP_ERR_INVALID_ARGUMENT, "null ctx");
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return PWN_HIP_OK;
}
int pwn_hip_ctx_wait_stream(pwn_hip_ctx* ctx, void* hip_stream) {
  if (!ctx) return fail(nullptr, PWN_HIP_ERR_INVALID_ARGUMENT, "null ctx");
  if (!ctx->foreign_ev) return fail(ctx, PWN_HIP_ERR_ALLOCATION, "no event");
  HIPCHK(ctx, hipSetDevice(ctx->device), PWN_HIP_ERR_NO_DEVICE);
  // everything the context queues from now on (its other streams fork from ctx->stream) runs after what the caller's stream holds now
  HIPCHK(ctx, hipEventRecord(ctx->foreign_ev, (hipStream_t)hip_stream), PWN_HIP_ERR_LAUNCH);
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->foreign_ev, 0), PWN_HIP_ERR_LAUNCH);
  return PWN_HIP_OK;
}
int pwn_hip_ctx_signal_stream(pwn_hip_ctx* ctx, void* hip_stream) {
  if (!ctx) return fail(nullptr, PWN_HIP_ERR_INVALID_ARGUMENT, "null ctx");
  if (!ctx->foreign_ev) return fail(ctx, PWN_HIP_ERR_ALLOCATION, "no event");
  HIPCHK(ctx, hipSetDevice(ctx->device), PWN_HIP_ERR_NO_DEVICE);
  // the mirror image of pwn_hip_ctx_wait_stream: what the caller queues on its stream from now on runs after everything the context has queued
  // so far (every batch call joins its streams back into ctx->stream before it packs its records)
  HIPCHK(ctx, hipEventRecord(ctx->foreign_ev, ctx->stream), PWN_HIP_ERR_LAUNCH);
  HIPCHK(ctx, hipStreamWaitEvent((hipStream_t)hip_stream, ctx->foreign_ev, 0), PWN_HIP_ERR_LAUNCH);
  return PWN_HIP_OK;
}
int pwn_hip_ctx_set_enqueued_callback(pwn_hip_ctx* ctx, void (*fn)(void*), void* user) {
  if (!ctx) return fail(nullptr, PWN_HIP_ERR_INVALID_ARGUMENT, "null ctx");
  ctx->enqueued_cb = fn; ctx->enqueued_user = fn ? user : nullptr;
  return PWN_HIP_OK;
}
int pwn_hip_ctx_synchronize(pwn_hip_ctx* ctx) {
  if (!ctx) return fail(nullptr, PWN_HIP_ERR_INVALID_ARGUMENT, "null ctx");
  if (int rc = absorb_copies(ctx)) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  return PWN_HIP_OK;
}
int pwn_hip_ctx_set_subbatch(pwn_hip_ctx* ctx, int frames, int pairs) {
  if (!ctx || frames <= 0 || pairs <= 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "bad subbatch");
  ctx->sub_frames = std::min(frames, ctx->max_batch); ctx->sub_pairs = std::min(pairs, ctx->max_batch);
  return PWN_HIP_OK;
}
int pwn_hip_ctx_set_omega_storage(pwn_hip_ctx* ctx, int mode) {
  if (!ctx || (mode != PWN_HIP_OMEGA_EXACT9 && mode != PWN_HIP_OMEGA_SYM6)) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "omega storage must be PWN_HIP_OMEGA_EXACT9 or PWN_HIP_OMEGA_SYM6");
  ctx->omega_sym = mode == PWN_HIP_OMEGA_SYM6 ? 1 : 0;
  return PWN_HIP_OK;
}
int pwn_hip_cloud_omega_storage(pwn_hip_ctx* ctx, const pwn_hip_cloud* c, int* mode) {
  if (!c || !mode) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  *mode = c->d.omSym ? PWN_HIP_OMEGA_SYM6 : PWN_HIP_OMEGA_EXACT9;
  return PWN_HIP_OK;
}
int pwn_hip_ctx_set_concurrency(pwn_hip_ctx* ctx, int streams) {
  if (!ctx || streams < 1 || streams > 4) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "streams must be 1..4");
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  ctx->concurrency = streams;
  return PWN_HIP_OK;
}
// Test hook for the strip hand-over of the converter's integral-image kernels: one hand-over word (strip, band, chain) of every
// frame is not written, so the chain to its right times out after `spin_limit` polls (0 = the default, ~1 s), the launch raises
// its fault flag and the convert call returns PWN_HIP_ERR_LAUNCH instead of hanging.  strip < 0 switches the hook off.
int pwn_hip_debug_withhold_carry(pwn_hip_ctx* ctx, int strip, int band, int chain, int rows, int spin_limit) {
  if (!ctx) return fail(nullptr, PWN_HIP_ERR_INVALID_ARGUMENT, "null ctx");
  if (strip < 0) { ctx->dbg_withhold = -1; ctx->spin_limit = kSpinLimit; ctx->dbg_withhold_once = 0; return PWN_HIP_OK; }
  if (band < 0 || chain < 0 || chain >= kII_Chains || rows <= 0 || band >= bands_of(rows)) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "bad hand-over word");
  ctx->dbg_withhold = (strip * bands_of(rows) + band) * kII_Chains + chain;
  ctx->dbg_withhold_once = spin_limit < 0 ? 1 : 0;                  // negative: |spin_limit| polls, and only the next launch is disturbed
  ctx->spin_limit = spin_limit > 0 ? spin_limit : (spin_limit < 0 ? -spin_limit : kSpinLimit);
  return PWN_HIP_OK;
}
int pwn_hip_debug_convert_retries(pwn_hip_ctx* ctx, int* retries) {
  if (!ctx || !retries) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  *retries = ctx->convert_retries;
  return PWN_HIP_OK;
}
// Test hook: 0 makes every alignment project both clouds in every iteration, also where a cloud's own index image is known to be that
// projection's result (the shortcut of align_batch_impl) -- so that tests can hold the shortcut against the projection it replaces.
int pwn_hip_debug_set_index_shortcut(pwn_hip_ctx* ctx, int enabled) {
  if (!ctx) return fail(nullptr, PWN_HIP_ERR_INVALID_ARGUMENT, "null ctx");
  ctx->index_shortcut = enabled ? 1 : 0;
  return PWN_HIP_OK;
}
// Test hooks of the projection's fault path: the rounds a thread of k_project spends on a contended pixel before it gives up (0 = every
// collision gives up at once; < 0 = the default), and how many alignment calls were repeated with the two-pass projection so far.
int pwn_hip_debug_set_settle_guard(pwn_hip_ctx* ctx, int rounds) {
  if (!ctx) return fail(nullptr, PWN_HIP_ERR_INVALID_ARGUMENT, "null ctx");
  ctx->settle_guard = rounds < 0 ? kSettleGuard : rounds;
  return PWN_HIP_OK;
}
int pwn_hip_debug_projection_fallbacks(pwn_hip_ctx* ctx, int* calls) {
  if (!ctx || !calls) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  *calls = ctx->projection_fallbacks;
  return PWN_HIP_OK;
}
int pwn_hip_set_profiling(pwn_hip_ctx* ctx, int enabled) {
  if (!ctx) return fail(nullptr, PWN_HIP_ERR_INVALID_ARGUMENT, "null ctx");
  ctx->profiling = enabled != 0;
  return PWN_HIP_OK;
}
// ---- what the box's HBM delivers (the measured figure next to the 8 TB/s spec) ---------------------------------
// float4 streaming kernels, 2048 workgroups x 256 threads, grid-stride with four independent loads in flight per thread
__global__ void __launch_bounds__(256) k_probe_read(const v4f* __restrict__ src, size_t n4, float* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  v4f a = { 0.f, 0.f, 0.f, 0.f }, b = a, c = a, d = a;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const v4f x0 = __builtin_nontemporal_load(src + i), x1 = __builtin_nontemporal_load(src + i + stride);
    const v4f x2 = __builtin_nontemporal_load(src + i + 2 * stride), x3 = __builtin_nontemporal_load(src + i + 3 * stride);
    a.x += x0.x; a.y += x0.y; a.z += x0.z; a.w += x0.w; b.x += x1.x; b.y += x1.y; b.z += x1.z; b.w += x1.w;
    c.x += x2.x; c.y += x2.y; c.z += x2.z; c.w += x2.w; d.x += x3.x; d.y += x3.y; d.z += x3.z; d.w += x3.w;
  }
  for (; i < n4; i += stride) { const v4f x0 = src[i]; a.x += x0.x; a.y += x0.y; a.z += x0.z; a.w += x0.w; }
  const float s = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w));
  if (s == 123456.789f) sink[0] = s;          // never true for the zero-filled buffer: keeps the loads alive
}
__global__ void __launch_bounds__(256) k_probe_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 x0 = src[i], x1 = src[i + stride], x2 = src[i + 2 * stride], x3 = src[i + 3 * stride];
    dst[i] = x0; dst[i + stride] = x1; dst[i + 2 * stride] = x2; dst[i + 3 * stride] = x3;
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}
int pwn_hip_measure_hbm(pwn_hip_ctx* ctx, size_t bytes, float* read_gbps, float* copy_gbps) {
  if (!ctx || bytes < (1u << 20)) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null ctx or fewer than 1 MiB");
  HIPCHK(ctx, hipSetDevice(ctx->device), PWN_HIP_ERR_NO_DEVICE);
  const size_t n4 = bytes / sizeof(float4);
  float4 *src = nullptr, *dst = nullptr;
  HIPCHK(ctx, hipMalloc(&src, n4 * sizeof(float4)), PWN_HIP_ERR_ALLOCATION);
  if (hipMalloc(&dst, n4 * sizeof(float4)) != hipSuccess) { (void)hipFree(src); return fail(ctx, PWN_HIP_ERR_ALLOCATION, "hipMalloc (probe)"); }
  hipStream_t st = ctx->stream;
  (void)hipMemsetAsync(src, 0, n4 * sizeof(float4), st); (void)hipMemsetAsync(dst, 0, n4 * sizeof(float4), st);
  float best_r = 1e30f, best_c = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    float ms = 0.f;
    (void)hipEventRecord(ctx->t0, st);
    hipLaunchKernelGGL(k_probe_read, dim3(2048), dim3(256), 0, st, (const v4f*)src, n4, (float*)dst);
    (void)hipEventRecord(ctx->t1, st); (void)hipEventSynchronize(ctx->t1); (void)hipEventElapsedTime(&ms, ctx->t0, ctx->t1);
    if (rep > 0 && ms < best_r) best_r = ms;
    (void)hipEventRecord(ctx->t0, st);
    hipLaunchKernelGGL(k_probe_copy, dim3(2048), dim3(256), 0, st, src, dst, n4);
    (void)hipEventRecord(ctx->t1, st); (void)hipEventSynchronize(ctx->t1); (void)hipEventElapsedTime(&ms, ctx->t0, ctx->t1);
    if (rep > 0 && ms < best_c) best_c = ms;
  }
  const hipError_t e = hipGetLastError();
  (void)hipFree(src); (void)hipFree(dst);
  if (e != hipSuccess) return fail(ctx, PWN_HIP_ERR_LAUNCH, hipGetErrorString(e));
  const double b = (double)n4 * sizeof(float4);
  if (read_gbps) *read_gbps = (float)(b / (best_r * 1e-3) / 1e9);
  if (copy_gbps) *copy_gbps = (float)(2.0 * b / (best_c * 1e-3) / 1e9);
  return PWN_HIP_OK;
}
int pwn_hip_last_stage_ms(pwn_hip_ctx* ctx, const char* stage, float* ms, int* launches) {
  if (!ctx || !stage) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  auto it = ctx->stages.find(stage);
  if (ms) *ms = it == ctx->stages.end() ? 0.f : it->second.ms;
  if (launches) *launches = it == ctx->stages.end() ? 0 : it->second.launches;
  return PWN_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------- clouds
int pwn_hip_cloud_create(pwn_hip_ctx* ctx, int capacity, pwn_hip_cloud** out) {
  if (!ctx || !out || capacity <= 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "bad cloud_create argument");
  if (capacity > kMaxCloudPoints) return fail(ctx, PWN_HIP_ERR_CAPACITY, "a cloud holds at most 2^25 points (index field of the scene stage's z-buffer word)");
  HIPCHK(ctx, hipSetDevice(ctx->device), PWN_HIP_ERR_NO_DEVICE);
  for (size_t k = 0; k < ctx->cloud_pool.size(); ++k) {
    pwn_hip_cloud* r = ctx->cloud_pool[k];
    if (r->d.capacity != capacity || r->d.omSym != ctx->omega_sym) continue;
    ctx->cloud_pool.erase(ctx->cloud_pool.begin() + (long)k);
    ctx->cloud_pool_bytes -= cloud_core_bytes(r);
    HIPCHK(ctx, hipMemsetAsync(r->d.count, 0, sizeof(int), ctx->stream), PWN_HIP_ERR_COPY);      // stream order: after whatever used the retired cloud
    *out = r;
    return PWN_HIP_OK;
  }
  pwn_hip_cloud* c = new pwn_hip_cloud();
  std::memset(&c->d, 0, sizeof(c->d));
  c->d.capacity = capacity;
  c->d.omSym = ctx->omega_sym;
  const size_t cap = (size_t)capacity;
  hipError_t e = hipMalloc((void**)&c->d.P3, cap * 3 * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&c->d.Nc, cap * sizeof(float4));
  if (e == hipSuccess) e = hipMalloc((void**)&c->d.Om, om_floats(c->d) * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&c->d.count, sizeof(int));
  if (e == hipSuccess) e = hipMemsetAsync(c->d.count, 0, sizeof(int), ctx->stream);
  if (e != hipSuccess) { pwn_hip_cloud_destroy(ctx, c); return fail(ctx, PWN_HIP_ERR_ALLOCATION, std::string("cloud allocation: ") + hipGetErrorString(e)); }
  *out = c;
  return PWN_HIP_OK;
}
int pwn_hip_cloud_destroy(pwn_hip_ctx* ctx, pwn_hip_cloud* c) {
  if (!c) return PWN_HIP_OK;
  if (ctx && ctx->async && ctx->async->cloud == c) (void)pwn_hip_convert_end(ctx, c);      // a conversion into this cloud is still in flight
  cloud_changes(ctx, c);
  // plain clouds (no scene-stage or uploaded extras) retire into the context's pool: every call that used them has joined its streams
  // back into ctx->stream before returning, and a reuse is enqueued on that stream
  const bool plain = !c->d.OmN && !c->d.St && !c->sb.G && !c->sb.Gf && !c->back.P3 && !c->back.Nc && !c->back.Om && !c->back.OmN && !c->back.St &&
                     !c->sback.G && !c->sback.Gf;
  if (ctx && plain && c->d.P3 && c->d.Nc && c->d.Om && c->d.count && ctx->cloud_pool_bytes + cloud_core_bytes(c) <= kCloudPoolBytes) {
    c->n_host = 0; c->has_stats = false; c->n_gauss = 0; c->idx_valid = false;
    ctx->cloud_pool.push_back(c);
    ctx->cloud_pool_bytes += cloud_core_bytes(c);
    return PWN_HIP_OK;
  }
  if (ctx) (void)hipStreamSynchronize(ctx->stream);
  cloud_free(c);
  return PWN_HIP_OK;
}
int pwn_hip_cloud_size(pwn_hip_ctx* ctx, const pwn_hip_cloud* c, int* n) {
  if (!c || !n) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  *n = c->n_host;
  return PWN_HIP_OK;
}
int pwn_hip_cloud_upload(pwn_hip_ctx* ctx, pwn_hip_cloud* c, int n, const float* points, const float* normals, const float* curvature,
                         const float* omega_p, const float* omega_n) {
  if (!ctx || !c || n < 0 || !points || !normals || !curvature || !omega_p || !omega_n) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (n > c->d.capacity) return fail(ctx, PWN_HIP_ERR_CAPACITY, "cloud capacity too small");
  cloud_changes(ctx, c);
  if (int rc = absorb_copies(ctx)) return rc;      // caller pointers may be the destination of a queued pwn_hip_copy_async
  // repack on the host into the device layout (upload is not on the hot path)
  std::vector<float> hp((size_t)n * 4), hn((size_t)n * 4), hc(n), hop((size_t)n * 16), hon((size_t)n * 16);
  HIPCHK(ctx, copy_any(hp.data(), points, hp.size() * 4, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, copy_any(hn.data(), normals, hn.size() * 4, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, copy_any(hc.data(), curvature, hc.size() * 4, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, copy_any(hop.data(), omega_p, hop.size() * 4, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, copy_any(hon.data(), omega_n, hon.size() * 4, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_COPY);
  const size_t cap = (size_t)c->d.capacity;
  const int sym = c->d.omSym;
  std::vector<float> P((size_t)n * 4), Nm((size_t)n * 4), Om(om_floats(c->d), 0.f), OmN(cap * 9, 0.f);
  for (int i = 0; i < n; ++i) {
    P[4 * i] = hp[4 * i]; P[4 * i + 1] = hp[4 * i + 1]; P[4 * i + 2] = hp[4 * i + 2]; P[4 * i + 3] = hc[i];
    Nm[4 * i] = hn[4 * i]; Nm[4 * i + 1] = hn[4 * i + 1]; Nm[4 * i + 2] = hn[4 * i + 2];
    const int one = 1; std::memcpy(&Nm[4 * i + 3], &one, 4);
    for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) {
      if (!(sym && om_is_lower(3 * r + q))) Om[omp_at(cap, i, 3 * r + q, sym)] = hop[(size_t)16 * i + r + 4 * q];      // column-major 4x4 -> entry (r,q); sym6 keeps the upper triangle
      OmN[om_at(cap, i, 3 * r + q)] = hon[(size_t)16 * i + r + 4 * q];
    }
  }
  if (!c->d.OmN) HIPCHK(ctx, hipMalloc((void**)&c->d.OmN, cap * 9 * sizeof(float)), PWN_HIP_ERR_ALLOCATION);
  if (int rc = cloud_store_records(ctx, c->d, n, P, Nm)) return rc;
  HIPCHK(ctx, hipMemcpy(c->d.Om, Om.data(), Om.size() * 4, hipMemcpyHostToDevice), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipMemcpy(c->d.OmN, OmN.data(), OmN.size() * 4, hipMemcpyHostToDevice), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipMemcpy(c->d.count, &n, sizeof(int), hipMemcpyHostToDevice), PWN_HIP_ERR_COPY);
  c->n_host = n; c->has_stats = false; c->idx_valid = false;
  return PWN_HIP_OK;
}
int pwn_hip_cloud_download(pwn_hip_ctx* ctx, const pwn_hip_cloud* c, float* points, float* normals, float* curvature, float* omega_p, float* omega_n) {
  if (!ctx || !c) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const int n = c->n_host; const size_t cap = (size_t)c->d.capacity;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  std::vector<float> P, Nm;
  if (int rc = cloud_fetch_records(ctx, c->d, n, P, Nm)) return rc;
  std::vector<float> hp, hn, hc, hop, hon;
  if (points) { hp.resize((size_t)n * 4); for (int i = 0; i < n; ++i) { hp[4*i] = P[4*i]; hp[4*i+1] = P[4*i+1]; hp[4*i+2] = P[4*i+2]; hp[4*i+3] = 1.0f; } }
  if (normals) { hn.resize((size_t)n * 4); for (int i = 0; i < n; ++i) { hn[4*i] = Nm[4*i]; hn[4*i+1] = Nm[4*i+1]; hn[4*i+2] = Nm[4*i+2]; hn[4*i+3] = 0.0f; } }
  if (curvature) { hc.resize(n); for (int i = 0; i < n; ++i) hc[i] = P[4*i+3]; }
  if (omega_p || omega_n) {
    std::vector<float> Om(om_floats(c->d)), OmN;
    HIPCHK(ctx, hipMemcpy(Om.data(), c->d.Om, Om.size() * 4, hipMemcpyDeviceToHost), PWN_HIP_ERR_COPY);
    if (c->d.OmN) { OmN.resize(cap * 9); HIPCHK(ctx, hipMemcpy(OmN.data(), c->d.OmN, OmN.size() * 4, hipMemcpyDeviceToHost), PWN_HIP_ERR_COPY); }
    if (omega_p) hop.assign((size_t)n * 16, 0.f);
    if (omega_n) hon.assign((size_t)n * 16, 0.f);
    for (int i = 0; i < n; ++i) {
      int cls; std::memcpy(&cls, &Nm[4 * i + 3], 4); cls &= kClsMask;
      for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) {
        if (omega_p) hop[(size_t)16 * i + r + 4 * q] = Om[omp_at(cap, i, 3 * r + q, c->d.omSym)];      // sym6: the stored upper triangle, mirrored
        if (omega_n) {
          float v = 0.f;
          if (c->d.OmN) v = OmN[om_at(cap, i, 3 * r + q)];
          else if (cls == 1) v = c->d.omN[0][3 * r + q];
          else if (cls == 2) v = c->d.omN[1][3 * r + q];
          hon[(size_t)16 * i + r + 4 * q] = v;
        }
      }
    }
  }
  if (points) HIPCHK(ctx, hipMemcpy(points, hp.data(), hp.size() * 4, hipMemcpyDefault), PWN_HIP_ERR_COPY);
  if (normals) HIPCHK(ctx, hipMemcpy(normals, hn.data(), hn.size() * 4, hipMemcpyDefault), PWN_HIP_ERR_COPY);
  if (curvature) HIPCHK(ctx, hipMemcpy(curvature, hc.data(), hc.size() * 4, hipMemcpyDefault), PWN_HIP_ERR_COPY);
  if (omega_p) HIPCHK(ctx, hipMemcpy(omega_p, hop.data(), hop.size() * 4, hipMemcpyDefault), PWN_HIP_ERR_COPY);
  if (omega_n) HIPCHK(ctx, hipMemcpy(omega_n, hon.data(), hon.size() * 4, hipMemcpyDefault), PWN_HIP_ERR_COPY);
  return PWN_HIP_OK;
}
int pwn_hip_cloud_download_stats(pwn_hip_ctx* ctx, const pwn_hip_cloud* c, float* stats, float* eigenvalues, int* npoints) {
  if (!ctx || !c) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (!c->has_stats || !c->d.St) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "cloud was not converted with keep_stats");
  const int n = c->n_host;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  std::vector<float> St((size_t)n * 16);
  HIPCHK(ctx, hipMemcpy(St.data(), c->d.St, St.size() * 4, hipMemcpyDeviceToHost), PWN_HIP_ERR_COPY);
  std::vector<float> hs, he; std::vector<int> hn;
  if (stats) hs.assign((size_t)n * 16, 0.f);
  if (eigenvalues) he.resize((size_t)n * 3);
  if (npoints) hn.resize(n);
  for (int i = 0; i < n; ++i) {
    const float* s = &St[(size_t)16 * i];
    if (stats) {
      float* o = &hs[(size_t)16 * i];
      for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) o[r + 4 * q] = s[r + 3 * q];
      o[12] = s[12]; o[13] = s[13]; o[14] = s[14]; o[15] = 1.0f;
    }
    if (eigenvalues) { he[3 * i] = s[9]; he[3 * i + 1] = s[10]; he[3 * i + 2] = s[11]; }
    if (npoints) hn[i] = (int)s[15];
  }
  if (stats) HIPCHK(ctx, hipMemcpy(stats, hs.data(), hs.size() * 4, hipMemcpyDefault), PWN_HIP_ERR_COPY);
  if (eigenvalues) HIPCHK(ctx, hipMemcpy(eigenvalues, he.data(), he.size() * 4, hipMemcpyDefault), PWN_HIP_ERR_COPY);
  if (npoints) HIPCHK(ctx, hipMemcpy(npoints, hn.data(), hn.size() * 4, hipMemcpyDefault), PWN_HIP_ERR_COPY);
  return PWN_HIP_OK;
}
int pwn_hip_cloud_transform_in_place(pwn_hip_ctx* ctx, pwn_hip_cloud* c, const float T[16]) {
  if (!ctx || !c || !T) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  const Mat4 m = forced(T);
  if (is_identity(m)) return PWN_HIP_OK;                       // cloud.cpp:176
  cloud_changes(ctx, c);
  if (!c->d.OmN) {                                             // class matrices transform with the cloud
    for (int k = 0; k < 2; ++k) {
      float* om = c->d.omN[k]; float t1[9], o2[9];
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t1[3 * i + j] = dot3seq(m(i,0), om[0 + j], m(i,1), om[3 + j], m(i,2), om[6 + j]);
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o2[3 * i + j] = dot3seq(t1[3 * i], m(j,0), t1[3 * i + 1], m(j,1), t1[3 * i + 2], m(j,2));
      std::memcpy(om, o2, sizeof(o2));
    }
  }
  c->idx_valid = false;
  hipLaunchKernelGGL(k_cloud_transform, dim3((c->d.capacity + 255) / 256), dim3(256), 0, ctx->stream, c->d, m);
  {   // StatsVector / Gaussian3fVector::transformInPlace (stats.h:125-131, gaussian3.h:65-73)
    CloudDev d = c->d; if (!c->has_stats) d.St = nullptr;
    const int ng = c->sb.G ? c->n_gauss : 0, cnt = std::max(c->n_host, ng);
    if (cnt > 0 && (d.St || ng > 0)) hipLaunchKernelGGL(k_scene_transform, dim3((cnt + 255) / 256), dim3(256), 0, ctx->stream, d, c->sb, c->n_host, ng, m);
  }
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  return PWN_HIP_OK;
}

// ---- a cloud as ONE flat buffer (replication of a cloud to the other GPUs of a node: PwnCloser::processPartition matches one `current` cloud
// against every cloud of the other partition, pwn_tracker/pwn_closer.cpp:85-111; SURVEY.md 8(e): "replicate current's cloud to all GPUs") ----
namespace {
struct CloudFlatHeader {            // 256 bytes; sections start at multiples of 256 bytes
  uint32_t magic, version;
  int32_t n, omSym, hasOmN, idxValid, idxRows, idxCols;
  float clsThr, idxMinD, idxMaxD;
  float omN[2][9];
  float idxK[9];
  uint64_t offP3, offNc, offOm, offOmN, offIdx, total;
  unsigned char pad[256 - (8 * 4 + 3 * 4 + 18 * 4 + 9 * 4 + 6 * 8)];
};
static_assert(sizeof(CloudFlatHeader) == 256, "flat cloud header");
constexpr uint32_t kFlatMagic = 0x464E5750u;      // "PWNF"
size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }
// section offsets for n points (plane stride inside the flat buffer = n points, not the cloud's capacity)
void flat_layout(CloudFlatHeader& h, size_t n, int omSym, bool omN, size_t idxPixels) {
  size_t o = sizeof(CloudFlatHeader);
  h.offP3 = o; o += up256(n * 12);
  h.offNc = o; o += up256(n * 16);
  h.offOm = o; o += (size_t)om_planes(omSym) * up256(n * 12);
  h.offOmN = o; if (omN) o += 3 * up256(n * 12);
  h.offIdx = o; o += up256(idxPixels * 4);
  h.total = o;
}
}  // namespace
size_t pwn_hip_cloud_export_bound(int capacity, int omega_storage, int index_pixels, int with_omega_n) {
  if (capacity < 0 || index_pixels < 0) return 0;
  CloudFlatHeader h;
  flat_layout(h, (size_t)capacity, omega_storage == PWN_HIP_OMEGA_SYM6 ? 1 : 0, with_omega_n != 0, (size_t)index_pixels);
  return (size_t)h.total;
}
int pwn_hip_cloud_export(pwn_hip_ctx* ctx, const pwn_hip_cloud* c, void* dst, size_t dst_bytes, size_t* written) {
  if (!ctx || !c || (!dst && !written)) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  HIPCHK(ctx, hipSetDevice(ctx->device), PWN_HIP_ERR_NO_DEVICE);
  const size_t n = (size_t)std::min(c->n_host, c->d.capacity), cap = (size_t)c->d.capacity;
  // the header is staged in page-locked memory of the context: the copy below is asynchronous, and an error return further down must not
  // leave it reading a dead stack frame
  CloudFlatHeader& h = *(CloudFlatHeader*)ctx->flat_hdr_host; std::memset(&h, 0, sizeof(h));
  h.magic = kFlatMagic; h.version = 1; h.n = (int32_t)n; h.omSym = c->d.omSym; h.hasOmN = c->d.OmN ? 1 : 0;
  h.clsThr = c->d.clsThr; std::memcpy(h.omN, c->d.omN, sizeof(h.omN));
  const bool idx = c->idx_valid && c->idximg && (size_t)c->idx_rows * c->idx_cols <= c->idx_cap;
  h.idxValid = idx ? 1 : 0; h.idxRows = idx ? c->idx_rows : 0; h.idxCols = idx ? c->idx_cols : 0;
  h.idxMinD = c->idx_minD; h.idxMaxD = c->idx_maxD; std::memcpy(h.idxK, c->idx_K, sizeof(h.idxK));
  const size_t npx = idx ? (size_t)c->idx_rows * c->idx_cols : 0;
  flat_layout(h, n, c->d.omSym, c->d.OmN != nullptr, npx);
  if (written) *written = (size_t)h.total;
  if (!dst) return PWN_HIP_OK;                                   // size query
  if (dst_bytes < h.total) return fail(ctx, PWN_HIP_ERR_CAPACITY, "flat cloud buffer too small (pwn_hip_cloud_export_bound)");
  char* out = (char*)dst; hipStream_t st = ctx->stream;
  HIPCHK(ctx, copy_any(out, &h, sizeof(h), st), PWN_HIP_ERR_COPY);
  if (n > 0) {
    HIPCHK(ctx, copy_section(out + h.offP3, c->d.P3, n * 12, st), PWN_HIP_ERR_COPY);
    HIPCHK(ctx, copy_section(out + h.offNc, c->d.Nc, n * 16, st), PWN_HIP_ERR_COPY);
    for (int r = 0; r < om_planes(c->d.omSym); ++r)
      HIPCHK(ctx, copy_section(out + h.offOm + (size_t)r * up256(n * 12), c->d.Om + (size_t)r * cap * 3, n * 12, st), PWN_HIP_ERR_COPY);
    if (c->d.OmN) for (int r = 0; r < 3; ++r)
      HIPCHK(ctx, copy_section(out + h.offOmN + (size_t)r * up256(n * 12), c->d.OmN + (size_t)r * cap * 3, n * 12, st), PWN_HIP_ERR_COPY);
  }
  if (npx > 0) HIPCHK(ctx, copy_section(out + h.offIdx, c->idximg, npx * 4, st), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(st), PWN_HIP_ERR_COPY);      // the buffer is complete on return
  return PWN_HIP_OK;
}
int pwn_hip_cloud_import(pwn_hip_ctx* ctx, pwn_hip_cloud* c, const void* src, size_t src_bytes) {
  if (!ctx || !c || !src) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (src_bytes < sizeof(CloudFlatHeader)) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "flat cloud buffer shorter than its header");
  HIPCHK(ctx, hipSetDevice(ctx->device), PWN_HIP_ERR_NO_DEVICE);
  if (int rc = absorb_copies(ctx)) return rc;
  hipStream_t st = ctx->stream;
  CloudFlatHeader h;
  HIPCHK(ctx, copy_any(&h, src, sizeof(h), st), PWN_HIP_ERR_COPY);      // on the context's stream: ordered after pwn_hip_ctx_wait_stream
  HIPCHK(ctx, hipStreamSynchronize(st), PWN_HIP_ERR_COPY);
  if (h.magic != kFlatMagic || h.version != 1 || h.n < 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "not a flat cloud buffer (pwn_hip_cloud_export)");
  CloudFlatHeader want; std::memset(&want, 0, sizeof(want));
  const bool idx = h.idxValid != 0 && h.idxRows > 0 && h.idxCols > 0;
  const size_t n = (size_t)h.n, npx = idx ? (size_t)h.idxRows * h.idxCols : 0, cap = (size_t)c->d.capacity;
  flat_layout(want, n, h.omSym ? 1 : 0, h.hasOmN != 0, npx);
  if (want.offP3 != h.offP3 || want.offNc != h.offNc || want.offOm != h.offOm || want.offOmN != h.offOmN || want.offIdx != h.offIdx || want.total != h.total)
    return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "flat cloud buffer: inconsistent header");
  if (src_bytes < h.total) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "flat cloud buffer shorter than its header says");
  if (n > cap) return fail(ctx, PWN_HIP_ERR_CAPACITY, "cloud capacity smaller than the flat cloud");
  if ((h.omSym ? 1 : 0) != c->d.omSym) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "flat cloud and destination cloud differ in omega storage (exact9 / sym6)");
  cloud_changes(ctx, c);
  c->has_stats = false; c->n_gauss = 0; c->idx_valid = false;
  // from here on the destination's previous content is being overwritten: it reads as EMPTY (host size and device count) until every section
  // has arrived, so that a copy that fails half way leaves an empty cloud behind, not a mixture with the old sizes
  c->n_host = 0;
  HIPCHK(ctx, hipMemsetAsync(c->d.count, 0, sizeof(int), st), PWN_HIP_ERR_COPY);
  CloudFlatHeader& hp = *(CloudFlatHeader*)ctx->flat_hdr_host; hp = h;      // page-locked copy: source of the asynchronous count copy below
  if (h.hasOmN && !c->d.OmN) HIPCHK(ctx, hipMalloc((void**)&c->d.OmN, cap * 9 * sizeof(float)), PWN_HIP_ERR_ALLOCATION);
  if (!h.hasOmN && c->d.OmN) { (void)hipFree(c->d.OmN); c->d.OmN = nullptr; }      // the stream is idle (synchronised above)
  if (idx && c->idx_cap < npx) {
    if (c->idximg) (void)hipFree(c->idximg);
    c->idximg = nullptr; c->idx_cap = 0;
    HIPCHK(ctx, hipMalloc((void**)&c->idximg, npx * sizeof(int)), PWN_HIP_ERR_ALLOCATION);
    c->idx_cap = npx;
  }
  const char* in = (const char*)src;
  if (n > 0) {
    HIPCHK(ctx, copy_section(c->d.P3, in + h.offP3, n * 12, st), PWN_HIP_ERR_COPY);
    HIPCHK(ctx, copy_section(c->d.Nc, in + h.offNc, n * 16, st), PWN_HIP_ERR_COPY);
    for (int r = 0; r < om_planes(c->d.omSym); ++r)
      HIPCHK(ctx, copy_section(c->d.Om + (size_t)r * cap * 3, in + h.offOm + (size_t)r * up256(n * 12), n * 12, st), PWN_HIP_ERR_COPY);
    if (h.hasOmN) for (int r = 0; r < 3; ++r)
      HIPCHK(ctx, copy_section(c->d.OmN + (size_t)r * cap * 3, in + h.offOmN + (size_t)r * up256(n * 12), n * 12, st), PWN_HIP_ERR_COPY);
  }
  if (idx) HIPCHK(ctx, copy_section(c->idximg, in + h.offIdx, npx * 4, st), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, copy_any(c->d.count, &hp.n, sizeof(int), st), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(st), PWN_HIP_ERR_COPY);
  c->d.clsThr = h.clsThr; std::memcpy(c->d.omN, h.omN, sizeof(h.omN));
  c->n_host = h.n;
  if (idx) {
    c->idx_valid = true; c->idx_rows = h.idxRows; c->idx_cols = h.idxCols; c->idx_minD = h.idxMinD; c->idx_maxD = h.idxMaxD;
    std::memcpy(c->idx_K, h.idxK, sizeof(h.idxK));
  }
  return PWN_HIP_OK;
}

// ------------------------------------------------------------------------------------------ input conditioning
int pwn_hip_depth_u16_to_f32(pwn_hip_ctx* ctx, const uint16_t* src, float* dst, int n, float scale) {
  if (!ctx || !src || !dst || n < 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if ((size_t)n > ctx->N * ctx->max_batch) return fail(ctx, PWN_HIP_ERR_CAPACITY, "image larger than the context workspaces");
  if (int rc = absorb_copies(ctx)) return rc;
  const uint16_t* s = src; float* d = dst;
  if (!is_device_ptr(src)) { HIPCHK(ctx, hipMemcpyAsync(ctx->raw_ws, src, (size_t)n * 2, hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY); s = ctx->raw_ws; }
  if (!is_device_ptr(dst)) d = ctx->depth_ws;
  ctx->raw_host[0].src = s; ctx->raw_host[0].dst = d;
  HIPCHK(ctx, hipMemcpyAsync(ctx->raw_dev, ctx->raw_host, sizeof(RawDesc), hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY);
  hipLaunchKernelGGL(k_u16_to_f32, dim3(std::min((n + 255) / 256, 2048), 1), dim3(256), 0, ctx->stream, ctx->raw_dev, n, scale);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  if (d != dst) HIPCHK(ctx, hipMemcpyAsync(dst, d, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  return PWN_HIP_OK;
}
int pwn_hip_depth_f32_to_u16(pwn_hip_ctx* ctx, const float* src, uint16_t* dst, int n, float scale) {
  if (!ctx || !src || !dst || n < 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if ((size_t)n > ctx->N * ctx->max_batch) return fail(ctx, PWN_HIP_ERR_CAPACITY, "image larger than the context workspaces");
  if (int rc = absorb_copies(ctx)) return rc;
  const float* s = src; uint16_t* d = dst;
  if (!is_device_ptr(src)) { HIPCHK(ctx, hipMemcpyAsync(ctx->depth_ws, src, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY); s = ctx->depth_ws; }
  if (!is_device_ptr(dst)) d = ctx->raw_ws;
  hipLaunchKernelGGL(k_f32_to_u16, dim3(std::min((n + 255) / 256, 2048)), dim3(256), 0, ctx->stream, s, d, n, scale);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  if (d != dst) HIPCHK(ctx, hipMemcpyAsync(dst, d, (size_t)n * 2, hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  return PWN_HIP_OK;
}
int pwn_hip_depth_scale(pwn_hip_ctx* ctx, const float* src, int rows, int cols, int step, float max_depth_cov, float* dst) {
  if (!ctx || !src || !dst || step <= 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "bad argument");
  if (int rc = check_image(ctx, rows, cols)) return rc;
  const size_t n = (size_t)rows * cols; const int orows = rows / step, ocols = cols / step; const size_t on = (size_t)orows * ocols;
  if (int rc = absorb_copies(ctx)) return rc;
  const float* s = src; float* d = dst;
  if (!is_device_ptr(src)) { HIPCHK(ctx, hipMemcpyAsync(ctx->depth_ws, src, n * 4, hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY); s = ctx->depth_ws; }
  if (!is_device_ptr(dst)) d = ctx->io_ws;
  if (on > 0) hipLaunchKernelGGL(k_depth_scale, dim3((unsigned)((on + 255) / 256)), dim3(256), 0, ctx->stream, s, rows, cols, step, max_depth_cov, d);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  if (d != dst) HIPCHK(ctx, hipMemcpyAsync(dst, d, on * 4, hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  return PWN_HIP_OK;
}

// ---------------------------------------------------------------------------------------------- converter stages
static int stage_depth(pwn_hip_ctx* ctx, const float* depth, size_t N, const float** out) {
  if (int rc = absorb_copies(ctx)) return rc;
  if (is_device_ptr(depth)) { *out = depth; return PWN_HIP_OK; }
  HIPCHK(ctx, hipMemcpyAsync(ctx->depth_ws, depth, N * 4, hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY);
  *out = ctx->depth_ws;
  return PWN_HIP_OK;
}
int pwn_hip_unproject(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float T[16], const float* depth, int rows, int cols,
                      pwn_hip_cloud* cloud, int* index_image) {
  if (!ctx || !p || !depth || !cloud) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (int rc = check_image(ctx, rows, cols)) return rc;
  const size_t N = (size_t)rows * cols;
  cloud->idx_valid = false;                      // the cloud gets new points
  cloud_changes(ctx, cloud);
  const ConvertParams cp = make_convert_params(ctx, p, T, rows, cols, 0);
  const float* d = nullptr;
  if (int rc = stage_depth(ctx, depth, N, &d)) return rc;
  HIPCHK(ctx, hipMemsetAsync(cloud->d.Nc, 0, sizeof(float4) * (size_t)cloud->d.capacity, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipMemsetAsync(cloud->d.Om, 0, sizeof(float) * om_floats(cloud->d), ctx->stream), PWN_HIP_ERR_COPY);
  fill_frame(ctx, 0, 0, d, cloud->d, rows);
  HIPCHK(ctx, hipMemcpyAsync(ctx->frames_dev, ctx->frames_host, sizeof(FrameDesc), hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY);
  hipLaunchKernelGGL(k_row_count, dim3(rows, 1), dim3(256), 0, ctx->stream, ctx->frames_dev, cp);
  hipLaunchKernelGGL(k_row_offsets, dim3(1), dim3(1024), 0, ctx->stream, ctx->frames_dev, rows);
  hipLaunchKernelGGL(k_unproject, dim3(rows, 1), dim3(256), 0, ctx->stream, ctx->frames_dev, cp);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  if (index_image) HIPCHK(ctx, copy_any(index_image, ctx->frames_host[0].index, N * 4, ctx->stream), PWN_HIP_ERR_COPY);
  cloud->has_stats = false;
  pwn_hip_cloud* arr[1] = { cloud };
  return sync_and_counts(ctx, arr, 1);
}
int pwn_hip_project_intervals(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float* depth, int rows, int cols, int* interval_image) {
  if (!ctx || !p || !depth || !interval_image) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (int rc = check_image(ctx, rows, cols)) return rc;
  const size_t N = (size_t)rows * cols;
  const ConvertParams cp = make_convert_params(ctx, p, nullptr, rows, cols, 0);
  const float* d = nullptr;
  if (int rc = stage_depth(ctx, depth, N, &d)) return rc;
  CloudDev none; std::memset(&none, 0, sizeof(none)); none.count = ctx->scratch_count; none.capacity = 0;
  fill_frame(ctx, 0, 0, d, none, rows);
  HIPCHK(ctx, hipMemcpyAsync(ctx->frames_dev, ctx->frames_host, sizeof(FrameDesc), hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY);
  hipLaunchKernelGGL(k_row_count, dim3(rows, 1), dim3(256), 0, ctx->stream, ctx->frames_dev, cp);
  hipLaunchKernelGGL(k_row_offsets, dim3(1), dim3(1024), 0, ctx->stream, ctx->frames_dev, rows);
  hipLaunchKernelGGL(k_unproject, dim3(rows, 1), dim3(256), 0, ctx->stream, ctx->frames_dev, cp);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  HIPCHK(ctx, copy_any(interval_image, ctx->frames_host[0].interval, N * 4, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  return PWN_HIP_OK;
}
int pwn_hip_integral_image(pwn_hip_ctx* ctx, const int* index_image, const pwn_hip_cloud* cloud, int rows, int cols, float* out) {
  if (!ctx || !index_image || !cloud || !out) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (int rc = check_image(ctx, rows, cols)) return rc;
  if (int rc = absorb_copies(ctx)) return rc;      // caller pointers may be the destination of a queued pwn_hip_copy_async
  const size_t N = (size_t)rows * cols;
  fill_frame(ctx, 0, 0, nullptr, cloud->d, rows);
  HIPCHK(ctx, copy_any(ctx->frames_host[0].index, index_image, N * 4, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipMemcpyAsync(ctx->frames_dev, ctx->frames_host, sizeof(FrameDesc), hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY);
  hipLaunchKernelGGL(k_integral_rows, dim3((rows + kIR_Rows - 1) / kIR_Rows, 1), dim3(256), 0, ctx->stream, ctx->frames_dev, rows, cols);
  hipLaunchKernelGGL(k_integral_cols, dim3((cols + kIC_Block - 1) / kIC_Block, kIntegralChannels, 1), dim3(kIC_Block), 0, ctx->stream, ctx->frames_dev, rows, cols, (const int*)nullptr, (int*)nullptr);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  HIPCHK(ctx, copy_any(out, ctx->frames_host[0].integral, N * kIntegralChannels * 4, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  return PWN_HIP_OK;
}
int pwn_hip_convert(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float* depth, int rows, int cols, pwn_hip_cloud* cloud,
                    int* index_image, int* interval_image, int keep_stats) {
  const float* frames[1] = { depth };
  pwn_hip_cloud* clouds[1] = { cloud };
  if (int rc = convert_batch_impl<float>(ctx, p, frames, 0.f, 1, rows, cols, clouds, keep_stats, interval_image != nullptr)) return rc;
  const size_t N = (size_t)rows * cols;
  if (index_image) HIPCHK(ctx, hipMemcpy(index_image, ctx->frames_host[0].index, N * 4, hipMemcpyDefault), PWN_HIP_ERR_COPY);
  if (interval_image) HIPCHK(ctx, hipMemcpy(interval_image, ctx->frames_host[0].interval, N * 4, hipMemcpyDefault), PWN_HIP_ERR_COPY);
  return PWN_HIP_OK;
}
int pwn_hip_convert_scaled(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float* depth, int rows, int cols, int step, float max_depth_cov,
                           pwn_hip_cloud* cloud) {
  if (!ctx || !p || !depth || !cloud || step <= 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "bad argument");
  if (int rc = check_image(ctx, rows, cols)) return rc;
  const int orows = rows / step, ocols = cols / step;
  if (orows <= 0 || ocols <= 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "scaled image has zero size");
  if (int rc = absorb_copies(ctx)) return rc;
  const size_t n = (size_t)rows * cols, on = (size_t)orows * ocols;
  const float* src = depth;
  if (!is_device_ptr(depth)) { HIPCHK(ctx, hipMemcpyAsync(ctx->depth_ws, depth, n * 4, hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY); src = ctx->depth_ws; }
  float* scaled = ctx->io_ws;                                        // device scratch (N*16 floats)
  hipLaunchKernelGGL(k_depth_scale, dim3((unsigned)((on + 255) / 256)), dim3(256), 0, ctx->stream, src, rows, cols, step, max_depth_cov, scaled);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  const float* frames[1] = { scaled };
  pwn_hip_cloud* clouds[1] = { cloud };
  return convert_batch_impl<float>(ctx, p, frames, 0.f, 1, orows, ocols, clouds, 0);
}
static void async_convert_loop(AsyncConvert* a, int device) {
  (void)hipSetDevice(device);
  for (;;) {
    std::unique_lock<std::mutex> lk(a->m);
    a->cv.wait(lk, [a] { return a->has_job || a->quit; });
    if (!a->has_job) return;                                    // quit, nothing pending
    lk.unlock();
    const auto t_begin = std::chrono::steady_clock::now();
    int rc;
    if (a->raw) {
      const uint16_t* frames[1] = { a->raw }; pwn_hip_cloud* clouds[1] = { a->cloud };
      rc = convert_batch_impl<uint16_t>(a->job_ctx, &a->p, frames, a->raw_scale, 1, a->rows, a->cols, clouds, 0);
    } else {
      rc = pwn_hip_convert_scaled(a->job_ctx, &a->p, a->depth, a->rows, a->cols, a->step, a->max_depth_cov, a->cloud);
    }
    size_t written = 0;
    if (rc == PWN_HIP_OK && a->flat_dst) rc = pwn_hip_cloud_export(a->job_ctx, a->cloud, a->flat_dst, a->flat_bytes, &written);
    const float job_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    lk.lock();
    a->flat_written = written; a->job_ms = job_ms;
    a->rc = rc; a->err = rc == PWN_HIP_OK ? std::string() : a->job_ctx->err;
    a->has_job = false; a->done = true;
    lk.unlock();
    a->cv.notify_all();
  }
}
// one job for the helper thread: a float frame through DepthImage_scale + the converter, or a raw uint16 frame through the converter; then
// (flat_dst != nullptr) the cloud's flat form
static int async_convert_begin(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float* depth, const uint16_t* raw, float raw_scale, int rows, int cols,
                               int step, float max_depth_cov, pwn_hip_cloud* cloud, void* flat_dst, size_t flat_bytes);
int pwn_hip_convert_scaled_begin(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float* depth, int rows, int cols, int step, float max_depth_cov,
                                 pwn_hip_cloud* cloud) {
  if (!ctx || !p || !depth || !cloud || step <= 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "bad argument");
  return async_convert_begin(ctx, p, depth, nullptr, 0.f, rows, cols, step, max_depth_cov, cloud, nullptr, 0);
}
int pwn_hip_convert_export_begin(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const uint16_t* raw_frame, float depth_scale, int rows, int cols,
                                 pwn_hip_cloud* cloud, void* flat_dst, size_t flat_bytes) {
  if (!ctx || !p || !raw_frame || !cloud) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "bad argument");
  if (flat_dst) {      // checked here, where the caller can still act on it: the largest flat form this frame can have must fit
    const size_t need = pwn_hip_cloud_export_bound(std::min(cloud->d.capacity, rows * cols), cloud->d.omSym ? PWN_HIP_OMEGA_SYM6 : PWN_HIP_OMEGA_EXACT9, rows * cols, 0);
    if (flat_bytes < need) return fail(ctx, PWN_HIP_ERR_CAPACITY, "flat cloud buffer too small for a frame of this size (pwn_hip_cloud_export_bound)");
  }
  return async_convert_begin(ctx, p, nullptr, raw_frame, depth_scale, rows, cols, 1, 0.f, cloud, flat_dst, flat_bytes);
}
int pwn_hip_convert_export_end(pwn_hip_ctx* ctx, pwn_hip_cloud* cloud, size_t* written, float* job_ms) {
  if (written) *written = 0;
  if (job_ms) *job_ms = 0.f;
  const int rc = pwn_hip_convert_end(ctx, cloud);
  if (ctx && ctx->async) {
    if (written) *written = rc == PWN_HIP_OK ? ctx->async->flat_written : 0;
    if (job_ms) *job_ms = ctx->async->job_ms;
  }
  return rc;
}
static int async_convert_begin(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float* depth, const uint16_t* raw, float raw_scale, int rows, int cols,
                               int step, float max_depth_cov, pwn_hip_cloud* cloud, void* flat_dst, size_t flat_bytes) {
  if (int rc = check_image(ctx, rows, cols)) return rc;
  if (rows / step <= 0 || cols / step <= 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "scaled image has zero size");
  if (!ctx->async) {
    AsyncConvert* a = new AsyncConvert();
    if (int rc = ctx_create(&a->helper, ctx->device, ctx->max_rows, ctx->max_cols, 1, false)) { const std::string m = g_err; delete a; return fail(ctx, rc, "helper context: " + m); }
    a->worker = std::thread(async_convert_loop, a, ctx->device);
    ctx->async = a;
  }
  AsyncConvert* a = ctx->async;
  {
    std::unique_lock<std::mutex> lk(a->m);
    if (a->cloud || !a->done) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "a conversion is already in flight on this context (pwn_hip_convert_end first)");
    // Which helper: the raw-frame form runs beside a batch whose streams hold hundreds of queued launches -- on default priority the one-frame job is
    // served when a hardware queue it shares with the batch has drained (5.8 ms of wall time instead of 0.5); beside a single alignment (the tracker's
    // look-ahead) high priority is the wrong way round: it delays the alignment's short dependent kernels (3 070 -> 2 300 frames/s measured).
    if (raw && !a->helper_hi) {
      if (int rc = ctx_create(&a->helper_hi, ctx->device, ctx->max_rows, ctx->max_cols, 1, true)) return fail(ctx, rc, "helper context: " + g_err);
    }
    a->job_ctx = raw ? a->helper_hi : a->helper;
    // the caller's copies (pwn_hip_copy_async into a device frame) must have landed before the helper's stream reads the frame
    if (int rc = absorb_copies(ctx)) return rc;
    // a device frame the context's own stream may still be writing (pwn_hip_copy_async above): wait for it.  The raw-frame form does not wait --
    // it is what a caller queues from inside pwn_hip_ctx_set_enqueued_callback while the context's stream is busy with a batch -- and asks for a
    // frame that is complete when the call is made (include/pwn_hip.h)
    if (depth && is_device_ptr(depth)) HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
    cloud_changes(ctx, cloud);
    a->raw = raw; a->raw_scale = raw_scale; a->flat_dst = flat_dst; a->flat_bytes = flat_bytes; a->flat_written = 0; a->job_ms = 0.f;
    a->p = *p; a->depth = depth; a->rows = rows; a->cols = cols; a->step = step; a->max_depth_cov = max_depth_cov; a->cloud = cloud;
    a->rc = PWN_HIP_OK; a->err.clear();
    a->done = false; a->has_job = true;
  }
  a->cv.notify_all();
  return PWN_HIP_OK;
}
int pwn_hip_convert_end(pwn_hip_ctx* ctx, pwn_hip_cloud* cloud) {
  if (!ctx || !cloud) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  AsyncConvert* a = ctx->async;
  if (!a || a->cloud != cloud) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "no conversion of this cloud was begun on this context");
  std::unique_lock<std::mutex> lk(a->m);
  a->cv.wait(lk, [a] { return a->done; });
  a->cloud = nullptr;
  if (a->rc != PWN_HIP_OK) return fail(ctx, a->rc, a->err);
  return PWN_HIP_OK;
}
int pwn_hip_convert_batch(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float* const* depth_frames, int n, int rows, int cols,
                          pwn_hip_cloud* const* clouds) {
  return convert_batch_impl<float>(ctx, p, depth_frames, 0.f, n, rows, cols, clouds, 0);
}
int pwn_hip_convert_batch_u16(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const uint16_t* const* raw_frames, float depth_scale, int n,
                              int rows, int cols, pwn_hip_cloud* const* clouds) {
  return convert_batch_impl<uint16_t>(ctx, p, raw_frames, depth_scale, n, rows, cols, clouds, 0);
}

// ------------------------------------------------------------------------------------------------ aligner stages
int pwn_hip_project(pwn_hip_ctx* ctx, const float K[9], const float T[16], float min_distance, float max_distance, int rows, int cols,
                    const pwn_hip_cloud* cloud, int* index_image, float* depth_image) {
  if (!ctx || !K || !T || !cloud) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (min_distance < 0.f) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "min_distance must be >= 0");
  if (int rc = check_image(ctx, rows, cols)) return rc;
  const size_t N = (size_t)rows * cols;
  Mat4 KRt, iKRt; Mat3 iK;
  projector_matrices(mat3_from(K), mat4_from(T), KRt, iKRt, iK);
  HIPCHK(ctx, hipMemsetAsync(ctx->zref_ws, 0xFF, N * 8, ctx->stream), PWN_HIP_ERR_COPY);
  { StageTimer t(ctx, "project");
    hipLaunchKernelGGL(k_project_single, dim3((cloud->d.capacity + 255) / 256), dim3(256), 0, ctx->stream, cloud->d, KRt, min_distance, max_distance, rows, cols, ctx->zref_ws, kZTag0); }
  int* di = index_image ? (is_device_ptr(index_image) ? index_image : ctx->index_ws) : nullptr;
  float* dd = depth_image ? (is_device_ptr(depth_image) ? depth_image : ctx->depth_ws) : nullptr;
  hipLaunchKernelGGL(k_zbuf_resolve, dim3((unsigned)std::min<size_t>((N + 255) / 256, 2048)), dim3(256), 0, ctx->stream, ctx->zref_ws, (int)N, di, dd, kZTag0);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  if (index_image && di != index_image) HIPCHK(ctx, hipMemcpyAsync(index_image, di, N * 4, hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
  if (depth_image && dd != depth_image) HIPCHK(ctx, hipMemcpyAsync(depth_image, dd, N * 4, hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  ctx->img_valid = false;
  collect_stage_times(ctx);
  return PWN_HIP_OK;
}
int pwn_hip_correspondences(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, const pwn_hip_cloud* ref, const pwn_hip_cloud* cur,
                            const int* ref_index, const int* cur_index, const float T[16], int* corr, int* n_corr, int* n_cand) {
  if (!ctx || !p || !ref || !cur || !ref_index || !cur_index || !T || !corr || !n_corr) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (int rc = check_image(ctx, p->rows, p->cols)) return rc;
  const size_t N = (size_t)p->rows * p->cols;
  const AlignParams ap = make_align_params(ctx, p);
  int* ri = ctx->index_ws; int* ci = ctx->interval_ws;
  if (int rc = absorb_copies(ctx)) return rc;      // caller pointers may be the destination of a queued pwn_hip_copy_async
  HIPCHK(ctx, copy_any(ri, ref_index, N * 4, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, copy_any(ci, cur_index, N * 4, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipMemsetAsync(ctx->counters_dev, 0, 16 * sizeof(int), ctx->stream), PWN_HIP_ERR_COPY);
  hipLaunchKernelGGL(k_correspondence_image, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, ref->d, cur->d, ri, ci, ap, forced(T), ctx->corr_ws, ctx->counters_dev);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  std::vector<int2> img(N); int cand = 0;
  HIPCHK(ctx, hipMemcpyAsync(img.data(), ctx->corr_ws, N * sizeof(int2), hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipMemcpyAsync(&cand, ctx->counters_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  std::vector<int> out(2 * N, -1);
  int C = 0;
  for (size_t i = 0; i < N; ++i) if (img[i].x >= 0) { out[2 * C] = img[i].x; out[2 * C + 1] = img[i].y; ++C; }   // row-major order
  HIPCHK(ctx, hipMemcpy(corr, out.data(), 2 * N * sizeof(int), hipMemcpyDefault), PWN_HIP_ERR_COPY);
  *n_corr = C;
  if (n_cand) *n_cand = cand;
  return PWN_HIP_OK;
}
int pwn_hip_linearize(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, const pwn_hip_cloud* ref, const pwn_hip_cloud* cur, const int* corr,
                      int C, const float T[16], float* H, float* b, float* error, int* inliers) {
  if (!ctx || !p || !ref || !cur || (!corr && C > 0) || !T || C < 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if ((size_t)C > ctx->N) return fail(ctx, PWN_HIP_ERR_CAPACITY, "more correspondences than pixels");
  const AlignParams ap = make_align_params(ctx, p);
  if (int rc = absorb_copies(ctx)) return rc;      // caller pointers may be the destination of a queued pwn_hip_copy_async
  if (C > 0) HIPCHK(ctx, copy_any(ctx->corr_ws, corr, (size_t)C * sizeof(int2), ctx->stream), PWN_HIP_ERR_COPY);
  const int nb = std::max(1, align_nblocks(C));
  { StageTimer t(ctx, "corr_linearize");
    hipLaunchKernelGGL(k_linearize_list, dim3(nb), dim3(kAlignBlock), 0, ctx->stream, ref->d, cur->d, ctx->corr_ws, C, ap, forced(T), ctx->partials_ws); }
  hipLaunchKernelGGL(k_reduce_only, dim3(1), dim3(256), 0, ctx->stream, ctx->partials_ws, nb, ctx->solve_dev);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  SolveOut so;
  HIPCHK(ctx, hipMemcpyAsync(&so, ctx->solve_dev, sizeof(so), hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  if (H) std::memcpy(H, so.H, sizeof(so.H));
  if (b) std::memcpy(b, so.b, sizeof(so.b));
  if (error) *error = so.chi2;
  if (inliers) *inliers = so.inliers;
  collect_stage_times(ctx);
  return PWN_HIP_OK;
}

static void finish_match(const MatchAcc& a, pwn_hip_match_result* r) {
  r->image_non_zeros = (int)a.nonZeros;
  r->image_inliers = (int)a.inliers;
  r->image_outliers = (int)a.nonZeros - (int)a.inliers;
  r->image_reprojection_distance = match_reprojection_distance(a);               // the expression k_pack_records evaluates for the score words of a record
}
// What a caller may weave into a batch alignment (pwn_hip_convert_align_batch_u16: the conversion of a sub-batch's frames goes in front of its
// alignment on the same stream, so that one sub-batch converts while the other aligns and nothing waits for the host in between).
struct AlignHooks {
  std::function<int(int base, int m, int k, hipStream_t st)> pre_sub;      // before the kernels of pairs [base, base + m) (sub-batch k) are enqueued on st
  std::function<int()> before_sync;                                        // on ctx->stream, after the streams have joined
  std::function<int()> after_sync;                                         // after the final wait, before the results are filled in
};
// records (optional): n * PWN_HIP_RECORD_FLOATS floats, device or host, written by k_pack_records; pair_ids (optional, host): the id in
// record word 19 (else first_pair_id + i).  results may be NULL when records are asked for.
// robust: every projection of the call by the two-pass kernels (what align_batch_impl repeats a call with whose k_project gave up on a pixel)
static int align_batch_once(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, int n, pwn_hip_cloud* const* refs, pwn_hip_cloud* const* curs,
                            const float* guesses, pwn_hip_align_result* results, pwn_hip_match_result* scores, float match_threshold,
                            pwn_hip_align_statistics* statistics, const AlignHooks* hooks, float* records,
                            const int* pair_ids, int first_pair_id, bool match_records, bool robust) {
  if (!ctx || !p || !refs || !curs || (!results && !records) || n < 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (int rc = check_image(ctx, p->rows, p->cols)) return rc;
  ctx->last_align_fault = 0;
  if (robust) { if (int rc = ensure_zdepth(ctx)) return rc; }
  ctx->img_valid = false;                 // whatever happens below, the finder images of an earlier alignment are gone (set again on success)
  const bool want_scores = scores != nullptr || (records && match_records);      // the score words of the long records come from the same accumulators
  if (p->min_distance < 0.f) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "min_distance must be >= 0");
  const int nit = p->outer_iterations * p->inner_iterations;
  if (p->outer_iterations < 0 || p->inner_iterations < 0 || nit > PWN_HIP_MAX_ITERATIONS)
    return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "outer*inner iterations exceeds PWN_HIP_MAX_ITERATIONS");
  const int N = p->rows * p->cols;
  const AlignParams ap = make_align_params(ctx, p);
  const int nb = align_nblocks(N);
  ctx->stages.clear();
  const StreamPlan plan = make_plan(ctx, ctx->sub_pairs, n);
  const int sub = plan.sub;
  if (int rc = ensure_desc(ctx, n)) return rc;
  HIPCHK(ctx, hipEventRecord(ctx->t0, ctx->stream), PWN_HIP_ERR_LAUNCH);
  // Batch calls (the finder's images belong to single alignments) skip the projection of a current cloud whose own index image is that
  // projection's result; the matchClouds score then reads the current depth image off the cloud itself (k_match_score, curOwn).
  // (single alignments too: the finder's current images are then made on demand, see img_cur_lazy)
  const bool batch_shortcut = n >= 1 && ctx->index_shortcut && is_identity(forced(p->current_sensor_offset));
  std::vector<char> own_index((size_t)std::max(n, 1), 0), own_ref((size_t)std::max(n, 1), 0);
  const bool ident_ref = is_identity(forced(p->reference_sensor_offset));
  const bool direct_state = n <= 4;
  const int omSym = (n > 0 && curs[0]) ? curs[0]->d.omSym : 0;      // the linearizer reads the CURRENT cloud's information matrices (linearizer.cpp:52-53)
  // descriptors + initial states of all pairs; workspace slots are reused round-robin across sub-batches.  This loop runs with the device
  // idle (the call's first launch comes after it): what does not depend on the pair is computed once, and only the head of a state is cleared
  // (the traces behind `it` are written before they are read: k_solve_update stores entry `it`, every reader stops at `it`)
  Mat4 KRtCur0;
  { Mat4 iKRt0; Mat3 iK0; projector_matrices(ap.K, mat4_from(p->current_sensor_offset), KRtCur0, iKRt0, iK0); }
  for (int i = 0; i < n; ++i) {
    const pwn_hip_cloud* r = refs[i]; const pwn_hip_cloud* c = curs[i];
    if (!r || !c) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null cloud in batch");
    if (c->d.omSym != omSym) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "current clouds of one batch must share one omega storage (exact9 / sym6)");
    // the aligner's 32-bit z-buffer word indexes 2^21 points (every frame up to 1448 x 1448 pixels); larger clouds are scenes (merge, voxelize)
    if (std::min(r->n_host, r->d.capacity) > kMaxAlignerPoints || std::min(c->n_host, c->d.capacity) > kMaxAlignerPoints)
      return fail(ctx, PWN_HIP_ERR_CAPACITY, "Aligner::align: a cloud holds more than 2^21 points (index field of the aligner's z-buffer word)");
    const int slot = plan.slot0(i / sub) + i % sub;
    PairDesc& pd = ctx->pairs_host[i];
    pd.ref = r->d; pd.cur = c->d;
    pd.zref = ctx->z32ref_ws + (size_t)slot * ctx->N;
    pd.zcur = ctx->z32cur_ws + (size_t)slot * ctx->N;
    pd.curidx = ctx->curidx_ws + (size_t)slot * ctx->N;
    // the converter's own index image is what projecting the current cloud would give (see pwn_hip_cloud::idximg)
    pd.refidx0 = nullptr;
    own_index[i] = batch_shortcut && c->idx_valid && c->idx_rows == p->rows && c->idx_cols == p->cols && c->idx_minD == p->min_distance &&
                   c->idx_maxD == p->max_distance && std::memcmp(c->idx_K, p->K, sizeof(c->idx_K)) == 0;
    pd.partials = ctx->partials_ws + (size_t)slot * ctx->nblocks_max * kAccN;
    pd.state = ctx->state_ws + i;
    // a few pairs (latency path): k_solve_update writes the pose and the traces into the page-locked host copy itself, no copy back at the end;
    // batches copy the states back in one transfer (64 workgroups storing across PCIe in every solve launch cost more than that: 16 against 11 us per launch)
    pd.state_out = direct_state ? ctx->state_host + i : nullptr;
    pd.fault = ctx->align_fault_host;
    pd.zdepth = robust ? ctx->zdepth_ws + (size_t)slot * ctx->N : nullptr;
    // initial state: aligner.cpp:60-64,72-73,79,84
    PairState& st = ctx->state_host[i];
    std::memset(&st, 0, offsetof(PairState, chi2));
    Mat4 T = mat4_from(guesses ? guesses + 16 * (size_t)i : p->initial_guess);
    set_last_row(T);
    // first reference projection with an identity pose (identity guess and reference offset): it returns the reference cloud's own index
    // image, like the current cloud's; later iterations (and the last one, whose z-buffer the statistics pass re-reads) project as usual
    own_ref[i] = batch_shortcut && p->outer_iterations > 1 && ident_ref && is_identity(T) && r->idx_valid && r->idx_rows == p->rows &&
                 r->idx_cols == p->cols && r->idx_minD == p->min_distance && r->idx_maxD == p->max_distance && std::memcmp(r->idx_K, p->K, sizeof(r->idx_K)) == 0;
    st.T = T;
    st.invTcorr = iso_inverse(T);
    st.invT = st.invTcorr; set_last_row(st.invT);
    Mat4 iKRt; Mat3 iK;
    projector_matrices(ap.K, iso_mul(T, ap.refOffset), st.KRt, iKRt, iK);
    st.KRtLast = st.KRt;
    st.KRtCur = KRtCur0;
    st.it = 0;
  }
  // a sub-batch skips the projection kernels only if every pair of it can
  std::vector<char> sub_own((size_t)(n + sub - 1) / sub + 1, 1);
  for (int i = 0; i < n; ++i) if (!own_index[i]) sub_own[i / sub] = 0;
  bool any_own = false;
  for (int i = 0; i < n; ++i) if (sub_own[i / sub]) { ctx->pairs_host[i].curidx = curs[i]->idximg; any_own = true; }
  std::vector<char> sub_ownref(sub_own.size(), 1);
  for (int i = 0; i < n; ++i) if (!own_ref[i]) sub_ownref[i / sub] = 0;
  for (int i = 0; i < n; ++i) if (sub_ownref[i / sub]) ctx->pairs_host[i].refidx0 = refs[i]->idximg;
  if (n > 0) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->pairs_dev, ctx->pairs_host, sizeof(PairDesc) * n, hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY);
    HIPCHK(ctx, hipMemcpyAsync(ctx->state_ws, ctx->state_host, sizeof(PairState) * n, hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY);
    if (want_scores) HIPCHK(ctx, hipMemsetAsync(ctx->match_dev, 0, sizeof(MatchAcc) * n, ctx->stream), PWN_HIP_ERR_COPY);
  }
  // Every sub-batch takes its own block of tags (workspace slots are reused from sub-batch to sub-batch): tag0 for the
  // current-cloud projection (its own buffer) and tag0 - i for the reference projection of outer iteration i.  A call with
  // more sub-batches than the tag space holds falls back to the fixed tags and clears the slots of every sub-batch.
  const unsigned tagsPerSub = (unsigned)std::max(1, p->outer_iterations);
  const unsigned nsub = (unsigned)((n + sub - 1) / sub);
  if (tagsPerSub > kZ32Tag0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "more projections per alignment than z-buffer epoch tags");
  const bool rolling = (unsigned long long)nsub * tagsPerSub <= kZ32Tag0;
  unsigned tagBase = kZ32Tag0;
  if (rolling && nsub > 0) { if (int rc = take_tags32(ctx, nsub * tagsPerSub, &tagBase)) return rc; }
  // fixed tags leave words in the buffers that could beat (smaller tag wins) the tags a later rolling call draws: make that call clear first
  if (!rolling) ctx->z32tag_next = 0;
  unsigned tag0 = tagBase, lastRefTag = tag0 - (tagsPerSub - 1);
  if (int rc = plan_fork(ctx, plan)) return rc;
  for (int base = 0, kk = 0; base < n; base += sub, ++kk) {
    const int m = std::min(sub, n - base);
    const PairDesc* pr = ctx->pairs_dev + base;
    hipStream_t st = plan.stream(kk);
    const size_t s0 = (size_t)plan.slot0(kk);
    int maxcap_ref = 0, maxcap_cur = 0;
    for (int i = 0; i < m; ++i) { maxcap_ref = std::max(maxcap_ref, refs[base + i]->d.capacity); maxcap_cur = std::max(maxcap_cur, curs[base + i]->d.capacity); }
    const unsigned subTag0 = rolling ? tagBase - (unsigned)kk * tagsPerSub : kZ32Tag0;
    if (!rolling) {      // z-buffers start empty; slots are contiguous
      HIPCHK(ctx, hipMemsetAsync(ctx->z32ref_ws + s0 * ctx->N, 0xFF, (size_t)m * ctx->N * 4, st), PWN_HIP_ERR_COPY);
      HIPCHK(ctx, hipMemsetAsync(ctx->z32cur_ws + s0 * ctx->N, 0xFF, (size_t)m * ctx->N * 4, st), PWN_HIP_ERR_COPY);
    }
    if (s0 == 0 || kk == 0) {      // slot 0: what pwn_hip_align_images / pwn_hip_match_score read
      tag0 = subTag0; lastRefTag = subTag0 - (tagsPerSub - 1); ctx->img_pair = base;
      ctx->img_ref_cloud = refs[base]; ctx->img_cur_cloud = curs[base];
    }
    const unsigned subLastRefTag = subTag0 - (tagsPerSub - 1);
    if (hooks && hooks->pre_sub) { if (int rc = hooks->pre_sub(base, m, kk, st)) return rc; }
    if (!sub_own[kk]) {
      StageTimer t(ctx, "project_cur", st);
      if (int rc = launch_project(ctx, maxcap_cur, m, st, pr, ap, 1, subTag0, robust ? ctx->zdepth_ws + s0 * ctx->N : nullptr)) return rc;
      hipLaunchKernelGGL(k_resolve_cur, dim3(std::min((N + 255) / 256, 1024), m), dim3(256), 0, st, pr, N, subTag0); }
    for (int i = 0; i < p->outer_iterations; ++i) {
      const unsigned tag = subTag0 - (unsigned)i;      // epoch of this outer iteration's reference projection
      const int ownRef = (i == 0 && sub_ownref[kk]) ? 1 : 0;
      if (!ownRef) { StageTimer t(ctx, "project_ref", st);
        if (int rc = launch_project(ctx, maxcap_ref, m, st, pr, ap, 0, tag, robust ? ctx->zdepth_ws + s0 * ctx->N : nullptr)) return rc; }
      for (int k = 0; k < p->inner_iterations; ++k) {
        const bool lastInner = (k == p->inner_iterations - 1);
        { StageTimer t(ctx, "corr_linearize", st);
          // first inner pass: the linearizer's transform is bitwise the finder's (aligner.cpp:79,84)
          if (k == 0) launch_corr_linearize<true, false>(ctx, omSym, nb, m, st, pr, ap, tag, 0, ownRef);
          else launch_corr_linearize<false, false>(ctx, omSym, nb, m, st, pr, ap, tag, 0, ownRef); }
        { StageTimer t(ctx, "solve", st);
          hipLaunchKernelGGL(k_solve_update, dim3(m), dim3(256), 0, st, pr, ap, nb, lastInner ? 1 : 0, (lastInner && i == p->outer_iterations - 1) ? 1 : 0); }
      }
    }
    if (statistics && p->outer_iterations > 0) {
      // Aligner::_computeStatistics' extra Linearizer::update: the finder's correspondences of the last outer iteration
      // (tests with that iteration's transform) re-linearized at the final transform (aligner.cpp:165-170)
      StageTimer t(ctx, "statistics", st);
      launch_corr_linearize<false, true>(ctx, omSym, nb, m, st, pr, ap, subLastRefTag, 1, 0);   // full H for _computeStatistics
      hipLaunchKernelGGL(k_reduce_pairs, dim3(m), dim3(256), 0, st, pr, nb, ctx->stats_dev + base);
    }
    if (want_scores && p->outer_iterations > 0) {
      StageTimer t(ctx, "match_score", st);     // the z-buffers of this sub-batch still hold the finder's last depth images
      // blocks per pair: enough to fill the device together with the other pairs of the launch, few enough that the per-block atomics on the
      // pair's one record stay rare
      hipLaunchKernelGGL(k_match_score, dim3(std::min((N + 255) / 256, m >= 8 ? 64 : 256), m), dim3(256), 0, st, pr, N, subLastRefTag, subTag0, 1000.0f,
                         match_threshold, ctx->match_dev + base, sub_own[kk] ? 1 : 0);
    }
    HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  }
  if (int rc = plan_join(ctx, plan)) return rc;
  if (n > 0 && records) {
    // the records leave the device as the kernel wrote them: straight into the caller's device buffer (what an all-gather sends), or through the
    // context's own buffer into host memory
    const bool dev = is_device_ptr(records);
    const int rlen = match_records ? kMatchRecordFloats : kRecordFloats;
    if (!dev && ctx->records_cap < n) {
      if (ctx->records_ws) (void)hipFree(ctx->records_ws);
      ctx->records_ws = nullptr; ctx->records_cap = 0;
      HIPCHK(ctx, hipMalloc((void**)&ctx->records_ws, (size_t)std::max(n, 64) * kMatchRecordFloats * sizeof(float)), PWN_HIP_ERR_ALLOCATION);
      ctx->records_cap = std::max(n, 64);
    }
    if (pair_ids) {
      if (ctx->ids_cap < n) {
        if (ctx->ids_dev) (void)hipFree(ctx->ids_dev);
        ctx->ids_dev = nullptr; ctx->ids_cap = 0;
        HIPCHK(ctx, hipMalloc((void**)&ctx->ids_dev, (size_t)std::max(n, 64) * sizeof(int)), PWN_HIP_ERR_ALLOCATION);
        ctx->ids_cap = std::max(n, 64);
      }
      HIPCHK(ctx, copy_any(ctx->ids_dev, pair_ids, sizeof(int) * n, ctx->stream), PWN_HIP_ERR_COPY);
    }
    float* dst = dev ? records : ctx->records_ws;
    hipLaunchKernelGGL(k_pack_records, dim3(n), dim3(match_records ? 128 : 64), 0, ctx->stream, ctx->pairs_dev, pair_ids ? (const int*)ctx->ids_dev : nullptr, first_pair_id, dst,
                       match_records ? (const MatchAcc*)ctx->match_dev : nullptr);
    if (!dev) HIPCHK(ctx, hipMemcpyAsync(records, ctx->records_ws, sizeof(float) * rlen * n, hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
  }
  if (hooks && hooks->before_sync) { if (int rc = hooks->before_sync()) return rc; }
  if (n > 0 && scores) HIPCHK(ctx, hipMemcpyAsync(ctx->match_host, ctx->match_dev, sizeof(MatchAcc) * n, hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
  if (n > 0 && statistics) HIPCHK(ctx, hipMemcpyAsync(ctx->stats_host, ctx->stats_dev, sizeof(SolveOut) * n, hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
  // direct_state: T, it and the traces of state_host[i] were written by the last k_solve_update of each pair (PairDesc::state_out); with no
  // iterations it still holds the initial state
  if (n > 0 && !direct_state && results) HIPCHK(ctx, hipMemcpyAsync(ctx->state_host, ctx->state_ws, sizeof(PairState) * n, hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipEventRecord(ctx->t1, ctx->stream), PWN_HIP_ERR_LAUNCH);      // before the wait: recording it afterwards costs a second round trip per call
  // everything of the call is queued, nothing has been waited for: the caller's moment to queue what depends on it on other streams, or to prepare
  // the next call, while the device works (pwn_hip_ctx_set_enqueued_callback)
  if (ctx->enqueued_cb && n > 0) ctx->enqueued_cb(ctx->enqueued_user);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  if (hooks && hooks->after_sync) { if (int rc = hooks->after_sync()) return rc; }
  if (take_align_fault(ctx)) return fail(ctx, PWN_HIP_ERR_LAUNCH, kSettleMessage);
  for (int i = 0; i < n && scores; ++i) finish_match(ctx->match_host[i], &scores[i]);      // with or without `results` (pwn_hip_match_batch_records)
  for (int i = 0; i < n && results; ++i) {
    const PairState& st = ctx->state_host[i];
    pwn_hip_align_result& r = results[i];
    std::memset(&r, 0, sizeof(r));
    std::memcpy(r.T, st.T.m, sizeof(r.T));
    r.iterations = st.it;
    for (int k = 0; k < st.it && k < PWN_HIP_MAX_ITERATIONS; ++k) {
      r.chi2[k] = st.chi2[k]; r.iter_inliers[k] = st.inliers[k]; r.iter_correspondences[k] = st.ncorr[k]; r.iter_candidates[k] = st.ncand[k];
    }
    if (st.it > 0) { r.error = st.chi2[st.it - 1]; r.inliers = st.inliers[st.it - 1]; }
    r.n_reference = refs[i]->n_host; r.n_current = curs[i]->n_host;
    if (statistics) {
      pwn_hip_align_statistics& q = statistics[i];
      std::memset(&q, 0, sizeof(q));
      if (p->outer_iterations > 0) {
        const SolveOut& so = ctx->stats_host[i];
        std::memcpy(q.H, so.H, sizeof(q.H)); std::memcpy(q.b, so.b, sizeof(q.b)); q.error = so.chi2; q.inliers = so.inliers;
        compute_statistics(so.H, st.T, q.mean, q.omega, &q.translational_eigen_ratio, &q.rotational_eigen_ratio);
      }
    }
  }
  float ms = 0.f; (void)hipEventElapsedTime(&ms, ctx->t0, ctx->t1);
  for (int i = 0; i < n && results; ++i) results[i].total_time_ms = n > 0 ? ms / n : 0.f;
  // batches: no current z-buffer after a skipped projection; a single alignment makes it on demand
  ctx->img_rows = p->rows; ctx->img_cols = p->cols; ctx->img_valid = n > 0 && (!any_own || n == 1);
  ctx->img_cur_lazy = n == 1 && any_own; ctx->img_ap = ap; ctx->img_cur_capacity = n == 1 ? curs[0]->d.capacity : 0;
  ctx->img_ref_tag = lastRefTag; ctx->img_cur_tag = tag0;
  collect_stage_times(ctx);
  return PWN_HIP_OK;
}
// The call; and once more with the two-pass projection if one of its projections gave up on a pixel (z32_settle): nothing of the failed
// attempt is kept (states and descriptors are rebuilt, z-buffer tags move on; the hooks of a fused step convert the same frames into the same
// clouds again).  A fault in the repeat cannot happen (k_project_robust has no loop) and would be reported.
static int align_batch_impl(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, int n, pwn_hip_cloud* const* refs, pwn_hip_cloud* const* curs,
                            const float* guesses, pwn_hip_align_result* results, pwn_hip_match_result* scores, float match_threshold,
                            pwn_hip_align_statistics* statistics = nullptr, const AlignHooks* hooks = nullptr, float* records = nullptr,
                            const int* pair_ids = nullptr, int first_pair_id = 0, bool match_records = false) {
  int rc = align_batch_once(ctx, p, n, refs, curs, guesses, results, scores, match_threshold, statistics, hooks, records, pair_ids, first_pair_id, match_records, false);
  if (rc == PWN_HIP_ERR_LAUNCH && ctx && ctx->last_align_fault) {
    ++ctx->projection_fallbacks;
    rc = align_batch_once(ctx, p, n, refs, curs, guesses, results, scores, match_threshold, statistics, hooks, records, pair_ids, first_pair_id, match_records, true);
  }
  return rc;
}
int pwn_hip_align_batch(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, int n, pwn_hip_cloud* const* refs, pwn_hip_cloud* const* curs,
                        const float* guesses, pwn_hip_align_result* results) {
  return align_batch_impl(ctx, p, n, refs, curs, guesses, results, nullptr, 0.f);
}
int pwn_hip_match_batch(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, int n, pwn_hip_cloud* const* refs, pwn_hip_cloud* const* curs,
                        const float* guesses, float threshold, pwn_hip_align_result* results, pwn_hip_match_result* scores) {
  if (!scores) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null scores");
  return align_batch_impl(ctx, p, n, refs, curs, guesses, results, scores, threshold);
}
int pwn_hip_align_with_priors(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, const pwn_hip_cloud* ref, const pwn_hip_cloud* cur, int n_priors,
                              const pwn_hip_prior* priors, pwn_hip_align_result* result) {
  return pwn_hip_align_with_priors_ex(ctx, p, ref, cur, n_priors, priors, result, nullptr);
}
static int align_with_priors_once(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, const pwn_hip_cloud* ref, const pwn_hip_cloud* cur, int n_priors,
                                  const pwn_hip_prior* priors, pwn_hip_align_result* result, pwn_hip_align_statistics* statistics, bool robust);
int pwn_hip_align_with_priors_ex(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, const pwn_hip_cloud* ref, const pwn_hip_cloud* cur, int n_priors,
                                 const pwn_hip_prior* priors, pwn_hip_align_result* result, pwn_hip_align_statistics* statistics) {
  int rc = align_with_priors_once(ctx, p, ref, cur, n_priors, priors, result, statistics, false);
  if (rc == PWN_HIP_ERR_LAUNCH && n_priors > 0 && ctx && ctx->last_align_fault) {      // see align_batch_impl (which handles the prior-less case itself)
    ++ctx->projection_fallbacks;
    rc = align_with_priors_once(ctx, p, ref, cur, n_priors, priors, result, statistics, true);
  }
  return rc;
}
static int align_with_priors_once(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, const pwn_hip_cloud* ref, const pwn_hip_cloud* cur, int n_priors,
                                  const pwn_hip_prior* priors, pwn_hip_align_result* result, pwn_hip_align_statistics* statistics, bool robust) {
  if (n_priors <= 0) {
    pwn_hip_cloud* r[1] = { const_cast<pwn_hip_cloud*>(ref) };
    pwn_hip_cloud* c[1] = { const_cast<pwn_hip_cloud*>(cur) };
    return align_batch_impl(ctx, p, 1, r, c, nullptr, result, nullptr, 0.f, statistics);
  }
  if (!ctx || !p || !ref || !cur || !priors || !result) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (int rc = check_image(ctx, p->rows, p->cols)) return rc;
  if (p->min_distance < 0.f) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "min_distance must be >= 0");
  const int nit = p->outer_iterations * p->inner_iterations;
  if (p->outer_iterations < 0 || p->inner_iterations < 0 || nit > PWN_HIP_MAX_ITERATIONS)
    return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "outer*inner iterations exceeds PWN_HIP_MAX_ITERATIONS");
  std::vector<PriorHost> pr(n_priors);
  for (int i = 0; i < n_priors; ++i) {
    pr[i].kind = priors[i].kind; pr[i].mean = mat4_from(priors[i].mean);
    pr[i].invReference = priors[i].kind == 1 ? iso_inverse(mat4_from(priors[i].reference_transform)) : mat4_identity();
    std::memcpy(pr[i].information, priors[i].information, sizeof(pr[i].information));
  }
  if (std::min(ref->n_host, ref->d.capacity) > kMaxAlignerPoints || std::min(cur->n_host, cur->d.capacity) > kMaxAlignerPoints)
    return fail(ctx, PWN_HIP_ERR_CAPACITY, "Aligner::align: a cloud holds more than 2^21 points (index field of the aligner's z-buffer word)");
  const int N = p->rows * p->cols;
  const AlignParams ap = make_align_params(ctx, p);
  const int nb = align_nblocks(N);
  hipStream_t st = ctx->stream;
  PairDesc& pd = ctx->pairs_host[0];
  pd.ref = ref->d; pd.cur = cur->d;
  pd.zref = ctx->z32ref_ws; pd.zcur = ctx->z32cur_ws; pd.curidx = ctx->curidx_ws; pd.partials = ctx->partials_ws; pd.state = ctx->state_ws;
  pd.refidx0 = nullptr; pd.state_out = nullptr;
  ctx->last_align_fault = 0;
  if (robust) { if (int rc = ensure_zdepth(ctx)) return rc; }
  pd.fault = ctx->align_fault_host; pd.zdepth = robust ? ctx->zdepth_ws : nullptr;
  ctx->img_valid = false;
  ctx->img_pair = 0; ctx->img_ref_cloud = ref; ctx->img_cur_cloud = cur;
  PairState& hs = ctx->state_host[0];
  std::memset(&hs, 0, sizeof(hs));
  Mat4 T = mat4_from(p->initial_guess); set_last_row(T);
  Mat4 iKRt; Mat3 iK;
  projector_matrices(ap.K, mat4_from(p->current_sensor_offset), hs.KRtCur, iKRt, iK);
  HIPCHK(ctx, hipEventRecord(ctx->t0, st), PWN_HIP_ERR_LAUNCH);
  HIPCHK(ctx, hipMemcpyAsync(ctx->pairs_dev, ctx->pairs_host, sizeof(PairDesc), hipMemcpyHostToDevice, st), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipMemcpyAsync(ctx->state_ws, &hs, sizeof(PairState), hipMemcpyHostToDevice, st), PWN_HIP_ERR_COPY);
  // z-buffer tags come from the context's running supply like the batch path's (a smaller tag wins atomicMin: fixed tags would
  // leave words behind that beat a later call's)
  unsigned tag0 = kZ32Tag0;
  if (int rc = take_tags32(ctx, (unsigned)std::max(1, p->outer_iterations), &tag0)) return rc;
  if (int rc = launch_project(ctx, cur->d.capacity, 1, st, ctx->pairs_dev, ap, 1, tag0, pd.zdepth)) return rc;
  hipLaunchKernelGGL(k_resolve_cur, dim3(std::min((N + 255) / 256, 1024), 1), dim3(256), 0, st, ctx->pairs_dev, N, tag0);
  std::memset(result, 0, sizeof(*result));
  int it = 0;
  for (int i = 0; i < p->outer_iterations; ++i) {
    const unsigned tag = tag0 - (unsigned)i;
    set_last_row(T);                                                                 // aligner.cpp:72
    hs.T = T; hs.invTcorr = iso_inverse(T);
    projector_matrices(ap.K, iso_mul(T, ap.refOffset), hs.KRt, iKRt, iK);            // :73
    hs.KRtLast = hs.KRt;
    Mat4 invT = iso_inverse(T);                                                      // :84
    for (int k = 0; k < p->inner_iterations; ++k, ++it) {
      set_last_row(invT);                                                            // :86
      hs.invT = invT;
      HIPCHK(ctx, hipMemcpyAsync(ctx->state_ws, &hs, sizeof(PairState), hipMemcpyHostToDevice, st), PWN_HIP_ERR_COPY);
      if (k == 0) { if (int rc = launch_project(ctx, ref->d.capacity, 1, st, ctx->pairs_dev, ap, 0, tag, pd.zdepth)) return rc; }
      if (k == 0) launch_corr_linearize<true, true>(ctx, cur->d.omSym, nb, 1, st, ctx->pairs_dev, ap, tag, 0, 0);
      else launch_corr_linearize<false, true>(ctx, cur->d.omSym, nb, 1, st, ctx->pairs_dev, ap, tag, 0, 0);
      hipLaunchKernelGGL(k_reduce_pairs, dim3(1), dim3(256), 0, st, ctx->pairs_dev, nb, ctx->stats_dev);
      HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
      HIPCHK(ctx, hipMemcpyAsync(ctx->stats_host, ctx->stats_dev, sizeof(SolveOut), hipMemcpyDeviceToHost, st), PWN_HIP_ERR_COPY);
      HIPCHK(ctx, hipStreamSynchronize(st), PWN_HIP_ERR_LAUNCH);
      const SolveOut& so = ctx->stats_host[0];
      result->chi2[it] = so.chi2; result->iter_inliers[it] = so.inliers; result->iter_correspondences[it] = so.ncorr; result->iter_candidates[it] = so.ncand;
      float H[36], b[6];
      std::memcpy(H, so.H, sizeof(H)); std::memcpy(b, so.b, sizeof(b));
      for (int d = 0; d < 6; ++d) H[d + 6 * d] = H[d + 6 * d] + 1.0f;                // :92
      for (int d = 0; d < 6; ++d) H[d + 6 * d] = H[d + 6 * d] + 1000.0f;             // :94
      for (int j = 0; j < n_priors; ++j) prior_accumulate(pr[j], invT, H, b);        // :97-108
      float nbv[6], dx[6];
      for (int d = 0; d < 6; ++d) nbv[d] = -b[d];
      ldlt_solve6(H, nbv, dx);                                                       // :110
      invT = iso_mul(v2t(dx), invT);                                                 // :111-112
    }
    T = iso_inverse(invT);                                                           // :115-117
    float v[6]; t2v(T, v); T = v2t(v); set_last_row(T);
  }
  const unsigned lastRefTag = tag0 - (unsigned)std::max(0, p->outer_iterations - 1);
  if (statistics) {
    // Aligner::_computeStatistics (aligner.cpp:127,152-199) runs after the loop whether or not priors exist: one more
    // Linearizer::update at the final transform on the finder's last correspondences, H + I without the prior terms (:168-170)
    std::memset(statistics, 0, sizeof(*statistics));
    if (p->outer_iterations > 0) {
      hs.invTcorrPrev = hs.invTcorr;
      hs.invT = iso_inverse(T); set_last_row(hs.invT);                                // :165-167
      HIPCHK(ctx, hipMemcpyAsync(ctx->state_ws, &hs, sizeof(PairState), hipMemcpyHostToDevice, st), PWN_HIP_ERR_COPY);
      launch_corr_linearize<false, true>(ctx, cur->d.omSym, nb, 1, st, ctx->pairs_dev, ap, lastRefTag, 1, 0);
      hipLaunchKernelGGL(k_reduce_pairs, dim3(1), dim3(256), 0, st, ctx->pairs_dev, nb, ctx->stats_dev);
      HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
      HIPCHK(ctx, hipMemcpyAsync(ctx->stats_host, ctx->stats_dev, sizeof(SolveOut), hipMemcpyDeviceToHost, st), PWN_HIP_ERR_COPY);
      HIPCHK(ctx, hipStreamSynchronize(st), PWN_HIP_ERR_LAUNCH);
      const SolveOut& so = ctx->stats_host[0];
      std::memcpy(statistics->H, so.H, sizeof(statistics->H)); std::memcpy(statistics->b, so.b, sizeof(statistics->b));
      statistics->error = so.chi2; statistics->inliers = so.inliers;
      compute_statistics(so.H, T, statistics->mean, statistics->omega, &statistics->translational_eigen_ratio, &statistics->rotational_eigen_ratio);
    }
  }
  HIPCHK(ctx, hipEventRecord(ctx->t1, st), PWN_HIP_ERR_LAUNCH);
  HIPCHK(ctx, hipEventSynchronize(ctx->t1), PWN_HIP_ERR_LAUNCH);
  if (take_align_fault(ctx)) return fail(ctx, PWN_HIP_ERR_LAUNCH, kSettleMessage);
  float ms = 0.f; (void)hipEventElapsedTime(&ms, ctx->t0, ctx->t1);
  std::memcpy(result->T, T.m, sizeof(result->T));
  result->iterations = it; result->total_time_ms = ms;
  if (it > 0) { result->error = result->chi2[it - 1]; result->inliers = result->iter_inliers[it - 1]; }
  result->n_reference = ref->n_host; result->n_current = cur->n_host;
  ctx->img_rows = p->rows; ctx->img_cols = p->cols; ctx->img_valid = true; ctx->img_cur_lazy = false;
  ctx->img_ref_tag = lastRefTag; ctx->img_cur_tag = tag0;
  return PWN_HIP_OK;
}
int pwn_hip_align_batch_ex(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, int n, pwn_hip_cloud* const* refs, pwn_hip_cloud* const* curs,
                           const float* guesses, pwn_hip_align_result* results, float threshold, pwn_hip_match_result* scores,
                           pwn_hip_align_statistics* statistics) {
  return align_batch_impl(ctx, p, n, refs, curs, guesses, results, scores, threshold, statistics);
}
int pwn_hip_align_batch_records(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, int n, pwn_hip_cloud* const* refs, pwn_hip_cloud* const* curs,
                                const float* guesses, const int* pair_ids, int first_pair_id, pwn_hip_align_result* results, float* records) {
  if (!records) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null records");
  return align_batch_impl(ctx, p, n, refs, curs, guesses, results, nullptr, 0.f, nullptr, nullptr, records, pair_ids, first_pair_id);
}
int pwn_hip_match_batch_records(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, int n, pwn_hip_cloud* const* refs, pwn_hip_cloud* const* curs,
                                const float* guesses, float threshold, const int* pair_ids, int first_pair_id, pwn_hip_align_result* results,
                                pwn_hip_match_result* scores, float* records) {
  if (!records) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null records");
  return align_batch_impl(ctx, p, n, refs, curs, guesses, results, scores, threshold, nullptr, nullptr, records, pair_ids, first_pair_id, true);
}
int pwn_hip_convert_align_batch_u16(pwn_hip_ctx* ctx, const pwn_hip_converter_params* cp, const pwn_hip_aligner_params* ap, int n,
                                    const uint16_t* const* ref_frames, const uint16_t* const* cur_frames, float depth_scale, int rows, int cols,
                                    pwn_hip_cloud* const* refs, pwn_hip_cloud* const* curs, const float* guesses, const int* pair_ids, int first_pair_id,
                                    pwn_hip_align_result* results, float* records) {
  if (!ctx || !cp || !ap || !ref_frames || !cur_frames || !refs || !curs || (!results && !records) || n < 0) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (int rc = check_image(ctx, rows, cols)) return rc;
  if (ap->rows != rows || ap->cols != cols) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "aligner image size differs from the frames'");
  // the clouds' host-side sizes are those of their PREVIOUS content while the step is being queued: bound what the conversion can produce instead
  if ((size_t)rows * cols > (size_t)kMaxAlignerPoints)
    return fail(ctx, PWN_HIP_ERR_CAPACITY, "Aligner::align: frames of more than 2^21 pixels (index field of the aligner's z-buffer word)");
  {   // sub-batches run on different streams: a cloud that appears twice would be written by two of them at once
    std::vector<const pwn_hip_cloud*> all; all.reserve((size_t)2 * std::max(n, 0));
    for (int i = 0; i < n; ++i) { all.push_back(refs[i]); all.push_back(curs[i]); }
    std::sort(all.begin(), all.end());
    if (std::adjacent_find(all.begin(), all.end()) != all.end()) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "the 2 n clouds of a step must be distinct");
  }
  if (int rc = absorb_copies(ctx)) return rc;
  if (int rc = ensure_desc(ctx, 2 * n)) return rc;       // once, before anything is queued: growing the descriptor arrays waits for the stream
  // the sub-batches and streams the alignment will use (align_batch_impl makes the same plan); the frames of sub-batch k -- its reference
  // frames, then its current frames -- are converted on k's stream in front of k's alignment, in launches of at most sub_frames frames that
  // reuse the stream's own block of frame slots
  const StreamPlan plan = make_plan(ctx, ctx->sub_pairs, n);
  const int sub = plan.sub;
  const int fslots = std::max(1, std::min(ctx->sub_frames, ctx->max_batch / plan.ns));
  std::vector<const uint16_t*> frames((size_t)2 * std::max(n, 0));
  std::vector<pwn_hip_cloud*> clouds((size_t)2 * std::max(n, 0));
  std::vector<int> slot((size_t)2 * std::max(n, 0));
  for (int base = 0, k = 0; base < n; base += sub, ++k) {
    const int m = std::min(sub, n - base);
    for (int j = 0; j < m; ++j) {
      frames[2 * base + j] = ref_frames[base + j]; clouds[2 * base + j] = refs[base + j];
      frames[2 * base + m + j] = cur_frames[base + j]; clouds[2 * base + m + j] = curs[base + j];
    }
    for (int f = 0; f < 2 * m; ++f) slot[2 * base + f] = (k % plan.ns) * fslots + f % fslots;
  }
  ConvertJob job;
  if (int rc = convert_prepare<uint16_t>(ctx, cp, frames.data(), depth_scale, 2 * n, rows, cols, clouds.data(), 0, false, slot, false, job)) return rc;
  AlignHooks hooks;
  hooks.pre_sub = [&](int base, int m, int, hipStream_t st) -> int {
    for (int f = 0; f < 2 * m; f += fslots) {
      const int cnt = std::min(fslots, 2 * m - f);
      if (job.host_input) { if (int rc = convert_stage_frames(ctx, job, 2 * base + f, cnt, st)) return rc; }
      if (int rc = launch_convert(ctx, job.cp, 2 * base + f, cnt, st)) return rc;
    }
    return PWN_HIP_OK;
  };
  hooks.before_sync = [&]() -> int { return counts_enqueue(ctx, 2 * n); };
  hooks.after_sync = [&]() -> int { return counts_apply(ctx, clouds.data(), 2 * n); };
  const int rc = align_batch_impl(ctx, ap, n, refs, curs, guesses, results, nullptr, 0.f, nullptr, &hooks, records, pair_ids, first_pair_id);
  if (rc != PWN_HIP_OK && ctx->last_convert_fault == 1 && !ctx->in_step_retry) {      // a hand-over time-out (see convert_batch_impl): the step once more
    ++ctx->convert_retries;
    ctx->in_step_retry = true;
    const int rc2 = pwn_hip_convert_align_batch_u16(ctx, cp, ap, n, ref_frames, cur_frames, depth_scale, rows, cols, refs, curs, guesses, pair_ids, first_pair_id, results, records);
    ctx->in_step_retry = false;
    return rc2;
  }
  return rc;
}
void pwn_hip_compute_statistics(const float H[36], const float T[16], float mean[6], float omega[36], float* tr, float* rr) {
  compute_statistics(H, mat4_from(T), mean, omega, tr, rr);
}
int pwn_hip_match_score(pwn_hip_ctx* ctx, float threshold, pwn_hip_match_result* out) {
  if (!ctx || !out) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "null argument");
  if (!ctx->img_valid) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "no alignment has run on this context");
  const int N = ctx->img_rows * ctx->img_cols;
  // the descriptor of the pair in slot 0 is still in pairs_dev (clouds, z-buffers, state with the projection matrices)
  HIPCHK(ctx, hipMemsetAsync(ctx->match_dev, 0, sizeof(MatchAcc), ctx->stream), PWN_HIP_ERR_COPY);
  hipLaunchKernelGGL(k_match_score, dim3(std::min((N + 255) / 256, 256), 1), dim3(256), 0, ctx->stream, ctx->pairs_dev + ctx->img_pair, N, ctx->img_ref_tag,
                     ctx->img_cur_tag, 1000.0f, threshold, ctx->match_dev, ctx->img_cur_lazy ? 1 : 0);
  HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
  HIPCHK(ctx, hipMemcpyAsync(ctx->match_host, ctx->match_dev, sizeof(MatchAcc), hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  finish_match(ctx->match_host[0], out);
  return PWN_HIP_OK;
}
int pwn_hip_align(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, const pwn_hip_cloud* ref, const pwn_hip_cloud* cur, pwn_hip_align_result* result) {
  pwn_hip_cloud* r[1] = { const_cast<pwn_hip_cloud*>(ref) };
  pwn_hip_cloud* c[1] = { const_cast<pwn_hip_cloud*>(cur) };
  return pwn_hip_align_batch(ctx, p, 1, r, c, nullptr, result);
}
int pwn_hip_align_images(pwn_hip_ctx* ctx, int* ref_index, float* ref_depth, int* cur_index, float* cur_depth) {
  if (!ctx) return fail(nullptr, PWN_HIP_ERR_INVALID_ARGUMENT, "null ctx");
  if (!ctx->img_valid) return fail(ctx, PWN_HIP_ERR_INVALID_ARGUMENT, "no alignment has run on this context");
  const size_t N = (size_t)ctx->img_rows * ctx->img_cols;
  if (ctx->img_cur_lazy && (cur_index || cur_depth)) {      // the projection the alignment skipped, with the tag it had reserved for it
    // two-pass form straight away: one projection of one cloud, off the hot path, and no repeat to arrange (the pair's descriptor on the device
    // gets the depth image's address first)
    if (int rc = ensure_zdepth(ctx)) return rc;
    ctx->pairs_host[ctx->img_pair].zdepth = ctx->zdepth_ws;
    HIPCHK(ctx, hipMemcpyAsync(&ctx->pairs_dev[ctx->img_pair].zdepth, &ctx->pairs_host[ctx->img_pair].zdepth, sizeof(unsigned*), hipMemcpyHostToDevice, ctx->stream), PWN_HIP_ERR_COPY);
    if (int rc = launch_project(ctx, ctx->img_cur_capacity, 1, ctx->stream, ctx->pairs_dev + ctx->img_pair, ctx->img_ap, 1, ctx->img_cur_tag, ctx->zdepth_ws)) return rc;
    HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
    ctx->img_cur_lazy = false;
  }
  for (int pass = 0; pass < 2; ++pass) {
    int* oi = pass == 0 ? ref_index : cur_index; float* od = pass == 0 ? ref_depth : cur_depth;
    if (!oi && !od) continue;
    int* di = oi ? (is_device_ptr(oi) ? oi : ctx->index_ws) : nullptr;
    float* dd = od ? (is_device_ptr(od) ? od : ctx->depth_ws) : nullptr;
    hipLaunchKernelGGL(k_pair_images, dim3((unsigned)std::min<size_t>((N + 255) / 256, 2048)), dim3(256), 0, ctx->stream, ctx->pairs_dev + ctx->img_pair, pass,
                       pass == 0 ? ctx->img_ref_tag : ctx->img_cur_tag, (int)N, di, dd);
    HIPCHK(ctx, hipGetLastError(), PWN_HIP_ERR_LAUNCH);
    if (oi && di != oi) HIPCHK(ctx, hipMemcpyAsync(oi, di, N * 4, hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
    if (od && dd != od) HIPCHK(ctx, hipMemcpyAsync(od, dd, N * 4, hipMemcpyDeviceToHost, ctx->stream), PWN_HIP_ERR_COPY);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream), PWN_HIP_ERR_LAUNCH);
  }
  return PWN_HIP_OK;
}

// ------------------------------------------------------------------------------------------------------- helpers
void pwn_hip_projector_matrices(const float K[9], const float T[16], float KRt[16], float iKRt[16], float iK[9]) {
  Mat4 a, b; Mat3 c;
  projector_matrices(mat3_from(K), mat4_from(T), a, b, c);
  if (KRt) std::memcpy(KRt, a.m, sizeof(a.m));
  if (iKRt) std::memcpy(iKRt, b.m, sizeof(b.m));
  if (iK) std::memcpy(iK, c.m, sizeof(c.m));
}
int pwn_hip_project_point(const float K[9], const float T[16], float min_distance, float max_distance, const float p[3], int* x, int* y, float* d) {
  if (!K || !T || !p) return 0;
  Mat4 KRt, iKRt; Mat3 iK;
  projector_matrices(mat3_from(K), mat4_from(T), KRt, iKRt, iK);
  // _project: ip = KRt * p; d = ip.z; ip *= 1 / d; round (the expressions of z32_insert / project_point in pwn_kernels.h)
  const float ix = dot4seq(KRt(0,0), p[0], KRt(0,1), p[1], KRt(0,2), p[2], KRt(0,3), 1.0f);
  const float iy = dot4seq(KRt(1,0), p[0], KRt(1,1), p[1], KRt(1,2), p[2], KRt(1,3), 1.0f);
  const float dd = dot4seq(KRt(2,0), p[0], KRt(2,1), p[1], KRt(2,2), p[2], KRt(2,3), 1.0f);
  if (d) *d = dd;
  if (dd < min_distance || dd > max_distance) return 0;
  const float inv = 1.0f / dd;
  const float fx = roundf(ix * inv), fy = roundf(iy * inv);
  // the reference converts whatever comes out to int (undefined for values an int cannot hold); saturate instead
  const float lim = 2147483520.0f;
  if (x) *x = (int)(fx > lim ? lim : (fx < -lim ? -lim : fx));
  if (y) *y = (int)(fy > lim ? lim : (fy < -lim ? -lim : fy));
  return 1;
}
int pwn_hip_unproject_pixel(const float K[9], const float T[16], float min_distance, float max_distance, int x, int y, float d, float p[3]) {
  if (!K || !T || !p) return 0;
  if (d < min_distance || d > max_distance) return 0;
  Mat4 KRt, iKRt; Mat3 iK;
  projector_matrices(mat3_from(K), mat4_from(T), KRt, iKRt, iK);
  const float a = (float)x * d, b = (float)y * d;                      // _unProject: iKRt * (x d, y d, d, 1): the expressions of k_unproject
  p[0] = dot4seq(iKRt(0,0), a, iKRt(0,1), b, iKRt(0,2), d, iKRt(0,3), 1.0f);
  p[1] = dot4seq(iKRt(1,0), a, iKRt(1,1), b, iKRt(1,2), d, iKRt(1,3), 1.0f);
  p[2] = dot4seq(iKRt(2,0), a, iKRt(2,1), b, iKRt(2,2), d, iKRt(2,3), 1.0f);
  return 1;
}
int pwn_hip_project_interval(const float K[9], float min_distance, float max_distance, float d, float world_radius) {
  if (!K) return -1;
  if (d < min_distance || d > max_distance) return -1;
  const Mat3 Km = mat3_from(K);
  // _projectInterval: p = K * (R, R, 0); p *= 1 / d; the larger of x, y truncated (make_convert_params / k_unproject evaluate the same)
  const float ivx = dot3seq(Km(0,0), world_radius, Km(0,1), world_radius, Km(0,2), 0.f);
  const float ivy = dot3seq(Km(1,0), world_radius, Km(1,1), world_radius, Km(1,2), 0.f);
  const float inv = 1.0f / d;
  const float px = ivx * inv, py = ivy * inv;
  return (px > py) ? (int)px : (int)py;
}
void pwn_hip_iso_inverse(const float T[16], float out[16]) { const Mat4 r = iso_inverse(mat4_from(T)); std::memcpy(out, r.m, sizeof(r.m)); }
void pwn_hip_iso_mul(const float A[16], const float B[16], float out[16]) { const Mat4 r = iso_mul(mat4_from(A), mat4_from(B)); std::memcpy(out, r.m, sizeof(r.m)); }
void pwn_hip_v2t(const float v[6], float T[16]) { const Mat4 t = v2t(v); std::memcpy(T, t.m, sizeof(t.m)); }
void pwn_hip_t2v(const float T[16], float v[6]) { t2v(mat4_from(T), v); }
void pwn_hip_ldlt_solve6(const float H[36], const float b[6], float x[6]) { ldlt_solve6(H, b, x); }

}  // extern "C"

#include "pwn_scene_capi.h"
