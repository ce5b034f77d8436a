// pwn_hip_partition_app -- PwnCloser::processPartition (pwn_tracker/pwn_closer.cpp:85-111) over the GPUs of one node in native code: one process per
// GPU, the C-ABI of include/pwn_hip.h through the C++ mirror, RCCL for the two collectives.  What INTEGRATION.md section 1 sketches, compiled and run:
//
//   every rank   converts and keeps its contiguous shard of the partition's keyframes (the PwnCache of its GPU: pwn_tracker_cache.cpp:24-51)
//   per step     ONE library call: matchFrames' data path for the rank's shard -- matchClouds(current, other, iT * other.T) = Aligner::align from the
//                odometry guess with the z translation zeroed + the depth-agreement score (pwn_matcher_base.cpp:88-183) -- pwn_hip_match_batch_records,
//                288-byte records written on the device.  Everything else is queued from INSIDE that call (pwn_hip_ctx_set_enqueued_callback: after its
//                device work is queued, before it waits), on one RCCL stream and a second small context, so that it runs beside the matches:
//                  the rank that owns keyframe k+3 (j % ranks): collect the look-ahead job of keyframe k+3 (its cloud converted and exported into flat buffer (k+3) % 4 by the library's
//                          helper thread: pwn_closer.cpp:92-93 _cache->get(current) one keyframe ahead), start the job of keyframe k+4
//                  ncclBroadcast of keyframe k+2's flat form -- only the bytes written; their count travelled in the control row of step k-1's all-gather
//                  ncclAllGather of step k's records (ordered behind the call's stream with pwn_hip_ctx_signal_stream) + one control row per rank
//                  pwn_hip_cloud_import of keyframe k+1's flat form into replica (k+1) % 2 (broadcast during step k-1)
//                the last all-gather is read back at the end; rank 0 applies PwnCloser's thresholds (pwn_closer.cpp:138-141) and prints one line per keyframe
//   `serial`     (5th argument) round 5's chain instead: convert, export, broadcast of the buffer's bound, import, match, all-gather -- one after the other
//
// The ranks are children of this process, forked before anything touches a GPU (the parent never does); rank 0 creates the ncclUniqueId and hands it
// to the others through pipes.  A rank that fails -- an unreadable frame, no device for it, a failed collective -- exits with a code; the parent reaps
// whichever child ends first and terminates the others (they would wait for the missing rank in ncclCommInitRank or a collective for ever), and gives
// up after a wall-clock limit (PWN_PARTITION_TIMEOUT_S, default 600).
//
//   pwn_hip_partition_app frames.txt [ranks=1] [steps=3] [guesses.txt|-] [serial]
//     frames.txt   16-bit PGM files, one per line: the first is `current` (every step's keyframe: the benchmark re-uses it), the others are the keyframes of the other partition
//     guesses.txt  optional: one line of 16 floats (column-major isometry) per keyframe = iT * other.T; identity when absent or "-"
//
// build (g2o_frontend_amd/build.py: build_tools):
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I. -I/opt/rocm/include tools/pwn_hip_partition_app.cpp -o tools/pwn_hip_partition_app
//       -Lg2o_frontend_amd -lpwn_hip -L/opt/rocm/lib -lrccl -lamdhip64 -Wl,-rpath,$ORIGIN/../g2o_frontend_amd -Wl,-rpath,/opt/rocm/lib
//   (the HIP runtime is on the link line for the program's own streams and events; device memory still comes from pwn_hip_device_alloc)
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "g2o_frontend_amd/host/pwn_hip.hpp"

using namespace pwn_hip;

static bool readPGM16(const std::string& fn, RawDepthImage& img) {
  std::ifstream f(fn, std::ios::binary);
  if (!f) return false;
  std::string magic; int w = 0, h = 0, maxv = 0;
  f >> magic >> w >> h >> maxv;
  if (magic != "P5" || w <= 0 || h <= 0 || maxv != 65535) return false;
  f.get();
  std::vector<unsigned char> buf((size_t)w * h * 2);
  f.read(reinterpret_cast<char*>(buf.data()), buf.size());
  if (!f) return false;
  img.rows = h; img.cols = w; img.data.resize((size_t)w * h);
  for (size_t i = 0; i < img.data.size(); ++i) img.data[i] = (uint16_t)((buf[2 * i] << 8) | buf[2 * i + 1]);
  return true;
}
// contiguous shard of `rank` (g2o_frontend_amd/shard.py: shard_range)
static void shardRange(int n, int rank, int world, int& lo, int& hi) { lo = (rank * n + world - 1) / world; hi = ((rank + 1) * n + world - 1) / world; }

#define NCCLCHK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { std::fprintf(stderr, "rank %d: %s: %s\n", rank, #call, ncclGetErrorString(r_)); return 3; } } while (0)
#define HIPCHK_(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "rank %d: %s: %s\n", rank, #call, hipGetErrorString(e_)); return 3; } } while (0)

// ---- the pipelined step's state: what the callback inside the match call works with ---------------------------------------------------------------
struct Pipeline {
  static constexpr int RING = 4;
  int rank = 0, world = 1, rows = 0, cols = 0, nmax = 0, k = 0, rc = 0;
  Context* ctx = nullptr; Context* io = nullptr;             // the matches' context; the small one the imports run on
  DepthImageConverterIntegralImage* converter = nullptr;
  const uint16_t* frame = nullptr;                           // rank 0: the raw `current` frame (device memory)
  Cloud* conv = nullptr;                                     // rank 0: what the look-ahead job converts into
  Cloud* rep[2] = { nullptr, nullptr };                      // the replicas the matches alternate between
  void* flat[RING] = { nullptr, nullptr, nullptr, nullptr }; size_t bound = 0, size[RING] = { 0, 0, 0, 0 };
  float* rec[2] = { nullptr, nullptr };                      // [nmax + 1][72]: the records + this rank's control row
  float* all[2] = { nullptr, nullptr };                      // [world][nmax + 1][72]
  float* ctrlSend[2] = { nullptr, nullptr }; float* ctrlHost[2] = { nullptr, nullptr };      // page-locked, 4 floats each
  ncclComm_t comm = nullptr;
  hipStream_t sN = nullptr, sImp = nullptr, sRec = nullptr;  // the collectives' stream; the stream an import waits on; the one a records buffer's reuse waits on
  hipEvent_t evB[RING] = { nullptr, nullptr, nullptr, nullptr }, evG[2] = { nullptr, nullptr };
  bool jobPending = false; double jobMs = 0.0, cbMs = 0.0;
  size_t recFloats() const { return (size_t)(nmax + 1) * PWN_HIP_MATCH_RECORD_FLOATS; }
};
static int overlapBody(Pipeline& P) {
  const int rank = P.rank, k = P.k, R = Pipeline::RING, W = P.world;
  // the look-ahead work rotates over the ranks: keyframe j is converted, exported and broadcast by rank j % world, so that no rank's step is longer than the
  // others' by a conversion (a deployment feeds keyframe j's raw frame to that rank)
  if (rank == (k + 3) % W) {
    float ms = 0.f;
    const size_t w = P.converter->computeExportEnd(*P.conv, &ms);                    // keyframe k+3's flat form is in its owner's buffer (k+3) % 4
    P.jobMs += ms; P.jobPending = false;
    P.ctrlSend[k % 2][0] = (float)(w / 256);                                         // its size rides in the owner's control row of this step's all-gather
    HIPCHK_(hipMemcpyAsync(P.rec[k % 2] + (size_t)P.nmax * PWN_HIP_MATCH_RECORD_FLOATS, P.ctrlSend[k % 2], 4 * sizeof(float), hipMemcpyHostToDevice, P.sN));
  }
  if (rank == (k + 4) % W) {
    P.converter->computeExportBegin(*P.conv, P.frame, 0.001f, P.rows, P.cols, P.flat[(k + 4) % R], P.bound);      // buffer k % 4: imported during step k-1
    P.jobPending = true;
  }
  if (k >= 1) {                                                                      // size of keyframe k+2: control row of step k-1's all-gather
    HIPCHK_(hipEventSynchronize(P.evG[(k - 1) % 2]));
    P.size[(k + 2) % R] = (size_t)P.ctrlHost[(k - 1) % 2][0] * 256;
  }
  const int j2 = (k + 2) % R, j1 = (k + 1) % R;
  NCCLCHK(ncclBroadcast(P.flat[j2], P.flat[j2], P.size[j2], ncclUint8, (k + 2) % W, P.comm, P.sN));      // keyframe k+2 travels from its owner while keyframe k is matched
  HIPCHK_(hipEventRecord(P.evB[j2], P.sN));
  P.ctx->signalStream(P.sN);                                                         // the all-gather runs after this call's records are packed
  NCCLCHK(ncclAllGather(P.rec[k % 2], P.all[k % 2], P.recFloats(), ncclFloat, P.comm, P.sN));
  HIPCHK_(hipMemcpyAsync(P.ctrlHost[k % 2], P.all[k % 2] + ((size_t)((k + 3) % W) * (size_t)(P.nmax + 1) + (size_t)P.nmax) * PWN_HIP_MATCH_RECORD_FLOATS, 4 * sizeof(float),
                         hipMemcpyDeviceToHost, P.sN));                                // the control row of keyframe k+3's owner
  HIPCHK_(hipEventRecord(P.evG[k % 2], P.sN));
  HIPCHK_(hipStreamWaitEvent(P.sImp, P.evB[j1], 0));                                 // keyframe k+1's broadcast (queued during step k-1)
  P.io->waitStream(P.sImp);
  P.rep[(k + 1) % 2]->importFlat(P.flat[j1], P.size[j1]);                            // on the small context: waits for that broadcast and its own copies only
  return 0;
}
static void overlap(void* user) {
  Pipeline& P = *static_cast<Pipeline*>(user);
  const auto t0 = std::chrono::steady_clock::now();
  try { if (int rc = overlapBody(P)) P.rc = rc; }
  catch (const Error& e) { std::fprintf(stderr, "rank %d (inside the match call): %s\n", P.rank, e.what()); P.rc = 2; }
  P.cbMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

static int runRank(int rank, int world, const std::vector<std::string>& files, const std::vector<Isometry3f>& guesses, int steps, bool serial, int idIn,
                   const std::vector<int>& idOut) {
  const int K = (int)files.size() - 1;                       // keyframes of the other partition
  int lo, hi; shardRange(K, rank, world, lo, hi);
  const int n = hi - lo, nmax = (K + world - 1) / world;
  // every input is read and checked before the rank joins the communicator: a rank that gives up later leaves the others waiting for it
  std::vector<RawDepthImage> others((size_t)n); RawDepthImage currentFrame;
  if (!readPGM16(files[0], currentFrame)) { std::fprintf(stderr, "rank %d: cannot read %s\n", rank, files[0].c_str()); return 1; }
  for (int i = 0; i < n; ++i) {
    const std::string& fn = files[(size_t)(1 + lo + i)];
    if (!readPGM16(fn, others[(size_t)i])) { std::fprintf(stderr, "rank %d: cannot read %s\n", rank, fn.c_str()); return 1; }
    if (others[(size_t)i].rows != currentFrame.rows || others[(size_t)i].cols != currentFrame.cols) { std::fprintf(stderr, "rank %d: %s: frame size differs\n", rank, fn.c_str()); return 1; }
  }
  const int rows = currentFrame.rows, cols = currentFrame.cols;
  if (pwn_hip_device_count() <= rank) { std::fprintf(stderr, "rank %d: one GPU per rank (devices: %d)\n", rank, pwn_hip_device_count()); return 1; }
  ncclComm_t comm = nullptr;
  int result = 0;
  try {
    Context ctx(rank, rows, cols, std::max(2, std::min(256, std::max(n, 1))));      // sym6 clouds: the library's default
    Context io(rank, rows, cols, 1);
    // the communicator: rank 0 makes the id, the others read it from their pipe
    ncclUniqueId id;
    if (rank == 0) { NCCLCHK(ncclGetUniqueId(&id)); for (int fd : idOut) if (write(fd, &id, sizeof(id)) != (ssize_t)sizeof(id)) return 3; }
    else if (read(idIn, &id, sizeof(id)) != (ssize_t)sizeof(id)) return 3;
    NCCLCHK(ncclCommInitRank(&comm, world, id, rank));
    // pwn_core/conf/pwn_aligner_1_1.conf as pwn_simple_aligner.cpp:214-269 applies it (bench.py's VGA tables)
    PinholePointProjector projector, alignerProjector;
    Matrix3f Kc; Kc(0,0) = 525.f; Kc(1,1) = 525.f; Kc(0,2) = 319.5f; Kc(1,2) = 239.5f;
    projector.setCameraMatrix(Kc); projector.setImageSize(rows, cols); projector.setMinDistance(0.5f); projector.setMaxDistance(4.5f);
    alignerProjector.setMinDistance(0.5f); alignerProjector.setMaxDistance(4.5f);
    StatsCalculatorIntegralImage stats; stats.setWorldRadius(0.1f); stats.setMinImageRadius(10); stats.setMaxImageRadius(30); stats.setMinPoints(50); stats.setCurvatureThreshold(0.2f);
    PointInformationMatrixCalculator pinfo; NormalInformationMatrixCalculator ninfo;
    DepthImageConverterIntegralImage converter(&ctx, &projector, &stats, &pinfo, &ninfo);
    CorrespondenceFinder finder; finder.setImageSize(rows, cols); finder.setInlierDistanceThreshold(1.0f); finder.setInlierNormalAngularThreshold(0.95f);
    finder.setFlatCurvatureThreshold(0.02f); finder.setInlierCurvatureRatioThreshold(1.3f);
    Linearizer linearizer; linearizer.setInlierMaxChi2(9e3f); linearizer.setRobustKernel(true);
    Aligner aligner(&ctx); aligner.setProjector(&alignerProjector); aligner.setCorrespondenceFinder(&finder); aligner.setLinearizer(&linearizer);
    aligner.setOuterIterations(10); aligner.setInnerIterations(1);
    PwnMatcherBase matcher(&ctx, &aligner, &converter); matcher.setScale(1);
    const Isometry3f I;

    // this rank's cache: its shard of the keyframes, converted once
    std::vector<Cloud*> cache((size_t)n); std::vector<const uint16_t*> raw((size_t)n);
    for (int i = 0; i < n; ++i) { cache[(size_t)i] = new Cloud(ctx, rows * cols); raw[(size_t)i] = others[(size_t)i].data.data(); }
    if (n) converter.computeBatchRaw(cache, raw, 0.001f, rows, cols);
    const size_t bound = pwn_hip_cloud_export_bound(rows * cols, PWN_HIP_OMEGA_SYM6, rows * cols, 0);
    std::vector<Isometry3f> g((size_t)n); std::vector<int> ids((size_t)n);
    for (int i = 0; i < n; ++i) { g[(size_t)i] = guesses.empty() ? I : guesses[(size_t)(lo + i)]; ids[(size_t)i] = lo + i; }
    const size_t rowFloats = PWN_HIP_MATCH_RECORD_FLOATS, recFloats = (size_t)(nmax + 1) * rowFloats;      // + the control row
    std::vector<float> pad(recFloats, -1.f);                   // rows past this rank's shard and the control row: pair id -1 = padding (shard.py)
    std::vector<float> host(recFloats * (size_t)world);
    double msPerStep = 0.0; size_t flatBytes = 0, sentBytes = 0;
    uint16_t* frameDev = nullptr;
    if (rank == 0 || !serial) {                                // pipelined: every rank owns keyframes (j % world)
      ctx.check(pwn_hip_device_alloc(ctx.handle(), (void**)&frameDev, currentFrame.data.size() * sizeof(uint16_t)));
      ctx.check(pwn_hip_copy(ctx.handle(), frameDev, currentFrame.data.data(), currentFrame.data.size() * sizeof(uint16_t)));
    }
    Pipeline P;
    if (serial) {
      // ---- round 5's chain, on the legacy default stream
      Cloud current(ctx, rows * cols);
      void* flat = nullptr; float* rec = nullptr; float* all = nullptr;
      ctx.check(pwn_hip_device_alloc(ctx.handle(), &flat, bound));
      ctx.check(pwn_hip_device_alloc(ctx.handle(), (void**)&rec, recFloats * sizeof(float)));
      ctx.check(pwn_hip_device_alloc(ctx.handle(), (void**)&all, recFloats * sizeof(float) * (size_t)world));
      ctx.check(pwn_hip_copy(ctx.handle(), rec, pad.data(), recFloats * sizeof(float)));
      std::vector<Cloud*> from((size_t)n, &current);
      for (int s = 0; s < steps + 1; ++s) {                    // step 0 = warm-up
        const auto t0 = std::chrono::steady_clock::now();
        if (rank == 0) {
          std::vector<Cloud*> c1(1, &current); std::vector<const uint16_t*> f1(1, frameDev);
          converter.computeBatchRaw(c1, f1, 0.001f, rows, cols);
          flatBytes = current.exportFlat(flat, bound);
        }
        NCCLCHK(ncclBroadcast(flat, flat, bound, ncclUint8, 0, comm, (hipStream_t)0));
        ctx.waitStream(nullptr);                               // what the context queues next runs after the broadcast (and after the previous all-gather)
        if (rank != 0) current.importFlat(flat, bound);
        if (n) matcher.matchCloudsBatchRecords(rec, from, cache, I, I, Kc, rows, cols, g, ids);
        NCCLCHK(ncclAllGather(rec, all, recFloats, ncclFloat, comm, (hipStream_t)0));
        ctx.waitStream(nullptr);
        ctx.check(pwn_hip_copy(ctx.handle(), host.data(), all, host.size() * sizeof(float)));      // ordered after the all-gather, complete on return
        if (s > 0) msPerStep += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      }
      sentBytes = bound;
      ctx.check(pwn_hip_device_free(ctx.handle(), flat)); ctx.check(pwn_hip_device_free(ctx.handle(), rec)); ctx.check(pwn_hip_device_free(ctx.handle(), all));
    } else {
      // ---- pipelined: see the head of this file
      P.rank = rank; P.world = world; P.rows = rows; P.cols = cols; P.nmax = nmax; P.ctx = &ctx; P.io = &io; P.converter = &converter; P.comm = comm;
      P.frame = frameDev; P.bound = bound;
      Cloud rep0(io, rows * cols), rep1(io, rows * cols), conv(ctx, rows * cols);
      P.rep[0] = &rep0; P.rep[1] = &rep1; P.conv = &conv;
      HIPCHK_(hipSetDevice(rank));
      HIPCHK_(hipStreamCreateWithFlags(&P.sN, hipStreamNonBlocking)); HIPCHK_(hipStreamCreateWithFlags(&P.sImp, hipStreamNonBlocking)); HIPCHK_(hipStreamCreateWithFlags(&P.sRec, hipStreamNonBlocking));
      for (auto& e : P.evB) HIPCHK_(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      for (auto& e : P.evG) HIPCHK_(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      for (auto& f : P.flat) ctx.check(pwn_hip_device_alloc(ctx.handle(), &f, bound));
      for (int j = 0; j < 2; ++j) {
        ctx.check(pwn_hip_device_alloc(ctx.handle(), (void**)&P.rec[j], recFloats * sizeof(float)));
        ctx.check(pwn_hip_device_alloc(ctx.handle(), (void**)&P.all[j], recFloats * sizeof(float) * (size_t)world));
        ctx.check(pwn_hip_copy(ctx.handle(), P.rec[j], pad.data(), recFloats * sizeof(float)));
        ctx.check(pwn_hip_host_alloc((void**)&P.ctrlSend[j], 4 * sizeof(float))); ctx.check(pwn_hip_host_alloc((void**)&P.ctrlHost[j], 4 * sizeof(float)));
        std::memset(P.ctrlSend[j], 0, 4 * sizeof(float)); std::memset(P.ctrlHost[j], 0, 4 * sizeof(float));
      }
      // fill the pipeline: keyframes 0-2 converted and exported, their sizes known everywhere; keyframe 0 in replica 0, keyframe 1 on its way, job 3 running
      unsigned long long* sizeDev = nullptr;
      ctx.check(pwn_hip_device_alloc(ctx.handle(), (void**)&sizeDev, sizeof(unsigned long long)));
      for (int j = 0; j < 3; ++j) {
        unsigned long long w = 0;
        if (rank == j % world) { converter.computeExportBegin(conv, frameDev, 0.001f, rows, cols, P.flat[j], bound); w = converter.computeExportEnd(conv); }
        ctx.check(pwn_hip_copy(ctx.handle(), sizeDev, &w, sizeof(w)));
        NCCLCHK(ncclBroadcast(sizeDev, sizeDev, 1, ncclUint64, j % world, comm, P.sN));
        HIPCHK_(hipStreamSynchronize(P.sN));
        ctx.check(pwn_hip_copy(ctx.handle(), &w, sizeDev, sizeof(w)));
        P.size[j] = (size_t)w;
      }
      NCCLCHK(ncclBroadcast(P.flat[0], P.flat[0], P.size[0], ncclUint8, 0, comm, P.sN));
      io.waitStream(P.sN); rep0.importFlat(P.flat[0], P.size[0]);
      NCCLCHK(ncclBroadcast(P.flat[1], P.flat[1], P.size[1], ncclUint8, 1 % world, comm, P.sN));
      HIPCHK_(hipEventRecord(P.evB[1], P.sN));
      if (rank == 3 % world) { converter.computeExportBegin(conv, frameDev, 0.001f, rows, cols, P.flat[3], bound); P.jobPending = true; }
      std::vector<Cloud*> from[2] = { std::vector<Cloud*>((size_t)n, &rep0), std::vector<Cloud*>((size_t)n, &rep1) };
      ctx.setEnqueuedCallback(overlap, &P);
      for (int s = 0; s < steps + 1 && !P.rc; ++s) {           // step 0 = warm-up
        const auto t0 = std::chrono::steady_clock::now();
        HIPCHK_(hipStreamWaitEvent(P.sRec, P.evG[P.k % 2], 0));  // this records buffer is free again: step k-2's all-gather read it (not step k-1's,
        ctx.waitStream(P.sRec);                                //   which may still be in flight)
        if (n) matcher.matchCloudsBatchRecords(P.rec[P.k % 2], from[P.k % 2], cache, I, I, Kc, rows, cols, g, ids);
        else overlap(&P);                                      // a rank without keyframes still takes part in the collectives
        ++P.k;
        if (s > 0) msPerStep += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      }
      ctx.setEnqueuedCallback(nullptr, nullptr);
      if (P.jobPending) (void)converter.computeExportEnd(conv);
      HIPCHK_(hipStreamSynchronize(P.sN));
      if (P.rc) { result = P.rc; }
      else {
        const int last = (P.k - 1) % 2;
        ctx.check(pwn_hip_copy(ctx.handle(), host.data(), P.all[last], host.size() * sizeof(float)));
        flatBytes = sentBytes = P.size[(P.k) % Pipeline::RING];
      }
      for (auto f : P.flat) ctx.check(pwn_hip_device_free(ctx.handle(), f));
      for (int j = 0; j < 2; ++j) {
        ctx.check(pwn_hip_device_free(ctx.handle(), P.rec[j])); ctx.check(pwn_hip_device_free(ctx.handle(), P.all[j]));
        pwn_hip_host_free(P.ctrlSend[j]); pwn_hip_host_free(P.ctrlHost[j]);
      }
      ctx.check(pwn_hip_device_free(ctx.handle(), sizeDev));
      for (auto e : P.evB) (void)hipEventDestroy(e);
      for (auto e : P.evG) (void)hipEventDestroy(e);
      (void)hipStreamDestroy(P.sN); (void)hipStreamDestroy(P.sImp); (void)hipStreamDestroy(P.sRec);
    }
    if (rank == 0 && !result) {
      PwnCloserAcceptance acceptance;
      std::printf("keyframes %d ranks %d steps %d ms_per_step %.4f flat_cloud_bytes %zu broadcast_bytes %zu mode %s lookahead_job_ms %.3f inside_call_ms %.3f\n", K, world, steps,
                  steps ? msPerStep / steps : 0.0, flatBytes, sentBytes, serial ? "serial" : "pipelined", P.k ? P.jobMs / P.k : 0.0, P.k ? P.cbMs / P.k : 0.0);
      for (int k = 0; k < K; ++k) {                            // record of keyframe k: the row whose pair id is k
        const float* q = nullptr;
        for (size_t r = 0; r < (size_t)world * (size_t)(nmax + 1) && !q; ++r) if ((int)host[r * rowFloats + 19] == k) q = &host[r * rowFloats];
        if (!q) { std::fprintf(stderr, "gather incomplete: no record of keyframe %d\n", k); result = 4; break; }
        if (q[63] != 0.f) { std::fprintf(stderr, "keyframe %d: record of a call that was being repeated (word 63)\n", k); result = 4; break; }
        PwnMatcherBase::MatcherResult m; m.image_nonZeros = (int)q[64]; m.image_outliers = (int)q[65]; m.image_inliers = (int)q[66];
        std::printf("keyframe %d %d %d %d %d %d %.9g", k, acceptance.accept(m) ? 1 : 0, (int)q[17], (int)q[64], (int)q[65], (int)q[66], q[67]);
        for (int t = 0; t < 16; ++t) std::printf(" %.9g", q[t]);
        for (int t = 20; t < 30; ++t) std::printf(" %.9g", q[t]);
        std::printf("\n");
      }
    }
    if (frameDev) ctx.check(pwn_hip_device_free(ctx.handle(), frameDev));
    for (Cloud* c : cache) delete c;
    if (result) ncclCommAbort(comm); else ncclCommDestroy(comm);
  } catch (const Error& e) {
    std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
    if (comm) ncclCommAbort(comm);                             // do not leave the other ranks inside a collective with this one
    return 2;
  }
  return result;
}

int main(int argc, char** argv) {
  if (argc < 2) { std::cout << "USAGE: pwn_hip_partition_app frames.txt [ranks=1] [steps=3] [guesses.txt|-] [serial]" << std::endl; return 0; }
  const int world = argc > 2 ? atoi(argv[2]) : 1, steps = argc > 3 ? atoi(argv[3]) : 3;
  const bool serial = argc > 5 && std::string(argv[5]) == "serial";
  std::vector<std::string> files;
  { std::ifstream is(argv[1]); std::string line; while (std::getline(is, line)) { std::istringstream iss(line); std::string fn; if ((iss >> fn) && fn[0] != '#') files.push_back(fn); } }
  if (files.size() < 2 || world < 1 || (int)files.size() - 1 < world) { std::cerr << "need `current` + at least one keyframe per rank" << std::endl; return 1; }
  std::vector<Isometry3f> guesses;
  if (argc > 4 && std::string(argv[4]) != "-") {
    std::ifstream is(argv[4]); std::string line;
    while (std::getline(is, line)) { std::istringstream iss(line); float v[16]; int k = 0; while (k < 16 && (iss >> v[k])) ++k; if (k == 16) guesses.push_back(Isometry3f(v)); }
    if (guesses.size() != files.size() - 1) { std::cerr << "guesses.txt: one line of 16 floats per keyframe" << std::endl; return 1; }
  }
  if (world == 1) return runRank(0, 1, files, guesses, steps, serial, -1, std::vector<int>());
  // pipes for the ncclUniqueId, then the ranks -- all before anything touches a GPU
  std::vector<int> rd((size_t)world, -1), wr((size_t)world, -1);
  for (int r = 1; r < world; ++r) { int fd[2]; if (pipe(fd) != 0) return 1; rd[(size_t)r] = fd[0]; wr[(size_t)r] = fd[1]; }
  std::vector<pid_t> pids;
  for (int r = 0; r < world; ++r) {
    const pid_t p = fork();
    if (p < 0) { for (pid_t q : pids) kill(q, SIGTERM); return 1; }
    if (p == 0) {
      std::vector<int> out; if (r == 0) for (int q = 1; q < world; ++q) out.push_back(wr[(size_t)q]);
      _exit(runRank(r, world, files, guesses, steps, serial, rd[(size_t)r], out));
    }
    pids.push_back(p);
  }
  for (int r = 1; r < world; ++r) { close(rd[(size_t)r]); close(wr[(size_t)r]); }      // the parent holds no end of the pipes: a dead writer reads as end-of-file
  // reap whichever rank ends first; after the first failure (or the wall-clock limit) the others are terminated -- they would wait for it for ever
  const char* lim = getenv("PWN_PARTITION_TIMEOUT_S");
  const double limit = lim ? atof(lim) : 600.0;
  const auto t0 = std::chrono::steady_clock::now();
  int rc = 0; size_t alive = pids.size(); bool killed = false;
  while (alive > 0) {
    int st = 0;
    const pid_t p = waitpid(-1, &st, WNOHANG);
    if (p > 0) {
      --alive;
      const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 5;
      if (code != 0 && rc == 0) rc = code;
    } else if (p < 0) break;
    else usleep(20000);
    const bool late = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit;
    if ((rc != 0 || late) && !killed) {
      if (late && rc == 0) { rc = 6; std::fprintf(stderr, "wall-clock limit of %.0f s reached: terminating the ranks\n", limit); }
      for (pid_t q : pids) kill(q, SIGTERM);                   // ended ranks: the signal goes nowhere
      killed = true;
    }
  }
  return rc;
}
