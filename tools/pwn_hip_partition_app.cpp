// pwn_hip_partition_app -- PwnCloser::processPartition (pwn_tracker/pwn_closer.cpp:85-111) over the GPUs of one node in native code: one process per
// GPU, the C-ABI of include/pwn_hip.h through the C++ mirror, RCCL for the two collectives.  What INTEGRATION.md section 1 sketches, compiled and run:
//
//   every rank   converts and keeps its contiguous shard of the partition's keyframes (the PwnCache of its GPU: pwn_tracker_cache.cpp:24-51)
//   per step     rank 0: makeCloud of `current` (pwn_closer.cpp:92-93), pwn_hip_cloud_export into one flat device buffer
//                ncclBroadcast of the buffer (the only collective that moves real data: ~17 MB per VGA cloud over xGMI)
//                ranks != 0: pwn_hip_cloud_import  (pwn_hip_ctx_wait_stream orders it after the broadcast)
//                every rank: matchFrames' data path for its shard -- matchClouds(current, other, iT * other.T) = Aligner::align from the odometry guess
//                with the z translation zeroed + the depth-agreement score (pwn_matcher_base.cpp:88-183) -- as ONE pwn_hip_match_batch_records call,
//                288-byte records written on the device
//                ncclAllGather of the records; rank 0 applies PwnCloser's thresholds (pwn_closer.cpp:138-141) and prints one line per keyframe
//
// The ranks are children of this process, forked before anything touches a GPU (the parent never does); rank 0 creates the ncclUniqueId and hands it
// to the others through pipes.  No HIP headers are needed by the program itself: device memory comes from pwn_hip_device_alloc, the collectives run on
// the legacy default stream (stream 0) and pwn_hip_ctx_wait_stream(ctx, NULL) / pwn_hip_copy order the library's work against them.
//
//   pwn_hip_partition_app frames.txt [ranks=1] [steps=3] [guesses.txt]
//     frames.txt   16-bit PGM files, one per line: the first is `current`, the others are the keyframes of the other partition
//     guesses.txt  optional: one line of 16 floats (column-major isometry) per keyframe = iT * other.T; identity when absent
//
// build (g2o_frontend_amd/build.py: build_tools):
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I. -I/opt/rocm/include tools/pwn_hip_partition_app.cpp -o tools/pwn_hip_partition_app
//       -Lg2o_frontend_amd -lpwn_hip -L/opt/rocm/lib -lrccl -Wl,-rpath,$ORIGIN/../g2o_frontend_amd -Wl,-rpath,/opt/rocm/lib
#include <rccl/rccl.h>
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

static int runRank(int rank, int world, const std::vector<std::string>& files, const std::vector<Isometry3f>& guesses, int steps, int idIn, const std::vector<int>& idOut) {
  const int K = (int)files.size() - 1;                       // keyframes of the other partition
  int lo, hi; shardRange(K, rank, world, lo, hi);
  const int n = hi - lo, nmax = (K + world - 1) / world;
  std::vector<RawDepthImage> others((size_t)n); RawDepthImage currentFrame;
  if (!readPGM16(files[0], currentFrame)) { std::fprintf(stderr, "cannot read %s\n", files[0].c_str()); return 1; }
  for (int i = 0; i < n; ++i) if (!readPGM16(files[(size_t)(1 + lo + i)], others[(size_t)i])) { std::fprintf(stderr, "cannot read %s\n", files[(size_t)(1 + lo + i)].c_str()); return 1; }
  const int rows = currentFrame.rows, cols = currentFrame.cols;
  if (pwn_hip_device_count() <= rank) { std::fprintf(stderr, "rank %d: one GPU per rank (devices: %d)\n", rank, pwn_hip_device_count()); return 1; }
  try {
    Context ctx(rank, rows, cols, std::max(2, std::min(256, std::max(n, 1))));
    ctx.setOmegaStorage(PWN_HIP_OMEGA_SYM6);
    // the communicator: rank 0 makes the id, the others read it from their pipe
    ncclUniqueId id;
    if (rank == 0) { NCCLCHK(ncclGetUniqueId(&id)); for (int fd : idOut) if (write(fd, &id, sizeof(id)) != (ssize_t)sizeof(id)) return 3; }
    else if (read(idIn, &id, sizeof(id)) != (ssize_t)sizeof(id)) return 3;
    ncclComm_t comm;
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
    Cloud current(ctx, rows * cols);
    const size_t bound = pwn_hip_cloud_export_bound(rows * cols, PWN_HIP_OMEGA_SYM6, rows * cols, 0);
    void* flat = nullptr; float* rec = nullptr; float* all = nullptr;
    ctx.check(pwn_hip_device_alloc(ctx.handle(), &flat, bound));
    const size_t recFloats = (size_t)nmax * PWN_HIP_MATCH_RECORD_FLOATS;
    ctx.check(pwn_hip_device_alloc(ctx.handle(), (void**)&rec, recFloats * sizeof(float)));
    ctx.check(pwn_hip_device_alloc(ctx.handle(), (void**)&all, recFloats * sizeof(float) * (size_t)world));
    std::vector<float> pad(recFloats, -1.f);                   // rows past this rank's shard: pair id -1 = padding (shard.py: gather_records)
    ctx.check(pwn_hip_copy(ctx.handle(), rec, pad.data(), recFloats * sizeof(float)));
    std::vector<Cloud*> from((size_t)n, &current);
    std::vector<Isometry3f> g((size_t)n); std::vector<int> ids((size_t)n);
    for (int i = 0; i < n; ++i) { g[(size_t)i] = guesses.empty() ? I : guesses[(size_t)(lo + i)]; ids[(size_t)i] = lo + i; }
    std::vector<float> host(recFloats * (size_t)world);

    double msPerStep = 0.0;
    for (int s = 0; s < steps + 1; ++s) {                      // step 0 = warm-up
      const auto t0 = std::chrono::steady_clock::now();
      if (rank == 0) {
        std::vector<Cloud*> c1(1, &current); std::vector<const uint16_t*> f1(1, currentFrame.data.data());
        converter.computeBatchRaw(c1, f1, 0.001f, rows, cols);
        current.exportFlat(flat, bound);
      }
      NCCLCHK(ncclBroadcast(flat, flat, bound, ncclUint8, 0, comm, (hipStream_t)0));
      ctx.waitStream(nullptr);                                 // what the context queues next runs after the broadcast (and after the previous all-gather)
      if (rank != 0) current.importFlat(flat, bound);
      if (n) matcher.matchCloudsBatchRecords(rec, from, cache, I, I, Kc, rows, cols, g, ids);
      NCCLCHK(ncclAllGather(rec, all, recFloats, ncclFloat, comm, (hipStream_t)0));
      ctx.waitStream(nullptr);
      ctx.check(pwn_hip_copy(ctx.handle(), host.data(), all, host.size() * sizeof(float)));      // ordered after the all-gather, complete on return
      if (s > 0) msPerStep += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (rank == 0) {
      PwnCloserAcceptance acceptance;
      std::printf("keyframes %d ranks %d steps %d ms_per_step %.4f flat_cloud_bytes %zu\n", K, world, steps, steps ? msPerStep / steps : 0.0, current.flatSize());
      for (int k = 0; k < K; ++k) {                            // record of keyframe k: rank (k * world) / K, row k - lo(rank)
        const float* q = nullptr;
        for (size_t r = 0; r < (size_t)world * (size_t)nmax && !q; ++r) if ((int)host[r * PWN_HIP_MATCH_RECORD_FLOATS + 19] == k) q = &host[r * PWN_HIP_MATCH_RECORD_FLOATS];
        if (!q) { std::fprintf(stderr, "gather incomplete: no record of keyframe %d\n", k); return 4; }
        PwnMatcherBase::MatcherResult m; m.image_nonZeros = (int)q[64]; m.image_outliers = (int)q[65]; m.image_inliers = (int)q[66];
        std::printf("keyframe %d %d %d %d %d %d %.9g", k, acceptance.accept(m) ? 1 : 0, (int)q[17], (int)q[64], (int)q[65], (int)q[66], q[67]);
        for (int t = 0; t < 16; ++t) std::printf(" %.9g", q[t]);
        for (int t = 20; t < 30; ++t) std::printf(" %.9g", q[t]);
        std::printf("\n");
      }
    }
    ncclCommDestroy(comm);
    for (Cloud* c : cache) delete c;
    ctx.check(pwn_hip_device_free(ctx.handle(), flat)); ctx.check(pwn_hip_device_free(ctx.handle(), rec)); ctx.check(pwn_hip_device_free(ctx.handle(), all));
  } catch (const Error& e) {
    std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
    return 2;
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) { std::cout << "USAGE: pwn_hip_partition_app frames.txt [ranks=1] [steps=3] [guesses.txt]" << std::endl; return 0; }
  const int world = argc > 2 ? atoi(argv[2]) : 1, steps = argc > 3 ? atoi(argv[3]) : 3;
  std::vector<std::string> files;
  { std::ifstream is(argv[1]); std::string line; while (std::getline(is, line)) { std::istringstream iss(line); std::string fn; if ((iss >> fn) && fn[0] != '#') files.push_back(fn); } }
  if (files.size() < 2 || world < 1 || (int)files.size() - 1 < world) { std::cerr << "need `current` + at least one keyframe per rank" << std::endl; return 1; }
  std::vector<Isometry3f> guesses;
  if (argc > 4) {
    std::ifstream is(argv[4]); std::string line;
    while (std::getline(is, line)) { std::istringstream iss(line); float v[16]; int k = 0; while (k < 16 && (iss >> v[k])) ++k; if (k == 16) guesses.push_back(Isometry3f(v)); }
    if (guesses.size() != files.size() - 1) { std::cerr << "guesses.txt: one line of 16 floats per keyframe" << std::endl; return 1; }
  }
  // pipes for the ncclUniqueId, then the ranks -- all before anything touches a GPU
  std::vector<int> rd((size_t)world, -1), wr((size_t)world, -1);
  for (int r = 1; r < world; ++r) { int fd[2]; if (pipe(fd) != 0) return 1; rd[(size_t)r] = fd[0]; wr[(size_t)r] = fd[1]; }
  if (world == 1) return runRank(0, 1, files, guesses, steps, -1, std::vector<int>());
  std::vector<pid_t> pids;
  for (int r = 0; r < world; ++r) {
    const pid_t p = fork();
    if (p < 0) return 1;
    if (p == 0) {
      std::vector<int> out; if (r == 0) for (int q = 1; q < world; ++q) out.push_back(wr[(size_t)q]);
      _exit(runRank(r, world, files, guesses, steps, rd[(size_t)r], out));
    }
    pids.push_back(p);
  }
  int rc = 0;
  for (pid_t p : pids) { int st = 0; waitpid(p, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = WIFEXITED(st) ? WEXITSTATUS(st) : 5; }
  return rc;
}
