// pwn_hip_tracker_app -- the reference's tracker + closer flow (pwn_tracker/pwn_tracker_app.cpp, pwn_tracker.cpp:106-215,
// pwn_closer.cpp:92-151) on the MI355X path, over the C++ host mirror:
//   1. PwnTracker::processFrame on every "timestamp filename" line (16-bit PGM depth images), key-cloud switching included;
//   2. the loop-closure pass: every pair of keyframes through a CloudCache (PwnCache: a miss re-converts the stored depth image) and one
//      batched matchClouds call, PwnCloser's acceptance rule on the scores;
//   3. Aligner extras on the first two keyframes: _computeStatistics (omega, eigen ratios) and an alignment with SE(3) priors.
// Output: <prefix>_track.txt, <prefix>_closures.txt, <prefix>_extras.txt (plain numbers, %.9g).
//
// build (g2o_frontend_amd/build.py: build_tools):
//   g++ -O2 -std=c++17 -I. tools/pwn_hip_tracker_app.cpp -o tools/pwn_hip_tracker_app -Lg2o_frontend_amd -lpwn_hip -Wl,-rpath,$ORIGIN/../g2o_frontend_amd
#include <algorithm>
#include <cstdio>
#include <chrono>
#include <fstream>
#include <iostream>
#include <map>
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
static bool readParameters(std::map<std::string, float>& m, const std::string& fn) {
  std::ifstream is(fn.c_str());
  if (!is) return false;
  std::string line;
  while (std::getline(is, line)) {
    std::istringstream iss(line);
    std::string key; float value;
    if (!(iss >> key >> value)) continue;
    if (key[0] == '#') continue;
    m.insert(std::make_pair(key, value));
  }
  return true;
}
static void put16(FILE* f, const float* m) { for (int i = 0; i < 16; ++i) std::fprintf(f, " %.9g", m[i]); }

int main(int argc, char** argv) {
  if (argc < 4) { std::cout << "USAGE: pwn_hip_tracker_app configuration.txt depthImageList.txt outputPrefix [device]" << std::endl; return 0; }
  std::map<std::string, float> P;
  if (!readParameters(P, argv[1])) { std::cerr << "Error while reading input parameters" << std::endl; return 1; }
  auto get = [&](const char* k, float d) { auto it = P.find(k); return it == P.end() ? d : it->second; };
  const float depthScale = get("depthScale", 0.001f);
  const int imageScale = (int)get("imageScale", 1);
  const int device = argc > 4 ? atoi(argv[4]) : 0;
  const std::string prefix = argv[3];
  try {
    PinholePointProjector converterProjector, alignerProjector;       // pwn_tracker_app: the converter and the aligner own separate projectors
    Matrix3f cameraMatrix;
    cameraMatrix(0,0) = get("fx", 525.0f); cameraMatrix(1,1) = get("fy", 525.0f); cameraMatrix(0,2) = get("cx", 319.5f); cameraMatrix(1,2) = get("cy", 239.5f);
    for (PinholePointProjector* p : { &converterProjector, &alignerProjector }) {
      if (P.count("minDistance")) p->setMinDistance(P["minDistance"]);
      if (P.count("maxDistance")) p->setMaxDistance(P["maxDistance"]);
    }
    StatsCalculatorIntegralImage statsCalculator;
    if (P.count("minImageRadius")) statsCalculator.setMinImageRadius((int)P["minImageRadius"]);
    if (P.count("maxImageRadius")) statsCalculator.setMaxImageRadius((int)P["maxImageRadius"]);
    if (P.count("minPoints")) statsCalculator.setMinPoints((int)P["minPoints"]);
    if (P.count("curvatureThreshold")) statsCalculator.setCurvatureThreshold(P["curvatureThreshold"]);
    if (P.count("worldRadius")) statsCalculator.setWorldRadius(P["worldRadius"]);
    PointInformationMatrixCalculator pointInfo; NormalInformationMatrixCalculator normalInfo;
    if (P.count("informationMatrixCurvatureThreshold")) { pointInfo.setCurvatureThreshold(P["informationMatrixCurvatureThreshold"]); normalInfo.setCurvatureThreshold(P["informationMatrixCurvatureThreshold"]); }
    CorrespondenceFinder finder;
    if (P.count("inlierDistanceThreshold")) finder.setInlierDistanceThreshold(P["inlierDistanceThreshold"]);
    if (P.count("inlierNormalAngularThreshold")) finder.setInlierNormalAngularThreshold(P["inlierNormalAngularThreshold"]);
    if (P.count("inlierCurvatureRatioThreshold")) finder.setInlierCurvatureRatioThreshold(P["inlierCurvatureRatioThreshold"]);
    if (P.count("flatCurvatureThreshold")) finder.setFlatCurvatureThreshold(P["flatCurvatureThreshold"]);
    Linearizer linearizer;
    if (P.count("inlierMaxChi2")) linearizer.setInlierMaxChi2(P["inlierMaxChi2"]);
    if (P.count("robustKernel")) linearizer.setRobustKernel(P["robustKernel"] != 0.f);

    std::ifstream is(argv[2]);
    if (!is) { std::cerr << "Impossible to open depth image list file: " << argv[2] << std::endl; return 1; }
    std::vector<DepthImage> frames;
    {
      std::string line; RawDepthImage raw;
      while (std::getline(is, line)) {
        std::istringstream iss(line);
        std::string timestamp, fn;
        if (!(iss >> timestamp >> fn) || timestamp[0] == '#') continue;
        if (!readPGM16(fn, raw)) { std::cerr << "cannot read " << fn << std::endl; return 1; }
        frames.emplace_back();
        DepthImage& d = frames.back();
        d.create(raw.rows, raw.cols);                                   // DepthImage_convert_16UC1_to_32FC1 (pwn_static.cpp:54-68): scale * raw, zeros stay 0
        for (size_t i = 0; i < d.data.size(); ++i) d.data[i] = raw.data[i] ? depthScale * (float)raw.data[i] : 0.f;
      }
    }
    if (frames.empty()) { std::cerr << "no frames" << std::endl; return 1; }
    Context ctx(device, frames[0].rows, frames[0].cols, 32);
    Aligner aligner(&ctx);
    if (P.count("outerIterations")) aligner.setOuterIterations((int)P["outerIterations"]);
    if (P.count("innerIterations")) aligner.setInnerIterations((int)P["innerIterations"]);
    aligner.setProjector(&alignerProjector); aligner.setCorrespondenceFinder(&finder); aligner.setLinearizer(&linearizer);
    DepthImageConverterIntegralImage converter(&ctx, &converterProjector, &statsCalculator, &pointInfo, &normalInfo);
    PwnTracker tracker(&ctx, &aligner, &converter);
    tracker.setScale(imageScale);
    if (P.count("newFrameInliersFraction")) tracker.setNewFrameInliersFraction(P["newFrameInliersFraction"]);
    const Isometry3f sensorOffset = Isometry3f::Identity();

    // 1. tracking
    FILE* ft = std::fopen((prefix + "_track.txt").c_str(), "w");
    if (!ft) { std::cerr << "cannot write " << prefix << "_track.txt" << std::endl; return 1; }
    std::vector<int> keyframes; std::vector<Isometry3f> keyPoses;
    const bool lookAhead = get("lookAhead", 0.f) != 0.f;       // hand frame k+1 over before frame k is aligned (PwnTracker::prefetch): same track, bit for bit
    if (get("warmUp", 0.f) != 0.f && frames.size() > 1) {      // timing runs: first-use costs (code objects, the look-ahead helper) outside the clock
      tracker.processFrame(frames[0], sensorOffset, cameraMatrix, Isometry3f::Identity(), lookAhead ? &frames[1] : nullptr);
      tracker.processFrame(frames[1], sensorOffset, cameraMatrix);
      tracker.init();
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t k = 0; k < frames.size(); ++k) {
      const DepthImage* next = (lookAhead && k + 1 < frames.size()) ? &frames[k + 1] : nullptr;
      const PwnTracker::FrameResult r = tracker.processFrame(frames[k], sensorOffset, cameraMatrix, Isometry3f::Identity(), next);
      std::fprintf(ft, "%zu %d %d %d %.9g %.9g", k, r.newFrame ? 1 : 0, r.aligned ? 1 : 0, r.inliers, r.error, r.inliersFraction);
      put16(ft, r.globalT.data()); std::fprintf(ft, "\n");
      if (r.newFrame) { keyframes.push_back((int)k); keyPoses.push_back(r.globalT); }
    }
    std::fclose(ft);
    { const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      std::cerr << "tracking: " << frames.size() << " frames, " << dt / frames.size() * 1e3 << " ms per frame (" << frames.size() / dt << " frames/s, file output included)"
                << (lookAhead ? ", look-ahead" : "") << std::endl; }

    // 2. loop-closure pass over the keyframes
    FILE* fc = std::fopen((prefix + "_closures.txt").c_str(), "w");
    CloudCache cache(&tracker, (size_t)get("cacheSize", 2));
    for (int k : keyframes) cache.addFrame(k, frames[k], cameraMatrix, sensorOffset);
    PwnCloserAcceptance acceptance;
    if (P.count("frameMinNonZeroThreshold")) acceptance.frameMinNonZeroThreshold = (int)P["frameMinNonZeroThreshold"];
    if (P.count("frameMaxOutliersThreshold")) acceptance.frameMaxOutliersThreshold = (int)P["frameMaxOutliersThreshold"];
    if (P.count("frameMinInliersThreshold")) acceptance.frameMinInliersThreshold = (int)P["frameMinInliersThreshold"];
    const int r = frames[0].rows, c = frames[0].cols;
    // the multi-GPU form of the same pass: `current` travels as ONE flat buffer, the results come back as 288-byte records.  Here on one
    // context: the replica of `current` that went through export -> host buffer -> import must give, record for record, what the original gives
    int recordCalls = 0, recordsEqual = 1;
    for (size_t a = 0; a < keyframes.size(); ++a) {           // processPartition: one `current` against every earlier keyframe
      std::vector<Cloud*> from, to; std::vector<Isometry3f> guesses; std::vector<int> other;
      for (size_t b = 0; b < a; ++b) { other.push_back(keyframes[b]); guesses.push_back(iso_mul(keyPoses[b].inverse(), keyPoses[a])); }
      if (other.empty()) continue;
      // the clouds of one batched call must all be resident at once: `current` + (cacheSize - 1) candidates per call
      const size_t cacheSize = (size_t)get("cacheSize", 2), chunk = cacheSize > 1 ? cacheSize - 1 : 1;
      for (size_t b0 = 0; b0 < other.size(); b0 += chunk) {
        from.clear(); to.clear();
        std::vector<Isometry3f> g;
        Cloud* cur = cache.get(keyframes[a]);
        const size_t b1 = std::min(other.size(), b0 + chunk);
        for (size_t b = b0; b < b1; ++b) { from.push_back(cache.get(other[b])); to.push_back(cur); g.push_back(guesses[b]); }
        std::vector<PwnMatcherBase::MatcherResult> results;
        tracker.matchCloudsBatch(results, from, to, sensorOffset, sensorOffset, cameraMatrix, r, c, g);
        {
          std::vector<unsigned char> flat(cur->flatSize());
          cur->exportFlat(flat.data(), flat.size());
          Cloud replica(ctx, (int)std::max<size_t>(1, cur->size()));
          replica.importFlat(flat.data(), flat.size());
          std::vector<Cloud*> toReplica(to.size(), &replica);
          std::vector<float> records(from.size() * PWN_HIP_MATCH_RECORD_FLOATS, -1.f);
          std::vector<int> ids; for (size_t i = 0; i < from.size(); ++i) ids.push_back(other[b0 + i]);
          std::vector<PwnMatcherBase::MatcherResult> viaRecords;
          tracker.matchCloudsBatchRecords(records.data(), from, toReplica, sensorOffset, sensorOffset, cameraMatrix, r, c, g, ids, 0, &viaRecords);
          ++recordCalls;
          for (size_t i = 0; i < results.size(); ++i) {
            const float* q = &records[i * PWN_HIP_MATCH_RECORD_FLOATS];
            const PwnMatcherBase::MatcherResult& m = results[i];
            bool ok = (int)q[19] == other[b0 + i] && (int)q[17] == m.cloud_inliers && (int)q[64] == m.image_nonZeros && (int)q[65] == m.image_outliers &&
                      (int)q[66] == m.image_inliers && std::memcmp(&q[67], &m.image_reprojectionDistance, sizeof(float)) == 0 &&
                      viaRecords[i].image_inliers == m.image_inliers && viaRecords[i].cloud_inliers == m.cloud_inliers;
            for (int t = 0; t < 16 && ok; ++t) ok = q[t] == (float)m.transform[t];
            if (!ok) recordsEqual = 0;
          }
        }
        for (size_t i = 0; i < results.size(); ++i) {
          const PwnMatcherBase::MatcherResult& m = results[i];
          std::fprintf(fc, "%d %d %d %d %d %d %d %.9g", other[b0 + i], keyframes[a], acceptance.accept(m) ? 1 : 0, m.cloud_inliers, m.image_nonZeros, m.image_outliers,
                       m.image_inliers, m.image_reprojectionDistance);
          for (int q = 0; q < 16; ++q) std::fprintf(fc, " %.9g", (float)m.transform[q]);
          std::fprintf(fc, "\n");
        }
      }
    }
    std::fprintf(fc, "# cache hits %d misses %d\n", cache.hits, cache.misses);
    std::fprintf(fc, "# replica_record_calls %d records_equal_results %d\n", recordCalls, recordsEqual);
    std::fclose(fc);

    // 3. statistics and priors on the first two keyframes
    FILE* fe = std::fopen((prefix + "_extras.txt").c_str(), "w");
    if (keyframes.size() >= 2) {
      Cloud* a = cache.get(keyframes[0]); Cloud* b = cache.get(keyframes[1]);
      alignerProjector.setCameraMatrix(cameraMatrix); alignerProjector.setImageSize(r, c); alignerProjector.scale(1.0f / imageScale);
      finder.setImageSize(alignerProjector.imageRows(), alignerProjector.imageCols());
      aligner.setSensorOffset(sensorOffset); aligner.setInitialGuess(Isometry3f::Identity());
      aligner.setReferenceCloud(a); aligner.setCurrentCloud(b);
      aligner.setComputeStatistics(true);
      aligner.align();
      aligner.setComputeStatistics(false);
      std::fprintf(fe, "statistics %d %.9g %.9g", aligner.solutionValid() ? 1 : 0, aligner.translationalEigenRatio(), aligner.rotationalEigenRatio());
      for (int i = 0; i < 36; ++i) std::fprintf(fe, " %.9g", aligner.omega().m[i]);
      for (int i = 0; i < 36; ++i) std::fprintf(fe, " %.9g", linearizer.H().m[i]);
      put16(fe, aligner.T().data()); std::fprintf(fe, "\n");
      // priors: an absolute prior pulling towards the identity and a relative prior on the increment (aligner.cpp:96-108)
      Matrix6f info = Matrix6f::Identity();
      for (int i = 0; i < 6; ++i) info(i,i) = 1000.f;
      aligner.setReferenceCloud(a); aligner.setCurrentCloud(b);        // clears priors
      aligner.addAbsolutePrior(Isometry3f::Identity(), Isometry3f::Identity(), info);
      aligner.addRelativePrior(Isometry3f::Identity(), info);
      aligner.align();
      std::fprintf(fe, "priors %zu %d %.9g", aligner.numPriors(), aligner.inliers(), aligner.error());
      put16(fe, aligner.T().data()); std::fprintf(fe, "\n");
      aligner.clearPriors();
      // stage-level calls: project both clouds, CorrespondenceFinder::compute and Linearizer::update at the identity
      IntImage ri, ci, itv; DepthImage rd, cd;
      alignerProjector.setTransform(Isometry3f::Identity());
      alignerProjector.project(ctx, ri, rd, *a); alignerProjector.project(ctx, ci, cd, *b);
      aligner.setReferenceCloud(a); aligner.setCurrentCloud(b);
      aligner.computeCorrespondences(ri, ci, Isometry3f::Identity());
      aligner.linearize(Isometry3f::Identity());
      long long sumIdx = 0; for (int v : ri.data) sumIdx += v;
      std::fprintf(fe, "stages %d %d %d %.9g %lld", finder.numCorrespondences(), finder.numCandidates(), linearizer.inliers(), linearizer.error(), sumIdx);
      for (int i = 0; i < 36; ++i) std::fprintf(fe, " %.9g", linearizer.H().m[i]);
      for (int i = 0; i < 6; ++i) std::fprintf(fe, " %.9g", linearizer.b()[i]);
      std::fprintf(fe, "\n");
      // unProject + projectIntervals of the first (scaled) frame through the projector alone
      DepthImage scaled; DepthImage_scale(ctx, scaled, frames[keyframes[0]], imageScale);
      Cloud pts(ctx, scaled.rows * scaled.cols);
      IntImage ui;
      alignerProjector.unProject(ctx, pts, ui, scaled);
      alignerProjector.projectIntervals(ctx, itv, scaled, statsCalculator.worldRadius());
      long long sumU = 0, sumI = 0; for (int v : ui.data) sumU += v; for (int v : itv.data) sumI += v;
      std::fprintf(fe, "projector %zu %lld %lld\n", pts.size(), sumU, sumI);
    }
    std::fclose(fe);
  } catch (const Error& e) {
    std::cerr << e.what() << std::endl;
    return 2;
  }
  return 0;
}
