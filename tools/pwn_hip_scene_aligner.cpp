// pwn_hip_scene_aligner -- the mapping loop of the reference's pwn_aligner (pwn_core/pwn_aligner.cpp:150-262) on the MI355X path,
// over the C++ host mirror: for every depth image convert it to a cloud; from the second one on render the scene into the current
// view, convert the rendering to a sub-scene, align the new cloud against it and chain the pose; add the cloud to the scene
// (Cloud::add) and merge (Merger::merge); every chunkStep frames save the scene (Cloud::save) and start a new one.
//
// Differences from the reference harness: depth images are binary PGM (P5, maxval 65535) instead of PNG (no OpenCV here); the
// configuration keys are the reference's (pwn_aligner.cpp:295-327) plus chunkStep.
//
//   g++ -O2 -std=c++17 -I. tools/pwn_hip_scene_aligner.cpp -o tools/pwn_hip_scene_aligner -Lg2o_frontend_amd -lpwn_hip -Wl,-rpath,$ORIGIN/../g2o_frontend_amd
#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
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
static bool fillInputParametersMap(std::map<std::string, float>& m, const std::string& fn) {      // pwn_aligner.cpp:264-289
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

int main(int argc, char** argv) {
  if (argc < 4) {
    std::cout << "USAGE: pwn_hip_scene_aligner configuration.txt depthImageList.txt outputPrefix" << std::endl;
    return 0;
  }
  std::map<std::string, float> P;
  if (!fillInputParametersMap(P, argv[1])) { std::cerr << "Error while reading input parameters" << std::endl; return 1; }
  auto get = [&](const char* k, float d) { auto it = P.find(k); return it == P.end() ? d : it->second; };
  const float depthScale = get("depthScale", 0.001f);
  const int imageScale = (int)get("imageScale", 1.f);
  const int chunkStep = (int)get("chunkStep", 1000.f);
  const std::string prefix = argv[3];
  try {
    std::ifstream list(argv[2]);
    if (!list) { std::cerr << "cannot open " << argv[2] << std::endl; return 1; }
    std::vector<std::string> files; std::string line;
    while (std::getline(list, line)) { if (line.empty() || line[0] == '#') continue; std::istringstream ls(line); std::string ts, fn; ls >> ts >> fn; files.push_back(fn); }
    RawDepthImage raw;
    if (files.empty() || !readPGM16(files[0], raw)) { std::cerr << "no readable depth image" << std::endl; return 1; }
    const int rows = raw.rows / imageScale, cols = raw.cols / imageScale;
    Context ctx(0, raw.rows, raw.cols, 2);
    PinholePointProjector pointProjector;
    Matrix3f K; K(0,0) = get("fx", 525.f); K(1,1) = get("fy", 525.f); K(0,2) = get("cx", 319.5f); K(1,2) = get("cy", 239.5f);
    const float invScale = 1.0f / imageScale;                                       // pwn_aligner.cpp:155-160
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) K(r,c) = K(r,c) * invScale;
    K(2,2) = 1.0f;
    pointProjector.setCameraMatrix(K);
    pointProjector.setMinDistance(get("minDistance", 0.5f)); pointProjector.setMaxDistance(get("maxDistance", 4.5f));
    pointProjector.setImageSize(rows, cols);
    StatsCalculatorIntegralImage stats;
    stats.setMinImageRadius((int)get("minImageRadius", 10)); stats.setMaxImageRadius((int)get("maxImageRadius", 30));
    stats.setMinPoints((int)get("minPoints", 50)); stats.setCurvatureThreshold(get("curvatureThreshold", 0.2f)); stats.setWorldRadius(get("worldRadius", 0.1f));
    PointInformationMatrixCalculator pinfo; NormalInformationMatrixCalculator ninfo;
    pinfo.setCurvatureThreshold(get("informationMatrixCurvatureThreshold", 0.02f)); ninfo.setCurvatureThreshold(get("informationMatrixCurvatureThreshold", 0.02f));
    DepthImageConverterIntegralImage converter(&ctx, &pointProjector, &stats, &pinfo, &ninfo);
    converter.setComputeGaussians(true);
    CorrespondenceFinder finder;
    finder.setInlierDistanceThreshold(get("inlierDistanceThreshold", 1.0f)); finder.setInlierNormalAngularThreshold(get("inlierNormalAngularThreshold", 0.95f));
    finder.setFlatCurvatureThreshold(get("flatCurvatureThreshold", 0.02f)); finder.setInlierCurvatureRatioThreshold(get("inlierCurvatureRatioThreshold", 1.3f));
    finder.setImageSize(rows, cols);
    Linearizer linearizer; linearizer.setInlierMaxChi2(get("inlierMaxChi2", 9e3f)); linearizer.setRobustKernel(get("robustKernel", 1.f) != 0.f);
    Aligner aligner(&ctx);
    aligner.setProjector(&pointProjector); aligner.setLinearizer(&linearizer); aligner.setCorrespondenceFinder(&finder);
    aligner.setOuterIterations((int)get("outerIterations", 10)); aligner.setInnerIterations((int)get("innerIterations", 1));
    Merger merger;
    merger.setDepthImageConverter(&converter); merger.setImageSize(rows, cols);
    merger.setMaxPointDepth(get("depthThreshold", 10.0f)); merger.setNormalThreshold(get("normalThreshold", std::cos(10 * (float)M_PI / 180.0f)));
    merger.setDistanceThreshold(get("distanceThreshold", 0.1f));

    const Isometry3f sensorOffset;                       // identity
    Isometry3f globalT, sceneT;
    const int sceneCapacity = (int)std::min<size_t>(files.size(), (size_t)chunkStep + 1) * rows * cols;
    std::unique_ptr<Cloud> referenceScene(new Cloud(ctx, sceneCapacity));
    Cloud subscene(ctx, rows * cols), cloud(ctx, rows * cols);
    DepthImage depth, scaled, rendered; IntImage renderedIndex;
    std::ofstream os((prefix + "_trajectory.txt").c_str());
    os.precision(9);
    int counter = 0; bool firstDepth = true;
    for (size_t fi = 0; fi < files.size(); ++fi) {
      if (!readPGM16(files[fi], raw)) { std::cerr << "cannot read " << files[fi] << std::endl; return 1; }
      depth.create(raw.rows, raw.cols);
      ctx.check(pwn_hip_depth_u16_to_f32(ctx.handle(), raw.data.data(), depth.data.data(), raw.rows * raw.cols, depthScale));
      scaled.create(rows, cols);
      ctx.check(pwn_hip_depth_scale(ctx.handle(), depth.data.data(), raw.rows, raw.cols, imageScale, 0.01f, scaled.data.data()));
      converter.compute(cloud, scaled, sensorOffset);
      if (!firstDepth) {                                                              // pwn_aligner.cpp:172-191
        pointProjector.setImageSize(rows, cols);
        pointProjector.setTransform(sceneT * sensorOffset);
        pointProjector.project(ctx, renderedIndex, rendered, *referenceScene);
        converter.compute(subscene, rendered, sensorOffset);
        pointProjector.setTransform(Isometry3f::Identity());
        aligner.setReferenceCloud(&subscene); aligner.setCurrentCloud(&cloud);
        aligner.setInitialGuess(Isometry3f::Identity()); aligner.setSensorOffset(sensorOffset);
        aligner.align();
        globalT = globalT * aligner.T(); globalT.forceLastRow();
        sceneT = sceneT * aligner.T(); sceneT.forceLastRow();
      }
      if (!firstDepth && counter++ % chunkStep == 0) {                                // :193-204 (counter is post-incremented in the reference)
        char buffer[1024];
        std::snprintf(buffer, sizeof(buffer), "%s_scene-%03d.pwn", prefix.c_str(), counter);
        referenceScene->save(buffer, sceneT.inverse() * globalT, 1, true);
        sceneT.setIdentity();
        referenceScene.reset(new Cloud(ctx, sceneCapacity));                         // delete referenceScene; new Cloud()
      }
      referenceScene->add(cloud, sceneT);                                             // :206-207
      merger.merge(referenceScene.get(), sceneT * sensorOffset);
      pointProjector.setTransform(Isometry3f::Identity());
      float v[6]; t2v(globalT, v);
      os << fi << " " << v[0] << " " << v[1] << " " << v[2] << " " << v[3] << " " << v[4] << " " << v[5] << " " << referenceScene->size() << std::endl;
      firstDepth = false;
    }
    char buffer[1024];
    std::snprintf(buffer, sizeof(buffer), "%s_scene-%03d.pwn", prefix.c_str(), counter);
    referenceScene->save(buffer, sceneT.inverse() * globalT, 1, true);
  } catch (const Error& e) {
    std::cerr << "pwn_hip error: " << e.what() << std::endl;
    return 2;
  }
  return 0;
}
