// pwn_hip_simple_aligner -- the reference's sequential-odometry harness (pwn_core/pwn_simple_aligner.cpp:28-269)
// on the MI355X path: text configuration -> pwn_hip objects; for each "timestamp filename" line read a 16-bit depth
// image, convert it to a cloud, align it against the previous cloud with an identity guess, chain the global pose and
// write a TUM-style trajectory line.
//
// Differences from the reference harness: depth images are binary PGM (P5, maxval 65535) instead of PNG (no OpenCV
// here), and the per-frame .pwn cloud dump (cloud.cpp:25-133) is not written.
//
// build (see g2o_frontend_amd/build.py: build_tools):
//   g++ -O2 -std=c++17 -I. tools/pwn_hip_simple_aligner.cpp -o tools/pwn_hip_simple_aligner -Lg2o_frontend_amd -lpwn_hip -Wl,-rpath,$ORIGIN/../g2o_frontend_amd
#include <cstdio>
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

// pwn_simple_aligner.cpp:190-212: "key value" lines, unparsable lines ignored, first occurrence wins
static bool fillInputParametersMap(std::map<std::string, float>& m, const std::string& fn) {
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
    std::cout << "USAGE: pwn_hip_simple_aligner configuration.txt depthImageList.txt visualOdometry.txt [device]" << std::endl;
    return 0;
  }
  std::map<std::string, float> P;
  if (!fillInputParametersMap(P, argv[1])) { std::cerr << "Error while reading input parameters" << std::endl; return 1; }
  auto get = [&](const char* k, float d) { auto it = P.find(k); return it == P.end() ? d : it->second; };
  const float depthScale = get("depthScale", 0.001f);
  const int imageScale = (int)get("imageScale", 1);
  const int device = argc > 4 ? atoi(argv[4]) : 0;
  try {
    // pwn_simple_aligner.cpp:214-269
    PinholePointProjector pointProjector;
    Matrix3f cameraMatrix;
    cameraMatrix(0,0) = get("fx", 525.0f); cameraMatrix(1,1) = get("fy", 525.0f); cameraMatrix(0,2) = get("cx", 319.5f); cameraMatrix(1,2) = get("cy", 239.5f);
    if (P.count("minDistance")) pointProjector.setMinDistance(P["minDistance"]);
    if (P.count("maxDistance")) pointProjector.setMaxDistance(P["maxDistance"]);
    pointProjector.setCameraMatrix(cameraMatrix);
    StatsCalculatorIntegralImage statsCalculator;
    if (P.count("minImageRadius")) statsCalculator.setMinImageRadius((int)P["minImageRadius"]);
    if (P.count("maxImageRadius")) statsCalculator.setMaxImageRadius((int)P["maxImageRadius"]);
    if (P.count("minPoints")) statsCalculator.setMinPoints((int)P["minPoints"]);
    if (P.count("curvatureThreshold")) statsCalculator.setCurvatureThreshold(P["curvatureThreshold"]);
    if (P.count("worldRadius")) statsCalculator.setWorldRadius(P["worldRadius"]);
    PointInformationMatrixCalculator pointInformationMatrixCalculator;
    NormalInformationMatrixCalculator normalInformationMatrixCalculator;
    if (P.count("informationMatrixCurvatureThreshold")) {
      pointInformationMatrixCalculator.setCurvatureThreshold(P["informationMatrixCurvatureThreshold"]);
      normalInformationMatrixCalculator.setCurvatureThreshold(P["informationMatrixCurvatureThreshold"]);
    }
    CorrespondenceFinder correspondenceFinder;
    if (P.count("inlierDistanceThreshold")) correspondenceFinder.setInlierDistanceThreshold(P["inlierDistanceThreshold"]);
    if (P.count("inlierNormalAngularThreshold")) correspondenceFinder.setInlierNormalAngularThreshold(P["inlierNormalAngularThreshold"]);
    if (P.count("inlierCurvatureRatioThreshold")) correspondenceFinder.setInlierCurvatureRatioThreshold(P["inlierCurvatureRatioThreshold"]);
    if (P.count("flatCurvatureThreshold")) correspondenceFinder.setFlatCurvatureThreshold(P["flatCurvatureThreshold"]);
    Linearizer linearizer;
    if (P.count("inlierMaxChi2")) linearizer.setInlierMaxChi2(P["inlierMaxChi2"]);
    if (P.count("robustKernel")) linearizer.setRobustKernel(P["robustKernel"] != 0.f);

    std::ifstream is(argv[2]);
    if (!is) { std::cerr << "Impossible to open depth image list file: " << argv[2] << std::endl; return 1; }
    std::ofstream os(argv[3]);
    if (!os) { std::cerr << "Impossible to open visual odometry file: " << argv[3] << std::endl; return 1; }
    os.precision(9);

    Context* ctx = nullptr; Aligner* aligner = nullptr; DepthImageConverterIntegralImage* converter = nullptr;
    Cloud* cloud = nullptr; Cloud* previousCloud = nullptr;
    bool firstDepth = true;
    Isometry3f sensorOffset = Isometry3f::Identity();
    float g0[6] = { get("tx", 0.f), get("ty", 0.f), get("tz", 0.f), get("qx", 0.f), get("qy", 0.f), get("qz", 0.f) };
    Isometry3f globalT = v2t(g0);
    RawDepthImage rawDepth; DepthImage depth, scaledDepth;
    std::string line;
    while (std::getline(is, line)) {
      std::istringstream iss(line);
      std::string timestamp, depthFilename;
      if (!(iss >> timestamp >> depthFilename)) continue;
      if (timestamp[0] == '#') continue;
      if (!readPGM16(depthFilename, rawDepth)) { std::cerr << "cannot read " << depthFilename << std::endl; return 1; }
      if (!ctx) {
        ctx = new Context(device, rawDepth.rows, rawDepth.cols, 1);
        aligner = new Aligner(ctx);
        if (P.count("outerIterations")) aligner->setOuterIterations((int)P["outerIterations"]);
        if (P.count("innerIterations")) aligner->setInnerIterations((int)P["innerIterations"]);
        aligner->setProjector(&pointProjector); aligner->setCorrespondenceFinder(&correspondenceFinder); aligner->setLinearizer(&linearizer);
        converter = new DepthImageConverterIntegralImage(ctx, &pointProjector, &statsCalculator, &pointInformationMatrixCalculator, &normalInformationMatrixCalculator);
      }
      DepthImage_convert_16UC1_to_32FC1(*ctx, depth, rawDepth, depthScale);        // :138
      DepthImage_scale(*ctx, scaledDepth, depth, imageScale);                       // :139
      if (firstDepth) {                                                             // :142-150
        const float invScale = 1.0f / imageScale;
        Matrix3f scaled = pointProjector.cameraMatrix();
        for (int i = 0; i < 9; ++i) scaled.m[i] = scaled.m[i] * invScale;
        scaled(2,2) = 1.0f;
        pointProjector.setCameraMatrix(scaled);
        pointProjector.setImageSize(scaledDepth.rows, scaledDepth.cols);
        correspondenceFinder.setImageSize(scaledDepth.rows, scaledDepth.cols);
      }
      cloud = new Cloud(*ctx, scaledDepth.rows * scaledDepth.cols);
      converter->compute(*cloud, scaledDepth, sensorOffset);                        // :155
      if (!firstDepth) {                                                            // :160-168
        aligner->setReferenceCloud(previousCloud);
        aligner->setCurrentCloud(cloud);
        aligner->setInitialGuess(Isometry3f::Identity());
        aligner->setSensorOffset(sensorOffset);
        aligner->align();
        globalT = globalT * aligner->T();
        globalT.forceLastRow();
        delete previousCloud;
      }
      float v[6]; t2v(globalT, v);
      const float n2 = v[3] * v[3] + v[4] * v[4] + v[5] * v[5];
      const float qw = std::sqrt(n2 < 1.f ? 1.f - n2 : 0.f);
      os << timestamp << " " << v[0] << " " << v[1] << " " << v[2] << " " << v[3] << " " << v[4] << " " << v[5] << " " << qw
         << " " << (firstDepth ? 0.f : aligner->error()) << " " << (firstDepth ? 0 : aligner->inliers()) << std::endl;
      previousCloud = cloud;
      firstDepth = false;
    }
    delete previousCloud; delete converter; delete aligner; delete ctx;
  } catch (const Error& e) {
    std::cerr << e.what() << std::endl;
    return 2;
  }
  return 0;
}
