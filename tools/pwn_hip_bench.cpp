// pwn_hip_bench -- the headline workload of bench.py driven from C++ through the host mirror: P depth pairs per step, every step converts
// the 2P raw uint16 frames (resident in HBM) and aligns the P pairs (10 Gauss-Newton iterations) in batched calls.  Frames come from a list
// of 16-bit PGM files; pair i is (frame 2i mod F, frame 2i+1 mod F).  Prints one line: pairs, steps, ms per step, alignments per second,
// then one line per distinct pair with the pose and the chi2 trace (for cross-checking against the Python mirror).
//
// build (g2o_frontend_amd/build.py: build_tools):
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I. -I/opt/rocm/include tools/pwn_hip_bench.cpp -o tools/pwn_hip_bench
//       -Lg2o_frontend_amd -lpwn_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$ORIGIN/../g2o_frontend_amd
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
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

int main(int argc, char** argv) {
  if (argc < 2) { std::cout << "USAGE: pwn_hip_bench depthImageList.txt [pairs=128] [steps=5] [warmup=1] [device=0]" << std::endl; return 0; }
  const int P = argc > 2 ? atoi(argv[2]) : 128, steps = argc > 3 ? atoi(argv[3]) : 5, warmup = argc > 4 ? atoi(argv[4]) : 1, device = argc > 5 ? atoi(argv[5]) : 0;
  std::vector<RawDepthImage> frames;
  {
    std::ifstream is(argv[1]);
    std::string line;
    while (std::getline(is, line)) {
      std::istringstream iss(line);
      std::string fn;
      if (!(iss >> fn) || fn[0] == '#') continue;
      frames.emplace_back();
      if (!readPGM16(fn, frames.back())) { std::cerr << "cannot read " << fn << std::endl; return 1; }
    }
  }
  if (frames.size() < 2 || frames.size() % 2) { std::cerr << "need an even number of frames (reference, current, reference, ...)" << std::endl; return 1; }
  const int rows = frames[0].rows, cols = frames[0].cols;
  try {
    Context ctx(device, rows, cols, 128);      // workspace slots: two streams x sub-batches of 64 frames / pairs
    // pwn_core/conf/pwn_aligner_1_1.conf as pwn_simple_aligner.cpp:214-269 applies it
    PinholePointProjector projector;
    Matrix3f K; K(0,0) = 525.f; K(1,1) = 525.f; K(0,2) = 319.5f; K(1,2) = 239.5f;     // VGA intrinsics
    projector.setCameraMatrix(K); projector.setImageSize(rows, cols); projector.setMinDistance(0.5f); projector.setMaxDistance(4.5f);
    StatsCalculatorIntegralImage stats; stats.setWorldRadius(0.1f); stats.setMinImageRadius(10); stats.setMaxImageRadius(30); stats.setMinPoints(50); stats.setCurvatureThreshold(0.2f);
    PointInformationMatrixCalculator pinfo; NormalInformationMatrixCalculator ninfo;
    DepthImageConverterIntegralImage converter(&ctx, &projector, &stats, &pinfo, &ninfo);
    CorrespondenceFinder finder; finder.setImageSize(rows, cols); finder.setInlierDistanceThreshold(1.0f); finder.setInlierNormalAngularThreshold(0.95f);
    finder.setFlatCurvatureThreshold(0.02f); finder.setInlierCurvatureRatioThreshold(1.3f);
    Linearizer linearizer; linearizer.setInlierMaxChi2(9e3f); linearizer.setRobustKernel(true);
    Aligner aligner(&ctx); aligner.setProjector(&projector); aligner.setCorrespondenceFinder(&finder); aligner.setLinearizer(&linearizer);
    aligner.setOuterIterations(10); aligner.setInnerIterations(1);

    // the raw frames live in HBM, as in bench.py: the timed region starts with the inputs resident
    if (hipSetDevice(device) != hipSuccess) { std::cerr << "hipSetDevice failed" << std::endl; return 1; }
    std::vector<uint16_t*> dev(frames.size(), nullptr);
    for (size_t i = 0; i < frames.size(); ++i) {
      if (hipMalloc((void**)&dev[i], frames[i].data.size() * 2) != hipSuccess ||
          hipMemcpy(dev[i], frames[i].data.data(), frames[i].data.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { std::cerr << "device upload failed" << std::endl; return 1; }
    }
    std::vector<Cloud*> clouds(2 * (size_t)P), refs(P), curs(P);
    std::vector<const uint16_t*> raw(2 * (size_t)P);
    for (int i = 0; i < 2 * P; ++i) { clouds[i] = new Cloud(ctx, rows * cols); raw[i] = dev[i % frames.size()]; }
    for (int i = 0; i < P; ++i) { refs[i] = clouds[2 * i]; curs[i] = clouds[2 * i + 1]; }
    std::vector<pwn_hip_align_result> results;
    auto step = [&]() {
      converter.computeBatchRaw(clouds, raw, 0.001f, rows, cols);
      projector.setImageSize(rows, cols);
      results = aligner.alignBatch(refs, curs);
    };
    for (int i = 0; i < warmup; ++i) step();
    ctx.synchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; ++i) step();
    ctx.synchronize();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("pairs %d steps %d ms_per_step %.4f alignments_per_s %.1f\n", P, steps, dt / steps * 1e3, (double)P * steps / dt);
    const size_t distinct = std::min((size_t)P, frames.size() / 2);
    for (size_t i = 0; i < distinct; ++i) {
      const pwn_hip_align_result& r = results[i];
      std::printf("pair %zu %d %.9g", i, r.inliers, r.error);
      for (int q = 0; q < 16; ++q) std::printf(" %.9g", r.T[q]);
      for (int q = 0; q < r.iterations; ++q) std::printf(" %.9g", r.chi2[q]);
      std::printf("\n");
    }
    for (Cloud* c : clouds) delete c;
    for (uint16_t* d : dev) (void)hipFree(d);
  } catch (const Error& e) {
    std::cerr << e.what() << std::endl;
    return 2;
  }
  return 0;
}
