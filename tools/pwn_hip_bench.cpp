// pwn_hip_bench -- the headline workload of bench.py driven from C++ through the host mirror: P depth pairs per step, every step converts
// the 2P raw uint16 frames and aligns the P pairs (10 Gauss-Newton iterations) in batched calls.  Frames come from a list of 16-bit PGM
// files; pair i is (frame 2i mod F, frame 2i+1 mod F).  Prints one line: pairs, steps, ms per step, alignments per second, then one line
// per distinct pair with the pose and the chi2 trace (for cross-checking against the Python mirror).
//   mode 0 (default): frames resident in HBM (bench.py's `value`)
//   mode 1: frames in one page-locked host block, handed to the convert call as host pointers (copied one sub-batch ahead of the kernels)
//   mode 2: the same host block, uploaded by the caller into one of two device blocks with pwn_hip_copy_async while the previous batch is
//           being aligned (the PCIe-inclusive rate of a caller that double-buffers)
// No HIP headers, no HIP runtime on the link line: device and page-locked memory come from the C-ABI (pwn_hip_device_alloc / _host_alloc).
//
// build (g2o_frontend_amd/build.py: build_tools):
//   g++ -O2 -std=c++17 -I. tools/pwn_hip_bench.cpp -o tools/pwn_hip_bench -Lg2o_frontend_amd -lpwn_hip -Wl,-rpath,$ORIGIN/../g2o_frontend_amd
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

int main(int argc, char** argv) {
  if (argc < 2) { std::cout << "USAGE: pwn_hip_bench depthImageList.txt [pairs=128] [steps=5] [warmup=1] [device=0] [mode=0: frames resident, convert then align | 1: host frames | 2: double-buffered uploads | 3: resident, one submission]" << std::endl; return 0; }
  const int P = argc > 2 ? atoi(argv[2]) : 128, steps = argc > 3 ? atoi(argv[3]) : 5, warmup = argc > 4 ? atoi(argv[4]) : 1, device = argc > 5 ? atoi(argv[5]) : 0;
  const int mode = argc > 6 ? atoi(argv[6]) : 0;
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

    const size_t fpix = (size_t)rows * cols, fbytes = fpix * 2, nfr = 2 * (size_t)P;
    auto check = [&](int rc, const char* what) { if (rc != PWN_HIP_OK) throw Error(rc, std::string(what) + ": " + pwn_hip_last_error_string(ctx.handle())); };
    // one page-locked host block with the 2P frames of a step (modes 1, 2), up to two device blocks (mode 0: one, filled once)
    uint16_t* host = nullptr;
    check(pwn_hip_host_alloc((void**)&host, nfr * fbytes), "pwn_hip_host_alloc");
    for (size_t i = 0; i < nfr; ++i) std::memcpy(host + i * fpix, frames[i % frames.size()].data.data(), fbytes);
    uint16_t* dev[2] = { nullptr, nullptr };
    for (int b = 0; b < (mode == 2 ? 2 : ((mode == 0 || mode == 3) ? 1 : 0)); ++b) check(pwn_hip_device_alloc(ctx.handle(), (void**)&dev[b], nfr * fbytes), "pwn_hip_device_alloc");
    if (mode == 0 || mode == 3) check(pwn_hip_copy(ctx.handle(), dev[0], host, nfr * fbytes), "pwn_hip_copy");
    std::vector<Cloud*> clouds(nfr), refs(P), curs(P);
    std::vector<const uint16_t*> raw(nfr);
    for (size_t i = 0; i < nfr; ++i) clouds[i] = new Cloud(ctx, rows * cols);
    for (int i = 0; i < P; ++i) { refs[i] = clouds[2 * i]; curs[i] = clouds[2 * i + 1]; }
    std::vector<pwn_hip_align_result> results;
    int flip = 0;
    if (mode == 2) check(pwn_hip_copy_async(ctx.handle(), dev[0], host, nfr * fbytes), "pwn_hip_copy_async");      // prime the first block
    std::vector<const uint16_t*> rawRef(P), rawCur(P);
    auto step = [&]() {
      const uint16_t* base = mode == 1 ? host : dev[mode == 2 ? flip : 0];
      for (size_t i = 0; i < nfr; ++i) raw[i] = base + i * fpix;
      if (mode == 3) {      // the whole step as ONE submission (pwn_hip_convert_align_batch_u16): frames resident, same bits as mode 0
        for (int i = 0; i < P; ++i) { rawRef[i] = raw[2 * i]; rawCur[i] = raw[2 * i + 1]; }
        projector.setImageSize(rows, cols);
        results = aligner.convertAlignBatch(converter, refs, curs, rawRef, rawCur, 0.001f, rows, cols);
        return;
      }
      converter.computeBatchRaw(clouds, raw, 0.001f, rows, cols);                       // waits for the copies queued into `base`
      if (mode == 2) { flip ^= 1; check(pwn_hip_copy_async(ctx.handle(), dev[flip], host, nfr * fbytes), "pwn_hip_copy_async"); }   // the next step's frames travel during the alignment
      projector.setImageSize(rows, cols);
      results = aligner.alignBatch(refs, curs);
    };
    for (int i = 0; i < warmup; ++i) step();
    ctx.synchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; ++i) step();
    ctx.synchronize();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("pairs %d steps %d ms_per_step %.4f alignments_per_s %.1f mode %d\n", P, steps, dt / steps * 1e3, (double)P * steps / dt, mode);
    const size_t distinct = std::min((size_t)P, frames.size() / 2);
    for (size_t i = 0; i < distinct; ++i) {
      const pwn_hip_align_result& r = results[i];
      std::printf("pair %zu %d %.9g", i, r.inliers, r.error);
      for (int q = 0; q < 16; ++q) std::printf(" %.9g", r.T[q]);
      for (int q = 0; q < r.iterations; ++q) std::printf(" %.9g", r.chi2[q]);
      std::printf("\n");
    }
    for (Cloud* c : clouds) delete c;
    for (uint16_t* d : dev) check(pwn_hip_device_free(ctx.handle(), d), "pwn_hip_device_free");
    check(pwn_hip_host_free(host), "pwn_hip_host_free");
  } catch (const Error& e) {
    std::cerr << e.what() << std::endl;
    return 2;
  }
  return 0;
}
