// pwn_hip.hpp -- header-only C++ host mirror of the reference's pwn_core operator interface over the C-ABI
// (include/pwn_hip.h).  Class / method names and argument meaning follow g2o_frontend/pwn_core so that a caller
// such as PwnMatcherBase (pwn_tracker/pwn_matcher_base.cpp:57-183) or pwn_simple_aligner.cpp compiles against
// these classes with a namespace switch; the heavy methods are one call into libpwn_hip.so each.
//
// Differences a maintainer has to know about:
//   * no Eigen here: Matrix3f / Isometry3f are plain column-major float arrays with Eigen's memory layout, so
//     `Isometry3f(eigenIso.matrix().data())` / `std::memcpy(eigen.data(), m.data(), ...)` convert both ways;
//   * Cloud lives in device memory (pwn_hip_cloud); points()/normals()/... download on demand;
//   * errors: the reference asserts (UB in Release); these classes throw pwn_hip::Error carrying the ABI status.
#pragma once

#include <cmath>
#include <cstring>
#include <list>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pwn_hip.h"

namespace pwn_hip {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error("pwn_hip status " + std::to_string(c) + ": " + m), code(c) {}
};

struct Matrix3f {
  float m[9];
  Matrix3f() { std::memset(m, 0, sizeof(m)); m[0] = m[4] = m[8] = 1.f; }
  float& operator()(int r, int c) { return m[r + 3 * c]; }
  float operator()(int r, int c) const { return m[r + 3 * c]; }
  const float* data() const { return m; }
  float* data() { return m; }
};
struct Isometry3f {
  float m[16];
  Isometry3f() { setIdentity(); }
  explicit Isometry3f(const float* colmajor16) { std::memcpy(m, colmajor16, sizeof(m)); }
  static Isometry3f Identity() { return Isometry3f(); }
  void setIdentity() { std::memset(m, 0, sizeof(m)); m[0] = m[5] = m[10] = m[15] = 1.f; }
  float& operator()(int r, int c) { return m[r + 4 * c]; }
  float operator()(int r, int c) const { return m[r + 4 * c]; }
  const float* data() const { return m; }
  float* data() { return m; }
  void forceLastRow() { m[3] = 0.f; m[7] = 0.f; m[11] = 0.f; m[15] = 1.f; }
  Isometry3f inverse() const { Isometry3f r; pwn_hip_iso_inverse(m, r.m); return r; }       // Isometry3f::inverse(): R^T, -R^T t
  // Isometry3f * Isometry3f (linear = Ra*Rb, translation = Ra*tb + ta), float, left-to-right sums
  Isometry3f operator*(const Isometry3f& b) const {
    Isometry3f r;
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) { float s = (*this)(i,0) * b(0,j); s = s + (*this)(i,1) * b(1,j); s = s + (*this)(i,2) * b(2,j); r(i,j) = s; }
    for (int i = 0; i < 3; ++i) { float s = (*this)(i,0) * b(0,3); s = s + (*this)(i,1) * b(1,3); s = s + (*this)(i,2) * b(2,3); r(i,3) = s + (*this)(i,3); }
    r.forceLastRow();
    return r;
  }
};
struct DepthImage {                       // pwn_typedefs.h:57-62: float metres, row-major
  int rows = 0, cols = 0;
  std::vector<float> data;
  void create(int r, int c) { rows = r; cols = c; data.assign((size_t)r * c, 0.f); }
  float& operator()(int r, int c) { return data[(size_t)r * cols + c]; }
};
struct RawDepthImage { int rows = 0, cols = 0; std::vector<uint16_t> data; };
struct IntImage { int rows = 0, cols = 0; std::vector<int> data; void create(int r, int c) { rows = r; cols = c; data.assign((size_t)r * c, -1); } };

inline Isometry3f v2t(const float v[6]) { Isometry3f T; pwn_hip_v2t(v, T.data()); return T; }      // bm_se3.h:37-43
inline void t2v(const Isometry3f& T, float v[6]) { pwn_hip_t2v(T.data(), v); }                     // bm_se3.h:45-52
// Isometry3f * Isometry3f with the evaluation order of the CPU path (the library's host helper; what the device-side pose chaining uses)
inline Isometry3f iso_mul(const Isometry3f& a, const Isometry3f& b) { Isometry3f r; pwn_hip_iso_mul(a.data(), b.data(), r.data()); return r; }
struct Matrix6f {                          // bm_defs.h:11, column-major
  float m[36];
  Matrix6f() { std::memset(m, 0, sizeof(m)); }
  static Matrix6f Identity() { Matrix6f r; for (int i = 0; i < 6; ++i) r.m[7 * i] = 1.f; return r; }
  float& operator()(int r, int c) { return m[r + 6 * c]; }
  float operator()(int r, int c) const { return m[r + 6 * c]; }
  const float* data() const { return m; }
  float* data() { return m; }
};
struct Vector6f { float v[6] = { 0, 0, 0, 0, 0, 0 }; float& operator[](int i) { return v[i]; } float operator[](int i) const { return v[i]; } };

// One per (GPU, host thread); every object below borrows it (non-owning raw pointer, like the reference's
// collaborator pointers: aligner.h:381-386).
class Context {
 public:
  Context(int device, int maxRows, int maxCols, int maxBatch = 1) {
    int rc = pwn_hip_ctx_create(&_ctx, device, maxRows, maxCols, maxBatch);
    if (rc) throw Error(rc, pwn_hip_last_error_string(nullptr));
  }
  ~Context() { pwn_hip_ctx_destroy(_ctx); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  pwn_hip_ctx* handle() const { return _ctx; }
  void check(int rc) const { if (rc) throw Error(rc, pwn_hip_last_error_string(_ctx)); }
  // batch calls: frames / pairs per kernel launch and the number of HIP streams the sub-batches are dealt over (include/pwn_hip.h)
  void setSubbatch(int frames, int pairs) { check(pwn_hip_ctx_set_subbatch(_ctx, frames, pairs)); }
  void setConcurrency(int streams) { check(pwn_hip_ctx_set_concurrency(_ctx, streams)); }
  // PWN_HIP_OMEGA_SYM6 (default) / PWN_HIP_OMEGA_EXACT9 (every entry of Omega_p bit-identical to pwn_core's): storage of the point information matrices of clouds created from now on
  void setOmegaStorage(int mode) { check(pwn_hip_ctx_set_omega_storage(_ctx, mode)); }
  void synchronize() { check(pwn_hip_ctx_synchronize(_ctx)); }
  void waitStream(void* hipStream) { check(pwn_hip_ctx_wait_stream(_ctx, hipStream)); }      // what the context queues from now on runs after that stream's work
  void signalStream(void* hipStream) { check(pwn_hip_ctx_signal_stream(_ctx, hipStream)); }  // what that stream gets from now on runs after everything the context has queued
  // fn(user) runs inside every alignment batch call, after its device work is queued and before the call waits (include/pwn_hip.h); nullptr = off
  void setEnqueuedCallback(void (*fn)(void*), void* user) { check(pwn_hip_ctx_set_enqueued_callback(_ctx, fn, user)); }
 private:
  pwn_hip_ctx* _ctx = nullptr;
};

// pwn_static.cpp:54-68 / :5-36
inline void DepthImage_convert_16UC1_to_32FC1(Context& ctx, DepthImage& dest, const RawDepthImage& src, float scale = 0.001f) {
  dest.create(src.rows, src.cols);
  ctx.check(pwn_hip_depth_u16_to_f32(ctx.handle(), src.data.data(), dest.data.data(), src.rows * src.cols, scale));
}
inline void DepthImage_scale(Context& ctx, DepthImage& dest, const DepthImage& src, int step, float maxDepthCov = 0.01f) {
  dest.create(src.rows / step, src.cols / step);
  ctx.check(pwn_hip_depth_scale(ctx.handle(), src.data.data(), src.rows, src.cols, step, maxDepthCov, dest.data.data()));
}

// cloud.h:20-187
class Cloud {
 public:
  Cloud(Context& ctx, int capacity) : _ctx(&ctx) { ctx.check(pwn_hip_cloud_create(ctx.handle(), capacity, &_h)); }
  ~Cloud() { pwn_hip_cloud_destroy(_ctx->handle(), _h); }
  Cloud(const Cloud&) = delete;
  Cloud& operator=(const Cloud&) = delete;
  pwn_hip_cloud* handle() const { return _h; }
  size_t size() const { int n = 0; _ctx->check(pwn_hip_cloud_size(_ctx->handle(), _h, &n)); return (size_t)n; }
  // n*4 floats each (Point w = 1, Normal w = 0)
  std::vector<float> points() const { std::vector<float> v(size() * 4); _ctx->check(pwn_hip_cloud_download(_ctx->handle(), _h, v.data(), nullptr, nullptr, nullptr, nullptr)); return v; }
  std::vector<float> normals() const { std::vector<float> v(size() * 4); _ctx->check(pwn_hip_cloud_download(_ctx->handle(), _h, nullptr, v.data(), nullptr, nullptr, nullptr)); return v; }
  std::vector<float> curvatures() const { std::vector<float> v(size()); _ctx->check(pwn_hip_cloud_download(_ctx->handle(), _h, nullptr, nullptr, v.data(), nullptr, nullptr)); return v; }
  std::vector<float> pointInformationMatrix() const { std::vector<float> v(size() * 16); _ctx->check(pwn_hip_cloud_download(_ctx->handle(), _h, nullptr, nullptr, nullptr, v.data(), nullptr)); return v; }
  std::vector<float> normalInformationMatrix() const { std::vector<float> v(size() * 16); _ctx->check(pwn_hip_cloud_download(_ctx->handle(), _h, nullptr, nullptr, nullptr, nullptr, v.data())); return v; }
  // n*16 floats: eigenvectors + mean as a column-major 4x4 per point (Stats, stats.h:13); needs a conversion that kept them (scene clouds)
  std::vector<float> stats() const { std::vector<float> v(size() * 16); _ctx->check(pwn_hip_cloud_download_stats(_ctx->handle(), _h, v.data(), nullptr, nullptr)); return v; }
  std::vector<int> traversabilityVector() const { return std::vector<int>(); }      // cloud.h:99: never written on this path
  size_t gaussians() const { return numGaussians(); }                              // cloud.h:131: the sensor-noise Gaussians live on the device (pwn_hip_cloud_download_gaussians)
  void transformInPlace(const Isometry3f& T) { _ctx->check(pwn_hip_cloud_transform_in_place(_ctx->handle(), _h, T.data())); }   // cloud.cpp:173-186
  // scene maintenance (cloud.cpp:11-171)
  void add(const Cloud& cloud, const Isometry3f& T = Isometry3f::Identity()) { _ctx->check(pwn_hip_cloud_add(_ctx->handle(), _h, cloud.handle(), T.data())); }
  bool save(const char* filename, const Isometry3f& T = Isometry3f::Identity(), int step = 1, bool binary = false) const {
    return pwn_hip_cloud_save(_ctx->handle(), _h, filename, T.data(), step, binary ? 1 : 0) == PWN_HIP_OK;
  }
  bool load(Isometry3f& T, const char* filename) { return pwn_hip_cloud_load(_ctx->handle(), _h, filename, T.data()) == PWN_HIP_OK; }
  // the cloud as one flat buffer, host or device (replication to the other GPUs of a node: pwn_closer.cpp:85-111; include/pwn_hip.h)
  size_t flatSize() const { size_t w = 0; _ctx->check(pwn_hip_cloud_export(_ctx->handle(), _h, nullptr, 0, &w)); return w; }
  size_t exportFlat(void* dst, size_t bytes) const { size_t w = 0; _ctx->check(pwn_hip_cloud_export(_ctx->handle(), _h, dst, bytes, &w)); return w; }
  void importFlat(const void* src, size_t bytes) { _ctx->check(pwn_hip_cloud_import(_ctx->handle(), _h, src, bytes)); }
  size_t numGaussians() const { int n = 0; _ctx->check(pwn_hip_cloud_num_gaussians(_ctx->handle(), _h, &n)); return (size_t)n; }
  Context* context() const { return _ctx; }
 private:
  Context* _ctx; pwn_hip_cloud* _h = nullptr;
};

// pointprojector.{h,cpp} + pinholepointprojector.{h,cpp}
class PinholePointProjector {
 public:
  PinholePointProjector() { _cameraMatrix(0,2) = 0.5f; _cameraMatrix(1,2) = 0.5f; }                      // pinholepointprojector.cpp:6-9
  const Matrix3f& cameraMatrix() const { return _cameraMatrix; }
  void setCameraMatrix(const Matrix3f& K) { _cameraMatrix = K; }
  const Isometry3f& transform() const { return _transform; }
  void setTransform(const Isometry3f& T) { _transform = T; }
  float minDistance() const { return _minDistance; }  void setMinDistance(float v) { _minDistance = v; }
  float maxDistance() const { return _maxDistance; }  void setMaxDistance(float v) { _maxDistance = v; }
  int imageRows() const { return _imageRows; }  int imageCols() const { return _imageCols; }
  void setImageSize(int r, int c) { _imageRows = r; _imageCols = c; }
  float baseline() const { return _baseline; }  void setBaseline(float v) { _baseline = v; }        // sensor-noise model of unProject's Gaussians
  float alpha() const { return _alpha; }        void setAlpha(float v) { _alpha = v; }
  void scale(float s) {                                                                                  // pinholepointprojector.cpp:149-154
    for (int c = 0; c < 3; ++c) { _cameraMatrix(0,c) *= s; _cameraMatrix(1,c) *= s; }
    _imageRows = (int)(_imageRows * s); _imageCols = (int)(_imageCols * s);
  }
  // unProject(points, indexImage, depthImage) (.cpp:68-91 / :93-133 without the Gaussians): fills the cloud's points with this projector's
  // transform, the other fields are reset to "invalid"
  Matrix3f inverseCameraMatrix() const { Matrix3f iK; float a[16], b[16]; pwn_hip_projector_matrices(_cameraMatrix.data(), _transform.data(), a, b, iK.data()); return iK; }   // :61
  // the single-point forms (pinholepointprojector.h:174,187,200): host code, the kernels' own expressions
  bool project(int& x, int& y, float& f, const float p[3]) const { return pwn_hip_project_point(_cameraMatrix.data(), _transform.data(), _minDistance, _maxDistance, p, &x, &y, &f) != 0; }
  bool unProject(float p[3], int x, int y, float d) const { return pwn_hip_unproject_pixel(_cameraMatrix.data(), _transform.data(), _minDistance, _maxDistance, x, y, d, p) != 0; }
  int projectInterval(int, int, float d, float worldRadius) const { return pwn_hip_project_interval(_cameraMatrix.data(), _minDistance, _maxDistance, d, worldRadius); }
  void unProject(Context& ctx, Cloud& cloud, IntImage& indexImage, const DepthImage& depthImage) const {
    indexImage.create(depthImage.rows, depthImage.cols);
    const pwn_hip_converter_params p = stageParams(0.1f);
    ctx.check(pwn_hip_unproject(ctx.handle(), &p, _transform.data(), depthImage.data.data(), depthImage.rows, depthImage.cols, cloud.handle(), indexImage.data.data()));
  }
  // projectIntervals(intervalImage, depthImage, worldRadius) (.cpp:135-147)
  void projectIntervals(Context& ctx, IntImage& intervalImage, const DepthImage& depthImage, float worldRadius) const {
    intervalImage.create(depthImage.rows, depthImage.cols);
    const pwn_hip_converter_params p = stageParams(worldRadius);
    ctx.check(pwn_hip_project_intervals(ctx.handle(), &p, depthImage.data.data(), depthImage.rows, depthImage.cols, intervalImage.data.data()));
  }
  void project(Context& ctx, IntImage& indexImage, DepthImage& depthImage, const Cloud& cloud) const {  // .cpp:33-66
    indexImage.create(_imageRows, _imageCols); depthImage.create(_imageRows, _imageCols);
    ctx.check(pwn_hip_project(ctx.handle(), _cameraMatrix.data(), _transform.data(), _minDistance, _maxDistance, _imageRows, _imageCols,
                              cloud.handle(), indexImage.data.data(), depthImage.data.data()));
  }
 private:
  pwn_hip_converter_params stageParams(float worldRadius) const {
    pwn_hip_converter_params p; pwn_hip_default_converter_params(&p);
    std::memcpy(p.K, _cameraMatrix.data(), sizeof(p.K));
    p.min_distance = _minDistance; p.max_distance = _maxDistance; p.world_radius = worldRadius;
    return p;
  }
  Matrix3f _cameraMatrix; Isometry3f _transform;
  float _minDistance = 0.01f, _maxDistance = 6.0f;                                                        // pointprojector.cpp:9-10
  float _baseline = 0.075f, _alpha = 0.1f;                                                                // pinholepointprojector.cpp:10-11
  int _imageRows = 0, _imageCols = 0;
};

// statscalculatorintegralimage.{h,cpp} (defaults .cpp:6-12)
class StatsCalculatorIntegralImage {
 public:
  float worldRadius() const { return _worldRadius; }            void setWorldRadius(float v) { _worldRadius = v; }
  int maxImageRadius() const { return _maxImageRadius; }        void setMaxImageRadius(int v) { _maxImageRadius = v; }
  int minImageRadius() const { return _minImageRadius; }        void setMinImageRadius(int v) { _minImageRadius = v; }
  int minPoints() const { return _minPoints; }                  void setMinPoints(int v) { _minPoints = v; }
  float curvatureThreshold() const { return _curvatureThreshold; } void setCurvatureThreshold(float v) { _curvatureThreshold = v; }
 private:
  float _worldRadius = 0.1f; int _maxImageRadius = 30, _minImageRadius = 10, _minPoints = 50; float _curvatureThreshold = 0.02f;
};
// informationmatrixcalculator.h:95-153
class InformationMatrixCalculator {
 public:
  InformationMatrixCalculator(float f0, float f1, float f2) { _flat[0] = f0; _flat[1] = f1; _flat[2] = f2; _nonFlat[0] = _nonFlat[1] = _nonFlat[2] = 1.f; }
  float curvatureThreshold() const { return _curvatureThreshold; } void setCurvatureThreshold(float v) { _curvatureThreshold = v; }
  const float* flatDiagonal() const { return _flat; }  const float* nonFlatDiagonal() const { return _nonFlat; }
  Matrix3f flatInformationMatrix() const { Matrix3f m; std::memset(m.m, 0, sizeof(m.m)); m(0,0) = _flat[0]; m(1,1) = _flat[1]; m(2,2) = _flat[2]; return m; }        // informationmatrixcalculator.h:58
  Matrix3f nonFlatInformationMatrix() const { Matrix3f m; std::memset(m.m, 0, sizeof(m.m)); m(0,0) = _nonFlat[0]; m(1,1) = _nonFlat[1]; m(2,2) = _nonFlat[2]; return m; }   // :74
  void setFlatInformationMatrix(float a, float b, float c) { _flat[0] = a; _flat[1] = b; _flat[2] = c; }
  void setNonFlatInformationMatrix(float a, float b, float c) { _nonFlat[0] = a; _nonFlat[1] = b; _nonFlat[2] = c; }
 protected:
  float _curvatureThreshold = 0.02f, _flat[3], _nonFlat[3];
};
struct PointInformationMatrixCalculator : InformationMatrixCalculator { PointInformationMatrixCalculator() : InformationMatrixCalculator(1000.f, 1.f, 1.f) {} };
struct NormalInformationMatrixCalculator : InformationMatrixCalculator { NormalInformationMatrixCalculator() : InformationMatrixCalculator(100.f, 100.f, 100.f) {} };

// depthimageconverter.{h,cpp} + depthimageconverterintegralimage.{h,cpp}
class DepthImageConverter {
 public:
  DepthImageConverter(Context* ctx, PinholePointProjector* projector, StatsCalculatorIntegralImage* statsCalculator,
                      PointInformationMatrixCalculator* pointInfo, NormalInformationMatrixCalculator* normalInfo)
      : _ctx(ctx), _projector(projector), _statsCalculator(statsCalculator), _pointInformationMatrixCalculator(pointInfo),
        _normalInformationMatrixCalculator(normalInfo) {}
  virtual ~DepthImageConverter() {}
  PinholePointProjector* projector() { return _projector; }                               void setProjector(PinholePointProjector* p) { _projector = p; }
  StatsCalculatorIntegralImage* statsCalculator() { return _statsCalculator; }            void setStatsCalculator(StatsCalculatorIntegralImage* s) { _statsCalculator = s; }      // depthimageconverter.h:63-104
  PointInformationMatrixCalculator* pointInformationMatrixCalculator() { return _pointInformationMatrixCalculator; }
  void setPointInformationMatrixCalculator(PointInformationMatrixCalculator* c) { _pointInformationMatrixCalculator = c; }
  NormalInformationMatrixCalculator* normalInformationMatrixCalculator() { return _normalInformationMatrixCalculator; }
  void setNormalInformationMatrixCalculator(NormalInformationMatrixCalculator* c) { _normalInformationMatrixCalculator = c; }
  IntImage& indexImage() { return _indexImage; }
  // The reference fills Cloud::gaussians() inside every compute() (depthimageconverterintegralimage.cpp:39); only Merger::merge reads
  // them, so here they are produced when asked for (clouds that will enter a scene).
  bool computeGaussians() const { return _computeGaussians; }  void setComputeGaussians(bool v) { _computeGaussians = v; }
  virtual void compute(Cloud& cloud, const DepthImage& depthImage, const Isometry3f& sensorOffset = Isometry3f::Identity()) = 0;
  pwn_hip_converter_params params(const Isometry3f& sensorOffset) const {
    if (!_projector || !_statsCalculator || !_pointInformationMatrixCalculator || !_normalInformationMatrixCalculator)
      throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "DepthImageConverter: missing collaborator");          // depthimageconverterintegralimage.cpp:18-21
    pwn_hip_converter_params p; pwn_hip_default_converter_params(&p);
    std::memcpy(p.K, _projector->cameraMatrix().data(), sizeof(p.K));
    p.min_distance = _projector->minDistance(); p.max_distance = _projector->maxDistance();
    p.world_radius = _statsCalculator->worldRadius(); p.min_image_radius = _statsCalculator->minImageRadius();
    p.max_image_radius = _statsCalculator->maxImageRadius(); p.min_points = _statsCalculator->minPoints();
    p.stats_curvature_threshold = _statsCalculator->curvatureThreshold();
    p.point_info_curvature_threshold = _pointInformationMatrixCalculator->curvatureThreshold();
    p.normal_info_curvature_threshold = _normalInformationMatrixCalculator->curvatureThreshold();
    for (int i = 0; i < 3; ++i) {
      p.point_flat_diag[i] = _pointInformationMatrixCalculator->flatDiagonal()[i]; p.point_nonflat_diag[i] = _pointInformationMatrixCalculator->nonFlatDiagonal()[i];
      p.normal_flat_diag[i] = _normalInformationMatrixCalculator->flatDiagonal()[i]; p.normal_nonflat_diag[i] = _normalInformationMatrixCalculator->nonFlatDiagonal()[i];
    }
    std::memcpy(p.sensor_offset, sensorOffset.data(), sizeof(p.sensor_offset));
    return p;
  }
 protected:
  Context* _ctx; PinholePointProjector* _projector; StatsCalculatorIntegralImage* _statsCalculator;
  PointInformationMatrixCalculator* _pointInformationMatrixCalculator; NormalInformationMatrixCalculator* _normalInformationMatrixCalculator;
  IntImage _indexImage;
  bool _computeGaussians = false;
};
class DepthImageConverterIntegralImage : public DepthImageConverter {
 public:
  using DepthImageConverter::DepthImageConverter;
  // n independent frames of equal size in one call (what a closure batch or a cache refill converts at once): depthFrames[i] -> *clouds[i].
  // The frame pointers may be host or device memory; the raw variant takes uint16 millimetre frames and fuses
  // DepthImage_convert_16UC1_to_32FC1(depthScale) in front.  Side effects on the projector as in compute().
  void computeBatch(const std::vector<Cloud*>& clouds, const std::vector<const float*>& depthFrames, int rows, int cols,
                    const Isometry3f& sensorOffset = Isometry3f::Identity()) {
    if (clouds.size() != depthFrames.size()) throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "computeBatch: list sizes differ");
    const pwn_hip_converter_params p = params(sensorOffset);
    _projector->setImageSize(rows, cols); _projector->setTransform(Isometry3f::Identity());
    std::vector<pwn_hip_cloud*> h(clouds.size());
    for (size_t i = 0; i < h.size(); ++i) h[i] = clouds[i]->handle();
    _ctx->check(pwn_hip_convert_batch(_ctx->handle(), &p, depthFrames.data(), (int)h.size(), rows, cols, h.data()));
  }
  void computeBatchRaw(const std::vector<Cloud*>& clouds, const std::vector<const uint16_t*>& rawFrames, float depthScale, int rows, int cols,
                       const Isometry3f& sensorOffset = Isometry3f::Identity()) {
    if (clouds.size() != rawFrames.size()) throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "computeBatchRaw: list sizes differ");
    const pwn_hip_converter_params p = params(sensorOffset);
    _projector->setImageSize(rows, cols); _projector->setTransform(Isometry3f::Identity());
    std::vector<pwn_hip_cloud*> h(clouds.size());
    for (size_t i = 0; i < h.size(); ++i) h[i] = clouds[i]->handle();
    _ctx->check(pwn_hip_convert_batch_u16(_ctx->handle(), &p, rawFrames.data(), depthScale, (int)h.size(), rows, cols, h.data()));
  }
  // the look-ahead of a sharded PwnCloser::processPartition (pwn_hip_convert_export_begin / _end): returns at once; the helper thread converts the raw
  // frame into `cloud` and then writes the cloud's flat form into flatDst (may be nullptr); computeExportEnd waits and returns the bytes written
  void computeExportBegin(Cloud& cloud, const uint16_t* rawFrame, float depthScale, int rows, int cols, void* flatDst, size_t flatBytes,
                          const Isometry3f& sensorOffset = Isometry3f::Identity()) {
    const pwn_hip_converter_params p = params(sensorOffset);
    _ctx->check(pwn_hip_convert_export_begin(_ctx->handle(), &p, rawFrame, depthScale, rows, cols, cloud.handle(), flatDst, flatBytes));
  }
  size_t computeExportEnd(Cloud& cloud, float* jobMs = nullptr) {
    size_t w = 0;
    _ctx->check(pwn_hip_convert_export_end(_ctx->handle(), cloud.handle(), &w, jobMs));
    return w;
  }
  void compute(Cloud& cloud, const DepthImage& depthImage, const Isometry3f& sensorOffset = Isometry3f::Identity()) override {
    if (depthImage.rows <= 0 || depthImage.cols <= 0) throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "DepthImageConverterIntegralImage: depthImage has zero size");
    const pwn_hip_converter_params p = params(sensorOffset);
    _projector->setImageSize(depthImage.rows, depthImage.cols);                 // depthimageconverterintegralimage.cpp:30
    _projector->setTransform(Isometry3f::Identity());                           // :38
    _indexImage.create(depthImage.rows, depthImage.cols);
    _ctx->check(pwn_hip_convert(_ctx->handle(), &p, depthImage.data.data(), depthImage.rows, depthImage.cols, cloud.handle(),
                                _indexImage.data.data(), nullptr, _computeGaussians ? 1 : 0));      // scene clouds keep their Stats (Cloud::save)
    if (_computeGaussians)
      _ctx->check(pwn_hip_cloud_gaussians(_ctx->handle(), &p, depthImage.data.data(), depthImage.rows, depthImage.cols, cloud.handle(),
                                          _projector->baseline(), _projector->alpha()));
  }
};

// merger.{h,cpp} (defaults merger.cpp:6-8)
class Merger {
 public:
  float distanceThreshold() const { return _distanceThreshold; }  void setDistanceThreshold(float v) { _distanceThreshold = v; }
  float normalThreshold() const { return _normalThreshold; }      void setNormalThreshold(float v) { _normalThreshold = v; }
  float maxPointDepth() const { return _maxPointDepth; }          void setMaxPointDepth(float v) { _maxPointDepth = v; }
  DepthImageConverter* depthImageConverter() const { return _depthImageConverter; }
  void setDepthImageConverter(DepthImageConverter* c) { _depthImageConverter = c; }
  void setImageSize(int r, int c) { _rows = r; _cols = c; }
  void imageSize(int& r, int& c) const { r = _rows; c = _cols; }                // merger.h: the size of the merger's index / depth images
  const std::vector<int>& collapsedIndices() const { return _collapsedIndices; }
  // needs the cloud's sensor-noise Gaussians (pwn_hip_cloud_gaussians after every convert that feeds the scene)
  void merge(Cloud* cloud, Isometry3f transform = Isometry3f::Identity()) {
    if (_rows <= 0 || _cols <= 0) throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "Merger: _indexImage has zero size");
    if (!_depthImageConverter || !_depthImageConverter->projector()) throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "Merger: missing _depthImageConverter / projector");
    PinholePointProjector* pp = _depthImageConverter->projector();
    pp->setTransform(transform);                                                 // merger.cpp:20-21
    _collapsedIndices.assign(cloud->size(), -1);
    int k = 0;
    Context* ctx = cloud->context();
    ctx->check(pwn_hip_merge(ctx->handle(), cloud->handle(), pp->cameraMatrix().data(), transform.data(), pp->minDistance(), pp->maxDistance(), _rows, _cols,
                             _distanceThreshold, _normalThreshold, _maxPointDepth, &k, _collapsedIndices.empty() ? nullptr : _collapsedIndices.data()));
  }
 protected:
  float _distanceThreshold = 0.1f, _normalThreshold = std::cos(10 * (float)M_PI / 180.0f), _maxPointDepth = 10.0f;
  DepthImageConverter* _depthImageConverter = nullptr;
  int _rows = 0, _cols = 0;
  std::vector<int> _collapsedIndices;
};

// voxelcalculator.{h,cpp}
class VoxelCalculator {
 public:
  float resolution() const { return _resolution; }  void setResolution(float r) { _resolution = r; }
  void compute(Cloud& cloud, float res) { const float old = _resolution; _resolution = res; compute(cloud); _resolution = old; }
  void compute(Cloud& cloud) { int k = 0; Context* ctx = cloud.context(); ctx->check(pwn_hip_voxelize(ctx->handle(), cloud.handle(), _resolution, &k, nullptr)); }
 protected:
  float _resolution = 0.01f;
};

// correspondencefinder.{h,cpp} (defaults .cpp:9-18)
class CorrespondenceFinder {
 public:
  float inlierDistanceThreshold() const { return _inlierDistanceThreshold; }            void setInlierDistanceThreshold(float v) { _inlierDistanceThreshold = v; }
  float inlierNormalAngularThreshold() const { return _inlierNormalAngularThreshold; }  void setInlierNormalAngularThreshold(float v) { _inlierNormalAngularThreshold = v; }
  float flatCurvatureThreshold() const { return _flatCurvatureThreshold; }              void setFlatCurvatureThreshold(float v) { _flatCurvatureThreshold = v; }
  float inlierCurvatureRatioThreshold() const { return _inlierCurvatureRatioThreshold; } void setInlierCurvatureRatioThreshold(float v) { _inlierCurvatureRatioThreshold = v; }
  float squaredThreshold() const { return _inlierDistanceThreshold * _inlierDistanceThreshold; }                          // correspondencefinder.h:133
  void setImageSize(int r, int c) { _rows = r; _cols = c; }
  int imageRows() const { return _rows; }  int imageCols() const { return _cols; }
  int numCorrespondences() const { return _numCorrespondences; }
  // (referenceIndex, currentIndex) pairs of the last Aligner::computeCorrespondences: the first numCorrespondences() are valid, in
  // row-major pixel order, the rest are (-1,-1) (correspondencefinder.cpp:116-117)
  const std::vector<int>& correspondences() const { return _correspondences; }
  int numCandidates() const { return _numCandidates; }
  IntImage& referenceIndexImage() { return _referenceIndexImage; }  IntImage& currentIndexImage() { return _currentIndexImage; }
  DepthImage& referenceDepthImage() { return _referenceDepthImage; }  DepthImage& currentDepthImage() { return _currentDepthImage; }
 private:
  friend class Aligner;
  float _inlierDistanceThreshold = 0.5f, _inlierNormalAngularThreshold = (float)std::cos(M_PI / 6), _flatCurvatureThreshold = 0.02f,
        _inlierCurvatureRatioThreshold = 1.3f;
  int _rows = 0, _cols = 0, _numCorrespondences = 0, _numCandidates = 0;
  std::vector<int> _correspondences;
  IntImage _referenceIndexImage, _currentIndexImage; DepthImage _referenceDepthImage, _currentDepthImage;
};
class Aligner;
// linearizer.{h,cpp} (defaults .cpp:9-15)
class Linearizer {
 public:
  void setAligner(Aligner* a) { _aligner = a; }  Aligner* aligner() { return _aligner; }
  Isometry3f T() const { return _T; }  void setT(Isometry3f T) { T.forceLastRow(); _T = T; }                         // linearizer.h:55,62-65
  float inlierMaxChi2() const { return _inlierMaxChi2; }  void setInlierMaxChi2(float v) { _inlierMaxChi2 = v; }
  bool robustKernel() const { return _robustKernel; }      void setRobustKernel(bool v) { _robustKernel = v; }
  float error() const { return _error; }  int inliers() const { return _inliers; }
  const Matrix6f& H() const { return _H; }  const Vector6f& b() const { return _b; }       // filled by Aligner::align when statistics are computed
 private:
  friend class Aligner;
  Aligner* _aligner = nullptr; float _inlierMaxChi2 = 9e3f; bool _robustKernel = true; float _error = 0.f; int _inliers = 0;
  Matrix6f _H; Vector6f _b; Isometry3f _T;
};

// aligner.{h,cpp}
class Aligner {
 public:
  explicit Aligner(Context* ctx) : _ctx(ctx) {}
  virtual ~Aligner() {}
  void setProjector(PinholePointProjector* p) { _projector = p; }            PinholePointProjector* projector() { return _projector; }
  void setLinearizer(Linearizer* l) { _linearizer = l; if (l) l->setAligner(this); }  Linearizer* linearizer() { return _linearizer; }
  void setCorrespondenceFinder(CorrespondenceFinder* f) { _correspondenceFinder = f; } CorrespondenceFinder* correspondenceFinder() { return _correspondenceFinder; }
  Cloud* referenceCloud() { return _referenceCloud; }  Cloud* currentCloud() { return _currentCloud; }
  const Isometry3f& initialGuess() const { return _initialGuess; }
  const Isometry3f& sensorOffset() const { return _referenceSensorOffset; }                                           // aligner.h:141
  const Isometry3f& referenceSensorOffset() const { return _referenceSensorOffset; }  const Isometry3f& currentSensorOffset() const { return _currentSensorOffset; }
  // aligner.h:216-239: _debug only switches the reference's terminal output on, _minInliers is never read (aligner.cpp:28): inert state, kept for configuration code
  bool debug() const { return _debug; }  void setDebug(bool v) { _debug = v; }  int minInliers() const { return _minInliers; }  void setMinInliers(int v) { _minInliers = v; }
  void setReferenceCloud(Cloud* c) { _referenceCloud = c; clearPriors(); }                                           // aligner.h:60-63: setting a cloud clears the priors
  void setCurrentCloud(Cloud* c) { _currentCloud = c; clearPriors(); }                                               // aligner.h:77-80
  // aligner.cpp:34-47, se3_prior.h
  void addRelativePrior(const Isometry3f& mean, const Matrix6f& informationMatrix) { _priors.push_back(makePrior(0, mean, Isometry3f::Identity(), informationMatrix)); }
  void addAbsolutePrior(const Isometry3f& referenceTransform, const Isometry3f& mean, const Matrix6f& informationMatrix) {
    _priors.push_back(makePrior(1, mean, referenceTransform, informationMatrix));
  }
  void clearPriors() { _priors.clear(); }
  size_t numPriors() const { return _priors.size(); }
  // Aligner::_computeStatistics (aligner.cpp:127,152-199) costs one more linearizer pass and 6x6 host math; the reference always runs it,
  // here it is run when asked for (the tracker and the closer never read its outputs)
  bool computeStatistics() const { return _computeStatistics; }  void setComputeStatistics(bool v) { _computeStatistics = v; }
  const Matrix6f& omega() const { return _omega; }                                                                    // aligner.h:314
  const Vector6f& mean() const { return _mean; }
  float translationalEigenRatio() const { return _translationalEigenRatio; }  float rotationalEigenRatio() const { return _rotationalEigenRatio; }
  float translationalMinEigenRatio() const { return _translationalMinEigenRatio; }  void setTranslationalMinEigenRatio(float v) { _translationalMinEigenRatio = v; }
  float rotationalMinEigenRatio() const { return _rotationalMinEigenRatio; }        void setRotationalMinEigenRatio(float v) { _rotationalMinEigenRatio = v; }
  bool solutionValid() const {                                                                                         // aligner.cpp:128-129
    return !(_rotationalEigenRatio > _rotationalMinEigenRatio || _translationalEigenRatio > _translationalMinEigenRatio);
  }
  int outerIterations() const { return _outerIterations; }  void setOuterIterations(int n) { _outerIterations = n; }
  int innerIterations() const { return _innerIterations; }  void setInnerIterations(int n) { _innerIterations = n; }
  const Isometry3f& T() const { return _T; }
  void setInitialGuess(Isometry3f g) { g.forceLastRow(); _initialGuess = g; }                                        // aligner.h:130-133
  void setSensorOffset(Isometry3f o) { o.forceLastRow(); _referenceSensorOffset = o; _currentSensorOffset = o; }     // :149-153
  void setReferenceSensorOffset(Isometry3f o) { o.forceLastRow(); _referenceSensorOffset = o; }
  void setCurrentSensorOffset(Isometry3f o) { o.forceLastRow(); _currentSensorOffset = o; }
  float error() const { return _error; }  int inliers() const { return _inliers; }  double totalTime() const { return _totalTime; }
  const pwn_hip_align_result& result() const { return _result; }
  pwn_hip_aligner_params params() const {
    if (!_projector || !_linearizer || !_correspondenceFinder) throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "Aligner: missing collaborator");   // aligner.cpp:50-52
    pwn_hip_aligner_params p; pwn_hip_default_aligner_params(&p);
    std::memcpy(p.K, _projector->cameraMatrix().data(), sizeof(p.K));
    p.min_distance = _projector->minDistance(); p.max_distance = _projector->maxDistance();
    p.rows = _projector->imageRows(); p.cols = _projector->imageCols();
    p.inlier_distance_threshold = _correspondenceFinder->inlierDistanceThreshold();
    p.inlier_normal_angular_threshold = _correspondenceFinder->inlierNormalAngularThreshold();
    p.flat_curvature_threshold = _correspondenceFinder->flatCurvatureThreshold();
    p.inlier_curvature_ratio_threshold = _correspondenceFinder->inlierCurvatureRatioThreshold();
    p.inlier_max_chi2 = _linearizer->inlierMaxChi2(); p.robust_kernel = _linearizer->robustKernel() ? 1 : 0;
    p.outer_iterations = _outerIterations; p.inner_iterations = _innerIterations;
    std::memcpy(p.reference_sensor_offset, _referenceSensorOffset.data(), 64);
    std::memcpy(p.current_sensor_offset, _currentSensorOffset.data(), 64);
    std::memcpy(p.initial_guess, _initialGuess.data(), 64);
    return p;
  }
  // aligner.cpp:49-125 ; fetchImages: also fill the finder's index / depth images (read by matchClouds)
  virtual void align(bool fetchImages = false) {
    if (!_referenceCloud || !_currentCloud) throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "Aligner: missing cloud");
    const pwn_hip_aligner_params p = params();
    pwn_hip_align_statistics q;
    if (!_priors.empty()) {                                                       // aligner.cpp:96-108: host-driven loop with the prior terms;
      _ctx->check(pwn_hip_align_with_priors_ex(_ctx->handle(), &p, _referenceCloud->handle(), _currentCloud->handle(), (int)_priors.size(), _priors.data(),
                                               &_result, _computeStatistics ? &q : nullptr));     // _computeStatistics runs after it as always (:127)
    } else if (_computeStatistics) {
      pwn_hip_cloud* r = _referenceCloud->handle(); pwn_hip_cloud* c = _currentCloud->handle();
      _ctx->check(pwn_hip_align_batch_ex(_ctx->handle(), &p, 1, &r, &c, nullptr, &_result, 0.f, nullptr, &q));
    } else {
      _ctx->check(pwn_hip_align(_ctx->handle(), &p, _referenceCloud->handle(), _currentCloud->handle(), &_result));
    }
    if (_computeStatistics) {
      std::memcpy(_omega.m, q.omega, sizeof(q.omega)); std::memcpy(_mean.v, q.mean, sizeof(q.mean));
      _translationalEigenRatio = q.translational_eigen_ratio; _rotationalEigenRatio = q.rotational_eigen_ratio;
      std::memcpy(_linearizer->_H.m, q.H, sizeof(q.H)); std::memcpy(_linearizer->_b.v, q.b, sizeof(q.b));
    }
    _T = Isometry3f(_result.T); _error = _result.error; _inliers = _result.inliers; _totalTime = _result.total_time_ms;
    _linearizer->_error = _error; _linearizer->_inliers = _inliers;
    _linearizer->setT(_T.inverse());      // what the reference leaves there: _linearizer->setT(_T.inverse()) in _computeStatistics (aligner.cpp:165-167)
    _correspondenceFinder->_numCorrespondences = _result.iterations > 0 ? _result.iter_correspondences[_result.iterations - 1] : 0;
    if (fetchImages) {
      CorrespondenceFinder& f = *_correspondenceFinder;
      f._referenceIndexImage.create(p.rows, p.cols); f._currentIndexImage.create(p.rows, p.cols);
      f._referenceDepthImage.create(p.rows, p.cols); f._currentDepthImage.create(p.rows, p.cols);
      _ctx->check(pwn_hip_align_images(_ctx->handle(), f._referenceIndexImage.data.data(), f._referenceDepthImage.data.data(),
                                       f._currentIndexImage.data.data(), f._currentDepthImage.data.data()));
    }
  }
  // n independent alignments with this aligner's parameters (the candidate loop of PwnCloser::processPartition, pwn_closer.cpp:92-111);
  // initialGuesses: n isometries or empty (= this aligner's initial guess for all); scores: optional depth-agreement scores of matchClouds
  std::vector<pwn_hip_align_result> alignBatch(const std::vector<Cloud*>& references, const std::vector<Cloud*>& currents,
                                               const std::vector<Isometry3f>& initialGuesses = std::vector<Isometry3f>(),
                                               std::vector<pwn_hip_match_result>* scores = nullptr, float frameInlierDepthThreshold = 50.f) {
    const size_t n = references.size();
    if (currents.size() != n || (!initialGuesses.empty() && initialGuesses.size() != n)) throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "Aligner::alignBatch: list sizes differ");
    std::vector<pwn_hip_align_result> results(n);
    if (n == 0) return results;
    const pwn_hip_aligner_params p = params();
    std::vector<pwn_hip_cloud*> r(n), c(n);
    for (size_t i = 0; i < n; ++i) {
      if (!references[i] || !currents[i]) throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "Aligner::alignBatch: null cloud");
      r[i] = references[i]->handle(); c[i] = currents[i]->handle();
    }
    std::vector<float> g;
    for (const Isometry3f& T : initialGuesses) { Isometry3f t = T; t.forceLastRow(); g.insert(g.end(), t.m, t.m + 16); }
    if (scores) scores->resize(n);
    _ctx->check(pwn_hip_align_batch_ex(_ctx->handle(), &p, (int)n, r.data(), c.data(), g.empty() ? nullptr : g.data(), results.data(),
                                       frameInlierDepthThreshold, scores ? scores->data() : nullptr, nullptr));
    return results;
  }
  // One candidate batch from raw uint16 frames as one submission (pwn_hip_convert_align_batch_u16): per pair PwnMatcherBase::makeCloud of both
  // frames (pwn_matcher_base.cpp:77-85), then align; `records` (optional): device or host buffer of n * PWN_HIP_RECORD_FLOATS floats that
  // receives the fixed-size result records (pair id = firstPairId + i) straight from the device.
  std::vector<pwn_hip_align_result> convertAlignBatch(DepthImageConverter& converter, const std::vector<Cloud*>& references, const std::vector<Cloud*>& currents,
                                                      const std::vector<const uint16_t*>& refFrames, const std::vector<const uint16_t*>& curFrames,
                                                      float depthScale, int rows, int cols, float* records = nullptr, int firstPairId = 0) {
    const size_t n = references.size();
    if (currents.size() != n || refFrames.size() != n || curFrames.size() != n) throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "Aligner::convertAlignBatch: list sizes differ");
    std::vector<pwn_hip_align_result> results(n);
    if (n == 0) return results;
    const pwn_hip_converter_params cp = converter.params(Isometry3f::Identity());
    converter.projector()->setImageSize(rows, cols); converter.projector()->setTransform(Isometry3f::Identity());      // side effects of compute() (depthimageconverterintegralimage.cpp:30,38)
    const pwn_hip_aligner_params p = params();
    std::vector<pwn_hip_cloud*> r(n), c(n);
    for (size_t i = 0; i < n; ++i) {
      if (!references[i] || !currents[i]) throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "Aligner::convertAlignBatch: null cloud");
      r[i] = references[i]->handle(); c[i] = currents[i]->handle();
    }
    _ctx->check(pwn_hip_convert_align_batch_u16(_ctx->handle(), &cp, &p, (int)n, refFrames.data(), curFrames.data(), depthScale, rows, cols, r.data(), c.data(),
                                                nullptr, nullptr, firstPairId, results.data(), records));
    return results;
  }
  // stage-level entry points with explicit inputs: CorrespondenceFinder::compute(reference, current, T) on two index images
  // (correspondencefinder.cpp:20-118) and Linearizer::update() with _T = T on the finder's correspondences (linearizer.cpp:17-115)
  void computeCorrespondences(const IntImage& referenceIndexImage, const IntImage& currentIndexImage, const Isometry3f& T) {
    if (!_referenceCloud || !_currentCloud) throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "Aligner: missing cloud");
    const pwn_hip_aligner_params p = params();
    CorrespondenceFinder& f = *_correspondenceFinder;
    f._correspondences.assign((size_t)p.rows * p.cols * 2, -1);
    _ctx->check(pwn_hip_correspondences(_ctx->handle(), &p, _referenceCloud->handle(), _currentCloud->handle(), referenceIndexImage.data.data(),
                                        currentIndexImage.data.data(), T.data(), f._correspondences.data(), &f._numCorrespondences, &f._numCandidates));
  }
  void linearize(const Isometry3f& T) {
    if (!_referenceCloud || !_currentCloud) throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "Aligner: missing cloud");
    const pwn_hip_aligner_params p = params();
    CorrespondenceFinder& f = *_correspondenceFinder;
    _ctx->check(pwn_hip_linearize(_ctx->handle(), &p, _referenceCloud->handle(), _currentCloud->handle(), f._correspondences.data(), f._numCorrespondences,
                                  T.data(), _linearizer->_H.m, _linearizer->_b.v, &_linearizer->_error, &_linearizer->_inliers));
  }
  Context* context() const { return _ctx; }
 protected:
  static pwn_hip_prior makePrior(int kind, const Isometry3f& mean, const Isometry3f& referenceTransform, const Matrix6f& info) {
    pwn_hip_prior q; q.kind = kind;
    std::memcpy(q.mean, mean.data(), sizeof(q.mean)); std::memcpy(q.reference_transform, referenceTransform.data(), sizeof(q.reference_transform));
    std::memcpy(q.information, info.data(), sizeof(q.information));
    return q;
  }
  Context* _ctx;
  PinholePointProjector* _projector = nullptr; Linearizer* _linearizer = nullptr; CorrespondenceFinder* _correspondenceFinder = nullptr;
  Cloud* _referenceCloud = nullptr; Cloud* _currentCloud = nullptr; bool _debug = false; int _minInliers = 100;
  std::vector<pwn_hip_prior> _priors;
  bool _computeStatistics = false;
  Matrix6f _omega; Vector6f _mean;
  float _translationalEigenRatio = 3.402823466e+38f, _rotationalEigenRatio = 3.402823466e+38f;
  float _translationalMinEigenRatio = 50.f, _rotationalMinEigenRatio = 50.f;                                          // aligner.cpp:29-30
  int _outerIterations = 10, _innerIterations = 1;                                                                    // aligner.cpp:19-20
  Isometry3f _T, _initialGuess, _referenceSensorOffset, _currentSensorOffset;
  float _error = 0.f; int _inliers = 0; double _totalTime = 0.0;
  pwn_hip_align_result _result;
};

// pwn_tracker/pwn_matcher_base.{h,cpp}: makeCloud + matchClouds with the reference's quirks kept (guess z zeroed :114,
// projector re-configured and scaled per call :117-119, information matrix = 100*I :147-148).
struct PwnMatcherBase {
  struct MatcherResult {
    double transform[16];            // column-major, Aligner::T() widened to double (convertScalar, .h:66-71)
    double informationMatrix[36];
    int cloud_inliers, image_nonZeros, image_outliers, image_inliers;
    float image_reprojectionDistance;
  };
  PwnMatcherBase(Context* ctx, Aligner* aligner, DepthImageConverter* converter) : _ctx(ctx), _aligner(aligner), _converter(converter) {}
  virtual ~PwnMatcherBase() { try { makeCloudDrop(); } catch (...) {} }
  int scale() const { return _scale; }  void setScale(int s) { _scale = s; }
  float frameInlierDepthThreshold() const { return _frameInlierDepthThreshold; }  void setFrameInlierDepthThreshold(float v) { _frameInlierDepthThreshold = v; }
  Aligner* aligner() { return _aligner; }  DepthImageConverter* converter() { return _converter; }
  void setAligner(Aligner* a) { _aligner = a; }  void setConverter(DepthImageConverter* c) { _converter = c; }      // pwn_matcher_base.h:30,33

  // .cpp:57-86: returns a new Cloud owned by the caller; r, c, cameraMatrix receive the scaled values.  DepthImage_scale (:72) and
  // converter->compute (:79) run as one device-side call (no scaled image on the host); same side effects on the converter's projector.
  Cloud* makeCloud(int& r, int& c, Matrix3f& cameraMatrix, const Isometry3f& sensorOffset, const DepthImage& depthImage) {
    PinholePointProjector* projector = _converter->projector();
    const float invScale = 1.0f / _scale;
    Matrix3f scaled = cameraMatrix;
    for (int i = 0; i < 9; ++i) scaled.m[i] = scaled.m[i] * invScale;
    scaled(2,2) = 1.0f;
    projector->setCameraMatrix(scaled);
    projector->setImageSize(depthImage.rows / _scale, depthImage.cols / _scale);
    projector->setTransform(Isometry3f::Identity());
    cameraMatrix = projector->cameraMatrix(); r = projector->imageRows(); c = projector->imageCols();
    Cloud* cloud = new Cloud(*_ctx, r * c > 0 ? r * c : 1);
    const pwn_hip_converter_params p = _converter->params(sensorOffset);
    const int rc = pwn_hip_convert_scaled(_ctx->handle(), &p, depthImage.data.data(), depthImage.rows, depthImage.cols, _scale, 0.01f, cloud->handle());
    if (rc) { delete cloud; _ctx->check(rc); }
    ++numCalls;
    return cloud;
  }
  // makeCloud in two halves (pwn_hip_convert_scaled_begin / _end; not in the reference): makeCloudBegin returns at once, the frame is
  // converted by the library's helper thread next to whatever runs on the context meanwhile; makeCloudEnd returns the cloud makeCloud would
  // have returned, bit for bit.  One at a time per context; depthImage must stay alive and unchanged in between.
  void makeCloudBegin(Matrix3f cameraMatrix, const Isometry3f& sensorOffset, const DepthImage& depthImage) {
    makeCloudDrop();
    PinholePointProjector* projector = _converter->projector();
    const float invScale = 1.0f / _scale;
    for (int i = 0; i < 9; ++i) cameraMatrix.m[i] = cameraMatrix.m[i] * invScale;
    cameraMatrix(2,2) = 1.0f;
    projector->setCameraMatrix(cameraMatrix);
    projector->setImageSize(depthImage.rows / _scale, depthImage.cols / _scale);
    projector->setTransform(Isometry3f::Identity());
    _pendingK = projector->cameraMatrix(); _pendingR = projector->imageRows(); _pendingC = projector->imageCols();
    Cloud* cloud = new Cloud(*_ctx, _pendingR * _pendingC > 0 ? _pendingR * _pendingC : 1);
    const pwn_hip_converter_params p = _converter->params(sensorOffset);
    const int rc = pwn_hip_convert_scaled_begin(_ctx->handle(), &p, depthImage.data.data(), depthImage.rows, depthImage.cols, _scale, 0.01f, cloud->handle());
    if (rc) { delete cloud; _ctx->check(rc); }
    _pending = cloud; _pendingImage = &depthImage;
  }
  bool makeCloudPending(const DepthImage& depthImage) const { return _pending && _pendingImage == &depthImage; }
  Cloud* makeCloudEnd(int& r, int& c, Matrix3f& cameraMatrix) {
    if (!_pending) throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "makeCloudEnd without makeCloudBegin");
    Cloud* cloud = _pending; _pending = nullptr; _pendingImage = nullptr;
    const int rc = pwn_hip_convert_end(_ctx->handle(), cloud->handle());
    if (rc) { delete cloud; _ctx->check(rc); }
    r = _pendingR; c = _pendingC; cameraMatrix = _pendingK;
    ++numCalls;
    return cloud;
  }
  void makeCloudDrop() { if (_pending) { int r, c; Matrix3f K; delete makeCloudEnd(r, c, K); } }
  // .cpp:88-183
  void matchClouds(MatcherResult& result, Cloud* fromCloud, Cloud* toCloud, const Isometry3f& fromOffset, const Isometry3f& toOffset,
                   const Matrix3f& toCameraMatrix, int toRows, int toCols, const Isometry3f& initialGuess = Isometry3f::Identity()) {
    configure(fromOffset, toOffset, toCameraMatrix, toRows, toCols, initialGuess);
    _aligner->setReferenceCloud(fromCloud);
    _aligner->setCurrentCloud(toCloud);
    _aligner->align();
    pwn_hip_match_result m;
    _ctx->check(pwn_hip_match_score(_ctx->handle(), _frameInlierDepthThreshold, &m));
    fill(result, _aligner->result(), m);
  }
  // the candidate loop of PwnCloser::processPartition (pwn_closer.cpp:92-111) as one batched call: results[i] = matchClouds(from[i], to[i])
  void matchCloudsBatch(std::vector<MatcherResult>& results, const std::vector<Cloud*>& fromClouds, const std::vector<Cloud*>& toClouds,
                        const Isometry3f& fromOffset, const Isometry3f& toOffset, const Matrix3f& toCameraMatrix, int toRows, int toCols,
                        const std::vector<Isometry3f>& initialGuesses = std::vector<Isometry3f>()) {
    configure(fromOffset, toOffset, toCameraMatrix, toRows, toCols, Isometry3f::Identity());
    std::vector<Isometry3f> g(fromClouds.size());
    for (size_t i = 0; i < g.size(); ++i) { if (!initialGuesses.empty()) g[i] = initialGuesses[i]; g[i](2,3) = 0.f; g[i].forceLastRow(); }
    std::vector<pwn_hip_match_result> scores;
    const std::vector<pwn_hip_align_result> r = _aligner->alignBatch(fromClouds, toClouds, g, &scores, _frameInlierDepthThreshold);
    results.resize(r.size());
    for (size_t i = 0; i < r.size(); ++i) fill(results[i], r[i], scores[i]);
  }
  // the same call with the results (also) leaving as PWN_HIP_MATCH_RECORD_FLOATS-float records, device or host: what the ranks of a sharded
  // processPartition exchange.  pairIds may be empty (record i then carries firstPairId + i).
  void matchCloudsBatchRecords(float* records, const std::vector<Cloud*>& fromClouds, const std::vector<Cloud*>& toClouds,
                               const Isometry3f& fromOffset, const Isometry3f& toOffset, const Matrix3f& toCameraMatrix, int toRows, int toCols,
                               const std::vector<Isometry3f>& initialGuesses = std::vector<Isometry3f>(), const std::vector<int>& pairIds = std::vector<int>(),
                               int firstPairId = 0, std::vector<MatcherResult>* results = nullptr) {
    configure(fromOffset, toOffset, toCameraMatrix, toRows, toCols, Isometry3f::Identity());
    const int n = (int)fromClouds.size();
    std::vector<float> g((size_t)n * 16);
    for (int i = 0; i < n; ++i) {
      Isometry3f t; if (!initialGuesses.empty()) t = initialGuesses[(size_t)i];
      t(2,3) = 0.f; t.forceLastRow();
      std::memcpy(&g[(size_t)i * 16], t.data(), 16 * sizeof(float));
    }
    std::vector<pwn_hip_cloud*> r((size_t)n), c((size_t)n);
    for (int i = 0; i < n; ++i) { r[(size_t)i] = fromClouds[(size_t)i]->handle(); c[(size_t)i] = toClouds[(size_t)i]->handle(); }
    std::vector<pwn_hip_align_result> res(results ? (size_t)n : 0);
    std::vector<pwn_hip_match_result> sc(results ? (size_t)n : 0);
    const pwn_hip_aligner_params p = _aligner->params();
    _ctx->check(pwn_hip_match_batch_records(_ctx->handle(), &p, n, r.data(), c.data(), g.data(), _frameInlierDepthThreshold,
                                            pairIds.empty() ? nullptr : pairIds.data(), firstPairId, results ? res.data() : nullptr,
                                            results ? sc.data() : nullptr, records));
    if (results) { results->resize((size_t)n); for (int i = 0; i < n; ++i) fill((*results)[(size_t)i], res[(size_t)i], sc[(size_t)i]); }
  }
  int numCalls = 0;
 protected:
  void configure(const Isometry3f& fromOffset, const Isometry3f& toOffset, const Matrix3f& toCameraMatrix, int toRows, int toCols, const Isometry3f& initialGuess) {
    PinholePointProjector* projector = _aligner->projector();
    _aligner->setReferenceSensorOffset(fromOffset);
    _aligner->setCurrentSensorOffset(toOffset);
    Isometry3f ig = initialGuess;
    ig(2,3) = 0.f;                                                               // :114
    _aligner->setInitialGuess(ig);
    projector->setCameraMatrix(toCameraMatrix);
    projector->setImageSize(toRows, toCols);
    projector->scale((float)(1. / _scale));                                      // :117-119
    _aligner->correspondenceFinder()->setImageSize(projector->imageRows(), projector->imageCols());
  }
  static void fill(MatcherResult& result, const pwn_hip_align_result& a, const pwn_hip_match_result& m) {
    for (int i = 0; i < 16; ++i) result.transform[i] = a.T[i];
    for (int i = 0; i < 36; ++i) result.informationMatrix[i] = (i % 7 == 0) ? 100.0 : 0.0;     // :147-148 HACK kept
    result.cloud_inliers = a.inliers;
    result.image_reprojectionDistance = m.image_reprojection_distance;
    result.image_nonZeros = m.image_non_zeros; result.image_outliers = m.image_outliers; result.image_inliers = m.image_inliers;
  }
  Context* _ctx; Aligner* _aligner; DepthImageConverter* _converter;
  Cloud* _pending = nullptr; const DepthImage* _pendingImage = nullptr; Matrix3f _pendingK; int _pendingR = 0, _pendingC = 0;      // makeCloudBegin
  float _frameInlierDepthThreshold = 50.f;   // .cpp:13
  int _scale = 2;                            // .cpp:12
};

// acceptance rule of PwnCloser::matchFrames (pwn_tracker/pwn_closer.cpp:56-58,138-141)
struct PwnCloserAcceptance {
  int frameMinNonZeroThreshold = 3000, frameMaxOutliersThreshold = 100, frameMinInliersThreshold = 1000;
  bool accept(const PwnMatcherBase::MatcherResult& r) const {
    return !(r.image_nonZeros < frameMinNonZeroThreshold || r.image_outliers > frameMaxOutliersThreshold || r.image_inliers < frameMinInliersThreshold);
  }
};

// pwn_tracker/pwn_tracker_cache.cpp:24-51 (+ boss_map_building cache.hpp): device-resident LRU of clouds keyed by frame; a miss re-runs the
// converter on the frame's stored depth image (PwnCache::loadFrame), so closure batches do not re-convert frames still resident in HBM.
class CloudCache {
 public:
  CloudCache(PwnMatcherBase* matcher, size_t capacity = 64) : _matcher(matcher), _capacity(capacity) {}
  ~CloudCache() { for (auto& e : _lru) delete e.second; }
  // the cache takes ownership of `cloud` (may be nullptr: the cloud is then made on the first get)
  void addFrame(int key, const DepthImage& depthImage, const Matrix3f& cameraMatrix, const Isometry3f& sensorOffset, Cloud* cloud = nullptr) {
    _frames[key] = Frame{ depthImage, cameraMatrix, sensorOffset };
    if (cloud) insert(key, cloud);
  }
  Cloud* get(int key) {
    for (auto it = _lru.begin(); it != _lru.end(); ++it)
      if (it->first == key) { ++hits; _lru.splice(_lru.end(), _lru, it); return _lru.back().second; }
    ++misses;
    auto f = _frames.find(key);
    if (f == _frames.end()) throw Error(PWN_HIP_ERR_INVALID_ARGUMENT, "CloudCache: unknown frame");
    int r, c; Matrix3f K = f->second.cameraMatrix;
    Cloud* cloud = _matcher->makeCloud(r, c, K, f->second.sensorOffset, f->second.depth);      // pwn_tracker_cache.cpp:38-44
    insert(key, cloud);
    return cloud;
  }
  size_t resident() const { return _lru.size(); }
  int hits = 0, misses = 0;
 private:
  struct Frame { DepthImage depth; Matrix3f cameraMatrix; Isometry3f sensorOffset; };
  void insert(int key, Cloud* cloud) {
    for (auto it = _lru.begin(); it != _lru.end(); ++it) if (it->first == key) { delete it->second; _lru.erase(it); break; }
    _lru.emplace_back(key, cloud);
    while (_lru.size() > _capacity) { delete _lru.front().second; _lru.pop_front(); }          // least recently used
  }
  PwnMatcherBase* _matcher; size_t _capacity;
  std::list<std::pair<int, Cloud*> > _lru;       // front = least recently used
  std::map<int, Frame> _frames;
};

// pwn_tracker/pwn_tracker.{h,cpp}: sequential odometry with key-cloud switching (processFrame, .cpp:106-215).  The BOSS map bookkeeping of
// the reference (frames / relations handed to the map manager, :217-281) is reported through FrameResult instead.
class PwnTracker : public PwnMatcherBase {
 public:
  struct FrameResult { bool newFrame = false, aligned = false; int inliers = 0; float error = 0.f, inliersFraction = 0.f; Isometry3f T, globalT; };
  using PwnMatcherBase::PwnMatcherBase;
  ~PwnTracker() { delete _previousCloud; if (_currentCloud != _previousCloud) delete _currentCloud; }
  const Isometry3f& globalT() const { return _globalT; }
  int numKeyframes() const { return _numKeyframes; }
  float newFrameInliersFraction() const { return _newFrameInliersFraction; }  void setNewFrameInliersFraction(float v) { _newFrameInliersFraction = v; }
  void init() {                                                                  // pwn_tracker.cpp:38-49
    makeCloudDrop();
    delete _previousCloud; if (_currentCloud != _previousCloud) delete _currentCloud;
    _previousCloud = _currentCloud = nullptr;
    _globalT.setIdentity(); _previousCloudTransform.setIdentity(); _counter = 0; _numKeyframes = 0;
  }
  // Not in the reference: hand over the NEXT frame of a recorded / buffered stream (same sensor offset and camera matrix as the call that
  // will process it) before processFrame of the current one; its makeCloud (:115, independent of the alignments before it) then runs next
  // to that alignment, and processFrame of the same DepthImage object picks the cloud up.  Same results, bit for bit.
  void prefetch(const DepthImage& depthImage, const Isometry3f& sensorOffset, const Matrix3f& cameraMatrix) { makeCloudBegin(cameraMatrix, sensorOffset, depthImage); }
  // pwn_tracker.cpp:106-215; nextDepthImage (not in the reference): prefetch()ed as soon as this frame's cloud exists
  FrameResult processFrame(const DepthImage& depthImage, const Isometry3f& sensorOffset, const Matrix3f& cameraMatrix, const Isometry3f& initialGuess = Isometry3f::Identity(),
                           const DepthImage* nextDepthImage = nullptr) {
    FrameResult out;
    int r, c; Matrix3f scaledCameraMatrix = cameraMatrix;
    Cloud* currentCloud;
    if (makeCloudPending(depthImage)) currentCloud = makeCloudEnd(r, c, scaledCameraMatrix);
    else { makeCloudDrop(); currentCloud = makeCloud(r, c, scaledCameraMatrix, sensorOffset, depthImage); }     // :115
    if (nextDepthImage) prefetch(*nextDepthImage, sensorOffset, cameraMatrix);
    if (_currentCloud != _previousCloud) delete _currentCloud;                                                   // the last non-key cloud
    _currentCloud = currentCloud;
    if (_previousCloud) {
      _aligner->setCurrentSensorOffset(sensorOffset); _aligner->setCurrentCloud(currentCloud);
      _aligner->setReferenceSensorOffset(_previousCloudOffset); _aligner->setReferenceCloud(_previousCloud);
      _aligner->correspondenceFinder()->setImageSize(r, c);
      _aligner->projector()->setCameraMatrix(scaledCameraMatrix); _aligner->projector()->setImageSize(r, c);
      const Isometry3f guess = iso_mul(iso_mul(_previousCloudTransform.inverse(), _globalT), initialGuess);     // :132
      _aligner->setInitialGuess(guess);
      _aligner->align();                                                                                         // :136
      if (_aligner->inliers() > 0) _globalT = iso_mul(_previousCloudTransform, _aligner->T());                   // :147
      else _globalT = iso_mul(_globalT, guess);                                                                  // :150
      if (!(_counter % 50)) {                                                                                    // :154-159: R <- R - 0.5 R (R^T R - I)
        float R[9], Rt[9], E[9], hR[9], D[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { R[i + 3 * j] = _globalT(i,j); Rt[j + 3 * i] = _globalT(i,j); hR[i + 3 * j] = 0.5f * _globalT(i,j); }
        mul3(Rt, R, E); E[0] -= 1.f; E[4] -= 1.f; E[8] -= 1.f;
        mul3(hR, E, D);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) _globalT(i,j) = R[i + 3 * j] - D[i + 3 * j];
      }
      _globalT.forceLastRow();
      out.aligned = true; out.inliers = _aligner->inliers(); out.error = _aligner->error(); out.T = _aligner->T();
      out.inliersFraction = (float)_aligner->inliers() / (float)(r * c);
      if (out.inliersFraction < _newFrameInliersFraction) {                                                      // :164-185
        out.newFrame = true; ++_numKeyframes;
        delete _previousCloud;
        _previousCloud = currentCloud; _previousCloudTransform = _globalT;
      }
    } else {                                                                                                     // :194-200
      out.newFrame = true; ++_numKeyframes;
      _previousCloud = currentCloud; _previousCloudTransform = _globalT; _previousCloudOffset = sensorOffset;
    }
    ++_counter;
    out.globalT = _globalT;
    return out;
  }
  Cloud* previousCloud() const { return _previousCloud; }  Cloud* currentCloud() const { return _currentCloud; }
 private:
  static void mul3(const float* A, const float* B, float* C) {      // 3x3 float product, left-to-right inner products (no FMA: build without -ffast-math)
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { float s = A[i] * B[3 * j]; s = s + A[i + 3] * B[1 + 3 * j]; s = s + A[i + 6] * B[2 + 3 * j]; C[i + 3 * j] = s; }
  }
  Cloud* _previousCloud = nullptr; Cloud* _currentCloud = nullptr;
  Isometry3f _globalT, _previousCloudTransform, _previousCloudOffset;
  float _newFrameInliersFraction = 0.4f;     // pwn_tracker.cpp:36
  int _counter = 0, _numKeyframes = 0;
};

}  // namespace pwn_hip
