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
#include <stdexcept>
#include <string>
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
  void transformInPlace(const Isometry3f& T) { _ctx->check(pwn_hip_cloud_transform_in_place(_ctx->handle(), _h, T.data())); }   // cloud.cpp:173-186
  // scene maintenance (cloud.cpp:11-171)
  void add(const Cloud& cloud, const Isometry3f& T = Isometry3f::Identity()) { _ctx->check(pwn_hip_cloud_add(_ctx->handle(), _h, cloud.handle(), T.data())); }
  bool save(const char* filename, const Isometry3f& T = Isometry3f::Identity(), int step = 1, bool binary = false) const {
    return pwn_hip_cloud_save(_ctx->handle(), _h, filename, T.data(), step, binary ? 1 : 0) == PWN_HIP_OK;
  }
  bool load(Isometry3f& T, const char* filename) { return pwn_hip_cloud_load(_ctx->handle(), _h, filename, T.data()) == PWN_HIP_OK; }
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
  void project(Context& ctx, IntImage& indexImage, DepthImage& depthImage, const Cloud& cloud) const {  // .cpp:33-66
    indexImage.create(_imageRows, _imageCols); depthImage.create(_imageRows, _imageCols);
    ctx.check(pwn_hip_project(ctx.handle(), _cameraMatrix.data(), _transform.data(), _minDistance, _maxDistance, _imageRows, _imageCols,
                              cloud.handle(), indexImage.data.data(), depthImage.data.data()));
  }
 private:
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
  PinholePointProjector* projector() { return _projector; }
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
  void setImageSize(int r, int c) { _rows = r; _cols = c; }
  int imageRows() const { return _rows; }  int imageCols() const { return _cols; }
  int numCorrespondences() const { return _numCorrespondences; }
  IntImage& referenceIndexImage() { return _referenceIndexImage; }  IntImage& currentIndexImage() { return _currentIndexImage; }
  DepthImage& referenceDepthImage() { return _referenceDepthImage; }  DepthImage& currentDepthImage() { return _currentDepthImage; }
 private:
  friend class Aligner;
  float _inlierDistanceThreshold = 0.5f, _inlierNormalAngularThreshold = (float)std::cos(M_PI / 6), _flatCurvatureThreshold = 0.02f,
        _inlierCurvatureRatioThreshold = 1.3f;
  int _rows = 0, _cols = 0, _numCorrespondences = 0;
  IntImage _referenceIndexImage, _currentIndexImage; DepthImage _referenceDepthImage, _currentDepthImage;
};
class Aligner;
// linearizer.{h,cpp} (defaults .cpp:9-15)
class Linearizer {
 public:
  void setAligner(Aligner* a) { _aligner = a; }
  float inlierMaxChi2() const { return _inlierMaxChi2; }  void setInlierMaxChi2(float v) { _inlierMaxChi2 = v; }
  bool robustKernel() const { return _robustKernel; }      void setRobustKernel(bool v) { _robustKernel = v; }
  float error() const { return _error; }  int inliers() const { return _inliers; }
 private:
  friend class Aligner;
  Aligner* _aligner = nullptr; float _inlierMaxChi2 = 9e3f; bool _robustKernel = true; float _error = 0.f; int _inliers = 0;
};

// aligner.{h,cpp}
class Aligner {
 public:
  explicit Aligner(Context* ctx) : _ctx(ctx) {}
  virtual ~Aligner() {}
  void setProjector(PinholePointProjector* p) { _projector = p; }            PinholePointProjector* projector() { return _projector; }
  void setLinearizer(Linearizer* l) { _linearizer = l; if (l) l->setAligner(this); }  Linearizer* linearizer() { return _linearizer; }
  void setCorrespondenceFinder(CorrespondenceFinder* f) { _correspondenceFinder = f; } CorrespondenceFinder* correspondenceFinder() { return _correspondenceFinder; }
  void setReferenceCloud(Cloud* c) { _referenceCloud = c; }  void setCurrentCloud(Cloud* c) { _currentCloud = c; }     // aligner.h:60-80
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
    _ctx->check(pwn_hip_align(_ctx->handle(), &p, _referenceCloud->handle(), _currentCloud->handle(), &_result));
    _T = Isometry3f(_result.T); _error = _result.error; _inliers = _result.inliers; _totalTime = _result.total_time_ms;
    _linearizer->_error = _error; _linearizer->_inliers = _inliers;
    _correspondenceFinder->_numCorrespondences = _result.iterations > 0 ? _result.iter_correspondences[_result.iterations - 1] : 0;
    if (fetchImages) {
      CorrespondenceFinder& f = *_correspondenceFinder;
      f._referenceIndexImage.create(p.rows, p.cols); f._currentIndexImage.create(p.rows, p.cols);
      f._referenceDepthImage.create(p.rows, p.cols); f._currentDepthImage.create(p.rows, p.cols);
      _ctx->check(pwn_hip_align_images(_ctx->handle(), f._referenceIndexImage.data.data(), f._referenceDepthImage.data.data(),
                                       f._currentIndexImage.data.data(), f._currentDepthImage.data.data()));
    }
  }
 protected:
  Context* _ctx;
  PinholePointProjector* _projector = nullptr; Linearizer* _linearizer = nullptr; CorrespondenceFinder* _correspondenceFinder = nullptr;
  Cloud* _referenceCloud = nullptr; Cloud* _currentCloud = nullptr;
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
  int scale() const { return _scale; }  void setScale(int s) { _scale = s; }
  Aligner* aligner() { return _aligner; }  DepthImageConverter* converter() { return _converter; }

  // .cpp:57-86: returns a new Cloud owned by the caller; r, c, cameraMatrix receive the scaled values
  Cloud* makeCloud(int& r, int& c, Matrix3f& cameraMatrix, const Isometry3f& sensorOffset, const DepthImage& depthImage) {
    PinholePointProjector* projector = _converter->projector();
    const float invScale = 1.0f / _scale;
    Matrix3f scaled = cameraMatrix;
    for (int i = 0; i < 9; ++i) scaled.m[i] = scaled.m[i] * invScale;
    scaled(2,2) = 1.0f;
    projector->setCameraMatrix(scaled);
    projector->setImageSize(depthImage.rows / _scale, depthImage.cols / _scale);
    DepthImage scaledImage;
    DepthImage_scale(*_ctx, scaledImage, depthImage, _scale);
    cameraMatrix = projector->cameraMatrix(); r = projector->imageRows(); c = projector->imageCols();
    Cloud* cloud = new Cloud(*_ctx, scaledImage.rows * scaledImage.cols > 0 ? scaledImage.rows * scaledImage.cols : 1);
    _converter->compute(*cloud, scaledImage, sensorOffset);
    ++numCalls;
    return cloud;
  }
  // .cpp:88-183
  void matchClouds(MatcherResult& result, Cloud* fromCloud, Cloud* toCloud, const Isometry3f& fromOffset, const Isometry3f& toOffset,
                   const Matrix3f& toCameraMatrix, int toRows, int toCols, const Isometry3f& initialGuess = Isometry3f::Identity()) {
    PinholePointProjector* projector = _aligner->projector();
    _aligner->setReferenceSensorOffset(fromOffset);
    _aligner->setCurrentSensorOffset(toOffset);
    Isometry3f ig = initialGuess;
    ig(2,3) = 0.f;
    _aligner->setInitialGuess(ig);
    projector->setCameraMatrix(toCameraMatrix);
    projector->setImageSize(toRows, toCols);
    projector->scale((float)(1. / _scale));
    _aligner->correspondenceFinder()->setImageSize(projector->imageRows(), projector->imageCols());
    _aligner->setReferenceCloud(fromCloud);
    _aligner->setCurrentCloud(toCloud);
    _aligner->align();
    for (int i = 0; i < 16; ++i) result.transform[i] = _aligner->T().m[i];
    for (int i = 0; i < 36; ++i) result.informationMatrix[i] = (i % 7 == 0) ? 100.0 : 0.0;
    result.cloud_inliers = _aligner->inliers();
    pwn_hip_match_result m;
    _ctx->check(pwn_hip_match_score(_ctx->handle(), _frameInlierDepthThreshold, &m));
    result.image_reprojectionDistance = m.image_reprojection_distance;
    result.image_nonZeros = m.image_non_zeros; result.image_outliers = m.image_outliers; result.image_inliers = m.image_inliers;
  }
  int numCalls = 0;
 protected:
  Context* _ctx; Aligner* _aligner; DepthImageConverter* _converter;
  float _frameInlierDepthThreshold = 50.f;   // .cpp:13
  int _scale = 2;                            // .cpp:12
};

}  // namespace pwn_hip
