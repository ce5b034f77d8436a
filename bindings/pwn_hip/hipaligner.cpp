// NOT COMPILED IN THIS REPOSITORY (needs the reference's headers, i.e. Eigen3 + OpenCV): see README.md.
#include "hipaligner.h"

#include <cassert>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <sys/time.h>
#include <vector>

#include "g2o_frontend/pwn_core/pinholepointprojector.h"
#include "g2o_frontend/pwn_core/se3_prior.h"

namespace pwn {

HipAligner::HipAligner(DeviceCloudRegistry* registry) : Aligner(), _registry(registry), _computeStatistics(true), _fetchFinderImages(false) {
  std::memset(&_last, 0, sizeof(_last));
}

pwn_hip_aligner_params HipAligner::params() const {
  const PinholePointProjector* pp = dynamic_cast<const PinholePointProjector*>(_projector);
  if (!pp) throw std::runtime_error("HipAligner: needs a PinholePointProjector");
  pwn_hip_aligner_params p;
  pwn_hip_default_aligner_params(&p);
  Eigen::Map<Eigen::Matrix3f>(p.K) = pp->cameraMatrix();
  p.min_distance = pp->minDistance(); p.max_distance = pp->maxDistance();
  p.rows = pp->imageRows(); p.cols = pp->imageCols();                                         // callers set both (aligner.cpp needs them: App. A #27)
  p.inlier_distance_threshold = _correspondenceFinder->inlierDistanceThreshold();             // correspondencefinder.h:133-186
  p.inlier_normal_angular_threshold = _correspondenceFinder->inlierNormalAngularThreshold();
  p.flat_curvature_threshold = _correspondenceFinder->flatCurvatureThreshold();
  p.inlier_curvature_ratio_threshold = _correspondenceFinder->inlierCurvatureRatioThreshold();
  p.inlier_max_chi2 = _linearizer->inlierMaxChi2(); p.robust_kernel = _linearizer->robustKernel() ? 1 : 0;   // linearizer.h:72-93
  p.outer_iterations = _outerIterations; p.inner_iterations = _innerIterations;
  Eigen::Map<Eigen::Matrix4f>(p.reference_sensor_offset) = _referenceSensorOffset.matrix();   // the setters already forced the last rows (aligner.h:130-190)
  Eigen::Map<Eigen::Matrix4f>(p.current_sensor_offset) = _currentSensorOffset.matrix();
  Eigen::Map<Eigen::Matrix4f>(p.initial_guess) = _initialGuess.matrix();
  return p;
}

void HipAligner::align() {
  assert(_projector && "HipAligner: missing _projector");                                     // aligner.cpp:50-54
  assert(_linearizer && "HipAligner: missing _linearizer");
  assert(_correspondenceFinder && "HipAligner: missing _correspondenceFinder");
  assert(_referenceCloud && "HipAligner: missing _referenceCloud");
  assert(_currentCloud && "HipAligner: missing _currentCloud");
  pwn_hip_ctx* ctx = _registry->context();
  struct timeval tvStart, tvEnd;
  gettimeofday(&tvStart, 0);

  const pwn_hip_aligner_params p = params();
  pwn_hip_cloud* ref = _registry->deviceCloudForAlign(_referenceCloud);
  pwn_hip_cloud* cur = _registry->deviceCloudForAlign(_currentCloud);

  // the priors of aligner.cpp:96-108 (Aligner::addRelativePrior / addAbsolutePrior, aligner.cpp:34-40)
  std::vector<pwn_hip_prior> priors(_priors.size());
  for (size_t j = 0; j < _priors.size(); ++j) {
    pwn_hip_prior& q = priors[j];
    std::memset(&q, 0, sizeof(q));
    const SE3AbsolutePrior* abs = dynamic_cast<const SE3AbsolutePrior*>(_priors[j]);
    q.kind = abs ? 1 : 0;
    Eigen::Map<Eigen::Matrix4f>(q.mean) = _priors[j]->mean().matrix();
    Eigen::Map<Eigen::Matrix4f>(q.reference_transform) = abs ? abs->referenceTransform().matrix() : Eigen::Matrix4f(Eigen::Matrix4f::Identity());
    Eigen::Map<Matrix6f>(q.information) = _priors[j]->information();
  }

  pwn_hip_align_statistics st;
  std::memset(&st, 0, sizeof(st));
  // priors or not, statistics or not: one entry point (n_priors = 0 runs the device-resident Gauss-Newton loop)
  const int rc = pwn_hip_align_with_priors_ex(ctx, &p, ref, cur, (int)priors.size(), priors.empty() ? 0 : &priors[0], &_last,
                                              _computeStatistics ? &st : 0);
  if (rc) throw std::runtime_error(pwn_hip_last_error_string(ctx));

  _T.matrix() = Eigen::Map<Eigen::Matrix4f>(_last.T);                                         // aligner.cpp:115-117 happened on the device
  _T.matrix().block<1, 4>(3, 0) << 0.0f, 0.0f, 0.0f, 1.0f;
  _error = _last.error; _inliers = _last.inliers;                                             // :124-125 (values of the last in-loop update)

  if (_computeStatistics) {                                                                   // :127-143
    _mean = Eigen::Map<Vector6f>(st.mean);
    _omega = Eigen::Map<Matrix6f>(st.omega);
    _translationalEigenRatio = st.translational_eigen_ratio;
    _rotationalEigenRatio = st.rotational_eigen_ratio;
    if (_debug && (_rotationalEigenRatio > _rotationalMinEigenRatio || _translationalEigenRatio > _translationalMinEigenRatio))
      std::cerr << "************** WARNING SOLUTION MIGHT BE INVALID (eigenratio failure) **************" << std::endl
                << "tr: " << _translationalEigenRatio << " rr: " << _rotationalEigenRatio << std::endl;
  }
  // what the collaborators expose afterwards
  if (HipLinearizer* hl = dynamic_cast<HipLinearizer*>(_linearizer)) {
    // _computeStatistics' extra update overwrites the linearizer's H, b, error, inliers (aligner.cpp:165-170); without it the
    // linearizer holds the last in-loop values
    if (_computeStatistics) hl->setResult(Eigen::Map<Matrix6f>(st.H), Eigen::Map<Vector6f>(st.b), st.error, st.inliers);
    else hl->setResult(_last.error, _last.inliers);
  }
  if (HipCorrespondenceFinder* hf = dynamic_cast<HipCorrespondenceFinder*>(_correspondenceFinder))
    hf->setNumCorrespondences(_last.iterations > 0 ? _last.iter_correspondences[_last.iterations - 1] : 0);
  if (_fetchFinderImages) {                                                                   // pwn_matcher_base.cpp:153-155
    IntImage& ri = _correspondenceFinder->referenceIndexImage(); IntImage& ci = _correspondenceFinder->currentIndexImage();
    DepthImage& rd = _correspondenceFinder->referenceDepthImage(); DepthImage& cd = _correspondenceFinder->currentDepthImage();
    ri.create(p.rows, p.cols); ci.create(p.rows, p.cols); rd.create(p.rows, p.cols); cd.create(p.rows, p.cols);
    const int rc2 = pwn_hip_align_images(ctx, reinterpret_cast<int*>(ri.data), reinterpret_cast<float*>(rd.data),
                                         reinterpret_cast<int*>(ci.data), reinterpret_cast<float*>(cd.data));
    if (rc2) throw std::runtime_error(pwn_hip_last_error_string(ctx));
  }
  // side effect of the reference that callers may observe: the projector ends up at the last reference projection's pose (:73)
  _projector->setTransform(_T * _referenceSensorOffset);

  gettimeofday(&tvEnd, 0);
  const double tStart = tvStart.tv_sec * 1000.0 + tvStart.tv_usec * 0.001, tEnd = tvEnd.tv_sec * 1000.0 + tvEnd.tv_usec * 0.001;
  _totalTime = tEnd - tStart;                                                                 // :120-123 (host wall clock, as the reference)
}

void HipAligner::alignBatch(const std::vector<Cloud*>& references, const std::vector<Cloud*>& currents,
                            const std::vector<Eigen::Isometry3f, Eigen::aligned_allocator<Eigen::Isometry3f> >& initialGuesses,
                            std::vector<pwn_hip_align_result>& results, std::vector<pwn_hip_match_result>* scores, float frameInlierDepthThreshold) {
  const size_t n = references.size();
  if (currents.size() != n || (!initialGuesses.empty() && initialGuesses.size() != n)) throw std::runtime_error("HipAligner::alignBatch: list sizes differ");
  results.resize(n);
  if (scores) scores->resize(n);
  if (n == 0) return;
  pwn_hip_ctx* ctx = _registry->context();
  const pwn_hip_aligner_params p = params();
  std::vector<pwn_hip_cloud*> r(n), c(n);
  for (size_t i = 0; i < n; ++i) { r[i] = _registry->deviceCloudForAlign(references[i]); c[i] = _registry->deviceCloudForAlign(currents[i]); }
  std::vector<float> g(16 * initialGuesses.size());
  for (size_t i = 0; i < initialGuesses.size(); ++i) {
    Eigen::Isometry3f T = initialGuesses[i];
    T.matrix().row(3) << 0.0f, 0.0f, 0.0f, 1.0f;                                               // aligner.h:130-133
    Eigen::Map<Eigen::Matrix4f> dst(&g[16 * i]);      // (a temporary `Map<..>(&g[..]) = ..` would parse as a declaration)
    dst = T.matrix();
  }
  const int rc = pwn_hip_align_batch_ex(ctx, &p, (int)n, &r[0], &c[0], g.empty() ? 0 : &g[0], &results[0], frameInlierDepthThreshold,
                                        scores ? &(*scores)[0] : 0, 0);
  if (rc) throw std::runtime_error(pwn_hip_last_error_string(ctx));
}

}  // namespace pwn
