// hipaligner.h -- pwn::Aligner whose align() runs on an MI355X through libpwn_hip.so; sits exactly where CuAligner::align
// (pwn_cuda/cualigner.h:8-15, cualigner.cpp:32-137) sat: the virtual of aligner.h:308, body aligner.cpp:49-150.
// NOT COMPILED IN THIS REPOSITORY (needs the reference's headers, i.e. Eigen3 + OpenCV): see README.md.
#ifndef PWN_HIP_HIPALIGNER_H
#define PWN_HIP_HIPALIGNER_H

#include <vector>

#include "g2o_frontend/pwn_core/aligner.h"
#include "devicecloudregistry.h"

namespace pwn {

// Linearizer / CorrespondenceFinder keep their results in protected members without setters (linearizer.h:139-143,
// correspondencefinder.h:245); callers that read them after align() (Linearizer::H(), ::error(), CorrespondenceFinder::
// numCorrespondences()) get them through these two subclasses.  Plain Linearizer / CorrespondenceFinder objects work too, their
// H() / numCorrespondences() then keep whatever they held.
class HipLinearizer : public Linearizer {
 public:
  void setResult(const Matrix6f& H, const Vector6f& b, float error, int inliers) { _H = H; _b = b; _error = error; _inliers = inliers; }
  void setResult(float error, int inliers) { _error = error; _inliers = inliers; }
};
class HipCorrespondenceFinder : public CorrespondenceFinder {
 public:
  void setNumCorrespondences(int n) { _numCorrespondences = n; }
};

class HipAligner : public Aligner {
 public:
  explicit HipAligner(DeviceCloudRegistry* registry);
  virtual ~HipAligner() {}

  virtual void align();

  // Aligner::_computeStatistics (aligner.cpp:127,152-199) costs one more linearizer pass + 6x6 host math per alignment; the
  // reference always runs it.  On by default for that reason; the loop-closure batch path switches it off.
  bool computeStatistics() const { return _computeStatistics; }
  void setComputeStatistics(bool v) { _computeStatistics = v; }
  // PwnMatcherBase::matchClouds reads the finder's two depth images after align() (pwn_tracker/pwn_matcher_base.cpp:153-155);
  // they are fetched from the device only when asked for
  bool fetchFinderImages() const { return _fetchFinderImages; }
  void setFetchFinderImages(bool v) { _fetchFinderImages = v; }

  // The candidate loop of PwnCloser::processPartition (pwn_tracker/pwn_closer.cpp:92-111: one matchFrames -> matchClouds -> align() per
  // candidate, strictly sequential) as ONE call: n independent alignments with this aligner's parameters, per-pair initial guesses
  // (already with matchClouds' z-translation reset, pwn_matcher_base.cpp:114) and, if `scores` is given, matchClouds' depth-agreement
  // score of every pair (pwn_matcher_base.cpp:153-182), to which the caller applies matchFrames' thresholds (pwn_closer.cpp:138-141).
  void alignBatch(const std::vector<Cloud*>& references, const std::vector<Cloud*>& currents,
                  const std::vector<Eigen::Isometry3f, Eigen::aligned_allocator<Eigen::Isometry3f> >& initialGuesses,
                  std::vector<pwn_hip_align_result>& results, std::vector<pwn_hip_match_result>* scores = 0,
                  float frameInlierDepthThreshold = 50.0f);

  const pwn_hip_align_result& lastResult() const { return _last; }      // per-iteration chi2 / inliers / C_i / K_i of the last align()
  pwn_hip_aligner_params params() const;

 protected:
  DeviceCloudRegistry* _registry;
  bool _computeStatistics;
  bool _fetchFinderImages;
  pwn_hip_align_result _last;
};

}  // namespace pwn
#endif
