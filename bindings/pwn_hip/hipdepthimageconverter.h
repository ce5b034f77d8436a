// hipdepthimageconverter.h -- DepthImageConverterIntegralImage whose compute() runs on an MI355X through libpwn_hip.so.
// Replaces the body of g2o_frontend/pwn_core/depthimageconverterintegralimage.cpp:15-55; everything else (constructor,
// collaborators, indexImage()) is the reference's.  NOT COMPILED IN THIS REPOSITORY (Eigen3 + OpenCV needed): see README.md.
#ifndef PWN_HIP_HIPDEPTHIMAGECONVERTER_H
#define PWN_HIP_HIPDEPTHIMAGECONVERTER_H

#include "g2o_frontend/pwn_core/depthimageconverterintegralimage.h"
#include "devicecloudregistry.h"

namespace pwn {

class HipDepthImageConverter : public DepthImageConverterIntegralImage {
 public:
  HipDepthImageConverter(DeviceCloudRegistry* registry,
                         PointProjector* projector = 0, StatsCalculator* statsCalculator = 0,
                         PointInformationMatrixCalculator* pointInformationMatrixCalculator = 0,
                         NormalInformationMatrixCalculator* normalInformationMatrixCalculator = 0);
  virtual ~HipDepthImageConverter() {}

  // depthimageconverter.h:47.  The cloud is produced on the device and stays there (the aligner takes it from the registry).
  // downloadToHost (default TRUE: what a drop-in owes callers that look at the host Cloud afterwards -- points().size() in
  // pwn_tracker, viewers, Cloud::save, a host-side Merger) also fills the host vectors; callers that only align set it to false and
  // save the PCIe copy of ~190 bytes per point.
  virtual void compute(Cloud& cloud, const DepthImage& depthImage, const Eigen::Isometry3f& sensorOffset = Eigen::Isometry3f::Identity());

  bool downloadToHost() const { return _downloadToHost; }
  void setDownloadToHost(bool v) { _downloadToHost = v; }
  // also produce the per-point sensor Gaussians (pinholepointprojector.cpp:104-123) on the device: only Merger::merge reads them
  bool computeGaussians() const { return _computeGaussians; }
  void setComputeGaussians(bool v) { _computeGaussians = v; }

  // the parameter block the C-ABI takes, filled from the four collaborators (depthimageconverter.h:114-117)
  pwn_hip_converter_params params(const Eigen::Isometry3f& sensorOffset) const;

 protected:
  DeviceCloudRegistry* _registry;
  bool _downloadToHost;
  bool _computeGaussians;
};

}  // namespace pwn
#endif
