// NOT COMPILED IN THIS REPOSITORY (needs the reference's headers, i.e. Eigen3 + OpenCV): see README.md.
#include "hipdepthimageconverter.h"

#include <cassert>
#include <stdexcept>

#include "g2o_frontend/pwn_core/pinholepointprojector.h"
#include "g2o_frontend/pwn_core/statscalculatorintegralimage.h"

namespace pwn {

HipDepthImageConverter::HipDepthImageConverter(DeviceCloudRegistry* registry, PointProjector* projector, StatsCalculator* statsCalculator,
                                               PointInformationMatrixCalculator* pointInformationMatrixCalculator,
                                               NormalInformationMatrixCalculator* normalInformationMatrixCalculator)
    : DepthImageConverterIntegralImage(projector, statsCalculator, pointInformationMatrixCalculator, normalInformationMatrixCalculator),
      _registry(registry), _downloadToHost(true), _computeGaussians(false) {}

pwn_hip_converter_params HipDepthImageConverter::params(const Eigen::Isometry3f& sensorOffset) const {
  const PinholePointProjector* pp = dynamic_cast<const PinholePointProjector*>(_projector);
  const StatsCalculatorIntegralImage* sc = dynamic_cast<const StatsCalculatorIntegralImage*>(_statsCalculator);
  if (!pp || !sc) throw std::runtime_error("HipDepthImageConverter: needs a PinholePointProjector and a StatsCalculatorIntegralImage");
  pwn_hip_converter_params p;
  pwn_hip_default_converter_params(&p);
  Eigen::Map<Eigen::Matrix3f>(p.K) = pp->cameraMatrix();                                  // column-major = Eigen's layout
  p.min_distance = pp->minDistance(); p.max_distance = pp->maxDistance();                 // pointprojector.h:55-76
  p.world_radius = sc->worldRadius();                                                     // statscalculatorintegralimage.h:47-110
  p.min_image_radius = sc->minImageRadius(); p.max_image_radius = sc->maxImageRadius();
  p.min_points = sc->minPoints(); p.stats_curvature_threshold = sc->curvatureThreshold();
  p.point_info_curvature_threshold = _pointInformationMatrixCalculator->curvatureThreshold();      // informationmatrixcalculator.h:68
  p.normal_info_curvature_threshold = _normalInformationMatrixCalculator->curvatureThreshold();
  const InformationMatrix pf = _pointInformationMatrixCalculator->flatInformationMatrix(), pn = _pointInformationMatrixCalculator->nonFlatInformationMatrix();
  const InformationMatrix nf = _normalInformationMatrixCalculator->flatInformationMatrix(), nn = _normalInformationMatrixCalculator->nonFlatInformationMatrix();
  for (int i = 0; i < 3; ++i) {                                                           // the calculators only ever hold diagonal matrices (:23-26,107-108,142-143)
    p.point_flat_diag[i] = pf(i, i); p.point_nonflat_diag[i] = pn(i, i);
    p.normal_flat_diag[i] = nf(i, i); p.normal_nonflat_diag[i] = nn(i, i);
  }
  Eigen::Isometry3f off = sensorOffset;
  off.matrix().row(3) << 0.0f, 0.0f, 0.0f, 1.0f;
  Eigen::Map<Eigen::Matrix4f>(p.sensor_offset) = off.matrix();
  return p;
}

void HipDepthImageConverter::compute(Cloud& cloud, const DepthImage& depthImage, const Eigen::Isometry3f& sensorOffset) {
  assert(_projector && "HipDepthImageConverter: missing _projector");                    // depthimageconverterintegralimage.cpp:18-26
  assert(_statsCalculator && "HipDepthImageConverter: missing _statsCalculator");
  assert(_pointInformationMatrixCalculator && "HipDepthImageConverter: missing _pointInformationMatrixCalculator");
  assert(_normalInformationMatrixCalculator && "HipDepthImageConverter: missing _normalInformationMatrixCalculator");
  assert(depthImage.rows > 0 && depthImage.cols > 0 && "HipDepthImageConverter: depthImage has zero size");
  pwn_hip_ctx* ctx = _registry->context();
  const int rows = depthImage.rows, cols = depthImage.cols;

  cloud.clear();                                                                          // :29
  _projector->setImageSize(rows, cols);                                                   // :30  (side effects the callers rely on)
  _projector->setTransform(Eigen::Isometry3f::Identity());                                // :38
  const pwn_hip_converter_params p = params(sensorOffset);

  DepthImage contiguous;                                                                  // cv::Mat_ rows may be strided (ROI): the ABI wants [rows][cols]
  const DepthImage& src = depthImage.isContinuous() ? depthImage : (contiguous = depthImage.clone());
  _indexImage.create(rows, cols);
  pwn_hip_cloud* dc = _registry->deviceCloud(&cloud, rows * cols);
  int rc = pwn_hip_convert(ctx, &p, reinterpret_cast<const float*>(src.data), rows, cols, dc, reinterpret_cast<int*>(_indexImage.data), 0,
                           _downloadToHost ? 1 : 0 /* keep the per-point Stats when the host will read them */);
  if (rc) throw std::runtime_error(pwn_hip_last_error_string(ctx));
  if (_computeGaussians) {
    const PinholePointProjector* pp = dynamic_cast<const PinholePointProjector*>(_projector);
    rc = pwn_hip_cloud_gaussians(ctx, &p, reinterpret_cast<const float*>(src.data), rows, cols, dc, pp->baseline(), pp->alpha());
    if (rc) throw std::runtime_error(pwn_hip_last_error_string(ctx));
  }
  _registry->markDeviceOnly(&cloud);
  if (_downloadToHost) _registry->download(&cloud);
}

}  // namespace pwn
