// NOT COMPILED IN THIS REPOSITORY (needs the reference's headers, i.e. Eigen3 + OpenCV): see README.md.
#include "devicecloudregistry.h"

#include <stdexcept>
#include <vector>

namespace pwn {

static void check(pwn_hip_ctx* ctx, int rc) { if (rc) throw std::runtime_error(pwn_hip_last_error_string(ctx)); }

DeviceCloudRegistry::~DeviceCloudRegistry() {
  for (std::map<const Cloud*, Entry>::iterator it = _entries.begin(); it != _entries.end(); ++it) pwn_hip_cloud_destroy(_ctx, it->second.dev);
}

pwn_hip_cloud* DeviceCloudRegistry::deviceCloud(const Cloud* cloud, int capacity) {
  std::map<const Cloud*, Entry>::iterator it = _entries.find(cloud);
  if (it != _entries.end() && it->second.capacity >= capacity) return it->second.dev;
  if (it != _entries.end()) { pwn_hip_cloud_destroy(_ctx, it->second.dev); _entries.erase(it); }      // retired into the context's pool
  Entry e; e.dev = 0; e.capacity = capacity < 1 ? 1 : capacity; e.deviceOnly = false; e.hostSynced = false; e.hostSizeAtSync = 0;
  check(_ctx, pwn_hip_cloud_create(_ctx, e.capacity, &e.dev));
  _entries[cloud] = e;
  return e.dev;
}

void DeviceCloudRegistry::markDeviceOnly(const Cloud* cloud) {
  std::map<const Cloud*, Entry>::iterator it = _entries.find(cloud);
  if (it != _entries.end()) { it->second.deviceOnly = true; it->second.hostSynced = false; it->second.hostSizeAtSync = 0; }
}

// The reference's element types carry a vptr (Point / Normal: 32 B, InformationMatrix: 80 B, Stats: 112 B), so
// the vectors cannot be handed to the C-ABI as they are: the data goes through plain float staging arrays, field by field.
void DeviceCloudRegistry::upload(const Cloud* cloud, Entry& e) {
  const size_t n = cloud->points().size();
  std::vector<float> P(4 * n), N(4 * n), C(n), OP(16 * n), ON(16 * n);
  for (size_t i = 0; i < n; ++i) {
    for (int k = 0; k < 4; ++k) { P[4 * i + k] = cloud->points()[i][k]; N[4 * i + k] = i < cloud->normals().size() ? cloud->normals()[i][k] : 0.f; }
    C[i] = i < cloud->stats().size() ? cloud->stats()[i].curvature() : 1.0f;                   // stats.h:98-103 (default curvature 1: stats.h:26)
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) {                                    // column-major 4x4
      OP[16 * i + r + 4 * c] = i < cloud->pointInformationMatrix().size() ? cloud->pointInformationMatrix()[i](r, c) : 0.f;
      ON[16 * i + r + 4 * c] = i < cloud->normalInformationMatrix().size() ? cloud->normalInformationMatrix()[i](r, c) : 0.f;
    }
  }
  check(_ctx, pwn_hip_cloud_upload(_ctx, e.dev, (int)n, P.data(), N.data(), C.data(), OP.data(), ON.data()));
}

pwn_hip_cloud* DeviceCloudRegistry::deviceCloudForAlign(const Cloud* cloud) {
  std::map<const Cloud*, Entry>::iterator it = _entries.find(cloud);
  if (it != _entries.end() && it->second.deviceOnly) {
    // the device twin is the valid copy as long as the host side cannot have diverged from it: the host vectors are still empty (nobody
    // downloaded), or they are the very download this registry made and nobody reported a change (markHostModified) or resized them
    const size_t hostSize = cloud->points().size();
    if (hostSize == 0 || (it->second.hostSynced && hostSize == it->second.hostSizeAtSync)) return it->second.dev;
    it->second.deviceOnly = false;          // the caller filled or changed the host vectors (Cloud::transformInPlace, add, load ...): they win
  }
  deviceCloud(cloud, (int)cloud->points().size());
  Entry& e = _entries[cloud];
  upload(cloud, e);
  e.hostSynced = true; e.hostSizeAtSync = cloud->points().size();
  return e.dev;
}

void DeviceCloudRegistry::markHostModified(const Cloud* cloud) {
  std::map<const Cloud*, Entry>::iterator it = _entries.find(cloud);
  if (it != _entries.end()) { it->second.hostSynced = false; it->second.deviceOnly = false; }
}

void DeviceCloudRegistry::download(Cloud* cloud) {
  std::map<const Cloud*, Entry>::iterator it = _entries.find(cloud);
  if (it == _entries.end()) return;
  int n = 0;
  check(_ctx, pwn_hip_cloud_size(_ctx, it->second.dev, &n));
  std::vector<float> P(4 * (size_t)n), N(4 * (size_t)n), C(n), OP(16 * (size_t)n), ON(16 * (size_t)n);
  check(_ctx, pwn_hip_cloud_download(_ctx, it->second.dev, P.data(), N.data(), C.data(), OP.data(), ON.data()));
  cloud->points().resize(n); cloud->normals().resize(n); cloud->stats().resize(n);            // what PinholePointProjector::unProject +
  cloud->pointInformationMatrix().resize(n); cloud->normalInformationMatrix().resize(n);       // the calculators leave (depthimageconverterintegralimage.cpp:39-52)
  std::vector<float> S, E; std::vector<int> NP;
  bool haveStats = true;
  S.resize(16 * (size_t)n); E.resize(3 * (size_t)n); NP.resize(n);
  if (pwn_hip_cloud_download_stats(_ctx, it->second.dev, S.data(), E.data(), NP.data()) != PWN_HIP_OK) haveStats = false;   // converted without keep_stats
  for (int i = 0; i < n; ++i) {
    cloud->points()[i] = Point(Eigen::Vector3f(P[4 * i], P[4 * i + 1], P[4 * i + 2]));
    cloud->normals()[i] = Normal(Eigen::Vector3f(N[4 * i], N[4 * i + 1], N[4 * i + 2]));
    Stats& st = cloud->stats()[i];
    if (haveStats) {
      st.Eigen::Matrix4f::operator=(Eigen::Map<Eigen::Matrix4f>(&S[16 * (size_t)i]));
      st.setEigenValues(Eigen::Vector3f(E[3 * i], E[3 * i + 1], E[3 * i + 2]));
      st.setN(NP[i]);
    }
    st.setCurvature(C[i]);
    cloud->pointInformationMatrix()[i] = InformationMatrix(Eigen::Matrix4f(Eigen::Map<Eigen::Matrix4f>(&OP[16 * (size_t)i])));
    cloud->normalInformationMatrix()[i] = InformationMatrix(Eigen::Matrix4f(Eigen::Map<Eigen::Matrix4f>(&ON[16 * (size_t)i])));
  }
  it->second.hostSynced = true; it->second.hostSizeAtSync = (size_t)n;      // host == device from here until markHostModified / a resize
}

void DeviceCloudRegistry::release(const Cloud* cloud) {
  std::map<const Cloud*, Entry>::iterator it = _entries.find(cloud);
  if (it == _entries.end()) return;
  pwn_hip_cloud_destroy(_ctx, it->second.dev);
  _entries.erase(it);
}

}  // namespace pwn
