// devicecloudregistry.h -- which device cloud (pwn_hip_cloud) belongs to which host pwn::Cloud.
//
// The reference passes clouds around as raw `Cloud*` owned by the caller (pwn_tracker/pwn_matcher_base.cpp:77-85 returns a
// `new Cloud`; aligner.h:381-386 keeps non-owning pointers), so the association lives outside the Cloud: a registry keyed by the
// Cloud's address.  A cloud converted by HipDepthImageConverter exists ONLY on the device until somebody asks for the host
// vectors (download()); a cloud that was filled on the host (Cloud::load, a CPU converter) is uploaded the first time the
// aligner sees it.  NOT COMPILED IN THIS REPOSITORY (needs the reference's headers, i.e. Eigen3 + OpenCV): see README.md.
#ifndef PWN_HIP_DEVICECLOUDREGISTRY_H
#define PWN_HIP_DEVICECLOUDREGISTRY_H

#include <map>

#include "g2o_frontend/pwn_core/cloud.h"
#include "pwn_hip.h"

namespace pwn {

class DeviceCloudRegistry {
 public:
  explicit DeviceCloudRegistry(pwn_hip_ctx* ctx) : _ctx(ctx) {}
  ~DeviceCloudRegistry();

  pwn_hip_ctx* context() const { return _ctx; }

  // device twin of `cloud` with room for `capacity` points, created on first use; a twin that is too small is replaced
  pwn_hip_cloud* deviceCloud(const Cloud* cloud, int capacity);
  // the twin the aligner needs: the registered one if the cloud came from HipDepthImageConverter and the host vectors are still empty or
  // are this registry's own download (size unchanged, no markHostModified), else an upload of the host vectors
  pwn_hip_cloud* deviceCloudForAlign(const Cloud* cloud);
  // marks the device twin as the only valid copy (after a device-side compute())
  void markDeviceOnly(const Cloud* cloud);
  // the caller changed the host vectors of a cloud this registry knows (Cloud::transformInPlace, Cloud::add, direct writes): the next
  // align uploads them again.  A change of points().size() is noticed without this call; an in-place change of values is not.
  void markHostModified(const Cloud* cloud);
  // fills the host vectors (points, normals, stats, both information-matrix vectors) from the device twin
  void download(Cloud* cloud);
  // MUST be called wherever the caller deletes the Cloud (pwn_matcher_base users own their clouds): the registry is keyed by the Cloud's
  // address, and a new Cloud allocated at a released address would otherwise inherit the old device twin
  void release(const Cloud* cloud);

 private:
  struct Entry { pwn_hip_cloud* dev; int capacity; bool deviceOnly; bool hostSynced; size_t hostSizeAtSync; };
  void upload(const Cloud* cloud, Entry& e);
  pwn_hip_ctx* _ctx;
  std::map<const Cloud*, Entry> _entries;
};

}  // namespace pwn
#endif
