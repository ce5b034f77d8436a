/*
 * pwn_hip.h -- C-ABI of the MI355X (gfx950) PWN dense-registration path.
 *
 * This is the drop-in boundary behind the reference's pwn_core C++ API
 * (grisetti/g2o_frontend, g2o_frontend/pwn_core/).  Every entry point names the reference
 * method it replaces (file:line relative to g2o_frontend/pwn_core/ unless a directory is given).
 * The in-tree precedent for a C-style device boundary is pwn_cuda/cudaaligner.h:59-80
 * (createContext / initComputation / simpleIteration / getHb); the shape below follows it:
 * opaque context, plain pointers and sizes, int status codes, nothing thrown across the ABI.
 *
 * Conventions
 *   - matrices: COLUMN-MAJOR float (Eigen default, as pwn_cuda/cualigner.cpp:56-67 passes them);
 *     3x3 = 9 floats, 4x4 isometries = 16 floats, 6x6 = 36 floats.
 *   - images: row-major [rows][cols]; index images are int32 (-1 = empty), depth images float32
 *     metres (pwn_typedefs.h:57-62).
 *   - data pointers may be HOST or DEVICE memory; the library detects which
 *     (hipPointerGetAttributes) and copies accordingly.  Parameter structs, result structs and
 *     handle arrays are always host memory.
 *   - a context is bound to one GPU; calls on one context must be serialised by the caller
 *     (the reference objects are not re-entrant either: aligner.cpp:60-76 mutates the shared
 *     projector).  Use one context per (GPU, host thread).
 *   - all calls are synchronous with respect to the host unless noted: outputs are valid on return.
 *   - status: 0 = ok; otherwise one of pwn_hip_status; pwn_hip_last_error_string() gives detail.
 */
#ifndef PWN_HIP_H
#define PWN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PWN_HIP_MAX_ITERATIONS 64

typedef enum pwn_hip_status {
  PWN_HIP_OK = 0,
  PWN_HIP_ERR_INVALID_ARGUMENT = 1,
  PWN_HIP_ERR_NO_DEVICE = 2,
  PWN_HIP_ERR_ALLOCATION = 3,      /* cf. pwn_cuda Operation::Allocation      (cudaaligner.h:11) */
  PWN_HIP_ERR_COPY = 4,            /* cf. CopyToDevice / CopyFromDevice        (:13-14) */
  PWN_HIP_ERR_LAUNCH = 5,          /* cf. KernelLaunch                          (:15) */
  PWN_HIP_ERR_CAPACITY = 6         /* image / batch / cloud larger than the context or handle allows */
} pwn_hip_status;

typedef struct pwn_hip_ctx pwn_hip_ctx;       /* one per (GPU, host thread) */
typedef struct pwn_hip_cloud pwn_hip_cloud;   /* device-resident pwn::Cloud (cloud.h:20-187) */

/* Parameters of DepthImageConverterIntegralImage and its four collaborators
 * (depthimageconverter.h:29-32).  Defaults = the class defaults of the reference. */
typedef struct pwn_hip_converter_params {
  float K[9];                            /* PinholePointProjector::setCameraMatrix (pinholepointprojector.h:51) */
  float min_distance;                    /* PointProjector::setMinDistance (pointprojector.h:62), default 0.01 (pointprojector.cpp:9) */
  float max_distance;                    /* setMaxDistance (:76), default 6.0 (pointprojector.cpp:10) */
  float world_radius;                    /* StatsCalculatorIntegralImage::setWorldRadius, 0.1 (statscalculatorintegralimage.cpp:7) */
  int   min_image_radius;                /* setMinImageRadius, 10 (:9) */
  int   max_image_radius;                /* setMaxImageRadius, 30 (:8) */
  int   min_points;                      /* setMinPoints, 50 (:10) */
  float stats_curvature_threshold;       /* setCurvatureThreshold, 0.02 (:11) */
  float point_info_curvature_threshold;  /* PointInformationMatrixCalculator::setCurvatureThreshold, 0.02 (informationmatrixcalculator.h:109) */
  float normal_info_curvature_threshold; /* NormalInformationMatrixCalculator, 0.02 (informationmatrixcalculator.h:144) */
  float point_flat_diag[3];              /* diag(1000,1,1) (informationmatrixcalculator.h:107) */
  float point_nonflat_diag[3];           /* diag(1,1,1) (:108); replaced by 1/eigenvalues at use (.cpp:27-29) */
  float normal_flat_diag[3];             /* diag(100,100,100) (:142) */
  float normal_nonflat_diag[3];          /* diag(1,1,1) (:143) */
  float sensor_offset[16];               /* third argument of DepthImageConverter::compute (depthimageconverter.h:47) */
} pwn_hip_converter_params;

/* Parameters of PinholePointProjector + CorrespondenceFinder + Linearizer + Aligner as the
 * callers set them before Aligner::align() (pwn_simple_aligner.cpp:214-269, pwn_matcher_base.cpp:107-135). */
typedef struct pwn_hip_aligner_params {
  float K[9];                             /* projector camera matrix */
  float min_distance, max_distance;       /* projector range */
  int   rows, cols;                       /* PointProjector::setImageSize / CorrespondenceFinder::setImageSize (correspondencefinder.h:211-218) */
  float inlier_distance_threshold;        /* 0.5  (correspondencefinder.cpp:10) */
  float inlier_normal_angular_threshold;  /* cos(pi/6) (:12) */
  float flat_curvature_threshold;         /* 0.02 (:13) */
  float inlier_curvature_ratio_threshold; /* 1.3  (:14) */
  float inlier_max_chi2;                  /* 9e3  (linearizer.cpp:13) */
  int   robust_kernel;                    /* 1    (linearizer.cpp:14) */
  int   outer_iterations;                 /* 10   (aligner.cpp:19) */
  int   inner_iterations;                 /* 1    (aligner.cpp:20) */
  float reference_sensor_offset[16];      /* Aligner::setReferenceSensorOffset (aligner.h:168-171) */
  float current_sensor_offset[16];        /* Aligner::setCurrentSensorOffset (aligner.h:187-190) */
  float initial_guess[16];                /* Aligner::setInitialGuess (aligner.h:130-133) */
} pwn_hip_aligner_params;

/* What Aligner exposes after align(): T() (aligner.h:115), error() (:320), inliers() (:326),
 * totalTime() (:332), plus the per-iteration counters SURVEY.md §8(d) asks the kernels to emit. */
typedef struct pwn_hip_align_result {
  float T[16];                                 /* Aligner::T(): current -> reference frame */
  float error;                                 /* value of the LAST in-loop Linearizer::update (aligner.cpp:124) */
  int   inliers;                               /* aligner.cpp:125 */
  int   iterations;                            /* outer*inner linearizer updates executed */
  float total_time_ms;                         /* GPU time of this alignment (batch: batch time / n) */
  float chi2[PWN_HIP_MAX_ITERATIONS];          /* Linearizer::error() after each update */
  int   iter_inliers[PWN_HIP_MAX_ITERATIONS];  /* Linearizer::inliers() */
  int   iter_correspondences[PWN_HIP_MAX_ITERATIONS]; /* C_i: CorrespondenceFinder::numCorrespondences() */
  int   iter_candidates[PWN_HIP_MAX_ITERATIONS];      /* K_i: pixels with both index images >= 0 */
  int   n_reference, n_current;                /* M_r, M_c */
} pwn_hip_align_result;

/* PwnMatcherBase::MatcherResult image fields (pwn_tracker/pwn_matcher_base.h:13-22), what PwnCloser::matchFrames
 * thresholds (pwn_tracker/pwn_closer.cpp:56-58,138-141). */
typedef struct pwn_hip_match_result {
  int   image_non_zeros;               /* pixels where both finder depth images are > 0 after the uint16-mm conversion */
  int   image_outliers;                /* non_zeros - inliers */
  int   image_inliers;                 /* masked pixels whose (bit-masked) |depth difference| < threshold */
  float image_reprojection_distance;   /* sum of the differences / non_zeros */
} pwn_hip_match_result;

/* What Aligner::_computeStatistics leaves behind (aligner.cpp:127,152-199): omega() (aligner.h:314), the eigen-ratio
 * validity measures compared with rotational/translationalMinEigenRatio (aligner.cpp:128-129), the remapped mean. */
typedef struct pwn_hip_align_statistics {
  float mean[6];                       /* (t, q) of the solution as reconstructed from the sigma points */
  float omega[36];                     /* column-major 6x6 information matrix of the estimate */
  float translational_eigen_ratio;
  float rotational_eigen_ratio;
  float H[36];                         /* Linearizer::H() of the extra update at the final transform (what the statistics start from) */
  float b[6];
  float error;                         /* Linearizer::error() / inliers() of that extra update (aligner.cpp:168 overwrites the */
  int   inliers;                       /* linearizer's values, not the Aligner's cached ones) */
} pwn_hip_align_statistics;

/* Aligner::addRelativePrior / addAbsolutePrior (aligner.h:342-361, se3_prior.h) */
typedef struct pwn_hip_prior {
  int   kind;                     /* 0 = SE3RelativePrior, 1 = SE3AbsolutePrior */
  float mean[16];                 /* prior mean (column-major isometry) */
  float reference_transform[16];  /* SE3AbsolutePrior::referenceTransform; ignored for relative priors */
  float information[36];          /* column-major 6x6 */
} pwn_hip_prior;

/* ------------------------------------------------------------------ context ------------------ */
/* cf. pwn_cuda createContext(AlignerContext**, maxRef, maxCur, rows, cols) (cudaaligner.h:59).
 * max_batch = largest number of frames (convert_batch) / pairs (align_batch) per call. */
int pwn_hip_ctx_create(pwn_hip_ctx** ctx, int device, int max_rows, int max_cols, int max_batch);
int pwn_hip_ctx_destroy(pwn_hip_ctx* ctx);                       /* cf. destroyContext (cudaaligner.h:61) */
/* hip_stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = the context's own stream */
int pwn_hip_ctx_set_stream(pwn_hip_ctx* ctx, void* hip_stream);
int pwn_hip_ctx_synchronize(pwn_hip_ctx* ctx);
/* Ordering against a stream of the caller (hip_stream: a hipStream_t, NULL = the legacy default stream): everything the context queues from
 * now on runs after the work that stream holds now.  The context runs its kernels on non-blocking streams of its own, so device buffers that
 * the caller's streams touch are NOT ordered against it by themselves.  Two cases need this call:
 *   - a device buffer the context WRITES call after call (the `records` of the *_records entry points) is still being READ on the caller's
 *     stream (an all-gather of step k while step k+1 packs its records): call it after queueing the reader, before the next writing call;
 *   - a device buffer the caller's stream WRITES (a broadcast that receives a flat cloud, a kernel that fills frames) is READ by the next call
 *     (pwn_hip_cloud_import, convert*): call it after queueing the writer.
 * The other direction needs nothing: every call is complete on return (outputs valid), pwn_hip_copy_async excepted (see there). */
int pwn_hip_ctx_wait_stream(pwn_hip_ctx* ctx, void* hip_stream);
/* The other direction: what the caller queues on `hip_stream` from now on runs after everything the context has queued so far.  With
 * pwn_hip_ctx_set_enqueued_callback this lets a collective that sends a call's device `records` be queued while the call's kernels still run. */
int pwn_hip_ctx_signal_stream(pwn_hip_ctx* ctx, void* hip_stream);
/* fn(user) is called on the calling thread by every alignment batch call of the context (pwn_hip_align_batch*, pwn_hip_match_batch*,
 * pwn_hip_convert_align_batch_u16; n > 0) after ALL device work of the call has been queued -- records packed, copies back queued -- and before
 * the call waits for it.  The reference's closer loop (pwn_tracker/pwn_closer.cpp:85-111) is sequential; on a GPU the host's share of a step
 * (queueing the exchange of the results, fetching and replicating the next `current` cloud) would otherwise run with the device idle between two
 * calls.  Inside fn: other streams and other contexts freely; on THIS context only pwn_hip_ctx_signal_stream, pwn_hip_ctx_wait_stream,
 * pwn_hip_convert_export_begin / _end, pwn_hip_convert_scaled_begin / pwn_hip_convert_end -- no call that converts, aligns or waits on it.  The
 * results the call returns (and host `records`) are not valid inside fn; device `records` are valid for work ordered with
 * pwn_hip_ctx_signal_stream.  If the call has to be repeated (word 63 of the records, PWN_HIP_RECORD_FLOATS) fn runs again for the repeat.
 * fn = NULL switches it off. */
int pwn_hip_ctx_set_enqueued_callback(pwn_hip_ctx* ctx, void (*fn)(void* user), void* user);
/* Batch calls are executed in sub-batches of at most this many frames / pairs (default 64, capped by
 * max_batch), which bounds the workspace the temporaries (integral images, z-buffers) need.  Measured on
 * MI355X: larger sub-batches are faster (fewer, fuller launches) -- as long as every stream has one: a call of
 * 32 items or more is cut into a multiple of `streams` (below) equal sub-batches of at least 16 and no more than this many items (64 pairs
 * run as 2 x 32 on two streams: +7 % over 1 x 64).  Results do not depend on it. */
int pwn_hip_ctx_set_subbatch(pwn_hip_ctx* ctx, int frames, int pairs);
/* streams = 4 (default): batch calls deal their sub-batches round-robin over up to `streams` (1..4) HIP streams of the context when the
 * workspaces hold one sub-batch per stream: the short dependent kernels of one sub-batch (projection, 6x6 solve) fill the gaps of the
 * others' large ones.  Measured on MI355X, 128 VGA pairs per call: 2 streams +12 % over 1; with the one-submission step
 * (pwn_hip_convert_align_batch_u16) 4 streams x 32 pairs another +2 % over 2 x 64 (10.04-10.09 against 10.19-10.35 ms; round 2 had measured
 * 3 and 4 no better than 2 on the two-call sequence).  streams = 1: strictly serial launches (use it when profiling per-kernel durations).
 * Results are identical. */
int pwn_hip_ctx_set_concurrency(pwn_hip_ctx* ctx, int streams);
/* Storage of the point information matrices (InformationMatrix, informationmatrix.h:13; PointInformationMatrixCalculator::compute,
 * informationmatrixcalculator.cpp:9-36) of the clouds created on the context FROM NOW ON (existing clouds keep theirs;
 * pwn_hip_cloud_omega_storage tells).  The clouds of one convert batch and the current clouds of one align batch must share one.
 *   PWN_HIP_OMEGA_SYM6 (the default since round 6): the upper triangle as the reference evaluates it, 24 bytes per point (the size
 *     SURVEY.md 8(d) counts); readers -- the Linearizer (linearizer.cpp:66-67,84-88), cloud_download, the scene stage -- mirror it.  The
 *     reference's nine entries are symmetric only up to the rounding of each entry's own products, so the lower triangle differs from it by
 *     <= 1 ulp of the entry's largest term: Omega_p within 1e-6 |Omega_p| (measured 1.7e-7), chi2 / H / b within 1e-5 (measured 6e-8 from the
 *     same iterate), every other converter output bit-identical.  12 of the 30 bytes the fused correspondence + linearize pass gathers per
 *     correspondence and 12 of k_stats' 64 stored bytes less: the 128-pair step runs 11-13 % faster than with exact9.
 *   PWN_HIP_OMEGA_EXACT9: all nine entries of U diag U^t as the reference evaluates them, 36 bytes per point; EVERY converter output,
 *     the lower triangle of Omega_p included, is bit-identical to the CPU path.  The mode to select when a caller compares clouds bit for
 *     bit with pwn_core's (the parity tests do), at the cost above. */
#define PWN_HIP_OMEGA_EXACT9 0
#define PWN_HIP_OMEGA_SYM6 1
int pwn_hip_ctx_set_omega_storage(pwn_hip_ctx* ctx, int mode);
int pwn_hip_cloud_omega_storage(pwn_hip_ctx* ctx, const pwn_hip_cloud* cloud, int* mode);
/* ctx may be NULL (errors of ctx_create). Never returns NULL. cf. AlignerStatus::toString (cudaaligner.h:54) */
const char* pwn_hip_last_error_string(const pwn_hip_ctx* ctx);
/* number of HIP devices visible; does not initialise a device */
int pwn_hip_device_count(void);

/* Page-locked host memory for depth frames handed to the convert calls from the host (a grabber's ring buffer): copies from it are
 * asynchronous DMA transfers that overlap the kernels of the other stream; from ordinary (pageable) memory every frame goes through the
 * runtime's bounce buffer and blocks the calling thread.  Does not need a context.  The reference has no counterpart (its images are
 * cv::Mat on the host, depthimageconverter.h:47); any host pointer is accepted by every entry point, this is the fast kind. */
int pwn_hip_host_alloc(void** ptr, size_t bytes);
int pwn_hip_host_free(void* ptr);
/* Device buffers for callers that do not link the HIP runtime themselves (the binding of INTEGRATION.md is plain C++): frames uploaded ahead
 * of time with pwn_hip_copy are used in place by the convert calls (no staging copy).  pwn_hip_copy: any direction, ordered after the work
 * already queued on the context, complete on return. */
int pwn_hip_device_alloc(pwn_hip_ctx* ctx, void** ptr, size_t bytes);
int pwn_hip_device_free(pwn_hip_ctx* ctx, void* ptr);      /* waits for the context's queued work first; ctx may be NULL once the context that
                                                             * allocated the buffer has been destroyed (the buffer is then simply released) */
int pwn_hip_copy(pwn_hip_ctx* ctx, void* dst, const void* src, size_t bytes);
/* The same copy queued on the context's copy stream; returns at once (for page-locked host memory -- pageable memory makes it wait).  The
 * next call on the context that reads data through a caller-supplied pointer -- convert*, unproject, project_intervals, the depth-image
 * helpers, cloud_gaussians, cloud_upload, integral_image, correspondences, linearize -- and pwn_hip_copy, ctx_synchronize, device_free
 * run after every copy issued so far.  The calls that take cloud handles only (align*, match*, project, merge, voxelize, cloud_add,
 * cloud_download*) do NOT wait: that is what lets a transfer overlap an alignment.  Pattern: upload the frames of batch k+1 into a second set of device
 * buffers, align batch k meanwhile, convert batch k+1 -- the transfers disappear behind the alignment.  The caller keeps source and
 * destination untouched until one of those calls has returned. */
int pwn_hip_copy_async(pwn_hip_ctx* ctx, void* dst, const void* src, size_t bytes);

void pwn_hip_default_converter_params(pwn_hip_converter_params* p);
void pwn_hip_default_aligner_params(pwn_hip_aligner_params* p);

/* ------------------------------------------------------------------ clouds ------------------- */
/* capacity = maximum number of points (rows*cols of the images it will be converted from; for a scene that grows by Cloud::add,
 * pwn_aligner.cpp:205-208, the sum of its frames): at most 2^25 = 33 554 432 (index field of the scene stage's 64-bit z-buffer word).
 * The clouds handed to the align / match calls may hold at most 2^21 = 2 097 152 points (index field of the aligner's 32-bit z-buffer
 * word: every frame up to 1448 x 1448 pixels) -- PWN_HIP_ERR_CAPACITY beyond that; merge, voxelize, cloud_add, save / load,
 * transform_in_place and project take the large ones. */
int pwn_hip_cloud_create(pwn_hip_ctx* ctx, int capacity, pwn_hip_cloud** cloud);
int pwn_hip_cloud_destroy(pwn_hip_ctx* ctx, pwn_hip_cloud* cloud);
int pwn_hip_cloud_size(pwn_hip_ctx* ctx, const pwn_hip_cloud* cloud, int* n);    /* Cloud::points().size() */
/* Upload a host/device pwn::Cloud: points/normals n*4 floats (Point/Normal, homogeneousvector4f.h:78-93),
 * curvature n floats (Stats::curvature(), stats.h:98-103), omega_p/omega_n n*16 floats
 * (InformationMatrix 4x4, informationmatrix.h:13).  cf. pwn_cuda initComputation (cudaaligner.h:63-75). */
int pwn_hip_cloud_upload(pwn_hip_ctx* ctx, pwn_hip_cloud* cloud, int n, const float* points, const float* normals,
                         const float* curvature, const float* omega_p, const float* omega_n);
/* Any output may be NULL.  Same layouts as upload. */
int pwn_hip_cloud_download(pwn_hip_ctx* ctx, const pwn_hip_cloud* cloud, float* points, float* normals,
                           float* curvature, float* omega_p, float* omega_n);
/* Stats of the last convert of this cloud (only if the convert ran with keep_stats != 0): per point
 * eigenvectors+mean as a column-major 4x4 (Stats, stats.h:13), eigenvalues[3], n (stats.h:30). */
int pwn_hip_cloud_download_stats(pwn_hip_ctx* ctx, const pwn_hip_cloud* cloud, float* stats, float* eigenvalues, int* npoints);
/* Cloud::transformInPlace (cloud.cpp:173-186): points, normals, both information matrices (T Omega T^t, informationmatrix.h:111-121), the
 * Stats when the cloud carries them (m * S, stats.h:125-131) and the Gaussians (gaussian3.h:65-73); skipped when T is the identity (:176) */
int pwn_hip_cloud_transform_in_place(pwn_hip_ctx* ctx, pwn_hip_cloud* cloud, const float T[16]);
/* A cloud as ONE flat buffer, host or device: what replicates a cloud to the other GPUs of a node.  PwnCloser::processPartition
 * (pwn_tracker/pwn_closer.cpp:85-111) matches the cloud of ONE `current` frame against every cached cloud of the other partition; sharded over
 * GPUs, the others stay where they were converted and `current` travels ("replicate current's cloud to all GPUs; one
 * broadcast") -- export on the rank that converted it, one broadcast of the buffer, import on every rank.  The reference's own whole-cloud
 * form is Cloud::save / load (cloud.cpp:11-136; pwn_hip_cloud_save / _load), which goes through a file and drops the information matrices.
 * Content: 256-byte header, points (12 B), normal + curvature (16 B), the point information matrices as stored (exact9 / sym6), the class
 * matrices of the normal information matrix (or its full planes for uploaded clouds), and the converter's index image with the projector it
 * belongs to when the cloud carries one (so that an imported cloud takes the same projection shortcuts as the original: bitwise the same
 * alignments).  Not carried: Stats, Gaussians (the scene stage's data; Aligner::align reads neither).
 * export_bound: buffer size that holds any cloud of that capacity.  export: *written (optional) = bytes used; PWN_HIP_ERR_CAPACITY when
 * dst_bytes is too small (*written then = bytes needed); dst = NULL: size query only.  import: the destination cloud must have been created with
 * the same omega storage and capacity >= the flat cloud's points.  Both are complete on return; with device buffers that another stream
 * produces / consumes see pwn_hip_ctx_wait_stream. */
size_t pwn_hip_cloud_export_bound(int capacity, int omega_storage, int index_pixels, int with_omega_n);
int pwn_hip_cloud_export(pwn_hip_ctx* ctx, const pwn_hip_cloud* cloud, void* dst, size_t dst_bytes, size_t* written);
int pwn_hip_cloud_import(pwn_hip_ctx* ctx, pwn_hip_cloud* cloud, const void* src, size_t src_bytes);

/* ------------------------------------------------------------------ input conditioning ------- */
/* DepthImage_convert_16UC1_to_32FC1 (pwn_static.cpp:54-68) */
int pwn_hip_depth_u16_to_f32(pwn_hip_ctx* ctx, const uint16_t* src, float* dst, int n, float scale);
/* DepthImage_convert_32FC1_to_16UC1 (pwn_static.cpp:38-52) */
int pwn_hip_depth_f32_to_u16(pwn_hip_ctx* ctx, const float* src, uint16_t* dst, int n, float scale);
/* DepthImage_scale (pwn_static.cpp:5-36); dst holds (rows/step)*(cols/step) floats */
int pwn_hip_depth_scale(pwn_hip_ctx* ctx, const float* src, int rows, int cols, int step, float max_depth_cov, float* dst);

/* ------------------------------------------------------------------ converter stages --------- */
/* PinholePointProjector::unProject(points, gaussians, indexImage, depthImage) (pinholepointprojector.cpp:93-133)
 * with the projector transform T (identity inside the converter).  Fills cloud points (normals etc. are
 * reset to "invalid").  The per-point sensor Gaussians are not produced (only Merger consumes them). */
int pwn_hip_unproject(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float T[16], const float* depth,
                      int rows, int cols, pwn_hip_cloud* cloud, int* index_image);
/* PinholePointProjector::projectIntervals (pinholepointprojector.cpp:135-147) */
int pwn_hip_project_intervals(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float* depth, int rows, int cols,
                              int* interval_image);
/* PointIntegralImage::compute (pointintegralimage.cpp:7-44): the 10 unique channels
 * (x,y,z,n,xx,xy,xz,yy,yz,zz) as planes out[10][rows][cols]. Diagnostic / parity entry point. */
int pwn_hip_integral_image(pwn_hip_ctx* ctx, const int* index_image, const pwn_hip_cloud* cloud, int rows, int cols, float* out);
/* DepthImageConverterIntegralImage::compute(cloud, depthImage, sensorOffset)
 * (depthimageconverterintegralimage.cpp:15-55).  index_image / interval_image: optional outputs
 * (DepthImageConverter::indexImage(), depthimageconverter.h:111). keep_stats: also keep per-point Stats. */
int pwn_hip_convert(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float* depth, int rows, int cols,
                    pwn_hip_cloud* cloud, int* index_image, int* interval_image, int keep_stats);
/* PwnMatcherBase::makeCloud's data path in one call (pwn_tracker/pwn_matcher_base.cpp:71-79): DepthImage_scale(depth, step,
 * max_depth_cov) followed by the converter on the (rows/step) x (cols/step) image, without leaving the device.
 * p->K must already be the scaled camera matrix. */
int pwn_hip_convert_scaled(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float* depth, int rows, int cols, int step,
                           float max_depth_cov, pwn_hip_cloud* cloud);
/* The same conversion in two halves, for callers that know their next frame before they are done with the current one (a recorded or
 * buffered stream through PwnTracker::processFrame, pwn_tracker/pwn_tracker.cpp:115: makeCloud of frame k+1 does not depend on the
 * alignment of frame k).  _begin returns at once: the frame is converted by a helper thread on streams and workspaces of its own (created
 * on first use: one more set of single-frame workspaces), next to whatever the caller runs on `ctx` meanwhile.  _end waits for it and
 * returns what pwn_hip_convert_scaled would have returned; the cloud then holds the same bits.  Between the two calls `depth` must stay
 * valid and unchanged and `cloud` must not be passed to any other entry point (pwn_hip_cloud_destroy and pwn_hip_ctx_destroy wait for a
 * conversion in flight).  One conversion in flight per context. */
int pwn_hip_convert_scaled_begin(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float* depth, int rows, int cols, int step,
                                 float max_depth_cov, pwn_hip_cloud* cloud);
int pwn_hip_convert_end(pwn_hip_ctx* ctx, pwn_hip_cloud* cloud);
/* The look-ahead of a sharded PwnCloser::processPartition (pwn_tracker/pwn_closer.cpp:85-111): the outer loop hands the closer one `current`
 * keyframe after the other, and _cache->get(current) of keyframe k+1 (:92-93; PwnCache::makeCloud, pwn_tracker_cache.cpp:24-51) depends on nothing
 * keyframe k's matches produce.  _begin returns at once; a helper thread converts the raw uint16 frame (depth = depth_scale * raw, as
 * pwn_hip_convert_batch_u16) into `cloud` and then, with flat_dst != NULL, writes the cloud's flat form (pwn_hip_cloud_export) into flat_dst (device
 * or host; at least pwn_hip_cloud_export_bound(min(capacity, rows*cols), storage, rows*cols, 0) bytes, checked by _begin) -- what the rank that owns
 * `current` broadcasts while every rank still matches the previous keyframe.  _end waits for the job, returns its status and reports the bytes
 * written (what to broadcast) and the job's wall time on the helper thread in milliseconds; both optional.  Same rules as
 * pwn_hip_convert_scaled_begin between the two calls (frame, cloud AND flat_dst belong to the job; one job in flight per context); the cloud holds
 * the bits pwn_hip_convert_batch_u16 gives, the buffer the bytes pwn_hip_cloud_export gives. */
int pwn_hip_convert_export_begin(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const uint16_t* raw_frame, float depth_scale, int rows, int cols,
                                 pwn_hip_cloud* cloud, void* flat_dst, size_t flat_bytes);
int pwn_hip_convert_export_end(pwn_hip_ctx* ctx, pwn_hip_cloud* cloud, size_t* written, float* job_ms);
/* n independent frames of equal size in one call.  depth[i] -> clouds[i].
 * depth_frames: n pointers (host array) to rows*cols floats each (each host or device). */
int pwn_hip_convert_batch(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float* const* depth_frames,
                          int n, int rows, int cols, pwn_hip_cloud* const* clouds);
/* Same, from raw uint16 millimetre frames: fuses DepthImage_convert_16UC1_to_32FC1 (scale) in front. */
int pwn_hip_convert_batch_u16(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const uint16_t* const* raw_frames,
                              float depth_scale, int n, int rows, int cols, pwn_hip_cloud* const* clouds);

/* ------------------------------------------------------------------ aligner stages ----------- */
/* PinholePointProjector::project(indexImage, depthImage, points) (pinholepointprojector.cpp:33-66) with
 * projector transform T.  Untouched depth pixels are FLT_MAX, index -1; ties keep the lowest index. */
int pwn_hip_project(pwn_hip_ctx* ctx, const float K[9], const float T[16], float min_distance, float max_distance,
                    int rows, int cols, const pwn_hip_cloud* cloud, int* index_image, float* depth_image);
/* CorrespondenceFinder::compute (correspondencefinder.cpp:20-118, single-thread canonical order).
 * correspondences: rows*cols pairs (referenceIndex,currentIndex) -- the first *n_correspondences are
 * valid, in row-major pixel order, the rest are (-1,-1) (correspondencefinder.cpp:116-117). */
int pwn_hip_correspondences(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, const pwn_hip_cloud* reference,
                            const pwn_hip_cloud* current, const int* reference_index_image, const int* current_index_image,
                            const float T[16], int* correspondences, int* n_correspondences, int* n_candidates);
/* Linearizer::update (linearizer.cpp:17-115) on an explicit correspondence list with transform T (= invT).
 * H: column-major 6x6 (Linearizer::H()), b: 6 (Linearizer::b()). cf. pwn_cuda getHb (cudaaligner.h:79). */
int pwn_hip_linearize(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, const pwn_hip_cloud* reference,
                      const pwn_hip_cloud* current, const int* correspondences, int n_correspondences, const float T[16],
                      float* H, float* b, float* error, int* inliers);
/* Aligner::align (aligner.cpp:49-125): project current once, then outer x inner Gauss-Newton
 * iterations of {project reference, find correspondences, linearize, damped LDLT solve, update}.
 * The whole loop runs on the device without host round trips. Priors and _computeStatistics
 * (aligner.cpp:96-108,127) are not part of this entry point. */
int pwn_hip_align(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, const pwn_hip_cloud* reference,
                  const pwn_hip_cloud* current, pwn_hip_align_result* result);
/* CorrespondenceFinder::{reference,current}{Index,Depth}Image() after the last pwn_hip_align
 * (correspondencefinder.h:99-117; read by pwn_tracker/pwn_matcher_base.cpp:153-155). Any may be NULL.
 * The aligner's z-buffers keep point indices only; the depth images are recomputed from the two clouds' points, so this call (and
 * pwn_hip_match_score) is valid until either cloud of that alignment is destroyed or gets new content (then: INVALID_ARGUMENT). */
int pwn_hip_align_images(pwn_hip_ctx* ctx, int* reference_index, float* reference_depth, int* current_index, float* current_depth);
/* n independent alignments (the loop-closure candidate batch, pwn_tracker/pwn_closer.cpp:92-111).
 * p is shared by all pairs; initial_guesses: n*16 floats or NULL (= p->initial_guess for all). */
int pwn_hip_align_batch(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, int n, pwn_hip_cloud* const* references,
                        pwn_hip_cloud* const* currents, const float* initial_guesses, pwn_hip_align_result* results);

/* ------------------------------------------------------------------ matcher ---------- */
/* Post-alignment depth-agreement score of PwnMatcherBase::matchClouds (pwn_tracker/pwn_matcher_base.cpp:153-182) on the
 * finder depth images of the last pwn_hip_align: DepthImage_convert_32FC1_to_16UC1 of both (scale 1000, FLT_MAX -> 0),
 * mask, |difference|, counts.  frame_inlier_depth_threshold: _frameInlierDepthThreshold (50, pwn_matcher_base.cpp:13). */
int pwn_hip_match_score(pwn_hip_ctx* ctx, float frame_inlier_depth_threshold, pwn_hip_match_result* out);
/* pwn_hip_align_batch + the score of every pair (the loop-closure check of PwnCloser::processPartition,
 * pwn_tracker/pwn_closer.cpp:92-111, without the host loop).  scores: n records. */
int pwn_hip_match_batch(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, int n, pwn_hip_cloud* const* references,
                        pwn_hip_cloud* const* currents, const float* initial_guesses, float frame_inlier_depth_threshold,
                        pwn_hip_align_result* results, pwn_hip_match_result* scores);

/* Everything Aligner::align() does for n pairs in one call: the Gauss-Newton loop, optionally the matcher's depth-agreement
 * score (scores != NULL) and optionally Aligner::_computeStatistics (statistics != NULL; aligner.cpp:127,152-199: one more
 * linearizer update at the final transform on the finder's last correspondences, then the 6x6 covariance / unscented math
 * on the host).  pwn_hip_align_batch / pwn_hip_match_batch are this call with the optional outputs left NULL. */
int pwn_hip_align_batch_ex(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, int n, pwn_hip_cloud* const* references,
                           pwn_hip_cloud* const* currents, const float* initial_guesses, pwn_hip_align_result* results,
                           float frame_inlier_depth_threshold, pwn_hip_match_result* scores, pwn_hip_align_statistics* statistics);
/* The result of a pair as ONE fixed-size record of PWN_HIP_RECORD_FLOATS floats (256 bytes) -- what the ranks of a sharded candidate batch
 * exchange (the reference runs the candidates of PwnCloser::process one after the other on one Aligner and keeps MatcherResult per pair,
 * pwn_tracker/pwn_closer.cpp:92-111):
 *   [0:16] T (column-major)   [16] chi2 of the last iteration   [17] its inliers   [18] iterations   [19] pair id
 *   [20:30] chi2_i   [30:40] inliers_i   [40:50] correspondences C_i   [50:60] candidates K_i   (first 10 iterations; 0 beyond the last)
 *   [60] points of the reference cloud   [61] of the current cloud   [62] iterations carried in the traces
 *   [63] 0; 1 = a projection of the call gave up on a pixel and the call is being repeated (the records a call RETURNS always carry 0: only a reader
 *        that takes device records inside pwn_hip_ctx_set_enqueued_callback can meet a 1, and drops those records)
 * Counts travel as float (exact below 2^24).  The records are written by a kernel from the pairs' device state: `records` may be a DEVICE
 * buffer (n * PWN_HIP_RECORD_FLOATS floats; e.g. the tensor an all-gather sends -- no trip through the host) or host memory.
 * pair_ids (host, may be NULL: then record i carries first_pair_id + i).  results may be NULL when only the records are wanted.
 * ORDERING: the kernel writes a device `records` buffer on the context's own stream.  If a stream of the caller still reads the buffer from
 * the previous call (an all-gather queued after it), call pwn_hip_ctx_wait_stream(ctx, that stream) before this call -- or alternate two buffers. */
#define PWN_HIP_RECORD_FLOATS 64
int pwn_hip_align_batch_records(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, int n, pwn_hip_cloud* const* references,
                                pwn_hip_cloud* const* currents, const float* initial_guesses, const int* pair_ids, int first_pair_id,
                                pwn_hip_align_result* results, float* records);
/* pwn_hip_match_batch whose results (also) leave as records of PWN_HIP_MATCH_RECORD_FLOATS floats (288 bytes): the 64 words of the alignment
 * record, then PwnMatcherBase::MatcherResult's image fields (pwn_tracker/pwn_matcher_base.cpp:175-181)
 *   [64] image_nonZeros   [65] image_outliers   [66] image_inliers   [67] image_reprojectionDistance   [68:72] 0
 * -- everything PwnCloser::matchFrames thresholds and stores in a PwnCloserRelation (pwn_closer.cpp:138-151), written on the device.  `records`:
 * device or host, n * PWN_HIP_MATCH_RECORD_FLOATS floats; results and scores may be NULL.  A device `records` buffer that another stream still
 * reads from the previous call: pwn_hip_ctx_wait_stream. */
#define PWN_HIP_MATCH_RECORD_FLOATS 72
int pwn_hip_match_batch_records(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, int n, pwn_hip_cloud* const* references,
                                pwn_hip_cloud* const* currents, const float* initial_guesses, float frame_inlier_depth_threshold,
                                const int* pair_ids, int first_pair_id, pwn_hip_align_result* results, pwn_hip_match_result* scores,
                                float* records);
/* One candidate batch from raw frames as ONE submission: per pair i, DepthImageConverter::compute of ref_frames[i] -> references[i] and
 * cur_frames[i] -> currents[i] (PwnMatcherBase::makeCloud, pwn_matcher_base.cpp:77-85), then Aligner::align of the pair
 * (pwn_matcher_base.cpp:120-128).  The conversion of a sub-batch's frames is queued in front of its alignment on the same stream, so one
 * sub-batch converts while the other aligns and nothing waits for the host between the two halves.  Results are bit for bit those of
 * pwn_hip_convert_batch_u16 followed by pwn_hip_align_batch.  The 2 n clouds must be distinct; frames: uint16 millimetres, host or device. */
int pwn_hip_convert_align_batch_u16(pwn_hip_ctx* ctx, const pwn_hip_converter_params* converter, const pwn_hip_aligner_params* aligner, int n,
                                    const uint16_t* const* ref_frames, const uint16_t* const* cur_frames, float depth_scale, int rows, int cols,
                                    pwn_hip_cloud* const* references, pwn_hip_cloud* const* currents, const float* initial_guesses,
                                    const int* pair_ids, int first_pair_id, pwn_hip_align_result* results, float* records);
/* Aligner::align with priors (aligner.cpp:96-108).  The prior terms (numeric Jacobians, se3_prior.cpp:8-52) are 6x6 host math
 * that changes the normal equations of every iteration, so this entry point keeps the reference's host-driven loop: per
 * iteration the GPU projects, finds correspondences and reduces H, b; the host adds damping + priors, solves and updates.
 * n_priors = 0 is the device-resident loop of pwn_hip_align. */
int pwn_hip_align_with_priors(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, const pwn_hip_cloud* reference,
                              const pwn_hip_cloud* current, int n_priors, const pwn_hip_prior* priors, pwn_hip_align_result* result);
/* The same with Aligner::_computeStatistics (aligner.cpp:127): the reference runs it after every align(), priors or not, on
 * Linearizer::H() + I at the final transform -- the prior terms are not part of that H (aligner.cpp:165-170).  statistics may
 * be NULL (= pwn_hip_align_with_priors); n_priors = 0 is pwn_hip_align_batch_ex for one pair. */
int pwn_hip_align_with_priors_ex(pwn_hip_ctx* ctx, const pwn_hip_aligner_params* p, const pwn_hip_cloud* reference,
                                 const pwn_hip_cloud* current, int n_priors, const pwn_hip_prior* priors, pwn_hip_align_result* result,
                                 pwn_hip_align_statistics* statistics);
/* the host-side part alone: H = Linearizer::H() at the final transform, T = Aligner::T() */
void pwn_hip_compute_statistics(const float H[36], const float T[16], float mean[6], float omega[36],
                                float* translational_eigen_ratio, float* rotational_eigen_ratio);

/* ---- scene maintenance (what follows the registration path in pwn_aligner.cpp:205-208 / pwn_merger.cpp:91-94) ---------------- */
/* The Gaussian half of PinholePointProjector::unProject(points, gaussians, index, depth) (pinholepointprojector.cpp:104-123; baseline
 * and alpha defaults :10-11) for a cloud made by pwn_hip_convert from the same depth image and parameters, followed by
 * Gaussian3fVector::transformInPlace(sensor_offset) (cloud.cpp:180).  The reference computes them inside every
 * DepthImageConverter::compute; only Merger::merge consumes them, so here they are produced on request. */
int pwn_hip_cloud_gaussians(pwn_hip_ctx* ctx, const pwn_hip_converter_params* p, const float* depth, int rows, int cols,
                            pwn_hip_cloud* cloud, float baseline, float alpha);
int pwn_hip_cloud_num_gaussians(pwn_hip_ctx* ctx, const pwn_hip_cloud* cloud, int* n);       /* Cloud::gaussians().size() */
/* host arrays: mean n*3, cov n*9 (column-major 3x3), info_vec n*3, info n*9, flags n (1 = moments valid, 2 = information form valid:
 * the two lazily evaluated forms of basemath/gaussian.h:9-94); any pointer may be NULL */
int pwn_hip_cloud_download_gaussians(pwn_hip_ctx* ctx, const pwn_hip_cloud* cloud, float* mean, float* cov, float* info_vec, float* info,
                                     int* flags);
/* Cloud::add (cloud.cpp:145-171): append a copy of `src` transformed by T (Cloud::transformInPlace) to `dst` */
int pwn_hip_cloud_add(pwn_hip_ctx* ctx, pwn_hip_cloud* dst, const pwn_hip_cloud* src, const float T[16]);
/* Merger::merge (merger.cpp:15-119).  K, min/max_distance: the projector of the merger's DepthImageConverter (merger.cpp:20-23),
 * T: its pose; thresholds: merger.cpp:6-8.  collapsed (optional, host, old size) receives Merger::_collapsedIndices. */
int pwn_hip_merge(pwn_hip_ctx* ctx, pwn_hip_cloud* cloud, const float K[9], const float T[16], float min_distance, float max_distance,
                  int rows, int cols, float distance_threshold, float normal_threshold, float max_point_depth, int* new_size,
                  int* collapsed);
/* VoxelCalculator::compute (voxelcalculator.cpp:15-73): the first point of every voxel survives, output sorted by voxel indices
 * (the intended lexicographic order; the reference's IndexComparator, voxelcalculator.h:41-48, is not a strict weak ordering).
 * kept (optional, host) receives the original indices of the survivors in output order. */
int pwn_hip_voxelize(pwn_hip_ctx* ctx, pwn_hip_cloud* cloud, float resolution, int* new_size, int* kept);
/* Cloud::save / Cloud::load (cloud.cpp:11-136): "PWNCLOUD n binary" header, t2v(T) line, POINTWITHSTATS text records or the
 * reference's 176-byte binary records (object layout of Point / Normal / Stats, non-data bytes zeroed) */
int pwn_hip_cloud_save(pwn_hip_ctx* ctx, const pwn_hip_cloud* cloud, const char* filename, const float T[16], int step, int binary);
int pwn_hip_cloud_load(pwn_hip_ctx* ctx, pwn_hip_cloud* cloud, const char* filename, float T_out[16]);

/* ------------------------------------------------------------------ helpers ------------------ */
/* PinholePointProjector::_updateMatrices (pinholepointprojector.cpp:17-31): KRt, iKRt (4x4), iK (3x3) */
void pwn_hip_projector_matrices(const float K[9], const float T[16], float KRt[16], float iKRt[16], float iK[9]);
/* The single-point forms of the projector (host code; the very expressions the kernels evaluate, so a point projected here lands on the pixel
 * pwn_hip_project gives it):
 *   PinholePointProjector::project(x, y, f, p)      (pinholepointprojector.h:174, _project :224-233): 1 = valid (depth inside [min, max]); the image
 *                                                   bounds are the caller's test, as in the reference (pinholepointprojector.cpp:51-55)
 *   PinholePointProjector::unProject(p, x, y, d)    (:187, _unProject :246-251): 1 = valid; x = column, y = row (pinholepointprojector.cpp:112)
 *   PinholePointProjector::projectInterval(x,y,d,R) (:200, _projectInterval :264-274): -1 for a depth outside [min, max] */
int pwn_hip_project_point(const float K[9], const float T[16], float min_distance, float max_distance, const float p[3], int* x, int* y, float* d);
int pwn_hip_unproject_pixel(const float K[9], const float T[16], float min_distance, float max_distance, int x, int y, float d, float p[3]);
int pwn_hip_project_interval(const float K[9], float min_distance, float max_distance, float d, float world_radius);
/* Eigen::Isometry3f::inverse() and Isometry3f * Isometry3f with the evaluation order the CPU path has (host code) */
void pwn_hip_iso_inverse(const float T[16], float out[16]);
void pwn_hip_iso_mul(const float A[16], const float B[16], float out[16]);
/* bm_se3.h:37-52 */
void pwn_hip_v2t(const float v[6], float T[16]);
void pwn_hip_t2v(const float T[16], float v[6]);
/* Matrix6f::ldlt().solve(b) as aligner.cpp:110 calls it (column-major H, lower triangle read): the host compilation of the
 * function k_solve_update runs on the device */
void pwn_hip_ldlt_solve6(const float H[36], const float b[6], float x[6]);
/* per-kernel device time (ms) of the stages of the last batch/single call, for bench.py:
 * names: "unproject","integral","integral_rows","integral_cols","stats","project_cur","project_ref","corr_linearize","solve",
 * "statistics","match_score" ("project": the stand-alone pwn_hip_project).  launches = timed launch groups (one per sub-batch). */
int pwn_hip_last_stage_ms(pwn_hip_ctx* ctx, const char* stage, float* ms, int* launches);
/* what this GPU's HBM delivers, for the roofline report (SURVEY 8(d) asks for the measured figure next to the 8 TB/s spec):
 * float4 streaming read and device-to-device copy of `bytes` (use >= 1 GiB: the Infinity Cache holds 256 MiB), best of 5,
 * GB/s; the copy counts bytes read + written.  Allocates and frees 2 x bytes. */
int pwn_hip_measure_hbm(pwn_hip_ctx* ctx, size_t bytes, float* read_gbps, float* copy_gbps);
/* enable/disable hipEvent timing around every kernel launch (adds host overhead; default off) */
int pwn_hip_set_profiling(pwn_hip_ctx* ctx, int enabled);

#ifdef __cplusplus
}
#endif
#endif
