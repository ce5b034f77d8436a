/*
 * pwn_oracle.h -- C interface of the CPU ORACLE for the PWN dense-registration path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (g2o_frontend_amd/, include/)
 * includes, links or calls this.  Allowed users: tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg.
 *
 * PARITY UNPINNED: the reference (grisetti/g2o_frontend, pwn_core) cannot be built
 * here (needs Eigen3 + OpenCV, neither installed, no network) and its own tests pin
 * no numerical result on this path (all pwn_test targets are disabled print-only
 * CLIs).  This oracle is therefore a line-by-line restatement of the reference
 * sources, with the Eigen routines it depends on restated from the published
 * Eigen 3.2.x/3.3 algorithms (SSE3 build, no FMA contraction, sequential
 * left-to-right inner products).  Each function cites the reference file:line.
 *
 * All matrices are COLUMN-MAJOR float (Eigen default): M[r + 4*c].
 * Images are row-major [rows][cols].
 */
#ifndef PWN_ORACLE_H
#define PWN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Parameters of DepthImageConverterIntegralImage and its collaborators.
 * Defaults / conf values: SURVEY.md App. C. */
typedef struct orc_converter_params {
  float K[9];                 /* camera matrix, column-major 3x3 (pinholepointprojector.h:51) */
  float min_distance;         /* pointprojector.cpp:9   */
  float max_distance;         /* pointprojector.cpp:10  */
  float world_radius;         /* statscalculatorintegralimage.cpp:7  */
  int   min_image_radius;     /* :9  */
  int   max_image_radius;     /* :8  */
  int   min_points;           /* :10 */
  float stats_curvature_threshold;  /* :11 */
  float point_info_curvature_threshold;   /* informationmatrixcalculator.h:109 */
  float normal_info_curvature_threshold;  /* informationmatrixcalculator.h:144 */
  float point_flat_diag[3];      /* diag(1000,1,1)   informationmatrixcalculator.h:107 */
  float point_nonflat_diag[3];   /* overwritten by 1/eigenvalues at use: .cpp:27-29 */
  float normal_flat_diag[3];     /* diag(100,100,100) .h:142 */
  float normal_nonflat_diag[3];  /* diag(1,1,1)       .h:143 */
  float sensor_offset[16];       /* column-major 4x4 isometry */
} orc_converter_params;

/* Parameters of CorrespondenceFinder + Linearizer + Aligner. */
typedef struct orc_aligner_params {
  float K[9];
  float min_distance, max_distance;
  int   rows, cols;                       /* projector / finder image size */
  float inlier_distance_threshold;        /* correspondencefinder.cpp:10 */
  float inlier_normal_angular_threshold;  /* :12 (a cosine) */
  float flat_curvature_threshold;         /* :13 */
  float inlier_curvature_ratio_threshold; /* :14 */
  float inlier_max_chi2;                  /* linearizer.cpp:13 */
  int   robust_kernel;                    /* linearizer.cpp:14 */
  int   outer_iterations;                 /* aligner.cpp:19 */
  int   inner_iterations;                 /* aligner.cpp:20 */
  float reference_sensor_offset[16];
  float current_sensor_offset[16];
  float initial_guess[16];
  int   accumulate_fp64;   /* 0: fp32 serial sums exactly as linearizer.cpp:56-89 (reference-faithful)
                              1: same per-term fp32 arithmetic, sums kept in fp64 (parity target) */
} orc_aligner_params;

/* Per-iteration trace of Aligner::align (one entry per linearizer update). */
typedef struct orc_iter_trace {
  int   K;            /* pixels with both indices >= 0 (correspondencefinder.cpp:60) */
  int   C;            /* correspondences emitted */
  int   inliers;      /* linearizer.cpp:80 */
  float chi2;         /* linearizer error(), fp32 value */
  double chi2_fp64;   /* same terms accumulated in double */
  float H[36];        /* column-major 6x6, linearizer H (no damping) */
  float b[6];
  float T_before[16]; /* _T at the start of the outer iteration */
} orc_iter_trace;

typedef struct orc_cloud orc_cloud;   /* opaque host cloud */

void orc_default_converter_params(orc_converter_params* p);
void orc_default_aligner_params(orc_aligner_params* p);

/* pwn_static.cpp:54-68 */
void orc_convert_16u_to_32f(const uint16_t* src, float* dst, int n, float scale);
/* pwn_static.cpp:38-52 */
void orc_convert_32f_to_16u(const float* src, uint16_t* dst, int n, float scale);
/* pwn_static.cpp:5-36; dst has (rows/step)*(cols/step) elements */
void orc_depth_scale(const float* src, int rows, int cols, int step, float max_depth_cov, float* dst);

/* pinholepointprojector.cpp:17-31 ; outputs column-major 4x4 / 3x3 */
void orc_projector_matrices(const float K[9], const float T[16], float KRt[16], float iKRt[16], float iK[9]);

orc_cloud* orc_cloud_create(void);
void orc_cloud_destroy(orc_cloud* c);
int  orc_cloud_size(const orc_cloud* c);
/* copies; any pointer may be NULL.  points/normals: n*4 floats; curvature n; stats n*16 (column-major 4x4);
 * eigenvalues n*3; npoints n; omega_p / omega_n: n*16 column-major 4x4 */
void orc_cloud_get(const orc_cloud* c, float* points, float* normals, float* curvature,
                   float* stats, float* eigenvalues, int* npoints, float* omega_p, float* omega_n);
/* build a cloud from raw arrays (stats are synthesised so that curvature() returns `curvature`) */
void orc_cloud_set(orc_cloud* c, int n, const float* points, const float* normals,
                   const float* curvature, const float* omega_p, const float* omega_n);

/* pinholepointprojector.cpp:93-133 (points + index image; gaussians are not produced) */
int  orc_unproject(const orc_converter_params* p, const float* depth, int rows, int cols,
                   float* points /* rows*cols*4 */, int* index_image);
/* pinholepointprojector.cpp:135-147 */
void orc_project_intervals(const orc_converter_params* p, const float* depth, int rows, int cols,
                           int* interval_image);
/* pointintegralimage.cpp:7-44.  out: 10 planes [10][rows][cols] of the unique sums
 * (x,y,z,n,xx,xy,xz,yy,yz,zz) after both prefix passes */
void orc_integral_image(const int* index_image, const float* points, int rows, int cols, float* out);

/* depthimageconverterintegralimage.cpp:15-55.  index_image/interval_image optional outputs */
void orc_convert(const orc_converter_params* p, const float* depth, int rows, int cols,
                 orc_cloud* cloud, int* index_image, int* interval_image);

/* pinholepointprojector.cpp:33-66 with transform T (projector pose) */
void orc_project(const float K[9], const float T[16], float min_distance, float max_distance,
                 int rows, int cols, const float* points, int n, int* index_image, float* depth_image);

/* correspondencefinder.cpp:20-118 (canonical single-thread semantics).  corr: up to rows*cols pairs
 * (ref,cur).  returns C; *K_out = #pixels with both indices valid */
int  orc_correspondences(const orc_aligner_params* p, const orc_cloud* ref, const orc_cloud* cur,
                         const int* ref_index, const int* cur_index, const float T[16],
                         int* corr, int* K_out);

/* linearizer.cpp:17-115 (single-thread). H col-major 6x6 */
void orc_linearize(const orc_aligner_params* p, const orc_cloud* ref, const orc_cloud* cur,
                   const int* corr, int C, const float T[16],
                   float* H, float* b, float* chi2, double* chi2_fp64, int* inliers);

/* aligner.cpp:49-125 (priors and _computeStatistics excluded).  trace: outer*inner entries or NULL.
 * Optional image outputs are the finder's images after the last iteration. */
void orc_align(const orc_aligner_params* p, const orc_cloud* ref, const orc_cloud* cur,
               float T_out[16], float* error_out, int* inliers_out, orc_iter_trace* trace,
               int* ref_index_out, float* ref_depth_out, int* cur_index_out, float* cur_depth_out);

/* PwnMatcherBase::matchClouds post-align scoring (pwn_tracker/pwn_matcher_base.cpp:153-182) on the finder's depth images */
void orc_match_score(const float* ref_depth, const float* cur_depth, int n, float threshold, int* non_zeros, int* outliers,
                     int* inliers, float* reprojection_distance);

/* Aligner::clearPriors / addRelativePrior (kind 0) / addAbsolutePrior (kind 1): priors used by the following orc_align calls
 * (aligner.cpp:34-47, 96-108; se3_prior.cpp) */
void orc_clear_priors(void);
void orc_add_prior(int kind, const float mean[16], const float reference_transform[16], const float information[36]);
/* Aligner::_computeStatistics (aligner.cpp:152-199): H = linearizer H at the final transform, T = Aligner::_T */
void orc_compute_statistics(const float H[36], const float T[16], float mean[6], float omega[36], float* translational_ratio, float* rotational_ratio);
/* the same after the last orc_align (runs the 11th linearizer update on the finder's last correspondences) */
void orc_align_statistics(const orc_aligner_params* p, const orc_cloud* ref, const orc_cloud* cur, const float T[16], float H_out[36],
                          float mean[6], float omega[36], float* translational_ratio, float* rotational_ratio);
/* Isometry3f::inverse / product (used by the tracker harness of the tests) */
void orc_iso_inverse(const float T[16], float out[16]);
void orc_iso_mul(const float A[16], const float B[16], float out[16]);
/* pwn_tracker/pwn_tracker.cpp:154-159: globalT.linear() -= 0.5 * R * (R^T R - I) (every 50th frame of PwnTracker::processFrame) */
void orc_reorthonormalize(const float T[16], float out[16]);
/* bm_se3.h:9-52 exposed for unit tests */
void orc_v2t(const float v[6], float T[16]);
void orc_t2v(const float T[16], float v[6]);
/* Eigen computeDirect restatement exposed for unit tests: A col-major 3x3 symmetric (lower read) */
void orc_eigen3(const float A[9], float evals[3], float evecs[9]);
/* pivoted LDLT solve of a 6x6 (col-major) system, fp32 */
void orc_ldlt_solve6(const float H[36], const float b[6], float x[6]);

/* ---- scene maintenance ---- */
/* orc_convert also fills the cloud's sensor-noise Gaussians (pinholepointprojector.cpp:104-123); off by default so that the timed
 * CPU baseline does the same work as the GPU path (which produces them on request) */
void orc_set_gaussians(int enabled, float baseline, float alpha);
int  orc_cloud_num_gaussians(const orc_cloud* c);
/* per Gaussian: mean[3], cov[9] column-major, info_vec[3], info[9], flags (1 = moments valid, 2 = information form valid) */
void orc_cloud_get_gaussians(const orc_cloud* c, float* mean, float* cov, float* info_vec, float* info, int* flags);
/* Cloud::transformInPlace (cloud.cpp:173-186) incl. Gaussian3fVector::transformInPlace (gaussian3.h:65-73) */
void orc_cloud_transform_in_place(orc_cloud* c, const float T[16]);
/* Cloud::add (cloud.cpp:145-171) */
void orc_cloud_add(orc_cloud* dst, const orc_cloud* src, const float T[16]);
/* Merger::merge (merger.cpp:15-119); collapsed (optional) = _collapsedIndices; returns the new cloud size */
int  orc_merge(orc_cloud* c, const float K[9], const float T[16], float min_distance, float max_distance, int rows, int cols,
               float distance_threshold, float normal_threshold, float max_point_depth, int* collapsed);
/* VoxelCalculator::compute (voxelcalculator.cpp:15-73); literal = 1: std::map with the reference's comparator (not a strict weak
 * order), literal = 0: the intended lexicographic order.  kept (optional) = original indices of the survivors in output order */
int  orc_voxelize(orc_cloud* c, float resolution, int literal, int* kept);
/* Cloud::save / Cloud::load (cloud.cpp:11-136) */
int  orc_cloud_save(const orc_cloud* c, const char* filename, const float T[16], int step, int binary);
int  orc_cloud_load(orc_cloud* c, const char* filename, float T_out[16]);

/* number of OpenMP threads the parallel (results-identical) loops will use */
int  orc_num_threads(void);
void orc_set_num_threads(int n);
/* 1: CorrespondenceFinder::compute and Linearizer::update are spread over the OpenMP threads the way the reference does it
 * (contiguous row blocks / correspondence chunks, per-thread partial sums reduced in thread order: correspondencefinder.cpp:38-51,
 * 110-114, linearizer.cpp:32-39,93-108) but WITHOUT its remainder dropping; 0 (default): the canonical one-thread loops.  Only the
 * timed CPU baseline switches it on. */
void orc_set_parallel_align(int enabled);
/* eigensolver trig: 0 (default, canonical) = fixed double-precision algorithms (+ - * / only, the same operations as the kernels) rounded
 * once to float; 1 = literal float libm calls as the reference makes them (last bit depends on the libm version); 2 = double libm rounded
 * to float */
void orc_set_trig_mode(int mode);
void orc_trig_eval(int mode, int n, const float* y, const float* x, float* theta, float* c, float* s);

#ifdef __cplusplus
}
#endif
#endif
