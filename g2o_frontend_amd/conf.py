"""Parameter sets of the reference's configuration files as `pwn_simple_aligner.cpp:214-269` applies them:
`pwn_core/conf/pwn_aligner_1_1.conf` (VGA, imageScale 1) and `pwn_core/conf/pwn_aligner_1_4.conf` (imageScale 4).  The keys are the
keyword arguments the host mirror's setters take (`bench.build_objects`, `tests/test_gpu_parity.gpu_objects`);
`tests/golden/reference_conf.json` holds the values of the two files and `tests/test_golden.py` pins these tables to it."""

# pwn_core/conf/pwn_aligner_1_1.conf
VGA_CONF_CONVERTER = dict(min_distance=0.5, max_distance=4.5, world_radius=0.1, min_image_radius=10,
                          max_image_radius=30, min_points=50, stats_curvature_threshold=0.2,
                          point_info_curvature_threshold=0.02, normal_info_curvature_threshold=0.02)
VGA_CONF_ALIGNER = dict(min_distance=0.5, max_distance=4.5, inlier_distance_threshold=1.0,
                        inlier_normal_angular_threshold=0.95, flat_curvature_threshold=0.02,
                        inlier_curvature_ratio_threshold=1.3, inlier_max_chi2=9000.0, robust_kernel=1,
                        outer_iterations=10, inner_iterations=1)
# pwn_core/conf/pwn_aligner_1_4.conf (imageScale 4)
QVGA4_CONF_CONVERTER = dict(VGA_CONF_CONVERTER, min_image_radius=3, max_image_radius=6, min_points=10)
QVGA4_CONF_ALIGNER = dict(VGA_CONF_ALIGNER, inlier_distance_threshold=0.5)
# BASELINE configs[4]: 1280x960 frames, stats radii x2, everything else as the VGA file
K2_CONF_CONVERTER = dict(VGA_CONF_CONVERTER, min_image_radius=20, max_image_radius=60, min_points=200)
