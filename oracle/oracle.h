/*
 * oracle.h -- CPU restatement ("oracle") of the per-frame front-end of
 * whwh747/A-Low-Texture-Robust-Hybrid-Feature-Based-Visual-Odometry.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped path is the HIP library in
 * a-low-texture-robust-hybrid-feature-based-visual-odometry_amd/csrc (libhvo.so).
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
 *  and cannot be compiled here (needs OpenCV 3.2 + contrib, Eigen,
 * PCL).  Every routine cites the reference file:line it follows; arithmetic that lives in
 * un-vendored OpenCV 3.2.0 / Eigen is restated from their published algorithms and marked
 * "ASSUMED".  The pins we do have are the known-answer constants derivable from the
 * reference text (tests/test_oracle_known_answers.py).
 *
 * Plain C11, no dependencies.  Build: make -C oracle  (gcc -O3 -march=native -ffp-contract=off)
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* == cv::KeyPoint, 28 bytes */
typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } orc_keypoint;

/* == cv::line_descriptor::KeyLine, 68 bytes
 * (Thirdparty/line_descriptor/include/line_descriptor/descriptor_custom.hpp:105-144) */
typedef struct {
    float angle; int32_t class_id, octave; float pt_x, pt_y, response, size;
    float sx, sy, ex, ey, sox, soy, eox, eoy, length; int32_t num_pixels;
} orc_keyline;

typedef struct { double normal[3], center[3], mse; int32_t n_points, rid; } orc_plane;

/* ---------------- alternative readings of un-vendored dependencies ----------------
 * Two places where the author's OpenCV build may have behaved differently from the oracle's default reading (SURVEY.md
 * Appendix A marks them "(?)").  Each is a switch, so that a box with OpenCV 3.2 settles it by flipping a flag; the
 * defaults (all 0) are what the golden vectors and the HIP path implement.
 *   ORC_READING_BLUR_FLOAT  cv::GaussianBlur on CV_8U served by IPP (ippiFilterGaussianBorder): float kernel, float
 *                           accumulation, one rounding at the end -- differs from the fixed-point path by at most 1.
 *                           Affects the 7x7 blur of ORB (ORBextractor.cc:1084) and the 5x5 blur of LBD.
 *   ORC_READING_LSD_8U      cv::LineSegmentDetector working on CV_8U (blur and 0.8x resize in the u8 fixed-point paths,
 *                           as OpenCV >= 3.2-ish asserts 8UC1) instead of on the image converted to CV_64F.            */
enum { ORC_READING_BLUR_FLOAT = 0, ORC_READING_LSD_8U = 1, ORC_READING_COUNT = 2 };
void  orc_set_reading(int which, int on);
int   orc_get_reading(int which);

/* ---------------- assumed OpenCV 3.2.0 primitives (cvsem.c) ---------------- */
int   orc_cvround_f(float v);
int   orc_cvround_d(double v);
float orc_fast_atan2(float y, float x);
/* cv::resize(u8, INTER_LINEAR) */
void  orc_resize_linear_u8_factor(const uint8_t *src, int sw, int sh, int sstride,
                                  uint8_t *dst, int dw, int dh, int dstride, double fx, double fy);
void  orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride,
                           uint8_t *dst, int dw, int dh, int dstride);
/* fixed-point (x256) 1-D Gaussian kernel as cv::GaussianBlur builds it for CV_8U; returns sum */
int   orc_gaussian_kernel_q8(int ksize, double sigma, int *k);
/* cv::GaussianBlur(u8, ksize x ksize, sigma, BORDER_REFLECT_101) */
void  orc_gaussian_blur_u8(const uint8_t *src, int w, int h, int sstride,
                           uint8_t *dst, int dstride, int ksize, double sigma);
/* cv::Sobel(u8 -> CV_16S, dx,dy, ksize 3, BORDER_REFLECT_101) */
void  orc_sobel3_u8_s16(const uint8_t *src, int w, int h, int sstride,
                        int16_t *dst, int dstride_elems, int dx, int dy);
/* cv::FAST(view, kps, threshold, nonmax=true), TYPE_9_16. Writes (x,y,score) int triples in
 * raster order; returns count (never more than cap). */
int   orc_fast9_16(const uint8_t *img, int stride, int vw, int vh, int threshold,
                   int *xys, int cap);
/* FAST corner score of a single pixel (cornerScore<16>), independent of the threshold that
 * admitted it: max over the 16 arcs of 9 of max(min(d), -max(d)) - 1. */
int   orc_fast_score(const uint8_t *p, int stride);
/* cv::LineIterator(img, p1, p2).count for float end points (8-connected) */
int   orc_line_iterator_count(int w, int h, float x1, float y1, float x2, float y2);

/* ---------------- ORB (orb.c): src/ORBextractor.cc ---------------- */
typedef struct {
    int   nfeatures;      /* ORBextractor.nFeatures  (TUM3.yaml:41) */
    float scale_factor;   /* ORBextractor.scaleFactor */
    int   nlevels;        /* ORBextractor.nLevels     */
    int   ini_th_fast;    /* ORBextractor.iniThFAST   */
    int   min_th_fast;    /* ORBextractor.minThFAST   */
} orc_orb_params;

typedef struct orc_orb orc_orb;
orc_orb *orc_orb_create(const orc_orb_params *p);
void     orc_orb_destroy(orc_orb *o);
/* ORBextractor::operator() (ORBextractor.cc:1041-1103). Returns 0, *n = #keypoints. */
int      orc_orb_extract(orc_orb *o, const uint8_t *gray, int w, int h, int stride,
                         orc_keypoint *kps, uint8_t *desc32, int cap, int *n);
/* table / intermediate accessors for known-answer and per-stage parity tests */
int         orc_orb_umax(const orc_orb *o, int *umax16);                 /* 16 ints */
int         orc_orb_features_per_level(const orc_orb *o, int *out);     /* nlevels ints */
float       orc_orb_scale(const orc_orb *o, int level);
const int8_t *orc_orb_pattern(void);                                    /* 256*4 */
int         orc_orb_level(const orc_orb *o, int level, int *w, int *h, int *stride,
                          const uint8_t **data);       /* mvImagePyramid[level] (no apron) */
int         orc_orb_level_blurred(const orc_orb *o, int level, const uint8_t **data, int *stride);
int         orc_orb_grid(const orc_orb *o, int level, int *ncols, int *nrows, int *wcell, int *hcell);
/* candidates fed to DistributeOctTree at `level`: n triples (x,y,response), cell-major order */
int         orc_orb_candidates(const orc_orb *o, int level, const int **xys);
int         orc_orb_level_count(const orc_orb *o, int level);            /* kps kept at level */

/* ---------------- PEAC planes (peac.c): src/PlaneExtractor.cpp, include/peac ---------------- */
/* readDepthImage + PlaneFitter::run.  labels: w*h int32 (-1 = none); planes sorted by N desc. */
int  orc_peac_run(const uint16_t *depth, int w, int h, int stride_bytes,
                  float fx, float fy, float cx, float cy, float depth_factor,
                  int32_t *labels, orc_plane *planes, int cap, int *nplanes);
void orc_eig33sym(const double K[3][3], double s[3], double V[3][3]);        /* cyclic Jacobi: cross-check only */
void orc_eig33_smallest(const double K[3][3], double *lambda0, double v[3]);  /* what Stats::compute uses */
double orc_peac_T_mse_init(double z);
double orc_peac_T_ang_init(double z);
double orc_peac_T_dz(double z);
double orc_peac_T_mse_merge(double z);
/* test hooks (tests/test_ref_pins.py): the fitter's union-find, the grid walk of one segment */
void *orc_ds_create(int n);
int   orc_ds_union(void *d, int x, int y);
int   orc_ds_find(void *d, int x);
int   orc_ds_set_size(void *d, int x);
void  orc_ds_free(void *d);
int   orc_grid_line_cells(double x1, double y1, double x2, double y2, int *cx, int *cy, int cap);

/* ---------------- lines (lsd.c, lbd.c) ---------------- */
/* cv::LineSegmentDetector (LSD_REFINE_STD defaults) on a CV_8UC1 image: segs = n x 4 floats */
int  orc_lsd_detect(const uint8_t *gray, int w, int h, int stride, float *segs, int cap, int *n_out);
/* LINEextractor::operator() (src/LineExtractor.cpp:329-380) */
int  orc_line_extract(const uint8_t *gray, int w, int h, int stride, int nfeatures,
                      orc_keyline *kls, uint8_t *desc32, double *linefn3, int cap, int *n_out);
/* BinaryDescriptor::compute for octave-0 keylines; desc72 (n x 72 floats) optional */
void orc_lbd_compute(const uint8_t *gray, int w, int h, int stride, const orc_keyline *kl, int n,
                     uint8_t *desc32, float *desc72);
void orc_lbd_weights(double *coefL21, double *coefG63);
const int *orc_lbd_combinations(void);

/* ---------------- Hamming (match.c): src/ORBmatcher.cc:1676, LSDmatcher.cpp:803-863,1137 ---- */
int  orc_descriptor_distance(const uint8_t *a, const uint8_t *b);
/* cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, 2): idx2/dist2 are nq*2, -1/INT_MAX padded */
void orc_hamming_knn2(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx2, int32_t *dist2);
void orc_hamming_matrix(const uint8_t *q, int nq, const uint8_t *t, int nt, uint16_t *d);
/* LSDmatcher::matchNNR */
int  orc_match_nnr(const uint8_t *d1, int n1, const uint8_t *d2, int n2, float nnr, int32_t *m12);

/* LSDmatcher::FrameBFMatch (src/LSDmatcher.cpp:942-966, lineDescriptorMAD 1110-1135) and SearchDouble's mutual check (902-939) */
int  orc_frame_bf_match(const uint8_t *d1, int n1, const uint8_t *d2, int n2, float TH, float nnratio, int32_t *m12);
int  orc_search_double(const uint8_t *d1, int n1, const uint8_t *d2, int n2, float TH, float nnratio, int32_t *m12);

/* ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, mono) core (src/ORBmatcher.cc:1353-1497) */
int  orc_search_by_projection(const uint8_t *q_desc, int nq, const float *q_u, const float *q_v, const float *q_radius,
                              const int32_t *q_min_level, const int32_t *q_max_level, const float *q_ur, const float *q_angle,
                              const uint8_t *q_blocks,
                              const void *t_kp, const float *t_uright, const uint8_t *t_occupied, const uint8_t *t_desc, int nt,
                              float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, int th_high, int check_orientation,
                              int32_t *match_idx, int32_t *match_dist);
/* ORBmatcher::SearchByProjection(F, vpMapPoints, th) core (src/ORBmatcher.cc:45-132): best / second best + ratio */
int  orc_search_by_projection_map(const uint8_t *q_desc, int nq, const float *q_u, const float *q_v, const float *q_radius,
                                  const int32_t *q_min_level, const int32_t *q_max_level, const float *q_ur, const uint8_t *q_blocks,
                                  const void *t_kp, const float *t_uright, const uint8_t *t_occupied, const uint8_t *t_desc, int nt,
                                  float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, int th_high, float nn_ratio,
                                  int32_t *match_idx, int32_t *match_dist);
/* the projection prologues of the two guided searches (src/ORBmatcher.cc:1364-1405; 55-70 + 134-140) */
void orc_project_last(const float *Tcw, const float *Tlw, int n, const float *x3Dw, const int32_t *octave,
                      float fx, float fy, float cx, float cy, float mbf, float mb, int mono, float th, const float *scale_factors,
                      float mnMinX, float mnMinY, float mnMaxX, float mnMaxY,
                      float *q_u, float *q_v, float *q_radius, int32_t *q_min_level, int32_t *q_max_level, float *q_ur, int32_t *fwd_bwd);
void orc_track_windows(int n, const int32_t *level, const float *view_cos, float th, const float *scale_factors,
                       float *q_radius, int32_t *q_min_level, int32_t *q_max_level);
/* Frame::ComputeStereoFromRGBD (src/Frame.cc:1940-1961) */
void orc_stereo_from_rgbd(const void *kp, const void *kp_un, int n, const uint16_t *depth, int w, int h, int stride_bytes,
                          float depth_factor, float bf, float *uright, float *zdepth);


/* ---------------- Frame post-processing (frame.c) ---------------- */
/* cv::undistortPoints(src, dst, K, dist, Mat(), K), OpenCV 3.2.0 (ASSUMED); dist5 = k1 k2 p1 p2 k3 */
void orc_undistort_points(const float *xy_in, int n, float fx, float fy, float cx, float cy, const float *dist5, float *xy_out);
/* Frame::UndistortKeyPoints (src/Frame.cc:1701-1731) */
void orc_undistort_keypoints(const orc_keypoint *kp, int n, float fx, float fy, float cx, float cy, const float *dist5, orc_keypoint *kp_un);
/* Frame::ComputeImageBounds (src/Frame.cc:1733-1762): {mnMinX, mnMaxX, mnMinY, mnMaxY} */
void orc_image_bounds(int w, int h, float fx, float fy, float cx, float cy, const float *dist5, float *bounds4);
/* Frame::AssignFeaturesToGrid (src/Frame.cc:832-847, 1680-1690) as CSR, cell = col*48 + row */
int  orc_assign_features_to_grid(const orc_keypoint *kp_un, int n, const float *bounds4, int32_t *cell_start, int32_t *cell_items);
/* Frame::AssignFeaturesToGridForLine (src/Frame.cc:849-872, src/lineIterator.cpp:34-76) as CSR */
int  orc_assign_lines_to_grid(const orc_keyline *kl, int n, const float *bounds4, int32_t *cell_start, int32_t *cell_items, int cap);

/* the line tracker's own calls (match.c): LSDmatcher::SearchByGeomNApearance (src/LSDmatcher.cpp:36-108) and
 * LSDmatcher::SearchByProjection(Cur, Last, th) (561-662) over Frame::GetFeaturesInAreaForLine (src/Frame.cc:1557-1627) */
int  orc_lines_geom_match(const uint8_t *d_last, const orc_keyline *kl_last, const uint8_t *last_has_mapline, int n_last,
                          const uint8_t *d_cur, const orc_keyline *kl_cur, int n_cur, float desc_th, const float *bounds4,
                          int32_t *matches12, uint8_t *accepted);
int  orc_search_lines_by_projection(int nq, const float *q_xyxy, const orc_keyline *q_kl, const uint8_t *q_desc, const uint8_t *q_blocks,
                                    const orc_keyline *t_kl, const double *t_linefn, const uint8_t *t_desc, const uint8_t *t_occupied, int nt,
                                    const int32_t *cell_start, const int32_t *cell_items, const float *bounds4, float th,
                                    int32_t *match_idx, int32_t *match_dist);

/* Frame::cullingLine (src/Frame.cc:952-1116; helpers 1117-1202), SURVEY.md 8f.2 */
int  orc_line_iterator_count_clipped(int w, int h, float x1, float y1, float x2, float y2);
int  orc_cull_lines(const uint8_t *gray, int w, int h, int stride, const orc_keyline *kl, const double *fn, int n,
                    double dis, double angle_deg, double endpoint_dis, orc_keyline *kl_out, uint8_t *desc32, double *fn_out);

/* Frame::isLineGood (src/Frame.cc:1205-1322): the 3-D line of every key line (line3d.c).  Layout == hvo_line3d. */
typedef struct {
    double A[3], B[3];          /* mvLines3D[i] (camera frame); zeros when no line was fitted */
    double line_nor[3];         /* mvLineNor[i] = A x B; (-1,-1,-1) when none */
    float  line_eq[3];          /* mvLineEq[i] = (B - A) / |B - A| in float; (-1,-1,-1) when none */
    int32_t good;               /* 1: |A - B| > 0.02, the line was pushed to mVF3DLines */
    int32_t n_samples;          /* samples with valid depth (<= 21) */
    int32_t n_inliers;          /* rl.pts.size() */
    uint32_t inlier_mask;       /* bit j: valid sample j is an inlier */
    int32_t pad;
} orc_line3d;
int  orc_lines_3d(const orc_keyline *kl, int n, const uint16_t *depth, int w, int h, int stride_bytes,
                  float fx, float fy, float cx, float cy, float depth_factor, uint32_t seed, orc_line3d *out);

/* The tail of Frame::ComputePlanes (src/Frame.cc:2110-2212, 2214-2274; planes_tail.c).  Layouts == hvo_plane_cloud / hvo_surface_normal. */
typedef struct {
    float coef[4];              /* mvPlaneCoefficients entry: the refit (sign rule applied) when valid, else (n, -n.c) of the extracted plane */
    int32_t valid;              /* 1: the plane passed the distance gate and the refit -> pushed to mvPlanePoints / mvPlaneCoefficients */
    int32_t gate_ok;            /* 1: no voxel point farther than Plane.DistanceThreshold */
    int32_t first, n_points;    /* its voxel-grid cloud = cloud_xyz[first .. first + n_points) */
    int32_t n_pixels;           /* plane_vertices_[i].size() */
    int32_t n_inliers;          /* inliers of the refined model */
} orc_plane_cloud;
typedef struct { float normal[3]; float position[3]; int32_t frame_x, frame_y; } orc_surface_normal;
int  orc_sac_plane(const float *xyz, int n, double threshold, float coef[4]);
int  orc_plane_clouds(const uint16_t *depth, int w, int h, int stride_bytes, float fx, float fy, float cx, float cy, float depth_factor,
                      const int32_t *labels, const orc_plane *planes, int nplanes, double dist_th,
                      float *cloud_xyz, int cap, orc_plane_cloud *out);
int  orc_surface_normals(const uint16_t *depth, int w, int h, int stride_bytes, float fx, float fy, float cx, float cy, float depth_factor,
                         orc_surface_normal *out, int cap);

/* ---------------- vanishing-point clustering of the key lines (vps.c; reference src/Frame.cc:442-778) ---------------- */
int  orc_normals_lpvo(const uint16_t *depth, int w, int h, int stride_bytes, float fx, float fy, float cx, float cy, float depth_factor,
                      double *normals, float *depth_out, int *pixel, int cap);    /* Manhattan::computeNormalsLPVO, the CV_32F reading */
int  orc_vp_iterations(void);                                                                     /* 105 */
void orc_vp_line_params(const orc_keyline *kl, int n, double *para, double *length, double *ori);
void orc_vp_sphere_grid(const double *para, const double *length, const double *ori, int n, double fx, double cx, double cy,
                        double *grid /* 90 x 360 */, double *raw /* before smoothing, or NULL */);
void orc_vp_line2vps(const orc_keyline *kl, int n, float fx, float fy, float cx, float cy, const double *vps, double th_angle, int32_t *vp_idx);
void orc_vp_hypothesis(const orc_keyline *kl, int n, float fx, float cx, float cy, uint32_t seed, int index, double *hyp9);
int  orc_vanishing_points(const orc_keyline *kl, int n, float fx, float fy, float cx, float cy, uint32_t seed, double th_angle,
                          double *vps /* 3 x 3 */, int *best_idx, double *best_score, int32_t *vp_idx /* n */,
                          double *scores /* iterations * 360 or NULL */, double *grid /* 90 x 360 or NULL */);

#ifdef __cplusplus
}
#endif
#endif
