/*
 * hvo.h -- C ABI of libhvo.so: the MI355X (gfx950) per-frame RGB-D front-end
 * (ORB + LSD/LBD lines + PEAC planes + Hamming matching) that replaces the bodies of
 *
 *   Frame::ExtractORB / ExtractORBNDepth   reference src/Frame.cc:886 / :874  (include/Frame.h:92-93)
 *       -> ORBextractor::operator()        reference src/ORBextractor.cc:1041 (include/ORBextractor.h:59-61)
 *   Frame::ExtractLSD                      reference src/Frame.cc:895          (include/Frame.h:96)
 *       -> LINEextractor::operator()       reference src/LineExtractor.cpp:329 (include/LineExtractor.h:193)
 *   Frame::ComputePlanes                   reference src/Frame.cc:2104         (include/Frame.h:415)
 *       -> PlaneDetection::readDepthImage / runPlaneDetection
 *                                          reference src/PlaneExtractor.cpp:26,60 (include/PlaneExtractor.h:50-54)
 *   ORBmatcher::DescriptorDistance         reference src/ORBmatcher.cc:1676    (include/ORBmatcher.h:44)
 *   LSDmatcher::match / matchNNR           reference src/LSDmatcher.cpp:828 / :803 (include/LSDmatcher.h:43)
 *   LSDmatcher::FrameBFMatch / SearchDouble reference src/LSDmatcher.cpp:942 / :902
 *   ORBmatcher::SearchByProjection         reference src/ORBmatcher.cc:1353 (frame to frame) and :45 (local map)
 * and, of the Frame constructor's post-processing:
 *   Frame::cullingLine                     reference src/Frame.cc:952
 *   Frame::UndistortKeyPoints / ComputeImageBounds / ComputeStereoFromRGBD / AssignFeaturesToGrid(ForLine)
 *                                          reference src/Frame.cc:1701 / 1733 / 1940 / 832 / 849
 *
 * The reference has no FFI of its own (single C++ process); INTEGRATION.md shows the
 * adaptor a maintainer adds to Frame.cc to call these entry points.
 *
 * Conventions
 *   - plain C types only; the caller owns every in/out buffer; nothing throws across the ABI
 *   - return value 0 = HVO_OK, negative = hvo_status; hvo_strerror() names it
 *   - empty image (NULL or w/h <= 0) -> *n = 0 and HVO_OK, like ORBextractor.cc:1044
 *   - a ctx is NOT thread-safe (neither is ORBextractor: mvImagePyramid is state); use one ctx
 *     per thread / per GPU.  Different ctx's may run concurrently (Frame.cc:210-215 pattern).
 *   - there is NO CPU fallback: every entry point fails with HVO_ERR_NO_DEVICE / HVO_ERR_HIP when
 *     the GPU or the gfx950 code object is unavailable.
 */
#ifndef HVO_H
#define HVO_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HVO_ABI_VERSION 3      /* 2: streamed-sequence entry points (hvo_stream_*), HVO_ERR_BUSY; 3: the Frame tail as pipeline stages
                                  (HVO_STAGE_LINES3D / _VP / _PLANE_TAIL / _GRIDS, hvo_frame_tail), hvo_stream_params grew three fields */

typedef enum {
    HVO_OK = 0,
    HVO_ERR_INVALID_ARG = -1,   /* NULL ctx/pointer, bad sizes */
    HVO_ERR_NO_DEVICE = -2,     /* no HIP device / wrong architecture */
    HVO_ERR_HIP = -3,           /* a HIP runtime call failed (hvo_last_error has the text) */
    HVO_ERR_UNSUPPORTED = -4,   /* image geometry outside what the kernels were sized for */
    HVO_ERR_CAPACITY = -5,      /* an internal fixed-capacity slab overflowed; results truncated */
    HVO_ERR_BAD_DTYPE = -6,     /* mirrors the CV_8UC1 assert (ORBextractor.cc:1048) and the
                                   CV_16U check (PlaneExtractor.cpp:34-38) */
    HVO_ERR_BUSY = -7           /* hvo_stream_submit: the ring slot still holds a frame that was not collected */
} hvo_status;

/* == cv::KeyPoint (28 bytes): what ORBextractor::operator() fills (ORBextractor.cc:1041-1103) */
typedef struct {
    float x, y;        /* pt, already multiplied by mvScaleFactor[octave] (ORBextractor.cc:1093-1099) */
    float size;        /* (int)(31 * scale) */
    float angle;       /* degrees [0,360), IC_Angle + fastAtan2 */
    float response;    /* FAST score */
    int32_t octave;
    int32_t class_id;  /* -1 */
} hvo_keypoint;

/* == cv::line_descriptor::KeyLine (68 bytes),
 * Thirdparty/line_descriptor/include/line_descriptor/descriptor_custom.hpp:105-144 */
typedef struct {
    float angle; int32_t class_id, octave;
    float pt_x, pt_y, response, size;
    float sx, sy, ex, ey;          /* startPointX/Y, endPointX/Y */
    float sox, soy, eox, eoy;      /* s/ePointInOctaveX/Y */
    float length; int32_t num_pixels;
} hvo_keyline;

/* one ahc::PlaneSeg of PlaneFitter::extractedPlanes (include/peac/AHCPlaneSeg.hpp:129-135) */
typedef struct {
    double normal[3], center[3], mse;
    int32_t n_points;   /* PlaneSeg::N */
    int32_t rid;        /* root block id */
} hvo_plane;

typedef struct {
    /* ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST): Tracking.cc:118-124,
     * Examples/RGB-D/TUM3.yaml:41-54 */
    int32_t orb_nfeatures;
    float   orb_scale_factor;
    int32_t orb_nlevels;
    int32_t orb_ini_th_fast;
    int32_t orb_min_th_fast;
    /* LINEextractor(numOctaves, scale, nLSDFeature, min_line_length): Tracking.cc:126-132,
     * TUM3.yaml:60-63.  NOTE the reference passes float scale into an int parameter (-> 1) and only
     * octave 0 runs for nLevels=1; num_octaves != 1 is HVO_ERR_UNSUPPORTED. */
    int32_t lsd_num_octaves;
    float   lsd_scale;
    int32_t lsd_nfeatures;
    /* camera / depth: TUM3.yaml:8-11,34; depth_map_factor = 1/DepthMapFactor as a float
     * (Tracking.cc:156-160) */
    float fx, fy, cx, cy;
    float depth_map_factor;
    /* execution */
    int32_t device;        /* HIP device ordinal */
    int32_t max_batch;     /* frames resident per batch call (>=1) */
} hvo_params;

typedef struct hvo_ctx hvo_ctx;

void        hvo_default_params(hvo_params *p);        /* TUM3.yaml values, device 0, max_batch 1 */
int         hvo_create(const hvo_params *p, hvo_ctx **out);
void        hvo_destroy(hvo_ctx *ctx);
const char *hvo_strerror(int status);
const char *hvo_last_error(const hvo_ctx *ctx);       /* text of the last HIP failure */
int         hvo_abi_version(void);

/* ---- single-frame entry points, host buffers (the drop-in boundary) ---- */

/* ORBextractor::operator()(image, mask(ignored), keypoints, descriptors).
 * gray: CV_8UC1 w x h, `stride` bytes per row.  kp/desc32: capacity `cap` entries (desc is cap x 32). */
int hvo_extract_orb(hvo_ctx *ctx, const uint8_t *gray, int w, int h, int stride,
                    hvo_keypoint *kp, uint8_t *desc32, int cap, int *n);

/* LINEextractor::operator()(image, mask(ignored), keylines, descriptors, lineVec2d).
 * linefn3: cap x 3 doubles (normalised 2-D line functions, LineExtractor.cpp:367-377). */
int hvo_extract_lsd(hvo_ctx *ctx, const uint8_t *gray, int w, int h, int stride,
                    hvo_keyline *kl, uint8_t *desc32, double *linefn3, int cap, int *n);

/* PlaneDetection::readDepthImage + runPlaneDetection on the raw 16-bit depth (Frame.cc:2104-2108).
 * depth: CV_16UC1, `stride` bytes per row.  labels: w*h int32 (PlaneFitter::membershipImg, -1 = none).
 * planes: extractedPlanes after refineDetails, sorted by N descending. */
int hvo_compute_planes(hvo_ctx *ctx, const uint16_t *depth, int w, int h, int stride,
                       int32_t *labels, hvo_plane *planes, int cap, int *n);

/* ---- Hamming matching (32-byte descriptors, row-major n x 32) ---- */
/* ORBmatcher::DescriptorDistance for every (q,t) pair */
int hvo_hamming_matrix(hvo_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, uint16_t *d);
/* cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, 2) (LSDmatcher.cpp:811-812): two best train indices
 * per query, ascending distance, ties -> lower train index; -1 / INT32_MAX where nt < 2 */
int hvo_hamming_knn2(hvo_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt,
                     int32_t *idx2, int32_t *dist2);
/* LSDmatcher::matchNNR(desc1, desc2, nnr, matches_12): returns #matches in *n_matches */
int hvo_match_nnr(hvo_ctx *ctx, const uint8_t *d1, int n1, const uint8_t *d2, int n2, float nnr,
                  int32_t *matches12, int *n_matches);

/* LSDmatcher::FrameBFMatch(ldesc1, ldesc2, LineMatches, TH) (reference src/LSDmatcher.cpp:942-966): knnMatch(k = 2),
 * lineDescriptorMAD's nn12 threshold (1110-1135), accepted if d1 - d0 > threshold && d0 < th && d0 < nnratio * d1.
 * hvo_search_double = the core of LSDmatcher::SearchDouble / SearchByDescriptor (902-939, 865-899): FrameBFMatch in both
 * directions (the reference uses two threads), i -> j kept only if j -> i. */
int hvo_frame_bf_match(hvo_ctx *ctx, const uint8_t *d1, int n1, const uint8_t *d2, int n2, float th, float nnratio,
                       int32_t *matches12, int *n_matches);
int hvo_search_double(hvo_ctx *ctx, const uint8_t *d1, int n1, const uint8_t *d2, int n2, float th, float nnratio,
                      int32_t *matches12, int *n_matches);

/* LSDmatcher::SearchByGeomNApearance (reference src/LSDmatcher.cpp:36-108) on host arrays: see hvo_stream_match_lines_geom.  bounds4 = {mnMinX,
 * mnMaxX, mnMinY, mnMaxY} of the current frame; matches12 / accepted: n_last entries. */
int hvo_match_lines_geom(hvo_ctx *ctx, const uint8_t *d_last, const hvo_keyline *kl_last, const uint8_t *last_has_mapline, int n_last,
                         const uint8_t *d_cur, const hvo_keyline *kl_cur, int n_cur, float desc_th, const float bounds4[4],
                         int32_t *matches12, uint8_t *accepted, int *n_accepted);
/* LSDmatcher::SearchByProjection(Cur, Last, th) core (reference src/LSDmatcher.cpp:561-662, Frame::GetFeaturesInAreaForLine src/Frame.cc:1557-1627)
 * on host arrays: q_kl[i] = LastFrame.mvKeylinesUn of query i; t_linefn (nt x 3) = mvKeyLineFunctions; cell_start / cell_items = the current frame's
 * line grid as hvo_assign_lines_to_grid returns it; at most 2048 current lines.  See hvo_stream_search_lines_by_projection. */
int hvo_search_lines_by_projection(hvo_ctx *ctx, int nq, const float *q_xyxy, const hvo_keyline *q_kl, const uint8_t *q_desc, const uint8_t *q_blocks,
                                   const hvo_keyline *t_kl, const double *t_linefn, const uint8_t *t_desc, const uint8_t *t_occupied, int nt,
                                   const int32_t *cell_start, const int32_t *cell_items, const float bounds4[4], float th,
                                   int32_t *match_idx, int32_t *match_dist, int *n_matches);

/* ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, mono) core (reference src/ORBmatcher.cc:1353-1497,
 * Frame::GetFeaturesInArea src/Frame.cc:1502-1555).  One query per last-frame map point that passed the
 * projection tests (:1381-1404): projected (u,v), radius = th * scale[octave], octave band [min,max] with
 * GetFeaturesInArea's conventions (min <= 0 and max < 0: no level check), ur = u - bf*invz (q_ur may be NULL),
 * key-point angle (rotation histogram) and q_blocks[i] != 0 when the map point has observations (the feature
 * it claims is then skipped by later queries, :1425-1427).  t_* describe the current frame: undistorted key
 * points, mvuRight (may be NULL), features already holding an observed map point (may be NULL), descriptors;
 * mnMin/Max are the frame's image bounds (64 x 48 grid).  match_idx[i] = current-frame index or -1. */
int hvo_search_by_projection(hvo_ctx *ctx, const uint8_t *q_desc, int nq, const float *q_u, const float *q_v, const float *q_radius,
                             const int32_t *q_min_level, const int32_t *q_max_level, const float *q_ur, const float *q_angle,
                             const uint8_t *q_blocks, const hvo_keypoint *t_kp, const float *t_uright, const uint8_t *t_occupied,
                             const uint8_t *t_desc, int nt, float mnMinX, float mnMinY, float mnMaxX, float mnMaxY,
                             int th_high, int check_orientation, int32_t *match_idx, int32_t *match_dist, int *n_matches);

/* Frame::ExtractLSD up to and including cullingLine (reference src/Frame.cc:895-934, 952-1116):
 * the LINEextractor output, then near-collinear segments merged (PointLineDistance / TwoLineAngle /
 * MergeTwoLines, 1117-1202), KeyLines rebuilt and re-sorted by response (class_id = rank), second LBD pass,
 * line functions.  isLineGood (the 3-D line fit with rand()) is not part of it.  Same conventions as
 * hvo_extract_lsd.  hvo_set_line_culling changes cullingLine's dis / angle (degrees) / endpoint_dis
 * (defaults 5, 2.5, 15: Frame.cc:934). */
int hvo_extract_lsd_culled(hvo_ctx *ctx, const uint8_t *gray, int w, int h, int stride,
                           hvo_keyline *kl, uint8_t *desc32, double *linefn3, int cap, int *n);
int hvo_set_line_culling(hvo_ctx *ctx, double dis, double angle_deg, double endpoint_dis);

/* Frame::isLineGood (reference src/Frame.cc:1205-1322): the 3-D line of every key line from the depth image --
 * <= 21 samples along the segment with nearest-pixel depth, LINEextractor::compPt3dCov (src/LineExtractor.cpp:44-97) and the RANSAC
 * on Mahalanobis point-line distances of LINEextractor::extract3dline_mahdist (220-327).  The reference draws from a time-seeded
 * rand(); here the caller passes a seed and every line draws from its own xorshift32 stream (seed, line index), so results are
 * reproducible and independent of the order of the lines.  depth / intrinsics as in hvo_stereo_from_rgbd. */
typedef struct {
    double A[3], B[3];          /* mvLines3D[i] (camera frame); zeros when no line was fitted */
    double line_nor[3];         /* mvLineNor[i] = A x B; (-1,-1,-1) when none */
    float  line_eq[3];          /* mvLineEq[i] = (B - A) / |B - A| (float); (-1,-1,-1) when none */
    int32_t good;               /* 1: |A - B| > 0.02: the line enters mVF3DLines */
    int32_t n_samples;          /* samples with a valid depth (<= 21) */
    int32_t n_inliers;          /* RandomLine3d::pts.size() */
    uint32_t inlier_mask;       /* bit j: valid sample j is an inlier */
    int32_t pad;
} hvo_line3d;
int hvo_lines_3d(hvo_ctx *ctx, const hvo_keyline *kl, int n, const uint16_t *depth, int w, int h, int stride, uint32_t seed, hvo_line3d *out);

/* The vanishing-point clustering of the key lines that the Frame constructor runs on every frame (reference src/Frame.cc:330-337,
 * SURVEY.md 8f.4): Frame::getVPHypVia2Lines (442-545: 105 random pairs of lines x 360 rotations = 37 800 hypotheses of three
 * orthogonal vanishing directions), getSphereGrids (546-650: the 90 x 360 grid of pairwise line intersections, weighted, 3x3
 * smoothed), getBestVpsHyp (651-707: the first hypothesis with the largest sum of its three cells) and line2Vps (708-778: the
 * cluster of every line, thAngle = 1 degree in Frame.h:365).  kl are the (undistorted) key lines (mvKeylinesUn), intrinsics
 * the context's.  The reference draws from a time-seeded rand(); the caller passes a seed, group i of 360 hypotheses draws its
 * pair of lines from its own xorshift32 stream (seed, i).  vp_idx[i] = 0..2 (isStructLine[i] = true) or 3 (none);
 * grid (optional) receives the 90 x 360 smoothed sphere grid.  n < 2: nothing is computed (as the reference), all vp_idx = 3. */
typedef struct {
    double vps[3][3];           /* tmp_vps: the best hypothesis, three unit vectors (camera frame) */
    double score;               /* its summed grid length (0 when no hypothesis scored) */
    int32_t best;               /* its index, group * 360 + rotation */
    int32_t n_hypotheses;       /* 37 800 */
} hvo_vp_result;
int hvo_vanishing_points(hvo_ctx *ctx, const hvo_keyline *kl, int n, uint32_t seed, double th_angle,
                         hvo_vp_result *res, int32_t *vp_idx, double *grid);

/* The tail of Frame::ComputePlanes after the plane detector (reference src/Frame.cc:2110-2212) and Frame::MaxPointDistanceFromPlane
 * (2214-2274), SURVEY.md 8f.3.  labels / planes are the outputs of hvo_compute_planes for the same depth image.
 * hvo_plane_clouds: per plane the points of its pixels (float), pcl::VoxelGrid(0.1 m), the gate |n.p + d| <= dist_th
 * (Plane.DistanceThreshold of the settings file) and the pcl::SACSegmentation refit with the sign rule; valid planes are the
 * entries of mvPlanePoints / mvPlaneCoefficients, in order.  cloud_xyz (cap x 3 floats) receives every plane's voxel cloud,
 * plane i at [first, first + n_points).  PCL's semantics are restated (it is not vendored by the reference): oracle/planes_tail.c.
 * hvo_surface_normals: the 1/3-resolution cloud and pcl::IntegralImageNormalEstimation(AVERAGE_3D_GRADIENT, 0.05, 10) at the odd
 * grid positions = vSurfaceNormal (normal NaN where PCL leaves it undefined); needs (h/3/2) * (w/3/2) entries (80 x 107 for 640x480). */
typedef struct {
    float coef[4];              /* mvPlaneCoefficients entry when valid (refit, sign rule applied), else (n, -n.c) of the extracted plane */
    int32_t valid;              /* 1: passed the gate and the refit: the plane enters mvPlanePoints / mvPlaneCoefficients */
    int32_t gate_ok;            /* 1: no voxel point farther than dist_th from the extracted plane */
    int32_t first, n_points;    /* its voxel-grid cloud in cloud_xyz */
    int32_t n_pixels;           /* plane_vertices_[i].size() */
    int32_t n_inliers;          /* inliers of the refined model */
} hvo_plane_cloud;
typedef struct { float normal[3]; float position[3]; int32_t frame_x, frame_y; } hvo_surface_normal;   /* SurfaceNormal: normal, cameraPosition, FramePosition */
int hvo_plane_clouds(hvo_ctx *ctx, const uint16_t *depth, int w, int h, int stride, const int32_t *labels, const hvo_plane *planes, int n_planes,
                     double dist_th, float *cloud_xyz, int cap, hvo_plane_cloud *out, int *n_total);
int hvo_surface_normals(hvo_ctx *ctx, const uint16_t *depth, int w, int h, int stride, hvo_surface_normal *out, int cap, int *n);

/* Manhattan::computeNormalsLPVO (reference src/Manhattan.cpp:237-393, the second half of SURVEY.md 8f.4; run by the RGB-D Frame constructor
 * through Frame::ExtractMainImgPtNormals, src/Frame.cc:222, until the coarse Manhattan frame is initialised): surface normals from 10 x 10
 * box averages of the central-difference tangents at every 15th pixel.  The INTENDED reading is implemented -- depth in metres as CV_32F,
 * integral images with their zero row / column removed -- not what the reference binary computes: as compiled it reads the raw CV_16U image
 * through at<float> and moves half rows of its CV_64F integral images (undefined behaviour; csrc/lpvo.hip).
 * normals3: cap x 3 doubles (unit, or 0 when the cross product vanishes); depth_out: the sample's z; pixel2: (u, v).  *n = samples found. */
int hvo_normals_lpvo(hvo_ctx *ctx, const uint16_t *depth, int w, int h, int stride, double *normals3, float *depth_out, int32_t *pixel2, int cap, int *n);

/* ---- Frame post-processing of the outputs above ----------------------------------------
 * dist5 = {k1, k2, p1, p2, k3} (Camera.k1.. of the settings file; k3 = 0 when absent); the intrinsics are the
 * context's (hvo_params fx, fy, cx, cy).  The 64 x 48 grids (FRAME_GRID_COLS x FRAME_GRID_ROWS) are returned as
 * CSR: cell = col * 48 + row (the reference's mGrid[col][row]), cell_start has 64*48+1 entries, cell_items holds
 * feature indices in the reference's push order. */
#define HVO_GRID_COLS 64
#define HVO_GRID_ROWS 48
/* Frame::UndistortKeyPoints (reference src/Frame.cc:1701-1731): k1 == 0 copies the key points (1703-1707), else
 * cv::undistortPoints(pts, K, dist, Mat(), K) replaces x, y and keeps the other fields. */
int hvo_undistort_keypoints(hvo_ctx *ctx, const hvo_keypoint *kp, int n, const float dist5[5], hvo_keypoint *kp_un);
/* Frame::ComputeImageBounds (reference src/Frame.cc:1733-1762): bounds4 = {mnMinX, mnMaxX, mnMinY, mnMaxY} */
int hvo_image_bounds(hvo_ctx *ctx, int w, int h, const float dist5[5], float bounds4[4]);
/* Frame::AssignFeaturesToGrid (reference src/Frame.cc:832-847, PosInGrid 1680-1690); cell_items needs n entries */
int hvo_assign_features_to_grid(hvo_ctx *ctx, const hvo_keypoint *kp_un, int n, const float bounds4[4],
                                int32_t *cell_start, int32_t *cell_items, int *n_assigned);
/* Frame::AssignFeaturesToGridForLine (reference src/Frame.cc:849-872, src/lineIterator.cpp:34-76); a line enters
 * every cell its Bresenham walk visits; HVO_ERR_CAPACITY (with *n_items = the needed count) if cap is too small */
int hvo_assign_lines_to_grid(hvo_ctx *ctx, const hvo_keyline *kl, int n, const float bounds4[4],
                             int32_t *cell_start, int32_t *cell_items, int cap, int *n_items);

/* ORBmatcher::SearchByProjection(Frame &F, vpMapPoints, th) core (reference src/ORBmatcher.cc:45-132), the local-map
 * variant: one query per map point in view (projected u, v = mTrackProjX/Y; radius = RadiusByViewingCos * th *
 * scale[level]; levels [level-1, level]; ur = mTrackProjXR), best and second-best distance, accepted if best <= th_high
 * and not (both in the same octave && best > nn_ratio * second) (:117-124).  t_occupied / q_blocks as above. */
int hvo_search_by_projection_map(hvo_ctx *ctx, const uint8_t *q_desc, int nq, const float *q_u, const float *q_v, const float *q_radius,
                                 const int32_t *q_min_level, const int32_t *q_max_level, const float *q_ur, const uint8_t *q_blocks,
                                 const hvo_keypoint *t_kp, const float *t_uright, const uint8_t *t_occupied, const uint8_t *t_desc, int nt,
                                 float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, int th_high, float nn_ratio,
                                 int32_t *match_idx, int32_t *match_dist, int *n_matches);

/* The same search from the tracker's own per-point fields, its prologue (src/ORBmatcher.cc:55-70, RadiusByViewingCos 134-140) on the
 * device: one query per map point in view (mbTrackInView, not bad) with mTrackProjX / mTrackProjY / mTrackProjXR (may be NULL),
 * mnTrackScaleLevel and mTrackViewCos; radius = (viewCos > 0.998 ? 2.5 : 4.0) [* th when th != 1] * scale[level], levels
 * [level - 1, level].  Everything else as hvo_search_by_projection_map. */
int hvo_search_by_projection_tracked(hvo_ctx *ctx, const uint8_t *q_desc, int nq, const float *proj_x, const float *proj_y, const float *proj_xr,
                                     const int32_t *level, const float *view_cos, const uint8_t *q_blocks, float th,
                                     const hvo_keypoint *t_kp, const float *t_uright, const uint8_t *t_occupied, const uint8_t *t_desc, int nt,
                                     float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, int th_high, float nn_ratio,
                                     int32_t *match_idx, int32_t *match_dist, int *n_matches);

/* Frame::ComputeStereoFromRGBD (reference src/Frame.cc:1940-1961): uright[i] = kp_un[i].x - bf/d and zdepth[i] = d
 * where d = depth(v,u) * depth_map_factor at the truncated key-point position, if 0 < d < 7; else -1. */
int hvo_stereo_from_rgbd(hvo_ctx *ctx, const hvo_keypoint *kp, const hvo_keypoint *kp_un, int n,
                         const uint16_t *depth, int w, int h, int stride, float bf, float *uright, float *zdepth);

/* ---- batch entry points (config 4: independent frames; inputs stay resident in HBM) ---- */
#define HVO_STAGE_ORB    1u
#define HVO_STAGE_LSD    2u
#define HVO_STAGE_PLANES 4u
#define HVO_STAGE_ALL    7u
#define HVO_STAGE_LSD_CULL 8u    /* HVO_STAGE_LSD followed by Frame::cullingLine: the frame's kl / ldesc / linefn are the merged lines */
/* The rest of the Frame constructor (reference src/Frame.cc:205-233) as stages of the same pipelines: they read the key lines, depth image,
 * label image, planes and key points where the stages above left them in HBM (nothing is uploaded twice, nothing allocated per frame). */
#define HVO_STAGE_LINES3D    16u  /* Frame::isLineGood of every key line (src/Frame.cc:934-939, 1205-1322) = hvo_lines_3d; needs LSD + depth */
#define HVO_STAGE_VP         32u  /* vanishing points + line2Vps (src/Frame.cc:328-337, 442-778) = hvo_vanishing_points; needs LSD */
#define HVO_STAGE_PLANE_TAIL 64u  /* ComputePlanes' tail (src/Frame.cc:2110-2274) = hvo_plane_clouds + hvo_surface_normals; needs PLANES */
#define HVO_STAGE_GRIDS     128u  /* AssignFeaturesToGrid / ForLine (src/Frame.cc:832-872) = hvo_assign_*_to_grid; needs ORB + LSD */
#define HVO_STAGE_FRAME     (HVO_STAGE_ORB | HVO_STAGE_LSD_CULL | HVO_STAGE_LSD | HVO_STAGE_PLANES | HVO_STAGE_LINES3D | HVO_STAGE_VP | HVO_STAGE_PLANE_TAIL | HVO_STAGE_GRIDS)

typedef struct {
    const uint8_t  *gray;  int gray_stride;    /* bytes */
    const uint16_t *depth; int depth_stride;   /* bytes; may be NULL when planes are not requested */
} hvo_frame_in;

typedef struct {
    hvo_keypoint *kp; uint8_t *desc; int kp_cap; int n_kp;
    hvo_keyline *kl; uint8_t *ldesc; double *linefn; int kl_cap; int n_kl;
    int32_t *labels; hvo_plane *planes; int pl_cap; int n_planes;
    int status;                                  /* per-frame hvo_status */
    int8_t *labels8;                             /* optional: the label image as int8 (w*h bytes, -1 = none; plane ids < 64), i.e. as it
                                                    crosses PCIe, without the widening to CV_32S that `labels` gets (ABI version 2) */
} hvo_frame_out;

/* Results of the tail stages for one frame; any pointer may be NULL.  Capacities: lines3d / vp_idx kl_cap entries, plane_clouds 64,
 * cloud_xyz cloud_cap x 3 floats (hvo_tail_capacity), normals normals_cap, pt_cell_start / ln_cell_start 64*48+1, pt_cell_items kp_cap,
 * ln_cell_items kl_cap * 128.  The random draws of isLineGood and of the vanishing-point hypotheses use seed + frame index (batch) or
 * seed + ticket (stream): hvo_set_tail_params / hvo_stream_params.seed. */
typedef struct {
    hvo_line3d *lines3d;
    hvo_vp_result *vp; int32_t *vp_idx;
    hvo_plane_cloud *plane_clouds; float *cloud_xyz; int cloud_cap; int n_cloud;
    hvo_surface_normal *normals; int normals_cap; int n_normals;
    int32_t *pt_cell_start, *pt_cell_items; int pt_items_cap; int n_pt_items;
    int32_t *ln_cell_start, *ln_cell_items; int ln_items_cap; int n_ln_items;
    int status;
} hvo_frame_tail;
/* capacities of the tail results for a geometry: voxel-cloud points per frame, surface normals, line-grid items */
int hvo_tail_capacity(int kl_cap, int w, int h, int *cloud_cap, int *n_normals, int *ln_items_cap);
/* seed of the random draws, Plane.DistanceThreshold (default 0.05) and line2Vps' angle in radians (default 1 degree) for hvo_batch_run's tail stages */
int hvo_set_tail_params(hvo_ctx *ctx, uint32_t seed, double plane_dist_th, double vp_th_angle);
/* results of the tail stages of the first n frames of the resident batch (after hvo_batch_run with those stages) */
int hvo_batch_download_tail(hvo_ctx *ctx, int n, hvo_frame_tail *out);

/* host -> HBM copy of n (<= max_batch) frames of one geometry */
int hvo_batch_upload(hvo_ctx *ctx, int n, const hvo_frame_in *in, int w, int h);
/* enqueue every kernel of the selected stages for the resident batch and wait for completion */
int hvo_batch_run(hvo_ctx *ctx, unsigned stages);
/* HBM -> host copy of results (any pointer in hvo_frame_out may be NULL to skip it) */
int hvo_batch_download(hvo_ctx *ctx, int n, hvo_frame_out *out);
/* Result slabs in device memory (the only multi-GPU exchange is a gather of these).  One record of *slab_bytes per
 * frame: int32 {n_kp, n_kl, n_planes, status}, kp[kp_cap] (28 B), desc[kp_cap] (32 B), kl[kl_cap] (68 B), ldesc[kl_cap] (32 B),
 * linefn[kl_cap] (3 doubles), planes[pl_cap] (64 B); entries beyond the counts are zero.  hvo_batch_pack_results writes the first
 * n frames of the resident batch to d_slabs, a DEVICE pointer with room for n * slab_bytes (e.g. the tensor handed to
 * ncclAllGather); stages that did not run report zero counts. */
int hvo_batch_slab_layout(hvo_ctx *ctx, int *kp_cap, int *kl_cap, int *pl_cap, size_t *slab_bytes);
int hvo_batch_pack_results(hvo_ctx *ctx, int n, void *d_slabs);
/* The same with options: HVO_SLAB_LABELS appends the frame's label image (membershipImg as int8, -1 = no plane; w * h bytes rounded up
 * to 16) to every slab, so that the one collective of a multi-GPU caller also carries the plane labels -- 307 200 B per 640x480 frame on
 * top of the 93.7 KB of the records (a 256-frame gather over xGMI grows from 24 MB to 103 MB: ~0.7 ms per GPU at 153 GB/s per link). */
#define HVO_SLAB_LABELS 1u
int hvo_batch_slab_layout_ex(hvo_ctx *ctx, unsigned flags, int *kp_cap, int *kl_cap, int *pl_cap, size_t *labels_off, size_t *slab_bytes);
int hvo_batch_pack_results_ex(hvo_ctx *ctx, int n, void *d_slabs, unsigned flags);
/* Double-buffered batches: the end-to-end rate of consecutive batches at the resident batch's efficiency.  While the resident batch runs,
 * the NEXT batch's images go up into staging slabs (hvo_batch_stage_upload: enqueued on a copy stream of its own, returns at once) and the
 * LAST batch's results come down from one packed slab (hvo_batch_results_async: hvo_batch_pack_results_ex into a device slab, then ONE
 * contiguous copy into `host_slabs`, n * slab_bytes of page-locked memory; hvo_batch_results_wait waits for it).  hvo_batch_commit_staged
 * waits for the staged upload and makes it the resident batch (a device-to-device copy: ~6 ms per 8192 frames).  One host thread:
 *     stage_upload(0); commit
 *     loop k:  stage_upload(k + 1);  hvo_batch_run;  results_async(k);  commit            -- run(k) overlaps upload(k + 1) and download(k - 1) */
int hvo_batch_stage_upload(hvo_ctx *ctx, int n, const hvo_frame_in *in, int w, int h);
int hvo_batch_commit_staged(hvo_ctx *ctx);
int hvo_batch_results_async(hvo_ctx *ctx, int n, unsigned flags, void *host_slabs);
int hvo_batch_results_wait(hvo_ctx *ctx);
/* upload + run + download */
int hvo_extract_batch(hvo_ctx *ctx, int n, const hvo_frame_in *in, hvo_frame_out *out, int w, int h,
                      unsigned stages);

/* ---- streamed sequence (BASELINE config 5) -----------------------------------------------------------------
 * The reference constructs one Frame per camera image (src/Tracking.cc:262 -> Frame ctor src/Frame.cc:205-233: ExtractORB ||
 * ExtractLSD || ComputePlanes on three threads, then UndistortKeyPoints / ComputeStereoFromRGBD) and matches it against the
 * previous frame (TrackWithMotionModel: SearchByProjection(Cur, Last) src/Tracking.cc:2396, LSDmatcher::match(Last.mLdesc,
 * Cur.mLdesc) src/Tracking.cc:2299 -> src/LSDmatcher.cpp:42).  A hvo_stream keeps `depth` frames in flight on the GPU (one
 * frame's serial chains -- AHC, region growing -- leave most of the chip idle, the next frames run beside them) and the
 * results of the last `depth` frames resident in HBM, so the frame-to-frame matching reads descriptors, undistorted key
 * points and mvuRight where they were produced: only the tracker's per-query projections cross PCIe.
 *   submit  : copies the images into pinned staging, enqueues uploads + every kernel + result downloads, returns at once
 *   collect : waits for that frame and copies its results out (tickets may be collected in any order; a slot is reused
 *             by ticket + depth, which is refused with HVO_ERR_BUSY until the slot's frame was collected)
 * A frame stays matchable until `depth` newer frames have been submitted.  Not thread-safe. */
typedef struct hvo_stream hvo_stream;
typedef struct {
    int32_t  width, height;
    int32_t  depth;             /* frames in flight / resident (ring slots), 2..16 */
    uint32_t stages;            /* HVO_STAGE_* mask */
    float    dist5[5];          /* k1 k2 p1 p2 k3 for UndistortKeyPoints (k1 == 0: key points are copied, Frame.cc:1703-1707) */
    float    bf;                /* ComputeStereoFromRGBD's mbf; <= 0: mvuRight / mvDepth are not computed */
    uint32_t seed;              /* tail stages: frame `ticket` draws with seed + ticket (ABI 3) */
    float    plane_dist_th;     /* Plane.DistanceThreshold of the settings file; <= 0: 0.05 */
    float    vp_th_angle;       /* line2Vps' thAngle in radians; <= 0: 1 degree (Frame.h:365) */
} hvo_stream_params;
#define HVO_LINE_MATCH_NNR 0    /* LSDmatcher::match -> matchNNR (src/LSDmatcher.cpp:803-863): d0 < nnr * d1 */
#define HVO_LINE_MATCH_BF 1     /* LSDmatcher::FrameBFMatch (942-966) */
#define HVO_LINE_MATCH_DOUBLE 2 /* LSDmatcher::SearchDouble (902-939): both directions + mutual check */

int  hvo_stream_create(const hvo_params *p, const hvo_stream_params *sp, hvo_stream **out);
void hvo_stream_destroy(hvo_stream *s);
const char *hvo_stream_last_error(const hvo_stream *s);
/* capacities of the per-frame result arrays (key points / key lines / planes) */
int  hvo_stream_capacity(const hvo_stream *s, int *kp_cap, int *kl_cap, int *pl_cap);
/* Frame::ComputeImageBounds with the stream's distortion: {mnMinX, mnMaxX, mnMinY, mnMaxY} */
int  hvo_stream_image_bounds(const hvo_stream *s, float bounds4[4]);
/* depth may be NULL when planes are not requested (then mvuRight / mvDepth are not computed either) */
int  hvo_stream_submit(hvo_stream *s, const uint8_t *gray, int gray_stride, const uint16_t *depth, int depth_stride, int64_t *ticket);
int  hvo_stream_poll(hvo_stream *s, int64_t ticket);                 /* 1: complete, 0: still running */
/* out as in hvo_batch_download (any pointer may be NULL); kp_un / uright / zdepth: kp_cap entries, may be NULL */
int  hvo_stream_collect(hvo_stream *s, int64_t ticket, hvo_frame_out *out, hvo_keypoint *kp_un, float *uright, float *zdepth);
/* results of the frame's tail stages (HVO_STAGE_LINES3D / _VP / _PLANE_TAIL / _GRIDS of hvo_stream_params.stages); call BEFORE hvo_stream_collect
 * releases the slot, or instead of it with out == NULL there: hvo_stream_collect_tail waits for the frame like hvo_stream_collect does */
int  hvo_stream_collect_tail(hvo_stream *s, int64_t ticket, hvo_frame_tail *tail);
/* device time from the start of the frame's upload to the end of its ORB / line / plane kernels (ms) */
int  hvo_stream_stage_ms(hvo_stream *s, int64_t ticket, float ms3[3]);
/* SearchByProjection(Cur, Last) core between two resident frames.  Query i = last-frame feature q_index[i] (its descriptor and
 * key-point angle are read from the last frame's slot; pass q_desc (nq x 32) when pMP->GetDescriptor() differs from the frame's
 * own descriptor); q_u .. q_blocks and t_occupied (n_kp(cur) entries, may be NULL) as in hvo_search_by_projection. */
int  hvo_stream_search_by_projection(hvo_stream *s, int64_t cur, int64_t last, int nq, const int32_t *q_index, const uint8_t *q_desc,
                                     const float *q_u, const float *q_v, const float *q_radius, const int32_t *q_min_level, const int32_t *q_max_level,
                                     const float *q_ur, const uint8_t *q_blocks, const uint8_t *t_occupied, int th_high, int check_orientation,
                                     int32_t *match_idx, int32_t *match_dist, int *n_matches);
/* ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) WHOLE, between two resident frames (src/ORBmatcher.cc:1353-1497):
 * the projection prologue (1364-1405) runs on the device too -- x3Dc = Rcw x3Dw + tcw, invzc, (u, v), the image-bounds tests,
 * bForward / bBackward from tlc = Rlw twc + tlw against mb, radius = th * mvScaleFactors[octave of the last frame's feature], the octave
 * band and ur = u - mbf invzc -- and feeds the search core where it stands: per query only the map point's world position crosses PCIe.
 * Tcw / Tlw: rows 0..2 of CurrentFrame.mTcw / LastFrame.mTcw, row-major 3 x 4.  Query i = last-frame feature q_index[i] whose map point
 * (not an outlier) has world position x3Dw[3 i ..]; q_blocks / q_desc / t_occupied as in hvo_stream_search_by_projection.  A point that
 * fails a projection test is never searched (match_idx -1).  q_uv (may be NULL, nq x 2): the projections, 1e30 where none was searched. */
typedef struct { float fx, fy, cx, cy, bf, b; } hvo_camera;      /* Frame::fx fy cx cy mbf mb */
int  hvo_stream_project_last(hvo_stream *s, int64_t cur, int64_t last, const hvo_camera *cam, const float Tcw[12], const float Tlw[12],
                             int nq, const int32_t *q_index, const float *x3Dw, const uint8_t *q_blocks, const uint8_t *q_desc,
                             const uint8_t *t_occupied, float th, int mono, int th_high, int check_orientation,
                             int32_t *match_idx, int32_t *match_dist, int *n_matches, float *q_uv);
/* line matching between two resident frames: query = lines of `from`, train = lines of `to`; matches12 needs kl_cap entries,
 * *n_from receives n_kl(from) */
int  hvo_stream_match_lines(hvo_stream *s, int64_t from, int64_t to, int mode, float th, float nnratio, int32_t *matches12, int *n_from, int *n_matches);

/* LSDmatcher::SearchByGeomNApearance(CurrentFrame, LastFrame, desc_th, matches_12) WHOLE between two resident frames (reference
 * src/Tracking.cc:2299 -> src/LSDmatcher.cpp:36-108; computeAngle2D 20-34): match(Last.mLdesc, Cur.mLdesc) and then, per pair, the 20-degree
 * angle gate on the in-octave end points and the position gate (start OR end point within a tenth of the image bounds in both axes).
 * last_has_mapline (n_kl(last) flags, NULL = every line has one): LastFrame.mvpMapLines[i] != NULL; a line without one is passed over and keeps
 * its descriptor match in matches12, as in the reference (57).  accepted[i] = 1 where the reference assigns CurrentFrame.mvpMapLines[matches12[i]]
 * = LastFrame.mvpMapLines[i]; *n_accepted = the reference's return value.  matches12 / accepted need kl_cap entries. */
int  hvo_stream_match_lines_geom(hvo_stream *s, int64_t cur, int64_t last, float desc_th, const uint8_t *last_has_mapline,
                                 int32_t *matches12, uint8_t *accepted, int *n_last, int *n_accepted);
/* LSDmatcher::SearchByProjection(CurrentFrame, LastFrame, th) core between two resident frames (reference src/LSDmatcher.cpp:561-662 over
 * Frame::GetFeaturesInAreaForLine src/Frame.cc:1557-1627), the tracker's retry when the descriptor match finds too few lines.  Query i =
 * last-frame line q_index[i] whose map line passed isInFrustum: q_xyxy[4 i ..] = (mTrackProjX1, mTrackProjY1, mTrackProjX2, mTrackProjY2);
 * q_desc (nq x 32, may be NULL: the last frame's own descriptors) = pML->GetDescriptor(); q_blocks[i] != 0 when the map line has observations;
 * t_occupied (n_kl(cur) flags, may be NULL): current lines already holding an observed map line.  The current frame's key lines, line functions,
 * descriptors and line grid are the resident ones: the stream must run HVO_STAGE_GRIDS.  match_idx[i] = current line or -1 (accepted at <= 95). */
int  hvo_stream_search_lines_by_projection(hvo_stream *s, int64_t cur, int64_t last, int nq, const int32_t *q_index, const float *q_xyxy, const uint8_t *q_desc,
                                           const uint8_t *q_blocks, const uint8_t *t_occupied, float th, int32_t *match_idx, int32_t *match_dist, int *n_matches);

/* Page-lock (hipHostRegister) / unlock a caller's host buffer.  Images handed to hvo_batch_upload / hvo_stream_submit and result
 * slabs handed to hvo_batch_download move by DMA at the link rate when they are pinned (no staging copy on either side); equally
 * sized, equally spaced pinned destinations (e.g. labels8 of consecutive frames in one slab) take a single strided DMA. */
int hvo_pin_host(void *p, size_t bytes);
int hvo_unpin_host(void *p);

/* What the line growing of the last small batch / streamed frame fell back on (lsd_async.inc: several waves per frame, used for up to 16 frames):
 * frames that were grown again by the one-wave kernel because the workers gave up (never an error: the result is the same lines), workers that
 * found themselves on another XCD than their frame's and counted themselves out (exact, slower; > 0 means the dispatcher does not deal
 * workgroups b, b + 8, ... to one XCD on this system: set HVO_LSD_ASYNC=0), and the workers per frame of that launch (0: it was not async). */
int hvo_lsd_async_report(hvo_ctx *ctx, int *frames_regrown, int *foreign_workers, int *workers_per_frame);

/* ---- alternative readings of two un-vendored OpenCV calls (SURVEY.md Appendix A "(?)"; csrc/readings.hip) ----
 * The defaults (mask 0) are what the golden vectors pin.  A maintainer with the author's OpenCV 3.2 decides by running one
 * cv::GaussianBlur(ramp image, 7 x 7, sigma 2) and comparing it with both readings; the oracle has the same
 * switches (oracle.h orc_set_reading), and the parity tests run with both sides flipped. */
#define HVO_READING_BLUR_FLOAT 1u   /* cv::GaussianBlur on CV_8U served by IPP: float kernel, one rounding (ORB's 7x7 blur, LBD's 5x5 blur; with LSD_8U the detector's too) */
#define HVO_READING_LSD_8U     2u   /* cv::LineSegmentDetector working on CV_8U: u8 blur and u8 0.8x resize, gradients of the rounded bytes */
int hvo_set_readings(hvo_ctx *ctx, unsigned mask);            /* takes effect with the next extraction; HVO_ERR_INVALID_ARG for unknown bits */
int hvo_stream_set_readings(hvo_stream *s, unsigned mask);     /* every frame submitted afterwards */

/* ---- measurement hooks (bench.py) ---- */
/* Per-kernel-group device time of the last hvo_batch_run, measured with hipEvents on the ctx
 * stream.  names[i] points at static strings.  Returns the number of groups written (<= cap). */
int hvo_profile_last(const hvo_ctx *ctx, const char **names, float *ms, int cap);
/* 0: off (default); 1: hipEvent bracketing of each kernel group inside hvo_batch_run; 2: as 1 and the
 * three subsystems run back to back on one stream, so that group times are free of cross-stream contention */
int hvo_profile_enable(hvo_ctx *ctx, int on);

#ifdef __cplusplus
}
#endif
#endif /* HVO_H */
