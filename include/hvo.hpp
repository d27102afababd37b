// hvo.hpp -- header-only C++ mirror of the reference's front-end interfaces on top of the C ABI
// (include/hvo.h).  Same class names, constructor arguments, call operators and error behaviour as
//
//   ORB_SLAM2::ORBextractor      reference include/ORBextractor.h:47-116, src/ORBextractor.cc:408,1041
//   ORB_SLAM2::LINEextractor     reference include/LineExtractor.h:186-283, src/LineExtractor.cpp:329
//   PlaneDetection               reference include/PlaneExtractor.h:36-56,  src/PlaneExtractor.cpp:26-66
//   ORB_SLAM2::ORBmatcher        reference include/ORBmatcher.h:44,         src/ORBmatcher.cc:1676
//   ORB_SLAM2::LSDmatcher        reference include/LSDmatcher.h:43,         src/LSDmatcher.cpp:803-863
//
// but without OpenCV/Eigen types: images are (pointer, width, height, stride) and results are
// std::vectors of PODs whose layout equals cv::KeyPoint / cv::line_descriptor::KeyLine, so a
// Frame.cc adaptor is a reinterpret of vector storage.  No CPU fallback: every call
// throws hvo::Error when libhvo.so / a gfx950 device is unavailable.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "hvo.h"

namespace hvo {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &what) : std::runtime_error(what + ": " + hvo_strerror(s)), status(s) {}
};
inline void check(int rc, const char *what) { if (rc != HVO_OK) throw Error(rc, what); }

using KeyPoint = hvo_keypoint;   // == cv::KeyPoint (28 B)
using KeyLine = hvo_keyline;     // == cv::line_descriptor::KeyLine (68 B)
using Plane = hvo_plane;
static_assert(sizeof(KeyPoint) == 28 && sizeof(KeyLine) == 68, "layouts must match OpenCV's");

struct Image8 { const uint8_t *data; int width, height, stride; bool empty() const { return !data || width <= 0 || height <= 0; } };
struct Image16 { const uint16_t *data; int width, height, stride; bool empty() const { return !data || width <= 0 || height <= 0; } };

// shared ownership of one hvo_ctx (one HIP stream set; NOT thread-safe, like ORBextractor)
class Context {
public:
    explicit Context(const hvo_params &p) { check(hvo_create(&p, &ctx_), "hvo_create"); }
    ~Context() { hvo_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    hvo_ctx *get() const { return ctx_; }
private:
    hvo_ctx *ctx_ = nullptr;
};

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0)
        : nfeatures_(nfeatures), nlevels_(nlevels), scaleFactor_(scaleFactor)
    {
        hvo_params p; hvo_default_params(&p);
        p.orb_nfeatures = nfeatures; p.orb_scale_factor = scaleFactor; p.orb_nlevels = nlevels;
        p.orb_ini_th_fast = iniThFAST; p.orb_min_th_fast = minThFAST; p.device = device;
        ctx_.reset(new Context(p));
        // scale tables exactly as ORBextractor.cc:413-428
        mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
        for (int i = 1; i < nlevels; i++) { mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor; mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i]; }
        for (int i = 0; i < nlevels; i++) { mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i]; mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i]; }
    }
    // operator()(image, mask, keypoints, descriptors): the mask is ignored (ORBextractor.h:58)
    void operator()(const Image8 &image, std::vector<KeyPoint> &keypoints, std::vector<uint8_t> &descriptors)
    {
        keypoints.clear(); descriptors.clear();
        if (image.empty()) return;                                   // ORBextractor.cc:1044
        const int cap = nfeatures_ + 8 * nlevels_ + 64;
        keypoints.resize(cap); descriptors.resize((size_t)cap * 32);
        int n = 0;
        check(hvo_extract_orb(ctx_->get(), image.data, image.width, image.height, image.stride, keypoints.data(), descriptors.data(), cap, &n), "hvo_extract_orb");
        keypoints.resize(n); descriptors.resize((size_t)n * 32);
    }
    int GetLevels() const { return nlevels_; }
    float GetScaleFactor() const { return scaleFactor_; }
    const std::vector<float> &GetScaleFactors() const { return mvScaleFactor; }
    const std::vector<float> &GetInverseScaleFactors() const { return mvInvScaleFactor; }
    const std::vector<float> &GetScaleSigmaSquares() const { return mvLevelSigma2; }
    const std::vector<float> &GetInverseScaleSigmaSquares() const { return mvInvLevelSigma2; }
    hvo_ctx *ctx() const { return ctx_->get(); }
private:
    int nfeatures_, nlevels_; float scaleFactor_;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::unique_ptr<Context> ctx_;
};

class LINEextractor {
public:
    // LINEextractor(numOctaves, scale, nLSDFeature, min_line_length) (LineExtractor.h:190)
    LINEextractor(int numOctaves, float scale, unsigned nLSDFeature, double /*min_line_length*/ = 0, int device = 0)
        : nfeat_((int)nLSDFeature), numOctaves_(numOctaves), scale_(scale)
    {
        hvo_params p; hvo_default_params(&p);
        p.lsd_num_octaves = numOctaves; p.lsd_scale = scale; p.lsd_nfeatures = (int)nLSDFeature; p.device = device;
        ctx_.reset(new Context(p));
    }
    // operator()(image, mask, keylines, descriptors, lineVec2d); lineVec2d is n x 3 doubles
    void operator()(const Image8 &image, std::vector<KeyLine> &keylines, std::vector<uint8_t> &descriptors, std::vector<double> &lineVec2d)
    {
        keylines.clear(); descriptors.clear(); lineVec2d.clear();
        if (image.empty()) return;                                   // LineExtractor.cpp:331-332
        const int cap = nfeat_ > 0 ? nfeat_ : 1;
        keylines.resize(cap); descriptors.resize((size_t)cap * 32); lineVec2d.resize((size_t)cap * 3);
        int n = 0;
        check(hvo_extract_lsd(ctx_->get(), image.data, image.width, image.height, image.stride, keylines.data(), descriptors.data(), lineVec2d.data(), cap, &n), "hvo_extract_lsd");
        keylines.resize(n); descriptors.resize((size_t)n * 32); lineVec2d.resize((size_t)n * 3);
    }
    // Frame::ExtractLSD up to and including cullingLine(im, dis, angle, endpoint_dis, .) (Frame.cc:895-934, 952-1116):
    // the extractor's lines with near-collinear segments merged, re-sorted, re-described
    void ExtractLSDCulled(const Image8 &image, std::vector<KeyLine> &keylines, std::vector<uint8_t> &descriptors, std::vector<double> &lineVec2d,
                          double dis = 5, double angle = 2.5, double endpoint_dis = 15)
    {
        keylines.clear(); descriptors.clear(); lineVec2d.clear();
        if (image.empty()) return;
        const int cap = nfeat_ > 0 ? nfeat_ : 1;
        keylines.resize(cap); descriptors.resize((size_t)cap * 32); lineVec2d.resize((size_t)cap * 3);
        int n = 0;
        check(hvo_set_line_culling(ctx_->get(), dis, angle, endpoint_dis), "hvo_set_line_culling");
        check(hvo_extract_lsd_culled(ctx_->get(), image.data, image.width, image.height, image.stride, keylines.data(), descriptors.data(), lineVec2d.data(), cap, &n), "hvo_extract_lsd_culled");
        keylines.resize(n); descriptors.resize((size_t)n * 32); lineVec2d.resize((size_t)n * 3);
    }
    // Frame::isLineGood (Frame.cc:1205-1322): the 3-D line of every key line from the raw depth image; intrinsics and depth factor
    // are the constructor's (setCamera); the reference's time-seeded rand() is an explicit seed here
    void setCamera(float fx, float fy, float cx, float cy, float depthMapFactor)
    {
        hvo_params p; hvo_default_params(&p);
        p.lsd_num_octaves = numOctaves_; p.lsd_scale = scale_; p.lsd_nfeatures = nfeat_; p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy; p.depth_map_factor = depthMapFactor;
        ctx_.reset(new Context(p));
    }
    void isLineGood(const std::vector<KeyLine> &keylines, const Image16 &depth, uint32_t seed, std::vector<hvo_line3d> &lines3d)
    {
        lines3d.assign(keylines.size(), hvo_line3d());
        if (keylines.empty() || depth.empty()) return;
        check(hvo_lines_3d(ctx_->get(), keylines.data(), (int)keylines.size(), depth.data, depth.width, depth.height, depth.stride, seed, lines3d.data()), "hvo_lines_3d");
    }
    // the vanishing-point block of the Frame constructor (Frame.cc:328-337): tmp_vps and local_vp_ids (3 = no structure line)
    hvo_vp_result line2Vps(const std::vector<KeyLine> &keylines, uint32_t seed, std::vector<int32_t> &vp_idx, double thAngle = 1.0 / 180.0 * 3.14159265358979323846)
    {
        hvo_vp_result r; vp_idx.assign(keylines.size(), 3);
        check(hvo_vanishing_points(ctx_->get(), keylines.data(), (int)keylines.size(), seed, thAngle, &r, vp_idx.data(), nullptr), "hvo_vanishing_points");
        return r;
    }
    int GetLevels() const { return numOctaves_; }
    float GetScaleFactor() const { return scale_; }
private:
    int nfeat_, numOctaves_; float scale_;
    std::unique_ptr<Context> ctx_;
};

class PlaneDetection {
public:
    std::vector<std::vector<int>> plane_vertices_;   // vertex indices each plane contains (PlaneExtractor.h:43)
    std::vector<Plane> planes;                        // plane_filter.extractedPlanes (normal, center, mse, N)
    std::vector<int32_t> membership;                  // plane_filter.membershipImg, -1 = none
    int plane_num_ = 0;

    explicit PlaneDetection(int device = 0) : device_(device) {}
    // readDepthImage(depthImg, K, kScaleFactor): CV_16U only (PlaneExtractor.cpp:34-38) -> false otherwise
    bool readDepthImage(const Image16 &depth, float fx, float fy, float cx, float cy, float kScaleFactor)
    {
        if (depth.empty()) return false;
        depth_ = depth;
        if (!ctx_ || fx != fx_ || fy != fy_ || cx != cx_ || cy != cy_ || kScaleFactor != sf_) {
            hvo_params p; hvo_default_params(&p);
            p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy; p.depth_map_factor = kScaleFactor; p.device = device_;
            ctx_.reset(new Context(p));
            fx_ = fx; fy_ = fy; cx_ = cx; cy_ = cy; sf_ = kScaleFactor;
        }
        return true;
    }
    void runPlaneDetection()
    {
        const int w = depth_.width, h = depth_.height;
        membership.assign((size_t)w * h, -1);
        planes.resize(64);
        int n = 0;
        check(hvo_compute_planes(ctx_->get(), depth_.data, w, h, depth_.stride, membership.data(), planes.data(), 64, &n), "hvo_compute_planes");
        planes.resize(n); plane_num_ = n;
        plane_vertices_.assign(n, std::vector<int>());
        for (int i = 0; i < w * h; i++) if (membership[i] >= 0) plane_vertices_[membership[i]].push_back(i);   // raster order, like refineDetails
    }
    // the tail of Frame::ComputePlanes (Frame.cc:2110-2212): per plane the 0.1 m voxel cloud, the distance gate and the RANSAC refit
    // (mvPlanePoints / mvPlaneCoefficients = the entries with valid == 1), and the integral-image surface normals (vSurfaceNormal)
    void planeClouds(double distanceThreshold, std::vector<hvo_plane_cloud> &clouds, std::vector<float> &xyz, int cap = 200000)
    {
        clouds.assign(planes.size(), hvo_plane_cloud()); xyz.assign((size_t)cap * 3, 0.f);
        int total = 0;
        if (!planes.empty())
            check(hvo_plane_clouds(ctx_->get(), depth_.data, depth_.width, depth_.height, depth_.stride, membership.data(), planes.data(), (int)planes.size(),
                                   distanceThreshold, xyz.data(), cap, clouds.data(), &total), "hvo_plane_clouds");
        xyz.resize((size_t)total * 3);
    }
    void surfaceNormals(std::vector<hvo_surface_normal> &normals)
    {
        const int cap = (depth_.height / 3 / 2 + 1) * (depth_.width / 3 / 2 + 1);
        normals.assign(cap, hvo_surface_normal());
        int n = 0;
        check(hvo_surface_normals(ctx_->get(), depth_.data, depth_.width, depth_.height, depth_.stride, normals.data(), cap, &n), "hvo_surface_normals");
        normals.resize(n);
    }
private:
    int device_; Image16 depth_{ nullptr, 0, 0, 0 };
    float fx_ = 0, fy_ = 0, cx_ = 0, cy_ = 0, sf_ = 0;
    std::unique_ptr<Context> ctx_;
};

// Frame post-processing of the extractor outputs: what the Frame constructor does right after
// ExtractORB / ExtractLSD (Frame.cc:231-262): UndistortKeyPoints, ComputeImageBounds, AssignFeaturesToGrid,
// AssignFeaturesToGridForLine.  Grids are CSR (cell = col * 48 + row), see hvo.h.
class FrameGrid {
public:
    static const int COLS = HVO_GRID_COLS, ROWS = HVO_GRID_ROWS;
    explicit FrameGrid(hvo_ctx *ctx) : ctx_(ctx) {}
    void UndistortKeyPoints(const std::vector<KeyPoint> &keys, const float distCoef[5], std::vector<KeyPoint> &keysUn) const
    {
        keysUn.resize(keys.size());
        check(hvo_undistort_keypoints(ctx_, keys.data(), (int)keys.size(), distCoef, keysUn.data()), "hvo_undistort_keypoints");
    }
    void ComputeImageBounds(int cols, int rows, const float distCoef[5], float &mnMinX, float &mnMaxX, float &mnMinY, float &mnMaxY) const
    {
        float b[4]; check(hvo_image_bounds(ctx_, cols, rows, distCoef, b), "hvo_image_bounds");
        mnMinX = b[0]; mnMaxX = b[1]; mnMinY = b[2]; mnMaxY = b[3];
    }
    // mGrid[col][row] == items[start[col*48+row] .. start[col*48+row+1])
    void AssignFeaturesToGrid(const std::vector<KeyPoint> &keysUn, const float bounds[4], std::vector<int> &start, std::vector<int> &items) const
    {
        start.assign(COLS * ROWS + 1, 0); items.assign(keysUn.size() ? keysUn.size() : 1, 0);
        int n = 0;
        check(hvo_assign_features_to_grid(ctx_, keysUn.data(), (int)keysUn.size(), bounds, start.data(), items.data(), &n), "hvo_assign_features_to_grid");
        items.resize(n);
    }
    void AssignFeaturesToGridForLine(const std::vector<KeyLine> &keylines, const float bounds[4], std::vector<int> &start, std::vector<int> &items) const
    {
        start.assign(COLS * ROWS + 1, 0); items.assign(keylines.size() * 128 + 1, 0);
        int n = 0;
        check(hvo_assign_lines_to_grid(ctx_, keylines.data(), (int)keylines.size(), bounds, start.data(), items.data(), (int)items.size(), &n), "hvo_assign_lines_to_grid");
        items.resize(n);
    }
private:
    hvo_ctx *ctx_;
};

class ORBmatcher {
public:
    static const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;       // ORBmatcher.cc:37-39
    explicit ORBmatcher(hvo_ctx *ctx) : ctx_(ctx) {}
    // DescriptorDistance(a, b): 32-byte descriptors
    int DescriptorDistance(const uint8_t *a, const uint8_t *b) const
    {
        uint16_t d = 0; check(hvo_hamming_matrix(ctx_, a, 1, b, 1, &d), "hvo_hamming_matrix"); return d;
    }
    // all-pairs distances for the guided searches (nq x nt, row-major)
    void DistanceMatrix(const uint8_t *q, int nq, const uint8_t *t, int nt, std::vector<uint16_t> &d) const
    {
        d.resize((size_t)nq * nt); check(hvo_hamming_matrix(ctx_, q, nq, t, nt, d.data()), "hvo_hamming_matrix");
    }
    // SearchByProjection(CurrentFrame, LastFrame, th, mono) core (ORBmatcher.cc:1353-1497): see hvo.h.
    // Returns the number of matches; match_idx[i] is the current-frame feature of query i or -1.
    int SearchByProjection(const uint8_t *q_desc, int nq, const float *q_u, const float *q_v, const float *q_radius,
                           const int32_t *q_min_level, const int32_t *q_max_level, const float *q_ur, const float *q_angle,
                           const uint8_t *q_blocks, const KeyPoint *t_kp, const float *t_uright, const uint8_t *t_occupied,
                           const uint8_t *t_desc, int nt, float mnMinX, float mnMinY, float mnMaxX, float mnMaxY,
                           bool checkOrientation, std::vector<int> &match_idx) const
    {
        match_idx.assign(nq, -1);
        std::vector<int> dist(nq);
        int n = 0;
        check(hvo_search_by_projection(ctx_, q_desc, nq, q_u, q_v, q_radius, q_min_level, q_max_level, q_ur, q_angle, q_blocks, t_kp, t_uright,
                                       t_occupied, t_desc, nt, mnMinX, mnMinY, mnMaxX, mnMaxY, TH_HIGH, checkOrientation ? 1 : 0,
                                       match_idx.data(), dist.data(), &n), "hvo_search_by_projection");
        return n;
    }
    // SearchByProjection(F, vpMapPoints, th) core (ORBmatcher.cc:45-132): best / second best with the same-octave
    // ratio test (mfNNratio); see hvo.h.  Returns the number of matches.
    int SearchByProjectionMap(const uint8_t *q_desc, int nq, const float *q_u, const float *q_v, const float *q_radius,
                              const int32_t *q_min_level, const int32_t *q_max_level, const float *q_ur, const uint8_t *q_blocks,
                              const KeyPoint *t_kp, const float *t_uright, const uint8_t *t_occupied, const uint8_t *t_desc, int nt,
                              float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, float nnratio, std::vector<int> &match_idx) const
    {
        match_idx.assign(nq, -1);
        std::vector<int> dist(nq);
        int n = 0;
        check(hvo_search_by_projection_map(ctx_, q_desc, nq, q_u, q_v, q_radius, q_min_level, q_max_level, q_ur, q_blocks, t_kp, t_uright,
                                           t_occupied, t_desc, nt, mnMinX, mnMinY, mnMaxX, mnMaxY, TH_HIGH, nnratio,
                                           match_idx.data(), dist.data(), &n), "hvo_search_by_projection_map");
        return n;
    }
    // SearchByProjection(F, vpMapPoints, th) from the tracker's own per-point fields: the window prologue (ORBmatcher.cc:55-70,
    // RadiusByViewingCos 134-140) runs on the device.  One entry per map point with mbTrackInView && !isBad().
    int SearchByProjection(const uint8_t *mp_desc, int nq, const float *mTrackProjX, const float *mTrackProjY, const float *mTrackProjXR,
                           const int32_t *mnTrackScaleLevel, const float *mTrackViewCos, const uint8_t *q_blocks, float th,
                           const KeyPoint *t_kp, const float *t_uright, const uint8_t *t_occupied, const uint8_t *t_desc, int nt,
                           float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, float nnratio, std::vector<int> &match_idx) const
    {
        match_idx.assign(nq, -1);
        std::vector<int> dist(nq);
        int n = 0;
        check(hvo_search_by_projection_tracked(ctx_, mp_desc, nq, mTrackProjX, mTrackProjY, mTrackProjXR, mnTrackScaleLevel, mTrackViewCos, q_blocks, th,
                                               t_kp, t_uright, t_occupied, t_desc, nt, mnMinX, mnMinY, mnMaxX, mnMaxY, TH_HIGH, nnratio,
                                               match_idx.data(), dist.data(), &n), "hvo_search_by_projection_tracked");
        return n;
    }
private:
    hvo_ctx *ctx_;
};

class LSDmatcher {
public:
    static const int TH_HIGH = 80, TH_LOW = 50;                            // LSDmatcher.cpp:12-14
    // FrameBFMatch(ldesc1, ldesc2, LineMatches, TH) (LSDmatcher.cpp:942-966)
    void FrameBFMatch(const uint8_t *ldesc1, int n1, const uint8_t *ldesc2, int n2, std::vector<int> &LineMatches, float TH, float nnratio) const
    {
        LineMatches.assign(n1 > 0 ? n1 : 1, -1); int n = 0;
        check(hvo_frame_bf_match(ctx_, ldesc1, n1, ldesc2, n2, TH, nnratio, LineMatches.data(), &n), "hvo_frame_bf_match");
        LineMatches.resize(n1);
    }
    // SearchDouble(InitialFrame, CurrentFrame, LineMatches) core (LSDmatcher.cpp:902-939): two-way FrameBFMatch at TH_LOW
    int SearchDouble(const uint8_t *ldesc1, int n1, const uint8_t *ldesc2, int n2, std::vector<int> &LineMatches, float nnratio) const
    {
        LineMatches.assign(n1 > 0 ? n1 : 1, -1); int n = 0;
        check(hvo_search_double(ctx_, ldesc1, n1, ldesc2, n2, (float)TH_LOW, nnratio, LineMatches.data(), &n), "hvo_search_double");
        LineMatches.resize(n1);
        return n;
    }
    explicit LSDmatcher(hvo_ctx *ctx) : ctx_(ctx) {}
    // int match(desc1, desc2, nnr, matches_12) -> matchNNR (LSDmatcher.cpp:828-863, 803-826)
    int match(const uint8_t *desc1, int n1, const uint8_t *desc2, int n2, float nnr, std::vector<int> &matches_12) const
    {
        matches_12.assign(n1, -1);
        int m = 0;
        check(hvo_match_nnr(ctx_, desc1, n1, desc2, n2, nnr, matches_12.data(), &m), "hvo_match_nnr");
        return m;
    }
    // int SearchByGeomNApearance(CurrentFrame, LastFrame, desc_th, matches_12) (LSDmatcher.cpp:36-108) on host arrays: the descriptor match and the
    // angle / end-point gates; accepted[i1] != 0 where the reference assigns CurrentFrame.mvpMapLines[matches_12[i1]] = LastFrame.mvpMapLines[i1]
    int SearchByGeomNApearance(const uint8_t *ldescLast, const hvo_keyline *klLast, const uint8_t *lastHasMapLine, int nLast,
                               const uint8_t *ldescCur, const hvo_keyline *klCur, int nCur, float desc_th, const float bounds4[4],
                               std::vector<int> &matches_12, std::vector<uint8_t> &accepted) const
    {
        matches_12.assign(nLast > 0 ? nLast : 1, -1); accepted.assign(nLast > 0 ? nLast : 1, 0);
        int n = 0;
        check(hvo_match_lines_geom(ctx_, ldescLast, klLast, lastHasMapLine, nLast, ldescCur, klCur, nCur, desc_th, bounds4, matches_12.data(), accepted.data(), &n), "hvo_match_lines_geom");
        matches_12.resize(nLast); accepted.resize(nLast);
        return n;
    }
    // int SearchByProjection(CurrentFrame, LastFrame, th) core (LSDmatcher.cpp:561-662 over Frame::GetFeaturesInAreaForLine, Frame.cc:1557-1627): one query per
    // last-frame map line in view (its projected end points, its key line, its descriptor, whether it has observations); the current frame's key lines, line
    // functions, descriptors, occupied flags and line grid (hvo::FrameGrid / hvo_assign_lines_to_grid)
    int SearchByProjection(int nq, const float *q_xyxy, const hvo_keyline *q_kl, const uint8_t *q_desc, const uint8_t *q_blocks,
                           const hvo_keyline *t_kl, const double *t_linefn, const uint8_t *t_desc, const uint8_t *t_occupied, int nt,
                           const int32_t *cell_start, const int32_t *cell_items, const float bounds4[4], float th, std::vector<int32_t> &match_idx) const
    {
        match_idx.assign(nq > 0 ? nq : 1, -1); std::vector<int32_t> dist(nq > 0 ? nq : 1, 256);
        int n = 0;
        check(hvo_search_lines_by_projection(ctx_, nq, q_xyxy, q_kl, q_desc, q_blocks, t_kl, t_linefn, t_desc, t_occupied, nt, cell_start, cell_items, bounds4, th,
                                             match_idx.data(), dist.data(), &n), "hvo_search_lines_by_projection");
        match_idx.resize(nq);
        return n;
    }
private:
    hvo_ctx *ctx_;
};


// One camera, one frame at a time (Tracking.cc:262 -> Frame.cc:205-233, then the matching of Tracking.cc:2299 / 2396): a ring of
// frames in flight whose results stay on the device for matching against the previous frame (hvo_stream_*, see hvo.h).
class FrameStream {
public:
    FrameStream(const hvo_params &p, const hvo_stream_params &sp)
    {
        check(hvo_stream_create(&p, &sp, &s_), "hvo_stream_create");
        check(hvo_stream_capacity(s_, &kp_cap_, &kl_cap_, &pl_cap_), "hvo_stream_capacity");
    }
    ~FrameStream() { if (s_) hvo_stream_destroy(s_); }
    FrameStream(const FrameStream &) = delete;
    FrameStream &operator=(const FrameStream &) = delete;
    int64_t submit(const Image8 &gray, const Image16 &depth)
    {
        int64_t t = -1;
        check(hvo_stream_submit(s_, gray.data, gray.stride, depth.data, depth.stride, &t), "hvo_stream_submit");
        return t;
    }
    bool done(int64_t ticket) { const int r = hvo_stream_poll(s_, ticket); if (r < 0) throw Error(r, "hvo_stream_poll"); return r == 1; }
    // waits for the frame; any pointer of out may be null (hvo_batch_download's conventions)
    void collect(int64_t ticket, hvo_frame_out &out, hvo_keypoint *kp_un = nullptr, float *uright = nullptr, float *zdepth = nullptr)
    {
        check(hvo_stream_collect(s_, ticket, &out, kp_un, uright, zdepth), "hvo_stream_collect");
    }
    // The rest of the Frame constructor (HVO_STAGE_LINES3D / _VP / _PLANE_TAIL / _GRIDS in hvo_stream_params.stages): isLineGood's 3-D
    // lines, the vanishing points and line2Vps clusters, mvPlanePoints / mvPlaneCoefficients / vSurfaceNormal, the two 64 x 48 grids.
    // Call before collect() releases the slot.  FrameTail owns its arrays, sized from the stream's capacities.
    struct FrameTail {
        std::vector<hvo_line3d> lines3d; hvo_vp_result vp{}; std::vector<int32_t> vp_idx;
        std::vector<hvo_plane_cloud> plane_clouds; std::vector<float> cloud_xyz; std::vector<hvo_surface_normal> normals;
        std::vector<int32_t> pt_cell_start, pt_cell_items, ln_cell_start, ln_cell_items;
        hvo_frame_tail c{};
    };
    void collectTail(int64_t ticket, int w, int h, FrameTail &t)
    {
        int cc = 0, nn = 0, lc = 0;
        check(hvo_tail_capacity(kl_cap_, w, h, &cc, &nn, &lc), "hvo_tail_capacity");
        t.lines3d.resize(kl_cap_); t.vp_idx.assign(kl_cap_, 3); t.plane_clouds.resize(64); t.cloud_xyz.resize(3 * (size_t)cc); t.normals.resize(nn > 0 ? nn : 1);
        t.pt_cell_start.resize(HVO_GRID_COLS * HVO_GRID_ROWS + 1); t.pt_cell_items.resize(kp_cap_ > 0 ? kp_cap_ : 1);
        t.ln_cell_start.resize(HVO_GRID_COLS * HVO_GRID_ROWS + 1); t.ln_cell_items.resize(lc > 0 ? lc : 1);
        hvo_frame_tail &c = t.c;
        c.lines3d = t.lines3d.data(); c.vp = &t.vp; c.vp_idx = t.vp_idx.data(); c.plane_clouds = t.plane_clouds.data(); c.cloud_xyz = t.cloud_xyz.data(); c.cloud_cap = cc;
        c.normals = t.normals.data(); c.normals_cap = nn; c.pt_cell_start = t.pt_cell_start.data(); c.pt_cell_items = t.pt_cell_items.data(); c.pt_items_cap = kp_cap_;
        c.ln_cell_start = t.ln_cell_start.data(); c.ln_cell_items = t.ln_cell_items.data(); c.ln_items_cap = lc;
        check(hvo_stream_collect_tail(s_, ticket, &c), "hvo_stream_collect_tail");
    }
    // ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) whole (ORBmatcher.cc:1353-1497) between two resident frames: the
    // projection prologue (1364-1405) and the search core on the device.  Tcw / Tlw: rows 0..2 of the frames' mTcw (row-major 3 x 4);
    // q_index[i] = last-frame feature whose map point (no outlier) has world position x3Dw[3 i ..].  Returns the number of matches.
    int SearchByProjection(int64_t cur, int64_t last, const hvo_camera &cam, const float *Tcw, const float *Tlw, int nq, const int32_t *q_index,
                           const float *x3Dw, const uint8_t *q_blocks, const uint8_t *t_occupied, float th, bool bMono, bool checkOrientation,
                           std::vector<int32_t> &match_idx)
    {
        match_idx.assign(nq, -1);
        std::vector<int32_t> dist(nq > 0 ? nq : 1);
        int n = 0;
        check(hvo_stream_project_last(s_, cur, last, &cam, Tcw, Tlw, nq, q_index, x3Dw, q_blocks, nullptr, t_occupied, th, bMono ? 1 : 0, 100,
                                      checkOrientation ? 1 : 0, match_idx.data(), dist.data(), &n, nullptr), "hvo_stream_project_last");
        return n;
    }
    // LSDmatcher::match / FrameBFMatch / SearchDouble between two resident frames (mode = HVO_LINE_MATCH_*)
    int matchLines(int64_t from, int64_t to, int mode, float th, float nnratio, std::vector<int32_t> &matches12)
    {
        matches12.assign(kl_cap_, -1);
        int nfrom = 0, nm = 0;
        check(hvo_stream_match_lines(s_, from, to, mode, th, nnratio, matches12.data(), &nfrom, &nm), "hvo_stream_match_lines");
        matches12.resize(nfrom);
        return nm;
    }
    // LSDmatcher::SearchByGeomNApearance(Cur, Last, desc_th, matches_12) whole between two resident frames (LSDmatcher.cpp:36-108)
    int searchByGeomNApearance(int64_t cur, int64_t last, float desc_th, const uint8_t *lastHasMapLine, std::vector<int32_t> &matches12, std::vector<uint8_t> &accepted)
    {
        matches12.assign(kl_cap_, -1); accepted.assign(kl_cap_, 0);
        int nlast = 0, nm = 0;
        check(hvo_stream_match_lines_geom(s_, cur, last, desc_th, lastHasMapLine, matches12.data(), accepted.data(), &nlast, &nm), "hvo_stream_match_lines_geom");
        matches12.resize(nlast); accepted.resize(nlast);
        return nm;
    }
    // LSDmatcher::SearchByProjection(Cur, Last, th) core between two resident frames (LSDmatcher.cpp:561-662); the stream must run HVO_STAGE_GRIDS
    int searchLinesByProjection(int64_t cur, int64_t last, const std::vector<int32_t> &q_index, const std::vector<float> &q_xyxy, const uint8_t *q_desc,
                                const uint8_t *q_blocks, const uint8_t *t_occupied, float th, std::vector<int32_t> &match_idx)
    {
        const int nq = (int)q_index.size();
        match_idx.assign(nq > 0 ? nq : 1, -1); std::vector<int32_t> dist(nq > 0 ? nq : 1, 256);
        int n = 0;
        check(hvo_stream_search_lines_by_projection(s_, cur, last, nq, q_index.data(), q_xyxy.data(), q_desc, q_blocks, t_occupied, th, match_idx.data(), dist.data(), &n),
              "hvo_stream_search_lines_by_projection");
        match_idx.resize(nq);
        return n;
    }
    void setReadings(unsigned mask) { check(hvo_stream_set_readings(s_, mask), "hvo_stream_set_readings"); }
    hvo_stream *get() const { return s_; }
    int kpCap() const { return kp_cap_; } int klCap() const { return kl_cap_; } int plCap() const { return pl_cap_; }
private:
    hvo_stream *s_ = nullptr; int kp_cap_ = 0, kl_cap_ = 0, pl_cap_ = 0;
};

}  // namespace hvo
