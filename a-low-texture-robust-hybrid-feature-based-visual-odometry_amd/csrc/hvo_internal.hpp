// hvo_internal.hpp -- shared declarations of libhvo.so (not part of the ABI).
//
// One hvo_ctx owns one HIP stream and every device slab of a batch ("plan") for a given image
// geometry.  Stages are written as batch kernels: grid.y (or a flattened index) walks the frames
// of the resident batch so that launch latency is amortised over the batch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/hvo.h"

#define HVO_MAX_LEVELS 16
#define HVO_EDGE_THRESHOLD 19      // ORBextractor.cc:72
#define HVO_CELL_TILE 72           // largest FAST cell view (wCell+6) the LDS tile holds
#define HVO_CELL_CAP 256           // max NMS survivors kept per cell
#define HVO_MAX_PROFILE 768        // intervals, not groups: a group that runs once per chunk of the batch has one interval per chunk
#define HVO_HAVE_PEAC 1           // peac.hip is built (stubs.hip drops its PEAC stubs)
#define HVO_HAVE_LSD 1            // lsd.hip is built

struct LevelGeom {
    int w, h, pitch;
    int minBX, minBY, maxBX, maxBY;     // ORBextractor.cc:771-774
    int nCols, nRows, wCell, hCell;     // ORBextractor.cc:782-785
    int cell_off, ncells;               // into the per-frame cell table
    int nfeat;                          // mnFeaturesPerLevel
    int cand_off, cand_cap;             // per-frame candidate scratch (entries)
    int node_off, node_cap;             // per-frame quadtree node scratch (entries)
    int kp_off, kp_cap;                 // per-frame per-level selected keypoints
    int scaled_patch;                   // (int)(31*scale)
    float scale;
    unsigned long long img_off;         // byte offset of this level inside a frame's all-levels slab (the blurred pyramid)
    unsigned long long lvl_off;         // byte offset inside a frame's levels >= 1 slab (level 0 is the frame's input image, a slab of its own)
    int rs_off;                         // offset into resize tables (x entries), y at ry_off
    int ry_off;
    int tile_off, ntx, nty;             // blur tiles
};

struct CellDesc {   // one FAST cell view (ORBextractor.cc:787-806)
    short level, x0, y0, vw, vh, ox, oy, pad;
};

// one tile of the fused per-level pass (orb_level.hip): the rectangle it owns for blur / resize, the FAST interior of the
// reference's cell it holds (fw == 0: a margin tile), and the next level's pixels whose upper-left source pixel it owns
struct OrbTile {
    short bx0, by0, bw, bh;             // owned rectangle (bx0, bw multiples of 4)
    short fx0, fw;                      // FAST interior columns [fx0, fx0 + fw); interior rows = the owned rows
    short cell;                         // index into the frame's cell table (-1: none)
    short dxa, dxb, dya, dyb;           // next level: columns [dxa, dxb) x rows [dya, dyb)
    short pad[5];
};

struct OrbPlan {
    int w = 0, h = 0, nlevels = 0, batch = 0;
    LevelGeom lev[HVO_MAX_LEVELS];
    int ncells = 0, cand_total = 0, node_total = 0, kp_total = 0, ntiles = 0, max_cell = 0;
    size_t pyr_bytes = 0;               // stride of d_pyr: level 0 only (the input image, one per resident frame)
    size_t lvl_bytes = 0, blur_bytes = 0;  // strides of d_lvl (levels >= 1) and d_blur (all levels): scratch, one per frame of a CHUNK
    int chunk = 0;                      // frames the scratch slabs exist for: orb_run walks the batch chunk by chunk
    bool resize_dw[HVO_MAX_LEVELS] = {};  // level is produced by k_resize_dw (dword loads) instead of k_resize
    int last_chunks = 0;                  // chunks the last orb_run walked (hvo_debug_orb_plan)
    bool fused = false;                 // the fused per-level pass (orb_level.hip) serves this geometry
    int lt_off[HVO_MAX_LEVELS] = {}, lt_cnt[HVO_MAX_LEVELS] = {}, lt_tpw = 4;
    int lt_nw = 4;                        // waves per workgroup of k_orb_level (HVO_ORB_NW, read when the plan is built)
    OrbTile *d_ltiles = nullptr;
    int2 *d_kpchunks = nullptr; int n_kpchunks = 0;     // (level, first slot) of every 32-slot chunk of the per-level key-point slabs (orb_describe.hip)
    int kp_cap = 0;                     // output capacity per frame
    // device
    LevelGeom *d_lev = nullptr;
    CellDesc *d_cells = nullptr;
    int *d_rs_xofs = nullptr; int *d_rs_xalpha = nullptr;   // per level: dw entries
    int *d_rs_yofs = nullptr; int *d_rs_ybeta = nullptr;    // per level: dh entries (yofs packs sy0|sy1<<16)
    int4 *d_tiles = nullptr;            // blur tiles (level, tx, ty, 0)
    uint8_t *d_pyr = nullptr, *d_blur = nullptr;            // batch * pyr_bytes; chunk * blur_bytes
    uint8_t *d_lvl = nullptr, *d_lvl_base = nullptr;        // chunk * lvl_bytes (256 guard bytes in front, like d_pyr)
    uint8_t *d_pyr_base = nullptr;                          // the allocation behind d_pyr (256 guard bytes in front: orb_level.hip's 16-byte tile loads may start 4 bytes before a row)
    uint32_t *d_cell_kp = nullptr; int *d_cell_cnt = nullptr;
    uint32_t *d_cand = nullptr; int *d_keys = nullptr, *d_keys_tmp = nullptr;
    int oct_slot_cap = 0;               // quadtree nodes alive at once, any level (k_octree keeps them in LDS: 36 bytes a slot)
    uint32_t *d_lvl_kp = nullptr; int *d_lvl_cnt = nullptr;
    hvo_keypoint *d_kp = nullptr; uint8_t *d_desc = nullptr; int *d_nkp = nullptr;
    int *d_flags = nullptr;             // per frame error flags
};

struct ProfileRec { const char *name; hipEvent_t e0, e1; float ms; bool used; hipStream_t st; };

// A tuning variable (environment) as it stood when a context or a plan was BUILT: launches read the copy, never the environment
// (~25 getenv calls per launch; tests set a variable and make a new context, tools use HVO_LIB for A/B builds).
struct Knob {
    bool set = false; int v = 0;
    void read(const char *name) { const char *e = getenv(name); set = e != nullptr; v = e ? atoi(e) : 0; }
    int or_(int dflt) const { return set ? v : dflt; }
    bool off() const { return set && v == 0; }
};
// hipFuncAttributeMaxDynamicSharedMemorySize for a kernel that is about to be launched with `bytes` of dynamic LDS (> 48 KB): raised under a
// lock, per device -- contexts are driven from several host threads (api.hip)
int hvo_ensure_dyn_lds(const void *kernel, size_t bytes);

struct hvo_ctx {
    hvo_params p;
    Knob kn_frame_perm, kn_upload_single;  // HVO_FRAME_PERM, HVO_UPLOAD_SINGLE (read by hvo_create)
    unsigned readings = 0;                 // HVO_READING_* (hvo_set_readings): alternative readings of two OpenCV calls, readings.hip
    int device = 0;
    hipStream_t stream = nullptr;          // ORB + matching + uploads
    hipStream_t s_lsd = nullptr;           // LSD/LBD kernels   } the three subsystems are independent and run
    hipStream_t s_peac = nullptr;          // PEAC kernels      } concurrently, like Frame.cc:210-215's threads
    hipEvent_t ev_lsd_pre = nullptr;       // recorded on s_lsd when the streaming LSD kernels are done (before k_lsd_grow)
    bool lsd_pre_recorded = false;
    hipEvent_t ev_fast = nullptr;          // recorded on the ORB stream after k_fast_cells (the only ORB kernel that needs LDS)
    bool fast_recorded = false;
    int sched = 1;                         // overlap policy in force (hvo_batch_run sets it per batch), see api.hip
    int sched_cfg = -1;                    // HVO_SCHED, or -1 = by batch size: 5 from 3072 resident frames on, 1 below
    bool orb_blur_late = false;            // k_blur7 behind k_fast_cells instead of before it (HVO_ORB_BLUR_LATE)
    // double-buffered batches (hvo_batch_stage_upload / _commit_staged / _results_async): the NEXT batch's images go into staging slabs and
    // the LAST batch's results leave from a packed slab while the resident batch runs
    uint8_t *d_stage_gray = nullptr; uint16_t *d_stage_depth = nullptr; size_t stage_gray_bytes = 0, stage_depth_bytes = 0;
    uint8_t *stage_gray_dst = nullptr; uint16_t *stage_depth_dst = nullptr;      // when set, orb_upload / peac_upload write here instead of the resident slabs
    int stage_n = 0, stage_w = 0, stage_h = 0; bool stage_depth = false;
    hipStream_t s_stage_up = nullptr, s_stage_down = nullptr; hipEvent_t ev_stage_up = nullptr;
    char *d_result_slab = nullptr; size_t result_slab_bytes = 0;
    hipStream_t s_copy = nullptr; bool copy_hi = false;   // batch uploads / downloads go on a stream of their own, above the compute streams (hvo_copy_stream)
    std::vector<std::pair<int, int *>> perms;   // launch orders (hvo_frame_perm), one device array per length asked for
    double cull_dis = 5.0, cull_angle = 2.5, cull_endpoint = 15.0;   // Frame::cullingLine(im, 5, 2.5, 15, 30), Frame.cc:934
    bool last_cull = false;                // the resident batch was run with HVO_STAGE_LSD_CULL
    unsigned last_stages = 0;              // stages hvo_batch_run has computed for the resident batch (hvo_batch_download reports only these)
    std::string last_error;
    OrbPlan orb;
    // ORB tables
    int umax[16];
    float scale[HVO_MAX_LEVELS], inv_scale[HVO_MAX_LEVELS];
    int nfeat[HVO_MAX_LEVELS];
    int8_t *d_pattern = nullptr; int *d_umax = nullptr;
    // resident batch
    int batch_n = 0, batch_w = 0, batch_h = 0;
    bool have_depth = false;
    // matcher staging arena (match.hip)
    void *marena = nullptr;
    // host staging (pinned)
    void *h_stage = nullptr; size_t h_stage_cap = 0;
    hipEvent_t ev_stage[2] = { nullptr, nullptr };     // download staging (hvo_staged_d2h)
    // profiling
    bool profile = false;
    bool serialize = false;                // profiling mode 2: all stages on one stream (clean per-kernel times)
    bool lsd_on_orb_stream = false;        // streamed mode: ORB (0.6 ms) then the line chain on one stream, planes on the other (two HW queues per frame in flight)
    ProfileRec prof[HVO_MAX_PROFILE]; int nprof = 0;
    // staging arena of the host-array entry points of the Frame tail (hvo_lines_3d, hvo_vanishing_points, hvo_plane_clouds, ...): one
    // grow-only device buffer per context instead of hipMalloc / hipFree per call (hvo_call_arena)
    void *call_arena = nullptr; size_t call_arena_cap = 0;
    void *tail = nullptr;                  // resident-batch Frame tail (tail.hip)
    // opaque per-subsystem state (peac.hip / lsd.hip own these)
    void *peac = nullptr;
    void *lsd = nullptr;
};

// Image-border handling of the 12-byte row window {W0, W1, W2} = pixels x0-4 .. x0+7 of a 4-pixel strip without a byte
// path: the three dwords are loaded from clamped in-row addresses and every window byte that lies outside the image is
// replaced by its REFLECT_101 source, which is always inside the same window for the taps of the strip's valid pixels
// (reach <= 3).  Three v_perm_b32 with per-thread selectors (identity for interior strips) and one operand select.
struct EdgeSel { unsigned s0, s1, s2; bool lo2; };
#ifdef __HIPCC__
static __device__ __forceinline__ EdgeSel edge_sel(int x0, int w)
{
    EdgeSel e; e.s0 = 0x03020100u; e.s1 = 0x07060504u; e.s2 = 0x07060504u;
    const int m = w - 1 - x0;                                  // last valid pixel relative to x0
    e.lo2 = m <= 2;                                            // W2' comes from {W0, W1} instead of {W1, W2}
    if (x0 >= 4 && m >= 7) return e;
    unsigned s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int b = 0; b < 12; b++) {
        const int px = x0 - 4 + b;
        int r = px < 0 ? -px : (px >= w ? 2 * (w - 1) - px : px);
        int sb = r - (x0 - 4);                                 // source byte in the window
        if (b < 8) { if (sb < 0 || sb > 7) sb = b; }            // bytes no valid pixel's taps reach: anything
        else if (e.lo2) { if (sb < 0 || sb > 7) sb = 7; }
        else { sb -= 4; if (sb < 0 || sb > 7) sb = b - 4; }
        if (b < 4) s0 |= (unsigned)sb << (8 * b);
        else if (b < 8) s1 |= (unsigned)sb << (8 * (b - 4));
        else s2 |= (unsigned)sb << (8 * (b - 8));
    }
    e.s0 = s0; e.s1 = s1; e.s2 = s2;
    return e;
}
static __device__ __forceinline__ void edge_fix(const EdgeSel &e, unsigned &W0, unsigned &W1, unsigned &W2)
{
    const unsigned a = e.lo2 ? W1 : W2, b = e.lo2 ? W0 : W1;
    const unsigned n0 = __builtin_amdgcn_perm(W1, W0, e.s0), n1 = __builtin_amdgcn_perm(W1, W0, e.s1);
    W2 = __builtin_amdgcn_perm(a, b, e.s2); W0 = n0; W1 = n1;
}
#endif

#define HVO_HIP(call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            ctx->last_error = std::string(#call) + ": " + hipGetErrorString(e_);            \
            return HVO_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

// Workgroup dispatch costs ~6 ns on MI355X: an EMPTY kernel over 834k tiny workgroups takes 5 ms.
// Streaming kernels therefore launch a bounded number of persistent workgroups that grid-stride over
// flattened work items (cdna_hip_programming.md Guideline 11).
// a / b for a divisor that is the same for a whole kernel (the camera's fx, fy), CORRECTLY ROUNDED -- the bits of the IEEE division the
// oracle and the reference perform -- in five instructions instead of the ~30 issue slots of v_div_scale / v_rcp_f64 / v_div_fmas / v_div_fixup:
// y = RN(1 / b) is computed once; q0 = RN(a y) is within two ulps; the residual a - q b is exact in an FMA; one correction makes q faithful,
// and by Markstein's theorem a second one from an exact residual rounds it correctly whenever y is the correctly rounded reciprocal and b's
// significand is not all ones (b is a float widened to double: 29 trailing zero bits).  No special cases are taken: a is finite, |a| is far
// from the subnormal range, and a is never -0 ((j - cx) z with z > 0; pixels with z == 0 are rejected before).  Checked against `/` on
// 9e8 operands of the path's own shape and random ones (tools/microbench/div_const_check.c; tests/test_oracle_known_answers.py runs it).
static __device__ __forceinline__ double hvo_div_const(double a, double b, double y)
{
    const double q0 = a * y;
    const double q1 = fma(fma(-q0, b, a), y, q0);
    return fma(fma(-q1, b, a), y, q1);
}

// cv::fastAtan2, float, evaluated without contraction (the same text as lsd.hip's and orb.hip's own copies)
static __device__ __forceinline__ float hvo_fatan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    const bool hi = ax >= ay;
    const float c = __fdiv_rn(hi ? ay : ax, __fadd_rn(hi ? ax : ay, eps)), c2 = __fmul_rn(c, c);
    float r = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    if (!hi) r = __fsub_rn(90.f, r);
    if (x < 0) r = __fsub_rn(180.f, r);
    if (y < 0) r = __fsub_rn(360.f, r);
    return r;
}

static inline int hvo_grid(long long items, int wg_per_cu) {
    const long long cap = 256LL * wg_per_cu;
    return (int)(items < cap ? (items < 1 ? 1 : items) : cap);
}

static inline hipStream_t hvo_stream_lsd(hvo_ctx *c);
static inline hipStream_t hvo_stream_peac(hvo_ctx *c);
// profiling scope helpers (api.hip)
int  hvo_prof_begin(hvo_ctx *ctx, const char *name, hipStream_t st);
void hvo_prof_end(hvo_ctx *ctx, int id);
void *hvo_stage_host(hvo_ctx *ctx, size_t bytes);
int hvo_staged_d2h(hvo_ctx *ctx, hipStream_t st, const void *dev_base, size_t dev_stride, int n, void *const *dst, const size_t *bytes, int widen8 = 0);

// orb.hip
int orb_init_tables(hvo_ctx *ctx);
int orb_ensure_plan(hvo_ctx *ctx, int w, int h, int batch);
bool orb_plan_covers(const hvo_ctx *ctx, int w, int h, int batch);      // the current plan serves (w, h, batch) without being rebuilt
void orb_free_plan(hvo_ctx *ctx);
// Bit-reversed order of the frames 0..n-1 (device array; nullptr when n < 2 or on failure = identity).  The kernels that give a
// frame one wave for its whole life (k_lsd_grow, k_peac_flood) take workgroup b's frame from it: the waves that share a SIMD are
// workgroups a fixed stride apart, and frames a fixed stride apart in a batch tend to be alike (the same camera, or a synthetic
// batch's period), so whole SIMDs got only long or only short frames and the kernel lasted as long as the unluckiest one.
const int *hvo_frame_perm(hvo_ctx *ctx, int n);
// Strided copies between pinned host memory and device slabs are executed by copy kernels, which queue like any other kernel: beside
// another context's compute on streams of equal priority a BatchPipeline's download took 80-140 ms instead of 30.  hvo_batch_upload /
// hvo_batch_download therefore move their copies to ctx->s_copy (highest priority); everything else keeps its compute stream's order.
static inline hipStream_t hvo_copy_stream(hvo_ctx *c, hipStream_t dflt) { return (c->copy_hi && c->s_copy) ? c->s_copy : dflt; }
int orb_upload(hvo_ctx *ctx, int n, const hvo_frame_in *in, int w, int h, bool sync = true);   // sync = false: the caller waits for ctx->stream
int orb_run(hvo_ctx *ctx, int n);
int orb_download(hvo_ctx *ctx, int n, hvo_frame_out *out);
// orb_level.hip
bool orb_level_build(OrbPlan &P, const std::vector<CellDesc> &cells, const std::vector<int> &xofs, const std::vector<int> &yofs, std::vector<OrbTile> &tiles);
int orb_level_run(hvo_ctx *ctx, int c0, int n, hipStream_t st, int k0, int k1, int k2, int k3);   // frames [c0, c0 + n) = one chunk
int orb_blur_float_run(hvo_ctx *ctx, int c0, int m, hipStream_t st);
// orb_describe.hip
int orb_describe_build(hvo_ctx *ctx);
int orb_describe_run(hvo_ctx *ctx, int c0, int n, hipStream_t st);

// match.hip
#define HVO_SBP_K 16
// device-resident arguments of the guided search (ORBmatcher::SearchByProjection cores)
struct SbpDev {
    const uint8_t *q_desc; const int *q_desc_index;            // query i's descriptor = q_desc + 32 * (q_desc_index ? q_desc_index[i] : i)
    const float *q_u, *q_v, *q_radius; const int *q_min_level, *q_max_level; const float *q_ur, *q_angle; const uint8_t *q_blocks;
    const hvo_keypoint *t_kp; const float *t_uright; const uint8_t *t_occ; const uint8_t *t_desc;
    int nq, nt; float mnMinX, mnMinY, mnMaxX, mnMaxY;
    int th_high, check_orientation, map_mode; float nn_ratio;
    unsigned long long *keys; int *cnt;                        // scratch (match_sbp_enqueue carves them)
    int32_t *match_idx, *match_dist; int *n_matches;          // results
};
// the projection prologue of SearchByProjection(Cur, Last) (match.hip k_project_last); sf = mvScaleFactors
struct ProjDev { float Rcw[9], tcw[3]; int fwd, bwd; float fx, fy, cx, cy, mbf, th; float sf[HVO_MAX_LEVELS]; float mnMinX, mnMinY, mnMaxX, mnMaxY; };
void match_project_setup(ProjDev &P, const float *Tcw, const float *Tlw, float mb, int mono);
int match_project_last_enqueue(hipStream_t st, const ProjDev &P, int n, const float *d_x3Dw, const int *d_qidx, const hvo_keypoint *d_last_kp,
                               float *q_u, float *q_v, float *q_radius, int *q_min, int *q_max, float *q_ur);
int match_search_by_projection_tracked(hvo_ctx *ctx, const uint8_t *q_desc, int nq, const float *proj_x, const float *proj_y, const float *proj_xr,
                                       const int32_t *level, const float *view_cos, const uint8_t *q_blocks, float th,
                                       const hvo_keypoint *t_kp, const float *t_uright, const uint8_t *t_occupied, const uint8_t *t_desc, int nt,
                                       float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, int th_high, float nn_ratio,
                                       int32_t *match_idx, int32_t *match_dist, int *n_matches);
int match_matrix(hvo_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, uint16_t *d);
int match_knn2(hvo_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx2, int32_t *dist2);
int match_knn2_enqueue(hipStream_t st, const uint8_t *dq, int nq, const uint8_t *dt, int nt, int32_t *d_idx2, int32_t *d_dist2);
void match_free(hvo_ctx *ctx);
size_t match_sbp_scratch_bytes(int nq);
int match_sbp_enqueue(hipStream_t st, SbpDev a, void *scratch);
int match_search_by_projection(hvo_ctx *ctx, const uint8_t *q_desc, int nq, const float *q_u, const float *q_v, const float *q_radius,
                               const int32_t *q_min_level, const int32_t *q_max_level, const float *q_ur, const float *q_angle, const uint8_t *q_blocks,
                               const hvo_keypoint *t_kp, const float *t_uright, const uint8_t *t_occupied, const uint8_t *t_desc, int nt,
                               float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, int th_high, int check_orientation, int map_mode, float nn_ratio,
                               int32_t *match_idx, int32_t *match_dist, int *n_matches);
size_t match_lines_scratch_bytes(int n1, int n2);
int match_lines_enqueue(hipStream_t st, const uint8_t *d1, int n1, const uint8_t *d2, int n2, float TH, float nnratio, int mode,
                        void *scratch, int32_t *d_m12, int *d_nmatch);
int match_lines(hvo_ctx *ctx, const uint8_t *d1, int n1, const uint8_t *d2, int n2, float th, float nnratio, int mode, int32_t *m12, int *n_matches);
// line_track.inc (part of match.hip): the line tracker's own calls -- LSDmatcher::SearchByGeomNApearance and SearchByProjection(Cur, Last, th)
// device-resident arguments of the guided line search; visit positions are 24 bits (three windows over < 2^22 grid items)
struct LsbpDev {
    int nq, nt;
    const float *q_xyxy; const hvo_keyline *q_kl; const int32_t *q_index;      // q_index != null: query q is line q_index[q] of q_kl / q_desc_all
    const uint8_t *q_desc; const uint8_t *q_desc_all; const uint8_t *q_blocks;
    const hvo_keyline *t_kl; const double *t_fn; const uint8_t *t_desc; const uint8_t *t_occ;
    const int32_t *cell_start, *cell_items; int n_items;
    float mnMinX, mnMaxX, mnMinY, mnMaxY, th; double cos_th;
    unsigned long long *keys; int32_t *match_idx, *match_dist; int *n_matches;
};

size_t match_lsbp_scratch_bytes(int nq, int nt);
int match_lsbp_enqueue(hipStream_t st, LsbpDev a, void *scratch);
int match_lines_geom_enqueue(hipStream_t st, const uint8_t *d_last, const hvo_keyline *kl_last, const uint8_t *has_ml, int n_last,
                             const uint8_t *d_cur, const hvo_keyline *kl_cur, int n_cur, float desc_th, const float *bounds4,
                             void *scratch, int32_t *d_m12, uint8_t *d_acc);
int match_lines_geom(hvo_ctx *ctx, const uint8_t *d_last, const hvo_keyline *kl_last, const uint8_t *has_ml, int n_last,
                     const uint8_t *d_cur, const hvo_keyline *kl_cur, int n_cur, float desc_th, const float *bounds4,
                     int32_t *m12, uint8_t *accepted, int *n_accepted);
int match_search_lines_by_projection(hvo_ctx *ctx, int nq, const float *q_xyxy, const hvo_keyline *q_kl, const uint8_t *q_desc, const uint8_t *q_blocks,
                                     const hvo_keyline *t_kl, const double *t_linefn, const uint8_t *t_desc, const uint8_t *t_occupied, int nt,
                                     const int32_t *cell_start, const int32_t *cell_items, int n_items, const float *bounds4, float th,
                                     int32_t *match_idx, int32_t *match_dist, int *n_matches);
int match_stereo_enqueue(hipStream_t st, const hvo_keypoint *d_kp, const hvo_keypoint *d_kpun, const int *d_n, int n_max, const uint16_t *d_depth, int pitch,
                         int w, int h, float dfac, float bf, float *d_uright, float *d_zdepth);
int match_stereo_from_rgbd(hvo_ctx *ctx, const hvo_keypoint *kp, const hvo_keypoint *kpun, int n, const uint16_t *depth, int w, int h, int stride,
                           float bf, float *uright, float *zdepth);

// readings.hip: the alternative readings of cv::GaussianBlur / cv::LineSegmentDetector (hvo_set_readings)
int readings_gblur_enqueue(hipStream_t st, const uint8_t *src, size_t sframe, int spitch, int w, int h, uint8_t *dst, size_t dframe, int dpitch,
                           int nframes, int ksize, double sigma, bool float_reading);
int readings_resize_tables(hipStream_t st, int sw, int sh, int dw, int dh, double factor, int *d_tab);
int readings_resize_enqueue(hipStream_t st, const uint8_t *src, size_t sframe, int spitch, int sw, uint8_t *dst, size_t dframe, int dpitch, int dw, int dh,
                            int nframes, const int *d_tab);
int readings_lsd_grad8_enqueue(hipStream_t st, const uint8_t *s8, size_t sframe, int sw, int sh, double4 *px4, unsigned *defined, int nwords, double rho, int nframes);

// frame.hip
int frame_undistort(hvo_ctx *ctx, const hvo_keypoint *kp, int n, const float *dist5, hvo_keypoint *kp_un);
int frame_image_bounds(hvo_ctx *ctx, int w, int h, const float *dist5, float *bounds4);
int frame_undistort_enqueue(hvo_ctx *ctx, hipStream_t st, const hvo_keypoint *d_kp, const int *d_n, int n_max, const float *dist5, hvo_keypoint *d_out);
void frame_gather_angles_enqueue(hipStream_t st, const hvo_keypoint *d_kp, const int *d_idx, int n, float *d_angle);
int frame_points_to_grid(hvo_ctx *ctx, const hvo_keypoint *kp_un, int n, const float *bounds4, int32_t *cell_start, int32_t *cell_items, int *n_out);
int frame_lines_to_grid(hvo_ctx *ctx, const hvo_keyline *kl, int n, const float *bounds4, int32_t *cell_start, int32_t *cell_items, int cap, int *n_out);

// api.hip: device buffer of at least `bytes` that lives as long as the context (grows by reallocation after draining the context's streams)
void *hvo_call_arena(hvo_ctx *ctx, size_t bytes);

// line3d.hip
int lines3d_enqueue(hvo_ctx *ctx, hipStream_t st, const hvo_keyline *d_kl, const int *d_n, int n_max, const uint16_t *d_depth, int pitch, int w, int h,
                    unsigned seed, hvo_line3d *d_out);

// vps.hip / planes_tail.hip / frame.hip: device-resident forms of the Frame tail (scratch from the caller, no allocation, no sync)
size_t vp_scratch_bytes(int nmax);
int vp_enqueue(hvo_ctx *ctx, hipStream_t st, const hvo_keyline *d_kl, const int *d_n, int nmax, unsigned seed, double th_angle,
               void *scratch, hvo_vp_result *d_res, int32_t *d_idx, double *d_grid_out);
size_t pc_scratch_bytes(int cap);
int pc_enqueue(hvo_ctx *ctx, hipStream_t st, const uint16_t *d_depth, int pitch, int w, int h, const int8_t *d_labels8, const hvo_plane *d_planes,
               const int *d_npl, int npl_fixed, double dist_th, void *scratch, float *d_cloud, int cap, hvo_plane_cloud *d_out, int *d_out_n);
size_t sn_scratch_bytes(int w, int h);
int sn_count(int w, int h);
int sn_enqueue(hvo_ctx *ctx, hipStream_t st, const uint16_t *d_depth, int pitch, int w, int h, void *scratch, hvo_surface_normal *d_out);
size_t frame_grid_scratch_ints(int n_max, bool lines);
int frame_points_grid_enqueue(hvo_ctx *ctx, hipStream_t st, const hvo_keypoint *d_kp_un, const int *d_n, int n_max, const float *b,
                              int *d_cell, int32_t *d_start, int32_t *d_items, int cap, int *d_total);
int frame_lines_grid_enqueue(hvo_ctx *ctx, hipStream_t st, const hvo_keyline *d_kl, const int *d_n, int n_max, const float *b,
                             int *d_cell, int32_t *d_start, int32_t *d_items, int cap, int *d_total);

// tail.hip: the rest of the Frame constructor on the resident results
#define HVO_TAIL_CLOUD_CAP 16384           // voxel-grid points of all planes of a frame (0.1 m leaves: a few thousand at most)
struct TailLayout {
    int w, h, kp_cap, nfeat, cloud_cap, n_normals, ln_cap;
    size_t counts, lines3d, vp_res, vp_idx, pclouds, cloud, normals, pt_start, pt_items, ln_start, ln_items, total;   // a frame's result block
    size_t s_vp, s_pc, s_sn, s_ptcell, s_lncell, scratch_total;                                                       // a frame's scratch
};
void tail_layout(int w, int h, int kp_cap, int nfeat, TailLayout &L);
int tail_enqueue_lines(hvo_ctx *ctx, hipStream_t st, unsigned stages, const TailLayout &L, char *d_out, char *d_scratch,
                       const hvo_keyline *d_kl, const int *d_nkl, const uint16_t *d_depth, int pitch, unsigned seed, double vp_th_angle, const float *bounds4);
int tail_enqueue_planes(hvo_ctx *ctx, hipStream_t st, unsigned stages, const TailLayout &L, char *d_out, char *d_scratch,
                        const uint16_t *d_depth, int pitch, const int8_t *d_labels8, const hvo_plane *d_planes, const int *d_npl, double dist_th);
int tail_enqueue_normals(hvo_ctx *ctx, hipStream_t st, unsigned stages, const TailLayout &L, char *d_out, char *d_scratch, const uint16_t *d_depth, int pitch);
int tail_enqueue_points(hvo_ctx *ctx, hipStream_t st, unsigned stages, const TailLayout &L, char *d_out, char *d_scratch,
                        const hvo_keypoint *d_kp_un, const int *d_nkp, const float *bounds4);
int tail_unpack(const TailLayout &L, unsigned stages, const char *ho, int n_kl, hvo_frame_tail *out);
int tail_batch_run(hvo_ctx *ctx, unsigned stages);
void tail_batch_free(hvo_ctx *ctx);

// peac.hip
struct PeacView { uint16_t *d_depth; int pitch; size_t dframe; int8_t *d_labels8; hvo_plane *d_planes; int *d_meta; int npix, max_planes; size_t lstride /* bytes between two frames' label images */; };
bool peac_plan_covers(const hvo_ctx *ctx, int w, int h, int batch);
int peac_prepare(hvo_ctx *ctx, int w, int h, int batch, PeacView *v);      // plan for this geometry + where its inputs / results live
int peac_upload(hvo_ctx *ctx, int n, const hvo_frame_in *in, int w, int h, bool sync = true);  // sync = false: the caller waits for ctx->s_peac
int peac_run(hvo_ctx *ctx, int n);
int peac_download(hvo_ctx *ctx, int n, hvo_frame_out *out);
void peac_free(hvo_ctx *ctx);

// lsd.hip (+ lbd)
struct LsdView { hvo_keyline *d_kl; uint8_t *d_desc; double *d_fn; int *d_nkl; int *d_flags; int nfeat; };
int lsd_prepare(hvo_ctx *ctx, int w, int h, int batch, bool culled, LsdView *v);
int lsd_run(hvo_ctx *ctx, int n, bool cull = false);
int lsd_download(hvo_ctx *ctx, int n, hvo_frame_out *out, bool culled = false);
void lsd_free(hvo_ctx *ctx);

static inline hipStream_t hvo_stream_lsd(hvo_ctx *c) { return (c->serialize || c->lsd_on_orb_stream) ? c->stream : c->s_lsd; }
static inline hipStream_t hvo_stream_peac(hvo_ctx *c) { return c->serialize ? c->stream : c->s_peac; }
