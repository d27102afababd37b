// stream.hip -- streamed-sequence mode (BASELINE.json config 5): one RGB-D frame at a time, a ring of `depth` frames
// in flight, the previous frames' results resident in HBM for the frame-to-frame matching.
//
// Reference call shape (src/Tracking.cc:262 -> Frame ctor src/Frame.cc:205-233, then TrackWithMotionModel):
//     Frame(imGray, imDepth, ...)           ExtractORB || ExtractLSD || ComputePlanes, UndistortKeyPoints, ComputeStereoFromRGBD
//     matcher.SearchByProjection(Cur, Last) src/Tracking.cc:2396 -> src/ORBmatcher.cc:1353-1497
//     lmatcher.SearchByGeomNApearance       src/Tracking.cc:2299 -> LSDmatcher::match(Last.mLdesc, Cur.mLdesc) src/LSDmatcher.cpp:42, 803-826
//
// A slot is a complete single-frame context (three HIP streams, its own plans) plus pinned input / output staging, so
// that `depth` consecutive frames overlap on the GPU: one frame's serial chains (AHC, region growing) occupy a handful
// of CUs, the next frames run beside them.  hvo_stream_submit copies the images into pinned memory, enqueues uploads,
// every kernel and the result downloads, and returns; hvo_stream_collect waits for that frame's events and hands the
// results out.  Nothing is allocated and no stream is drained on the submit path.  The matching calls take device-resident
// descriptors, undistorted key points and mvuRight of the two frames (nothing but the per-query projections crosses PCIe).
#include "hvo_internal.hpp"
#include <string.h>
#include <new>
#include <vector>

#define ST_MAX_DEPTH 16

struct StreamSlot {
    hvo_ctx *ctx = nullptr;
    PeacView pv; LsdView lv;
    // pinned host
    uint8_t *h_gray = nullptr; uint16_t *h_depth = nullptr;
    char *h_out = nullptr;
    // device extras
    hvo_keypoint *d_kp_un = nullptr; float *d_uright = nullptr, *d_zdepth = nullptr;
    char *d_tail = nullptr, *d_tail_scratch = nullptr, *h_tail = nullptr;     // the Frame tail's result block (HBM + pinned copy) and scratch (tail.hip)
    hipEvent_t ev_gray = nullptr, ev_depth = nullptr, ev_orb = nullptr, ev_lsd = nullptr, ev_peac = nullptr;
    hipEvent_t ev_kern[3] = { nullptr, nullptr, nullptr };      // kernels done (before the downloads), per subsystem: latency accounting
    hipEvent_t ev_t0 = nullptr;
    int64_t ticket = -1; bool busy = false, had_depth = false;
};

// layout of a slot's pinned result block
struct OutLayout {
    size_t counts, kp, desc, kp_un, uright, zdepth, kl, ldesc, fn, planes, labels, total;
};

struct hvo_stream {
    hvo_params p; hvo_stream_params sp;
    int depth = 0, w = 0, h = 0, kp_cap = 0, nfeat = 0;
    bool culled = false;
    StreamSlot slot[ST_MAX_DEPTH];
    OutLayout lay;
    int64_t next = 0;
    TailLayout tl; unsigned tail_stages = 0; double tail_dist_th = 0.05, tail_vp_th = 1.0 / 180.0 * 3.1415926535897932384626433832795;
    float bounds[4];                       // mnMinX, mnMaxX, mnMinY, mnMaxY (Frame::ComputeImageBounds)
    const float *bounds4() const { return bounds; }
    // matching scratch (device + pinned), sized for kp_cap queries
    char *d_ms = nullptr, *h_ms = nullptr; size_t ms_bytes = 0;
    hipStream_t s_match = nullptr;         // the matching calls run here, behind the two frames' events (not behind a frame's line chain)
    std::string last_error;
};

#define ST_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { s->last_error = std::string(#call) + ": " + hipGetErrorString(e_); return HVO_ERR_HIP; } } while (0)

static size_t al64(size_t v) { return (v + 63) & ~(size_t)63; }

extern "C" {

void hvo_stream_destroy(hvo_stream *s)
{
    if (!s) return;
    (void)hipSetDevice(s->p.device);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < ST_MAX_DEPTH; i++) {
        StreamSlot &S = s->slot[i];
        hipEvent_t evs[] = { S.ev_gray, S.ev_depth, S.ev_orb, S.ev_lsd, S.ev_peac, S.ev_kern[0], S.ev_kern[1], S.ev_kern[2], S.ev_t0 };
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
        if (S.h_gray) (void)hipHostFree(S.h_gray);
        if (S.h_depth) (void)hipHostFree(S.h_depth);
        if (S.h_out) (void)hipHostFree(S.h_out);
        if (S.d_kp_un) (void)hipFree(S.d_kp_un);
        if (S.d_uright) (void)hipFree(S.d_uright);
        if (S.d_zdepth) (void)hipFree(S.d_zdepth);
        if (S.d_tail) (void)hipFree(S.d_tail);
        if (S.d_tail_scratch) (void)hipFree(S.d_tail_scratch);
        if (S.h_tail) (void)hipHostFree(S.h_tail);
        if (S.ctx) hvo_destroy(S.ctx);
    }
    if (s->d_ms) (void)hipFree(s->d_ms);
    if (s->h_ms) (void)hipHostFree(s->h_ms);
    if (s->s_match) (void)hipStreamDestroy(s->s_match);
    delete s;
}

const char *hvo_stream_last_error(const hvo_stream *s) { return s ? s->last_error.c_str() : ""; }

int hvo_stream_create(const hvo_params *p, const hvo_stream_params *sp, hvo_stream **out)
{
    if (!p || !sp || !out) return HVO_ERR_INVALID_ARG;
    *out = nullptr;
    if (sp->depth < 2 || sp->depth > ST_MAX_DEPTH || sp->width <= 0 || sp->height <= 0) return HVO_ERR_INVALID_ARG;
    if (!(sp->stages & (HVO_STAGE_ORB | HVO_STAGE_LSD | HVO_STAGE_LSD_CULL | HVO_STAGE_PLANES))) return HVO_ERR_INVALID_ARG;
    hvo_stream *s = new (std::nothrow) hvo_stream();
    if (!s) return HVO_ERR_INVALID_ARG;
    s->p = *p; s->sp = *sp; s->depth = sp->depth; s->w = sp->width; s->h = sp->height;
    s->p.max_batch = 1;
    s->culled = (sp->stages & HVO_STAGE_LSD_CULL) != 0;
    s->tail_stages = sp->stages & (HVO_STAGE_LINES3D | HVO_STAGE_VP | HVO_STAGE_PLANE_TAIL | HVO_STAGE_GRIDS);
    if (sp->plane_dist_th > 0) s->tail_dist_th = sp->plane_dist_th;
    if (sp->vp_th_angle > 0) s->tail_vp_th = sp->vp_th_angle;
    {   // every tail stage needs the stage that produces its input
        const bool lsd = (sp->stages & (HVO_STAGE_LSD | HVO_STAGE_LSD_CULL)) != 0, orb = (sp->stages & HVO_STAGE_ORB) != 0, pl = (sp->stages & HVO_STAGE_PLANES) != 0;
        if (((s->tail_stages & (HVO_STAGE_LINES3D | HVO_STAGE_VP)) && !lsd) || ((s->tail_stages & HVO_STAGE_PLANE_TAIL) && !pl) || ((s->tail_stages & HVO_STAGE_GRIDS) && !(lsd && orb))) { delete s; return HVO_ERR_INVALID_ARG; }
    }
    const int w = s->w, h = s->h;
    int rc = HVO_OK;
    for (int i = 0; i < s->depth && !rc; i++) {
        StreamSlot &S = s->slot[i];
        if ((rc = hvo_create(&s->p, &S.ctx))) break;
        S.ctx->sched = 0; S.ctx->sched_cfg = 0;                                      // no cross-stream ordering inside a slot: the frames overlap instead
        // Three HIP streams per frame in flight (points, lines, planes; priorities 0 / -1 / +1), as in the batch mode.  The runtime
        // maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues per priority level and streams that share a queue
        // serialise: measured (profiles/r02_stream_scaling.txt) 32.8 / 64.6 / 64.9 / 93.9 frames/s at 1 / 2 / 3 / 5 frames in
        // flight.  Putting the line chain behind ORB on one stream (HVO_STREAM_LSD_OWN=0) serialises it with the plane chain on
        // the ROCm 7.2 runtime (20.8 frames/s at one frame in flight); GPU_MAX_HW_QUEUES=8 collapses to < 10 frames/s at five.
        S.ctx->lsd_on_orb_stream = getenv("HVO_STREAM_LSD_OWN") && atoi(getenv("HVO_STREAM_LSD_OWN")) == 0;
        if ((rc = orb_ensure_plan(S.ctx, w, h, 1))) break;
        if ((rc = lsd_prepare(S.ctx, w, h, 1, s->culled, &S.lv))) break;
        if ((rc = peac_prepare(S.ctx, w, h, 1, &S.pv))) break;
        if (i == 0) {
            s->kp_cap = S.ctx->orb.kp_cap; s->nfeat = S.lv.nfeat;
            OutLayout &L = s->lay; size_t o = 0;
            L.counts = o; o += al64(32 * sizeof(int));
            L.kp = o; o += al64((size_t)s->kp_cap * sizeof(hvo_keypoint));
            L.desc = o; o += al64((size_t)s->kp_cap * 32);
            L.kp_un = o; o += al64((size_t)s->kp_cap * sizeof(hvo_keypoint));
            L.uright = o; o += al64((size_t)s->kp_cap * 4);
            L.zdepth = o; o += al64((size_t)s->kp_cap * 4);
            L.kl = o; o += al64((size_t)s->nfeat * sizeof(hvo_keyline));
            L.ldesc = o; o += al64((size_t)s->nfeat * 32);
            L.fn = o; o += al64((size_t)s->nfeat * 24);
            L.planes = o; o += al64((size_t)S.pv.max_planes * sizeof(hvo_plane));
            L.labels = o; o += al64((size_t)w * h);
            L.total = o;
        }
        if (hipHostMalloc((void **)&S.h_gray, (size_t)w * h, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&S.h_depth, (size_t)w * h * 2, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&S.h_out, s->lay.total, hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void **)&S.d_kp_un, (size_t)s->kp_cap * sizeof(hvo_keypoint)) != hipSuccess ||
            hipMalloc((void **)&S.d_uright, (size_t)s->kp_cap * 4) != hipSuccess ||
            hipMalloc((void **)&S.d_zdepth, (size_t)s->kp_cap * 4) != hipSuccess) { rc = HVO_ERR_HIP; break; }
        if (s->tail_stages) {
            if (i == 0) tail_layout(w, h, S.ctx->orb.kp_cap, S.lv.nfeat > 0 ? S.lv.nfeat : s->p.lsd_nfeatures, s->tl);
            if (hipMalloc((void **)&S.d_tail, s->tl.total) != hipSuccess || hipMalloc((void **)&S.d_tail_scratch, 3 * s->tl.scratch_total) != hipSuccess ||
                hipHostMalloc((void **)&S.h_tail, s->tl.total, hipHostMallocDefault) != hipSuccess) { rc = HVO_ERR_HIP; break; }
            (void)hipMemset(S.d_tail, 0, s->tl.total); memset(S.h_tail, 0, s->tl.total);
        }
        hipEvent_t *evs[] = { &S.ev_gray, &S.ev_depth, &S.ev_orb, &S.ev_lsd, &S.ev_peac };
        for (hipEvent_t *e : evs) if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) rc = HVO_ERR_HIP;
        for (int k = 0; k < 3; k++) if (hipEventCreate(&S.ev_kern[k]) != hipSuccess) rc = HVO_ERR_HIP;
        if (hipEventCreate(&S.ev_t0) != hipSuccess) rc = HVO_ERR_HIP;
        memset(S.h_out, 0, s->lay.total);
    }
    if (!rc) {
        // Frame::ComputeImageBounds with the stream's distortion (src/Frame.cc:1733-1762)
        float b4[4];
        rc = frame_image_bounds(s->slot[0].ctx, w, h, sp->dist5, b4);
        for (int k = 0; k < 4; k++) s->bounds[k] = b4[k];
    }
    if (!rc) {
        const int nq = s->kp_cap;
        s->ms_bytes = al64((size_t)nq * 32) + 13 * al64((size_t)nq * 4) + 2 * al64((size_t)nq) + al64((size_t)s->kp_cap) + 3 * al64((size_t)nq * 4 + 64) +
                      match_sbp_scratch_bytes(nq) + match_lines_scratch_bytes(s->nfeat, s->nfeat) + al64((size_t)s->nfeat * 4 + 64) + 4096 +
                      match_lsbp_scratch_bytes(s->nfeat, s->nfeat) + 8 * al64((size_t)s->nfeat * 32);      // (the guided line search: a key per (query, line))
        if (hipMalloc((void **)&s->d_ms, s->ms_bytes) != hipSuccess || hipHostMalloc((void **)&s->h_ms, s->ms_bytes, hipHostMallocDefault) != hipSuccess ||
            hipStreamCreateWithPriority(&s->s_match, hipStreamNonBlocking, -1) != hipSuccess) rc = HVO_ERR_HIP;
    }
    if (rc) { hvo_stream_destroy(s); return rc; }
    *out = s;
    return HVO_OK;
}

int hvo_stream_image_bounds(const hvo_stream *s, float bounds4[4])
{
    if (!s || !bounds4) return HVO_ERR_INVALID_ARG;
    for (int k = 0; k < 4; k++) bounds4[k] = s->bounds[k];
    return HVO_OK;
}

int hvo_stream_capacity(const hvo_stream *s, int *kp_cap, int *kl_cap, int *pl_cap)
{
    if (!s) return HVO_ERR_INVALID_ARG;
    if (kp_cap) *kp_cap = s->kp_cap;
    if (kl_cap) *kl_cap = s->nfeat;
    if (pl_cap) *pl_cap = s->slot[0].pv.max_planes;
    return HVO_OK;
}

static int stream_submit_enqueue(hvo_stream *s, StreamSlot &S, const uint8_t *gray, int gray_stride, const uint16_t *depth, int depth_stride);

int hvo_stream_submit(hvo_stream *s, const uint8_t *gray, int gray_stride, const uint16_t *depth, int depth_stride, int64_t *ticket)
{
    if (!s || !gray || gray_stride < s->w || !ticket) return HVO_ERR_INVALID_ARG;
    const unsigned stages = s->sp.stages;
    if ((stages & HVO_STAGE_PLANES) && !depth) return HVO_ERR_INVALID_ARG;
    if (depth && depth_stride < 2 * s->w) return HVO_ERR_INVALID_ARG;
    StreamSlot &S = s->slot[s->next % s->depth];
    if (S.busy) return HVO_ERR_BUSY;                           // the frame that used this slot has not been collected
    if (hipSetDevice(s->p.device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    const int rc = stream_submit_enqueue(s, S, gray, gray_stride, depth, depth_stride);
    if (rc) {
        // Part of the frame may be in flight (uploads reading the pinned staging, kernels, only some events recorded): drain the slot's
        // three streams before handing it back, so that a retried submit does not overwrite staging an async copy is still reading and
        // no later poll / collect waits on events of a frame that was never completely enqueued.
        hvo_ctx *c = S.ctx;
        (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->s_lsd); (void)hipStreamSynchronize(c->s_peac);
        (void)hipGetLastError();
        S.busy = false; S.ticket = -1;
        return rc;
    }
    S.ticket = s->next; S.busy = true;
    *ticket = s->next++;
    return HVO_OK;
}

static int stream_submit_enqueue(hvo_stream *s, StreamSlot &S, const uint8_t *gray, int gray_stride, const uint16_t *depth, int depth_stride)
{
    const unsigned stages = s->sp.stages;
    hvo_ctx *c = S.ctx;
    const int w = s->w, h = s->h;
    const bool want_orb = (stages & HVO_STAGE_ORB) != 0, want_lsd = (stages & (HVO_STAGE_LSD | HVO_STAGE_LSD_CULL)) != 0, want_pl = (stages & HVO_STAGE_PLANES) != 0;
    // 1. images -> pinned staging -> HBM (gray on the ORB stream, depth on the plane stream)
    for (int y = 0; y < h; y++) memcpy(S.h_gray + (size_t)y * w, gray + (size_t)y * gray_stride, (size_t)w);
    if (depth) for (int y = 0; y < h; y++) memcpy(S.h_depth + (size_t)y * w, (const char *)depth + (size_t)y * depth_stride, (size_t)w * 2);
    OrbPlan &O = c->orb;
    ST_HIP(hipEventRecord(S.ev_t0, c->stream));
    ST_HIP(hipMemcpy2DAsync(O.d_pyr, O.lev[0].pitch, S.h_gray, w, w, h, hipMemcpyHostToDevice, c->stream));
    ST_HIP(hipEventRecord(S.ev_gray, c->stream));
    if (depth) {
        ST_HIP(hipMemcpy2DAsync(S.pv.d_depth, S.pv.pitch * sizeof(uint16_t), S.h_depth, (size_t)w * 2, (size_t)w * 2, h, hipMemcpyHostToDevice, c->s_peac));
        ST_HIP(hipEventRecord(S.ev_depth, c->s_peac));
    }
    S.had_depth = depth != nullptr;
    char *ho = S.h_out;
    const OutLayout &L = s->lay;
    int *hc = (int *)(ho + L.counts);
    int rc;
    // 2. planes (the longest chain first), lines, points: three streams, no host synchronisation
    if (want_pl) {
        if ((rc = peac_run(c, 1))) { s->last_error = c->last_error; return rc; }
        // ComputePlanes' tail on the resident depth, labels and planes (src/Frame.cc:2110-2274)
        if ((rc = tail_enqueue_planes(c, c->s_peac, s->tail_stages, s->tl, S.d_tail, S.d_tail_scratch + s->tl.scratch_total, S.pv.d_depth, S.pv.pitch, S.pv.d_labels8, S.pv.d_planes,
                                      S.pv.d_meta + 4, s->tail_dist_th))) return rc;
        ST_HIP(hipEventRecord(S.ev_kern[2], c->s_peac));
        if (s->tail_stages & HVO_STAGE_PLANE_TAIL) {
            const TailLayout &T = s->tl;
            ST_HIP(hipMemcpyAsync(S.h_tail + T.counts, S.d_tail + T.counts, 2 * sizeof(int), hipMemcpyDeviceToHost, c->s_peac));
            ST_HIP(hipMemcpyAsync(S.h_tail + T.pclouds, S.d_tail + T.pclouds, T.normals - T.pclouds, hipMemcpyDeviceToHost, c->s_peac));
        }
        ST_HIP(hipMemcpyAsync(hc + 16, S.pv.d_meta, 16 * sizeof(int), hipMemcpyDeviceToHost, c->s_peac));
        ST_HIP(hipMemcpyAsync(ho + L.planes, S.pv.d_planes, (size_t)S.pv.max_planes * sizeof(hvo_plane), hipMemcpyDeviceToHost, c->s_peac));
        ST_HIP(hipMemcpyAsync(ho + L.labels, S.pv.d_labels8, (size_t)w * h, hipMemcpyDeviceToHost, c->s_peac));
    }
    ST_HIP(hipEventRecord(S.ev_peac, c->s_peac));
    if (want_orb) {
        if ((rc = orb_run(c, 1))) { s->last_error = c->last_error; return rc; }
        // Frame::UndistortKeyPoints (src/Frame.cc:1701-1731) and ComputeStereoFromRGBD (1940-1961) on the resident key points
        if ((rc = frame_undistort_enqueue(c, c->stream, O.d_kp, O.d_nkp, s->kp_cap, s->sp.dist5, S.d_kp_un))) return rc;
        if (depth && s->sp.bf > 0) {
            ST_HIP(hipStreamWaitEvent(c->stream, S.ev_depth, 0));
            if ((rc = match_stereo_enqueue(c->stream, O.d_kp, S.d_kp_un, O.d_nkp, s->kp_cap, S.pv.d_depth, S.pv.pitch, w, h, s->p.depth_map_factor, s->sp.bf,
                                           S.d_uright, S.d_zdepth))) return rc;
        }
        if ((rc = tail_enqueue_points(c, c->stream, s->tail_stages, s->tl, S.d_tail, S.d_tail_scratch + 2 * s->tl.scratch_total, S.d_kp_un, O.d_nkp, s->bounds4()))) return rc;
        ST_HIP(hipEventRecord(S.ev_kern[0], c->stream));
        if (s->tail_stages & HVO_STAGE_GRIDS) {
            const TailLayout &T = s->tl;
            ST_HIP(hipMemcpyAsync(S.h_tail + T.counts + 2 * sizeof(int), S.d_tail + T.counts + 2 * sizeof(int), sizeof(int), hipMemcpyDeviceToHost, c->stream));
            ST_HIP(hipMemcpyAsync(S.h_tail + T.pt_start, S.d_tail + T.pt_start, T.ln_start - T.pt_start, hipMemcpyDeviceToHost, c->stream));
        }
        ST_HIP(hipMemcpyAsync(hc + 0, O.d_nkp, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        ST_HIP(hipMemcpyAsync(hc + 1, O.d_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        ST_HIP(hipMemcpyAsync(ho + L.kp, O.d_kp, (size_t)s->kp_cap * sizeof(hvo_keypoint), hipMemcpyDeviceToHost, c->stream));
        ST_HIP(hipMemcpyAsync(ho + L.desc, O.d_desc, (size_t)s->kp_cap * 32, hipMemcpyDeviceToHost, c->stream));
        ST_HIP(hipMemcpyAsync(ho + L.kp_un, S.d_kp_un, (size_t)s->kp_cap * sizeof(hvo_keypoint), hipMemcpyDeviceToHost, c->stream));
        if (depth && s->sp.bf > 0) {
            ST_HIP(hipMemcpyAsync(ho + L.uright, S.d_uright, (size_t)s->kp_cap * 4, hipMemcpyDeviceToHost, c->stream));
            ST_HIP(hipMemcpyAsync(ho + L.zdepth, S.d_zdepth, (size_t)s->kp_cap * 4, hipMemcpyDeviceToHost, c->stream));
        }
    }
    // surface normals (ComputePlanes' tail, src/Frame.cc:2160-2212): they need the depth image only, so they run on this short stream
    // beside the plane chain instead of behind its 20 ms
    if (depth && want_pl && (s->tail_stages & HVO_STAGE_PLANE_TAIL)) {
        const TailLayout &T = s->tl;
        ST_HIP(hipStreamWaitEvent(c->stream, S.ev_depth, 0));
        if ((rc = tail_enqueue_normals(c, c->stream, s->tail_stages, T, S.d_tail, S.d_tail_scratch + 2 * T.scratch_total, S.pv.d_depth, S.pv.pitch))) return rc;
        ST_HIP(hipMemcpyAsync(S.h_tail + T.normals, S.d_tail + T.normals, (size_t)T.n_normals * sizeof(hvo_surface_normal), hipMemcpyDeviceToHost, c->stream));
    }
    ST_HIP(hipEventRecord(S.ev_orb, c->stream));
    hipStream_t LS = hvo_stream_lsd(c);
    if (want_lsd) {
        if (!c->lsd_on_orb_stream) ST_HIP(hipStreamWaitEvent(c->s_lsd, S.ev_gray, 0));
        if ((rc = lsd_run(c, 1, s->culled))) { s->last_error = c->last_error; return rc; }
        // isLineGood, vanishing points and the line grid on the resident key lines (src/Frame.cc:328-337, 934-939, 849-872)
        if (s->tail_stages & (HVO_STAGE_LINES3D | HVO_STAGE_VP | HVO_STAGE_GRIDS)) {
            const bool l3 = (s->tail_stages & HVO_STAGE_LINES3D) && depth;
            if (l3) ST_HIP(hipStreamWaitEvent(LS, S.ev_depth, 0));
            if ((rc = tail_enqueue_lines(c, LS, s->tail_stages, s->tl, S.d_tail, S.d_tail_scratch, S.lv.d_kl, S.lv.d_nkl, l3 ? S.pv.d_depth : nullptr, S.pv.pitch,
                                         s->sp.seed + (unsigned)s->next, s->tail_vp_th, s->bounds4()))) return rc;
        }
        ST_HIP(hipEventRecord(S.ev_kern[1], LS));
        if (s->tail_stages & (HVO_STAGE_LINES3D | HVO_STAGE_VP | HVO_STAGE_GRIDS)) {
            const TailLayout &T = s->tl;
            ST_HIP(hipMemcpyAsync(S.h_tail + T.lines3d, S.d_tail + T.lines3d, T.pclouds - T.lines3d, hipMemcpyDeviceToHost, LS));
            if (s->tail_stages & HVO_STAGE_GRIDS) {
                ST_HIP(hipMemcpyAsync(S.h_tail + T.counts + 3 * sizeof(int), S.d_tail + T.counts + 3 * sizeof(int), sizeof(int), hipMemcpyDeviceToHost, LS));
                ST_HIP(hipMemcpyAsync(S.h_tail + T.ln_start, S.d_tail + T.ln_start, T.total - T.ln_start, hipMemcpyDeviceToHost, LS));
            }
        }
        ST_HIP(hipMemcpyAsync(hc + 4, S.lv.d_nkl, sizeof(int), hipMemcpyDeviceToHost, LS));
        ST_HIP(hipMemcpyAsync(hc + 5, S.lv.d_flags, sizeof(int), hipMemcpyDeviceToHost, LS));
        ST_HIP(hipMemcpyAsync(ho + L.kl, S.lv.d_kl, (size_t)s->nfeat * sizeof(hvo_keyline), hipMemcpyDeviceToHost, LS));
        ST_HIP(hipMemcpyAsync(ho + L.ldesc, S.lv.d_desc, (size_t)s->nfeat * 32, hipMemcpyDeviceToHost, LS));
        ST_HIP(hipMemcpyAsync(ho + L.fn, S.lv.d_fn, (size_t)s->nfeat * 24, hipMemcpyDeviceToHost, LS));
    }
    ST_HIP(hipEventRecord(S.ev_lsd, LS));
    return HVO_OK;
}

static StreamSlot *slot_of(hvo_stream *s, int64_t ticket)
{
    if (ticket < 0 || ticket >= s->next || ticket < s->next - s->depth) return nullptr;
    StreamSlot &S = s->slot[ticket % s->depth];
    return S.ticket == ticket ? &S : nullptr;
}

// 0: not finished, 1: every stage of that frame (and its downloads) is complete
int hvo_stream_poll(hvo_stream *s, int64_t ticket)
{
    StreamSlot *S = s ? slot_of(s, ticket) : nullptr;
    if (!S) return HVO_ERR_INVALID_ARG;
    hipEvent_t evs[] = { S->ev_orb, S->ev_lsd, S->ev_peac };
    for (hipEvent_t e : evs) { const hipError_t r = hipEventQuery(e); if (r == hipErrorNotReady) return 0; if (r != hipSuccess) return HVO_ERR_HIP; }
    return 1;
}

int hvo_stream_collect(hvo_stream *s, int64_t ticket, hvo_frame_out *out, hvo_keypoint *kp_un, float *uright, float *zdepth)
{
    if (!s) return HVO_ERR_INVALID_ARG;
    StreamSlot *Sp = slot_of(s, ticket);
    if (!Sp || !Sp->busy) return HVO_ERR_INVALID_ARG;
    StreamSlot &S = *Sp;
    ST_HIP(hipEventSynchronize(S.ev_orb));
    ST_HIP(hipEventSynchronize(S.ev_lsd));
    ST_HIP(hipEventSynchronize(S.ev_peac));
    S.busy = false;
    const unsigned stages = s->sp.stages;
    const OutLayout &L = s->lay;
    const char *ho = S.h_out;
    const int *hc = (const int *)(ho + L.counts);
    if (out) {
        out->status = HVO_OK; out->n_kp = out->n_kl = out->n_planes = 0;
        if (stages & HVO_STAGE_ORB) {
            int m = hc[0];
            if (hc[1]) out->status = HVO_ERR_CAPACITY;
            if (out->kp) {
                if (m > out->kp_cap) { m = out->kp_cap; out->status = HVO_ERR_CAPACITY; }
                memcpy(out->kp, ho + L.kp, (size_t)m * sizeof(hvo_keypoint));
                if (out->desc) memcpy(out->desc, ho + L.desc, (size_t)m * 32);
                if (kp_un) memcpy(kp_un, ho + L.kp_un, (size_t)m * sizeof(hvo_keypoint));
                if (uright && S.had_depth && s->sp.bf > 0) memcpy(uright, ho + L.uright, (size_t)m * 4);
                if (zdepth && S.had_depth && s->sp.bf > 0) memcpy(zdepth, ho + L.zdepth, (size_t)m * 4);
            }
            out->n_kp = m;
        }
        if (stages & (HVO_STAGE_LSD | HVO_STAGE_LSD_CULL)) {
            int m = hc[4];
            if (hc[5]) out->status = HVO_ERR_CAPACITY;
            if (out->kl) {
                if (m > out->kl_cap) { m = out->kl_cap; out->status = HVO_ERR_CAPACITY; }
                memcpy(out->kl, ho + L.kl, (size_t)m * sizeof(hvo_keyline));
                if (out->ldesc) memcpy(out->ldesc, ho + L.ldesc, (size_t)m * 32);
                if (out->linefn) memcpy(out->linefn, ho + L.fn, (size_t)m * 24);
            }
            out->n_kl = m;
        }
        if (stages & HVO_STAGE_PLANES) {
            const int *meta = hc + 16;
            int m = meta[4];
            if (meta[3]) out->status = HVO_ERR_CAPACITY;
            if (out->labels) {                                  // int8 on the wire -> int32 (PlaneFitter::membershipImg is CV_32SC1)
                const int8_t *l8 = (const int8_t *)(ho + L.labels);
                const size_t npix = (size_t)s->w * s->h;
                for (size_t k = 0; k < npix; k++) out->labels[k] = (int32_t)l8[k];
            }
            if (out->labels8) memcpy(out->labels8, ho + L.labels, (size_t)s->w * s->h);
            if (out->planes) {
                if (m > out->pl_cap) { m = out->pl_cap; out->status = HVO_ERR_CAPACITY; }
                memcpy(out->planes, ho + L.planes, (size_t)m * sizeof(hvo_plane));
            }
            out->n_planes = m;
        }
    }
    return HVO_OK;
}

// the frame's tail results (waits for the frame like hvo_stream_collect; the slot stays busy until hvo_stream_collect releases it)
int hvo_stream_collect_tail(hvo_stream *s, int64_t ticket, hvo_frame_tail *tail)
{
    if (!s || !tail) return HVO_ERR_INVALID_ARG;
    StreamSlot *Sp = slot_of(s, ticket);
    if (!Sp || !Sp->busy || !s->tail_stages) return HVO_ERR_INVALID_ARG;
    ST_HIP(hipEventSynchronize(Sp->ev_orb));
    ST_HIP(hipEventSynchronize(Sp->ev_lsd));
    ST_HIP(hipEventSynchronize(Sp->ev_peac));
    unsigned st = s->tail_stages;
    if (!Sp->had_depth) st &= ~(HVO_STAGE_LINES3D | HVO_STAGE_PLANE_TAIL);
    const int n_kl = ((const int *)(Sp->h_out + s->lay.counts))[4];
    return tail_unpack(s->tl, st, Sp->h_tail, n_kl, tail);
}

// device time of the frame's stages: ms from the start of its upload to the end of the ORB / line / plane kernels
// (0 for a stage that did not run); valid after hvo_stream_collect or a positive hvo_stream_poll
int hvo_stream_stage_ms(hvo_stream *s, int64_t ticket, float ms3[3])
{
    StreamSlot *S = s ? slot_of(s, ticket) : nullptr;
    if (!S || !ms3) return HVO_ERR_INVALID_ARG;
    const unsigned stages = s->sp.stages;
    const bool on[3] = { (stages & HVO_STAGE_ORB) != 0, (stages & (HVO_STAGE_LSD | HVO_STAGE_LSD_CULL)) != 0, (stages & HVO_STAGE_PLANES) != 0 };
    for (int k = 0; k < 3; k++) {
        ms3[k] = 0.f;
        if (on[k] && hipEventElapsedTime(&ms3[k], S->ev_t0, S->ev_kern[k]) != hipSuccess) ms3[k] = -1.f;
    }
    return HVO_OK;
}

// ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, mono) core (src/ORBmatcher.cc:1353-1497) between two resident
// frames.  One query per last-frame map point that passed the projection tests: its last-frame feature index q_index[i]
// (descriptor and key-point angle are read from the last frame's slot unless q_desc is given: pMP->GetDescriptor() may differ
// from the last frame's own descriptor), the projected (u, v), radius, octave band, ur and "has observations" flag computed by
// the tracker.  The current frame's undistorted key points, mvuRight and descriptors never leave HBM.
int hvo_stream_search_by_projection(hvo_stream *s, int64_t cur, int64_t last, int nq, const int32_t *q_index, const uint8_t *q_desc,
                                    const float *q_u, const float *q_v, const float *q_radius, const int32_t *q_min_level, const int32_t *q_max_level,
                                    const float *q_ur, const uint8_t *q_blocks, const uint8_t *t_occupied, int th_high, int check_orientation,
                                    int32_t *match_idx, int32_t *match_dist, int *n_matches)
{
    if (!s || !n_matches || nq < 0) return HVO_ERR_INVALID_ARG;
    *n_matches = 0;
    if (nq == 0) return HVO_OK;
    if (nq > s->kp_cap || !q_index || !q_u || !q_v || !q_radius || !q_min_level || !q_max_level || !q_blocks || !match_idx || !match_dist) return HVO_ERR_INVALID_ARG;
    if (!(s->sp.stages & HVO_STAGE_ORB)) return HVO_ERR_INVALID_ARG;
    StreamSlot *C = slot_of(s, cur), *Lz = slot_of(s, last);
    if (!C || !Lz || C == Lz) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(s->p.device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    hipStream_t st = s->s_match;
    ST_HIP(hipStreamWaitEvent(st, Lz->ev_orb, 0));
    ST_HIP(hipStreamWaitEvent(st, C->ev_orb, 0));
    // the current frame's key-point count is needed on the host for the launch geometry: it arrived with the frame's download
    ST_HIP(hipEventSynchronize(C->ev_orb));
    const int nt = ((const int *)(C->h_out + s->lay.counts))[0];
    // q_index addresses the last frame's key points and descriptors on the device: an index outside [0, n_last) would be an
    // out-of-bounds gather there (a GPU fault aborts the process), so it is refused here
    ST_HIP(hipEventSynchronize(Lz->ev_orb));
    const int n_last = ((const int *)(Lz->h_out + s->lay.counts))[0];
    for (int i = 0; i < nq; i++) if (q_index[i] < 0 || q_index[i] >= n_last) return HVO_ERR_INVALID_ARG;
    for (int i = 0; i < nq; i++) { match_idx[i] = -1; match_dist[i] = 256; }
    if (nt <= 0) return HVO_OK;
    char *d = s->d_ms, *hh = s->h_ms; size_t off = 0;
    auto up = [&](const void *src, size_t bytes) -> void * {
        if (!src) return nullptr;
        void *dp = d + off; memcpy(hh + off, src, bytes);
        (void)hipMemcpyAsync(dp, hh + off, bytes, hipMemcpyHostToDevice, st);
        off += al64(bytes);
        return dp;
    };
    SbpDev a; memset(&a, 0, sizeof(a));
    if (q_desc) { a.q_desc = (const uint8_t *)up(q_desc, (size_t)nq * 32); a.q_desc_index = nullptr; }
    else { a.q_desc = Lz->ctx->orb.d_desc; a.q_desc_index = (const int *)up(q_index, (size_t)nq * 4); }
    a.q_u = (const float *)up(q_u, (size_t)nq * 4); a.q_v = (const float *)up(q_v, (size_t)nq * 4); a.q_radius = (const float *)up(q_radius, (size_t)nq * 4);
    a.q_min_level = (const int *)up(q_min_level, (size_t)nq * 4); a.q_max_level = (const int *)up(q_max_level, (size_t)nq * 4);
    a.q_ur = (const float *)up(q_ur, (size_t)nq * 4); a.q_blocks = (const uint8_t *)up(q_blocks, (size_t)nq);
    a.t_occ = (const uint8_t *)up(t_occupied, (size_t)nt);
    // key-point angles of the queries: LastFrame.mvKeysUn[i].angle, gathered on the device
    float *d_angle = (float *)(d + off); off += al64((size_t)nq * 4);
    const int *d_qidx = a.q_desc_index ? a.q_desc_index : (const int *)up(q_index, (size_t)nq * 4);
    frame_gather_angles_enqueue(st, Lz->d_kp_un, d_qidx, nq, d_angle);
    a.q_angle = d_angle;
    a.t_kp = C->d_kp_un; a.t_uright = (C->had_depth && s->sp.bf > 0) ? C->d_uright : nullptr; a.t_desc = C->ctx->orb.d_desc;
    a.nq = nq; a.nt = nt; a.mnMinX = s->bounds[0]; a.mnMaxX = s->bounds[1]; a.mnMinY = s->bounds[2]; a.mnMaxY = s->bounds[3];
    a.th_high = th_high; a.check_orientation = check_orientation; a.map_mode = 0; a.nn_ratio = 0.f;
    int32_t *dout = (int32_t *)(d + off); int32_t *hout = (int32_t *)(hh + off); off += al64((2 * (size_t)nq + 1) * 4);
    a.match_idx = dout; a.match_dist = dout + nq; a.n_matches = dout + 2 * nq;
    void *scratch = d + off; off += match_sbp_scratch_bytes(nq);
    if (off > s->ms_bytes) { s->last_error = "matching scratch too small"; return HVO_ERR_CAPACITY; }
    int rc = match_sbp_enqueue(st, a, scratch);
    if (rc) return rc;
    ST_HIP(hipMemcpyAsync(hout, dout, (2 * (size_t)nq + 1) * 4, hipMemcpyDeviceToHost, st));
    ST_HIP(hipStreamSynchronize(st));
    memcpy(match_idx, hout, (size_t)nq * 4); memcpy(match_dist, hout + nq, (size_t)nq * 4);
    *n_matches = hout[2 * nq];
    return HVO_OK;
}

int hvo_stream_project_last(hvo_stream *s, int64_t cur, int64_t last, const hvo_camera *cam, const float Tcw[12], const float Tlw[12],
                            int nq, const int32_t *q_index, const float *x3Dw, const uint8_t *q_blocks, const uint8_t *q_desc,
                            const uint8_t *t_occupied, float th, int mono, int th_high, int check_orientation,
                            int32_t *match_idx, int32_t *match_dist, int *n_matches, float *q_uv)
{
    if (!s || !n_matches || nq < 0) return HVO_ERR_INVALID_ARG;
    *n_matches = 0;
    if (nq == 0) return HVO_OK;
    if (nq > s->kp_cap || !cam || !Tcw || !Tlw || !q_index || !x3Dw || !q_blocks || !match_idx || !match_dist) return HVO_ERR_INVALID_ARG;
    if (!(s->sp.stages & HVO_STAGE_ORB)) return HVO_ERR_INVALID_ARG;
    StreamSlot *C = slot_of(s, cur), *Lz = slot_of(s, last);
    if (!C || !Lz || C == Lz) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(s->p.device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    hipStream_t st = s->s_match;
    ST_HIP(hipStreamWaitEvent(st, Lz->ev_orb, 0));
    ST_HIP(hipStreamWaitEvent(st, C->ev_orb, 0));
    ST_HIP(hipEventSynchronize(C->ev_orb));
    const int nt = ((const int *)(C->h_out + s->lay.counts))[0];
    ST_HIP(hipEventSynchronize(Lz->ev_orb));
    const int n_last = ((const int *)(Lz->h_out + s->lay.counts))[0];
    for (int i = 0; i < nq; i++) if (q_index[i] < 0 || q_index[i] >= n_last) return HVO_ERR_INVALID_ARG;     // (a gather out of bounds on the device otherwise)
    for (int i = 0; i < nq; i++) { match_idx[i] = -1; match_dist[i] = 256; }
    if (q_uv) for (int i = 0; i < 2 * nq; i++) q_uv[i] = 1e30f;
    if (nt <= 0) return HVO_OK;
    char *d = s->d_ms, *hh = s->h_ms; size_t off = 0;
    auto up = [&](const void *src, size_t bytes) -> void * {
        if (!src) return nullptr;
        void *dp = d + off; memcpy(hh + off, src, bytes);
        (void)hipMemcpyAsync(dp, hh + off, bytes, hipMemcpyHostToDevice, st);
        off += al64(bytes);
        return dp;
    };
    auto dev = [&](size_t bytes) -> void * { void *dp = d + off; off += al64(bytes); return dp; };
    SbpDev a; memset(&a, 0, sizeof(a));
    const int *d_qidx = (const int *)up(q_index, (size_t)nq * 4);
    if (q_desc) { a.q_desc = (const uint8_t *)up(q_desc, (size_t)nq * 32); a.q_desc_index = nullptr; }
    else { a.q_desc = Lz->ctx->orb.d_desc; a.q_desc_index = d_qidx; }
    const float *d_x = (const float *)up(x3Dw, (size_t)nq * 12);
    a.q_blocks = (const uint8_t *)up(q_blocks, (size_t)nq);
    a.t_occ = (const uint8_t *)up(t_occupied, (size_t)nt);
    float *d_u = (float *)dev((size_t)nq * 8), *d_v = d_u + nq;                              // u then v: one copy back for q_uv
    float *d_radius = (float *)dev((size_t)nq * 4), *d_ur = (float *)dev((size_t)nq * 4);
    int *d_min = (int *)dev((size_t)nq * 4), *d_max = (int *)dev((size_t)nq * 4);
    float *d_angle = (float *)dev((size_t)nq * 4);
    ProjDev P; memset(&P, 0, sizeof(P));
    match_project_setup(P, Tcw, Tlw, cam->b, mono);
    P.fx = cam->fx; P.fy = cam->fy; P.cx = cam->cx; P.cy = cam->cy; P.mbf = cam->bf; P.th = th;
    for (int l = 0; l < HVO_MAX_LEVELS; l++) P.sf[l] = C->ctx->scale[l];
    P.mnMinX = s->bounds[0]; P.mnMaxX = s->bounds[1]; P.mnMinY = s->bounds[2]; P.mnMaxY = s->bounds[3];
    int32_t *dout = (int32_t *)dev((2 * (size_t)nq + 1) * 4); int32_t *hout = (int32_t *)(hh + ((char *)dout - d));
    float *huv = (float *)(hh + ((char *)d_u - d));
    void *scratch = d + off; off += match_sbp_scratch_bytes(nq);
    if (off > s->ms_bytes) { s->last_error = "matching scratch too small"; return HVO_ERR_CAPACITY; }
    int rc = match_project_last_enqueue(st, P, nq, d_x, d_qidx, Lz->d_kp_un, d_u, d_v, d_radius, d_min, d_max, d_ur);
    if (rc) return rc;
    frame_gather_angles_enqueue(st, Lz->d_kp_un, d_qidx, nq, d_angle);
    a.q_u = d_u; a.q_v = d_v; a.q_radius = d_radius; a.q_min_level = d_min; a.q_max_level = d_max; a.q_ur = d_ur; a.q_angle = d_angle;
    a.t_kp = C->d_kp_un; a.t_uright = (C->had_depth && s->sp.bf > 0) ? C->d_uright : nullptr; a.t_desc = C->ctx->orb.d_desc;
    a.nq = nq; a.nt = nt; a.mnMinX = s->bounds[0]; a.mnMaxX = s->bounds[1]; a.mnMinY = s->bounds[2]; a.mnMaxY = s->bounds[3];
    a.th_high = th_high; a.check_orientation = check_orientation; a.map_mode = 0; a.nn_ratio = 0.f;
    a.match_idx = dout; a.match_dist = dout + nq; a.n_matches = dout + 2 * nq;
    if ((rc = match_sbp_enqueue(st, a, scratch))) return rc;
    ST_HIP(hipMemcpyAsync(hout, dout, (2 * (size_t)nq + 1) * 4, hipMemcpyDeviceToHost, st));
    if (q_uv) ST_HIP(hipMemcpyAsync(huv, d_u, (size_t)nq * 8, hipMemcpyDeviceToHost, st));
    ST_HIP(hipStreamSynchronize(st));
    memcpy(match_idx, hout, (size_t)nq * 4); memcpy(match_dist, hout + nq, (size_t)nq * 4);
    *n_matches = hout[2 * nq];
    if (q_uv) for (int i = 0; i < nq; i++) { q_uv[2 * i] = huv[i]; q_uv[2 * i + 1] = huv[nq + i]; }
    return HVO_OK;
}

// Line matching between two resident frames: query = the lines of frame `from`, train = the lines of frame `to`, e.g.
// LSDmatcher::match(LastFrame.mLdesc, CurrentFrame.mLdesc, nnr, matches_12) (src/LSDmatcher.cpp:42) is from = last, to = cur.
// mode: HVO_LINE_MATCH_NNR (th unused) / _BF / _DOUBLE.  matches12 needs n_kl(from) entries.
int hvo_stream_match_lines(hvo_stream *s, int64_t from, int64_t to, int mode, float th, float nnratio, int32_t *matches12, int *n_from, int *n_matches)
{
    if (!s || !matches12 || !n_matches || mode < 0 || mode > 2) return HVO_ERR_INVALID_ARG;
    *n_matches = 0;
    if (!(s->sp.stages & (HVO_STAGE_LSD | HVO_STAGE_LSD_CULL))) return HVO_ERR_INVALID_ARG;
    StreamSlot *A = slot_of(s, from), *B = slot_of(s, to);
    if (!A || !B || A == B) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(s->p.device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    ST_HIP(hipEventSynchronize(A->ev_lsd));
    ST_HIP(hipEventSynchronize(B->ev_lsd));
    const int n1 = ((const int *)(A->h_out + s->lay.counts))[4], n2 = ((const int *)(B->h_out + s->lay.counts))[4];
    if (n_from) *n_from = n1;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 <= 0 || n2 < 2 || (mode == HVO_LINE_MATCH_DOUBLE && n1 < 2)) return HVO_OK;
    hipStream_t st = s->s_match;
    ST_HIP(hipStreamWaitEvent(st, A->ev_lsd, 0));
    ST_HIP(hipStreamWaitEvent(st, B->ev_lsd, 0));
    char *d = s->d_ms, *hh = s->h_ms; size_t off = 0;
    int32_t *dm = (int32_t *)(d + off); int32_t *hm = (int32_t *)(hh + off); off += al64(((size_t)n1 + 1) * 4);
    void *scratch = d + off; off += match_lines_scratch_bytes(n1, n2);
    if (off > s->ms_bytes) { s->last_error = "matching scratch too small"; return HVO_ERR_CAPACITY; }
    int rc = match_lines_enqueue(st, A->lv.d_desc, n1, B->lv.d_desc, n2, th, nnratio, mode, scratch, dm, (int *)(dm + n1));
    if (rc) return rc;
    ST_HIP(hipMemcpyAsync(hm, dm, ((size_t)n1 + 1) * 4, hipMemcpyDeviceToHost, st));
    ST_HIP(hipStreamSynchronize(st));
    memcpy(matches12, hm, (size_t)n1 * 4);
    *n_matches = hm[n1];
    return HVO_OK;
}

// LSDmatcher::SearchByGeomNApearance(Cur, Last) between two resident frames (src/Tracking.cc:2299 -> src/LSDmatcher.cpp:36-108): the descriptor
// match AND the angle / end-point gates on the resident key lines; only the per-line "has a map line" flags go up
int hvo_stream_match_lines_geom(hvo_stream *s, int64_t cur, int64_t last, float desc_th, const uint8_t *last_has_mapline,
                                int32_t *matches12, uint8_t *accepted, int *n_last, int *n_accepted)
{
    if (!s || !matches12 || !accepted || !n_accepted) return HVO_ERR_INVALID_ARG;
    *n_accepted = 0;
    if (!(s->sp.stages & (HVO_STAGE_LSD | HVO_STAGE_LSD_CULL))) return HVO_ERR_INVALID_ARG;
    StreamSlot *A = slot_of(s, last), *B = slot_of(s, cur);
    if (!A || !B || A == B) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(s->p.device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    ST_HIP(hipEventSynchronize(A->ev_lsd));
    ST_HIP(hipEventSynchronize(B->ev_lsd));
    const int n1 = ((const int *)(A->h_out + s->lay.counts))[4], n2 = ((const int *)(B->h_out + s->lay.counts))[4];
    if (n_last) *n_last = n1;
    for (int i = 0; i < n1; i++) { matches12[i] = -1; accepted[i] = 0; }
    if (n1 <= 0 || n2 < 2) return HVO_OK;
    hipStream_t st = s->s_match;
    char *d = s->d_ms, *hh = s->h_ms; size_t off = 0;
    int32_t *dm = (int32_t *)(d + off); int32_t *hm = (int32_t *)(hh + off); off += al64(((size_t)n1 + 1) * 4);
    uint8_t *da = (uint8_t *)(d + off); uint8_t *ha = (uint8_t *)(hh + off); off += al64((size_t)n1);
    uint8_t *dml = (uint8_t *)(d + off); uint8_t *hml = (uint8_t *)(hh + off); off += al64((size_t)n1);
    void *scratch = d + off; off += match_lines_scratch_bytes(n1, n2);
    if (off > s->ms_bytes) { s->last_error = "matching scratch too small"; return HVO_ERR_CAPACITY; }
    if (last_has_mapline) { memcpy(hml, last_has_mapline, (size_t)n1); ST_HIP(hipMemcpyAsync(dml, hml, (size_t)n1, hipMemcpyHostToDevice, st)); }
    int rc = match_lines_geom_enqueue(st, A->lv.d_desc, A->lv.d_kl, last_has_mapline ? dml : nullptr, n1, B->lv.d_desc, B->lv.d_kl, n2, desc_th, s->bounds4(), scratch, dm, da);
    if (rc) return rc;
    ST_HIP(hipMemcpyAsync(hm, dm, ((size_t)n1 + 1) * 4, hipMemcpyDeviceToHost, st));
    ST_HIP(hipMemcpyAsync(ha, da, (size_t)n1, hipMemcpyDeviceToHost, st));
    ST_HIP(hipStreamSynchronize(st));
    memcpy(matches12, hm, (size_t)n1 * 4); memcpy(accepted, ha, (size_t)n1);
    *n_accepted = hm[n1];
    return HVO_OK;
}

// LSDmatcher::SearchByProjection(Cur, Last, th) core between two resident frames (src/LSDmatcher.cpp:561-662): query i = last-frame line q_index[i]
// (its key line and -- unless q_desc is given -- its descriptor are read from the last frame's slot), the current frame's key lines, line functions,
// descriptors and LINE GRID are the resident ones (the stream must run HVO_STAGE_GRIDS); per query only the four projected coordinates go up
int hvo_stream_search_lines_by_projection(hvo_stream *s, int64_t cur, int64_t last, int nq, const int32_t *q_index, const float *q_xyxy, const uint8_t *q_desc,
                                          const uint8_t *q_blocks, const uint8_t *t_occupied, float th, int32_t *match_idx, int32_t *match_dist, int *n_matches)
{
    if (!s || !match_idx || !match_dist || !n_matches || nq < 0) return HVO_ERR_INVALID_ARG;
    *n_matches = 0;
    if (!(s->sp.stages & (HVO_STAGE_LSD | HVO_STAGE_LSD_CULL)) || !(s->tail_stages & HVO_STAGE_GRIDS)) return HVO_ERR_INVALID_ARG;
    StreamSlot *A = slot_of(s, last), *B = slot_of(s, cur);
    if (!A || !B || A == B) return HVO_ERR_INVALID_ARG;
    for (int i = 0; i < nq; i++) { match_idx[i] = -1; match_dist[i] = 256; }
    if (nq == 0) return HVO_OK;
    if (!q_index || !q_xyxy) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(s->p.device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    ST_HIP(hipEventSynchronize(A->ev_lsd));
    ST_HIP(hipEventSynchronize(B->ev_lsd));                     // (recorded behind the line grid and its download)
    const int n1 = ((const int *)(A->h_out + s->lay.counts))[4], n2 = ((const int *)(B->h_out + s->lay.counts))[4];
    for (int i = 0; i < nq; i++) if (q_index[i] < 0 || q_index[i] >= n1) return HVO_ERR_INVALID_ARG;
    if (n2 <= 0) return HVO_OK;
    const TailLayout &T = s->tl;
    const int n_items = ((const int *)(B->h_tail + T.counts))[3];
    if (n_items < 0 || n_items > T.ln_cap) { s->last_error = "line grid overflowed its capacity"; return HVO_ERR_CAPACITY; }
    hipStream_t st = s->s_match;
    char *d = s->d_ms, *hh = s->h_ms; size_t off = 0;
    auto up = [&](const void *src, size_t bytes) -> void * {
        void *dp = d + off, *hp = hh + off; off += al64(bytes);
        if (off > s->ms_bytes) return nullptr;
        memcpy(hp, src, bytes);
        if (hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return nullptr;
        return dp;
    };
    LsbpDev a; memset(&a, 0, sizeof(a));
    a.nq = nq; a.nt = n2;
    a.q_xyxy = (const float *)up(q_xyxy, (size_t)nq * 16); a.q_index = (const int32_t *)up(q_index, (size_t)nq * 4);
    a.q_kl = A->lv.d_kl; a.q_desc_all = A->lv.d_desc;
    a.q_desc = q_desc ? (const uint8_t *)up(q_desc, (size_t)nq * 32) : nullptr;
    a.q_blocks = q_blocks ? (const uint8_t *)up(q_blocks, (size_t)nq) : nullptr;
    a.t_occ = t_occupied ? (const uint8_t *)up(t_occupied, (size_t)n2) : nullptr;
    if (!a.q_xyxy || !a.q_index || (q_desc && !a.q_desc) || (q_blocks && !a.q_blocks) || (t_occupied && !a.t_occ)) { s->last_error = "matching scratch too small"; return HVO_ERR_CAPACITY; }
    a.t_kl = B->lv.d_kl; a.t_fn = B->lv.d_fn; a.t_desc = B->lv.d_desc;
    a.cell_start = (const int32_t *)(B->d_tail + T.ln_start); a.cell_items = (const int32_t *)(B->d_tail + T.ln_items); a.n_items = n_items;
    a.mnMinX = s->bounds[0]; a.mnMaxX = s->bounds[1]; a.mnMinY = s->bounds[2]; a.mnMaxY = s->bounds[3]; a.th = th; a.cos_th = cos(10.0 / 180.0 * M_PI);
    int32_t *dout = (int32_t *)(d + off); int32_t *hout = (int32_t *)(hh + off); off += al64((2 * (size_t)nq + 1) * 4);
    void *scratch = d + off; off += match_lsbp_scratch_bytes(nq, n2);
    if (off > s->ms_bytes) { s->last_error = "matching scratch too small"; return HVO_ERR_CAPACITY; }
    a.match_idx = dout; a.match_dist = dout + nq; a.n_matches = dout + 2 * nq;
    int rc = match_lsbp_enqueue(st, a, scratch);
    if (rc) return rc;
    ST_HIP(hipMemcpyAsync(hout, dout, (2 * (size_t)nq + 1) * 4, hipMemcpyDeviceToHost, st));
    ST_HIP(hipStreamSynchronize(st));
    memcpy(match_idx, hout, (size_t)nq * 4); memcpy(match_dist, hout + nq, (size_t)nq * 4);
    *n_matches = hout[2 * nq];
    return HVO_OK;
}

int hvo_stream_set_readings(hvo_stream *s, unsigned mask)
{
    if (!s) return HVO_ERR_INVALID_ARG;
    for (int i = 0; i < s->depth; i++) { const int rc = hvo_set_readings(s->slot[i].ctx, mask); if (rc) return rc; }
    return HVO_OK;
}

}  // extern "C"
