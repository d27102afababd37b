// tail.hip -- the rest of the Frame constructor on the device-resident results (reference src/Frame.cc:205-233):
//   HVO_STAGE_LINES3D     Frame::isLineGood for every key line                  src/Frame.cc:934-939, 1205-1322      (line3d.hip)
//   HVO_STAGE_VP          vanishing-point hypotheses + line2Vps                 src/Frame.cc:328-337, 442-778        (vps.hip)
//   HVO_STAGE_PLANE_TAIL  per-plane voxel clouds, gate, SAC refit, 1/3-resolution surface normals   src/Frame.cc:2110-2212, 2214-2274   (planes_tail.hip)
//   HVO_STAGE_GRIDS       AssignFeaturesToGrid / AssignFeaturesToGridForLine    src/Frame.cc:832-872                 (frame.hip)
// Each piece reads what the front-end left in HBM -- the culled key lines and their count, the raw depth image, the int8 label image,
// the plane records, the undistorted key points -- through the *_enqueue forms of those files: nothing is uploaded a second time,
// nothing is allocated per frame, nothing synchronises.  One TailBuf holds a frame's scratch and one block of results laid out as
// TailLayout says (the same layout in HBM and in the pinned block the streamed mode downloads into).
#include "hvo_internal.hpp"
#include <string.h>
#include <algorithm>
#include <new>
#include <vector>

// device buffer of at least `bytes` that lives as long as the context: the staging arena of the host-array entry points
void *hvo_call_arena(hvo_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->call_arena_cap && ctx->call_arena) return ctx->call_arena;
    (void)hipStreamSynchronize(ctx->stream);                     // nothing enqueued may still use the old buffer
    if (ctx->call_arena) (void)hipFree(ctx->call_arena);
    ctx->call_arena = nullptr; ctx->call_arena_cap = 0;
    const size_t cap = (bytes + (bytes >> 2) + 4095) & ~(size_t)4095;
    if (hipMalloc(&ctx->call_arena, cap) != hipSuccess) { ctx->last_error = "hipMalloc(call arena)"; ctx->call_arena = nullptr; return nullptr; }
    ctx->call_arena_cap = cap;
    return ctx->call_arena;
}

static size_t tl_al(size_t v) { return (v + 255) & ~(size_t)255; }

void tail_layout(int w, int h, int kp_cap, int nfeat, TailLayout &L)
{
    memset(&L, 0, sizeof(L));
    L.w = w; L.h = h; L.kp_cap = kp_cap; L.nfeat = nfeat;
    L.cloud_cap = HVO_TAIL_CLOUD_CAP; L.n_normals = sn_count(w, h); L.ln_cap = nfeat * 128;
    size_t o = 0;
    L.counts = o; o += tl_al(16 * sizeof(int));                 // [0] n_cloud [1] cloud capacity flag [2] pt items [3] ln items
    L.lines3d = o; o += tl_al((size_t)nfeat * sizeof(hvo_line3d));
    L.vp_res = o; o += tl_al(sizeof(hvo_vp_result));
    L.vp_idx = o; o += tl_al((size_t)nfeat * 4);
    L.pclouds = o; o += tl_al(64 * sizeof(hvo_plane_cloud));
    L.cloud = o; o += tl_al((size_t)L.cloud_cap * 12);
    L.normals = o; o += tl_al((size_t)std::max(L.n_normals, 1) * sizeof(hvo_surface_normal));
    L.pt_start = o; o += tl_al((HVO_GRID_COLS * HVO_GRID_ROWS + 1) * 4);
    L.pt_items = o; o += tl_al((size_t)kp_cap * 4);
    L.ln_start = o; o += tl_al((HVO_GRID_COLS * HVO_GRID_ROWS + 1) * 4);
    L.ln_items = o; o += tl_al((size_t)L.ln_cap * 4);
    L.total = o;
    // scratch
    o = 0;
    L.s_vp = o; o += tl_al(vp_scratch_bytes(nfeat));
    L.s_pc = o; o += tl_al(pc_scratch_bytes(L.cloud_cap));
    L.s_sn = o; o += tl_al(sn_scratch_bytes(w, h));
    L.s_ptcell = o; o += tl_al(frame_grid_scratch_ints(kp_cap, false) * 4);
    L.s_lncell = o; o += tl_al(frame_grid_scratch_ints(nfeat, true) * 4);
    L.scratch_total = o;
}

// lines of a frame: 3-D lines, vanishing points, line grid -- on the stream the frame's line chain runs on
int tail_enqueue_lines(hvo_ctx *ctx, hipStream_t st, unsigned stages, const TailLayout &L, char *d_out, char *d_scratch,
                       const hvo_keyline *d_kl, const int *d_nkl, const uint16_t *d_depth, int pitch, unsigned seed, double vp_th_angle, const float *bounds4)
{
    int rc;
    if ((stages & HVO_STAGE_LINES3D) && d_depth &&
        (rc = lines3d_enqueue(ctx, st, d_kl, d_nkl, L.nfeat, d_depth, pitch, L.w, L.h, seed, (hvo_line3d *)(d_out + L.lines3d)))) return rc;
    if ((stages & HVO_STAGE_VP) &&
        (rc = vp_enqueue(ctx, st, d_kl, d_nkl, L.nfeat, seed, vp_th_angle, d_scratch + L.s_vp, (hvo_vp_result *)(d_out + L.vp_res), (int32_t *)(d_out + L.vp_idx), nullptr))) return rc;
    if ((stages & HVO_STAGE_GRIDS) &&
        (rc = frame_lines_grid_enqueue(ctx, st, d_kl, d_nkl, L.nfeat, bounds4, (int *)(d_scratch + L.s_lncell), (int32_t *)(d_out + L.ln_start), (int32_t *)(d_out + L.ln_items),
                                       L.ln_cap, (int *)(d_out + L.counts) + 3))) return rc;
    return HVO_OK;
}

// planes of a frame: per-plane clouds + refit -- on the stream the frame's plane chain runs on
int tail_enqueue_planes(hvo_ctx *ctx, hipStream_t st, unsigned stages, const TailLayout &L, char *d_out, char *d_scratch,
                        const uint16_t *d_depth, int pitch, const int8_t *d_labels8, const hvo_plane *d_planes, const int *d_npl, double dist_th)
{
    if (!(stages & HVO_STAGE_PLANE_TAIL)) return HVO_OK;
    return pc_enqueue(ctx, st, d_depth, pitch, L.w, L.h, d_labels8, d_planes, d_npl, 64, dist_th, d_scratch + L.s_pc, (float *)(d_out + L.cloud), L.cloud_cap,
                      (hvo_plane_cloud *)(d_out + L.pclouds), (int *)(d_out + L.counts));
}

// the 1/3-resolution surface normals need the depth image only, not the planes: they run beside the plane chain (on the short ORB
// stream) instead of behind its 20 ms
int tail_enqueue_normals(hvo_ctx *ctx, hipStream_t st, unsigned stages, const TailLayout &L, char *d_out, char *d_scratch, const uint16_t *d_depth, int pitch)
{
    if (!(stages & HVO_STAGE_PLANE_TAIL) || !d_depth) return HVO_OK;
    return sn_enqueue(ctx, st, d_depth, pitch, L.w, L.h, d_scratch + L.s_sn, (hvo_surface_normal *)(d_out + L.normals));
}

// points of a frame: the 64 x 48 grid of the undistorted key points -- on the stream the frame's ORB chain runs on
int tail_enqueue_points(hvo_ctx *ctx, hipStream_t st, unsigned stages, const TailLayout &L, char *d_out, char *d_scratch,
                        const hvo_keypoint *d_kp_un, const int *d_nkp, const float *bounds4)
{
    if (!(stages & HVO_STAGE_GRIDS)) return HVO_OK;
    return frame_points_grid_enqueue(ctx, st, d_kp_un, d_nkp, L.kp_cap, bounds4, (int *)(d_scratch + L.s_ptcell), (int32_t *)(d_out + L.pt_start), (int32_t *)(d_out + L.pt_items),
                                     L.kp_cap, (int *)(d_out + L.counts) + 2);
}

// a result block (host copy, layout L) -> the caller's arrays
int tail_unpack(const TailLayout &L, unsigned stages, const char *ho, int n_kl, hvo_frame_tail *out)
{
    if (!out) return HVO_OK;
    const int *cnt = (const int *)(ho + L.counts);
    out->status = HVO_OK; out->n_cloud = out->n_normals = out->n_pt_items = out->n_ln_items = 0;
    const int nl = std::min(n_kl, L.nfeat);
    if (stages & HVO_STAGE_LINES3D) { if (out->lines3d) memcpy(out->lines3d, ho + L.lines3d, (size_t)nl * sizeof(hvo_line3d)); }
    if (stages & HVO_STAGE_VP) {
        if (out->vp) memcpy(out->vp, ho + L.vp_res, sizeof(hvo_vp_result));
        if (out->vp_idx) memcpy(out->vp_idx, ho + L.vp_idx, (size_t)nl * 4);
    }
    if (stages & HVO_STAGE_PLANE_TAIL) {
        if (out->plane_clouds) memcpy(out->plane_clouds, ho + L.pclouds, 64 * sizeof(hvo_plane_cloud));
        int m = cnt[0];
        out->n_cloud = m;
        if (cnt[1]) out->status = HVO_ERR_CAPACITY;
        if (out->cloud_xyz) { if (m > out->cloud_cap) { m = out->cloud_cap; out->status = HVO_ERR_CAPACITY; } memcpy(out->cloud_xyz, ho + L.cloud, (size_t)std::min(m, L.cloud_cap) * 12); }
        int k = L.n_normals;
        out->n_normals = k;
        if (out->normals) { if (k > out->normals_cap) { k = out->normals_cap; out->status = HVO_ERR_CAPACITY; } memcpy(out->normals, ho + L.normals, (size_t)k * sizeof(hvo_surface_normal)); }
    }
    if (stages & HVO_STAGE_GRIDS) {
        const int ncell = HVO_GRID_COLS * HVO_GRID_ROWS + 1;
        out->n_pt_items = cnt[2]; out->n_ln_items = cnt[3];
        if (out->pt_cell_start) memcpy(out->pt_cell_start, ho + L.pt_start, (size_t)ncell * 4);
        if (out->pt_cell_items) { int m = std::min(cnt[2], L.kp_cap); if (m > out->pt_items_cap) { m = out->pt_items_cap; out->status = HVO_ERR_CAPACITY; } memcpy(out->pt_cell_items, ho + L.pt_items, (size_t)m * 4); }
        if (out->ln_cell_start) memcpy(out->ln_cell_start, ho + L.ln_start, (size_t)ncell * 4);
        if (cnt[3] > L.ln_cap) out->status = HVO_ERR_CAPACITY;
        if (out->ln_cell_items) { int m = std::min(cnt[3], L.ln_cap); if (m > out->ln_items_cap) { m = out->ln_items_cap; out->status = HVO_ERR_CAPACITY; } memcpy(out->ln_cell_items, ho + L.ln_items, (size_t)m * 4); }
    }
    return HVO_OK;
}

// ---- resident batch (hvo_batch_run with tail stages): one scratch, a result block per frame ---------------------------------
struct TailBatch { TailLayout L; char *d_out = nullptr, *d_scratch = nullptr; int frames = 0; unsigned seed = 1; double dist_th = 0.05, vp_th = 1.0 / 180.0 * 3.1415926535897932384626433832795; };

void tail_batch_free(hvo_ctx *ctx)
{
    TailBatch *T = (TailBatch *)ctx->tail;
    if (!T) return;
    if (T->d_out) (void)hipFree(T->d_out);
    if (T->d_scratch) (void)hipFree(T->d_scratch);
    delete T;
    ctx->tail = nullptr;
}

extern "C" int hvo_set_tail_params(hvo_ctx *ctx, uint32_t seed, double plane_dist_th, double vp_th_angle)
{
    if (!ctx) return HVO_ERR_INVALID_ARG;
    if (!ctx->tail) { ctx->tail = new (std::nothrow) TailBatch(); if (!ctx->tail) return HVO_ERR_HIP; }
    TailBatch *T = (TailBatch *)ctx->tail;
    T->seed = seed;
    if (plane_dist_th > 0) T->dist_th = plane_dist_th;
    if (vp_th_angle > 0) T->vp_th = vp_th_angle;
    return HVO_OK;
}

extern "C" int hvo_tail_capacity(int kl_cap, int w, int h, int *cloud_cap, int *n_normals, int *ln_items_cap)
{
    if (w < 3 || h < 3 || kl_cap < 0) return HVO_ERR_INVALID_ARG;
    if (cloud_cap) *cloud_cap = HVO_TAIL_CLOUD_CAP;
    if (n_normals) *n_normals = sn_count(w, h);
    if (ln_items_cap) *ln_items_cap = kl_cap * 128;
    return HVO_OK;
}

// The tail of every frame of the resident batch, frame after frame on the three subsystem streams (each frame's pieces behind the
// stage that produces their input; the frames share one scratch, so a stream runs them in order).  Launch-bound (~25 small kernels
// per frame): meant for the moderate batches of a tracker, not for the 8192-frame throughput batch.
int tail_batch_run(hvo_ctx *ctx, unsigned stages)
{
    const unsigned ts = stages & (HVO_STAGE_LINES3D | HVO_STAGE_VP | HVO_STAGE_PLANE_TAIL | HVO_STAGE_GRIDS);
    if (!ts) return HVO_OK;
    const int n = ctx->batch_n, w = ctx->batch_w, h = ctx->batch_h;
    if ((ts & (HVO_STAGE_LINES3D | HVO_STAGE_VP)) && !(ctx->last_stages & HVO_STAGE_LSD)) return HVO_ERR_INVALID_ARG;
    if ((ts & HVO_STAGE_PLANE_TAIL) && !(ctx->last_stages & HVO_STAGE_PLANES)) return HVO_ERR_INVALID_ARG;
    if ((ts & HVO_STAGE_GRIDS) && (ctx->last_stages & (HVO_STAGE_ORB | HVO_STAGE_LSD)) != (HVO_STAGE_ORB | HVO_STAGE_LSD)) return HVO_ERR_INVALID_ARG;
    if ((ts & HVO_STAGE_LINES3D) && !ctx->have_depth) return HVO_ERR_INVALID_ARG;
    if (!ctx->tail) { ctx->tail = new (std::nothrow) TailBatch(); if (!ctx->tail) return HVO_ERR_HIP; }
    TailBatch *T = (TailBatch *)ctx->tail;
    LsdView lv; PeacView pv; memset(&lv, 0, sizeof(lv)); memset(&pv, 0, sizeof(pv));
    int rc;
    int nfeat = ctx->p.lsd_nfeatures;
    if (ctx->last_stages & HVO_STAGE_LSD) { if ((rc = lsd_prepare(ctx, w, h, std::max(n, ctx->p.max_batch), ctx->last_cull, &lv))) return rc; nfeat = lv.nfeat; }
    if (ctx->have_depth && (rc = peac_prepare(ctx, w, h, std::max(n, ctx->p.max_batch), &pv))) return rc;
    const int kp_cap = ctx->orb.kp_cap > 0 ? ctx->orb.kp_cap : 1;
    if (T->frames < n || T->L.w != w || T->L.h != h || T->L.kp_cap != kp_cap || T->L.nfeat != nfeat) {
        if (T->d_out) (void)hipFree(T->d_out);
        if (T->d_scratch) (void)hipFree(T->d_scratch);
        T->d_out = T->d_scratch = nullptr; T->frames = 0;
        tail_layout(w, h, kp_cap, nfeat, T->L);
        const int cap = std::max(n, ctx->p.max_batch);
        HVO_HIP(hipMalloc((void **)&T->d_out, (size_t)cap * T->L.total));
        HVO_HIP(hipMalloc((void **)&T->d_scratch, 3 * T->L.scratch_total));      // one per subsystem stream
        T->frames = cap;
    }
    const TailLayout &L = T->L;
    const float bounds[4] = { 0.f, (float)w, 0.f, (float)h };      // batch contexts carry no distortion: k1 = 0, mvKeysUn = mvKeys (Frame.cc:1703-1707)
    hipStream_t s_orb = ctx->stream, s_lsd = hvo_stream_lsd(ctx), s_pl = hvo_stream_peac(ctx);
    const size_t lstride = ((size_t)w * h + 3) & ~(size_t)3;
    for (int f = 0; f < n; f++) {
        char *out = T->d_out + (size_t)f * L.total;
        const uint16_t *depth = ctx->have_depth ? pv.d_depth + (size_t)f * pv.dframe : nullptr;
        if (ts & (HVO_STAGE_LINES3D | HVO_STAGE_VP | HVO_STAGE_GRIDS))
            if ((rc = tail_enqueue_lines(ctx, s_lsd, ts, L, out, T->d_scratch, lv.d_kl + (size_t)f * nfeat, lv.d_nkl + f, depth, pv.pitch, T->seed + (unsigned)f, T->vp_th, bounds))) return rc;
        if (ts & HVO_STAGE_PLANE_TAIL)
            if ((rc = tail_enqueue_planes(ctx, s_pl, ts, L, out, T->d_scratch + L.scratch_total, depth, pv.pitch, pv.d_labels8 + (size_t)f * lstride, pv.d_planes + (size_t)f * 64,
                                          pv.d_meta + (size_t)f * 16 + 4, T->dist_th))) return rc;
        if ((ts & HVO_STAGE_PLANE_TAIL) && (rc = tail_enqueue_normals(ctx, s_orb, ts, L, out, T->d_scratch + 2 * L.scratch_total, depth, pv.pitch))) return rc;
        if (ts & HVO_STAGE_GRIDS)
            if ((rc = tail_enqueue_points(ctx, s_orb, ts, L, out, T->d_scratch + 2 * L.scratch_total, ctx->orb.d_kp + (size_t)f * kp_cap, ctx->orb.d_nkp + f, bounds))) return rc;
    }
    HVO_HIP(hipStreamSynchronize(s_lsd)); HVO_HIP(hipStreamSynchronize(s_pl)); HVO_HIP(hipStreamSynchronize(s_orb));
    ctx->last_stages |= ts;
    return HVO_OK;
}

extern "C" int hvo_batch_download_tail(hvo_ctx *ctx, int n, hvo_frame_tail *out)
{
    if (!ctx || !out || n < 1 || n > ctx->batch_n) return HVO_ERR_INVALID_ARG;
    TailBatch *T = (TailBatch *)ctx->tail;
    const unsigned ts = ctx->last_stages & (HVO_STAGE_LINES3D | HVO_STAGE_VP | HVO_STAGE_PLANE_TAIL | HVO_STAGE_GRIDS);
    if (!T || !T->d_out || !ts) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    const TailLayout &L = T->L;
    char *hb = (char *)hvo_stage_host(ctx, L.total + 64);
    if (!hb) return HVO_ERR_HIP;
    LsdView lv; memset(&lv, 0, sizeof(lv));
    int rc;
    std::vector<int> nkl(n, 0);
    if (ctx->last_stages & HVO_STAGE_LSD) {
        if ((rc = lsd_prepare(ctx, ctx->batch_w, ctx->batch_h, std::max(n, ctx->p.max_batch), ctx->last_cull, &lv))) return rc;
        HVO_HIP(hipMemcpy(nkl.data(), lv.d_nkl, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    }
    for (int f = 0; f < n; f++) {
        HVO_HIP(hipMemcpy(hb, T->d_out + (size_t)f * L.total, L.total, hipMemcpyDeviceToHost));
        tail_unpack(L, ts, hb, nkl[f], &out[f]);
    }
    return HVO_OK;
}
