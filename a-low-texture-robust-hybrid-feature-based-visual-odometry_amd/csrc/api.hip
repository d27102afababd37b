// api.hip -- the extern "C" boundary of libhvo.so (include/hvo.h).  No compute here: argument
// checks, device selection, staging, and dispatch to the per-subsystem batch runners.
// There is deliberately no CPU path: without a usable gfx950 device every call fails.
#include "hvo_internal.hpp"
#include <stdio.h>
#include <algorithm>
#include <math.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <new>
#include <vector>

#include <map>
#include <mutex>
int hvo_ensure_dyn_lds(const void *kernel, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> have;
    if (bytes <= 48 * 1024) return HVO_OK;
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    size_t &h = have[std::make_pair(dev, kernel)];
    if (bytes > h) {
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) { (void)hipGetLastError(); return HVO_ERR_HIP; }
        h = bytes;
    }
    return HVO_OK;
}

const int *hvo_frame_perm(hvo_ctx *ctx, int n)
{
    if (n < 2) return nullptr;
    if (ctx->kn_frame_perm.off()) return nullptr;                // HVO_FRAME_PERM=0: A/B knob
    for (auto &e : ctx->perms) if (e.first == n) return e.second;
    int bits = 0; while ((1 << bits) < n) bits++;
    std::vector<int> p; p.reserve(n);
    for (unsigned i = 0; i < (1u << bits); i++) {
        unsigned r = 0; for (int b = 0; b < bits; b++) r |= ((i >> b) & 1u) << (bits - 1 - b);
        if ((int)r < n) p.push_back((int)r);
    }
    if (ctx->perms.size() >= 8) { for (auto &e : ctx->perms) (void)hipFree(e.second); ctx->perms.clear(); }      // (lengths change with the batch: keep a few)
    int *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)n * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, p.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    ctx->perms.emplace_back(n, d);
    return d;
}

extern "C" {

int hvo_abi_version(void) { return HVO_ABI_VERSION; }

void hvo_default_params(hvo_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    // Examples/RGB-D/TUM3.yaml:41-54, 60-63, 8-11, 34
    p->orb_nfeatures = 1000; p->orb_scale_factor = 1.2f; p->orb_nlevels = 8;
    p->orb_ini_th_fast = 20; p->orb_min_th_fast = 7;
    p->lsd_num_octaves = 1; p->lsd_scale = 1.2f; p->lsd_nfeatures = 200;
    p->fx = 535.4f; p->fy = 539.2f; p->cx = 320.1f; p->cy = 247.6f;
    p->depth_map_factor = 1.0f / 5000.0f;      // Tracking.cc:156-160
    p->device = 0; p->max_batch = 1;
}

const char *hvo_strerror(int s)
{
    switch (s) {
    case HVO_OK: return "ok";
    case HVO_ERR_INVALID_ARG: return "invalid argument";
    case HVO_ERR_NO_DEVICE: return "no usable HIP device (gfx950 required; there is no CPU fallback)";
    case HVO_ERR_HIP: return "HIP runtime error";
    case HVO_ERR_UNSUPPORTED: return "unsupported configuration or image geometry";
    case HVO_ERR_CAPACITY: return "internal capacity exceeded, results truncated";
    case HVO_ERR_BAD_DTYPE: return "wrong image type";
    case HVO_ERR_BUSY: return "stream slot still holds an uncollected frame";
    default: return "unknown status";
    }
}

const char *hvo_last_error(const hvo_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int hvo_create(const hvo_params *p, hvo_ctx **out)
{
    if (!p || !out) return HVO_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return HVO_ERR_NO_DEVICE;
    if (p->device < 0 || p->device >= ndev) return HVO_ERR_NO_DEVICE;
    if (p->max_batch < 1) return HVO_ERR_INVALID_ARG;
    if (p->lsd_num_octaves != 1) return HVO_ERR_UNSUPPORTED;
    hvo_ctx *ctx = new (std::nothrow) hvo_ctx();
    if (!ctx) return HVO_ERR_INVALID_ARG;
    ctx->p = *p; ctx->device = p->device;
    for (auto &r : ctx->prof) { r.name = nullptr; r.e0 = r.e1 = nullptr; r.ms = 0; r.used = false; r.st = nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { delete ctx; return HVO_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) { delete ctx; return HVO_ERR_NO_DEVICE; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { delete ctx; return HVO_ERR_NO_DEVICE; }   // code object is gfx950 only
    // stream priorities (experiment knob HVO_PRIO="orb,lsd,peac", lower number = higher priority)
    // Three priority classes, not one: a priority class has its own hardware queues (4 by default), and the streamed mode keeps
    // depth x 3 streams busy -- with equal priorities its frames in flight serialised (92 -> 65 frames/s at depth 4).  For the
    // batch that fills the machine equal priorities are 1.5 % faster (profiles/r02_sched_sweep.txt): contexts made for one get them.
    int pr[3] = { 0, -1, 1 };
    if (ctx->p.max_batch >= 3072) pr[1] = pr[2] = 0;
    { const char *e = getenv("HVO_PRIO"); if (e) sscanf(e, "%d,%d,%d", &pr[0], &pr[1], &pr[2]); }
    if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, pr[0]) != hipSuccess ||
        hipStreamCreateWithPriority(&ctx->s_lsd, hipStreamNonBlocking, pr[1]) != hipSuccess ||
        hipStreamCreateWithPriority(&ctx->s_peac, hipStreamNonBlocking, pr[2]) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_lsd_pre, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fast, hipEventDisableTiming) != hipSuccess) { hvo_destroy(ctx); return HVO_ERR_HIP; }
    { const char *e = getenv("HVO_SCHED"); if (e) ctx->sched_cfg = atoi(e); }
    ctx->kn_frame_perm.read("HVO_FRAME_PERM"); ctx->kn_upload_single.read("HVO_UPLOAD_SINGLE");
    { const char *e = getenv("HVO_ORB_BLUR_LATE"); if (e) ctx->orb_blur_late = atoi(e) != 0; }
    int rc = orb_init_tables(ctx);
    if (rc) { hvo_destroy(ctx); return rc; }
    *out = ctx;
    return HVO_OK;
}

void hvo_destroy(hvo_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto &e : ctx->perms) (void)hipFree(e.second);
    ctx->perms.clear();
    tail_batch_free(ctx);
    if (ctx->call_arena) (void)hipFree(ctx->call_arena);
    orb_free_plan(ctx);
    match_free(ctx);
    peac_free(ctx);
    lsd_free(ctx);
    if (ctx->d_pattern) (void)hipFree(ctx->d_pattern);
    if (ctx->d_umax) (void)hipFree(ctx->d_umax);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    for (int i = 0; i < 2; i++) if (ctx->ev_stage[i]) (void)hipEventDestroy(ctx->ev_stage[i]);
    for (auto &r : ctx->prof) { if (r.e0) (void)hipEventDestroy(r.e0); if (r.e1) (void)hipEventDestroy(r.e1); }
    if (ctx->s_copy) (void)hipStreamDestroy(ctx->s_copy);
    if (ctx->s_stage_up) (void)hipStreamDestroy(ctx->s_stage_up);
    if (ctx->s_stage_down) (void)hipStreamDestroy(ctx->s_stage_down);
    if (ctx->ev_stage_up) (void)hipEventDestroy(ctx->ev_stage_up);
    if (ctx->d_stage_gray) (void)hipFree(ctx->d_stage_gray);
    if (ctx->d_stage_depth) (void)hipFree(ctx->d_stage_depth);
    if (ctx->d_result_slab) (void)hipFree(ctx->d_result_slab);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->s_lsd) (void)hipStreamDestroy(ctx->s_lsd);
    if (ctx->s_peac) (void)hipStreamDestroy(ctx->s_peac);
    if (ctx->ev_lsd_pre) (void)hipEventDestroy(ctx->ev_lsd_pre);
    if (ctx->ev_fast) (void)hipEventDestroy(ctx->ev_fast);
    delete ctx;
}

// alternative readings of two OpenCV calls (readings.hip); takes effect with the next extraction
int hvo_set_readings(hvo_ctx *ctx, unsigned mask)
{
    if (!ctx || (mask & ~(unsigned)(HVO_READING_BLUR_FLOAT | HVO_READING_LSD_8U))) return HVO_ERR_INVALID_ARG;
    ctx->readings = mask;
    return HVO_OK;
}

int hvo_profile_enable(hvo_ctx *ctx, int on)
{
    if (!ctx) return HVO_ERR_INVALID_ARG;
    ctx->profile = on != 0;
    ctx->serialize = on == 2;              // 2: also run ORB, LSD, PEAC back to back on one stream
    return HVO_OK;
}

int hvo_profile_last(const hvo_ctx *ctx, const char **names, float *ms, int cap)
{
    if (!ctx) return HVO_ERR_INVALID_ARG;
    int n = 0;
    for (int i = 0; i < ctx->nprof; i++) {
        if (!ctx->prof[i].used) continue;
        int at = -1;
        for (int k = 0; k < i; k++) if (ctx->prof[k].used && ctx->prof[k].name == ctx->prof[i].name) { at = k; break; }
        if (at >= 0) continue;                              // a later interval of a group already reported
        if (n >= cap) break;
        float tot = 0.f;
        for (int k = i; k < ctx->nprof; k++) if (ctx->prof[k].used && ctx->prof[k].name == ctx->prof[i].name) tot += ctx->prof[k].ms;
        if (names) names[n] = ctx->prof[i].name;
        if (ms) ms[n] = tot;
        n++;
    }
    // two counts ride along when the last line growing was the async one (small batches): frames the one-wave kernel had to grow again and
    // workers that found themselves on a foreign XCD (hvo_lsd_async_report; > 0 = slower than it should be, never wrong)
    int regrown = 0, foreign = 0, wpf = 0;
    if (ctx->profile && hvo_lsd_async_report(const_cast<hvo_ctx *>(ctx), &regrown, &foreign, &wpf) == HVO_OK && wpf > 0) {
        if (n < cap) { if (names) names[n] = "lsd_async_regrown_frames"; if (ms) ms[n] = (float)regrown; n++; }
        if (n < cap) { if (names) names[n] = "lsd_async_foreign_workers"; if (ms) ms[n] = (float)foreign; n++; }
    }
    return n;
}

// (the copy stream is created on first use: contexts of the streamed mode never need one, and every stream costs a hardware queue)
struct CopyHi {
    hvo_ctx *c;
    explicit CopyHi(hvo_ctx *c_) : c(c_)
    {
        if (!c->s_copy) { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); if (hipStreamCreateWithPriority(&c->s_copy, hipStreamNonBlocking, hi) != hipSuccess) c->s_copy = nullptr; }
        c->copy_hi = true;
    }
    ~CopyHi() { c->copy_hi = false; }
};

int hvo_batch_upload(hvo_ctx *ctx, int n, const hvo_frame_in *in, int w, int h)
{
    if (!ctx || !in || n < 1) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    CopyHi hi_(ctx);                                       // after hipSetDevice: the copy stream is created on the CONTEXT's device, whatever the calling thread's was
    // grey images, then depth images, one after the other: both at once (two streams, two DMA engines) upload no faster -- the link's
    // ~23.6 GB/s is the bound -- and leave no engine to the download another context of a BatchPipeline is making meanwhile (measured:
    // its download leg went from ~40 to 80-140 ms per 2048 frames)
    int rc = orb_upload(ctx, n, in, w, h);
    if (rc) return rc;
    ctx->have_depth = true;
    for (int f = 0; f < n; f++) if (!in[f].depth) ctx->have_depth = false;
    if (ctx->have_depth) { rc = peac_upload(ctx, n, in, w, h); if (rc) return rc; }
    ctx->batch_n = n; ctx->batch_w = w; ctx->batch_h = h;
    ctx->last_stages = 0;                                  // nothing has been computed for this batch yet
    return HVO_OK;
}

int hvo_batch_stage_upload(hvo_ctx *ctx, int n, const hvo_frame_in *in, int w, int h)
{
    if (!ctx || !in || n < 1) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    if (!ctx->s_stage_up) {
        int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        HVO_HIP(hipStreamCreateWithPriority(&ctx->s_stage_up, hipStreamNonBlocking, hi)); HVO_HIP(hipStreamCreateWithPriority(&ctx->s_stage_down, hipStreamNonBlocking, hi));
        HVO_HIP(hipEventCreateWithFlags(&ctx->ev_stage_up, hipEventDisableTiming));
    }
    bool depth = true;
    for (int f = 0; f < n; f++) if (!in[f].depth) depth = false;
    // A resident batch must not be touched: building a plan for another geometry or a larger batch frees and zeroes the resident slabs
    // (batch k would then run on blank images in the documented stage(k + 1) / run(k) loop).  Such a call is refused; a first batch, or a
    // context whose resident batch was given up (hvo_batch_upload of the new geometry), builds the plans here.
    const int pb = std::max(n, ctx->p.max_batch);
    if (ctx->batch_n > 0 && (!orb_plan_covers(ctx, w, h, pb) || (depth && !peac_plan_covers(ctx, w, h, pb)))) return HVO_ERR_INVALID_ARG;
    // the plans first (their geometry gives the staging slabs' sizes)
    int rc = orb_ensure_plan(ctx, w, h, pb);
    if (rc) return rc;
    const size_t gb = (size_t)ctx->orb.batch * ctx->orb.pyr_bytes;
    if (ctx->stage_gray_bytes < gb) { if (ctx->d_stage_gray) (void)hipFree(ctx->d_stage_gray); ctx->d_stage_gray = nullptr; ctx->stage_gray_bytes = 0; HVO_HIP(hipMalloc((void **)&ctx->d_stage_gray, gb)); ctx->stage_gray_bytes = gb;
        HVO_HIP(hipMemsetAsync(ctx->d_stage_gray, 0, gb, ctx->s_stage_up)); }      // pitch padding and the 256-byte round-up travel with a commit: zero, as in the resident slab
    PeacView pv; memset(&pv, 0, sizeof(pv));
    if (depth) {
        if ((rc = peac_prepare(ctx, w, h, std::max(n, ctx->p.max_batch), &pv))) return rc;
        const size_t db = (size_t)std::max(n, ctx->p.max_batch) * pv.dframe * sizeof(uint16_t);
        if (ctx->stage_depth_bytes < db) { if (ctx->d_stage_depth) (void)hipFree(ctx->d_stage_depth); ctx->d_stage_depth = nullptr; ctx->stage_depth_bytes = 0; HVO_HIP(hipMalloc((void **)&ctx->d_stage_depth, db)); ctx->stage_depth_bytes = db;
            HVO_HIP(hipMemsetAsync(ctx->d_stage_depth, 0, db, ctx->s_stage_up)); }      // row h and the pitch padding of every frame
    }
    ctx->stage_gray_dst = ctx->d_stage_gray; ctx->stage_depth_dst = depth ? ctx->d_stage_depth : nullptr;
    rc = orb_upload(ctx, n, in, w, h, false);
    if (!rc && depth) rc = peac_upload(ctx, n, in, w, h, false);
    ctx->stage_gray_dst = nullptr; ctx->stage_depth_dst = nullptr;
    if (rc) return rc;
    HVO_HIP(hipEventRecord(ctx->ev_stage_up, ctx->s_stage_up));
    ctx->stage_n = n; ctx->stage_w = w; ctx->stage_h = h; ctx->stage_depth = depth;
    return HVO_OK;
}

int hvo_batch_commit_staged(hvo_ctx *ctx)
{
    if (!ctx || ctx->stage_n < 1) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    const int n = ctx->stage_n, w = ctx->stage_w, h = ctx->stage_h;
    // the plans the staging slabs were sized with must still be the current ones (another call may have rebuilt them for another geometry
    // since): the copies below use the CURRENT plan's frame sizes
    if (!orb_plan_covers(ctx, w, h, n) || (size_t)n * ctx->orb.pyr_bytes > ctx->stage_gray_bytes) return HVO_ERR_INVALID_ARG;
    if (ctx->stage_depth && !peac_plan_covers(ctx, w, h, n)) return HVO_ERR_INVALID_ARG;
    HVO_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_stage_up, 0));
    HVO_HIP(hipMemcpyAsync(ctx->orb.d_pyr, ctx->d_stage_gray, (size_t)n * ctx->orb.pyr_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    if (ctx->stage_depth) {
        PeacView pv; memset(&pv, 0, sizeof(pv));
        int rc = peac_prepare(ctx, w, h, std::max(n, ctx->p.max_batch), &pv);
        if (rc) return rc;
        if ((size_t)n * pv.dframe * sizeof(uint16_t) > ctx->stage_depth_bytes) return HVO_ERR_INVALID_ARG;
        HVO_HIP(hipMemcpyAsync(pv.d_depth, ctx->d_stage_depth, (size_t)n * pv.dframe * sizeof(uint16_t), hipMemcpyDeviceToDevice, ctx->stream));
    }
    HVO_HIP(hipStreamSynchronize(ctx->stream));
    ctx->have_depth = ctx->stage_depth;
    ctx->batch_n = n; ctx->batch_w = w; ctx->batch_h = h; ctx->last_stages = 0; ctx->stage_n = 0;
    return HVO_OK;
}

int hvo_batch_results_async(hvo_ctx *ctx, int n, unsigned flags, void *host_slabs)
{
    if (!ctx || !host_slabs || n < 1 || n > ctx->batch_n) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    size_t sb = 0;
    int rc = hvo_batch_slab_layout_ex(ctx, flags, nullptr, nullptr, nullptr, nullptr, &sb);
    if (rc) return rc;
    if (!ctx->s_stage_down) { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); HVO_HIP(hipStreamCreateWithPriority(&ctx->s_stage_down, hipStreamNonBlocking, hi)); }
    HVO_HIP(hipStreamSynchronize(ctx->s_stage_down));       // the slab's previous content has left
    if (ctx->result_slab_bytes < (size_t)n * sb) { if (ctx->d_result_slab) (void)hipFree(ctx->d_result_slab); ctx->d_result_slab = nullptr; ctx->result_slab_bytes = 0; HVO_HIP(hipMalloc((void **)&ctx->d_result_slab, (size_t)n * sb)); ctx->result_slab_bytes = (size_t)n * sb; }
    if ((rc = hvo_batch_pack_results_ex(ctx, n, ctx->d_result_slab, flags))) return rc;       // device to device, waited for: the next run may overwrite the results
    HVO_HIP(hipMemcpyAsync(host_slabs, ctx->d_result_slab, (size_t)n * sb, hipMemcpyDeviceToHost, ctx->s_stage_down));
    return HVO_OK;
}

int hvo_batch_results_wait(hvo_ctx *ctx)
{
    if (!ctx) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    if (ctx->s_stage_down) HVO_HIP(hipStreamSynchronize(ctx->s_stage_down));
    return HVO_OK;
}

int hvo_batch_run(hvo_ctx *ctx, unsigned stages)
{
    if (!ctx || ctx->batch_n < 1) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    for (int i = 0; i < ctx->nprof; i++) ctx->prof[i].used = false;
    int rc;
    if ((stages & HVO_STAGE_PLANES) && !ctx->have_depth) return HVO_ERR_INVALID_ARG;
    // The serial stages go first so that their long single-wave kernels overlap the streaming ones (sched 0-2);
    // sched 3 / 4 (experiments): the plane stage is enqueued last, after LSD then ORB (3) or ORB then LSD (4)
    // measured (profiles/r02_sched_sweep.txt): policy 5 wins once the batch fills the wave slots (8192 frames: 207 against 218 ms,
    // 4096: 112 against 115), policy 1 below (2048 frames: 69 against 77 ms; 2048 frames of 1280x960: 344 against 387)
    // round 3 (tools/sweep.sh preset knobs1280): frames of 1280x960 prefer 7 -- their growing kernel is six times longer and loses more by waiting
    // for FAST than FAST loses beside it (3072 frames: 340 against 389 ms); 640x480 keeps 5 (188.8 against 191.3 ms at 8192 frames)
    const bool big_frames = (long long)ctx->batch_w * ctx->batch_h >= 2LL * 640 * 480;
    ctx->sched = ctx->sched_cfg >= 0 ? ctx->sched_cfg : (ctx->batch_n >= 3072 ? (big_frames ? 7 : 5) : 1);
    const bool peac_last = ctx->sched >= 3 && ctx->sched != 5 && ctx->sched != 7 && !ctx->serialize;
    const bool orb_first = (ctx->sched == 5 || ctx->sched == 7) && !ctx->serialize && (stages & HVO_STAGE_ORB);      // sched 5 (experiment): ORB, planes (flood behind k_fast_cells), LSD
    if (orb_first) { ctx->fast_recorded = false; rc = orb_run(ctx, ctx->batch_n); if (rc) return rc; }
    // a few frames at once: the slowest frame's line growing is the longest chain (it grows with the number of frames, the AHC does not), so
    // its kernels are enqueued before the plane stage's dozen launches (8 <= n <= 64; a lone frame waits for its planes, 256 frames are a throughput case
    // that prefers the planes first: 9.6 against 9.1 k frames/s)
    const bool lsd_first = !orb_first && !peac_last && !ctx->serialize && ctx->sched != 2 && ctx->sched != 4 && ctx->sched != 6 &&
                           ctx->batch_n >= 8 && ctx->batch_n <= 64 && (stages & (HVO_STAGE_LSD | HVO_STAGE_LSD_CULL)) != 0;
    if ((stages & HVO_STAGE_PLANES) && !peac_last && !lsd_first) { rc = peac_run(ctx, ctx->batch_n); if (rc) return rc; }
    // Overlap policy (HVO_SCHED; measured in profiles/r02_sched_sweep.txt).  The long serial kernels mostly exclude each other and
    // stretch whatever streams beside them; what the order CAN do is keep them from starving a kernel the others wait for.
    //   5 (default): ORB is enqueued first, then the plane stage, then LSD; k_peac_flood (all of a CU's LDS for ~40 ms) and
    //      k_lsd_grow (every wave slot) both wait for k_fast_cells, the one ORB kernel that needs LDS: FAST runs beside the AHC
    //      (registers only), then flood, growing and ORB's quadtree share the machine, then the short tails.
    //   2: ORB first, only k_lsd_grow waits for k_fast_cells.  1: ORB waits for the LSD preamble (round 1's best).
    //   0: no cross-stream ordering.  3 / 4: planes enqueued last.  6: the AHC waits for all streaming kernels.  7: as 5, growing does not wait.
    ctx->lsd_pre_recorded = false; if (!orb_first) ctx->fast_recorded = false;
    const bool want_cull = (stages & HVO_STAGE_LSD_CULL) != 0;
    const bool want_orb = (stages & HVO_STAGE_ORB) != 0 && !orb_first, want_lsd = (stages & HVO_STAGE_LSD) != 0 || want_cull;
    if (want_lsd) ctx->last_cull = want_cull;
    if ((ctx->sched == 4 || ctx->sched == 6) && want_orb && !ctx->serialize) {      // 6 (experiment): as 4, and the AHC waits for the streaming kernels
        rc = orb_run(ctx, ctx->batch_n); if (rc) return rc;
        if (want_lsd) { rc = lsd_run(ctx, ctx->batch_n, want_cull); if (rc) return rc; }
    } else if (ctx->sched == 2 && want_orb && !ctx->serialize) {
        rc = orb_run(ctx, ctx->batch_n); if (rc) return rc;                 // records ev_fast
        if (want_lsd) { rc = lsd_run(ctx, ctx->batch_n, want_cull); if (rc) return rc; }   // k_lsd_grow waits for ev_fast
    } else {
        if (want_lsd) { rc = lsd_run(ctx, ctx->batch_n, want_cull); if (rc) return rc; }
        if ((stages & HVO_STAGE_PLANES) && lsd_first) { rc = peac_run(ctx, ctx->batch_n); if (rc) return rc; }
        if (want_orb) {
            if (ctx->sched == 1 && ctx->lsd_pre_recorded && !ctx->serialize) HVO_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_lsd_pre, 0));
            rc = orb_run(ctx, ctx->batch_n); if (rc) return rc;
        }
    }
    if ((stages & HVO_STAGE_PLANES) && peac_last) { rc = peac_run(ctx, ctx->batch_n); if (rc) return rc; }
    HVO_HIP(hipStreamSynchronize(ctx->s_peac));
    HVO_HIP(hipStreamSynchronize(ctx->s_lsd));
    HVO_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->profile)
        for (int i = 0; i < ctx->nprof; i++)
            if (ctx->prof[i].used) HVO_HIP(hipEventElapsedTime(&ctx->prof[i].ms, ctx->prof[i].e0, ctx->prof[i].e1));
    ctx->last_stages |= stages & (HVO_STAGE_ORB | HVO_STAGE_PLANES);
    if (want_lsd) ctx->last_stages = (ctx->last_stages & ~(HVO_STAGE_LSD | HVO_STAGE_LSD_CULL)) | HVO_STAGE_LSD | (want_cull ? HVO_STAGE_LSD_CULL : 0u);
    // the rest of the Frame constructor on the resident results (tail.hip)
    ctx->last_stages &= ~(HVO_STAGE_LINES3D | HVO_STAGE_VP | HVO_STAGE_PLANE_TAIL | HVO_STAGE_GRIDS);
    return tail_batch_run(ctx, stages);
}

int hvo_batch_download(hvo_ctx *ctx, int n, hvo_frame_out *out)
{
    if (!ctx || !out || n < 1 || n > ctx->batch_n) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    CopyHi hi_(ctx);                                       // after hipSetDevice (see hvo_batch_upload)
    for (int f = 0; f < n; f++) { out[f].status = HVO_OK; out[f].n_kp = out[f].n_kl = out[f].n_planes = 0; }
    // only what hvo_batch_run computed for THIS resident batch is reported: a stage that did not run leaves its counts
    // at 0 (its device slabs hold the results of some earlier batch or nothing at all)
    int rc;
    bool want_kp = false, want_pl = false, want_kl = false;
    for (int f = 0; f < n; f++) { want_kp |= (out[f].kp != nullptr); want_pl |= (out[f].labels || out[f].labels8 || out[f].planes); want_kl |= (out[f].kl != nullptr); }
    if (want_kp && (ctx->last_stages & HVO_STAGE_ORB)) { rc = orb_download(ctx, n, out); if (rc) return rc; }
    if (want_pl && (ctx->last_stages & HVO_STAGE_PLANES)) { rc = peac_download(ctx, n, out); if (rc) return rc; }
    if (want_kl && (ctx->last_stages & HVO_STAGE_LSD)) { rc = lsd_download(ctx, n, out, ctx->last_cull); if (rc) return rc; }
    return HVO_OK;
}

// ---- result slabs on the device (the multi-GPU gather of SURVEY.md 8e hands these to RCCL without a host round trip) ----
static __global__ void k_pack_header(const int *__restrict__ nkp, const int *__restrict__ oflags, const int *__restrict__ nkl, const int *__restrict__ lflags,
                                     const int *__restrict__ pmeta, int n, int kp_cap, int kl_cap, int pl_cap, char *__restrict__ slabs, size_t slab_bytes)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n) return;
    int a = nkp ? nkp[f] : 0, b = nkl ? nkl[f] : 0, c = pmeta ? pmeta[(size_t)f * 16 + 4] : 0, st = HVO_OK;
    if ((oflags && oflags[f]) || (lflags && lflags[f]) || (pmeta && pmeta[(size_t)f * 16 + 3])) st = HVO_ERR_CAPACITY;
    if (a > kp_cap) a = kp_cap;
    if (b > kl_cap) b = kl_cap;
    if (c > pl_cap) c = pl_cap;
    int *h = reinterpret_cast<int *>(slabs + (size_t)f * slab_bytes);
    h[0] = a; h[1] = b; h[2] = c; h[3] = st;
}

int hvo_batch_slab_layout_ex(hvo_ctx *ctx, unsigned flags, int *kp_cap, int *kl_cap, int *pl_cap, size_t *labels_off, size_t *slab_bytes)
{
    if (!ctx || ctx->batch_n < 1 || ctx->orb.kp_cap <= 0 || (flags & ~HVO_SLAB_LABELS)) return HVO_ERR_INVALID_ARG;
    const int kc = ctx->orb.kp_cap, lc = std::max(ctx->p.lsd_nfeatures, 1), pc = 64;
    if (kp_cap) *kp_cap = kc;
    if (kl_cap) *kl_cap = lc;
    if (pl_cap) *pl_cap = pc;
    size_t sb = 16 + (size_t)kc * (sizeof(hvo_keypoint) + 32) + (size_t)lc * (sizeof(hvo_keyline) + 32 + 24) + (size_t)pc * sizeof(hvo_plane);
    if (labels_off) *labels_off = (flags & HVO_SLAB_LABELS) ? sb : 0;
    if (flags & HVO_SLAB_LABELS) sb += ((size_t)ctx->batch_w * ctx->batch_h + 15) & ~(size_t)15;
    if (slab_bytes) *slab_bytes = sb;
    return HVO_OK;
}
int hvo_batch_slab_layout(hvo_ctx *ctx, int *kp_cap, int *kl_cap, int *pl_cap, size_t *slab_bytes)
{
    return hvo_batch_slab_layout_ex(ctx, 0, kp_cap, kl_cap, pl_cap, nullptr, slab_bytes);
}

int hvo_batch_pack_results_ex(hvo_ctx *ctx, int n, void *d_slabs, unsigned flags)
{
    if (!ctx || !d_slabs || n < 1 || n > ctx->batch_n) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    int kc, lc, pc; size_t sb, lab_off;
    int rc = hvo_batch_slab_layout_ex(ctx, flags, &kc, &lc, &pc, &lab_off, &sb);
    if (rc) return rc;
    const unsigned done = ctx->last_stages;
    OrbPlan &O = ctx->orb;
    LsdView lv; memset(&lv, 0, sizeof(lv));
    PeacView pv; memset(&pv, 0, sizeof(pv));
    if (done & HVO_STAGE_LSD) { if ((rc = lsd_prepare(ctx, ctx->batch_w, ctx->batch_h, ctx->batch_n, ctx->last_cull, &lv))) return rc; }
    if (done & HVO_STAGE_PLANES) { if ((rc = peac_prepare(ctx, ctx->batch_w, ctx->batch_h, ctx->batch_n, &pv))) return rc; }
    hipStream_t st = ctx->stream;
    char *S = (char *)d_slabs;
    // the records' part of every slab is zeroed (unused capacity must not carry stale bytes into the collective); the labels are whole
    HVO_HIP(hipMemset2DAsync(S, sb, 0, (flags & HVO_SLAB_LABELS) ? lab_off : sb, (size_t)n, st));
    hipLaunchKernelGGL(k_pack_header, dim3((n + 255) / 256), dim3(256), 0, st, (done & HVO_STAGE_ORB) ? O.d_nkp : nullptr, (done & HVO_STAGE_ORB) ? O.d_flags : nullptr,
                       lv.d_nkl, lv.d_flags, pv.d_meta, n, kc, lc, pc, S, sb);
    size_t off = 16;
    auto field = [&](const void *src, size_t src_stride, size_t bytes, bool have) -> hipError_t {
        hipError_t e = hipSuccess;
        if (have) e = hipMemcpy2DAsync(S + off, sb, src, src_stride, bytes, (size_t)n, hipMemcpyDeviceToDevice, st);
        off += bytes;
        return e;
    };
    HVO_HIP(field(O.d_kp, (size_t)O.kp_cap * sizeof(hvo_keypoint), (size_t)kc * sizeof(hvo_keypoint), (done & HVO_STAGE_ORB) != 0));
    HVO_HIP(field(O.d_desc, (size_t)O.kp_cap * 32, (size_t)kc * 32, (done & HVO_STAGE_ORB) != 0));
    HVO_HIP(field(lv.d_kl, (size_t)lc * sizeof(hvo_keyline), (size_t)lc * sizeof(hvo_keyline), lv.d_kl != nullptr));
    HVO_HIP(field(lv.d_desc, (size_t)lc * 32, (size_t)lc * 32, lv.d_kl != nullptr));
    HVO_HIP(field(lv.d_fn, (size_t)lc * 24, (size_t)lc * 24, lv.d_kl != nullptr));
    HVO_HIP(field(pv.d_planes, (size_t)pc * sizeof(hvo_plane), (size_t)pc * sizeof(hvo_plane), pv.d_planes != nullptr));
    if (flags & HVO_SLAB_LABELS) {
        const size_t lb = (size_t)ctx->batch_w * ctx->batch_h, lbp = (lb + 15) & ~(size_t)15;
        if (pv.d_labels8) HVO_HIP(hipMemcpy2DAsync(S + lab_off, sb, pv.d_labels8, pv.lstride, lb, (size_t)n, hipMemcpyDeviceToDevice, st));
        else HVO_HIP(hipMemset2DAsync(S + lab_off, sb, 0xFF, lb, (size_t)n, st));          // no plane stage ran: every pixel "no plane"
        if (lbp > lb) HVO_HIP(hipMemset2DAsync(S + lab_off + lb, sb, 0, lbp - lb, (size_t)n, st));
    }
    HVO_HIP(hipStreamSynchronize(st));
    return HVO_OK;
}

int hvo_batch_pack_results(hvo_ctx *ctx, int n, void *d_slabs) { return hvo_batch_pack_results_ex(ctx, n, d_slabs, 0); }

int hvo_extract_batch(hvo_ctx *ctx, int n, const hvo_frame_in *in, hvo_frame_out *out, int w, int h, unsigned stages)
{
    int rc = hvo_batch_upload(ctx, n, in, w, h);
    if (rc) return rc;
    if ((rc = hvo_batch_run(ctx, stages))) return rc;
    return hvo_batch_download(ctx, n, out);
}

int hvo_extract_orb(hvo_ctx *ctx, const uint8_t *gray, int w, int h, int stride,
                    hvo_keypoint *kp, uint8_t *desc32, int cap, int *n)
{
    if (!ctx || !n) return HVO_ERR_INVALID_ARG;
    *n = 0;
    if (!gray || w <= 0 || h <= 0) return HVO_OK;        // ORBextractor.cc:1044
    if (!kp || !desc32 || cap < 0 || stride < w) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    hvo_frame_in in; memset(&in, 0, sizeof(in));
    in.gray = gray; in.gray_stride = stride;
    int rc = orb_upload(ctx, 1, &in, w, h);
    if (rc) return rc;
    ctx->last_stages = 0;                                  // slot 0 of the resident batch has been overwritten
    for (int i = 0; i < ctx->nprof; i++) ctx->prof[i].used = false;
    if ((rc = orb_run(ctx, 1))) return rc;
    hvo_frame_out out; memset(&out, 0, sizeof(out));
    out.kp = kp; out.desc = desc32; out.kp_cap = cap;
    if ((rc = orb_download(ctx, 1, &out))) return rc;
    *n = out.n_kp;
    return out.status;
}

int hvo_undistort_keypoints(hvo_ctx *ctx, const hvo_keypoint *kp, int n, const float dist5[5], hvo_keypoint *kp_un)
{
    if (!ctx || n < 0 || !dist5) return HVO_ERR_INVALID_ARG;
    if (n == 0) return HVO_OK;
    if (!kp || !kp_un) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    return frame_undistort(ctx, kp, n, dist5, kp_un);
}

int hvo_image_bounds(hvo_ctx *ctx, int w, int h, const float dist5[5], float bounds4[4])
{
    if (!ctx || !dist5 || !bounds4 || w <= 0 || h <= 0) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    return frame_image_bounds(ctx, w, h, dist5, bounds4);
}

int hvo_assign_features_to_grid(hvo_ctx *ctx, const hvo_keypoint *kp_un, int n, const float bounds4[4],
                                int32_t *cell_start, int32_t *cell_items, int *n_assigned)
{
    if (!ctx || n < 0 || !bounds4 || !cell_start || !n_assigned) return HVO_ERR_INVALID_ARG;
    *n_assigned = 0;
    if (!(bounds4[1] > bounds4[0]) || !(bounds4[3] > bounds4[2])) return HVO_ERR_INVALID_ARG;
    if (n == 0) { memset(cell_start, 0, (HVO_GRID_COLS * HVO_GRID_ROWS + 1) * sizeof(int32_t)); return HVO_OK; }
    if (!kp_un || !cell_items) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    return frame_points_to_grid(ctx, kp_un, n, bounds4, cell_start, cell_items, n_assigned);
}

int hvo_assign_lines_to_grid(hvo_ctx *ctx, const hvo_keyline *kl, int n, const float bounds4[4],
                             int32_t *cell_start, int32_t *cell_items, int cap, int *n_items)
{
    if (!ctx || n < 0 || cap < 0 || !bounds4 || !cell_start || !n_items) return HVO_ERR_INVALID_ARG;
    *n_items = 0;
    if (!(bounds4[1] > bounds4[0]) || !(bounds4[3] > bounds4[2])) return HVO_ERR_INVALID_ARG;
    if (n == 0) { memset(cell_start, 0, (HVO_GRID_COLS * HVO_GRID_ROWS + 1) * sizeof(int32_t)); return HVO_OK; }
    if (!kl || (cap > 0 && !cell_items)) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    return frame_lines_to_grid(ctx, kl, n, bounds4, cell_start, cell_items, cap, n_items);
}

int hvo_hamming_matrix(hvo_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, uint16_t *d)
{
    if (!ctx || nq < 0 || nt < 0) return HVO_ERR_INVALID_ARG;
    if (nq == 0 || nt == 0) return HVO_OK;
    if (!q || !t || !d) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    return match_matrix(ctx, q, nq, t, nt, d);
}

int hvo_hamming_knn2(hvo_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx2, int32_t *dist2)
{
    if (!ctx || nq < 0 || nt < 0) return HVO_ERR_INVALID_ARG;
    if (nq == 0) return HVO_OK;
    if (!q || !idx2 || !dist2 || (nt > 0 && !t)) return HVO_ERR_INVALID_ARG;
    if (nt == 0) {
        for (int i = 0; i < 2 * nq; i++) { idx2[i] = -1; dist2[i] = INT32_MAX; }
        return HVO_OK;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    return match_knn2(ctx, q, nq, t, nt, idx2, dist2);
}

int hvo_match_nnr(hvo_ctx *ctx, const uint8_t *d1, int n1, const uint8_t *d2, int n2, float nnr,
                  int32_t *m12, int *n_matches)
{
    // LSDmatcher::matchNNR (LSDmatcher.cpp:803-826): knn-2 search and ratio test on the device (match.hip)
    if (!ctx || !m12 || !n_matches || n1 < 0 || n2 < 0) return HVO_ERR_INVALID_ARG;
    *n_matches = 0;
    if (n1 == 0) return HVO_OK;
    for (int i = 0; i < n1; i++) m12[i] = -1;
    if (n2 < 2) return HVO_OK;
    if (!d1 || !d2) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    return match_lines(ctx, d1, n1, d2, n2, 0.f, nnr, HVO_LINE_MATCH_NNR, m12, n_matches);
}

// LSDmatcher::FrameBFMatch (LSDmatcher.cpp:942-966): knn-2, lineDescriptorMAD's threshold and the three tests all run on the
// device (match.hip: k_hamming_knn2 + k_frame_bf_epilogue); the call stages the two descriptor sets and fetches n1 + 1 ints
int hvo_frame_bf_match(hvo_ctx *ctx, const uint8_t *d1, int n1, const uint8_t *d2, int n2, float th, float nnratio,
                       int32_t *m12, int *n_matches)
{
    if (!ctx || !m12 || !n_matches || n1 < 0 || n2 < 0) return HVO_ERR_INVALID_ARG;
    *n_matches = 0;
    for (int i = 0; i < n1; i++) m12[i] = -1;
    if (n1 == 0 || n2 < 2) return HVO_OK;                        // knnMatch(k = 2) needs two train descriptors
    if (!d1 || !d2) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    return match_lines(ctx, d1, n1, d2, n2, th, nnratio, HVO_LINE_MATCH_BF, m12, n_matches);
}

// LSDmatcher::SearchByGeomNApearance (LSDmatcher.cpp:36-108): matchNNR, then the angle and end-point gates (line_track.inc)
int hvo_match_lines_geom(hvo_ctx *ctx, const uint8_t *d_last, const hvo_keyline *kl_last, const uint8_t *last_has_mapline, int n_last,
                         const uint8_t *d_cur, const hvo_keyline *kl_cur, int n_cur, float desc_th, const float bounds4[4],
                         int32_t *matches12, uint8_t *accepted, int *n_accepted)
{
    if (!ctx || !matches12 || !accepted || !n_accepted || !bounds4 || n_last < 0 || n_cur < 0) return HVO_ERR_INVALID_ARG;
    *n_accepted = 0;
    for (int i = 0; i < n_last; i++) { matches12[i] = -1; accepted[i] = 0; }
    if (n_last == 0 || n_cur < 2) return HVO_OK;                  // knnMatch(k = 2) needs two train rows (as hvo_match_nnr)
    if (!d_last || !kl_last || !d_cur || !kl_cur) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    return match_lines_geom(ctx, d_last, kl_last, last_has_mapline, n_last, d_cur, kl_cur, n_cur, desc_th, bounds4, matches12, accepted, n_accepted);
}

// LSDmatcher::SearchByProjection(Cur, Last, th) core (LSDmatcher.cpp:561-662) over Frame::GetFeaturesInAreaForLine (Frame.cc:1557-1627)
int hvo_search_lines_by_projection(hvo_ctx *ctx, int nq, const float *q_xyxy, const hvo_keyline *q_kl, const uint8_t *q_desc, const uint8_t *q_blocks,
                                   const hvo_keyline *t_kl, const double *t_linefn, const uint8_t *t_desc, const uint8_t *t_occupied, int nt,
                                   const int32_t *cell_start, const int32_t *cell_items, const float bounds4[4], float th,
                                   int32_t *match_idx, int32_t *match_dist, int *n_matches)
{
    if (!ctx || !match_idx || !match_dist || !n_matches || !bounds4 || nq < 0 || nt < 0) return HVO_ERR_INVALID_ARG;
    *n_matches = 0;
    for (int i = 0; i < nq; i++) { match_idx[i] = -1; match_dist[i] = 256; }
    if (nq == 0 || nt == 0) return HVO_OK;
    if (!q_xyxy || !q_kl || !q_desc || !t_kl || !t_linefn || !t_desc || !cell_start) return HVO_ERR_INVALID_ARG;
    if (!(bounds4[1] > bounds4[0]) || !(bounds4[3] > bounds4[2])) return HVO_ERR_INVALID_ARG;
    const int n_items = cell_start[HVO_GRID_COLS * HVO_GRID_ROWS];
    if (n_items < 0 || (n_items > 0 && !cell_items)) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    return match_search_lines_by_projection(ctx, nq, q_xyxy, q_kl, q_desc, q_blocks, t_kl, t_linefn, t_desc, t_occupied, nt, cell_start, cell_items, n_items, bounds4, th,
                                            match_idx, match_dist, n_matches);
}

// LSDmatcher::SearchDouble / SearchByDescriptor core (LSDmatcher.cpp:902-939): FrameBFMatch in both directions + mutual check
int hvo_search_double(hvo_ctx *ctx, const uint8_t *d1, int n1, const uint8_t *d2, int n2, float th, float nnratio,
                      int32_t *m12, int *n_matches)
{
    if (!ctx || !m12 || !n_matches || n1 < 0 || n2 < 0) return HVO_ERR_INVALID_ARG;
    *n_matches = 0;
    for (int i = 0; i < n1; i++) m12[i] = -1;
    if (n1 == 0 || n2 == 0) return HVO_OK;                       // LSDmatcher.cpp:910-911
    if (n1 < 2 || n2 < 2) return HVO_OK;                         // one of the two directions has no second neighbour: nothing survives
    if (!d1 || !d2) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    return match_lines(ctx, d1, n1, d2, n2, th, nnratio, HVO_LINE_MATCH_DOUBLE, m12, n_matches);
}

// ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, mono) core: ranked window candidates per query, then the
// reference's sequential pass (occupancy, TH_HIGH, rotation histogram) as a one-wave kernel (match.hip: k_sbp_epilogue)
int hvo_search_by_projection(hvo_ctx *ctx, const uint8_t *q_desc, int nq, const float *q_u, const float *q_v, const float *q_radius,
                             const int32_t *q_min_level, const int32_t *q_max_level, const float *q_ur, const float *q_angle,
                             const uint8_t *q_blocks, const hvo_keypoint *t_kp, const float *t_uright, const uint8_t *t_occupied,
                             const uint8_t *t_desc, int nt, float mnMinX, float mnMinY, float mnMaxX, float mnMaxY,
                             int th_high, int check_orientation, int32_t *match_idx, int32_t *match_dist, int *n_matches)
{
    if (!ctx || !n_matches || nq < 0 || nt < 0) return HVO_ERR_INVALID_ARG;
    *n_matches = 0;
    if (nq == 0) return HVO_OK;
    if (!q_desc || !q_u || !q_v || !q_radius || !q_min_level || !q_max_level || !q_angle || !q_blocks || !match_idx || !match_dist) return HVO_ERR_INVALID_ARG;
    for (int i = 0; i < nq; i++) { match_idx[i] = -1; match_dist[i] = 256; }
    if (nt == 0) return HVO_OK;
    if (!t_kp || !t_desc || nt > 65535 || !(mnMaxX > mnMinX) || !(mnMaxY > mnMinY)) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    return match_search_by_projection(ctx, q_desc, nq, q_u, q_v, q_radius, q_min_level, q_max_level, q_ur, q_angle, q_blocks, t_kp, t_uright, t_occupied,
                                      t_desc, nt, mnMinX, mnMinY, mnMaxX, mnMaxY, th_high, check_orientation, 0, 0.f, match_idx, match_dist, n_matches);
}

// ORBmatcher::SearchByProjection(F, vpMapPoints, th) core (ORBmatcher.cc:45-132): same ranked candidates, best and second best
// still-free candidate per query, same-octave ratio test
int hvo_search_by_projection_tracked(hvo_ctx *ctx, const uint8_t *q_desc, int nq, const float *proj_x, const float *proj_y, const float *proj_xr,
                                     const int32_t *level, const float *view_cos, const uint8_t *q_blocks, float th,
                                     const hvo_keypoint *t_kp, const float *t_uright, const uint8_t *t_occupied, const uint8_t *t_desc, int nt,
                                     float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, int th_high, float nn_ratio,
                                     int32_t *match_idx, int32_t *match_dist, int *n_matches)
{
    if (!ctx || !n_matches || nq < 0 || nt < 0) return HVO_ERR_INVALID_ARG;
    *n_matches = 0;
    if (nq == 0) return HVO_OK;
    if (!q_desc || !proj_x || !proj_y || !level || !view_cos || !q_blocks || !match_idx || !match_dist) return HVO_ERR_INVALID_ARG;
    for (int i = 0; i < nq; i++) if (level[i] < 0 || level[i] >= ctx->p.orb_nlevels) return HVO_ERR_INVALID_ARG;      // mvScaleFactors[level] on the device
    for (int i = 0; i < nq; i++) { match_idx[i] = -1; match_dist[i] = 256; }
    if (nt == 0) return HVO_OK;
    if (!t_kp || !t_desc || nt > 65535 || !(mnMaxX > mnMinX) || !(mnMaxY > mnMinY)) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    return match_search_by_projection_tracked(ctx, q_desc, nq, proj_x, proj_y, proj_xr, level, view_cos, q_blocks, th, t_kp, t_uright, t_occupied,
                                              t_desc, nt, mnMinX, mnMinY, mnMaxX, mnMaxY, th_high, nn_ratio, match_idx, match_dist, n_matches);
}

int hvo_search_by_projection_map(hvo_ctx *ctx, const uint8_t *q_desc, int nq, const float *q_u, const float *q_v, const float *q_radius,
                                 const int32_t *q_min_level, const int32_t *q_max_level, const float *q_ur, const uint8_t *q_blocks,
                                 const hvo_keypoint *t_kp, const float *t_uright, const uint8_t *t_occupied, const uint8_t *t_desc, int nt,
                                 float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, int th_high, float nn_ratio,
                                 int32_t *match_idx, int32_t *match_dist, int *n_matches)
{
    if (!ctx || !n_matches || nq < 0 || nt < 0) return HVO_ERR_INVALID_ARG;
    *n_matches = 0;
    if (nq == 0) return HVO_OK;
    if (!q_desc || !q_u || !q_v || !q_radius || !q_min_level || !q_max_level || !q_blocks || !match_idx || !match_dist) return HVO_ERR_INVALID_ARG;
    for (int i = 0; i < nq; i++) { match_idx[i] = -1; match_dist[i] = 256; }
    if (nt == 0) return HVO_OK;
    if (!t_kp || !t_desc || nt > 65535 || !(mnMaxX > mnMinX) || !(mnMaxY > mnMinY)) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    return match_search_by_projection(ctx, q_desc, nq, q_u, q_v, q_radius, q_min_level, q_max_level, q_ur, nullptr, q_blocks, t_kp, t_uright, t_occupied,
                                      t_desc, nt, mnMinX, mnMinY, mnMaxX, mnMaxY, th_high, 0, 1, nn_ratio, match_idx, match_dist, n_matches);
}

int hvo_stereo_from_rgbd(hvo_ctx *ctx, const hvo_keypoint *kp, const hvo_keypoint *kp_un, int n, const uint16_t *depth, int w, int h, int stride,
                         float bf, float *uright, float *zdepth)
{
    if (!ctx || n < 0) return HVO_ERR_INVALID_ARG;
    if (n == 0) return HVO_OK;
    if (!kp || !kp_un || !depth || !uright || !zdepth || w <= 0 || h <= 0 || stride < 2 * w) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    return match_stereo_from_rgbd(ctx, kp, kp_un, n, depth, w, h, stride, bf, uright, zdepth);
}

}  // extern "C"

// ---- helpers shared by the subsystem files ----
int hvo_prof_begin(hvo_ctx *ctx, const char *name, hipStream_t st)
{
    if (!ctx->profile) return -1;
    // one interval per call: a group enqueued once per chunk of the batch gets several, hvo_profile_last adds them up
    int id = -1;
    for (int i = 0; i < ctx->nprof; i++) if (ctx->prof[i].name == name && !ctx->prof[i].used) { id = i; break; }
    if (id < 0) {
        if (ctx->nprof >= HVO_MAX_PROFILE) return -1;
        id = ctx->nprof++;
        ctx->prof[id].name = name;
        (void)hipEventCreate(&ctx->prof[id].e0);
        (void)hipEventCreate(&ctx->prof[id].e1);
    }
    ctx->prof[id].used = true; ctx->prof[id].st = st;
    (void)hipEventRecord(ctx->prof[id].e0, st);
    return id;
}

void hvo_prof_end(hvo_ctx *ctx, int id)
{
    if (id >= 0) (void)hipEventRecord(ctx->prof[id].e1, ctx->prof[id].st);
}

// Batch download of one per-frame slab: frame f's record starts at dev_base + f * dev_stride and its first bytes[f]
// bytes go to dst[f] (either may be null / 0).  The slab is moved in chunks of whole frames with one DMA each into
// pinned memory (two buffers: the DMA of chunk c+1 runs while the host scatters chunk c), instead of one pageable
// hipMemcpy per frame and field (each of those is staged synchronously by the runtime, ~2 GB/s).
// widen8 != 0: the slab holds int8 values (the plane label image, -1 = none) that the caller's buffers receive as int32:
// bytes[f] counts SOURCE bytes, dst[f] gets 4 * bytes[f].  The labels cross PCIe as 1 byte per pixel.
int hvo_staged_d2h(hvo_ctx *ctx, hipStream_t st, const void *dev_base, size_t dev_stride, int n, void *const *dst, const size_t *bytes, int widen8)
{
    const size_t HALF = 64u << 20;
    if (n <= 0) return HVO_OK;
    // Direct path: equally sized, equally spaced destinations in page-locked memory (a caller's pinned result slab, see
    // hvo_pin_host) take ONE strided DMA from the device slab; no staging copy, no host memcpy.
    if (!widen8 && n >= 2 && dst[0] && bytes[0]) {
        bool regular = true;
        const ptrdiff_t D = (const char *)dst[1] - (const char *)dst[0];
        for (int f = 0; f < n && regular; f++) regular = dst[f] == (void *)((char *)dst[0] + (ptrdiff_t)f * D) && bytes[f] == bytes[0];
        if (regular && D >= (ptrdiff_t)bytes[0]) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, dst[0]) == hipSuccess && at.type == hipMemoryTypeHost) {
                HVO_HIP(hipMemcpy2DAsync(dst[0], (size_t)D, dev_base, dev_stride, bytes[0], (size_t)n, hipMemcpyDeviceToHost, st));
                HVO_HIP(hipStreamSynchronize(st));
                return HVO_OK;
            }
            (void)hipGetLastError();                               // "not a registered pointer" is not an error here
        }
    }
    if (dev_stride > HALF && !widen8) {                        // a frame larger than a buffer: plain copies
        for (int f = 0; f < n; f++)
            if (dst[f] && bytes[f]) HVO_HIP(hipMemcpyAsync(dst[f], (const char *)dev_base + (size_t)f * dev_stride, bytes[f], hipMemcpyDeviceToHost, st));
        HVO_HIP(hipStreamSynchronize(st));
        return HVO_OK;
    }
    char *stage = (char *)hvo_stage_host(ctx, 2 * HALF);
    if (!stage) { ctx->last_error = "hipHostMalloc (download staging)"; return HVO_ERR_HIP; }
    if (!ctx->ev_stage[0]) for (int i = 0; i < 2; i++) HVO_HIP(hipEventCreateWithFlags(&ctx->ev_stage[i], hipEventDisableTiming));
    if (dev_stride > HALF) { ctx->last_error = "label slab larger than the staging buffer"; return HVO_ERR_UNSUPPORTED; }
    const int per = (int)std::max<size_t>(1, HALF / dev_stride);
    const int nchunk = (n + per - 1) / per;
    for (int c = 0; c <= nchunk; c++) {
        if (c < nchunk) {
            const int f0 = c * per, F = std::min(per, n - f0);
            HVO_HIP(hipMemcpyAsync(stage + (size_t)(c & 1) * HALF, (const char *)dev_base + (size_t)f0 * dev_stride, (size_t)F * dev_stride, hipMemcpyDeviceToHost, st));
            HVO_HIP(hipEventRecord(ctx->ev_stage[c & 1], st));
        }
        if (c > 0) {
            const int pc = c - 1, f0 = pc * per, F = std::min(per, n - f0);
            HVO_HIP(hipEventSynchronize(ctx->ev_stage[pc & 1]));
            const char *src = stage + (size_t)(pc & 1) * HALF;
            for (int f = 0; f < F; f++) {
                if (!dst[f0 + f] || !bytes[f0 + f]) continue;
                if (!widen8) memcpy(dst[f0 + f], src + (size_t)f * dev_stride, bytes[f0 + f]);
                else {
                    const int8_t *s8 = (const int8_t *)(src + (size_t)f * dev_stride); int32_t *d32 = (int32_t *)dst[f0 + f];
                    for (size_t k = 0; k < bytes[f0 + f]; k++) d32[k] = (int32_t)s8[k];
                }
            }
        }
    }
    return HVO_OK;
}

// Page-lock / unlock a caller's buffer (hipHostRegister): images and result slabs in pinned memory move by DMA at the link
// rate without the runtime's staging copy.  For callers that do not link HIP themselves.
extern "C" int hvo_pin_host(void *p, size_t bytes)
{
    if (!p || !bytes) return HVO_ERR_INVALID_ARG;
    return hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess ? HVO_OK : HVO_ERR_HIP;
}
extern "C" int hvo_unpin_host(void *p)
{
    if (!p) return HVO_ERR_INVALID_ARG;
    return hipHostUnregister(p) == hipSuccess ? HVO_OK : HVO_ERR_HIP;
}

void *hvo_stage_host(hvo_ctx *ctx, size_t bytes)
{
    if (ctx->h_stage_cap >= bytes) return ctx->h_stage;
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    ctx->h_stage = nullptr; ctx->h_stage_cap = 0;
    if (hipHostMalloc(&ctx->h_stage, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    ctx->h_stage_cap = bytes;
    return ctx->h_stage;
}
