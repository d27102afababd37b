// stubs.hip -- entry points of subsystems that are not built yet report HVO_ERR_UNSUPPORTED
// (never a CPU fallback).  Each stub disappears when its subsystem file lands.
#include "hvo_internal.hpp"
#ifndef HVO_HAVE_PEAC
int peac_upload(hvo_ctx *, int, const hvo_frame_in *, int, int) { return HVO_OK; }
int peac_run(hvo_ctx *, int) { return HVO_ERR_UNSUPPORTED; }
int peac_download(hvo_ctx *, int, hvo_frame_out *) { return HVO_ERR_UNSUPPORTED; }
void peac_free(hvo_ctx *) {}
extern "C" int hvo_compute_planes(hvo_ctx *, const uint16_t *, int, int, int, int32_t *, hvo_plane *, int, int *) { return HVO_ERR_UNSUPPORTED; }
#endif
#ifndef HVO_HAVE_LSD
int lsd_run(hvo_ctx *, int) { return HVO_ERR_UNSUPPORTED; }
int lsd_download(hvo_ctx *, int, hvo_frame_out *) { return HVO_ERR_UNSUPPORTED; }
void lsd_free(hvo_ctx *) {}
extern "C" int hvo_extract_lsd(hvo_ctx *, const uint8_t *, int, int, int, hvo_keyline *, uint8_t *, double *, int, int *) { return HVO_ERR_UNSUPPORTED; }
#endif
