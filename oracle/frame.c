/*
 * frame.c -- oracle for the Frame post-processing of the front-end's outputs:
 *   Frame::UndistortKeyPoints      reference src/Frame.cc:1701-1731
 *   Frame::ComputeImageBounds      reference src/Frame.cc:1733-1762
 *   Frame::AssignFeaturesToGrid    reference src/Frame.cc:832-847  (PosInGrid 1680-1690)
 *   Frame::AssignFeaturesToGridForLine  reference src/Frame.cc:849-872  (src/lineIterator.cpp:34-76)
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  cv::undistortPoints is restated from OpenCV 3.2.0
 * (imgproc/src/undistort.cpp cvUndistortPoints: 5 fixed-point iterations in double) -- ASSUMED.
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

#define GRID_COLS 64   /* FRAME_GRID_COLS, include/Frame.h */
#define GRID_ROWS 48   /* FRAME_GRID_ROWS */

/* cv::undistortPoints(src CV_32FC2, dst, K, dist, R = I, P = K); dist = k1 k2 p1 p2 k3 (k3 = 0 for a 4-vector).
 * K entries arrive as the floats of mK promoted to double. */
void orc_undistort_points(const float *xy_in, int n, float fx, float fy, float cx, float cy, const float *dist5, float *xy_out)
{
    const double dfx = fx, dfy = fy, dcx = cx, dcy = cy, ifx = 1. / dfx, ify = 1. / dfy;
    const double k0 = dist5[0], k1 = dist5[1], p1 = dist5[2], p2 = dist5[3], k4 = dist5[4];
    for (int i = 0; i < n; i++) {
        double x = xy_in[2 * i], y = xy_in[2 * i + 1];
        x = (x - dcx) * ifx; y = (y - dcy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            /* k[5..7] (rational model) and k[8..11] (thin prism) are zero for the 4/5-coefficient model */
            const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
            const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
            const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y + 0 * r2 + 0 * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        /* RR = P * R = K: xx = fx*x + 0*y + cx, ww = 1/(0*x + 0*y + 1) */
        const double xx = dfx * x + 0 * y + dcx, yy = 0 * x + dfy * y + dcy, ww = 1. / (0 * x + 0 * y + 1);
        xy_out[2 * i] = (float)(xx * ww); xy_out[2 * i + 1] = (float)(yy * ww);
    }
}

/* Frame::UndistortKeyPoints: k1 == 0 -> copy; else positions through cv::undistortPoints, other fields kept */
void orc_undistort_keypoints(const orc_keypoint *kp, int n, float fx, float fy, float cx, float cy, const float *dist5, orc_keypoint *kp_un)
{
    memcpy(kp_un, kp, (size_t)n * sizeof(orc_keypoint));
    if (dist5[0] == 0.0f) return;
    for (int i = 0; i < n; i++) {
        float in[2] = { kp[i].x, kp[i].y }, out[2];
        orc_undistort_points(in, 1, fx, fy, cx, cy, dist5, out);
        kp_un[i].x = out[0]; kp_un[i].y = out[1];
    }
}

/* Frame::ComputeImageBounds: bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY} */
void orc_image_bounds(int w, int h, float fx, float fy, float cx, float cy, const float *dist5, float *bounds4)
{
    if (dist5[0] != 0.0f) {
        const float in[8] = { 0, 0, (float)w, 0, 0, (float)h, (float)w, (float)h };
        float o[8];
        orc_undistort_points(in, 4, fx, fy, cx, cy, dist5, o);
        bounds4[0] = fminf(o[0], o[4]); bounds4[1] = fmaxf(o[2], o[6]);
        bounds4[2] = fminf(o[1], o[3]); bounds4[3] = fmaxf(o[5], o[7]);
    } else { bounds4[0] = 0.f; bounds4[1] = (float)w; bounds4[2] = 0.f; bounds4[3] = (float)h; }
}

/* Frame::AssignFeaturesToGrid as CSR: cell = col * 48 + row (mGrid[col][row]); cell_start has 64*48+1 entries,
 * items are key-point indices in push order (ascending).  Returns the number of assigned key points. */
int orc_assign_features_to_grid(const orc_keypoint *kp_un, int n, const float *bounds4, int32_t *cell_start, int32_t *cell_items)
{
    const float winv = (float)GRID_COLS / (bounds4[1] - bounds4[0]), hinv = (float)GRID_ROWS / (bounds4[3] - bounds4[2]);
    memset(cell_start, 0, (GRID_COLS * GRID_ROWS + 1) * sizeof(int32_t));
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < n; i++) {
            const int px = (int)roundf((kp_un[i].x - bounds4[0]) * winv), py = (int)roundf((kp_un[i].y - bounds4[2]) * hinv);
            if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
            const int c = px * GRID_ROWS + py;
            if (pass == 0) cell_start[c + 1]++;
            else cell_items[cell_start[c]++] = i;
        }
        if (pass == 0) for (int c = 0; c < GRID_COLS * GRID_ROWS; c++) cell_start[c + 1] += cell_start[c];
        else { for (int c = GRID_COLS * GRID_ROWS; c > 0; c--) cell_start[c] = cell_start[c - 1]; cell_start[0] = 0; }
    }
    return cell_start[GRID_COLS * GRID_ROWS];
}

/* ORB_SLAM2::LineIterator (src/lineIterator.cpp:34-76): cells visited by the segment, in visiting order */
static int line_cells(double x1, double y1, double x2, double y2, int *cx, int *cy, int cap)
{
    const int steep = fabs(y2 - y1) > fabs(x2 - x1);
    double t;
    if (steep) { t = x1; x1 = y1; y1 = t; t = x2; x2 = y2; y2 = t; }
    if (x1 > x2) { t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
    const double dx = x2 - x1, dy = fabs(y2 - y1);
    double error = dx / 2.0;
    const int ystep = (y1 < y2) ? 1 : -1;
    int x = (int)x1, y = (int)y1, n = 0;
    const int maxX = (int)x2;
    while (x <= maxX) {
        if (n < cap) { cx[n] = steep ? y : x; cy[n] = steep ? x : y; }
        n++;
        error -= dy;
        if (error < 0) { y += ystep; error += dx; }
        x++;
    }
    return n;
}

/* test hook: the walk above for one segment (tests/test_ref_pins.py compares it with the reference's src/lineIterator.cpp) */
int orc_grid_line_cells(double x1, double y1, double x2, double y2, int *cx, int *cy, int cap) { return line_cells(x1, y1, x2, y2, cx, cy, cap); }

/* Frame::AssignFeaturesToGridForLine as CSR (same cell numbering); the end points are scaled by the grid
 * element inverses only (no mnMinX offset, as written at Frame.cc:862).  Returns the number of items, or
 * -1 if cap is too small (cell_start is valid either way). */
int orc_assign_lines_to_grid(const orc_keyline *kl, int n, const float *bounds4, int32_t *cell_start, int32_t *cell_items, int cap)
{
    const float winv = (float)GRID_COLS / (bounds4[1] - bounds4[0]), hinv = (float)GRID_ROWS / (bounds4[3] - bounds4[2]);
    memset(cell_start, 0, (GRID_COLS * GRID_ROWS + 1) * sizeof(int32_t));
    int cxs[4096], cys[4096];
    int total = 0;
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < n; i++) {
            const int m = line_cells((double)(kl[i].sx * winv), (double)(kl[i].sy * hinv), (double)(kl[i].ex * winv), (double)(kl[i].ey * hinv), cxs, cys, 4096);
            for (int q = 0; q < m && q < 4096; q++) {
                if (cxs[q] < 0 || cxs[q] >= GRID_COLS || cys[q] < 0 || cys[q] >= GRID_ROWS) continue;
                const int c = cxs[q] * GRID_ROWS + cys[q];
                if (pass == 0) cell_start[c + 1]++;
                else { if (cell_start[c] < cap) cell_items[cell_start[c]] = i; cell_start[c]++; }
            }
        }
        if (pass == 0) { for (int c = 0; c < GRID_COLS * GRID_ROWS; c++) cell_start[c + 1] += cell_start[c]; total = cell_start[GRID_COLS * GRID_ROWS]; }
        else { for (int c = GRID_COLS * GRID_ROWS; c > 0; c--) cell_start[c] = cell_start[c - 1]; cell_start[0] = 0; }
    }
    return total <= cap ? total : -1;
}
