/*
 * lbd.c -- ORACLE (test infrastructure only; see oracle.h).  Parity unpinned.
 *
 * Line Band Descriptor as compiled into the reference via opencv_contrib 3.2 line_descriptor
 * (the vendored orphan copy is the in-tree text):
 *   BinaryDescriptor ctor (weights)    Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp:217-259
 *   computeGaussianPyramid / Sobel     binary_descriptor_custom.cpp:350-398
 *   computeImpl                        binary_descriptor_custom.cpp:539-687
 *   computeLBD                         binary_descriptor_custom.cpp:1026-1372
 *   binaryConversion / combinations    binary_descriptor_custom.cpp:401-412, 74-107
 *
 * Float evaluation rules (ASSUMED): no FMA contraction; cos/sin/round resolve to
 * the global double versions and are narrowed to float; sqrt resolves to std::sqrt(float) through
 * cv's `using std::sqrt`, so 1/sqrt(x) is a float division.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NUM_OF_BANDS 9
#define WIDTH_OF_BAND 7

static const int combinations[32][2] = {
    { 0, 1 }, { 0, 2 }, { 0, 3 }, { 0, 4 }, { 0, 5 }, { 0, 6 }, { 1, 2 }, { 1, 3 }, { 1, 4 }, { 1, 5 }, { 1, 6 },
    { 2, 3 }, { 2, 4 }, { 2, 5 }, { 2, 6 }, { 2, 7 }, { 2, 8 }, { 3, 4 }, { 3, 5 }, { 3, 6 }, { 3, 7 }, { 3, 8 },
    { 4, 5 }, { 4, 6 }, { 4, 7 }, { 4, 8 }, { 5, 6 }, { 5, 7 }, { 5, 8 }, { 6, 7 }, { 6, 8 }, { 7, 8 } };

const int *orc_lbd_combinations(void) { return &combinations[0][0]; }

/* gaussCoefL_ (21) and gaussCoefG_ (63) with the integer divisions of the reference */
void orc_lbd_weights(double *coefL21, double *coefG63)
{
    double u = (WIDTH_OF_BAND * 3 - 1) / 2;
    double sigma = (WIDTH_OF_BAND * 2 + 1) / 2;
    double invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < WIDTH_OF_BAND * 3; i++) { double dis = i - u; coefL21[i] = exp(dis * dis * invsigma2); }
    u = (NUM_OF_BANDS * WIDTH_OF_BAND - 1) / 2;
    sigma = u;
    invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < NUM_OF_BANDS * WIDTH_OF_BAND; i++) { double dis = i - u; coefG63[i] = exp(dis * dis * invsigma2); }
}

/* one line: 72 floats */
static void lbd_one(const int16_t *dxImg, const int16_t *dyImg, int realWidth, int imgH,
                    const orc_keyline *kl, const double *gL, const double *gG, float *desVec)
{
    const short imageWidth = (short)(realWidth - 1), imageHeight = (short)(imgH - 1);
    const short heightOfLSP = WIDTH_OF_BAND * NUM_OF_BANDS;
    float pgdLBandSum[NUM_OF_BANDS], ngdLBandSum[NUM_OF_BANDS], pgdL2BandSum[NUM_OF_BANDS], ngdL2BandSum[NUM_OF_BANDS];
    float pgdOBandSum[NUM_OF_BANDS], ngdOBandSum[NUM_OF_BANDS], pgdO2BandSum[NUM_OF_BANDS], ngdO2BandSum[NUM_OF_BANDS];
    memset(pgdLBandSum, 0, sizeof(pgdLBandSum)); memset(ngdLBandSum, 0, sizeof(ngdLBandSum));
    memset(pgdL2BandSum, 0, sizeof(pgdL2BandSum)); memset(ngdL2BandSum, 0, sizeof(ngdL2BandSum));
    memset(pgdOBandSum, 0, sizeof(pgdOBandSum)); memset(ngdOBandSum, 0, sizeof(ngdOBandSum));
    memset(pgdO2BandSum, 0, sizeof(pgdO2BandSum)); memset(ngdO2BandSum, 0, sizeof(ngdO2BandSum));
    const short halfHeight = (heightOfLSP - 1) / 2;
    const short lengthOfLSP = (short)kl->num_pixels;
    const short halfWidth = (lengthOfLSP - 1) / 2;
    const float lineMiddlePointX = (float)(0.5 * (kl->sox + kl->eox));
    const float lineMiddlePointY = (float)(0.5 * (kl->soy + kl->eoy));
    float dL[2], dO[2];
    dL[0] = (float)cos((double)kl->angle);
    dL[1] = (float)sin((double)kl->angle);
    dO[0] = -dL[1]; dO[1] = dL[0];
    float sCorX0 = -dL[0] * halfWidth + dL[1] * halfHeight + lineMiddlePointX;
    float sCorY0 = -dL[1] * halfWidth - dL[0] * halfHeight + lineMiddlePointY;
    for (short hID = 0; hID < heightOfLSP; hID++) {
        float sCorX = sCorX0, sCorY = sCorY0;
        float pgdLRowSum = 0, ngdLRowSum = 0, pgdORowSum = 0, ngdORowSum = 0;
        for (short wID = 0; wID < lengthOfLSP; wID++) {
            short tempCor = (short)round(sCorX);
            short xCor = (tempCor < 0) ? 0 : (tempCor > imageWidth) ? imageWidth : tempCor;
            tempCor = (short)round(sCorY);
            short yCor = (tempCor < 0) ? 0 : (tempCor > imageHeight) ? imageHeight : tempCor;
            short dx = dxImg[yCor * realWidth + xCor], dy = dyImg[yCor * realWidth + xCor];
            float gDL = dx * dL[0] + dy * dL[1];
            float gDO = dx * dO[0] + dy * dO[1];
            if (gDL > 0) pgdLRowSum += gDL; else ngdLRowSum -= gDL;
            if (gDO > 0) pgdORowSum += gDO; else ngdORowSum -= gDO;
            sCorX += dL[0];
            sCorY += dL[1];
        }
        sCorX0 -= dL[1];
        sCorY0 += dL[0];
        float coef = (float)gG[hID];
        pgdLRowSum = coef * pgdLRowSum; ngdLRowSum = coef * ngdLRowSum;
        float pgdL2RowSum = pgdLRowSum * pgdLRowSum, ngdL2RowSum = ngdLRowSum * ngdLRowSum;
        pgdORowSum = coef * pgdORowSum; ngdORowSum = coef * ngdORowSum;
        float pgdO2RowSum = pgdORowSum * pgdORowSum, ngdO2RowSum = ngdORowSum * ngdORowSum;
        short bandID = (short)(hID / WIDTH_OF_BAND);
        for (int pass = 0; pass < 3; pass++) {
            short b; int gi;
            if (pass == 0) { b = bandID; gi = hID % WIDTH_OF_BAND + WIDTH_OF_BAND; }
            else if (pass == 1) { b = bandID - 1; gi = hID % WIDTH_OF_BAND + 2 * WIDTH_OF_BAND; if (b < 0) continue; }
            else { b = bandID + 1; gi = hID % WIDTH_OF_BAND; if (b >= NUM_OF_BANDS) continue; }
            coef = (float)gL[gi];
            pgdLBandSum[b] += coef * pgdLRowSum;
            ngdLBandSum[b] += coef * ngdLRowSum;
            pgdL2BandSum[b] += coef * coef * pgdL2RowSum;
            ngdL2BandSum[b] += coef * coef * ngdL2RowSum;
            pgdOBandSum[b] += coef * pgdORowSum;
            ngdOBandSum[b] += coef * ngdORowSum;
            pgdO2BandSum[b] += coef * coef * pgdO2RowSum;
            ngdO2BandSum[b] += coef * coef * ngdO2RowSum;
        }
    }
    const float invN2 = (float)(1.0 / (WIDTH_OF_BAND * 2.0)), invN3 = (float)(1.0 / (WIDTH_OF_BAND * 3.0));
    for (short b = 0; b < NUM_OF_BANDS; b++) {
        const float invN = (b == 0 || b == NUM_OF_BANDS - 1) ? invN2 : invN3;
        const short id = b * 8;
        float temp = pgdLBandSum[b] * invN;
        desVec[id] = temp; desVec[id + 4] = sqrtf(pgdL2BandSum[b] * invN - temp * temp);
        temp = ngdLBandSum[b] * invN;
        desVec[id + 1] = temp; desVec[id + 5] = sqrtf(ngdL2BandSum[b] * invN - temp * temp);
        temp = pgdOBandSum[b] * invN;
        desVec[id + 2] = temp; desVec[id + 6] = sqrtf(pgdO2BandSum[b] * invN - temp * temp);
        temp = ngdOBandSum[b] * invN;
        desVec[id + 3] = temp; desVec[id + 7] = sqrtf(ngdO2BandSum[b] * invN - temp * temp);
    }
    float tempM = 0, tempS = 0;
    for (int b = 0; b < NUM_OF_BANDS; b++) {
        const float *d = desVec + 8 * b;
        tempM += d[0] * d[0]; tempM += d[1] * d[1]; tempM += d[2] * d[2]; tempM += d[3] * d[3];
        tempS += d[4] * d[4]; tempS += d[5] * d[5]; tempS += d[6] * d[6]; tempS += d[7] * d[7];
    }
    tempM = 1 / sqrtf(tempM);
    tempS = 1 / sqrtf(tempS);
    for (int b = 0; b < NUM_OF_BANDS; b++) {
        float *d = desVec + 8 * b;
        d[0] = d[0] * tempM; d[1] = d[1] * tempM; d[2] = d[2] * tempM; d[3] = d[3] * tempM;
        d[4] = d[4] * tempS; d[5] = d[5] * tempS; d[6] = d[6] * tempS; d[7] = d[7] * tempS;
    }
    for (int i = 0; i < NUM_OF_BANDS * 8; i++) if ((double)desVec[i] > 0.4) desVec[i] = (float)0.4;
    float temp = 0;
    for (int i = 0; i < NUM_OF_BANDS * 8; i++) temp += desVec[i] * desVec[i];
    temp = 1 / sqrtf(temp);
    for (int i = 0; i < NUM_OF_BANDS * 8; i++) desVec[i] = desVec[i] * temp;
}

/* BinaryDescriptor::compute for octave-0 keylines: blur 5x5 sigma 1, Sobel, LBD, 32 bytes each */
void orc_lbd_compute(const uint8_t *gray, int w, int h, int stride, const orc_keyline *kl, int n,
                     uint8_t *desc32, float *desc72)
{
    uint8_t *blur = (uint8_t *)malloc((size_t)w * h);
    int16_t *dx = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * h), *dy = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * h);
    orc_gaussian_blur_u8(gray, w, h, stride, blur, w, 5, 1.0);
    orc_sobel3_u8_s16(blur, w, h, w, dx, w, 1, 0);
    orc_sobel3_u8_s16(blur, w, h, w, dy, w, 0, 1);
    double gL[21], gG[63];
    orc_lbd_weights(gL, gG);
    for (int i = 0; i < n; i++) {
        float d[72];
        lbd_one(dx, dy, w, h, &kl[i], gL, gG, d);
        if (desc72) memcpy(desc72 + 72 * (size_t)i, d, sizeof(d));
        if (desc32)
            for (int c = 0; c < 32; c++) {
                const float *f1 = d + 8 * combinations[c][0], *f2 = d + 8 * combinations[c][1];
                unsigned r = 0;
                for (int b = 0; b < 8; b++) if (f1[b] > f2[b]) r += 1u << b;
                desc32[32 * (size_t)i + c] = (uint8_t)r;
            }
    }
    free(blur); free(dx); free(dy);
}
