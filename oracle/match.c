/*
 * match.c -- ORACLE (test infrastructure only; see oracle.h).
 *
 * 256-bit Hamming distance and brute-force matching:
 *   ORBmatcher::DescriptorDistance   src/ORBmatcher.cc:1676-1692
 *   LSDmatcher::DescriptorDistance   src/LSDmatcher.cpp:1137-1153
 *   LSDmatcher::matchNNR             src/LSDmatcher.cpp:803-826  (cv::BFMatcher knnMatch k=2, ASSUMED:
 *        exhaustive, ascending distance, ties -> lower train index)
 */
#include "oracle.h"
#include <limits.h>
#include <string.h>

int orc_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    /* the SWAR bit-trick of the reference, word by word (8 x 32 bit) */
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y;
        memcpy(&x, a + 4 * i, 4); memcpy(&y, b + 4 * i, 4);
        uint32_t v = x ^ y;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

void orc_hamming_matrix(const uint8_t *q, int nq, const uint8_t *t, int nt, uint16_t *d)
{
    for (int i = 0; i < nq; i++)
        for (int j = 0; j < nt; j++)
            d[(size_t)i * nt + j] = (uint16_t)orc_descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)j);
}

void orc_hamming_knn2(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx2, int32_t *dist2)
{
    for (int i = 0; i < nq; i++) {
        int b0 = INT_MAX, b1 = INT_MAX, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            int d = orc_descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)j);
            if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = j; }
            else if (d < b1) { b1 = d; i1 = j; }
        }
        idx2[2 * i] = i0; idx2[2 * i + 1] = i1; dist2[2 * i] = b0; dist2[2 * i + 1] = b1;
    }
}

int orc_match_nnr(const uint8_t *d1, int n1, const uint8_t *d2, int n2, float nnr, int32_t *m12)
{
    int matches = 0;
    for (int i = 0; i < n1; i++) {
        int32_t idx[2], dist[2];
        orc_hamming_knn2(d1 + 32 * (size_t)i, 1, d2, n2, idx, dist);
        m12[i] = -1;
        /* the reference indexes matches_[idx][1] unconditionally: needs n2 >= 2 */
        if (n2 >= 2 && (float)dist[0] < (float)dist[1] * nnr) { m12[i] = idx[0]; matches++; }
    }
    return matches;
}
