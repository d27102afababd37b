// ref_shim.cpp -- extern "C" doors into the three pieces of the REFERENCE that compile from their own files
// (no OpenCV / Eigen / PCL): include/peac/DisjointSet.hpp, include/peac/AHCParamSet.hpp and src/lineIterator.cpp
// (+ include/lineIterator.h).  Built by `make -C oracle ref` from the sources where they lie under /root/reference
// into oracle/_ref/libhvoref.so (git-ignored; nothing of the reference is copied into this repository).
// TEST INFRASTRUCTURE: tests/test_ref_pins.py checks the oracle's restatements against it.  Nothing here is a
// stand-in for a missing header: the three files only need the C++ standard library.
#include <cmath>
#include <algorithm>
#include <utility>
#include "peac/DisjointSet.hpp"
#include "peac/AHCParamSet.hpp"
#include "lineIterator.h"

extern "C" {

void *ref_ds_create(int n) { return new DisjointSet(n); }
int   ref_ds_union(void *d, int x, int y) { return static_cast<DisjointSet *>(d)->Union(x, y); }
int   ref_ds_find(void *d, int x) { return static_cast<DisjointSet *>(d)->Find(x); }
int   ref_ds_set_size(void *d, int x) { return static_cast<DisjointSet *>(d)->getSetSize(x); }
void  ref_ds_free(void *d) { delete static_cast<DisjointSet *>(d); }

// ahc::ParamSet with its defaults (AHCParamSet.hpp:68-76); phase 0 = P_INIT, 1 = P_MERGING, 2 = P_REFINE
double ref_T_mse(int phase, double z) { ahc::ParamSet p; return p.T_mse(static_cast<ahc::ParamSet::Phase>(phase), z); }
double ref_T_ang(int phase, double z) { ahc::ParamSet p; return p.T_ang(static_cast<ahc::ParamSet::Phase>(phase), z); }
double ref_T_dz(double z) { ahc::ParamSet p; return p.T_dz(z); }

// ORB_SLAM2::LineIterator (src/lineIterator.cpp:34-76): the pixels of one walk, in order
int ref_line_iterator(double x1, double y1, double x2, double y2, int *px, int *py, int cap)
{
    ORB_SLAM2::LineIterator it(x1, y1, x2, y2);
    std::pair<int, int> p;
    int n = 0;
    while (it.getNext(p)) { if (n < cap) { px[n] = p.first; py[n] = p.second; } n++; }
    return n;
}

}
