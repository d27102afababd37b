// frontend_demo.cpp -- uses the C++ mirror (include/hvo.hpp) the way Frame::Frame() uses the
// reference extractors (src/Frame.cc:205-233): ORB, lines and planes of one RGB-D frame, then a
// Hamming match of the ORB descriptors against themselves.  Reads a raw 640x480 gray (u8) + depth
// (u16) pair written by tests/test_cpp_adaptor.py and prints one line of counts/checksums.
//
// build:  g++ -std=c++14 -Iinclude examples/frontend_demo.cpp -L<csrc> -lhvo -Wl,-rpath,<csrc> -o demo
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hvo.hpp"

static uint64_t fnv(const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } return h; }

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s gray.u8 depth.u16\n", argv[0]); return 2; }
    const int W = 640, H = 480;
    std::vector<uint8_t> gray(W * H); std::vector<uint16_t> depth(W * H);
    FILE *f = fopen(argv[1], "rb"); if (!f || fread(gray.data(), 1, gray.size(), f) != gray.size()) return 3; fclose(f);
    f = fopen(argv[2], "rb"); if (!f || fread(depth.data(), 2, depth.size(), f) != depth.size()) return 3; fclose(f);
    try {
        hvo::ORBextractor orb(1000, 1.2f, 8, 20, 7);                  // Tracking.cc:124 / TUM3.yaml:41-54
        hvo::LINEextractor lsd(1, 1.2f, 200, 0);                      // Tracking.cc:132 / TUM3.yaml:60-63
        hvo::PlaneDetection planes;
        std::vector<hvo::KeyPoint> kps; std::vector<uint8_t> desc;
        orb(hvo::Image8{ gray.data(), W, H, W }, kps, desc);
        std::vector<hvo::KeyLine> kls; std::vector<uint8_t> ldesc; std::vector<double> fn;
        lsd(hvo::Image8{ gray.data(), W, H, W }, kls, ldesc, fn);
        planes.readDepthImage(hvo::Image16{ depth.data(), W, H, W * 2 }, 535.4f, 539.2f, 320.1f, 247.6f, 1.0f / 5000.0f);
        planes.runPlaneDetection();
        hvo::LSDmatcher lm(orb.ctx());
        std::vector<int> m12;
        int nm = lm.match(desc.data(), (int)kps.size(), desc.data(), (int)kps.size(), 0.9f, m12);
        // what the Frame constructor does next (Frame.cc:231-262): undistort, image bounds, feature grids
        hvo::FrameGrid fg(orb.ctx());
        const float dist[5] = { 0.f, 0.f, 0.f, 0.f, 0.f };           // TUM3.yaml:13-17
        std::vector<hvo::KeyPoint> kps_un; float b[4];
        fg.UndistortKeyPoints(kps, dist, kps_un);
        fg.ComputeImageBounds(W, H, dist, b[0], b[1], b[2], b[3]);
        std::vector<int> gstart, gitems, lstart, litems;
        fg.AssignFeaturesToGrid(kps_un, b, gstart, gitems);
        fg.AssignFeaturesToGridForLine(kls, b, lstart, litems);
        printf("grid %zu %016llx linegrid %zu %016llx ", gitems.size(), (unsigned long long)fnv(gitems.data(), gitems.size() * 4),
               litems.size(), (unsigned long long)fnv(litems.data(), litems.size() * 4));
        printf("kp %zu desc %016llx lines %zu ldesc %016llx planes %d labels %016llx matches %d\n", kps.size(),
               (unsigned long long)fnv(desc.data(), desc.size()), kls.size(), (unsigned long long)fnv(ldesc.data(), ldesc.size()),
               planes.plane_num_, (unsigned long long)fnv(planes.membership.data(), planes.membership.size() * 4), nm);
        // round 2: isLineGood, the vanishing-point clustering, the ComputePlanes tail, one frame through the streamed mode
        lsd.setCamera(535.4f, 539.2f, 320.1f, 247.6f, 1.0f / 5000.0f);
        std::vector<hvo_line3d> l3; lsd.isLineGood(kls, hvo::Image16{ depth.data(), W, H, W * 2 }, 7u, l3);
        int good = 0; for (const hvo_line3d &l : l3) good += l.good;
        std::vector<int32_t> vpi; const hvo_vp_result vp = lsd.line2Vps(kls, 7u, vpi);
        int vc[4] = { 0, 0, 0, 0 }; for (int32_t v : vpi) vc[v]++;
        std::vector<hvo_plane_cloud> pc; std::vector<float> xyz; planes.planeClouds(0.05, pc, xyz);
        int valid = 0; for (const hvo_plane_cloud &c : pc) valid += c.valid;
        std::vector<hvo_surface_normal> sn; planes.surfaceNormals(sn);
        hvo_params p; hvo_default_params(&p);
        hvo_stream_params sp = hvo_stream_params(); sp.width = W; sp.height = H; sp.depth = 2; sp.stages = HVO_STAGE_ORB | HVO_STAGE_LSD | HVO_STAGE_PLANES | HVO_STAGE_LINES3D | HVO_STAGE_VP | HVO_STAGE_PLANE_TAIL | HVO_STAGE_GRIDS; sp.bf = 40.f; sp.seed = 7u;   // the whole Frame constructor
        hvo::FrameStream fs(p, sp);
        const int64_t t = fs.submit(hvo::Image8{ gray.data(), W, H, W }, hvo::Image16{ depth.data(), W, H, W * 2 });
        std::vector<hvo::KeyPoint> skp(fs.kpCap()); std::vector<uint8_t> sdesc((size_t)fs.kpCap() * 32);
        hvo_frame_out fo = hvo_frame_out(); fo.kp = skp.data(); fo.desc = sdesc.data(); fo.kp_cap = fs.kpCap();
        hvo::FrameStream::FrameTail tail; fs.collectTail(t, W, H, tail);        // before collect() hands the slot back
        fs.collect(t, fo);
        int tgood = 0; for (int i = 0; i < fs.klCap(); i++) tgood += tail.lines3d[i].good;
        int tvalid = 0; for (const hvo_plane_cloud &c : tail.plane_clouds) tvalid += c.valid;
        printf("l3d %d vp %d %d %d %d best %d clouds %d cloudpts %zu normals %zu stream %d %016llx\n", good, vc[0], vc[1], vc[2], vc[3], vp.best, valid, xyz.size() / 3,
               sn.size(), fo.n_kp, (unsigned long long)fnv(sdesc.data(), (size_t)fo.n_kp * 32));
        printf("tail l3d %d vpbest %d clouds %d cloudpts %d normals %d ptitems %d lnitems %d\n", tgood, tail.vp.best, tvalid, tail.c.n_cloud, tail.c.n_normals, tail.c.n_pt_items, tail.c.n_ln_items);
        // round 5: the line tracker's own calls (LSDmatcher::SearchByGeomNApearance, SearchByProjection(Cur, Last, th)): the frame against itself
        {
            std::vector<int> m12g; std::vector<uint8_t> acc;
            const int ng = lm.SearchByGeomNApearance(ldesc.data(), kls.data(), nullptr, (int)kls.size(), ldesc.data(), kls.data(), (int)kls.size(), 0.9f, b, m12g, acc);
            std::vector<float> q; for (const hvo::KeyLine &k : kls) { q.push_back(k.sx + 1.5f); q.push_back(k.sy - 0.5f); q.push_back(k.ex + 1.5f); q.push_back(k.ey - 0.5f); }
            std::vector<uint8_t> blocks(kls.size(), 1), occ(kls.size(), 0);
            std::vector<int32_t> mi;
            const int ns = lm.SearchByProjection((int)kls.size(), q.data(), kls.data(), ldesc.data(), blocks.data(), kls.data(), fn.data(), ldesc.data(), occ.data(), (int)kls.size(),
                                                 lstart.data(), litems.data(), b, 15.f, mi);
            printf("linetrack geom %d %016llx sbp %d %016llx\n", ng, (unsigned long long)fnv(m12g.data(), m12g.size() * 4), ns, (unsigned long long)fnv(mi.data(), mi.size() * 4));
        }
    } catch (const hvo::Error &e) { fprintf(stderr, "hvo error: %s\n", e.what()); return 1; }
    return 0;
}
