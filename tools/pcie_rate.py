#!/usr/bin/env python3
"""PCIe-inclusive rate of the batch boundary (host buffers in, host results out).

    python tools/pcie_rate.py [--batch 1024] [--iters 3] > profiles/rNN_pcie_note.txt

Times hvo_batch_upload + hvo_batch_run + hvo_batch_download (== hvo_extract_batch) with pageable host
buffers, i.e. what a caller that hands over cv::Mat-like host images pays.  This is NOT bench.py's `value`
(which is measured with the batch already resident in HBM); DESIGN.md section 5 quotes it beside it.
"""
import argparse
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    hvo = ge.package()
    synth = importlib.import_module("hvo_amd.synth")
    nd = 16
    g, d = synth.make_batch("std", 0x5EED1000, nd, 640, 480)
    reps = max(1, args.batch // nd)
    B = reps * nd
    ctx = hvo.Context(max_batch=B)
    ctx.batch_upload(g, d, repeat=reps); ctx.batch_run(); ctx.batch_download()     # warm-up
    t_up = t_run = t_dn = 0.0
    for _ in range(args.iters):
        t0 = time.perf_counter(); ctx.batch_upload(g, d, repeat=reps)
        t1 = time.perf_counter(); ctx.batch_run()
        t2 = time.perf_counter(); ctx.batch_download()
        t3 = time.perf_counter()
        t_up += t1 - t0; t_run += t2 - t1; t_dn += t3 - t2
    n = B * args.iters
    tot = t_up + t_run + t_dn
    in_b = 640 * 480 * 3
    out_b = 640 * 480 * 4 + 1128 * 60 + 200 * 124 + 64 * 64
    print("PCIe-inclusive rate of the batch boundary, %d frames per batch, %d iterations, pageable host memory" % (B, args.iters))
    print("(python binding: the download leg includes allocating the numpy result arrays)")
    print("  upload   (u8 gray + u16 depth, %.2f MB/frame): %8.1f frames/s  %.2f GB/s" % (in_b / 1e6, n / t_up, n * in_b / t_up / 1e9))
    print("  run      (resident, what bench.py times)      : %8.1f frames/s" % (n / t_run))
    print("  download (labels i32 + kp/desc/lines, %.2f MB/frame): %8.1f frames/s  %.2f GB/s" % (out_b / 1e6, n / t_dn, n * out_b / t_dn / 1e9))
    print("  upload + run + download, not overlapped       : %8.1f frames/s" % (n / tot))
    ctx.close()


if __name__ == "__main__":
    main()
