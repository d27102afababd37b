"""N>1 path on CPU: frame sharding + result-slab gather with torch.distributed (gloo, world_size 2).
There is no GPU here, so each rank fills its shard's results with the CPU oracle (tests may use
the oracle); what is under test is the partition and the gather plumbing bench.py / a multi-GPU
caller uses."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, load_pkg, load_oracle, load_synth


def test_shard_range_partitions_exactly():
    pkg = load_pkg()
    import importlib
    dist = importlib.import_module("hvo_amd.dist")
    for n in (0, 1, 7, 8, 255, 256, 257):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                lo, hi = dist.shard_range(n, world, r)
                assert 0 <= lo <= hi <= n
                covered += list(range(lo, hi))
            assert covered == list(range(n))
            sizes = [dist.shard_range(n, world, r)[1] - dist.shard_range(n, world, r)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _oracle_results(orc, gray, depth):
    res = []
    o = orc.Orb()
    for g, d in zip(gray, depth):
        kp, desc = o.extract(g)
        kl, ld, fn = orc.line_extract(g)
        lab, pl = orc.peac(d)
        res.append({"kp": kp, "desc": desc, "kl": kl, "ldesc": ld, "linefn": fn, "planes": pl, "labels": lab, "status": 0})
    return res


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_pkg(); orc = load_oracle(); synth = load_synth()
    hd = importlib.import_module("hvo_amd.dist")
    gray, depth = synth.make_batch("std", 0x5EED2000, n_frames, 320, 240)     # small frames: CPU test
    lo, hi = hd.shard_range(n_frames, world, rank)
    local = _oracle_results(orc, gray[lo:hi], depth[lo:hi])
    allres = hd.gather_results(pkg, local, n_frames, kp_cap=1100, kl_cap=200)
    withlab = hd.gather_results(pkg, local, n_frames, kp_cap=1100, kl_cap=200, label_shape=(240, 320))      # the slab that also carries membershipImg
    if rank == 0:
        ref = _oracle_results(orc, gray, depth)
        ok = len(allres) == n_frames and len(withlab) == n_frames
        for a, a2, b in zip(allres, withlab, ref):
            for k in ("kp", "desc", "kl", "ldesc", "linefn", "planes"):
                ok = ok and np.array_equal(a[k], b[k]) and np.array_equal(a2[k], b[k])
            ok = ok and "labels" not in a and np.array_equal(a2["labels"], b["labels"]) and (b["labels"] >= 0).any()
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _worker_ragged(rank, world, port, n_frames, q):
    """the DEVICE path's plumbing on CPU tensors: every rank packs its shard's slabs, pads to the largest shard, one all_gather of equal
    pieces (gather_padded), rank 0 unpacks in global frame order (unpack_gathered)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_pkg(); orc = load_oracle(); synth = load_synth()
    hd = importlib.import_module("hvo_amd.dist")
    gray, depth = synth.make_batch("std", 0x5EED2100, n_frames, 320, 240)
    lo, hi = hd.shard_range(n_frames, world, rank)
    local = _oracle_results(orc, gray[lo:hi], depth[lo:hi])
    t = torch.from_numpy(hd.pack_results(pkg, local, 1100, 200, 64, label_shape=(240, 320)))
    out = hd.gather_padded(t, hd.max_shard(n_frames, world), reduce_device="cpu")
    if rank == 0:
        allres = hd.unpack_gathered(pkg, out, n_frames, 1100, 200, 64, label_shape=(240, 320))
        ref = _oracle_results(orc, gray, depth)
        ok = len(allres) == n_frames and tuple(out.shape[:2]) == (world, hd.max_shard(n_frames, world))
        for a, b in zip(allres, ref):
            for k in ("kp", "desc", "kl", "ldesc", "linefn", "planes", "labels"):
                ok = ok and np.array_equal(a[k], b[k])
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_padded_world3_ragged_gloo():
    """7 frames over 3 ranks (3 + 2 + 2): the padded all_gather of the device path, on CPU tensors"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_ragged, args=(r, 3, port, 7, q)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
