"""Frame sharding across ranks (one process per GPU) and the result gather.

The hot path has no exchange step: frames are independent units (the reference builds one Frame
at a time, src/Tracking.cc:262), so a batch is split into contiguous blocks, every rank runs the
whole front-end on its block, and nothing crosses GPUs while computing.  The only collective is an
optional all_gather of fixed-size result slabs (RCCL over xGMI with backend "nccl", gloo on CPU):
93.7 KB per frame of records; with `labels=True` the slab also carries the frame's int8 label image (membershipImg; 307 200 B at
640x480) -- plane labels are one of the three things the consumer wants bit-exact, and the slab is the one thing that crosses GPUs.
"""
import numpy as np

HDR = 16  # n_kp, n_kl, n_planes, status (int32 each)


def shard_range(n_frames, world, rank):
    """contiguous block of frames for `rank`: the first n % world ranks get one extra frame"""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def slab_layout(pkg, kp_cap, kl_cap, pl_cap, label_shape=None):
    """byte layout of one frame's slab (= hvo_batch_slab_layout_ex); label_shape = (h, w): the int8 label image at the end (HVO_SLAB_LABELS)"""
    kpb = kp_cap * pkg.KEYPOINT_DT.itemsize; db = kp_cap * 32
    klb = kl_cap * pkg.KEYLINE_DT.itemsize; ldb = kl_cap * 32; fnb = kl_cap * 24
    plb = pl_cap * pkg.PLANE_DT.itemsize
    offs = np.cumsum([HDR, kpb, db, klb, ldb, fnb, plb])
    L = dict(kp=(HDR, kpb), desc=(offs[1], db), kl=(offs[2], klb), ldesc=(offs[3], ldb), linefn=(offs[4], fnb),
             planes=(offs[5], plb), size=int(offs[6]))
    if label_shape is not None:
        lb = int(label_shape[0]) * int(label_shape[1])
        L["labels"] = (L["size"], lb); L["label_shape"] = (int(label_shape[0]), int(label_shape[1])); L["size"] += (lb + 15) & ~15
    return L


def pack_results(pkg, results, kp_cap, kl_cap, pl_cap=64, label_shape=None):
    """list of per-frame result dicts (Context.batch_download) -> uint8 array [n, slab_size]"""
    L = slab_layout(pkg, kp_cap, kl_cap, pl_cap, label_shape)
    out = np.zeros((len(results), L["size"]), np.uint8)
    if label_shape is not None:
        o, lb = L["labels"]
        for i, r in enumerate(results):
            out[i, o:o + lb] = 0xFF if "labels" not in r else np.ascontiguousarray(r["labels"]).astype(np.int8).view(np.uint8).reshape(-1)
    for i, r in enumerate(results):
        hdr = np.array([len(r.get("kp", ())), len(r.get("kl", ())), len(r.get("planes", ())), r.get("status", 0)], np.int32)
        out[i, :HDR] = hdr.view(np.uint8)
        for key in ("kp", "desc", "kl", "ldesc", "linefn", "planes"):
            if key in r and len(r[key]):
                b = np.ascontiguousarray(r[key]).view(np.uint8).reshape(-1)
                o, cap = L[key]
                assert len(b) <= cap, key
                out[i, o:o + len(b)] = b
    return out


def unpack_results(pkg, slabs, kp_cap, kl_cap, pl_cap=64, label_shape=None):
    L = slab_layout(pkg, kp_cap, kl_cap, pl_cap, label_shape)
    res = []
    for row in slabs:
        nkp, nkl, npl, status = row[:HDR].view(np.int32)
        r = {"status": int(status)}
        o, _ = L["kp"]; r["kp"] = row[o:o + nkp * 28].view(pkg.KEYPOINT_DT).copy()
        o, _ = L["desc"]; r["desc"] = row[o:o + nkp * 32].reshape(nkp, 32).copy()
        o, _ = L["kl"]; r["kl"] = row[o:o + nkl * 68].view(pkg.KEYLINE_DT).copy()
        o, _ = L["ldesc"]; r["ldesc"] = row[o:o + nkl * 32].reshape(nkl, 32).copy()
        o, _ = L["linefn"]; r["linefn"] = row[o:o + nkl * 24].view(np.float64).reshape(nkl, 3).copy()
        o, _ = L["planes"]; r["planes"] = row[o:o + npl * 64].view(pkg.PLANE_DT).copy()
        if label_shape is not None:
            o, lb = L["labels"]; r["labels"] = row[o:o + lb].view(np.int8).reshape(L["label_shape"]).astype(np.int32)
        res.append(r)
    return res


def device_slabs(ctx, n, labels=False):
    """the first n frames' result slabs of ctx's resident batch as a CUDA uint8 tensor [n, slab_bytes]: packed on the device by
    hvo_batch_pack_results_ex (no host round trip); the layout is slab_layout(pkg, *ctx.slab_layout()[:3], label_shape)"""
    from . import torch_order_check
    torch_order_check()
    import torch
    sb = ctx.slab_layout(labels)[3]
    t = torch.empty((n, sb), dtype=torch.uint8, device="cuda")
    ctx.pack_results(n, t.data_ptr(), labels)
    return t


def max_shard(n_frames, world):
    return max(shard_range(n_frames, world, r)[1] - shard_range(n_frames, world, r)[0] for r in range(world))


def gather_padded(t, per, reduce_device="cuda"):
    """all_gather of every rank's [n_r, slab_bytes] uint8 tensor where the n_r may differ (257 frames over 8 ranks: 33 + 7 x 32): every
    rank pads to `per` rows -- zero slabs, whose header says that nothing arrived -- so that ONE all_gather_into_tensor of equal pieces
    serves ragged shards on the device path too (round 5; before, only the host path gather_results took them).  -> [world, per, slab_bytes]"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if t.shape[0] != per:
        pad = torch.zeros((per, t.shape[1]), dtype=torch.uint8, device=t.device)
        pad[: t.shape[0]] = t
        t = pad
    if reduce_device == "cuda":
        out = torch.empty((world,) + tuple(t.shape), dtype=torch.uint8, device="cuda")
        dist.all_gather_into_tensor(out, t)
        torch.cuda.synchronize()
    else:
        tc = t.cpu()
        outs = [torch.empty_like(tc) for _ in range(world)]
        dist.all_gather(outs, tc)
        out = torch.stack(outs)
    return out


def unpack_gathered(pkg, out, n_frames, kp_cap, kl_cap, pl_cap=64, label_shape=None):
    """[world, per, slab_bytes] as gather_padded returns it -> the results of all n_frames frames in global frame order"""
    world = out.shape[0]
    arr = out.cpu().numpy()
    res = []
    for r in range(world):
        lo, hi = shard_range(n_frames, world, r)
        res += unpack_results(pkg, arr[r][: hi - lo], kp_cap, kl_cap, pl_cap, label_shape)
    return res


def gather_device_slabs(ctx, n, reduce_device="cuda", labels=False, n_frames=None, want_tensor=False):
    """the path's one collective: all_gather of every rank's n result slabs.  With the nccl (= RCCL) backend the slabs go
    from HBM to HBM over xGMI (all_gather_into_tensor on the packed tensor); with gloo (CPU rehearsal) they are staged
    through the host.  n_frames: the batch's total when the ranks' shards differ in size (shard_range): every rank pads to the largest.
    Returns (ranks whose slabs arrived with results in them, slab bytes per frame[, the gathered tensor])."""
    import torch.distributed as dist
    world = dist.get_world_size()
    t = device_slabs(ctx, n, labels)
    per = n if n_frames is None else max_shard(n_frames, world)
    out = gather_padded(t, per, reduce_device)
    hdr = out[:, 0, :HDR].cpu().numpy().view(np.int32).reshape(world, 4)
    seen = int(((hdr[:, 0] > 0) | (hdr[:, 1] > 0) | (hdr[:, 2] > 0)).sum())
    return (seen, int(t.shape[1]), out) if want_tensor else (seen, int(t.shape[1]))


def gather_results(pkg, local_results, n_frames, kp_cap, kl_cap, pl_cap=64, device="cpu", label_shape=None):
    """all_gather the per-rank slabs; returns the results of all frames in global frame order.
    Blocks may differ by one frame, so every rank pads to the largest block."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    L = slab_layout(pkg, kp_cap, kl_cap, pl_cap, label_shape)
    per = max(shard_range(n_frames, world, r)[1] - shard_range(n_frames, world, r)[0] for r in range(world))
    mine = np.zeros((per, L["size"]), np.uint8)
    packed = pack_results(pkg, local_results, kp_cap, kl_cap, pl_cap, label_shape)
    mine[: len(packed)] = packed
    t = torch.from_numpy(mine).to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    res = []
    for r in range(world):
        lo, hi = shard_range(n_frames, world, r)
        res += unpack_results(pkg, outs[r].cpu().numpy()[: hi - lo], kp_cap, kl_cap, pl_cap, label_shape)
    return res
