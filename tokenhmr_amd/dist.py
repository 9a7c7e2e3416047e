"""Data-parallel inference across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has no collective on this path (inference is single-device, eval.py:52-54); crops are
independent units (no BatchNorm at inference), so the path shards with NO data-path collective:
rank r takes crops [r*B/N, (r+1)*B/N).  Two collectives exist around it:
  * start-up: ONE broadcast of the packed weight arena (≈2.6 GB fp32) from rank 0, so only rank 0
    reads the checkpoint;
  * per batch (optional): ONE all-gather of a packed per-crop record (≈85 KB/crop: verts, joints,
    kp2d, rotmats, betas, cam, cam_t, token indices) instead of one collective per tensor — the
    gather is latency-bound on xGMI, so a single fused message is the right shape.
Works with backend 'nccl' (= RCCL on ROCm) on GPUs and 'gloo' on CPU tensors (tests).
"""
import torch
import torch.distributed as dist

# packed per-crop record layout (float32 words); token_idx is stored bit-exactly via view(int32)
RECORD_FIELDS = [
    ("pred_vertices", 6890 * 3), ("pred_keypoints_3d", 44 * 3), ("pred_keypoints_2d", 44 * 2),
    ("rotmat", 24 * 9), ("betas", 10), ("pred_cam", 3), ("pred_cam_t", 3), ("token_idx", 160),
]
RECORD_WORDS = sum(n for _, n in RECORD_FIELDS)


def shard_range(total: int, world: int, rank: int):
    """Contiguous, balanced split: the first (total % world) ranks get one extra crop."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_sizes(total: int, world: int):
    return [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]


def pack_records(o):
    """dict of per-crop tensors (engine output) -> (B, RECORD_WORDS) float32.  On the GPU this is ONE launch of the library's
    pack kernel (thmr_pack_records, the same call a non-torch host makes); CPU tensors (gloo tests) are concatenated."""
    B = o["pred_cam"].shape[0]
    if o["pred_cam"].is_cuda:
        import ctypes as C
        from . import _cabi
        lib = _cabi.load()
        dev = o["pred_cam"].device
        ts = {}
        for name, n in RECORD_FIELDS:
            # thmr_pack_records reads raw 32-bit words: a forward_fn that hands back e.g. an int64 token_idx or a tensor on another
            # device must not be packed as garbage
            want = torch.int32 if name == "token_idx" else torch.float32
            t = o[name]
            if t.device != dev:
                raise ValueError(f"pack_records: '{name}' lives on {t.device}, 'pred_cam' on {dev}")
            if t.dtype != want:
                if name == "token_idx" and t.dtype in (torch.int64, torch.int16, torch.uint8, torch.int8):
                    t = t.to(torch.int32)
                elif name != "token_idx" and t.is_floating_point():
                    t = t.to(torch.float32)
                else:
                    raise TypeError(f"pack_records: '{name}' must be {want}, got {t.dtype}")
            if t.numel() != B * n:
                raise ValueError(f"pack_records: '{name}' has {t.numel()} elements, expected {B} x {n}")
            ts[name] = t.contiguous()
        st = _cabi.Outputs(**{k: (ts[k].data_ptr() if k in ts else None) for k in _cabi.OUTPUT_FIELDS})
        rec = torch.empty(B, RECORD_WORDS, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.thmr_pack_records(C.byref(st), B, C.c_void_p(rec.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise _cabi.EngineError(f"thmr_pack_records: {lib.thmr_collective_last_error().decode()}")
        return rec
    parts = []
    for name, n in RECORD_FIELDS:
        t = o[name]
        if name == "token_idx":
            t = t.to(torch.int32).contiguous().view(torch.float32)
        elif t.dtype != torch.float32:
            t = t.to(torch.float32)
        parts.append(t.reshape(B, n))
    return torch.cat(parts, dim=1).contiguous()


def unpack_records(rec):
    B = rec.shape[0]
    shapes = {"pred_vertices": (6890, 3), "pred_keypoints_3d": (44, 3), "pred_keypoints_2d": (44, 2),
              "rotmat": (24, 3, 3), "betas": (10,), "pred_cam": (3,), "pred_cam_t": (3,), "token_idx": (160,)}
    out, off = {}, 0
    for name, n in RECORD_FIELDS:
        t = rec[:, off:off + n]
        off += n
        if name == "token_idx":
            t = t.contiguous().view(torch.int32)
        out[name] = t.reshape(B, *shapes[name])
    return out


def _gloo_with_gpu_tensor(t) -> bool:
    """The gloo backend moves host memory; GPU tensors are staged through the host here (RCCL absent, or — what this exists for
    in this repo — two ranks on ONE GPU: RCCL refuses duplicate devices, gloo does not, so the N > 1 code path can run with real
    engines on a one-GPU box: bench.py --single-device, tests/test_bench_cli.py)."""
    return dist.is_initialized() and dist.get_backend() == "gloo" and t.is_cuda


def broadcast_weights(engine, src: int = 0):
    """ONE collective for the whole model: rank `src` has loaded the checkpoint; everyone else
    receives the packed arena and only has to finalize."""
    if not dist.is_initialized():
        return
    arena = engine.weight_arena
    if not _gloo_with_gpu_tensor(arena):
        dist.broadcast(arena, src=src)
        return
    chunk = 256 << 20                                        # bound the host staging buffer
    flat = arena.view(-1)
    for off in range(0, flat.numel(), chunk):
        part = flat[off:off + chunk]
        host = part.cpu() if dist.get_rank() == src else torch.empty(part.numel(), dtype=part.dtype)
        dist.broadcast(host, src=src)
        if dist.get_rank() != src:
            part.copy_(host)


class GatherHandle:
    """An all-gather in flight on the process group's own (RCCL) stream.  wait() makes the CURRENT stream wait for it (no host
    block) and returns the (total, RECORD_WORDS) records in crop order."""

    def __init__(self, work, buf, sizes, mx, device=None):
        self.work, self.buf, self.sizes, self.mx, self.device = work, buf, sizes, mx, device

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.device is not None and self.buf.device != self.device:      # gloo with GPU records: gathered on the host
            self.buf = self.buf.to(self.device)
        if all(s == self.mx for s in self.sizes):
            return self.buf                                  # equal shards: the gathered buffer IS the result (no copy)
        return torch.cat([self.buf[r * self.mx: r * self.mx + s] for r, s in enumerate(self.sizes)], dim=0)


def all_gather_records(local_rec, total: int, async_op: bool = False):
    """All-gather the packed records of every rank's shard; returns (total, RECORD_WORDS) in crop order.
    Shards may differ by one crop, so each rank pads to the maximum shard size.  async_op=True returns a GatherHandle
    instead: the collective then runs on the RCCL stream while the caller's stream goes on with the next batch (xGMI traffic
    under the next ViT), and is joined with handle.wait()."""
    if not dist.is_initialized():
        return GatherHandle(None, local_rec, [local_rec.shape[0]], local_rec.shape[0]) if async_op else local_rec
    world = dist.get_world_size()
    sizes = shard_sizes(total, world)
    mx = max(sizes)
    pad = local_rec
    if local_rec.shape[0] < mx:
        pad = torch.zeros(mx, local_rec.shape[1], dtype=local_rec.dtype, device=local_rec.device)
        pad[: local_rec.shape[0]] = local_rec
    dev = local_rec.device
    if _gloo_with_gpu_tensor(local_rec):
        pad = pad.cpu()                                      # staged through the host (see _gloo_with_gpu_tensor)
    buf = torch.empty(world * mx, local_rec.shape[1], dtype=local_rec.dtype, device=pad.device)
    work = dist.all_gather_into_tensor(buf, pad.contiguous(), async_op=True)
    h = GatherHandle(work, buf, sizes, mx, device=dev)
    return h if async_op else h.wait()


class ShardedRunner:
    """Runs `forward_fn(img_shard) -> engine-output dict` on this rank's shard of a global batch and
    (optionally) all-gathers the packed records so every rank sees all crops."""

    def __init__(self, forward_fn, gather: bool = True, device=None):
        """device: where this rank's records live (the engine's device).  Default: the current CUDA device under the nccl
        backend; under gloo the device of the first records this rank packs.  Pass it explicitly under gloo with GPU engines when a
        rank's FIRST shard can be empty and the batch is a host tensor — otherwise that one call returns host tensors on this rank."""
        self.forward_fn = forward_fn
        self.gather = gather
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if device is None and dist.is_initialized() and dist.get_backend() == "nccl":
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device) if device is not None else None

    def local_slice(self, total):
        return shard_range(total, self.world, self.rank)

    def __call__(self, img_global):
        total = img_global.shape[0]
        s, e = self.local_slice(total)
        if e > s:
            rec = pack_records(self.forward_fn(img_global[s:e]))
            if self.device is None:
                self.device = rec.device            # remembered: a later EMPTY shard of this rank puts its zero rows on the same device
        else:
            # fewer crops than ranks (e.g. 3 detections in a frame on 8 GPUs): this rank has nothing to run — the engine
            # rejects B < 1 — but it must still enter the collective, with a zero-row record block
            # (on the device the OTHER ranks' records live on — the batch itself may still be a host tensor that forward_fn uploads)
            rec = torch.zeros(0, RECORD_WORDS, dtype=torch.float32, device=self.device if self.device is not None else img_global.device)
        if not self.gather:
            return unpack_records(rec)
        return unpack_records(all_gather_records(rec, total))


def merge_evaluator(ev, total=None):
    """Sharded evaluation (BASELINE config 5: eval.py over the 8 GPUs of a node).  Every rank has run the reference's
    evaluator protocol (pose_utils.py:201-275: per-sample metric arrays filled up to `counter`, `imgnames`) over ITS
    contiguous shard shard_range(total, world, rank) of the dataset; after this call every rank's evaluator holds the metric
    arrays and image names of ALL samples in dataset order, so `log()` / `get_metrics_dict()` (eval.py:154-158) report the
    whole-dataset means — identical to a one-process run because only per-sample values travel, never partial means.
    Two small collectives: the shard lengths and one padded (samples, metrics) float64 gather."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        if total is not None and ev.counter != total:
            raise ValueError(f"evaluator saw {ev.counter} samples, expected {total}")
        return ev
    import numpy as np
    world = dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    cnt = torch.tensor([ev.counter], dtype=torch.int64, device=dev)
    cnts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(cnts, cnt)
    counts = [int(c) for c in cnts.cpu()]
    if total is not None and sum(counts) != total:
        raise ValueError(f"ranks evaluated {counts} samples, expected {total} in all")
    mx, nm = max(counts), len(ev.metrics)
    local = torch.zeros(mx, nm, dtype=torch.float64)
    if ev.counter:
        local[: ev.counter] = torch.from_numpy(np.stack([getattr(ev, m)[: ev.counter] for m in ev.metrics], 1))
    buf = torch.empty(world * mx, nm, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(buf, local.to(dev))
    buf = buf.cpu().numpy()
    merged = np.concatenate([buf[r * mx: r * mx + c] for r, c in enumerate(counts)], 0)
    names = [None] * world
    dist.all_gather_object(names, list(ev.imgnames))
    n = merged.shape[0]
    for i, m in enumerate(ev.metrics):
        arr = getattr(ev, m)
        if arr.shape[0] < n:
            arr = np.zeros((n,))
            setattr(ev, m, arr)
        arr[:n] = merged[:, i]
    ev.counter = n
    ev.imgnames = [x for part in names for x in part]
    return ev
