"""CPU (-m "not gpu"): the N>1 path with world_size 2 on the gloo backend.

The data path has no collective (crops are independent); what is tested is exactly the code that runs
under RCCL on GPUs: weight-arena broadcast, contiguous sharding, packed-record all-gather with ragged
shards — and that the N-rank result is bit-identical to the 1-rank result."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tokenhmr_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_forward(img):
    """Deterministic per-crop 'engine': every output is a pure function of the crop alone.  Like the real engine
    (check_ready, csrc/engine.hip) it refuses an empty batch."""
    B = img.shape[0]
    if B < 1:
        raise ValueError("batch 0 outside [1, max_batch]")
    key = img.reshape(B, -1)[:, :8].sum(dim=1)
    def f(n, k):
        return (key[:, None] * (torch.arange(n, dtype=torch.float32)[None] + k)).contiguous()
    return {"pred_vertices": f(6890 * 3, 1).reshape(B, 6890, 3), "pred_keypoints_3d": f(132, 2).reshape(B, 44, 3),
            "pred_keypoints_2d": f(88, 3).reshape(B, 44, 2), "rotmat": f(216, 4).reshape(B, 24, 3, 3),
            "betas": f(10, 5), "pred_cam": f(3, 6), "pred_cam_t": f(3, 7),
            "token_idx": (key[:, None].abs() * 100 + torch.arange(160)[None]).to(torch.int32) % 2048}


class _FakeEngine:
    def __init__(self, rank):
        self.weight_arena = torch.full((1024,), 7 if rank == 0 else 0, dtype=torch.uint8)


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = _FakeEngine(rank)
        D.broadcast_weights(eng, src=0)
        assert int(eng.weight_arena.sum()) == 7 * 1024
        g = torch.Generator().manual_seed(123)
        img = torch.randn(total, 3, 4, 4, generator=g)          # same global batch on every rank
        runner = D.ShardedRunner(_fake_forward, gather=True)
        s, e = runner.local_slice(total)
        out = runner(img)
        ref = D.unpack_records(D.pack_records(_fake_forward(img)))
        ok = all(torch.equal(out[k], ref[k]) for k in ref)
        # pipelined form used by bench.py: two gathers issued back to back, joined later, in order
        def rec(x):     # a rank whose shard is empty contributes a zero-row block (what ShardedRunner does)
            return D.pack_records(_fake_forward(x)) if x.shape[0] else torch.zeros(0, D.RECORD_WORDS)
        h1 = D.all_gather_records(rec(img[s:e]), total, async_op=True)
        h2 = D.all_gather_records(rec(2.0 * img[s:e]), total, async_op=True)
        o1, o2 = D.unpack_records(h1.wait()), D.unpack_records(h2.wait())
        ref2 = D.unpack_records(D.pack_records(_fake_forward(2.0 * img)))
        ok = ok and all(torch.equal(o1[k], ref[k]) for k in ref) and all(torch.equal(o2[k], ref2[k]) for k in ref2)
        q.put((rank, (s, e), ok, int(out["pred_cam"].shape[0])))
    finally:
        dist.destroy_process_group()


def _run(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_two_ranks_even_split():
    res = _run(8)
    assert [r[1] for r in res] == [(0, 4), (4, 8)]
    assert all(r[2] for r in res) and all(r[3] == 8 for r in res)


def test_two_ranks_ragged_split():
    res = _run(7)       # shards of 4 and 3 crops: padded all-gather must drop the pad row
    assert [r[1] for r in res] == [(0, 4), (4, 7)]
    assert all(r[2] for r in res) and all(r[3] == 7 for r in res)


def test_fewer_crops_than_ranks():
    """total < world (one detection in a frame, two ranks): rank 1's shard is empty.  It must skip the forward (the engine
    rejects B = 0) yet still enter the all-gather, or rank 0 hangs until the collective times out."""
    res = _run(1)
    assert [r[1] for r in res] == [(0, 1), (1, 1)]
    assert all(r[2] for r in res) and all(r[3] == 1 for r in res)


def test_single_process_is_identity():
    g = torch.Generator().manual_seed(1)
    img = torch.randn(5, 3, 4, 4, generator=g)
    out = D.ShardedRunner(_fake_forward)(img)
    ref = _fake_forward(img)
    assert all(torch.equal(out[k], ref[k]) for k in ref)


class _StubEvaluator:
    """The state the reference's Evaluator keeps (pose_utils.py:160-175): metric arrays, a fill counter, image names."""
    metrics = ["mode_re", "mode_mpjpe", "mode_pve"]

    def __init__(self, n=16):
        import numpy as np
        for m in self.metrics:
            setattr(self, m, np.zeros((n,)))
        self.counter, self.imgnames = 0, []

    def feed(self, idx):
        for i in idx:
            for k, m in enumerate(self.metrics):
                getattr(self, m)[self.counter] = 10.0 * (k + 1) + 0.37 * i * (k + 1)
            self.imgnames.append(f"img{i}")
            self.counter += 1


def _eval_worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ev = _StubEvaluator(n=8)                       # smaller than the merged length: merge must grow the arrays
        s, e = D.shard_range(total, world, rank)
        ev.feed(range(s, e))
        D.merge_evaluator(ev, total)
        q.put((rank, ev.counter, [float(x) for x in ev.mode_mpjpe[:ev.counter]], list(ev.imgnames)))
        try:
            D.merge_evaluator(ev, total + 1)
            q.put((rank, "no error"))
        except ValueError:
            q.put((rank, "raised"))
    finally:
        dist.destroy_process_group()


def test_sharded_evaluators_merge_to_the_single_process_result():
    total = 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_eval_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in range(4)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    one = _StubEvaluator(n=16)
    one.feed(range(total))
    assert D.merge_evaluator(one, total) is one                     # no process group: identity
    res = [g for g in got if len(g) == 4]
    assert len(res) == 2 and sorted(g[1] for g in got if len(g) == 2) == ["raised", "raised"]
    for _, counter, mpjpe, names in res:
        assert counter == total and names == one.imgnames
        assert mpjpe == [float(x) for x in one.mode_mpjpe[:total]]


class _ToyDataset(torch.utils.data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {"img": torch.full((3, 2, 2), float(i)), "gt": torch.tensor([float(i) * 0.5]), "imgname": f"img{i}"}


class _ToyEvaluator(_StubEvaluator):
    """Reference protocol (pose_utils.py:201-275): called per batch, appends per-sample metrics."""

    def __call__(self, out, batch):
        err = (out["pred"] - batch["gt"]).abs().reshape(-1)
        for j in range(err.shape[0]):
            for k, m in enumerate(self.metrics):
                getattr(self, m)[self.counter] = float(err[j]) * (k + 1)
            self.imgnames.append(batch["imgname"][j])
            self.counter += 1

    def log(self):
        pass

    def get_metrics_dict(self):
        return {m: float(getattr(self, m)[:self.counter].mean()) for m in self.metrics}


def _toy_model(batch):
    return {"pred": batch["img"][:, 0, 0, :1] * 0.75}         # per-crop function of the crop alone


def _run_eval_worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tokenhmr_amd.eval_dp import run_eval
        ev = _ToyEvaluator(n=total + 4)
        res = run_eval(_toy_model, _ToyDataset(total), ev, batch_size=4, device="cpu")
        q.put((rank, res, ev.counter, list(ev.imgnames)))
    finally:
        dist.destroy_process_group()


def test_run_eval_sharded_over_two_ranks_equals_one_process():
    """tokenhmr_amd.eval_dp.run_eval (eval.py:116-158 as a data-parallel job): 2 ranks x contiguous shards -> every rank reports
    the metrics and image order a single process computes."""
    from tokenhmr_amd.eval_dp import run_eval
    total = 13
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_run_eval_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    one = _ToyEvaluator(n=total)
    ref = run_eval(_toy_model, _ToyDataset(total), one, batch_size=4, device="cpu")
    for _, res, counter, names in got:
        assert counter == total and names == one.imgnames == [f"img{i}" for i in range(total)]
        assert res.keys() == ref.keys() and all(abs(res[k] - ref[k]) < 1e-12 for k in ref)


def test_pack_records_casts_and_checks_dtypes():
    """a forward_fn that hands back an int64 token_idx or float64 joints must not be packed as garbage words: the CPU
    path casts (token indices bit-exactly through int32), and the packed block round-trips."""
    o = _fake_forward(torch.randn(3, 3, 4, 4, generator=torch.Generator().manual_seed(1)))
    ref = D.pack_records(o)
    o2 = dict(o)
    o2["token_idx"] = o["token_idx"].to(torch.int64)
    o2["pred_keypoints_3d"] = o["pred_keypoints_3d"].double()
    got = D.pack_records(o2)
    assert got.dtype == torch.float32 and got.shape == ref.shape and torch.equal(got.view(torch.int32), ref.view(torch.int32))
    back = D.unpack_records(got)
    assert back["token_idx"].dtype == torch.int32 and torch.equal(back["token_idx"], o["token_idx"])


def _worker_empty_shard_device(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seen = {}

        def fwd(x):
            seen["dev"] = x.device
            return _fake_forward(x)
        # 1 crop on 2 ranks: rank 1's shard is empty.  The runner is told where this rank's records live (device=...): the
        # zero-row block must be created THERE, not wherever the (possibly host-side) batch lives.
        runner = D.ShardedRunner(fwd, gather=True, device="cpu")
        img = torch.randn(1, 3, 4, 4, generator=torch.Generator().manual_seed(5))
        out = runner(img)
        ref = D.unpack_records(D.pack_records(_fake_forward(img)))
        q.put((rank, all(torch.equal(out[k], ref[k]) for k in ref), str(runner.device), "dev" in seen))
    finally:
        dist.destroy_process_group()


def test_empty_shard_record_lives_on_the_runner_device():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_empty_shard_device, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == [(0, True, "cpu", True), (1, True, "cpu", False)], res
