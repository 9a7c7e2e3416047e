"""bench.py's launch contract.  CPU (-m "not gpu"): `python bench.py --gpus 2` with no launcher re-executes itself under
torch.distributed.run with 2 ranks; `--backend gloo --fake-engine` runs exactly the N-rank orchestration (rank 0 loads,
ONE broadcast of the arena, per-step async packed all-gather joined one step later, barrier-bracketed timing, MAX over
ranks, one JSON line from rank 0) on CPU tensors.  GPU (-m gpu): the same self-launch path on the nccl (= RCCL) backend at
world size 1, and the broadcast -> finalize -> forward -> async all-gather -> unpack chain against the direct call."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=timeout, cwd=ROOT)


def _line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_gpus2_self_launches_two_gloo_ranks():
    j = _line(_run(["--gpus", "2", "--backend", "gloo", "--fake-engine", "--steps", "3", "--warmup", "1", "--batch", "4"]))
    assert j["n_gpus"] == 2 and j["config"]["ranks"] == 2 and j["config"]["backend"] == "gloo"
    assert j["config"]["global_batch"] == 8 and j["config"]["batch_per_gpu"] == 4 and j["config"]["parallelism"] == "dp2"
    assert j["config"]["allgather_outputs"] is True and j["gathered_records_ok"] is True
    assert j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak" and j["metric"] == "crops_per_sec"
    assert j["value"] > 0 and abs(j["value"] - 8 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-3
    assert "dry_run" in j
    _check_multi_gpu_block(j, 2, [4, 4])


def _check_multi_gpu_block(j, world, sizes):
    m = j["multi_gpu"]
    assert m["checkpoint_readers"] == [0], m["checkpoint_readers"]            # only rank 0 touches the "checkpoint"
    assert [d["rank"] for d in m["devices"]] == list(range(world)) and [d["local_rank"] for d in m["devices"]] == list(range(world))
    assert [r["crops"] for r in m["per_rank"]] == sizes and [r["rank"] for r in m["per_rank"]] == list(range(world))
    assert m["bcast_ms"] is not None and m["bcast_ms"] >= 0 and m["bcast_bytes"] == 4096
    assert m["gather_ms_exposed"] >= 0 and m["gather_bytes_per_step"] == sum(sizes) * 21282 * 4
    assert m["cross_rank_check"]["ranks_checked"] == world - 1 and m["cross_rank_check"]["bit_identical"] is True
    assert j["gathered_records_ok"] is True


def test_gpus8_dry_run_global_batch_512_and_ragged_509():
    """The driver's first 8-GPU run must not be a debugging session: the whole 8-rank orchestration (rank 0 loads, ONE broadcast,
    per-step async packed all-gather joined one step later, per-rank diagnostics, cross-rank check, one JSON line) runs here on
    gloo / CPU tensors, weak-scaled at 64 crops per rank (global 512, BASELINE configs[3]) and with ONE ragged global batch of
    509 crops dealt 64,64,64,64,64,63,63,63."""
    j = _line(_run(["--gpus", "8", "--backend", "gloo", "--fake-engine", "--steps", "3", "--warmup", "1"], timeout=900))
    assert j["n_gpus"] == 8 and j["config"]["ranks"] == 8 and j["config"]["global_batch"] == 512 and j["config"]["batch_per_gpu"] == 64
    assert j["scaling"] == "weak" and j["config"]["parallelism"] == "dp8"
    assert abs(j["value"] - 512 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-3
    _check_multi_gpu_block(j, 8, [64] * 8)
    j = _line(_run(["--gpus", "8", "--backend", "gloo", "--fake-engine", "--steps", "2", "--warmup", "1", "--global-batch", "509"], timeout=900))
    assert j["config"]["global_batch"] == 509 and j["config"]["batch_per_gpu"] == [64, 64, 64, 64, 64, 63, 63, 63] and j["scaling"] == "strong"
    assert abs(j["value"] - 509 * 2 / (j["ms_per_step"] * 2e-3)) / j["value"] < 1e-3
    _check_multi_gpu_block(j, 8, [64, 64, 64, 64, 64, 63, 63, 63])


def test_global_batch_smaller_than_world_is_refused():
    r = _run(["--gpus", "2", "--backend", "gloo", "--fake-engine", "--global-batch", "1"])
    assert r.returncode != 0 and "leaves a rank without crops" in r.stderr


def test_gpus_flag_must_agree_with_the_launcher():
    """Under a launcher (WORLD_SIZE set) a mismatching --gpus is a one-line error, not an AssertionError traceback."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = _run(["--gpus", "2", "--backend", "gloo", "--fake-engine"],
             {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr and "Traceback" not in r.stderr


def test_fake_engine_is_refused_on_the_gpu_backend():
    r = _run(["--fake-engine", "--backend", "nccl"])
    assert r.returncode != 0 and "dry run" in r.stderr


@pytest.mark.gpu
def test_self_launch_on_rccl_world1(built_lib, cuda_dev):
    """`--gpus 1 --self-launch`: the N-rank code path (torch.distributed.run, nccl backend, broadcast, per-step all-gather)
    with one rank, at a reduced depth so it takes seconds."""
    j = _line(_run(["--gpus", "1", "--self-launch", "--vit-depth", "2", "--steps", "4", "--warmup", "1", "--batch", "8",
                    "--no-cpu-baseline", "--no-extras"], {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}))
    assert j["n_gpus"] == 1 and j["config"]["ranks"] == 1 and j["config"]["backend"] == "nccl"
    assert j["config"]["allgather_outputs"] is True and j["gathered_records_ok"] is True
    assert j["roofline"]["frac"] > 0 and "src:" in j["build"]
    m = j["multi_gpu"]
    assert m["checkpoint_readers"] == [0] and m["bcast_ms"] > 0 and m["bcast_bytes"] > 1e8 and m["devices"][0]["device"] == 0
    assert m["gather_ms_exposed"] >= 0 and m["rank_step_ms"]["min"] > 0


@pytest.mark.gpu
def test_rccl_world1_chain_equals_direct_call(built_lib, cuda_dev):
    """broadcast -> finalize -> forward -> async packed all-gather -> unpack on the nccl backend (world size 1) is bit-for-bit
    the direct engine call."""
    import torch
    import torch.distributed as dist
    from tokenhmr_amd.config import HMRConfig
    from tokenhmr_amd import weights as W, dist as D
    from tokenhmr_amd.smpl_assets import make_synthetic_smpl
    from tokenhmr_amd.engine import Engine
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda_dev)
    try:
        cfg = HMRConfig(vit_depth=2, dec_depth=2)
        eng = Engine(cfg, max_batch=6, device=cuda_dev)
        eng.load_state(W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0))
        eng.load_smpl(make_synthetic_smpl(cfg, 0))
        torch.cuda.synchronize()
        D.broadcast_weights(eng, src=0)
        torch.cuda.synchronize()
        eng.finalize()
        img = torch.randn(6, 3, 256, 256, generator=torch.Generator().manual_seed(5)).to(cuda_dev)
        direct = {k: v.clone() for k, v in eng.forward(img).items()}
        runner = D.ShardedRunner(lambda x: eng.forward(x), gather=True)
        out = runner(img)
        h = D.all_gather_records(D.pack_records(eng.forward(img)), 6, async_op=True)
        piped = D.unpack_records(h.wait())
        torch.cuda.synchronize()
        for k in out:
            assert torch.equal(out[k], direct[k]) and torch.equal(piped[k], direct[k]), k
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--vit-gemm", "f32"], []], ids=["f32", "split3"])
def test_two_ranks_with_real_engines_on_one_gpu(built_lib, cuda_dev, extra):
    """The N > 1 orchestration with REAL engines: two gloo ranks, both on cuda:0 (RCCL refuses duplicate devices, gloo stages GPU
    tensors through the host).  Rank 0 loads, the arena is broadcast BEFORE rank 0 finalizes, rank 1 finalizes with
    assume_all_loaded, both run their shard, records are gathered, and rank 0 recomputes rank 1's seeded shard on its own engine:
    bit-identical or the receiver's model is wrong (the round-2 regression: uninitialised resample tables on ranks != 0).
    In the split3 mode every rank builds its own split copy of the ViT weights from the broadcast arena (16 crops per rank: its unsplit range); the receiver's results must again be rank 0's bit for bit."""
    split = not extra                     # bench.py's default with real engines and >= 3 crops per GPU is the split3 mode
    n = 16 if split else 5
    j = _line(_run(["--gpus", "2", "--backend", "gloo", "--single-device", "--vit-depth", "2", "--batch", str(n), "--steps", "3", "--warmup", "1",
                    "--no-cpu-baseline"] + extra, timeout=900))
    assert j["n_gpus"] == 2 and j["config"]["ranks"] == 2 and "single-device" in j["dry_run"]
    m = j["multi_gpu"]
    assert m["checkpoint_readers"] == [0] and j["gathered_records_ok"] is True
    assert m["cross_rank_check"]["ranks_checked"] == 1 and m["cross_rank_check"]["bit_identical"] is True, m["cross_rank_check"]
    assert [r["crops"] for r in m["per_rank"]] == [n, n] and m["bcast_bytes"] > 1e8
    assert ("bf16" in j["dtype"]) == split and j["vit_gemm"] == ("split3" if split else "f32")
