"""The evaluation loop of the reference (tokenhmr/eval.py:116-158, `run_eval`) as a data-parallel job: one process per GPU,
rank r evaluates the contiguous dataset shard [r*len/N, (r+1)*len/N), the per-sample metrics are merged at the end
(tokenhmr_amd.dist.merge_evaluator).  With one process it is exactly the reference's loop: DataLoader(shuffle=False) ->
recursive_to -> model(batch) under no_grad -> evaluator(out, batch) -> evaluator.log() every `log_freq` batches.

Dataset construction (`create_dataset`, lib/datasets/__init__.py) and rendering stay with the reference; any map-style
dataset whose items carry the keys the evaluator reads (`img`, `keypoints_3d`, `vertices`, `imgname`) works."""
import torch
import torch.distributed as dist

from . import dist as D


def recursive_to(x, device):
    """lib/utils/__init__.py:9-25 `recursive_to`: move every tensor of a nested dict / list to `device`."""
    if torch.is_tensor(x):
        return x.to(device, non_blocking=True)
    if isinstance(x, dict):
        return {k: recursive_to(v, device) for k, v in x.items()}
    if isinstance(x, list):
        return [recursive_to(v, device) for v in x]
    return x


def run_eval(model, dataset, evaluator, batch_size=64, device=None, num_workers=0, log_freq=0):
    """Returns `evaluator.get_metrics_dict()` over the WHOLE dataset on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    total = len(dataset)
    s, e = D.shard_range(total, world, rank)
    shard = torch.utils.data.Subset(dataset, range(s, e)) if world > 1 else dataset
    loader = torch.utils.data.DataLoader(shard, batch_size, shuffle=False, num_workers=num_workers)
    device = device if device is not None else getattr(model, "device", None)
    for i, batch in enumerate(loader):
        batch = recursive_to(batch, device)
        with torch.no_grad():
            out = model(batch)
        evaluator(out, batch)
        if log_freq and i % log_freq == log_freq - 1 and rank == 0:
            evaluator.log()
    D.merge_evaluator(evaluator, total)
    if rank == 0:
        evaluator.log()
    return evaluator.get_metrics_dict()
