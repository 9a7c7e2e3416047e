#!/bin/bash
# Round 3, session F: 64x128 / 128x64 tiles (unsplit and split-K) on the N = 1280 GEMMs at 7 ... 16 crops; tile sweep with the new tiles;
# B = 1 with the ViT's weights resident in the Infinity Cache (depth 1-3) against cold weights (depth 32)
set -u
O=gpurun_out/r3f; mkdir -p $O; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $O/pytest_ops.log 2>&1; tail -2 $O/pytest_ops.log
timeout 400 python scripts/n1280_sweep.py 3 2>/dev/null > $O/n1280_sweep.log; cat $O/n1280_sweep.log | cut -c1-330
echo "t=$(( $(date +%s) - t0 ))"
for env in "THMR_MID_SPLIT=22 THMR_MID_TILE=12" "THMR_MID_SPLIT=22 THMR_MID_TILE=13" "THMR_MID_SPLIT=44 THMR_MID_TILE=12" "THMR_MID_SPLIT=02 THMR_MID_TILE=12" "THMR_MID_SPLIT=00"; do
  echo "== $env" >> $O/mid_tile.log
  env $env timeout 300 python scripts/mid_split_sweep.py 7 8 9 10 12 16 2>/dev/null | grep '^{' >> $O/mid_tile.log
done
cat $O/mid_tile.log | cut -c1-500
echo "t=$(( $(date +%s) - t0 ))"
python - > $O/weights_resident_b1.log 2>/dev/null <<'PY'
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
from tokenhmr_amd.config import HMRConfig
from tokenhmr_amd import weights as W
from tokenhmr_amd.smpl_assets import make_synthetic_smpl
from tokenhmr_amd.engine import Engine
dev = torch.device("cuda:0")
res = {}
for depth in (1, 2, 3, 32):
    cfg = HMRConfig(vit_depth=depth)
    eng = Engine(cfg, max_batch=8, device=dev)
    eng.load_state(W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0)); eng.load_smpl(make_synthetic_smpl(cfg, 0)); eng.finalize()
    for B in (1, 2, 8):
        img = torch.randn(B, 3, 256, 256, generator=torch.Generator().manual_seed(B)).to(dev)
        feats = torch.empty(B, 192, 1280, device=dev)
        for _ in range(5): eng.vit_forward(img, out=feats)
        torch.cuda.synchronize(); n = 50; t0 = time.perf_counter()
        for _ in range(n): eng.vit_forward(img, out=feats)
        torch.cuda.synchronize()
        res[f"depth{depth}_B{B}"] = round((time.perf_counter() - t0) / n * 1e3, 4)
    del eng; torch.cuda.empty_cache()
for B in (1, 2, 8):
    warm = res[f"depth3_B{B}"] - res[f"depth2_B{B}"]; warm1 = res[f"depth2_B{B}"] - res[f"depth1_B{B}"]
    cold = (res[f"depth32_B{B}"] - res[f"depth2_B{B}"]) / 30
    print(f"B={B}: ViT-only ms per call at depth 1/2/3/32 = {res[f'depth1_B{B}']}, {res[f'depth2_B{B}']}, {res[f'depth3_B{B}']}, {res[f'depth32_B{B}']};  per layer: weights resident in the 256 MB Infinity Cache (depth 2->3) {warm*1e3:.1f} us (1->2: {warm1*1e3:.1f}), cold from HBM (depth 32) {cold*1e3:.1f} us")
print(json.dumps(res))
PY
cat $O/weights_resident_b1.log
echo "total t=$(( $(date +%s) - t0 ))"
