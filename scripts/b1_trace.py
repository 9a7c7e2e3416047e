import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokenhmr_amd.config import HMRConfig
from tokenhmr_amd import weights as W
from tokenhmr_amd.smpl_assets import make_synthetic_smpl
from tokenhmr_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0"); cfg = HMRConfig()
eng = Engine(cfg, max_batch=B, device=dev)
eng.load_state(W.make_synthetic_state(cfg, 0), W.make_synthetic_tokenizer(cfg, 0)); eng.load_smpl(make_synthetic_smpl(cfg, 0)); eng.finalize()
img = torch.randn(B, 3, 256, 256, device=dev); outs = eng._alloc_outputs(B, taps=False, want_probs=True)
for _ in range(10): eng.forward(img, outputs=outs)
torch.cuda.synchronize()
