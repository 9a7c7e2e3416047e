"""TEST INFRASTRUCTURE ONLY — golden code indices from the reference's OWN tokenizer encoder + quantiser
(PoseSPEncoderV1 / QuantizeEMAReset imported in place).  Run in the build container:  python oracle/gen_golden_encode.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tokenhmr_amd.config import RELEASE  # noqa: E402
from tokenhmr_amd import weights as W  # noqa: E402
from oracle import ref_import, tokenhmr_oracle as O  # noqa: E402


def make_pose(B=3, seed=0):
    g = torch.Generator().manual_seed(6000 + seed)
    return torch.randn(B, 21, 6, generator=g)


def main():
    ns = ref_import.load()
    enc_sd = W.make_synthetic_encoder(RELEASE, 0)
    tok = W.make_synthetic_tokenizer(RELEASE, 0)
    enc = ns.vqvae.PoseSPEncoderV1(rot_type="rot6d", input_dim=6, output_emb_width=256, down_t=1, width=512, depth=2,
                                   token_size_mul=4, dilation_growth_rate=3)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in enc_sd.items()}, strict=True)   # proves the name contract
    enc.eval()
    q = ns.quantize_cnn.QuantizeEMAReset(2048, 256)
    q.load_state_dict({"codebook": tok["quantizer.codebook"]}, strict=True)
    pose = make_pose()
    with torch.no_grad():
        x = enc(pose)                                   # EncodeTokens.forward :336
        lat = q.preprocess(x)                           # :339
        idx = q.quantize(lat)                           # :340
        oidx, olat, dist = O.vq_encode(pose, enc_sd, tok["quantizer.codebook"])
    print("oracle vs reference: latent max|diff| =", (olat - lat).abs().max().item(), " idx equal:", bool(torch.equal(oidx, idx)))
    two = dist.topk(2, dim=-1, largest=False).values
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "encode_small.npz"), idx=idx.numpy().astype(np.int32),
                        latent_sample=lat[::7].numpy(), gap=(two[:, 1] - two[:, 0]).numpy(),
                        checksum=np.array([W.checksum(enc_sd)]))
    print("wrote encode_small.npz", idx.shape, "min top-2 distance gap", float((two[:, 1] - two[:, 0]).min()))


if __name__ == "__main__":
    main()
