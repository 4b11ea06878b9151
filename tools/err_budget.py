"""Developer helper (not a test): where does the SFT model's valence error come from?  Runs the HIP model on the GPU,
then replays each downstream stage with the CPU oracle FROM THE GPU's intermediate, so each stage's own error shows.
    python tools/err_budget.py [T] [B]
"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests/golden")
import oracle                                    # noqa: E402
import recipe as R                               # noqa: E402
from multimodal_transformer_amd import multiTransformer as MT, eval_ccc, functional as F   # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-300)))


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda:0")
    model = MT.NLPTransformer(512, embed_dim=128, h=8, device=dev)
    p32 = R.gen_params(R.shapes_of(model.state_dict()), 9)
    model.load_state_dict(p32)
    model = model.to(dev).eval()
    lengths = [T] * B
    mask_c = R.prefix_mask(lengths, T)
    x_c = torch.tanh(R.gen_normal("full:sft:x", (B, T, 512), 9))
    torch.set_num_threads(8)
    with torch.no_grad():
        x, mask = x_c.to(dev), mask_c.to(dev)
        e = F.linear(x, model.embed[1].weight, model.embed[1].bias, act=1)
        enc = model.encoder(e, mask)
        y = model._decode(enc, mask)
        # oracle stages
        e_o = torch.relu(x_c @ p32["embed.1.weight"].T + p32["embed.1.bias"])
        enc_o = oracle.encoder_stack(p32, "encoder.", e_o, mask_c, 8)
        y_o = oracle.lstm_decoder_head(p32, enc_o) * mask_c
        enc_from_gpu_e = oracle.encoder_stack(p32, "encoder.", e.cpu(), mask_c, 8)
        y_from_gpu_enc = oracle.lstm_decoder_head(p32, enc.cpu()) * mask_c
    print("embed           rel_l2 %.3e" % rel(e.cpu(), e_o))
    print("encoder (total) rel_l2 %.3e   own (oracle from GPU embed) %.3e" % (rel(enc.cpu(), enc_o), rel(enc.cpu(), enc_from_gpu_e)))
    print("valence (total) rel_l2 %.3e   decoder own (oracle from GPU enc) %.3e" % (rel(y.cpu(), y_o), rel(y.cpu(), y_from_gpu_enc)))
    for b in range(min(B, 3)):
        print("  seq %d CCC total %.6f  decoder-own %.6f" % (b, eval_ccc(y_o[b].reshape(-1).numpy(), y[b].cpu().reshape(-1).numpy()),
                                                         eval_ccc(y_from_gpu_enc[b].reshape(-1).numpy(), y[b].cpu().reshape(-1).numpy())))
    print("valence std %.4e mean %.4e" % (float(y_o.std()), float(y_o.mean())))


if __name__ == "__main__":
    main()
