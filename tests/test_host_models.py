"""CPU: the host-side EEG models of BASELINE configs[4] (eeg2video_amd/host_models.py) produce tensors of the shapes the
accelerated path takes; the Seq2Seq decoder is autoregressive (frame i depends on the windows only through frames < i's tokens)."""
import torch

from eeg2video_amd.host_models import GLMNet, Seq2SeqLatents


def test_glmnet_and_seq2seq_shapes():
    torch.manual_seed(0)
    g = GLMNet(out_dim=2, emb_dim=64, C=62, T=200).eval()
    assert g(torch.randn(3, 1, 62, 200)).shape == (3, 2)
    s = Seq2SeqLatents(d_model=64, latent_shape=(4, 4, 6), frames=3).eval()
    x = torch.randn(2, 7, 62, 100)
    y = s(x)
    assert y.shape == (2, 3, 4, 4, 6) and torch.isfinite(y).all()
    assert torch.equal(s(x), y)                          # eval mode: deterministic
    assert not torch.equal(s(x.flip(0)), y)
