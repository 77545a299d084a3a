"""CPU: the host-side frame writer (`save_videos_grid`, EEG2Video_New/Generation/tuneavideo/util.py:20-32) -- grid layout as
torchvision.utils.make_grid builds it, the truncating uint8 conversion, GIF / NPY output."""
import numpy as np
import torch

from eeg2video_amd.util import make_grid, save_videos_grid


def test_make_grid_layout():
    x = torch.arange(5 * 3 * 2 * 3, dtype=torch.float32).reshape(5, 3, 2, 3)
    g = make_grid(x, nrow=4)                                   # 4 per row -> 2 rows, padding 2, zeros between
    assert g.shape == (3, 2 * (2 + 2) + 2, 4 * (3 + 2) + 2)
    assert torch.equal(g[:, 2:4, 2:5], x[0]) and torch.equal(g[:, 2:4, 7:10], x[1]) and torch.equal(g[:, 6:8, 2:5], x[4])
    assert float(g[:, :2].abs().sum()) == 0 and float(g[:, 6:8, 7:].abs().sum()) == 0      # padding and the empty cells
    assert torch.equal(make_grid(x[:1], nrow=4), x[0])                                       # one image: returned as it is


def test_save_videos_grid_gif_and_npy(tmp_path):
    from oracle import frames_to_uint8
    from PIL import Image
    g = torch.Generator().manual_seed(0)
    v = torch.rand(2, 3, 6, 16, 24, generator=g)
    out = save_videos_grid(v, str(tmp_path / "a" / "clip.npy"))
    assert out.dtype == np.uint8 and out.shape == (6, 16 + 4, 2 * (24 + 2) + 2, 3)
    assert np.array_equal(np.load(tmp_path / "a" / "clip.npy"), out)
    want = frames_to_uint8(v)                                  # (x * 255).astype(uint8), truncation
    assert np.array_equal(out[:, 2:18, 2:26], np.transpose(want[0], (1, 2, 3, 0)))
    assert np.array_equal(out[:, 2:18, 28:52], np.transpose(want[1], (1, 2, 3, 0)))
    save_videos_grid(v, str(tmp_path / "clip.gif"), fps=3)
    im = Image.open(tmp_path / "clip.gif")
    assert im.n_frames == 6 and im.size == (54, 20) and im.info["duration"] == 330      # GIF delays are centiseconds: 1000 / 3 ms -> 33 cs
    r = save_videos_grid(v * 2 - 1, str(tmp_path / "r.npy"), rescale=True)
    assert np.abs(r[:, 2:18, 2:26].astype(int) - out[:, 2:18, 2:26].astype(int)).max() <= 1      # image cells agree ...
    assert int(r[0, 0, 0, 0]) == 127                                   # ... and the zero padding is rescaled with them, as the reference does
    u8 = torch.from_numpy(want)                                        # frames already uint8 (Engine.frames_to_uint8): laid out as they are
    assert np.array_equal(save_videos_grid(u8, str(tmp_path / "u8.npy")), out)
