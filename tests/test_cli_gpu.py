"""The CLI counterpart of scripts/vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile.py end to end on a tiny synthetic PNG sequence:
PNG decode (host) -> device pre-processing (bicubic x4, reflect pad) -> RAFT flows + occlusion masks -> aggregation sampling
with motion guidance -> video-VAE decode + AdaIN -> crop + uint8 -> PNG."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cli_runs_on_png_sequence(tmp_path):
    from PIL import Image
    seq = tmp_path / "seqs" / "clip0"
    seq.mkdir(parents=True)
    rng = np.random.default_rng(0)
    base = rng.random((35, 46, 3))
    for i in range(3):                                        # 3 frames with n_frames=2: exercises the repeat-last padding
        im = np.kron(np.roll(base, i, 1), np.ones((4, 4, 1)))   # 140 x 184 LR -> x4 = 560 x 736 (not a multiple of 32)
        Image.fromarray((im * 255).astype(np.uint8)).save(seq / f"{i:03d}.png")
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile.py"), "--seqs-path",
           str(tmp_path / "seqs"), "--outdir", str(out), "--ddpm_steps", "2", "--n_frames", "2", "--latent-dir",
           str(tmp_path / "lat")]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    files = sorted(os.listdir(out / "clip0"))
    assert files == ["000.png", "001.png", "002.png"]
    for f in files:
        im = np.asarray(Image.open(out / "clip0" / f))
        assert im.shape == (560, 736, 3) and im.dtype == np.uint8
        assert im.std() > 1.0                                  # not a constant image
    lat = np.load(tmp_path / "lat" / "clip0" / "001.npy")
    assert lat.shape == (4, 576 // 8, 768 // 8) and np.isfinite(lat).all()      # 560 x 736 reflect-padded to 576 x 768 (both sides grow)
