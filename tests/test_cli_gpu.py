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
           str(tmp_path / "seqs"), "--outdir", str(out), "--ddpm_steps", "2", "--n_frames", "2", "--dec_w", "1.0", "--colorfix_type",
           "adain", "--device", "cuda", "--latent-dir", str(tmp_path / "lat")]
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


@pytest.mark.gpu
def test_bench_json_line_contract():
    """bench.py prints exactly ONE JSON line with the driver's keys, the roofline object of the dominant kernel and (when
    asked) the CPU baseline; run here on the reduced-width nets (a plumbing check — `config.reduced_width` says so)."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--small", "--frames", "2", "--size", "128", "--ddpm-steps", "3",
           "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f16" and d["data"] == "synthetic" and d["value"] > 0
    assert d["config"]["finite"] is True and d["config"]["reduced_width"] is True and "workload" in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
