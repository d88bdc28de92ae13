"""bench.py argument handling that needs no GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout) and "n_gpus" not in p.stdout


def test_self_launch_fails_without_devices():
    """no launcher, --gpus 2, no GPU in this container: both rank processes are started, find no device and the command fails"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.device_count() >= 2:
        return
    assert p.returncode != 0 and "n_gpus" not in p.stdout
    assert "rank exit codes" in p.stderr


def test_every_baseline_config_has_a_bench_mode():
    sys.path.insert(0, ROOT)
    import bench
    assert {"space", "space4096", "ball", "space_n3", "space_bm", "ball_bm", "reach", "reach_brake"} <= set(bench.CONFIGS)
    for name in bench.CONFIGS:
        cfg, kw, algo, what = bench._env_config(name)
        from safemotionsrisk_amd.scene import build_scene
        kw = {k: v for k, v in kw.items() if k != "risk_networks"}
        assert build_scene(cfg, **kw).struct.obs_size > 0 and algo > 0
