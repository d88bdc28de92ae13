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


def test_counters_of_other_kernel_sources_are_withheld(tmp_path):
    """bench.load_pmc: a counter file is used only if it was collected from the kernel sources of this tree (run.csrc_sha256, written by
    scripts/make_pmc_json.py) at this batch size and contact cadence; anything else gives traffic: null and a note, never stale numbers"""
    import json
    sys.path.insert(0, ROOT)
    import bench
    sha = bench.csrc_sha256()
    assert len(sha) == 64 and sha == bench.csrc_sha256()
    run = {"config": "space", "envs_per_gpu": 65536, "collision_checks_per_step": 24, "csrc_sha256": sha}
    kernels = {"smr::jobs_kernel": {"hbm_bytes": 1.0}}
    json.dump({"run": run, "kernels": kernels}, open(tmp_path / "round7_pmc_space.json", "w"))
    pmc, note = bench.load_pmc("space", 65536, 24, profiles_dir=str(tmp_path))
    assert pmc is not None and note.endswith("round7_pmc_space.json")
    assert bench.load_pmc("space", 4096, 24, profiles_dir=str(tmp_path))[0] is None
    json.dump({"run": dict(run, csrc_sha256="0" * 64), "kernels": kernels}, open(tmp_path / "round8_pmc_space.json", "w"))
    pmc, note = bench.load_pmc("space", 65536, 24, profiles_dir=str(tmp_path))        # the newest round's file is the one that counts
    assert pmc is None and "other kernel sources" in note
    json.dump({"run": {k: v for k, v in run.items() if k != "csrc_sha256"}, "kernels": kernels}, open(tmp_path / "round9_pmc_space.json", "w"))
    assert bench.load_pmc("space", 65536, 24, profiles_dir=str(tmp_path))[0] is None  # a file without a hash proves nothing
    # every counter file committed for the current round belongs to this tree or is withheld - load_pmc never raises on them
    for name in bench.CONFIGS:
        bench.load_pmc(name, 65536, 24)


def test_multi_gpu_keys_of_the_bench_line():
    sys.path.insert(0, ROOT)
    import bench
    rep = bench.multi_gpu_report(8, ["AMD Instinct MI355X:%d" % i for i in range(8)], [], True)
    assert rep["rccl_world"] == 8 and len(rep["devices"]) == 8 and rep["gather_check"]["mismatches"] == [] and rep["gather_check"]["mode"] == "overlapped"
    assert bench.multi_gpu_report(2, ["a", "b"], [(0, 1)], False)["gather_check"]["mismatches"] == [[0, 1]]
