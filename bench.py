#!/usr/bin/env python3
"""Benchmark of the SafeMotions hot path on MI355X (contract: see the task statement / DESIGN.md).

    python bench.py --gpus N --steps K --warmup W

A "step" is one env.step() of every environment of the batch (plan + collide + finish + reset kernels,
with auto-reset, so finished episodes are re-sampled inside the timed region).  Workload = BASELINE.json
configs[2]: 65 536 envs per GPU, 7-DoF space/planet scene, collision_check_time=0.033 (3 contact samples
per step).  N > 1: weak scaling, every rank steps its own 65 536 envs and the (obs, reward, done) shards
are all-gathered over RCCL once per step (configs[4] at N = 8).  Actions are resident on the GPU before
the timed region; rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 342      # SURVEY section 8d, space scene (DESIGN.md "algorithmic bytes")
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3


def cpu_baseline(scene_config, seconds_target=12.0):
    """The float64 CPU restatement (oracle/, kind "port") on the host cores, bounded sample."""
    import numpy as np
    from safemotionsrisk_amd.scene import build_scene
    from oracle.oracle import OracleEnv
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    B = 4096
    scene = build_scene(scene_config, auto_reset=True)
    env = OracleEnv(scene, B, seed=0, num_threads=cores)
    env.reset()
    rng = np.random.default_rng(0)
    actions = [rng.uniform(-1, 1, (B, 7)) for _ in range(4)]
    env.step(actions[0])
    steps, t0 = 0, time.perf_counter()
    while True:
        env.step(actions[steps % 4])
        steps += 1
        el = time.perf_counter() - t0
        if el > seconds_target or steps >= 400:
            break
    env.close()
    return dict(value=B * steps / el, unit="env-steps/s", cores=cores, kind="port",
                sample="%d envs x %d steps of the same space scene (auto-reset, uniform actions, OpenMP over envs, "
                       "float64 oracle; kinematic step only, no Bullet physics sub-steps)" % (B, steps))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=65536)
    ap.add_argument("--checks", type=int, default=None, help="contact samples per step (default: from collision_check_time = 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the per-step all-gather")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as entry
    from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG
    from safemotionsrisk_amd.dist import ShardedRollout

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    distributed = world > 1
    if rank == 0 and not os.path.exists(os.path.join(ROOT, "safemotionsrisk_amd", "libsmr_hip.so")):
        entry.build()
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        dist.barrier()

    B = args.envs_per_gpu
    env = VecSafeMotionsEnv(B, device=local_rank, seed=0, auto_reset=True, collision_checks_per_step=args.checks,
                            return_info=False, env_index_offset=rank, **SPACE_ENV_CONFIG)
    rollout = ShardedRollout(env) if (distributed and not args.no_gather) else None
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234 + rank)
    actions = [torch.rand((B, 7), device="cuda", generator=gen) * 2 - 1 for _ in range(8)]   # resident before timing
    env.reset()

    def one_step(i):
        if rollout is not None:
            rollout.step(actions[i % 8])
        else:
            env.step(actions[i % 8])

    for i in range(args.warmup):
        one_step(i)
    env.set_profiling(True)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = env.get_profile()
    env.set_profiling(False)
    if distributed:
        tmax = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        n_checks = int(env.scene.struct.collision_checks_per_step)
        total_env_steps = B * world * args.steps
        kernels = {k: (v["ms"] / max(v["launches"], 1)) for k, v in prof.items()}
        dominant = max(kernels, key=kernels.get)
        dom_ms = kernels[dominant]
        achieved = ALGO_BYTES_PER_ENV_STEP * B / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM bytes per launch of the dominant kernel: measured in separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
        # FETCH_SIZE doubled on gfx950) of this command at the default configuration, committed under profiles/
        traffic, pmc_note = None, ""
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "round1_final_pmc.json")))
            if B == 65536 and n_checks == 3 and world == 1:
                traffic = pmc["kernels"]["smr::%s_kernel" % dominant]["hbm_bytes"]
                pmc_note = "; traffic = (2 x FETCH_SIZE + WRITE_SIZE) per launch from profiles/round1_final_pmc.json"
        except Exception:
            pass
        # kernels on the critical path: the reset kernel of a scene without a ball runs on the side stream, overlapped
        overlapped = ["reset"] if not env.scene.struct.use_ball and not os.environ.get("SMR_SYNC_RESET") == "1" else []
        critical = {k: v for k, v in kernels.items() if k not in overlapped}
        if critical:
            dominant = max(critical, key=critical.get)
            dom_ms = critical[dominant]
            achieved = ALGO_BYTES_PER_ENV_STEP * B / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
            if traffic is not None:
                try:
                    traffic = pmc["kernels"]["smr::%s_kernel" % dominant]["hbm_bytes"]
                except Exception:
                    traffic = None
        # what actually bounds the dominant kernel: vector issue.  Wave-level VALU instructions per launch (SQ_INSTS_VALU of the same
        # PMC passes) x 4 cycles on one of the chip's 1024 SIMDs at 2.4 GHz, against the kernel's measured time
        issue = None
        try:
            if traffic is not None:
                valu = pmc["kernels"]["smr::%s_kernel" % dominant]["SQ_INSTS_VALU"]
                floor_ms = valu * 4.0 / (1024 * 2.4e9) * 1e3
                issue = {"valu_insts_per_launch": valu, "issue_floor_ms": floor_ms, "kernel_ms": dom_ms, "frac_of_peak_issue": floor_ms / dom_ms}
        except Exception:
            issue = None
        out = {
            "metric": "env-steps/sec at batch 65 536, 7-DoF space scene",
            "value": total_env_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: %d envs/GPU, space/planet scene (robot_scene=0, obstacle_scene=5, "
                                   "planet_mode), dt=0.1, T=2.0, %d contact samples/step, auto-reset, uniform(-1,1) actions"
                                   % (B, n_checks),
                       "envs_per_gpu": B, "global_batch": B * world, "collision_checks_per_step": n_checks,
                       "all_gather": bool(rollout is not None), "parallelism": "env-sharded x%d" % world},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "note": "path is FP32-VALU/LDS-latency bound, not HBM bound (SURVEY section 0.5): %d algorithmic B per "
                                 "env-step x %d envs per launch / mean %s-kernel time%s" % (ALGO_BYTES_PER_ENV_STEP, B, dominant, pmc_note)},
            "kernel_ms": kernels,
            "overlapped_kernels": overlapped,
        }
        if issue is not None:
            out["roofline"]["vector_issue"] = issue
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(SPACE_ENV_CONFIG)
        print(json.dumps(out))
    env.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
