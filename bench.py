#!/usr/bin/env python3
"""Benchmark of the SafeMotions hot path on MI355X (contract: see the task statement / DESIGN.md).

    python bench.py --gpus N --steps K --warmup W [--config NAME]

A "step" is one env.step() of every environment of the batch (plan + contact sweep + collide + finish kernels and the
refill of finished episodes, with auto-reset, so finished episodes are re-sampled inside the timed region; the info
matrix RLlib's callbacks read is written every step).  Default workload = BASELINE.json configs[2]: 65 536 envs per GPU,
7-DoF space/planet scene of the README command, contact checks at the reference's cadence (24 per step, one after every
1/240 s obstacle update, ctlp.py:2631-2637).  Other BASELINE configurations and variants: --config (see CONFIGS).
N > 1: weak scaling, every rank steps its own 65 536 envs and the (obs, reward, done) shards are all-gathered over RCCL
once per step (configs[4] at N = 8).  Without WORLD_SIZE in the environment `--gpus N` starts its N rank processes
itself (fresh children, before this process touches the GPU); under torchrun it uses the ranks it is given.
Actions are resident on the GPU before the timed region; rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s spec
VALU_CYCLES_PER_WAVE_INST = 2.0    # MI355X_MICROARCH.md constants table: "v_fma_f32 (wave64): 2 cyc (SIMD-32); one wave alone: 4"
N_SIMD, CLOCK_HZ = 1024, 2.4e9     # 256 CUs x 4 SIMD-32

#: name -> (env_config name, overrides, constructor keywords, algorithmic HBM bytes per env-step, what it is)
CONFIGS = {
    "space": ("SPACE", {}, {}, 342, "BASELINE configs[2]: space/planet scene (robot_scene=0, obstacle_scene=5, planet_mode), README command; "
                                    "24 contact samples per step = the reference's cadence"),
    "space_n3": ("SPACE", {}, dict(collision_checks_per_step=3), 342, "space scene with 3 contact samples per step (round-1 headline configuration)"),
    "space_n1": ("SPACE", {}, dict(collision_checks_per_step=1), 342, "space scene, contact tested at the end of the step only"),
    "space4096": ("SPACE", {}, {}, 342, "BASELINE configs[1]: 4096 envs on one GPU, space/planet scene"),
    "ball": ("BALL", {}, {}, 422, "BASELINE configs[3]: ball scene (ballistic moving obstacle), self + static + moving distance rewards"),
    "space_bm": ("SPACE", dict(ball_machine_mode=True), {}, 342, "space scene on robot_ball_machine.urdf = trained_networks/backup_networks/space/params.json"),
    "ball_bm": ("BALL", dict(ball_machine_mode=True), {}, 422, "ball scene on robot_ball_machine.urdf = trained_networks/backup_networks/ball/params.json"),
    "reach": ("REACH", {}, {}, 390, "target-point reaching task (SafeMotionsEnv, TargetPointReachingReward), ball machine mounted, 8 s episodes"),
    "reach_risk": ("REACH", {}, dict(risk_networks=True), 390,
                   "reaching task with risk-aware stepping (actions.py:303-340): risk network 30-256-128-1 and backup policy 23-256-128-14 "
                   "on the MFMA units, synthetic weights"),
    "reach_risk4": ("REACH", {}, dict(risk_networks="rollout4"), 390,
                    "reaching task with risk_state_config 4 (RISK_CHECK_NEXT_STATE_SIMULATE_NEXT_STEP_AND_BACKUP_TRAJECTORY, safe_motions_base.py:1518-1577): "
                    "every env-step steps a background copy of the batch 1 + 20 times on the device (the action, then 20 actions of the backup policy "
                    "23-256-128-14, synthetic weights) and replaces the action where the copy collides; value counts the OUTER env-steps"),
    "reach_brake": ("REACH", dict(check_braking_trajectory_collisions=True), {}, 390,
                    "reaching task with the braking-trajectory safety layer (README.md:95: --check_braking_trajectory_collisions "
                    "--closest_point_safety_distance=0.01 --collision_check_time=0.033)"),
}


def _env_config(name):
    import safemotionsrisk_amd as smr
    base, over, kw, algo, what = CONFIGS[name]
    cfg = dict(getattr(smr, base + "_ENV_CONFIG"), **over)
    return cfg, kw, algo, what


def cpu_baseline(config_name, seconds_target=8.0):
    """The float64 CPU restatement (oracle/, kind "port") on the host cores, bounded samples: all cores over 4096 envs, one
    thread over 256 envs, and BASELINE configs[0] - ONE env, 1000 steps of uniform(-1,1,7) actions (PCG64 seed 0), auto-reset,
    the loop shape of evaluate.py:198-303 with --random_agent (base.py:1047-1048)."""
    import numpy as np
    from safemotionsrisk_amd.scene import build_scene
    from oracle.oracle import OracleEnv
    cfg, kw, _, _ = _env_config(config_name)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    kw = {k: v for k, v in kw.items() if k != "risk_networks"}
    scene = build_scene(cfg, auto_reset=True, **kw)

    def run(B, threads, budget, max_steps):
        env = OracleEnv(scene, B, seed=0, num_threads=threads)
        env.reset()
        rng = np.random.default_rng(0)
        actions = [rng.uniform(-1, 1, (B, 7)) for _ in range(4)]
        env.step(actions[0])
        steps, t0 = 0, time.perf_counter()
        while True:
            env.step(actions[steps % 4])
            steps += 1
            el = time.perf_counter() - t0
            if el > budget or steps >= max_steps:
                break
        env.close()
        return B * steps / el, steps

    all_cores, steps_all = run(4096, cores, seconds_target, 400)
    one_thread, steps_one = run(256, 1, seconds_target, 400)
    # configs[0]: 1 env, 1000 random-action steps, auto-reset
    env = OracleEnv(scene, 1, seed=0, num_threads=1)
    env.reset()
    rng = np.random.default_rng(0)
    t0, episodes = time.perf_counter(), 0
    for _ in range(1000):
        _, _, done, _, _ = env.step(rng.uniform(-1, 1, (1, 7)))
        episodes += int(done[0])
    single = 1000 / (time.perf_counter() - t0)
    env.close()
    return dict(value=all_cores, unit="env-steps/s", cores=cores, kind="port",
                sample="%d envs x %d steps of the same scene (auto-reset, uniform actions, OpenMP over envs, float64 oracle; kinematic step "
                       "only - the reference additionally runs 24 Bullet physics sub-steps per step)" % (4096, steps_all),
                one_thread={"value": one_thread, "cores": 1, "sample": "256 envs x %d steps" % steps_one},
                config1_single_env={"value": single, "cores": 1, "episodes": episodes,
                                    "sample": "BASELINE configs[0] shape: 1 env, 1000 uniform(-1,1,7) steps (PCG64 seed 0), auto-reset"})


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: N fresh rank processes (this process has not touched the GPU and does not)."""
    port = int(os.environ.get("MASTER_PORT", 0)) or _free_port()    # (the port is free when probed; a launcher that must not race passes its own)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained while the ranks run (a full pipe would block it for ever)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # a rank that fails leaves the others waiting in the rendezvous: stop them (our own children, by pid) instead of hanging
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    reader.join(timeout=30)
    out = b"".join(chunks)
    if any(codes):
        raise SystemExit("bench.py --gpus %d: rank exit codes %s" % (args.gpus, codes))
    sys.stdout.write(out.decode())
    sys.stdout.flush()


CSRC_FILES = ("smr_hip.hip", "smr_device.h", "smr_tables.h", "smr_mlp.h")


def csrc_sha256():
    """one hash over the four kernel sources (safemotionsrisk_amd/csrc): what scripts/make_pmc_json.py records with a counter file and
    what load_pmc() demands of it - counters of other kernels than the ones that ran are not reported"""
    import hashlib
    h = hashlib.sha256()
    for name in CSRC_FILES:
        h.update(name.encode() + b"\0")
        h.update(open(os.path.join(ROOT, "safemotionsrisk_amd", "csrc", name), "rb").read())
    return h.hexdigest()


def load_pmc(config_name, B, n_checks, profiles_dir=None):
    """profiles/round<R>_pmc_<config>.json of the latest round that has one (scripts/make_pmc_json.py from separate rocprofv3 --pmc
    passes of this command).  The file must have been collected from THESE kernel sources (run.csrc_sha256) at this batch size and
    contact cadence; anything else gives (None, why) and the bench line carries traffic: null - never another build's counters."""
    profiles_dir = profiles_dir or os.path.join(ROOT, "profiles")
    for rnd in range(9, 1, -1):
        path = os.path.join(profiles_dir, "round%d_pmc_%s.json" % (rnd, config_name))
        if os.path.exists(path):
            break
    else:
        return None, "no profiles/round*_pmc_%s.json" % config_name
    pmc = json.load(open(path))
    meta = pmc.get("run", {})
    rel = os.path.join("profiles", os.path.basename(path))
    if meta.get("envs_per_gpu") != B or meta.get("collision_checks_per_step") != n_checks:
        return None, "%s was collected at %s envs / %s checks, this run has %d / %d" % (rel, meta.get("envs_per_gpu"),
                                                                                     meta.get("collision_checks_per_step"), B, n_checks)
    have = csrc_sha256()
    if meta.get("csrc_sha256") != have:
        return None, "%s was collected from other kernel sources (csrc sha256 %s, this tree %s): counters withheld" % (
            rel, str(meta.get("csrc_sha256"))[:12], have[:12])
    return pmc, rel


def multi_gpu_report(world, devices, mismatches, overlap):
    """the keys of the bench line that describe the N > 1 exchange (and the self-check that ran before the timed region)"""
    return {"rccl_world": int(world), "devices": list(devices),
            "gather_check": {"steps": 2, "mode": "overlapped" if overlap else "synchronous", "mismatches": [list(m) for m in mismatches],
                             "what": "before the timed region: every rank's int64 checksums of its local (obs, reward, done) bytes against "
                                     "the slice every rank gathered (one all-gather of 3 x int64 per rank and step); a mismatch ends the run"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="space", choices=sorted(CONFIGS))
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--checks", type=int, default=None, help="contact samples per step (default 24 = the reference's cadence)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the per-step all-gather")
    ap.add_argument("--sync-gather", action="store_true", help="N > 1: every step waits for its all-gather (default: the gather of step t "
                                                               "runs beside the kernels of step t + 1, all of them inside the timed region)")
    ap.add_argument("--no-info", action="store_true", help="do not write the per-step info matrix")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)                       # before anything in this process touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start one rank per GPU (python -m torch.distributed.run --nproc-per-node %d ... "
                         "bench.py --gpus %d), or run without a launcher" % (args.gpus, world, args.gpus, args.gpus))

    import torch
    import __graft_entry__ as entry
    from safemotionsrisk_amd import VecSafeMotionsEnv
    from safemotionsrisk_amd.dist import ShardedRollout

    n_dev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if local_rank >= n_dev or local_world > n_dev:       # every rank sees this, so none is left waiting in the rendezvous
        raise SystemExit("rank %d: --gpus %d needs %d GPUs on this node, %d visible" % (rank, args.gpus, local_world, n_dev))
    distributed = world > 1
    if rank == 0 and not os.path.exists(os.path.join(ROOT, "safemotionsrisk_amd", "libsmr_hip.so")):
        entry.build()
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        dist.barrier()

    cfg, kw, algo_bytes, what = _env_config(args.config)
    B = args.envs_per_gpu or (4096 if args.config == "space4096" else 65536)
    if args.checks is not None:
        kw = dict(kw, collision_checks_per_step=args.checks)
    risk_nets = kw.pop("risk_networks", False)
    env = VecSafeMotionsEnv(B, device=local_rank, seed=0, auto_reset=True, return_info=not args.no_info, env_index_base=rank * B, **kw, **cfg)
    if risk_nets:                                       # synthetic weights of the shipped architectures (the checkpoints are not loaded)
        import numpy as np
        from safemotionsrisk_amd import backup_policy, risk_network
        wrng = np.random.default_rng(0)
        mk = lambda sizes: ([(wrng.normal(size=(a, b)) * 2 / np.sqrt(a)).astype(np.float32) for a, b in zip(sizes[:-1], sizes[1:])],
                            [np.zeros(b, np.float32) for b in sizes[1:]])
        r = env.risk_observation_size
        if risk_nets == "rollout4":                     # no risk network: the background copy rates the action (base.py:1560-1566)
            nets = (None, backup_policy(*mk([r, 256, 128, 14]), device=local_rank))
            env.set_risk_networks(*nets, risk_threshold=0.5, risk_state_config=4)
            env.set_risk_rollout(backup_steps=20)       # risk_state_backup_trajectory_steps of the shipped backup policies (2.0 s / 0.1 s, base.py:573-576)
        else:
            nets = (risk_network(*mk([r + 7, 256, 128, 1]), device=local_rank), backup_policy(*mk([r, 256, 128, 14]), device=local_rank))
            env.set_risk_networks(*nets, risk_threshold=0.5)
    rollout = ShardedRollout(env, overlap=not args.sync_gather) if (distributed and not args.no_gather) else None
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234 + rank)
    actions = [torch.rand((B, 7), device="cuda", generator=gen) * 2 - 1 for _ in range(8)]   # resident before timing
    env.reset()

    def one_step(i):
        if rollout is not None:
            rollout.step(actions[i % 8])
        else:
            env.step(actions[i % 8])

    gather_report = None
    if rollout is not None:
        # the exchange checks itself before anything is timed: a broken overlap is an error, not a silently wrong number
        bad = rollout.verify([actions[0], actions[1]])
        names = [None] * world
        dist.all_gather_object(names, "%s:%d" % (torch.cuda.get_device_name(local_rank), local_rank))
        flags = [None] * world
        dist.all_gather_object(flags, bad)
        gather_report = multi_gpu_report(dist.get_world_size(), names, [m for f in flags for m in f], rollout.overlap)
        if any(flags):
            raise SystemExit("rank %d: all-gather self-check failed, (step, rank) slices that differ from what their rank wrote, per rank: %s" % (rank, flags))
    for i in range(args.warmup):
        one_step(i)
    # HIP events around the launches of every `period`-th timed step: a pair of events is a pair of barrier packets in the queue, and around
    # every launch of every step they cost the step 12 % (0.485 -> 0.550 ms at 65 536 envs, measured)
    repeats = 1 if args.steps >= 100 else 5              # (see the timed regions below)
    period = max(1, min(16, args.steps * repeats // 6))  # at least six sampled steps; 16: the timed launches cost the run under 1 %
    env.set_profiling(period)
    # A short run (the driver's 20 steps after 5 warm-up steps) is timed `repeats` times back to back and the MEDIAN region is reported:
    # a single 10 ms region right after reset() holds the burst in which every surviving env ends its first episode (39 577 of
    # 65 536 at step 20) and reads 8 % low.  Every region is exactly --steps steps between barrier + synchronize.
    regions = []
    for rep in range(repeats):
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            one_step(rep * args.steps + i)
        if rollout is not None:
            rollout.wait()                              # every gather of the timed steps has landed before the clock stops
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if distributed:
            tmax = torch.tensor([el], device="cuda", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        regions.append(el)
    elapsed = sorted(regions)[len(regions) // 2]
    env.synchronize()                                   # raises if a device-side error condition was latched during the run
    prof = env.get_profile()
    env.set_profiling(False)

    if rank == 0:
        s = env.scene.struct
        n_checks = int(s.collision_checks_per_step)
        total_env_steps = B * world * args.steps
        kernels = {k: (v["ms"] / max(v["launches"], 1)) for k, v in prof.items()}
        # kernels on the critical path: the staged refill (reset, and in ball / reaching scenes the staged launches / targets) runs on
        # side streams, overlapped with the step kernels
        overlapped = ["reset"] if os.environ.get("SMR_SYNC_RESET") != "1" else []
        # "collide" is the whole distance stage; where it runs as keys / enumerate / jobs kernels (planet scenes) those are timed one
        # by one as well, and the dominant KERNEL - what the rocprofv3 summary under profiles/ lists - is picked among them
        stage_parts = [k for k in ("keys", "enumerate", "jobs") if kernels.get(k, 0.0) > 0.0]
        critical = {k: v for k, v in kernels.items() if k not in overlapped and not (k == "collide" and stage_parts)}
        dominant = max(critical, key=critical.get)
        dom_ms = critical[dominant]
        achieved = algo_bytes * B / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # measured counters of the same command (separate rocprofv3 --pmc passes, committed under profiles/)
        pmc, pmc_note = load_pmc(args.config, B, n_checks) if world == 1 else (None, "PMC profiles are collected at N = 1")
        traffic = traffic_step = issue = None
        kk = {}
        if pmc is not None:
            # ("brake" times brake_plan + brake_check + brake_apply; the counters are those of brake_check, 97 % of it)
            kname = {"brake": "smr::brake_check_kernel"}.get(dominant, "smr::%s_kernel" % dominant)
            kk = pmc["kernels"].get(kname, {})
            traffic = kk.get("hbm_bytes")
            step_kernels = [k for k in pmc["kernels"] if not any(x in k for x in ("reset", "spawn<staged>", "target<staged>"))]
            traffic_step = {"hbm_bytes": sum(pmc["kernels"][k].get("hbm_bytes", 0.0) * pmc["kernels"][k].get("launches_per_step", 1.0) for k in step_kernels),
                            "algorithmic_bytes": algo_bytes * B, "kernels": step_kernels}
            traffic_step["ratio"] = traffic_step["hbm_bytes"] / traffic_step["algorithmic_bytes"]
            if "SQ_INSTS_VALU" in kk:
                valu = kk["SQ_INSTS_VALU"]
                floor_ms = valu * VALU_CYCLES_PER_WAVE_INST / (N_SIMD * CLOCK_HZ) * 1e3
                issue = {"valu_insts_per_launch": valu, "cycles_per_wave_inst": VALU_CYCLES_PER_WAVE_INST, "issue_floor_ms": floor_ms,
                         "kernel_ms": dom_ms, "frac_of_peak_issue": floor_ms / dom_ms,
                         "lane_utilisation": kk.get("lane_utilisation"),
                         "note": "wave-level VALU instructions x 2 cycles (wave64 on a SIMD-32, >= 2 waves per SIMD) / 1024 SIMDs / 2.4 GHz against "
                                 "the kernel's measured time; lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): the share of "
                                 "the issued lanes that are switched on"}
        # wave-slot occupancy of the dominant kernel: the wavefronts' summed life time against (kernel time x wave slots of the chip)
        slot = None
        if pmc is not None and "SQ_WAVE_CYCLES" in kk and kk.get("SQ_WAVES"):
            waves_per_simd = {"collide": 4, "brake": 3, "jobs": 4}.get(dominant, 2)
            # the jobs kernel is a persistent grid (one 512-lane block per CU, two up to 32 768 envs): the wavefronts it launches ARE its slots
            slots = min(N_SIMD * waves_per_simd, int(kk["SQ_WAVES"])) if dominant == "jobs" else N_SIMD * waves_per_simd
            # SQ_WAVE_CYCLES counts in units of 4 cycles (MI355X_MICROARCH.md, SQ counters "quad-cycles")
            mean_wave_us = kk["SQ_WAVE_CYCLES"] * 4.0 / kk["SQ_WAVES"] / CLOCK_HZ * 1e6
            packed_ms = kk["SQ_WAVES"] * mean_wave_us * 1e-3 / slots
            slot = {"waves_per_launch": kk["SQ_WAVES"], "mean_wave_life_us": mean_wave_us, "wave_slots": slots, "packed_ms": packed_ms,
                    "slot_utilisation": packed_ms / dom_ms,
                    "note": "SQ_WAVE_CYCLES x 4 / SQ_WAVES = mean life of a wavefront; packed_ms = all wavefronts' life time spread evenly over the "
                            "%d wave slots (at most %d per SIMD at this kernel's register budget; a persistent grid holds as many as it launched); "
                            "slot_utilisation = packed_ms / measured kernel time: what is lost to the tail of the schedule (the kernel is as long "
                            "as its slowest wavefront)" % (slots, waves_per_simd)}
        label = "env-steps/sec at batch 65 536, 7-DoF space scene" if (args.config == "space" and B == 65536) else \
                "env-steps/sec, %s, %d envs/GPU" % (args.config, B)
        out = {
            "metric": label,
            "value": total_env_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "repeats": repeats,
            "region_ms": [round(r * 1e3, 4) for r in regions],
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s; %d envs/GPU, dt=%.3g, T=%.3g, %d contact samples/step, auto-reset, uniform(-1,1) actions, info matrix %s"
                                   % (what, B, s.dt, s.dt * s.episode_steps, n_checks, "off" if args.no_info else "on"),
                       "name": args.config, "envs_per_gpu": B, "global_batch": B * world, "collision_checks_per_step": n_checks,
                       "all_gather": (False if rollout is None else ("every step waits for it" if args.sync_gather else
                                                                        "overlapped with the next step (all gathers land inside the timed region)")), "parallelism": "env-sharded x%d" % world},
            # the contract's roofline object prices the kernel against HBM ("achieved" = algorithmic bytes / kernel time); this path is
            # NOT HBM bound (342 B per env-step), so "bound" names the resource that does bind it and the figures
            # that matter are vector_issue (share of the chip's VALU issue rate) and slot_utilisation (share of its wave slots)
            "roofline": {"bound": "valu", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_step": traffic_step,
                         "binding_resource": "fp32 vector issue + LDS latency of a few long-running wavefronts (not HBM, SURVEY section 0.5): "
                                             "achieved / peak / frac are the HBM figures the contract asks for, see vector_issue and wave_slots",
                         "vector_issue": issue, "wave_slots": slot, "pmc": pmc_note,
                         "stage_ms": {"collide": kernels.get("collide"), "parts": {k: kernels[k] for k in stage_parts}},
                         "note": "achieved = %d algorithmic B per env-step x %d envs per launch / mean %s-kernel time (HIP events on the "
                                 "handle's stream); traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch; stage_ms: the distance stage as a "
                                 "whole (HIP events around all its launches) and its kernels" % (algo_bytes, B, dominant)},
            "kernel_ms": kernels,
            "kernel_timing": {"every_nth_step": period, "timed_launches": {k: v["launches"] for k, v in prof.items() if v["launches"]},
                              "note": "HIP events on the stream each kernel is launched on, around its launches in every n-th of the timed steps"},
            "overlapped_kernels": overlapped,
        }
        if gather_report is not None:
            out.update(gather_report)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.config)
        print(json.dumps(out))
        sys.stdout.flush()
    env.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
