#!/usr/bin/env python3
"""Headline benchmark: particle-steps/s of the per-step force path (tree gravity + Ewald + SPH
density + hydro) on BASELINE.json's config c2 (64^3 DM + 64^3 gas periodic box), on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full pass of the path over the resident particle set, exactly the work of
compute_accelerations() on a full step (accel.c:61-106): tree build, Newtonian walk with the
relative opening criterion, Ewald-correction walk, OldAcc/G finish, SPH density (h iteration),
hmax refresh, SPH hydro.  Inputs are resident in HBM when the timed region starts.
Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "tests"))

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_INTERACTION = 32.0    # s[3] + mass, fp64
BYTES_PER_GRAV_TARGET = 68.0    # 40 B in + 28 B out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ng", type=int, default=64, help="particles per dimension per species")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the force path has no CPU fallback")
    # BENCH_BACKEND=gloo: self-test of the multi-rank path with all ranks on GPU 0 (RCCL refuses
    # two ranks on one device); the driver's runs use nccl = RCCL, one rank per GPU
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from common import Problem  # seeded workload definition shared with the parity tests
    B = importlib.import_module("gadget-leicester_amd.bindings")
    S = importlib.import_module("gadget-leicester_amd.sharded")

    pr = Problem(ng=args.ng, gas=True, periodic=1)
    fp = pr.device(local_rank)
    drv = S.ShardedForceStep(fp, rank, world, dist=dist if world > 1 else None, device=device)
    tree_args = (pr.extent[0], pr.extent[1], pr.extent[2], pr.force_soft)
    gp_bh, gp_rel = pr.g_grav(pr.theta), pr.g_grav(0.0)
    dp, hp = pr.g_dens(), pr.g_hydro()
    walks = [B.WALK_NEWTON, B.WALK_EWALD]

    def sync():
        fp.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def allreduce(t, op):
        if backend == "nccl":
            dist.all_reduce(t, op=op)
            return t
        h = t.cpu()
        dist.all_reduce(h, op=op)
        return h

    # step 0 of a run (accel.c:61-68): Barnes-Hut pass to obtain OldAcc, then relative criterion
    fp.set_field(B.F_OLDACC, np.zeros(pr.n))
    drv.step(tree_args, gp_bh, dp, hp, pr.G, walks)
    for _ in range(args.warmup):
        drv.step(tree_args, gp_rel, dp, hp, pr.G, walks)
    sync()

    phase_ms = {k: 0.0 for k in ("tree", "grav", "ewald", "dens", "hmax", "hydro")}
    grav_int = ewald_int = dens_ngb = hyd_pairs = dens_iter = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        drv.step(tree_args, gp_rel, dp, hp, pr.G, walks)
        st = fp.stats()          # device-event times + interaction counters of this step
        for k in phase_ms:
            phase_ms[k] += st["ms_" + k]
        grav_int += st["grav_interactions"]
        ewald_int += st["ewald_interactions"]
        dens_ngb += st["dens_neighbours"]
        hyd_pairs += st["hydro_pairs"]
        dens_iter += st["dens_iterations"]
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        t = allreduce(t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # whole-job interaction counts for the roofline line
        c = torch.tensor([grav_int, ewald_int, dens_ngb, hyd_pairs], dtype=torch.float64,
                         device=device)
        c = allreduce(c, dist.ReduceOp.SUM)
        grav_int_all = float(c[0].item())
    else:
        grav_int_all = float(grav_int)

    # the dominant kernel alone (outside the timed region): in the timed steps it shares the chip
    # with the Ewald walk and the SPH kernels, which stretches its own duration
    iso = []
    for _ in range(3):
        fp.gravity(gp_rel, B.WALK_NEWTON)
        iso.append(fp.stats()["ms_grav"])
    iso_ms = float(np.median(iso))

    K = max(args.steps, 1)
    ms_per_step = 1e3 * elapsed / K
    value = pr.n * K / elapsed

    out = None
    if rank == 0:
        # dominant kernel: k_grav_walk<NEWTON>.  Algorithmic bytes per launch on THIS rank.
        my_targets = st["grav_targets"]
        alg_bytes = BYTES_PER_INTERACTION * (grav_int / K) + BYTES_PER_GRAV_TARGET * my_targets
        kern_s = 1e-3 * phase_ms["grav"] / K
        achieved = alg_bytes / kern_s / 1e9 if kern_s > 0 else 0.0
        traffic = None
        tf = os.path.join(HERE, "profiles", "grav_walk_traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf))
                if rec.get("ng") == args.ng and world == 1:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "particle-steps/sec (gravity+SPH)",
            "value": value,
            "unit": "particle-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "c2: %d^3 DM + %d^3 gas periodic box, tree gravity (relative "
                            "criterion, ErrTolForceAcc=0.005) + Ewald correction + SPH density "
                            "(DesNumNgb=33+-2) + hydro, all particles active, tree rebuilt every "
                            "step" % (args.ng, args.ng),
                "n_particles": pr.n,
                "n_gas": pr.ngas,
                "parallelism": "targets sharded %d-way along the space-filling curve, sources "
                               "replicated, 3 all-gathers/step" % world,
            },
            "roofline": {
                "kernel": "k_grav_walk<NEWTON>",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": 1e3 * kern_s,
                "note": "kernel_ms is the kernel's duration inside the timed steps, where it runs "
                        "concurrently with the Ewald walk and the SPH kernels (DESIGN.md 4.3); "
                        "kernel_ms_alone / frac_alone: the same launch with the chip to itself",
                "kernel_ms_alone": iso_ms,
                "frac_alone": (alg_bytes / (1e-3 * iso_ms) / 1e9 / HBM_PEAK_GBS) if iso_ms > 0 else None,
                "interactions_per_particle": grav_int_all / K / pr.n,
            },
            "phases_ms_rank0": {k: v / K for k, v in phase_ms.items()},
            "work_per_step_rank0": {"grav_interactions": grav_int / K,
                                    "ewald_interactions": ewald_int / K,
                                    "dens_neighbours": dens_ngb / K,
                                    "dens_extra_iterations": dens_iter / K,
                                    "hydro_pairs": hyd_pairs / K},
        }

    if rank == 0 and world == 1:
        # "next" row N1, outside the timed region: the timestep + kick kernel on the same
        # resident state, against its own (HBM) roofline
        try:
            out["next_rows"] = {"N1_kick": kick_roofline(pr, fp, B), "N3_pm": pm_timing(pr, fp)}
        except Exception as e:   # never let an auxiliary measurement break the bench line
            out["next_rows"] = {"N1_kick": {"error": str(e)}}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pr, fp, B, args)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def kick_roofline(pr, fp, B, reps=5):
    """ghip_advance_timesteps (timestep.c: get_timestep + do_the_kick) for all particles of the
    workload, using the accelerations the timed steps left on the device.  Algorithmic bytes:
    92 B per collisionless particle (Type, GravAccel, Vel, TimeBin, Ti_begstep in; Vel, TimeBin,
    Ti_begstep out), 188 B per gas particle (+ HydroAccel, Hsml, MaxSignalVel, Entropy, DtEntropy
    in; VelPred, Entropy, DtEntropy out)."""
    import numpy as np
    P = B.KickParams()
    P.Ti_Current, P.Timebase_interval, P.ComovingIntegrationOn = 0, 1.0 / (1 << 29), 0
    P.Time, P.hubble_a = 1.0, 1.0
    P.ErrTolIntAccuracy, P.CourantFac = 0.025, 0.15
    P.MaxSizeTimestep, P.MinSizeTimestep, P.dt_displacement = 0.03, 0.0, 0.03
    for t in range(6):
        P.SofteningTable[t] = pr.force_soft[t] / 2.8
    P.MinEgySpec = 0.0
    P.TimeBinActive = 0xffffffff
    fp.set_field(B.F_TIMEBIN, np.full(pr.n, 20, np.int32))
    fp.set_field(B.F_TI_BEGSTEP, np.zeros(pr.n, np.int32))
    ms = []
    for _ in range(reps):
        fp.set_field(B.F_TI_BEGSTEP, np.zeros(pr.n, np.int32))
        cnt, _sph = fp.advance_timesteps(P)
        ms.append(fp.stats()["ms_kick"])
    nbytes = 92.0 * (pr.n - pr.ngas) + 188.0 * pr.ngas
    t = float(np.median(ms)) * 1e-3
    ach = nbytes / t / 1e9
    return {"kernel": "k_advance_timesteps", "bound": "hbm", "achieved": ach,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "kernel_ms": 1e3 * t, "algorithmic_bytes_per_launch": nbytes,
            "timebins_populated": int((cnt > 0).sum())}


def pm_timing(pr, fp, pmgrid=128, reps=5):
    """ghip_pm_periodic (pmforce_periodic: CIC deposit, hipFFT, Green's function, gradient, CIC
    interpolation) for all particles of the workload on a PMGRID^3 mesh.  Algorithmic bytes: 32 B
    in + 8 x 8 B mesh updates + 24 x 8 B mesh reads + 24 B out per particle, and per mesh point the
    FFT pair (2 x 3 passes x 16 B), Green's function (12 B), gradient (5 x 8 B in, 24 B out)."""
    import numpy as np
    ms = []
    for _ in range(reps):
        fp.pm_periodic(pmgrid, pr.box, pr.G)
        ms.append(fp.stats()["ms_pm"])
    t = float(np.median(ms)) * 1e-3
    nbytes = pr.n * (32.0 + 64.0 + 192.0 + 24.0) + float(pmgrid) ** 3 * (96.0 + 12.0 + 64.0)
    ach = nbytes / t / 1e9
    return {"kernel": "ghip_pm_periodic", "pmgrid": pmgrid, "bound": "hbm", "achieved": ach,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "ms": 1e3 * t,
            "algorithmic_bytes": nbytes}


def cpu_baseline(pr, fp, B, args):
    """The CPU restatement of the same step (oracle/, kind "port"), timed on the host cores of
    this box on the FULL workload.  The converged smoothing lengths and OldAcc of the device run
    are the starting state, so both sides do the same work (same interaction sets)."""
    from oracle import oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = args.cpu_threads if args.cpu_threads > 0 else min(16, avail)
    O.set_num_threads(threads)
    n, ng = pr.n, pr.ngas
    hs = fp.get_field(B.F_HSML)
    oldacc = fp.get_field(B.F_OLDACC)
    tab = np.ascontiguousarray(fp.ewald_table())      # identical to the oracle's to 1e-12
    tg = np.arange(n, dtype=np.int32)
    act = np.arange(ng, dtype=np.int32)
    t0 = time.perf_counter()
    T = pr.oracle_tree(hsml=hs)
    acc, cost = T.gravity(pr.o_grav(0.0), tg, oldacc)
    T.gravity_ewald_add(pr.o_grav(0.0), tab, tg, oldacc, acc, cost)
    od = T.density(pr.o_dens(), act, pr.velpred, pr.entropy, pr.dtentropy, pr.timebin,
                   pr.ti_begstep, hs)
    T.update_hmax(act, od["hsml"], od["divvel"])
    T.hydro(pr.o_hydro(), act, pr.velpred, od["hsml"], od["density"], od["pressure"],
            od["dhsmlfac"], od["divvel"], od["curlvel"], pr.timebin)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "particle-steps/s", "cores": threads, "kind": "port",
            "sample": "one full step of the same %d-particle workload (tree build + Newtonian "
                      "+ Ewald walks + density + hydro), %.2f s wall on %d OpenMP threads"
                      % (n, dt, threads),
            "interactions_per_particle": float(cost.mean())}


if __name__ == "__main__":
    main()
