#!/usr/bin/env python3
"""Headline benchmark: particle-steps/s of the per-step force path (tree gravity + Ewald + SPH
density + hydro) on BASELINE.json's config c2 (64^3 DM + 64^3 gas periodic box), on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of run.c's loop over the resident particle set with everybody active
(a full step, run.c:293-296): drift to the sync point (predict.c:129), tree build, Newtonian walk with
the relative opening criterion + Ewald-correction walk, SPH density (h iteration), hmax refresh,
SPH hydro, the OldAcc / G post-pass -- exactly the work of compute_accelerations() (accel.c:61-106)
-- and the timestep + kick (timestep.c:29).  The state ADVANCES between steps: smoothing lengths
really iterate and the wavefront plan of the walks is one step stale.  Inputs are resident in HBM
when the timed region starts.

N > 1 (one process per GPU, strong scaling of the same workload): Peano-Hilbert domain
decomposition; every step migrates the particles that left their shard, exchanges locally essential
trees and SPH ghosts over RCCL from C (ghip_dd_* in include/ghip.h).  torch.distributed (gloo) is the
control plane only: it carries the 128-byte RCCL id once and the barriers / reductions of this
script.  `--mode replicated` selects the round-1 design (all sources on every rank, results
all-gathered by torch.distributed over RCCL).

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
FP64_VALU_PEAK_TFLOPS = 78.6    # 256 CU x 4 SIMD x 2.4 GHz x 32 flop/clk (one wave64 fp64 FMA per 4 cycles)
N_SIMD = 1024
MAX_CLOCK_HZ = 2.4e9
VALU_CYCLES_PER_INST = 4        # an fp64 (or any full-rate 64-bit) VALU wave-instruction holds its SIMD 4 cycles
BYTES_PER_INTERACTION = 32.0    # s[3] + mass, fp64
BYTES_PER_GRAV_TARGET = 68.0    # 40 B in + 28 B out
FLOP_PER_INTERACTION = 38.0     # ~25 flop + sqrt + div
DT_STEP = 4.0e-3                # the common particle step of the workload (code units)
BIN = 20                        # its time bin: Timebase_interval = DT_STEP / 2^BIN


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ng", type=int, default=64, help="particles per dimension per species")
    ap.add_argument("--mode", choices=["dd", "replicated"], default="dd",
                    help="N > 1: domain decomposition (default) or replicated sources")
    ap.add_argument("--frozen", action="store_true",
                    help="re-evaluate one frozen state (no drift / kick between steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args()


class Timeline:
    """run.c's integer timeline for a workload in which MaxSizeTimestep binds: everybody shares
    time bin BIN, every sync point is a full step."""

    def __init__(self, B, pr):
        self.B, self.pr = B, pr
        self.tb = DT_STEP / (1 << BIN)
        self.ti = 0
        P = B.KickParams()
        P.Timebase_interval, P.ComovingIntegrationOn = self.tb, 0
        P.Time, P.hubble_a = 1.0, 1.0
        P.ErrTolIntAccuracy, P.CourantFac = 1.0e3, 1.0e3      # the criteria do not bind ...
        P.MaxSizeTimestep, P.MinSizeTimestep = DT_STEP, 0.0   # ... MaxSizeTimestep does
        P.dt_displacement = 1.0
        for t in range(6):
            P.SofteningTable[t] = pr.force_soft[t] / 2.8
        P.MinEgySpec = 0.0
        self.kick = P

    def kick_params(self):
        self.kick.Ti_Current = self.ti
        self.kick.TimeBinActive = sum(1 << b for b in range(30) if self.ti % (1 << b) == 0)
        return self.kick

    def next_sync_point(self):
        self.ti += 1 << BIN
        return self.ti


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    # stdout carries ONE JSON line: everything libraries print on fd 1 meanwhile (gloo's connection
    # banner, ...) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # (the host driver of this pool supports dmabuf IPC only: RCCL's peer buffers need this)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the force path has no CPU fallback")
    ndev = torch.cuda.device_count()
    local_rank = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    replicated = world > 1 and args.mode == "replicated"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if replicated and os.environ.get("BENCH_BACKEND", "nccl") == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from common import Problem  # seeded workload definition shared with the parity tests
    B = importlib.import_module("gadget-leicester_amd.bindings")
    S = importlib.import_module("gadget-leicester_amd.sharded")

    pr = Problem(ng=args.ng, gas=True, periodic=1)
    n_total, ngas_total = pr.n, pr.ngas
    tl = Timeline(B, pr)
    pr.timebase = tl.tb
    # The domain cube (DomainCorner / DomainCenter / DomainLen).  The reference recomputes the particles'
    # extent at every decomposition (domain.c:1972-2014); a resident loop keeps one cube for the run, and
    # in a periodic box that cube is the box -- every wrapped position lies inside it, whatever drifts.
    pr.extent = (np.zeros(3), np.full(3, 0.5 * pr.box), pr.box)
    tree_args = (pr.extent[0], pr.extent[1], pr.extent[2], pr.force_soft)
    gp_bh, gp_rel = pr.g_grav(pr.theta), pr.g_grav(0.0)
    hp = pr.g_hydro()
    zeros_i = np.zeros(pr.n, np.int32)
    pr.timebin, pr.ti_begstep = zeros_i.copy(), zeros_i.copy()   # run.c: everybody starts in bin 0

    dom = None
    rep = None
    parallelism = "1 GPU"
    if world == 1:
        fp = pr.device(local_rank)
    elif replicated:
        fp = pr.device(local_rank)
        rep = S.ShardedForceStep(fp, rank, world, dist=dist, device=device)
        parallelism = ("targets sharded %d-way along the space-filling curve, sources replicated, "
                       "3 all-gathers/step (torch.distributed)" % world)
    else:
        # cut the Peano-Hilbert curve (same arithmetic on every rank), keep this rank's range
        probe = B.ForcePath(local_rank)
        probe.set_counts(pr.n, 0)
        probe.set_field(B.F_POS, pr.ic["pos"])
        probe.dd_init(0, 1)
        probe.dd_set_domain(*tree_args)
        keys = probe.dd_keys()
        probe.close()
        splits, owner = S.decompose(keys, world)
        mine = np.where(owner == rank)[0]
        gid = np.concatenate([mine[mine < pr.ngas], mine[mine >= pr.ngas]])
        ngl = int((mine < pr.ngas).sum())
        fp = B.ForcePath(local_rank)
        fp.set_counts(len(gid), ngl)
        ic = pr.ic
        for fid, arr in ((B.F_POS, ic["pos"]), (B.F_VEL, ic["vel"]), (B.F_MASS, ic["mass"]),
                         (B.F_TYPE, ic["type"]), (B.F_HSML, pr.hsml0), (B.F_TIMEBIN, pr.timebin),
                         (B.F_TI_BEGSTEP, pr.ti_begstep)):
            fp.set_field(fid, arr[gid])
        fp.set_field(B.F_ID, gid.astype(np.int32))
        gg = gid[:ngl]
        fp.set_field(B.F_VELPRED, pr.velpred[gg])
        fp.set_field(B.F_ENTROPY, pr.entropy[gg])
        fp.set_field(B.F_DTENTROPY, pr.dtentropy[gg])

        def bcast(obj):
            box = [obj]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        transport = os.environ.get("BENCH_TRANSPORT", "rccl")

        def allgather_bytes(data):
            # rehearsal transport (several ranks on one GPU): gloo all-gather of host bytes
            t = torch.frombuffer(bytearray(data), dtype=torch.uint8)
            out = torch.empty(world * len(data), dtype=torch.uint8)
            dist.all_gather_into_tensor(out, t)
            return out.numpy().tobytes()

        rccl_error = None
        if transport == "rccl":
            # ncclCommInitRank is collective: nobody may enter it unless every rank has a librccl to
            # enter it with -- agreed over gloo BEFORE the connect attempt
            have = torch.tensor([1.0 if B.dd_rccl_library() else 0.0], dtype=torch.float64)
            dist.all_reduce(have, op=dist.ReduceOp.MIN)
            if float(have.item()) < 0.5:
                transport = "host"
                rccl_error = "librccl could not be loaded on every rank"
                sys.stderr.write("bench: %s: exchanges staged through the host\n" % rccl_error)
        if transport == "rccl":
            # every rank must end up on the same transport: agree after the connect attempt
            try:
                dom = S.DomainRank(fp, rank, world, bcast, transport="rccl")
                ok = 1.0
            except Exception as e:   # noqa: BLE001 -- reported in the JSON line
                rccl_error, ok = str(e), 0.0
            t = torch.tensor([ok], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if float(t.item()) < 0.5:
                transport = "host"
                rccl_error = rccl_error or "another rank could not connect"
                sys.stderr.write("bench: RCCL unavailable (%s): exchanges staged through the host\n"
                                 % rccl_error)
        if transport != "rccl":
            dom = S.DomainRank(fp, rank, world, bcast, transport="host", allgather=allgather_bytes)
        fp.dd_set_domain(*tree_args)
        fp.dd_set_splits(splits)
        parallelism = ("Peano-Hilbert domain decomposition over %d GPUs: migration, locally "
                       "essential trees and SPH ghosts over RCCL (%s), one merged tree per rank"
                       % (world, "NOT AVAILABLE (%s): exchanges staged through the host + gloo"
                          % (rccl_error or "rehearsal") if transport != "rccl"
                          else "lib: " + B.dd_rccl_library()))
    for fid in (B.F_TIMEBIN, B.F_TI_BEGSTEP, B.F_TI_CURRENT):
        fp.set_field(fid, np.zeros(fp.n, np.int32))
    fp.set_field(B.F_OLDACC, np.zeros(fp.n))

    def sync():
        fp.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def allreduce(vals, op):
        t = torch.tensor(vals, dtype=torch.float64)
        if world > 1:
            if dist.get_backend() == "nccl":
                t = t.to(device)
            dist.all_reduce(t, op=op)
        return [float(v) for v in t.cpu()]

    walks = [B.WALK_NEWTON, B.WALK_EWALD]

    def force_step(gp):
        """compute_accelerations() + the post-pass on the resident state"""
        dp = pr.g_dens()
        dp.Ti_Current, dp.Timebase_interval = tl.ti, tl.tb
        if dom is not None:
            dom.gravity(gp, B.WALK_NEWTON_EWALD)
            dom.density(dp)
            fp.update_hmax()
            dom.hydro(hp)
            fp.gravity_finish(pr.G)
        elif rep is not None:
            rep.step(tree_args, gp, dp, hp, pr.G, walks)
        else:
            fp.tree_build(*tree_args)
            fp.gravity(gp, B.WALK_NEWTON_EWALD)
            fp.density(dp)
            fp.update_hmax()
            fp.hydro(hp)
            fp.gravity_finish(pr.G)

    def step(gp):
        if not args.frozen:
            fp.drift(tl.next_sync_point(), tl.tb, box_wrap=True, boxsize=pr.box)
            if dom is not None:
                dom.migrate()
        force_step(gp)
        if not args.frozen:
            # (everybody shares one bin in this workload: the host's timeline needs no recount)
            fp.advance_timesteps(tl.kick_params(), counts=False)

    # The host never waits for the device inside a step, except where the h iteration reads its
    # unconverged count (underneath the gravity walks): drift and kick report errors at the next
    # synchronisation, the tree build is enqueued without its node counts (include/ghip.h,
    # ghip_set_async), statistics are kept per step on the device and read once after the run.
    fp.set_async(True)

    # step 0 of a run (accel.c:61-68): Barnes-Hut pass to obtain OldAcc, then relative criterion;
    # advance_and_find_timesteps puts everybody into bin BIN
    force_step(gp_bh)
    force_step(gp_rel)
    if not args.frozen:
        fp.advance_timesteps(tl.kick_params())
    for _ in range(args.warmup):
        step(gp_rel)
    sync()

    migrated = 0
    dd_bytes = [0, 0, 0]
    fp.run_begin(max(args.steps, 1))
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fp.step_begin()
        step(gp_rel)
        fp.step_end()
        if dom is not None:
            i = fp.dd_info()     # host-side bookkeeping of the exchanges, no device access
            migrated += i["migrated_out"]
            dd_bytes[0] += i["bytes_migrate"]
            dd_bytes[1] += i["bytes_gravity"]
            dd_bytes[2] += i["bytes_density"]
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = allreduce([elapsed], dist.ReduceOp.MAX)[0]
    K = max(args.steps, 1)
    rs = fp.run_end()            # per-step device events and counters, summed over the run
    st = fp.stats()
    phase_ms = {k: rs["ms_" + k] for k in ("tree", "grav", "ewald", "dens", "hmax", "hydro")}
    work = {k: rs[k] for k in ("grav_interactions", "ewald_interactions", "dens_neighbours",
                               "hydro_pairs", "grav_wave_steps", "ewald_wave_steps")}
    work["dens_iterations"] = rs["dens_extra_iterations"]
    work["grav_targets"] = st["grav_targets"]
    tot = [work["grav_interactions"], work["ewald_interactions"], work["dens_neighbours"],
           work["hydro_pairs"], fp.n]
    if world > 1:
        tot = allreduce(tot, dist.ReduceOp.SUM)
        if rep is not None:
            tot[4] = n_total
    grav_int_all = float(tot[0])
    assert int(round(tot[4])) == n_total, "particles lost or doubled by the decomposition"

    # the dominant kernel alone (outside the timed region): in the timed steps it shares the chip
    # with the Ewald walk and the SPH kernels, which stretches its own duration
    iso_ms, iso_steps, iso_int = None, None, None
    if world == 1:
        iso, isteps = [], []
        for _ in range(3):
            fp.gravity(gp_rel, B.WALK_NEWTON)
            s2 = fp.stats()
            iso.append(s2["ms_grav"])
            isteps.append(s2["grav_wave_steps"])
            iso_int = s2["grav_interactions"]
        iso_ms, iso_steps = float(np.median(iso)), float(np.median(isteps))

    # a rank's share of a step phase by phase (outside the timed region; the device drained between
    # phases, so the numbers add up -- unlike the overlapping spans of the timed steps)
    shard_phases = None
    if dom is not None:
        for _ in range(2):
            dp = pr.g_dens()
            if not args.frozen:
                fp.drift(tl.next_sync_point(), tl.tb, box_wrap=True, boxsize=pr.box)
            dp.Ti_Current, dp.Timebase_interval = tl.ti, tl.tb
            rows = []
            c, x = dom.timed(B.DD_MIGRATE, None)
            rows += [("migrate", sum(c)), ("migrate: exchange", sum(x))]
            c, x = dom.timed(B.DD_GRAVITY, gp_rel, B.WALK_NEWTON_EWALD)
            names = ["gravity: own tree + groups", "gravity: LET selection + packing",
                     "gravity: merged tree + both walks"]
            rows += [(names[i] if i < 3 else "gravity: phase %d" % i, v) for i, v in enumerate(c)]
            rows.append(("gravity: exchanges", sum(x)))
            c, x = dom.timed(B.DD_DENSITY, dp)
            names = ["density: gas groups", "density: ghost selection",
                     "density: gas tree + h iteration", "density: growth check + refresh packing",
                     "density: refresh of the ghosts"]
            rows += [(names[i] if i < 5 else "density: phase %d" % i, v) for i, v in enumerate(c)]
            rows.append(("density: exchanges", sum(x)))
            fp.sync()
            t1 = time.perf_counter()
            fp.update_hmax()
            fp.sync()
            rows.append(("force_update_hmax", 1e3 * (time.perf_counter() - t1)))
            c, x = dom.timed(B.DD_HYDRO, hp)
            rows += [("hydro", sum(c)), ("hydro: exchanges", sum(x))]
            fp.gravity_finish(pr.G)
            if not args.frozen:
                fp.advance_timesteps(tl.kick_params(), counts=False)
            fp.sync()
        vals = [v for _, v in rows]
        vmax = allreduce(vals, dist.ReduceOp.MAX)
        vsum = allreduce(vals, dist.ReduceOp.SUM)
        shard_phases = {nm: {"max": a, "mean": b / world} for (nm, _), a, b in zip(rows, vmax, vsum)}
        shard_phases["sum_of_slowest"] = sum(vmax)

    ms_per_step = 1e3 * elapsed / K
    value = n_total * K / elapsed

    out = None
    if rank == 0:
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
                "workload": "c2: %d^3 DM + %d^3 gas periodic box, every step: drift, tree build, "
                            "tree gravity (relative criterion, ErrTolForceAcc=0.005) + Ewald "
                            "correction, SPH density (DesNumNgb=33+-2, h iterated) + hydro, OldAcc/G "
                            "post-pass, timestep + kick; all particles active (full steps of %g), "
                            "state %s between steps" % (args.ng, args.ng, DT_STEP,
                                                        "frozen" if args.frozen else "advancing"),
                "n_particles": n_total,
                "n_gas": ngas_total,
                "parallelism": parallelism,
            },
            # how much of a step the host is responsible for: wall time per step minus the device's
            # own span of a step (step-begin mark -> step-end mark on the device clock)
            "host_gap_ms": ms_per_step - rs["ms_steps_device"] / max(rs["steps_timed"], 1),
            "device_ms_per_step": rs["ms_steps_device"] / max(rs["steps_timed"], 1),
            "device_idle_between_steps_ms": rs["ms_between_steps"] / max(rs["steps_timed"], 1),
            "launches_per_step": rs["launches"] / K,
            "blocking_syncs_per_step": rs["blocking_syncs"] / K,
            # work-normalised throughput (the advancing state changes the work per step)
            "interactions_per_s": (tot[0] + tot[1]) / elapsed,
            "roofline": walk_roofline(work, phase_ms, K, st, iso_ms, iso_steps, args, world,
                                      grav_int_all, n_total, iso_int),
            "phases_ms_rank0": {k: v / K for k, v in phase_ms.items()},
            "work_per_step_rank0": {"grav_interactions": work["grav_interactions"] / K,
                                    "ewald_interactions": work["ewald_interactions"] / K,
                                    "dens_neighbours": work["dens_neighbours"] / K,
                                    "dens_extra_iterations": work["dens_iterations"] / K,
                                    "hydro_pairs": work["hydro_pairs"] / K,
                                    "grav_wave_steps": work["grav_wave_steps"] / K,
                                    "ewald_wave_steps": work["ewald_wave_steps"] / K},
        }
        if dom is not None:
            out["transport"] = transport     # "rccl": RCCL from C (ghip_dd_run); "host": staged through gloo
            out["rccl_library"] = B.dd_rccl_library() if transport == "rccl" else None
            out["shard_phases_ms"] = shard_phases
            out["exchange_per_step_rank0"] = {
                "particles_migrated": migrated / K,
                "bytes_sent_migration": dd_bytes[0] / K,
                "bytes_sent_tree_nodes": dd_bytes[1] / K,
                "bytes_sent_ghosts": dd_bytes[2] / K,
                "note": "bytes this rank sent over links per step (all-gathered group tables "
                        "included); tree nodes = 64-B elements of the locally essential trees, "
                        "ghosts = 128-B gas records sent twice (before density, before hydro)"}

    if rank == 0 and world == 1:
        # "next" rows outside the timed region, against their own (HBM) rooflines
        try:
            out["next_rows"] = {"N1_kick": kick_roofline(pr, fp, B), "N3_pm": pm_timing(pr, fp)}
        except Exception as e:   # never let an auxiliary measurement break the bench line
            out["next_rows"] = {"N1_kick": {"error": str(e)}}
    if rank == 0 and world == 1 and not args.no_dropin:
        try:
            state = device_state(pr, fp, B, tl)
            out["dropin_ms_per_step"] = dropin_timing(pr, state)
            out["dropin_ms_per_step"]["with_overlap_sph"] = dropin_timing(pr, state, overlap_sph=1)
            out["dropin_ms_per_step"]["with_overlap_sph_and_pin_records"] = dropin_timing(
                pr, state, overlap_sph=1, pin_records=1)
            del state
        except Exception as e:
            out["dropin_ms_per_step"] = {"error": str(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        pr.ti_current = tl.ti

        def device_eval():
            # one more force evaluation of the state the run reached, nothing advanced
            dp = pr.g_dens()
            dp.Ti_Current, dp.Timebase_interval = tl.ti, tl.tb
            fp.tree_build(*tree_args)
            fp.gravity(gp_rel, B.WALK_NEWTON_EWALD)
            fp.density(dp)
            fp.update_hmax()
            fp.hydro(hp)

        out["cpu_baseline"] = cpu_baseline(pr, fp, B, args, device_eval)

    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def walk_roofline(work, phase_ms, K, st, iso_ms, iso_steps, args, world, grav_int_all, n_total,
                  iso_int=None):
    """Dominant kernel: k_grav_walk<NEWTON>.  It is irregular fp64 pairwise work out of the caches
    (the walk records of c2 fit the L2 / Infinity Cache), so the bound that can actually bind is the
    fp64 VALU issue rate, not HBM:
      frac             = VALU wave-instructions x 4 cycles / (1024 SIMDs x 2.4 GHz x kernel time):
                         <= 1 by construction.  The instruction count is the live element-visit
                         counter of this run x the VALU instructions per element visit of this
                         binary, measured with rocprofv3 --pmc SQ_INSTS_VALU (profiles/*_walk_pmc.json)
      achieved / peak  = the same number expressed in TFLOP/s (a wave-instruction slot = one fp64 FMA
                         over 64 lanes = 128 flop) against the fp64 vector peak
      flop_frac        = algorithmic flops (38 per interaction) / time / peak
      algorithmic_frac = SURVEY 8(d)'s 32 B per interaction + 68 B per target / time / HBM peak -- NOT a
                         physical bound (one 64-B scalar load feeds 64 lanes), kept for continuity
      hbm_frac         = counter traffic (2 x FETCH_SIZE + WRITE_SIZE, per launch) / time / HBM peak"""
    grav_int = work["grav_interactions"] / K
    my_targets = st["grav_targets"]
    alg_bytes = BYTES_PER_INTERACTION * grav_int + BYTES_PER_GRAV_TARGET * my_targets
    kern_s = 1e-3 * phase_ms["grav"] / K
    wave_steps = work["grav_wave_steps"] / K
    pmc = None
    for tag in ("r03", "r02", "r01"):
        f = os.path.join(HERE, "profiles", "%s_walk_valu.json" % tag)
        if os.path.exists(f):
            try:
                pmc = json.load(open(f))
                break
            except Exception:
                pmc = None
    ipw = pmc.get("valu_insts_per_wave_step") if pmc else None
    traffic = None
    if pmc and pmc.get("ng") == args.ng and world == 1:
        traffic = pmc.get("hbm_bytes_per_launch")

    def issue_frac(steps, seconds):
        if not ipw or not seconds:
            return None
        return steps * ipw * VALU_CYCLES_PER_INST / (N_SIMD * MAX_CLOCK_HZ * seconds)

    frac = issue_frac(wave_steps, kern_s)
    r = {
        "kernel": "k_grav_walk<NEWTON>",
        "bound": "fp64-valu",
        "achieved": (frac * FP64_VALU_PEAK_TFLOPS) if frac is not None else None,
        "peak": FP64_VALU_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": frac,
        "traffic": traffic,
        "kernel_ms": 1e3 * kern_s,
        "valu_insts_per_element_visit": ipw,
        "element_visits_per_launch": wave_steps,
        "flop_frac": (FLOP_PER_INTERACTION * grav_int / kern_s / 1e12 / FP64_VALU_PEAK_TFLOPS)
        if kern_s > 0 else None,
        "algorithmic_bytes_per_launch": alg_bytes,
        "algorithmic_frac": (alg_bytes / kern_s / 1e9 / HBM_PEAK_GBS) if kern_s > 0 else None,
        "hbm_frac": (traffic / kern_s / 1e9 / HBM_PEAK_GBS) if (traffic and kern_s > 0) else None,
        "interactions_per_particle": grav_int_all / K / n_total,
        "ps_per_interaction": (1e12 * kern_s / grav_int) if grav_int else None,
        "lanes_interacting_per_visit": grav_int / (64.0 * wave_steps) if wave_steps else None,
        "note": "kernel_ms is the kernel's duration inside the timed steps, where it shares the chip "
                "with the Ewald walk and the SPH kernels (DESIGN.md 4.3); *_alone: the same launch "
                "with the chip to itself.  frac is an issue-slot fraction (<= 1 by construction); "
                "algorithmic_frac is SURVEY 8(d)'s byte model and is not a physical bound.",
    }
    if iso_ms:
        s = 1e-3 * iso_ms
        r["kernel_ms_alone"] = iso_ms
        r["ps_per_interaction_alone"] = (1e12 * s / iso_int) if iso_int else None
        r["interactions_alone"] = iso_int
        r["frac_alone"] = issue_frac(iso_steps, s)
        r["algorithmic_frac_alone"] = alg_bytes / s / 1e9 / HBM_PEAK_GBS
        r["hbm_frac_alone"] = (traffic / s / 1e9 / HBM_PEAK_GBS) if traffic else None
    return r


def device_state(pr, fp, B, tl):
    """What the timed steps left on the device, as host arrays (for the drop-in leg)."""
    names = {"pos": B.F_POS, "vel": B.F_VEL, "oldacc": B.F_OLDACC, "hsml": B.F_HSML,
             "velpred": B.F_VELPRED, "entropy": B.F_ENTROPY, "dtentropy": B.F_DTENTROPY,
             "timebin": B.F_TIMEBIN, "ti_begstep": B.F_TI_BEGSTEP}
    st = {k: fp.get_field(f) for k, f in names.items()}
    st["ti_current"], st["timebase"] = tl.ti, tl.tb
    return st


def dropin_timing(pr_bench, state, reps=3, overlap_sph=0, pin_records=0):
    """The PCIe-inclusive drop-in path accel.c would see: gravity_tree(), density(),
    force_update_hmax(), hydro_force() of libgadget_force.so on AoS P[]/SphP[] records, each with its
    H2D / D2H of the record blocks (SURVEY 8d's metric "including host<->device packing"), on the
    particle state the timed steps reached (same interaction sets as `value`'s last step).  Never
    part of `value`."""
    from common import Problem
    from test_gpu_parity import _host_problem
    H = importlib.import_module("gadget-leicester_amd.hostapi")
    pr = Problem(ng=round((pr_bench.n // 2) ** (1 / 3)), gas=True, periodic=1)
    host, P, S = _host_problem(pr, H, 1, overlap_sph=overlap_sph, pin_records=pin_records)
    ng = pr.ngas
    P["Pos"], P["Vel"], P["OldAcc"] = state["pos"], state["vel"], state["oldacc"]
    P["TimeBin"], P["Ti_begstep"] = state["timebin"], state["ti_begstep"]
    S["VelPred"], S["Entropy"], S["DtEntropy"] = state["velpred"], state["entropy"], state["dtentropy"]
    S["Hsml"] = state["hsml"][:ng]
    host.All.Ti_Current, host.All.Timebase_interval = state["ti_current"], state["timebase"]
    host.domain()
    L = host.L
    host.All.ErrTolTheta = 0                     # relative criterion on the run's OldAcc
    L.gravity_tree()
    L.gadget_force_flush()
    best = None
    for _ in range(reps):
        host.All.ErrTolTheta = 0
        t = [time.perf_counter()]
        for f in (L.gravity_tree, L.density, L.force_update_hmax, L.hydro_force):
            f()
            t.append(time.perf_counter())
        d = [1e3 * (b - a) for a, b in zip(t, t[1:])]
        if best is None or sum(d) < sum(best):
            best = d
    ok = host.endrun_codes == []
    host.close()
    return {"total": sum(best), "gravity_tree": best[0], "density": best[1],
            "force_update_hmax": best[2], "hydro_force": best[3], "ok": ok,
            "particle_steps_per_s": pr.n / (1e-3 * sum(best)),
            "overlap_sph": overlap_sph, "pin_records": pin_records,
            "note": "host calls the four drivers one after the other on 112-B / 184-B records in the "
                    "state the timed steps reached; each uploads / downloads the record blocks over PCIe"
                    + ("; gadget_force_config.overlap_sph: the gravity walks stay in flight underneath "
                       "density / hydro_force and their results arrive with hydro_force()"
                       if overlap_sph else "")
                    + ("; gadget_force_config.pin_records: P[] / SphP[] page-locked once"
                       if pin_records else "")}


def kick_roofline(pr, fp, B, reps=5):
    """ghip_advance_timesteps (timestep.c: get_timestep + do_the_kick) for all particles of the
    workload, using the accelerations the timed steps left on the device.  Algorithmic bytes:
    92 B per collisionless particle (Type, GravAccel, Vel, TimeBin, Ti_begstep in; Vel, TimeBin,
    Ti_begstep out), 188 B per gas particle (+ HydroAccel, Hsml, MaxSignalVel, Entropy, DtEntropy
    in; VelPred, Entropy, DtEntropy out)."""
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


def cpu_baseline(pr, fp, B, args, device_eval=None):
    """The CPU restatement of the force path (oracle/, kind "port"), timed on the host cores of
    this box on the FULL workload in the state the device run left it: same positions, smoothing
    lengths and OldAcc, so both sides do the same work (same interaction sets).  With device_eval the
    device evaluates the same state once more and BASELINE's accuracy metric (max |da| / |a| over all
    particles) is reported next to the timing."""
    from oracle import oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = args.cpu_threads if args.cpu_threads > 0 else min(16, avail)
    O.set_num_threads(threads)
    n, ng = pr.n, pr.ngas
    pos = fp.get_field(B.F_POS)
    vel = fp.get_field(B.F_VEL)
    hs = fp.get_field(B.F_HSML)
    oldacc = fp.get_field(B.F_OLDACC)
    velpred = fp.get_field(B.F_VELPRED)
    entropy = fp.get_field(B.F_ENTROPY)
    dtentropy = fp.get_field(B.F_DTENTROPY)
    timebin = fp.get_field(B.F_TIMEBIN)
    tibeg = fp.get_field(B.F_TI_BEGSTEP)
    tab = np.ascontiguousarray(fp.ewald_table())      # identical to the oracle's to 1e-12
    dev = None
    if device_eval is not None:
        device_eval()
        dev = {"acc": fp.get_field(B.F_GRAVACCEL), "cost": fp.get_field(B.F_GRAVCOST),
               "density": fp.get_field(B.F_DENSITY), "hsml": fp.get_field(B.F_HSML)[:ng],
               "hydro": fp.get_field(B.F_HYDROACCEL), "dtentropy": fp.get_field(B.F_DTENTROPY)}
    tg = np.arange(n, dtype=np.int32)
    act = np.arange(ng, dtype=np.int32)
    t0 = time.perf_counter()
    T = O.Tree(pos, vel, pr.ic["mass"], pr.ic["type"], pr.force_soft, hsml=hs, extent=pr.extent)
    acc, cost = T.gravity(pr.o_grav(0.0), tg, oldacc)
    T.gravity_ewald_add(pr.o_grav(0.0), tab, tg, oldacc, acc, cost)
    od = T.density(pr.o_dens(), act, velpred, entropy, dtentropy, timebin, tibeg, hs)
    T.update_hmax(act, od["hsml"], od["divvel"])
    oh = T.hydro(pr.o_hydro(), act, velpred, od["hsml"], od["density"], od["pressure"],
                 od["dhsmlfac"], od["divvel"], od["curlvel"], timebin)
    dt = time.perf_counter() - t0
    accuracy = None
    if dev is not None:
        def vrel(a, b):
            return float((np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)).max())
        tot_d, tot_o = pr.G * dev["acc"], pr.G * acc
        tot_d[:ng] += dev["hydro"]
        tot_o[:ng] += oh["hydroaccel"][:ng]
        accuracy = {
            "max_rel_dacc_gravity": vrel(dev["acc"], acc),
            "max_rel_dacc_total": vrel(tot_d, tot_o),
            "max_rel_density": float(np.abs(dev["density"] / od["density"][:ng] - 1).max()),
            "max_rel_hsml": float(np.abs(dev["hsml"] / od["hsml"][:ng] - 1).max()),
            "max_dtentropy_over_max": float(np.abs(dev["dtentropy"] - oh["dtentropy"][:ng]).max() /
                                            np.abs(oh["dtentropy"][:ng]).max()),
            "interaction_counts_equal": bool(np.array_equal(dev["cost"], cost)),
            "target": "max |da|/|a| < 1e-5 (BASELINE north star); all %d particles, device vs the "
                      "CPU port on the same state" % n}
    return {"value": n / dt, "unit": "particle-steps/s", "cores": threads, "kind": "port",
            "accuracy_vs_cpu": accuracy,
            "sample": "one full force step (tree build + Newtonian + Ewald walks + density + hydro) "
                      "of the same %d-particle workload in the state the device run reached, "
                      "%.2f s wall on %d OpenMP threads" % (n, dt, threads),
            "interactions_per_particle": float(cost.mean())}


if __name__ == "__main__":
    main()
