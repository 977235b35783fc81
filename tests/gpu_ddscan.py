"""Development aid: what a rank's share of a domain-decomposed step costs on one GPU.  The shards of
a P-way run are executed one after the other (logical shards, device-to-device exchanges); every
compute phase of every shard is timed with the device idle otherwise.  A P-GPU step takes at least
sum over phases of the slowest shard's phase (the ranks meet at every exchange) plus the exchanges.
  python tests/gpu_ddscan.py [ng] [P ...]"""
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common import Problem, ShardSet, bindings   # noqa: E402


def run_op(S, op, params, walk=0):
    B = S.B
    for fp in S.fp:
        fp.dd_begin(op, params, walk)
    phases = []
    while True:
        ts, rcs = [], []
        for fp in S.fp:
            fp.sync()
            t0 = time.perf_counter()
            rcs.append(fp.dd_step())
            fp.sync()
            ts.append(1e3 * (time.perf_counter() - t0))
        phases.append(ts)
        if rcs[0] == 0:
            break
        B.dd_exchange_local(S.fp)
    return phases


def main():
    ng = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    Ps = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
    B = bindings()
    pr = Problem(ng=ng, gas=True, periodic=1)
    base = None
    balance = os.environ.get("DDSCAN_BALANCE", "0") == "1"
    for P in Ps:
        work = None
        if balance and P > 1:
            # domain.c:378-384: cut the curve by cumulative (1 + GravCost) of a first step
            S0 = ShardSet(pr, P)
            run_op(S0, B.DD_GRAVITY, pr.g_grav(pr.theta), B.WALK_NEWTON_EWALD)
            S0.each(lambda fp: fp.gravity_finish(pr.G))
            run_op(S0, B.DD_GRAVITY, pr.g_grav(0.0), B.WALK_NEWTON_EWALD)
            work = 1.0 + S0.get_field(B.F_GRAVCOST).astype(np.float64)
            S0.close()
        S = ShardSet(pr, P, work=work)
        # step 0: Barnes-Hut pass for OldAcc, density to converge h; then the measured step
        run_op(S, B.DD_GRAVITY, pr.g_grav(pr.theta), B.WALK_NEWTON_EWALD)
        S.each(lambda fp: fp.gravity_finish(pr.G))
        run_op(S, B.DD_DENSITY, pr.g_dens())
        for rep in range(int(os.environ.get("DDSCAN_REPS", "20"))):   # (the pair's balance needs a re-probe of the other setting, 16 pairs)
            mig = run_op(S, B.DD_MIGRATE, None)
            g = run_op(S, B.DD_GRAVITY, pr.g_grav(0.0), B.WALK_NEWTON_EWALD)
            d = run_op(S, B.DD_DENSITY, pr.g_dens())
            th = []
            for fp in S.fp:
                fp.sync()
                t0 = time.perf_counter()
                fp.update_hmax()
                fp.sync()
                th.append(1e3 * (time.perf_counter() - t0))
            h = run_op(S, B.DD_HYDRO, pr.g_hydro())
            S.each(lambda fp: fp.gravity_finish(pr.G))
        def label(base, known, k):
            return [known[i] if i < len(known) else "%s:phase%d" % (base, i) for i in range(k)]
        names = (["mig%d" % i for i in range(len(mig))] +
                 label("grav", ["grav:localtree+groups", "grav:let-select", "grav:merged-tree+walks"], len(g)) +
                 label("dens", ["dens:groups", "dens:ghost-select", "dens:gastree+iter",
                                "dens:growth-check+refresh-pack", "dens:refresh-unpack"], len(d)) +
                 ["hmax"] + label("hydro", ["hydro"], len(h)))
        rows = mig + g + d + [th] + h
        tot = sum(max(r) for r in rows)
        info = S.each(lambda fp: fp.dd_info())
        print("ng=%d P=%d: step >= %.2f ms (sum over phases of the slowest shard)" % (ng, P, tot))
        for nm, r in zip(names, rows):
            print("   %-26s max %.3f  mean %.3f" % (nm, max(r), float(np.mean(r))))
        print("   imported tree elements %s, ghosts %s" % ([i["let_imported"] for i in info],
                                                          [i["ghosts_imported"] for i in info]))
        st = S.each(lambda fp: fp.stats())
        print("   walk kernels per shard (ms): newton %s  ewald %s; tree %s" % (
            ["%.2f" % s_["ms_grav"] for s_ in st], ["%.2f" % s_["ms_ewald"] for s_ in st],
            ["%.2f" % s_["ms_tree"] for s_ in st]))
        if base is None:
            base = tot
        print("   speed-up over P=%d: %.2f" % (Ps[0], base / tot), flush=True)
        S.close()


if __name__ == "__main__":
    main()
