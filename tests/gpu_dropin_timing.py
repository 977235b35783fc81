"""Development aid: wall time of the PCIe-inclusive drop-in path (gravity_tree / density /
force_update_hmax / hydro_force on AoS records, each with its H2D/D2H of the record blocks) at the
c2 size.  Pinning the blocks with hipHostRegister was tried and changed nothing (21.7 vs 21.5 ms)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common import Problem                      # noqa: E402
from test_gpu_parity import _host_problem       # noqa: E402


def run(ng):
    H = importlib.import_module("gadget-leicester_amd.hostapi")
    pr = Problem(ng=ng, gas=True, periodic=1)
    host, P, S = _host_problem(pr, H, 1)
    L = host.L
    L.gravity_tree()                             # Barnes-Hut pass for OldAcc
    best = None
    for _ in range(4):
        host.All.ErrTolTheta = 0
        t = [time.perf_counter()]
        for f in (L.gravity_tree, L.density, L.force_update_hmax, L.hydro_force):
            f()
            t.append(time.perf_counter())
        d = [1e3 * (b - a) for a, b in zip(t, t[1:])]
        if best is None or sum(d) < sum(best):
            best = d
    assert host.endrun_codes == []
    host.close()
    return best


if __name__ == "__main__":
    ng = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    d = run(ng)
    print("gravity_tree %.2f  density %.2f  hmax %.2f  hydro_force %.2f  total %.2f ms"
          % (tuple(d) + (sum(d),)), flush=True)
