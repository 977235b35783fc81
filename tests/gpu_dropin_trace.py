"""Timeline of the PCIe-inclusive drop-in sequence (bench.py's dropin_timing) for rocprofv3:
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/dropin -- python3 tests/gpu_dropin_trace.py [pin]
Prints the host-side times of the four calls."""
import importlib
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from common import Problem  # noqa: E402
from test_gpu_parity import _host_problem  # noqa: E402

H = importlib.import_module("gadget-leicester_amd.hostapi")
pin = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pr = Problem(ng=64, gas=True, periodic=1)
host, P, S = _host_problem(pr, H, 1, overlap_sph=1, pin_records=pin)
L = host.L
L.gravity_tree()
L.gadget_force_flush()
for rep in range(4):
    host.All.ErrTolTheta = 0
    t = [time.perf_counter()]
    for f in (L.gravity_tree, L.density, L.force_update_hmax, L.hydro_force):
        f()
        t.append(time.perf_counter())
    print("rep %d: " % rep + " ".join("%.3f" % (1e3 * (b - a)) for a, b in zip(t, t[1:])),
          "total %.3f ms" % (1e3 * (t[-1] - t[0])), flush=True)
print("last error text:", L.gadget_force_last_error().decode())
host.close()
