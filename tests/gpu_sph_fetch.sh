#!/bin/bash
# Fabric traffic (FETCH_SIZE) of the SPH kernels alone, with the XCD-contiguous bucket order and without
# (DESIGN.md 4.4).  Run on the GPU box from the repo root:  bash tests/gpu_sph_fetch.sh  -> gpurun_out/sph_fetch.json
set -e
R=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$R/tests"
for v in 1 0; do
  GHIP_SPH_XCD=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/sphpmc_$v" -o p -- \
    python3 gpu_sphperf.py 64 3 > "$R/gpurun_out/sphpmc_$v.log" 2>&1
done
python3 - "$R" <<'PY'
import csv, glob, json, sys
R = sys.argv[1]
out = {"command": "GHIP_SPH_XCD={1,0} rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tests/gpu_sphperf.py 64 3",
       "unit": "GB per launch = 2 * FETCH_SIZE[KB] * 1024 / 1e9 (gfx950 correction, MI355X_MICROARCH.md)",
       "algorithmic_GB": {"hydro": 1.00, "density_pass": 0.58}}
for v in (1, 0):
    f = glob.glob("%s/gpurun_out/sphpmc_%d/**/p_counter_collection.csv" % (R, v), recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        k = "hydro" if "k_hydro<" in r["Kernel_Name"] else ("density_pass" if "k_density<" in r["Kernel_Name"] else None)
        if k and r["Counter_Name"] == "FETCH_SIZE":
            acc.setdefault(k, []).append(float(r["Counter_Value"]))
    line = [ln for ln in open("%s/gpurun_out/sphpmc_%d.log" % (R, v)) if ln.startswith("ng=")]
    out["xcd_order" if v else "plain_order"] = {
        "fetch_GB": {k: sum(a) / len(a) * 2 * 1024 / 1e9 for k, a in acc.items()},
        "launches": {k: len(a) for k, a in acc.items()}, "timing_under_counters": line[-1].strip() if line else None}
json.dump(out, open("%s/gpurun_out/sph_fetch.json" % R, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
