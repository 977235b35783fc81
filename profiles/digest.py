"""Digest of profiles/collect.sh's output into the committed summaries:
  profiles/<tag>_bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats summary of the bench command
  profiles/<tag>_walk_pmc.json           per-kernel PMC means for the two gravity walks and the SPH kernels
  profiles/grav_walk_traffic.json        HBM bytes per relative-criterion launch of k_grav_walk<NEWTON>
                                         (bench.py reads this for roofline.traffic)
Usage: python profiles/digest.py gpurun_out/prof_r01 r01
The first launch of each gravity walk in a bench run is the Barnes-Hut pass that seeds OldAcc
(accel.c:61-68); it is left out so the numbers describe the relative-criterion launches the bench times.
"""
import csv
import glob
import json
import os
import shutil
import sys

KERNELS = {
    "k_grav_walk<NEWTON>": "void k_grav_walk<0, true",
    "k_grav_walk<EWALD>": "void k_grav_walk<2, true",
    "k_density": ("k_density(", "void k_density<"),
    "k_hydro": ("k_hydro(", "void k_hydro<"),
}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"),
                os.path.join(here, "%s_bench_kernel_stats.csv" % tag))
    counters = {k: {} for k in KERNELS}
    for f in sorted(glob.glob(os.path.join(src, "pmc*", "p_counter_collection.csv"))):
        seen = {}
        acc = {}
        for row in csv.DictReader(open(f)):
            for label, prefix in KERNELS.items():
                if row["Kernel_Name"].startswith(prefix):   # (str.startswith takes a tuple too)
                    key = (label, row["Counter_Name"])
                    seen[key] = seen.get(key, 0) + 1
                    if label.startswith("k_grav_walk") and seen[key] == 1:
                        continue   # the Barnes-Hut launch
                    a = acc.setdefault(key, [0.0, 0, 0.0])
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
                    a[2] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        for (label, cname), a in acc.items():
            counters[label][cname] = {"mean": a[0] / a[1], "launches": a[1],
                                      "mean_duration_ns": a[2] / a[1]}
    derived = {}
    for label, c in counters.items():
        d = {}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            d["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024
        if "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c:
            d["valu_active_fraction_of_wave_cycles"] = (c["SQ_ACTIVE_INST_VALU"]["mean"] /
                                                        c["SQ_WAVE_CYCLES"]["mean"])
        if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
            d["wait_inst_fraction_of_wave_cycles"] = (c["SQ_WAIT_INST_ANY"]["mean"] /
                                                      c["SQ_WAVE_CYCLES"]["mean"])
        if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
            # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS
            # note): cycles of the launch = GRBM / 8.  256 CUs x 4 SIMDs; a 64-bit VALU
            # wave-instruction holds its SIMD for 4 cycles.
            cyc = c["GRBM_GUI_ACTIVE"]["mean"] / 8.0
            d["valu_insts_per_simd_cycle"] = c["SQ_INSTS_VALU"]["mean"] / (cyc * 1024)
            d["valu_issue_slot_fraction"] = 4.0 * c["SQ_INSTS_VALU"]["mean"] / (cyc * 1024)
        if "SQ_WAVE_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            # SQ_WAVE_CYCLES counts quad-cycles
            d["mean_resident_waves_per_simd"] = (4.0 * c["SQ_WAVE_CYCLES"]["mean"] /
                                                 (c["GRBM_GUI_ACTIVE"]["mean"] / 8.0 * 1024))
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            d["l2_hit_rate"] = c["TCC_HIT_sum"]["mean"] / (c["TCC_HIT_sum"]["mean"] +
                                                          c["TCC_MISS_sum"]["mean"])
        if "GRBM_GUI_ACTIVE" in c:
            d["mean_clock_GHz"] = (c["GRBM_GUI_ACTIVE"]["mean"] / 8.0 /
                                   c["GRBM_GUI_ACTIVE"]["mean_duration_ns"])
        derived[label] = d
    out = {"note": "rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 bench.py --steps 3 "
                   "--warmup 1 --no-cpu-baseline (profiles/collect.sh); relative-criterion launches "
                   "only; FETCH_SIZE/WRITE_SIZE in KB as reported (gfx950: FETCH_SIZE counts half the "
                   "bytes of wide reads, MI355X_MICROARCH.md HBM section)",
           "counters": counters, "derived": derived}
    json.dump(out, open(os.path.join(here, "%s_walk_pmc.json" % tag), "w"), indent=1)
    c = counters["k_grav_walk<NEWTON>"]
    traffic = {"ng": 64, "kernel": "k_grav_walk<NEWTON>",
               "FETCH_SIZE_KB": c["FETCH_SIZE"]["mean"], "WRITE_SIZE_KB": c["WRITE_SIZE"]["mean"],
               "hbm_bytes_per_launch": derived["k_grav_walk<NEWTON>"]["hbm_bytes_per_launch"],
               "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 correction of MI355X_MICROARCH.md",
               "source": "profiles/%s_walk_pmc.json" % tag}
    json.dump(traffic, open(os.path.join(here, "grav_walk_traffic.json"), "w"), indent=1)
    # VALU wave-instructions per element visit of this binary: SQ_INSTS_VALU per launch of the PMC
    # pass over the element visits per launch the same run reported (its bench line).  bench.py
    # multiplies it with its own live visit counter for the issue-slot fraction of the roofline.
    valu = None
    for f in sorted(glob.glob(os.path.join(src, "pmc*.log"))):
        csvf = os.path.join(f[:-4], "p_counter_collection.csv")
        if not os.path.exists(csvf) or "SQ_INSTS_VALU" not in open(csvf).read(400000):
            continue
        for line in open(f).read().splitlines():
            if line.startswith('{"metric"'):
                rec = json.loads(line)
                steps = rec["work_per_step_rank0"]["grav_wave_steps"]
                # the launches of the TIMED steps only (the visits the bench line reports are theirs):
                # launch 0 is the Barnes-Hut pass, 1 the first relative-criterion pass, then the
                # warm-up steps, then the timed ones; later launches (the kernel alone, the drop-in
                # legs) run on the state the run reached and would bias the mean
                first = 2 + int(rec["warmup"])
                vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(csvf))
                        if r["Kernel_Name"].startswith(KERNELS["k_grav_walk<NEWTON>"])
                        and r["Counter_Name"] == "SQ_INSTS_VALU"]
                timed = vals[first:first + int(rec["steps"])]
                per_launch = sum(timed) / len(timed)
                valu = {"ng": 64, "kernel": "k_grav_walk<NEWTON>",
                        "SQ_INSTS_VALU_per_launch": per_launch,
                        "launches_averaged": len(timed),
                        "element_visits_per_launch": steps,
                        "valu_insts_per_wave_step": per_launch / steps,
                        "hbm_bytes_per_launch": traffic["hbm_bytes_per_launch"],
                        "source": "profiles/%s_walk_pmc.json + the bench line of the same PMC pass" % tag,
                        "note": "SQ_INSTS_VALU counts wave-instructions; a full-rate 64-bit VALU "
                                "instruction holds its SIMD for 4 cycles"}
    if valu:
        json.dump(valu, open(os.path.join(here, "%s_walk_valu.json" % tag), "w"), indent=1)
        print(json.dumps(valu, indent=1))
    print(json.dumps(derived, indent=1))
    # Agreement of the bench line's roofline.kernel_ms (HIP events) with the kernel trace of the SAME
    # run: the stats summary averages over every launch of the command (Barnes-Hut pass, warm-up, the
    # kernel alone, the drop-in legs); here only the launches of the timed steps are taken.
    tr = os.path.join(src, "stats", "bench_kernel_trace.csv")
    bl = os.path.join(src, "bench_under_profiler.json")
    if os.path.exists(tr) and os.path.exists(bl):
        rec = json.loads(open(bl).read().strip().splitlines()[-1])
        first, k = 2 + int(rec["warmup"]), int(rec["steps"])
        agree = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps %d --warmup %d "
                            "--no-cpu-baseline" % (k, rec["warmup"]),
                 "bench_ms_per_step": rec["ms_per_step"]}
        for label, key in (("k_grav_walk<NEWTON>", "grav"), ("k_grav_walk<EWALD>", "ewald")):
            rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
                          for r in csv.DictReader(open(tr)) if r["Kernel_Name"].startswith(KERNELS[label]))
            timed = rows[first:first + k]
            ms = sum(e - b for b, e in timed) / len(timed) / 1e6
            agree[label] = {"trace_mean_ms_of_the_timed_launches": ms,
                            "bench_line_phase_ms": rec["phases_ms_rank0"][key],
                            "ratio": ms / rec["phases_ms_rank0"][key]}
        json.dump(agree, open(os.path.join(here, "%s_trace_vs_bench.json" % tag), "w"), indent=1)
        print(json.dumps(agree, indent=1))


main()
