"""gadget-leicester_amd -- MI355X-native force path (tree gravity + SPH density/hydro) of the
GADGET-3 Leicester fork, behind a C-ABI (include/ghip.h, include/gadget_force.h).

The directory name contains a hyphen, so import it with
    importlib.import_module("gadget-leicester_amd")
This module only builds and loads the native libraries; there is NO CPU fallback: every compute
entry point fails loudly when the HIP library or a GPU is missing.
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
HOST = os.path.join(PKG_DIR, "host")
LIBGHIP = os.path.join(PKG_DIR, "libghip.so")
LIBHOST = os.path.join(PKG_DIR, "libgadget_force.so")

HIP_SOURCES = ["ghip_api.hip", "ghip_tree.hip", "ghip_gravity.hip", "ghip_sph.hip",
               "ghip_shard.hip", "ghip_drift.hip", "ghip_kick.hip", "ghip_export.hip", "ghip_pm.hip",
               "ghip_dd.hip", "ghip_comm.hip", "ghip_sink.hip"]
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the gfx950 extension cannot be built")


def build(force=False, verbose=False):
    """Compile libghip.so (HIP, gfx950) and libgadget_force.so (host C mirror) in-tree."""
    hipcc = hipcc_path()
    hdrs = [os.path.join(CSRC, "ghip_internal.h"), os.path.join(CSRC, "ghip_walk.h"), os.path.join(CSRC, "ghip_timefac.h"),
            os.path.join(CSRC, "ghip_keys.h"),
            os.path.join(REPO_DIR, "include", "ghip.h")]
    objs = []
    procs = []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            cmd = [hipcc] + HIPCC_FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or _newer(LIBGHIP, objs + [os.path.join(CSRC, "ghip.map")]):
        cmd = ([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950"] + objs +
               ["-Wl,--version-script=" + os.path.join(CSRC, "ghip.map"), "-lhipfft", "-ldl", "-o", LIBGHIP])
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    hsrc = os.path.join(HOST, "gadget_force.c")
    if os.path.exists(hsrc):
        hdeps = [hsrc, os.path.join(REPO_DIR, "include", "gadget_force.h"),
                 os.path.join(REPO_DIR, "include", "ghip.h"), LIBGHIP]
        if force or _newer(LIBHOST, hdeps):
            cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=gnu99", "-Wall",
                   "-I", os.path.join(REPO_DIR, "include"), hsrc, "-o", LIBHOST,
                   "-L", PKG_DIR, "-lghip", "-Wl,-rpath,$ORIGIN", "-lm"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    return LIBGHIP


def lib_path():
    # (development: an A/B build of the library, tests/build_variant.sh)
    alt = os.environ.get("GHIP_LIBGHIP")
    if alt:
        if not os.path.exists(alt):
            raise RuntimeError("GHIP_LIBGHIP=%s does not exist" % alt)
        return alt
    if not os.path.exists(LIBGHIP):
        raise RuntimeError(
            "libghip.so is missing (%s): run __graft_entry__.build() -- there is no CPU fallback"
            % LIBGHIP)
    return LIBGHIP
