"""tools/ helper: point HISPARSE_HIP_LIB at libhisparse_hip_prof.so (make prof) -- the build that carries the HISPARSE_ABLATE / HISPARSE_DEPTH
/ timeline instantiations.  The product library refuses to launch while one of those switches is set (hisparse_hip.h: hs_set_option)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "hisparse_amd", "lib", "libhisparse_hip_prof.so")


def use_profiling_library():
    if "HISPARSE_HIP_LIB" in os.environ:
        return
    if not os.path.exists(PROF):
        subprocess.check_call(["make", "-C", ROOT, "-j8", "prof"])
    os.environ["HISPARSE_HIP_LIB"] = PROF


def needs_profiling_library(env):
    return any(k in env for k in ("HISPARSE_ABLATE", "HISPARSE_DEPTH", "HISPARSE_TIMELINE_OUT"))
