"""Variant builds of the HIP library the GPU suite needs (fault injection): built on demand, and again when a source is newer."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rl-offline-simulation_amd", "csrc")


def fault_lib():
    """variants/lib_fault.so: -DSHUF_FAULT_INJECT (role A of the resident shuffle never starts) -DSHC_TEST_SMALL_LISTS (16-entry message lists)."""
    lib = os.path.join(CSRC, "variants", "lib_fault.so")
    srcs = glob.glob(os.path.join(CSRC, "*.h*")) + [os.path.join(ROOT, "include", "offsim.h")]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["bash", os.path.join(CSRC, "build.sh"), "-DSHUF_FAULT_INJECT", "-DSHC_TEST_SMALL_LISTS"], env=dict(os.environ, OUT="variants/lib_fault.so"))
    return lib
