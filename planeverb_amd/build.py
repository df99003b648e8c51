"""Build helpers: compile planeverb_amd/libplaneverb_amd.so (hipcc, gfx950) in-tree."""
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
# PLANEVERB_AMD_LIB: development aid for A/B runs of two builds on one GPU box (tools/gpu_tune.py)
LIB_PATH = os.environ.get("PLANEVERB_AMD_LIB") or os.path.join(PKG_DIR, "libplaneverb_amd.so")


KERNEL_SOURCES = ["pv_kernels.hip", "pv_resident.hip", "pv_rt60.hip", "pv_fused.hip", "pv_analysis_dev.h", "pv_probe.hip", "pv_stream.h", "pv_seg.h", "pv_device.h", "pv_libm.h", "pv_prims.h", "pv_analysis.h", "Makefile"]


def kernel_source_hash():
    """sha256 (first 16 hex digits) of the sources that determine the device code: the stamp of profiles/hbm_traffic.json
    (counter profiles are only quoted by bench.py for the kernels they were collected on)"""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


EXP_LIB_PATH = os.path.join(PKG_DIR, "libplaneverb_amd_exp.so")


def build(force=False, jobs=4, experimental=True):
    """Compile every HIP/C++ source of the package for gfx950.  Works without a GPU (cross-compile).
    experimental: also the EXPERIMENTAL build (libplaneverb_amd_exp.so, -DPV_EXPERIMENTAL): the product's sources plus the
    arms that were built, measured and switched off (row-streaming segments, patch kernel, stacked tiles) and the tuning
    tiles of rounds 1-3 -- what their equivalence tests load (tests/conftest.py pvlib_exp); nothing else uses it."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC, "-j%d" % jobs], stdout=subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("build finished but %s is missing" % LIB_PATH)
    if experimental and not os.environ.get("PLANEVERB_AMD_LIB"):
        subprocess.check_call(["make", "-C", CSRC, "-j%d" % jobs, "BUILD=build_exp", "OUT=../libplaneverb_amd_exp.so",
                               "EXTRA=-DPV_EXPERIMENTAL"], stdout=subprocess.DEVNULL)
    return LIB_PATH
