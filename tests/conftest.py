import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(ROOT, "tests", "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    """element-wise: bit-identical float32, modulo the sign of zero, NaN == NaN"""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    return (bits(a) == bits(b)) | ((a == 0) & (b == 0)) | (np.isnan(a) & np.isnan(b))


CELL_DTYPE = np.dtype([("pr", np.float32), ("vx", np.float32), ("vy", np.float32), ("b", np.int16), ("by", np.int16)])


def same_cells(got, want):
    """two arrays of reference Cells (PvTypes.h:106-121; any of: structured, uint8 [T, 16], bytes): pr / vx / vy
    bit-identical modulo the sign of zero (the reference's ghost cells hold -0: beta * negative), b / by equal"""
    a = np.frombuffer(np.ascontiguousarray(got).tobytes(), CELL_DTYPE)
    b = np.frombuffer(np.ascontiguousarray(want).tobytes(), CELL_DTYPE)
    return (a.shape == b.shape and all(same_bits(a[f], b[f]).all() for f in ("pr", "vx", "vy")) and
            np.array_equal(a["b"], b["b"]) and np.array_equal(a["by"], b["by"]))


def rel_err(a, b):
    a = np.atleast_1d(np.asarray(a, np.float64))
    b = np.atleast_1d(np.asarray(b, np.float64))
    with np.errstate(all="ignore"):
        e = np.abs(a - b) / np.maximum(np.abs(b), 1e-30)
    e[(a == b) | (np.isnan(a) & np.isnan(b))] = 0
    return e


def valid_mask(delay, T, fs):
    """cells whose analysis windows lie inside the IR (onset + N_dry + 2 <= T - N_cut); elsewhere the
    reference reads past its own vector and only `delay` is comparable."""
    n_dry = int(np.float32(0.01) * np.float32(fs))
    n_cut = int(np.float32(0.01) * np.float32(fs))
    return (delay < 1e30) & (delay + n_dry + 2 <= T - n_cut)


@pytest.fixture(scope="session")
def pvlib():
    # torch first where a GPU is present (as bench.py does): libplaneverb_amd.so then binds to the HIP runtime torch has
    # already loaded, and the tests that hand torch DEVICE tensors to the library (slab halos by address) see one runtime
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # noqa: BLE001 -- torch is plumbing for a few tests, not a requirement of the library
        pass
    import planeverb_amd
    planeverb_amd.build()
    from planeverb_amd import api
    return api


# tiles of the PRODUCT library (csrc/pv_kernels.hip PV_PRODUCT_STEP_CONFIGS); everything else -- other (K, rows), stacked tiles,
# row-streaming segments, the patch kernel, the unpacked air kernel -- exists in the experimental build only
PRODUCT_TILES = {(8, 24), (10, 36), (12, 36), (8, 40), (12, 12), (10, 20)}


def needs_experimental(opts):
    k, r = opts.get("steps_per_launch", 0), opts.get("tile_rows", 0)
    if (k or r) and (k or 8, r or 24) not in PRODUCT_TILES:
        return True
    if opts.get("packed_math", 1) == 0:  # (the unpacked air kernel: a validation form)
        return True
    return bool(opts.get("stream_rows") or opts.get("patch_kernel", 0) > 0)


@pytest.fixture(scope="session")
def pvlib_exp(pvlib):
    """planeverb_amd.api bound to the EXPERIMENTAL build of the library (libplaneverb_amd_exp.so)"""
    from planeverb_amd.build import EXP_LIB_PATH
    if not os.path.exists(EXP_LIB_PATH):
        pytest.skip("experimental build of the library not present")
    return pvlib.variant(EXP_LIB_PATH)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pvoracle
    pvoracle.build()
    return pvoracle
