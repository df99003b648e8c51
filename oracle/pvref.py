"""oracle/pvref.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding for oracle/_ref/libpvref.so: the unmodified reference (Grid / FreeGrid / Analyzer) driven
directly by oracle/ref_harness.cpp.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product package never does.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libpvref.so")


DSP_LIB_PATH = os.path.join(_HERE, "_ref", "libpvrefdsp.so")


def available():
    return os.path.exists(LIB_PATH)


def dsp_available():
    return os.path.exists(DSP_LIB_PATH)


_dsp = None


def find_gains(rt60, wet):
    """(A, B, C) from the reference's own compiled FindGainA/B/C (PlaneverbDSP/src/PvDSPContext.cpp:165-228,
    oracle/ref_dsp_harness.cpp)"""
    global _dsp
    if _dsp is None:
        L = C.CDLL(DSP_LIB_PATH)
        for f in "abc":
            fn = getattr(L, "pvrefdsp_find_gain_" + f)
            fn.restype = C.c_float
            fn.argtypes = [C.c_float, C.c_float]
        _dsp = L
    return tuple(getattr(_dsp, "pvrefdsp_find_gain_" + f)(float(rt60), float(wet)) for f in "abc")


def load_pv(path):
    """Parse a .pv scene (PlaneverbSandbox/src/Editor/Editor.cpp:245-281): count, then id x y w h R per box.
    The id is read and discarded, exactly as the reference loader does."""
    with open(path) as f:
        tok = f.read().split()
    n = int(tok[0])
    vals = np.array(tok[1:1 + 6 * n], dtype=np.float32).reshape(n, 6)
    return np.ascontiguousarray(vals[:, 1:6])


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.pvref_create.restype = C.c_void_p
        L.pvref_create.argtypes = [C.c_float, C.c_float, C.c_int, fp, C.c_int, C.c_int]
        L.pvref_destroy.argtypes = [C.c_void_p]
        L.pvref_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4 + [fp] * 3 + [C.POINTER(C.c_double)] * 2
        L.pvref_pulse.argtypes = [C.c_void_p, fp]
        L.pvref_material.argtypes = [C.c_void_p, C.POINTER(C.c_short), fp]
        L.pvref_add_aabb.argtypes = [C.c_void_p, fp]
        L.pvref_remove_aabb.argtypes = [C.c_void_p, fp]
        L.pvref_generate.restype = C.c_double
        L.pvref_generate.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.pvref_analyze.restype = C.c_double
        L.pvref_analyze.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.pvref_results.argtypes = [C.c_void_p, fp, fp]
        L.pvref_ir.argtypes = [C.c_void_p, C.c_int, C.c_int, fp]
        L.pvref_ir_cells.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.pvref_snapshot.argtypes = [C.c_void_p, C.c_int, fp, fp, fp]
        L.pvref_output.restype = C.c_int
        L.pvref_output.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, fp]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class RefSolver:
    """One reference Grid (+ FreeGrid + Analyzer) for a scene."""

    def __init__(self, size_x, size_y, res, aabbs=None, with_free_grid=True):
        L = lib()
        aabbs = np.zeros((0, 5), np.float32) if aabbs is None else np.ascontiguousarray(aabbs, np.float32)
        self._h = L.pvref_create(size_x, size_y, res, _fp(aabbs), len(aabbs), int(with_free_grid))
        gx, gy, T, fs = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        dx, dt, ef = C.c_float(), C.c_float(), C.c_float()
        cg, cf = C.c_double(), C.c_double()
        L.pvref_info(self._h, gx, gy, T, fs, dx, dt, ef, cg, cf)
        self.gx, self.gy, self.T, self.fs = gx.value, gy.value, T.value, fs.value
        self.dx, self.dt, self.efree = dx.value, dt.value, ef.value
        self.ctor_grid_s, self.ctor_free_s = cg.value, cf.value

    def close(self):
        if self._h:
            lib().pvref_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def pulse(self):
        out = np.empty(self.T, np.float32)
        lib().pvref_pulse(self._h, _fp(out))
        return out

    def material(self):
        n = (self.gx + 1) * (self.gy + 1)
        b = np.empty(n, np.int16)
        R = np.empty(n, np.float32)
        lib().pvref_material(self._h, b.ctypes.data_as(C.POINTER(C.c_short)), _fp(R))
        return b.reshape(self.gx + 1, self.gy + 1), R.reshape(self.gx + 1, self.gy + 1)

    def add_aabb(self, a5):
        a = np.ascontiguousarray(a5, np.float32)
        lib().pvref_add_aabb(self._h, _fp(a))

    def remove_aabb(self, a5):
        a = np.ascontiguousarray(a5, np.float32)
        lib().pvref_remove_aabb(self._h, _fp(a))

    def generate(self, listener):
        return lib().pvref_generate(self._h, *[float(v) for v in listener])

    def analyze(self, listener):
        return lib().pvref_analyze(self._h, *[float(v) for v in listener])

    def results(self):
        n = self.gx * self.gy
        res = np.empty((n, 8), np.float32)
        delay = np.empty(n, np.float32)
        lib().pvref_results(self._h, _fp(res), _fp(delay))
        return res.reshape(self.gx, self.gy, 8), delay.reshape(self.gx, self.gy)

    def ir(self, cx, cy):
        out = np.empty((self.T, 3), np.float32)
        lib().pvref_ir(self._h, int(cx), int(cy), _fp(out))
        return out

    def ir_cells(self, cx, cy):
        """raw reference Cells of one IR: uint8 [T, 16] = {f32 pr, vx, vy; i16 b, by} (PvTypes.h:106-121)"""
        out = np.empty((self.T, 16), np.uint8)
        lib().pvref_ir_cells(self._h, int(cx), int(cy), out.ctypes.data_as(C.c_void_p))
        return out

    def snapshot(self, t):
        n = (self.gx + 1) * (self.gy + 1)
        pr, vx, vy = (np.empty(n, np.float32) for _ in range(3))
        lib().pvref_snapshot(self._h, int(t), _fp(pr), _fp(vx), _fp(vy))
        shp = (self.gx + 1, self.gy + 1)
        return pr.reshape(shp), vx.reshape(shp), vy.reshape(shp)

    def output(self, emitter):
        out = np.empty(8, np.float32)
        ok = lib().pvref_output(self._h, *[float(v) for v in emitter], _fp(out))
        return out if ok else None
