"""oracle/pvoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding for oracle/libpvoracle.so (pv_oracle.c, the plain-C restatement of the reference hot path).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpvoracle.so")


class _Grid(C.Structure):
    _fields_ = [("res", C.c_int), ("sizeX", C.c_float), ("sizeY", C.c_float), ("dx", C.c_float),
                ("dt", C.c_float), ("fs", C.c_uint), ("T", C.c_int), ("gridSizeXf", C.c_float),
                ("gridSizeYf", C.c_float), ("gx", C.c_int), ("gy", C.c_int), ("ncell", C.c_int),
                ("b", C.POINTER(C.c_short)), ("R", C.POINTER(C.c_float)), ("pulse", C.POINTER(C.c_float)),
                ("hist_pr", C.POINTER(C.c_float)), ("hist_vx", C.POINTER(C.c_float)),
                ("hist_vy", C.POINTER(C.c_float))]


def build():
    src = os.path.join(_HERE, "pv_oracle.c")
    if (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libpvoracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        gp = C.POINTER(_Grid)
        L.pvo_grid_params.argtypes = [C.c_int, fp, fp, C.POINTER(C.c_uint)]
        L.pvo_response_length.restype = C.c_int
        L.pvo_response_length.argtypes = [C.c_uint]
        L.pvo_gaussian_pulse.argtypes = [C.c_int, C.c_uint, fp, C.c_int]
        L.pvo_grid_create.restype = gp
        L.pvo_grid_create.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int]
        L.pvo_grid_destroy.argtypes = [gp]
        L.pvo_add_aabb.argtypes = [gp, fp]
        L.pvo_remove_aabb.argtypes = [gp, fp]
        L.pvo_listener_cell.argtypes = [gp, C.c_float, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.pvo_fdtd.argtypes = [gp, C.c_float, C.c_float, fp]
        L.pvo_free_energy.restype = C.c_float
        L.pvo_free_energy.argtypes = [C.c_float, C.c_float, C.c_int]
        L.pvo_efree_per_r.restype = C.c_float
        L.pvo_efree_per_r.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        L.pvo_analyze.argtypes = [gp, C.c_float, C.c_float, C.c_float, fp, fp, C.POINTER(C.c_ubyte)]
        L.pvo_analyze_at.argtypes = [gp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, fp, fp,
                                     C.POINTER(C.c_ubyte)]
        L.pvo_result_index.restype = C.c_int
        L.pvo_result_index.argtypes = [gp, C.c_float, C.c_float]
        L.pvo_find_gains.argtypes = [C.c_float, C.c_float, fp, fp, fp]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def grid_params(res):
    dx, dt, fs = C.c_float(), C.c_float(), C.c_uint()
    lib().pvo_grid_params(res, dx, dt, fs)
    return dx.value, dt.value, fs.value


def gaussian_pulse(res, fs, n):
    out = np.empty(n, np.float32)
    lib().pvo_gaussian_pulse(res, fs, _fp(out), n)
    return out


def free_energy(size_x, size_y, res):
    return lib().pvo_free_energy(size_x, size_y, res)


def find_gains(rt60, wet):
    a, b, c = C.c_float(), C.c_float(), C.c_float()
    lib().pvo_find_gains(rt60, wet, a, b, c)
    return a.value, b.value, c.value


class OracleGrid:
    def __init__(self, size_x, size_y, res, aabbs=None, with_history=True):
        self._g = lib().pvo_grid_create(size_x, size_y, res, int(with_history))
        g = self._g.contents
        self.gx, self.gy, self.T, self.fs, self.dx, self.dt = g.gx, g.gy, g.T, g.fs, g.dx, g.dt
        self.ncell = g.ncell
        self.res = res
        self.size = (size_x, size_y)
        if aabbs is not None:
            for a in np.ascontiguousarray(aabbs, np.float32):
                self.add_aabb(a)

    def close(self):
        if self._g:
            lib().pvo_grid_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_aabb(self, a5):
        a = np.ascontiguousarray(a5, np.float32)
        lib().pvo_add_aabb(self._g, _fp(a))

    def remove_aabb(self, a5):
        a = np.ascontiguousarray(a5, np.float32)
        lib().pvo_remove_aabb(self._g, _fp(a))

    def pulse(self):
        return np.ctypeslib.as_array(self._g.contents.pulse, (self.T,)).copy()

    def material(self):
        shp = (self.gx + 1, self.gy + 1)
        b = np.ctypeslib.as_array(self._g.contents.b, (self.ncell,)).copy().reshape(shp)
        R = np.ctypeslib.as_array(self._g.contents.R, (self.ncell,)).copy().reshape(shp)
        return b, R

    def listener_cell(self, lx, lz):
        cx, cy = C.c_int(), C.c_int()
        lib().pvo_listener_cell(self._g, lx, lz, cx, cy)
        return cx.value, cy.value

    def fdtd(self, listener, want_fields=False):
        f = np.empty(3 * self.ncell, np.float32) if want_fields else None
        lib().pvo_fdtd(self._g, float(listener[0]), float(listener[2]), _fp(f) if want_fields else None)
        if want_fields:
            return f.reshape(3, self.gx + 1, self.gy + 1)

    def history(self):
        """views [T, gx+1, gy+1] of the recorded pr, vx, vy"""
        g = self._g.contents
        shp = (self.T, self.gx + 1, self.gy + 1)
        n = self.T * self.ncell
        return tuple(np.ctypeslib.as_array(p, (n,)).reshape(shp) for p in (g.hist_pr, g.hist_vx, g.hist_vy))

    def analyze(self, efree, listener, offset=(0, 0)):
        """offset != (0, 0): this grid is the window at that cell offset of a larger open grid and `listener` is in
        the larger grid's metres (pvo_analyze_at)"""
        n = self.gx * self.gy
        res = np.zeros((n, 8), np.float32)
        delay = np.empty(n, np.float32)
        valid = np.zeros(n, np.uint8)
        lib().pvo_analyze_at(self._g, efree, float(listener[0]), float(listener[2]), int(offset[0]), int(offset[1]),
                             _fp(res), _fp(delay), valid.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return (res.reshape(self.gx, self.gy, 8), delay.reshape(self.gx, self.gy),
                valid.reshape(self.gx, self.gy).astype(bool))

    def result_index(self, emitter):
        return lib().pvo_result_index(self._g, float(emitter[0]), float(emitter[2]))
