"""ctypes binding of libplaneverb_amd.so -- the host-side mirror of Planeverb's interface for the FDTD path.

Two layers, both thin:

* module-level functions `Init / Exit / Emit / UpdateEmission / EndEmission / GetOutput / AddGeometry /
  UpdateGeometry / RemoveGeometry / SetListenerPosition` with the names, argument meaning and sentinel behaviour of
  the reference's C++ API (ProjectPlaneverb/include/Planeverb.h:12-47) on top of the flat C-ABI
  (PlaneverbUnityPluginAPI/PlaneverbUnity.cpp:25-135);
* `Solver`, the synchronous batch handle (PvAmd* extension) used by the benchmarks, the parity tests and the
  multi-GPU sharding layer.

There is no CPU implementation behind this module: if the shared library is missing it raises, and if no HIP device
is visible every call that needs one fails loudly.
"""
import ctypes as C
import os
from collections import namedtuple

import numpy as np

from .build import LIB_PATH

PV_INVALID_DRY_GAIN = -1.0
PV_INVALID_ID = -1

PVA_OPT_DENSE_HISTORY = 1
PVA_OPT_NUM_STEPS = 2
PVA_OPT_SKIP_ANALYSIS = 3
PVA_OPT_USE_GRAPH = 4
PVA_OPT_STEPS_PER_LAUNCH = 5
PVA_OPT_TILE_ROWS = 6
PVA_OPT_NO_FREE_GRID = 7
PVA_OPT_TIME_KERNELS = 8
PVA_OPT_TILE_ORDER = 9
PVA_OPT_SMALL_GRID_KERNEL = 10
PVA_OPT_PACKED_MATH = 11
PVA_OPT_STREAMING_ANALYSIS = 12
PVA_OPT_STREAM_ROWS = 13
PVA_OPT_MERGED_LAUNCH = 14
PVA_OPT_EDGE_TILES = 15
PVA_OPT_ROW_BANDS = 16
PVA_OPT_PATCH_KERNEL = 17
PVA_OPT_PATCH_STRIP = 18
PVA_OPT_LAZY_FAR_CELLS = 19
PVA_OPT_STREAM_FUSE = 20
PVA_OPT_AUX_STREAMS = 21
PVA_OPT_RESIDENT_KERNEL = 22
PVA_OPT_RT60_LANES = 23
PVA_OPT_DEBUG_LOSE_FIRST_CAPTURE = 24
PVA_OPT_STREAM_PRIORITY = 25
PVA_OPT_ALTERNATE_SWEEPS = 26
PVA_OPT_XCD_REGIONS = 27
PVA_OPT_ANALYSIS_FORK = 28
PVA_OPT_FUSED_ANALYSIS = 29


class PlaneverbOutput(C.Structure):
    """PlaneverbUnity.cpp:66-76"""
    _fields_ = [("occlusion", C.c_float), ("wetGain", C.c_float), ("rt60", C.c_float), ("lowpass", C.c_float),
                ("directionX", C.c_float), ("directionY", C.c_float), ("sourceDirectionX", C.c_float),
                ("sourceDirectionY", C.c_float)]

    def as_array(self):
        return np.array([self.occlusion, self.wetGain, self.rt60, self.lowpass, self.directionX, self.directionY,
                         self.sourceDirectionX, self.sourceDirectionY], np.float32)


class PlaneverbCell(C.Structure):
    """PvTypes.h:106-121"""
    _fields_ = [("pr", C.c_float), ("vx", C.c_float), ("vy", C.c_float), ("b", C.c_short), ("by", C.c_short)]


CELL_DTYPE = np.dtype([("pr", np.float32), ("vx", np.float32), ("vy", np.float32), ("b", np.int16), ("by", np.int16)])


class PvAmdInfo(C.Structure):
    _fields_ = [("gx", C.c_int), ("gy", C.c_int), ("T", C.c_int), ("fs", C.c_int), ("res", C.c_int),
                ("dx", C.c_float), ("dt", C.c_float), ("efree", C.c_float), ("device", C.c_int),
                ("stepsPerLaunch", C.c_int), ("tileRows", C.c_int), ("tileCols", C.c_int), ("pitch", C.c_int),
                ("rows", C.c_int), ("histRows", C.c_int), ("histPitch", C.c_int), ("numGeometry", C.c_int),
                ("deviceBytes", C.c_longlong), ("streamFuse", C.c_int), ("residentKernel", C.c_int)]


class PvAmdSlabInfo(C.Structure):
    _fields_ = [("nslabs", C.c_int), ("row0", C.c_int * 16), ("rows", C.c_int * 16), ("device", C.c_int * 16),
                ("haloBytesPerLaunch", C.c_longlong), ("exchangeBytesPerRun", C.c_longlong),
                ("deviceBytes", C.c_longlong * 16), ("handoffWords", C.c_int), ("streamRedeals", C.c_int),
                ("dryRunUsPerSweep", C.c_float)]


class PvAmdTimings(C.Structure):
    _fields_ = [("fdtdMs", C.c_float), ("analysisMs", C.c_float), ("geometryMs", C.c_float),
                ("stepKernelMs", C.c_float), ("stepLaunches", C.c_int), ("airKernelMs", C.c_float), ("generalKernelMs", C.c_float), ("airLaunches", C.c_int),
                ("generalLaunches", C.c_int), ("stepLoopMs", C.c_float), ("reachedCells", C.c_int), ("activeCells", C.c_int), ("silentCells", C.c_int)]


# every symbol include/planeverb_amd.h declares: name -> (restype, argtypes)
_fp = C.POINTER(C.c_float)
_vp = C.c_void_p
SYMBOLS = {
    "UnityPluginLoad": (None, [_vp]),
    "UnityPluginUnload": (None, []),
    "PlaneverbInit": (None, [C.c_float, C.c_float, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int]),
    "PlaneverbExit": (None, []),
    "PlaneverbEmit": (C.c_int, [C.c_float] * 3),
    "PlaneverbUpdateEmission": (None, [C.c_int] + [C.c_float] * 3),
    "PlaneverbEndEmission": (None, [C.c_int]),
    "PlaneverbGetOutput": (PlaneverbOutput, [C.c_int]),
    "PlaneverbAddGeometry": (C.c_int, [C.c_float] * 5),
    "PlaneverbUpdateGeometry": (None, [C.c_int] + [C.c_float] * 5),
    "PlaneverbRemoveGeometry": (None, [C.c_int]),
    "PlaneverbSetListenerPosition": (None, [C.c_float] * 3),
    "PlaneverbLoadScene": (C.c_int, [C.c_char_p]),
    "PlaneverbIterationCount": (C.c_longlong, []),
    "PlaneverbWaitIterations": (C.c_longlong, [C.c_longlong, C.c_int]),
    "PlaneverbIsRunning": (C.c_int, []),
    "PlaneverbWorkerError": (C.c_char_p, []),
    "PlaneverbIsStreaming": (C.c_int, []),
    "PlaneverbGetImpulseResponse": (C.c_int, [C.c_float] * 3 + [C.POINTER(PlaneverbCell), C.c_int]),
    "PvAmdDeviceCount": (C.c_int, []),
    "PvAmdLastError": (C.c_char_p, []),
    "PvAmdVersion": (C.c_char_p, []),
    "PvAmdCreate": (_vp, [C.c_float, C.c_float, C.c_int, C.c_int]),
    "PlaneverbCreateGrid": (_vp, [C.c_float, C.c_float, C.c_int, C.c_int]),
    "PvAmdCreateSlabs": (_vp, [C.c_float, C.c_float, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "PvAmdGetSlabInfo": (C.c_int, [_vp, C.POINTER(PvAmdSlabInfo)]),
    "PvAmdCreateSlabRank": (_vp, [C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]),
    "PvAmdComputeEfree": (C.c_int, [C.c_float, C.c_float, C.c_int, C.c_int, _fp]),
    "PvAmdSlabSetEfree": (C.c_int, [_vp, C.c_float]),
    "PvAmdSlabBegin": (C.c_int, [_vp] + [C.c_float] * 3),
    "PvAmdSlabNumLaunches": (C.c_int, [_vp]),
    "PvAmdSlabLaunch": (C.c_int, [_vp, C.c_int]),
    "PvAmdSlabHaloFloats": (C.c_int, [_vp]),
    "PvAmdSlabExportHalo": (C.c_int, [_vp, C.c_int, _fp]),
    "PvAmdSlabImportHalo": (C.c_int, [_vp, C.c_int, _fp]),
    "PvAmdSlabHistoryFloats": (C.c_int, [_vp]),
    "PvAmdSlabExportEdgeHistory": (C.c_int, [_vp, _fp]),
    "PvAmdSlabImportAboveHistory": (C.c_int, [_vp, _fp]),
    "PvAmdSlabAnalyze": (C.c_int, [_vp]),
    "PvAmdSlabWindowBlock": (C.c_longlong, [_vp, C.POINTER(C.c_int), _fp, C.c_longlong]),
    "PvAmdSlabRootCreate": (_vp, [_vp, C.c_int]),
    "PvAmdSlabRootDestroy": (None, [_vp]),
    "PvAmdSlabRootBegin": (C.c_int, [_vp] + [C.c_float] * 3),
    "PvAmdSlabRootImportBlock": (C.c_int, [_vp, C.POINTER(C.c_int), _fp]),
    "PvAmdSlabRootFinish": (C.c_int, [_vp]),
    "PvAmdSlabRootGetOutput": (C.c_int, [_vp] + [C.c_float] * 3 + [C.POINTER(PlaneverbOutput)]),
    "PvAmdSlabRootCopyResults": (C.c_int, [_vp, _fp, _fp]),
    "PvAmdDestroy": (None, [_vp]),
    "PvAmdSetOption": (C.c_int, [_vp, C.c_int, C.c_longlong]),
    "PvAmdGetInfo": (C.c_int, [_vp, C.POINTER(PvAmdInfo)]),
    "PvAmdAddGeometry": (C.c_int, [_vp] + [C.c_float] * 5),
    "PvAmdUpdateGeometry": (C.c_int, [_vp, C.c_int] + [C.c_float] * 5),
    "PvAmdRemoveGeometry": (C.c_int, [_vp, C.c_int]),
    "PvAmdLoadScene": (C.c_int, [_vp, C.c_char_p]),
    "PvAmdSaveScene": (C.c_int, [_vp, C.c_char_p]),
    "PvAmdRun": (C.c_int, [_vp] + [C.c_float] * 3),
    "PvAmdRunAsync": (C.c_int, [_vp] + [C.c_float] * 3),
    "PvAmdRunAsyncAfter": (C.c_int, [_vp, _vp] + [C.c_float] * 3),
    "PvAmdSync": (C.c_int, [_vp]),
    "PvAmdRunBatch": (C.c_int, [C.POINTER(_vp), C.c_int, _fp, C.c_int]),
    "PvAmdGetTimings": (C.c_int, [_vp, C.POINTER(PvAmdTimings)]),
    "PvAmdClockProbe": (C.c_float, [C.c_int, _fp]),
    "PvAmdBandwidthProbe": (C.c_int, [C.c_int, _fp]),
    "PvAmdSetEmitters": (C.c_int, [_vp, _fp, C.c_int]),
    "PvAmdGetOutput": (C.c_int, [_vp] + [C.c_float] * 3 + [C.POINTER(PlaneverbOutput)]),
    "PvAmdSetOutputQueries": (C.c_int, [_vp, _fp, C.c_int]),
    "PvAmdGetQueriedOutputs": (C.c_int, [_vp, C.POINTER(PlaneverbOutput), C.c_int]),
    "PvAmdCopyResults": (C.c_int, [_vp, _fp, _fp]),
    "PvAmdCopyResultsBlock": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "PvAmdGetImpulseResponse": (C.c_int, [_vp, C.c_int, C.c_int, _fp]),
    "PvAmdGetImpulseResponseCells": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(PlaneverbCell)]),
    "PvAmdCopyFields": (C.c_int, [_vp, _fp, _fp, _fp]),
    "PvAmdCopyHistoryPlane": (C.c_int, [_vp, C.c_int, _fp]),
    "PvAmdCopyPulse": (C.c_int, [_vp, _fp]),
    "PvAmdCopyMaterial": (C.c_int, [_vp, C.POINTER(C.c_ubyte), _fp]),
    "PvAmdSetFields": (C.c_int, [_vp, _fp, _fp, _fp]),
    "PvAmdRunSteps": (C.c_int, [_vp, C.c_int, C.c_int, C.c_float, C.c_float]),
    "PvAmdShardPlan": (C.c_int, [C.c_int] * 4 + [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]),
    "PvAmdPlanSegments": (C.c_int, [C.POINTER(C.c_ubyte)] + [C.c_int] * 5 + [C.POINTER(C.c_int), C.c_int]),
    "PvAmdCommUniqueId": (C.c_int, [C.c_char_p]),
    "PvAmdCommCreate": (_vp, [C.c_char_p, C.c_int, C.c_int, C.c_int]),
    "PvAmdCommDestroy": (None, [_vp]),
    "PvAmdCommAllGather": (C.c_int, [_vp, _fp, C.c_int, _fp]),
    "PvAmdRunSharded": (C.c_int, [C.POINTER(_vp), C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, _vp,
                                  C.POINTER(PlaneverbOutput)]),
    "PvAmdReverbBusGains": (None, [C.c_float, C.c_float, _fp, _fp, _fp]),
    "PvAmdHostGridInfo": (C.c_int, [C.c_float, C.c_float, C.c_int, C.POINTER(PvAmdInfo)]),
    "PvAmdHostPulse": (C.c_int, [C.c_float, C.c_float, C.c_int, _fp]),
    "PvAmdHostPulseSelfCheck": (C.c_int, []),
    "PvAmdHostRasterize": (C.c_int, [C.c_float, C.c_float, C.c_int, _fp, C.POINTER(C.c_int), C.c_int,
                                     C.POINTER(C.c_ubyte), _fp]),
    "PvAmdHostLoadPv": (C.c_int, [C.c_char_p, _fp, C.c_int]),
    "PvAmdHostSavePv": (C.c_int, [C.c_char_p, _fp, C.POINTER(C.c_int), C.c_int]),
    "PvAmdHostCells": (C.c_int, [C.c_float, C.c_float, C.c_int, C.c_float, C.c_float] + [C.POINTER(C.c_int)] * 5),
}

_lib = None


def lib():
    """Load libplaneverb_amd.so (built by planeverb_amd.build.build()).  Raises if it is missing: there is no
    fallback implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def variant(lib_path):
    """This module once more, bound to ANOTHER build of the library -- e.g. libplaneverb_amd_exp.so, the experimental build
    (make EXTRA=-DPV_EXPERIMENTAL) that carries the kernel arms and tile configurations the product library leaves out.
    Both libraries can be used side by side in one process (each has its own state)."""
    import importlib.util
    import sys
    name = "%s_variant_%s" % (__name__, os.path.splitext(os.path.basename(lib_path))[0])
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, __file__)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = __package__
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    mod.LIB_PATH = lib_path
    return mod


def last_error():
    e = lib().PvAmdLastError()
    return e.decode() if e else ""


class PlaneverbError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise PlaneverbError(last_error())


def _f(a):
    return a.ctypes.data_as(_fp)


# --------------------------------------------------------------------------------------------------------------
# reference-shaped live API (Planeverb.h:12-47)
# --------------------------------------------------------------------------------------------------------------

Config = namedtuple("PlaneverbConfig", "gridSizeInMeters gridResolution gridBoundaryType tempFileDirectory "
                                       "maxThreadUsage threadExecutionType")
pv_CPU, pv_GPU = 0, 1  # PvTypes.h:13-17 / PlaneverbConfig.cs:23-29


def Init(config):
    """Planeverb::Init (PvContext.cpp:25-32).  Raises PlaneverbError where the reference throws."""
    d = config.tempFileDirectory
    lib().PlaneverbInit(config.gridSizeInMeters[0], config.gridSizeInMeters[1], config.gridResolution,
                        config.gridBoundaryType, d.encode() if d is not None else None, config.maxThreadUsage,
                        config.threadExecutionType)
    if not lib().PlaneverbIsRunning():
        raise PlaneverbError(last_error() or "pv_InvalidConfig")


def Exit():
    lib().PlaneverbExit()


def Emit(pos):
    return lib().PlaneverbEmit(*[float(v) for v in pos])


def UpdateEmission(eid, pos):
    lib().PlaneverbUpdateEmission(int(eid), *[float(v) for v in pos])


def EndEmission(eid):
    lib().PlaneverbEndEmission(int(eid))


def GetOutput(eid):
    return lib().PlaneverbGetOutput(int(eid))


def AddGeometry(aabb):
    """aabb = (posX, posY, width, height, absorption), PvMathTypes.h:31-49"""
    return lib().PlaneverbAddGeometry(*[float(v) for v in aabb])


def UpdateGeometry(gid, aabb):
    lib().PlaneverbUpdateGeometry(int(gid), *[float(v) for v in aabb])


def RemoveGeometry(gid):
    lib().PlaneverbRemoveGeometry(int(gid))


def SetListenerPosition(pos):
    lib().PlaneverbSetListenerPosition(*[float(v) for v in pos])


def GetImpulseResponse(pos):
    """Planeverb::GetImpulseResponse (Planeverb.h:47): structured array [T] of (pr, vx, vy, b, by) at a world position,
    from the last completed iteration; empty for a position outside the cell array"""
    n = lib().PlaneverbGetImpulseResponse(float(pos[0]), float(pos[1]), float(pos[2]), None, 0)
    if n < 0:
        raise PlaneverbError(last_error())
    out = np.zeros(n, CELL_DTYPE)
    if n:
        got = lib().PlaneverbGetImpulseResponse(float(pos[0]), float(pos[1]), float(pos[2]),
                                                out.ctypes.data_as(C.POINTER(PlaneverbCell)), n)
        if got != n:
            raise PlaneverbError(last_error())
    return out


def IsRunning():
    return bool(lib().PlaneverbIsRunning())


def LoadScene(path):
    n = lib().PlaneverbLoadScene(path.encode())
    if n < 0:
        raise PlaneverbError(last_error())
    return n


def WaitIterations(count, timeout_ms=60000):
    return lib().PlaneverbWaitIterations(int(count), int(timeout_ms))


def IterationCount():
    return lib().PlaneverbIterationCount()


def reverb_bus_gains(rt60, wet):
    a, b, c = C.c_float(), C.c_float(), C.c_float()
    lib().PvAmdReverbBusGains(rt60, wet, a, b, c)
    return a.value, b.value, c.value


def host_grid_info(size_x, size_y, res):
    i = PvAmdInfo()
    _check(lib().PvAmdHostGridInfo(float(size_x), float(size_y), int(res), i))
    return i


def host_pulse(size_x, size_y, res):
    i = host_grid_info(size_x, size_y, res)
    out = np.empty(i.T, np.float32)
    _check(lib().PvAmdHostPulse(float(size_x), float(size_y), int(res), _f(out)))
    return out


def host_rasterize(size_x, size_y, res, boxes, ops=None):
    i = host_grid_info(size_x, size_y, res)
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 5)
    ops_a = np.ascontiguousarray(ops if ops is not None else np.ones(len(boxes)), np.int32)
    beta = np.empty((i.gx + 1, i.gy + 1), np.uint8)
    R = np.empty((i.gx + 1, i.gy + 1), np.float32)
    _check(lib().PvAmdHostRasterize(float(size_x), float(size_y), int(res), _f(boxes),
                                    ops_a.ctypes.data_as(C.POINTER(C.c_int)), len(boxes),
                                    beta.ctypes.data_as(C.POINTER(C.c_ubyte)), _f(R)))
    return beta, R


def load_pv(path, max_boxes=4096):
    """.pv scene -> (n, 5) float32 array of (posX, posY, width, height, absorption)"""
    buf = np.empty((max_boxes, 5), np.float32)
    n = lib().PvAmdHostLoadPv(path.encode(), _f(buf), max_boxes)
    if n < 0:
        raise PlaneverbError(last_error())
    return buf[:n].copy()


def save_pv(path, boxes, ids=None):
    """write boxes [(posX, posY, width, height, absorption), ...] as a .pv scene (Editor.cpp:219-243)"""
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 5)
    ida = None if ids is None else np.ascontiguousarray(ids, np.int32)
    _check(lib().PvAmdHostSavePv(path.encode(), _f(b), None if ida is None else ida.ctypes.data_as(C.POINTER(C.c_int)),
                                 len(b)))


def host_cells(size_x, size_y, res, x, z):
    v = [C.c_int() for _ in range(5)]
    _check(lib().PvAmdHostCells(float(size_x), float(size_y), int(res), float(x), float(z), *v))
    return (v[0].value, v[1].value), ((v[2].value, v[3].value) if v[4].value else None)


def clock_probe(device=0):
    """(MHz by a timed s_sleep, MHz by s_memtime) of the device's shader clock at this moment (PvAmdClockProbe)"""
    m = C.c_float(0.0)
    v = lib().PvAmdClockProbe(int(device), C.byref(m))
    return float(v), float(m.value)


def bandwidth_probe(device=0):
    """the device's own streaming bandwidth in GB/s (PvAmdBandwidthProbe): copy with 16 B and with 4 B per lane (bytes read +
    written per second), read only, write only"""
    v = (C.c_float * 4)()
    _check(lib().PvAmdBandwidthProbe(int(device), v))
    return {"copy_x4": float(v[0]), "copy_dword": float(v[1]), "read_dword": float(v[2]), "write_dword": float(v[3])}


def device_count():
    return lib().PvAmdDeviceCount()


# --------------------------------------------------------------------------------------------------------------
# batch solver
# --------------------------------------------------------------------------------------------------------------

def batch_solver_options(n):
    """Solver options for grids of about n x n cells that are run in batches (run_batch): the mirror-pair tiles whose
    batched kernel has the edge-tile arm (grid-border tiles on the air path, tile class 2), measured on MI355X:
    +13 % at 512^2, +25 % at 1024^2 over the default tile of the size"""
    if n <= 768:
        return dict(steps_per_launch=8, tile_rows=40, edge_tiles=1, aux_streams=1)
    if n <= 1536:
        return dict(steps_per_launch=10, tile_rows=36, edge_tiles=1, aux_streams=1)
    return dict(steps_per_launch=12, tile_rows=36, edge_tiles=1, aux_streams=1)


def shard_plan(n_runs, world, rank, n_local_solvers):
    """PvAmdShardPlan: [(run index, local solver index), ...] of this rank (run k -> rank k mod world)"""
    cap = max(1, (n_runs + max(world, 1) - 1) // max(world, 1))
    r, s = (C.c_int * cap)(), (C.c_int * cap)()
    n = lib().PvAmdShardPlan(int(n_runs), int(world), int(rank), int(n_local_solvers), r, s, cap)
    return [(r[i], s[i]) for i in range(min(n, cap))]


class Comm:
    """RCCL communicator of the sharded runs (PvAmdComm*): one per process, bound to the process's HIP device"""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().PvAmdCommUniqueId(buf))
        return buf.raw

    def __init__(self, unique_id, rank, world, device):
        assert len(unique_id) == 128
        self._h = lib().PvAmdCommCreate(unique_id, int(rank), int(world), int(device))
        if not self._h:
            raise PlaneverbError(last_error())
        self.rank, self.world = rank, world

    def all_gather(self, mine):
        mine = np.ascontiguousarray(mine, np.float32).ravel()
        out = np.empty(self.world * mine.size, np.float32)
        _check(lib().PvAmdCommAllGather(self._h, _f(mine), mine.size, _f(out)))
        return out.reshape(self.world, -1)

    def close(self):
        if getattr(self, "_h", None):
            lib().PvAmdCommDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_segments(air, tile_rows, max_tile_columns, target):
    """PvAmdPlanSegments: the row-streaming segments (PVA_OPT_STREAM_ROWS) covering the air tiles of a [ntx, nty] 0/1
    array: int array [n, 4] of (first array row, rows, first tile column, tile columns)"""
    a = np.ascontiguousarray(air, np.uint8)
    ntx, nty = a.shape
    cap = 2 * a.size + 8
    out = np.zeros((cap, 4), np.int32)
    n = lib().PvAmdPlanSegments(a.ctypes.data_as(C.POINTER(C.c_ubyte)), ntx, nty, int(tile_rows), int(max_tile_columns),
                                int(target), out.ctypes.data_as(C.POINTER(C.c_int)), cap)
    return out[:min(n, cap)].copy()


def run_sharded(solvers, listeners, emitters, rank=0, world=1, comm=None):
    """PvAmdRunSharded: listeners [n, 3], emitters [n, E, 3] -> float32 [n, E, 8] on every rank"""
    L = np.ascontiguousarray(listeners, np.float32).reshape(-1, 3)
    Em = np.ascontiguousarray(emitters, np.float32).reshape(len(L), -1, 3)
    n, E = len(L), Em.shape[1]
    hs = (_vp * len(solvers))(*[sv._h for sv in solvers])
    out = (PlaneverbOutput * max(1, n * E))()
    _check(lib().PvAmdRunSharded(hs, len(solvers), _f(L), n, _f(Em), E, int(rank), int(world),
                                 comm._h if comm is not None else None, out))
    return np.frombuffer(out, np.float32).reshape(-1, 8)[:n * E].reshape(n, E, 8).copy()


def compute_efree(size_x, size_y, res, device=0):
    """FreeGrid energy of a config (FreeGrid.cpp:71-110): computed once, handed to every slab rank"""
    e = C.c_float()
    _check(lib().PvAmdComputeEfree(float(size_x), float(size_y), int(res), int(device), e))
    return e.value


class SlabRank:
    """ONE slab of a decomposed grid, owned by this process (PvAmdCreateSlabRank + PvAmdSlab*): the per-rank primitives of
    planeverb_amd.dist_slabs.  Buffers that cross ranks are numpy arrays."""

    def __init__(self, size_x, size_y, res, device, index, count, efree, **options):
        self.solver = Solver.__new__(Solver)
        self.solver._h = lib().PvAmdCreateSlabRank(float(size_x), float(size_y), int(res), int(device), int(index), int(count))
        if not self.solver._h:
            raise PlaneverbError(last_error())
        self._h = self.solver._h
        keys = {"steps_per_launch": PVA_OPT_STEPS_PER_LAUNCH, "tile_rows": PVA_OPT_TILE_ROWS, "num_steps": PVA_OPT_NUM_STEPS}
        for k, v in options.items():
            _check(lib().PvAmdSetOption(self._h, keys[k], int(v)))
        _check(lib().PvAmdSlabSetEfree(self._h, float(efree)))
        self.index, self.count = index, count
        self.num_launches = lib().PvAmdSlabNumLaunches(self._h)
        self.halo_floats = lib().PvAmdSlabHaloFloats(self._h)
        self.history_floats = lib().PvAmdSlabHistoryFloats(self._h)

    def close(self):
        self.solver.close()
        self._h = None

    def add_geometry(self, aabb):
        return lib().PvAmdAddGeometry(self._h, *[float(v) for v in aabb])

    def begin(self, listener):
        _check(lib().PvAmdSlabBegin(self._h, *[float(v) for v in listener]))

    def launch(self, li):
        _check(lib().PvAmdSlabLaunch(self._h, int(li)))

    def export_halo(self, side):
        out = np.empty(self.halo_floats, np.float32)
        _check(lib().PvAmdSlabExportHalo(self._h, int(side), _f(out)))
        return out

    def import_halo(self, side, buf):
        buf = np.ascontiguousarray(buf, np.float32)
        assert buf.size == self.halo_floats
        _check(lib().PvAmdSlabImportHalo(self._h, int(side), _f(buf)))

    # the same four transfers with raw addresses of buffers that may live on the slab's device (torch tensors: .data_ptr()):
    # what an RCCL transport uses, nothing is staged through the host (dist_slabs.TorchTransport)
    def export_halo_to(self, side, ptr):
        _check(lib().PvAmdSlabExportHalo(self._h, int(side), C.cast(C.c_void_p(int(ptr)), _fp)))

    def import_halo_from(self, side, ptr):
        _check(lib().PvAmdSlabImportHalo(self._h, int(side), C.cast(C.c_void_p(int(ptr)), _fp)))

    def export_edge_history_to(self, ptr):
        _check(lib().PvAmdSlabExportEdgeHistory(self._h, C.cast(C.c_void_p(int(ptr)), _fp)))

    def import_above_history_from(self, ptr):
        _check(lib().PvAmdSlabImportAboveHistory(self._h, C.cast(C.c_void_p(int(ptr)), _fp)))

    def export_edge_history(self):
        out = np.empty(self.history_floats, np.float32)
        _check(lib().PvAmdSlabExportEdgeHistory(self._h, _f(out)))
        return out

    def import_above_history(self, buf):
        buf = np.ascontiguousarray(buf, np.float32)
        assert buf.size == self.history_floats
        _check(lib().PvAmdSlabImportAboveHistory(self._h, _f(buf)))

    def analyze(self):
        _check(lib().PvAmdSlabAnalyze(self._h))

    def window_block(self):
        """(info4 int32 [row0 of the whole grid, col0, rows, cols], float32 [7, rows, cols])"""
        info = (C.c_int * 4)()
        n = lib().PvAmdSlabWindowBlock(self._h, info, None, 0)
        if n < 0:
            raise PlaneverbError(last_error())
        data = np.empty(max(n, 0), np.float32)
        if n > 0 and lib().PvAmdSlabWindowBlock(self._h, info, _f(data), n) != n:
            raise PlaneverbError(last_error())
        return np.array(list(info), np.int32), data


class SlabRoot:
    """whole-grid result maps of a decomposed grid (rank 0): far cells, the ranks' blocks, the direction descent"""

    def __init__(self, slab_rank, device=0):
        self._h = lib().PvAmdSlabRootCreate(slab_rank._h, int(device))
        if not self._h:
            raise PlaneverbError(last_error())
        i = PvAmdInfo()
        _check(lib().PvAmdGetInfo(slab_rank._h, i))
        self.gx, self.gy = i.gx, i.gy

    def close(self):
        if getattr(self, "_h", None):
            lib().PvAmdSlabRootDestroy(self._h)
            self._h = None

    def begin(self, listener):
        _check(lib().PvAmdSlabRootBegin(self._h, *[float(v) for v in listener]))

    def import_block(self, info4, data):
        info = (C.c_int * 4)(*[int(v) for v in info4])
        data = np.ascontiguousarray(data, np.float32)
        _check(lib().PvAmdSlabRootImportBlock(self._h, info, _f(data) if data.size else None))

    def finish(self):
        _check(lib().PvAmdSlabRootFinish(self._h))

    def get_output(self, emitter):
        o = PlaneverbOutput()
        _check(lib().PvAmdSlabRootGetOutput(self._h, *[float(v) for v in emitter], o))
        return o

    def results(self):
        res = np.empty((self.gx, self.gy, 8), np.float32)
        delay = np.empty((self.gx, self.gy), np.float32)
        _check(lib().PvAmdSlabRootCopyResults(self._h, _f(res), _f(delay)))
        return res, delay


def run_batch(solvers, listeners, wait=True):
    """PvAmdRunBatch: len(solvers) <= 8 independent runs (one listener each) advanced by ONE launch per K steps.
    The solvers must share device, grid and tile configuration; afterwards each holds its own run's results."""
    n = len(solvers)
    if n != len(listeners):
        raise ValueError("one listener position per solver")
    hs = (_vp * n)(*[sv._h for sv in solvers])
    xyz = (C.c_float * (3 * n))(*[float(v) for L in listeners for v in L])
    _check(lib().PvAmdRunBatch(hs, n, xyz, 1 if wait else 0))


class Solver:
    """Grid + FreeGrid + Analyzer of one config on one MI355X (PvAmd* handle API)."""

    def __init__(self, size_x, size_y, res, device=0, slabs=None, **options):
        """slabs = list of HIP devices, one per row slab: ONE grid decomposed into len(slabs) slabs (PvAmdCreateSlabs;
        all devices equal = several slabs on one GPU).  Same results, bit for bit."""
        if slabs is not None:
            dev = (C.c_int * len(slabs))(*[int(d) for d in slabs])
            self._h = lib().PvAmdCreateSlabs(float(size_x), float(size_y), int(res), dev, len(slabs))
        else:
            self._h = lib().PvAmdCreate(float(size_x), float(size_y), int(res), int(device))
        if not self._h:
            raise PlaneverbError(last_error())
        keys = {"dense_history": PVA_OPT_DENSE_HISTORY, "num_steps": PVA_OPT_NUM_STEPS,
                "skip_analysis": PVA_OPT_SKIP_ANALYSIS, "use_graph": PVA_OPT_USE_GRAPH,
                "steps_per_launch": PVA_OPT_STEPS_PER_LAUNCH, "tile_rows": PVA_OPT_TILE_ROWS,
                "no_free_grid": PVA_OPT_NO_FREE_GRID, "time_kernels": PVA_OPT_TIME_KERNELS,
                "tile_order": PVA_OPT_TILE_ORDER, "small_grid_kernel": PVA_OPT_SMALL_GRID_KERNEL,
                "packed_math": PVA_OPT_PACKED_MATH, "streaming_analysis": PVA_OPT_STREAMING_ANALYSIS,
                "stream_rows": PVA_OPT_STREAM_ROWS, "merged_launch": PVA_OPT_MERGED_LAUNCH,
                "edge_tiles": PVA_OPT_EDGE_TILES, "row_bands": PVA_OPT_ROW_BANDS,
                "patch_kernel": PVA_OPT_PATCH_KERNEL, "patch_strip": PVA_OPT_PATCH_STRIP,
                "lazy_far_cells": PVA_OPT_LAZY_FAR_CELLS, "stream_fuse": PVA_OPT_STREAM_FUSE, "aux_streams": PVA_OPT_AUX_STREAMS,
                "resident_kernel": PVA_OPT_RESIDENT_KERNEL, "rt60_lanes": PVA_OPT_RT60_LANES,
                "debug_lose_first_capture": PVA_OPT_DEBUG_LOSE_FIRST_CAPTURE, "stream_priority": PVA_OPT_STREAM_PRIORITY,
                "alternate_sweeps": PVA_OPT_ALTERNATE_SWEEPS, "xcd_regions": PVA_OPT_XCD_REGIONS,
                "analysis_fork": PVA_OPT_ANALYSIS_FORK, "fused_analysis": PVA_OPT_FUSED_ANALYSIS}
        for k, v in options.items():
            _check(lib().PvAmdSetOption(self._h, keys[k], int(v)))
        self.info = PvAmdInfo()
        _check(lib().PvAmdGetInfo(self._h, self.info))
        i = self.info
        self.gx, self.gy, self.T, self.fs, self.dx, self.dt, self.efree = i.gx, i.gy, i.T, i.fs, i.dx, i.dt, i.efree

    def close(self):
        if getattr(self, "_h", None):
            lib().PvAmdDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def slab_info(self):
        i = PvAmdSlabInfo()
        _check(lib().PvAmdGetSlabInfo(self._h, i))
        return i

    def load_scene(self, path):
        n = lib().PvAmdLoadScene(self._h, path.encode())
        if n < 0:
            raise PlaneverbError(last_error())
        return n

    def save_scene(self, path):
        _check(lib().PvAmdSaveScene(self._h, path.encode()))

    def add_geometry(self, aabb):
        return lib().PvAmdAddGeometry(self._h, *[float(v) for v in aabb])

    def update_geometry(self, gid, aabb):
        _check(lib().PvAmdUpdateGeometry(self._h, int(gid), *[float(v) for v in aabb]))

    def remove_geometry(self, gid):
        _check(lib().PvAmdRemoveGeometry(self._h, int(gid)))

    def run(self, listener):
        _check(lib().PvAmdRun(self._h, *[float(v) for v in listener]))

    def run_async(self, listener):
        _check(lib().PvAmdRunAsync(self._h, *[float(v) for v in listener]))

    def run_async_after(self, prev, listener):
        """a run that continues `prev`'s result map (PvAmdRunAsyncAfter: two solvers taking turns on one sequence of iterations)"""
        _check(lib().PvAmdRunAsyncAfter(self._h, prev._h, *[float(v) for v in listener]))

    def sync(self):
        _check(lib().PvAmdSync(self._h))

    def run_steps(self, nsteps, with_pulse=False, listener=(0.0, 0.0, 0.0)):
        _check(lib().PvAmdRunSteps(self._h, int(nsteps), int(with_pulse), float(listener[0]), float(listener[2])))

    def timings(self):
        t = PvAmdTimings()
        _check(lib().PvAmdGetTimings(self._h, t))
        return t

    def set_emitters(self, emitters):
        """streaming-analysis mode: the emitter positions whose wet gain / RT60 are computed"""
        e = np.ascontiguousarray(emitters, np.float32).reshape(-1, 3)
        _check(lib().PvAmdSetEmitters(self._h, _f(e), len(e)))

    def get_output(self, emitter):
        o = PlaneverbOutput()
        _check(lib().PvAmdGetOutput(self._h, *[float(v) for v in emitter], o))
        return o

    def set_output_queries(self, emitters):
        """emitter positions whose outputs every following run leaves in pinned host memory (<= 64)"""
        e = np.ascontiguousarray(np.asarray(emitters, np.float32).reshape(-1, 3))
        self._nq = len(e)
        _check(lib().PvAmdSetOutputQueries(self._h, _f(e), len(e)))

    def queried_outputs(self):
        """float32 [n_queries, 8] of the last run (after sync; no GPU work)"""
        n = getattr(self, "_nq", 0)
        out = (PlaneverbOutput * max(n, 1))()
        _check(lib().PvAmdGetQueriedOutputs(self._h, out, n))
        return np.frombuffer(out, np.float32).reshape(-1, 8)[:n].copy()

    def results(self):
        res = np.empty((self.gx, self.gy, 8), np.float32)
        delay = np.empty((self.gx, self.gy), np.float32)
        _check(lib().PvAmdCopyResults(self._h, _f(res), _f(delay)))
        return res, delay

    def results_block(self, r0, c0, nr, nc):
        """(records [nr, nc, 8], onsets [nr, nc]) of result cells [r0, r0 + nr) x [c0, c0 + nc)"""
        res = np.empty((nr, nc, 8), np.float32)
        delay = np.empty((nr, nc), np.float32)
        _check(lib().PvAmdCopyResultsBlock(self._h, int(r0), int(c0), int(nr), int(nc), _f(res), _f(delay)))
        return res, delay

    def impulse_response(self, cx, cy):
        out = np.empty((self.T, 3), np.float32)
        _check(lib().PvAmdGetImpulseResponse(self._h, int(cx), int(cy), _f(out)))
        return out

    def impulse_response_cells(self, cx, cy):
        """structured array [T] of reference Cells (pr, vx, vy, b, by)"""
        out = np.zeros(self.T, CELL_DTYPE)
        _check(lib().PvAmdGetImpulseResponseCells(self._h, int(cx), int(cy), out.ctypes.data_as(C.POINTER(PlaneverbCell))))
        return out

    def fields(self):
        shp = (self.gx + 1, self.gy + 1)
        pr, vx, vy = (np.empty(shp, np.float32) for _ in range(3))
        _check(lib().PvAmdCopyFields(self._h, _f(pr), _f(vx), _f(vy)))
        return pr, vx, vy

    def fields_local(self):
        """a slab rank's own rows of the final fields (PvAmdCopyFields on a slab copies the rows it owns)"""
        i = PvAmdInfo()
        _check(lib().PvAmdGetInfo(self._h, i))
        # rows owned = result rows, + the ghost row on the last slab: read generously, trim by what the library wrote
        bufs = [np.full((i.rows, i.gy + 1), np.nan, np.float32) for _ in range(3)]
        _check(lib().PvAmdCopyFields(self._h, _f(bufs[0]), _f(bufs[1]), _f(bufs[2])))
        n = int((~np.isnan(bufs[0][:, 0])).sum())
        return [b[:n] for b in bufs]

    def set_fields(self, pr, vx, vy):
        a = [np.ascontiguousarray(x, np.float32) for x in (pr, vx, vy)]
        _check(lib().PvAmdSetFields(self._h, _f(a[0]), _f(a[1]), _f(a[2])))

    def history_plane(self, t):
        out = np.empty((self.gx + 1, self.gy + 1), np.float32)
        _check(lib().PvAmdCopyHistoryPlane(self._h, int(t), _f(out)))
        return out

    def pulse(self):
        out = np.empty(self.T, np.float32)
        _check(lib().PvAmdCopyPulse(self._h, _f(out)))
        return out

    def material(self):
        shp = (self.gx + 1, self.gy + 1)
        beta = np.empty(shp, np.uint8)
        R = np.empty(shp, np.float32)
        _check(lib().PvAmdCopyMaterial(self._h, beta.ctypes.data_as(C.POINTER(C.c_ubyte)), _f(R)))
        return beta, R
