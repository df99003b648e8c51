"""CPU: the N > 1 path (run sharding + the one gather of per-emitter outputs) over gloo with world_size 2."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_runs", [8, 5, 1])
def test_gather_outputs_gloo_world2(n_runs, tmp_path):
    import subprocess
    from _dist_worker import fake_result
    port = str(_free_port())
    worker = os.path.join(ROOT, "tests", "_dist_worker.py")
    outs = [str(tmp_path / ("rank%d.npz" % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, str(n_runs), outs[r]]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    want = np.stack([fake_result(k) for k in range(n_runs)])
    owned = []
    for r in range(2):
        d = np.load(outs[r])
        assert np.array_equal(d["out"], want), r
        owned += list(d["mine"])
    assert sorted(owned) == list(range(n_runs))  # every run simulated exactly once


def test_shard_runs_partition():
    from planeverb_amd import dist as pvd
    for n in (0, 1, 7, 8, 64):
        for w in (1, 2, 4, 8):
            parts = [pvd.shard_runs(n, w, r) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_single_process_gather():
    from planeverb_amd import dist as pvd
    local = {0: np.ones((2, 8), np.float32), 1: np.full((2, 8), 2, np.float32)}
    out = pvd.gather_outputs(local, 2, None)
    assert out.shape == (2, 2, 8) and out[1, 0, 0] == 2
