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


@pytest.mark.parametrize("n_runs", [8, 64, 3])
def test_gather_outputs_gloo_world8(n_runs, tmp_path):
    """the REAL multi-GPU shape on CPU: eight ranks under gloo -- BASELINE config 4 (8 runs, one per rank),
    config 5 (64 runs, eight per rank, two in flight) and fewer runs than ranks; run -> rank mapping, gather order, the native
    gather's packing against torch's (tests/_dist_worker.py)"""
    import subprocess
    from _dist_worker import fake_result
    port = str(_free_port())
    worker = os.path.join(ROOT, "tests", "_dist_worker.py")
    W = 8
    outs = [str(tmp_path / ("rank%d.npz" % r)) for r in range(W)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(W), port, str(n_runs), outs[r]]) for r in range(W)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    want = np.stack([fake_result(k) for k in range(n_runs)])
    for r in range(W):
        d = np.load(outs[r])
        assert np.array_equal(d["out"], want), r
        assert list(d["mine"]) == list(range(r, n_runs, W))


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


@pytest.mark.parametrize("world", [2, 3])
def test_slab_exchange_schedule_gloo(world, tmp_path):
    """SURVEY.md 8f N4 across processes: planeverb_amd.dist_slabs.run_rank (per launch: K boundary rows both ways between
    neighbouring ranks; per run: the last row's history to the rank below, the window blocks to rank 0) over gloo, on a toy
    slab with exact integer arithmetic (tests/_slab_toy.py): final fields and gathered maps EQUAL the undivided domain's"""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _slab_toy import analysis, whole_domain
    port = str(_free_port())
    worker = os.path.join(ROOT, "tests", "_slab_worker.py")
    outs = [str(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), port, outs[r]]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    NX, cols, T, K = 60, 17, 23, 4
    data = [np.load(o) for o in outs]
    for i, src in enumerate([(NX // world, 5), (NX // world - 1, 9), (3, 2)]):
        final, hist = whole_domain(NX, cols, T, K, src)
        got = np.concatenate([d["f%d" % i] for d in data])
        assert np.array_equal(got, final), "final field, source %s" % (src,)
        want = analysis(hist, np.zeros((T, cols)))
        assert np.array_equal(data[0]["m%d" % i], want), "gathered maps, source %s" % (src,)
    assert final.any() and want.any()


def test_slab_schedule_local_equals_undivided():
    """the lock-step form (dist_slabs.run_local: all ranks in one process, how a one-GPU box runs the decomposition)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from planeverb_amd import dist_slabs
    from _slab_toy import ToyRoot, ToySlab, analysis, whole_domain
    NX, cols, T, K = 60, 17, 23, 4
    for world in (2, 4):
        src = (NX // world, 5)
        slabs = [ToySlab(NX, cols, T, K, r, world, src) for r in range(world)]
        root = ToyRoot(NX, cols)
        dist_slabs.run_local(slabs, root, (0, 0, 0))
        final, hist = whole_domain(NX, cols, T, K, src)
        assert np.array_equal(np.concatenate([s.u[K:K + s.n] for s in slabs]), final)
        assert np.array_equal(root.maps, analysis(hist, np.zeros((T, cols)))) and root.finished


@pytest.mark.parametrize("shape", ["config4", "config5"])
def test_bench_orchestration_gloo_world8(shape, tmp_path):
    """bench.py's orchestration with EIGHT ranks on CPU, in the two shapes the driver's 8-GPU node will see: config 4 (one run
    per GPU and step: --inflight 1) and config 5 (two runs in flight per GPU; 4 steps = 64 runs, eight per rank), once through the
    native gather's code path (a stand-in communicator over gloo) and once through torch's (PV_BENCH_GATHER=torch): both verify
    every gathered record against the reference's, and must agree on every number that does not depend on time."""
    import json
    import subprocess
    worker = os.path.join(ROOT, "tests", "_bench_worker.py")
    W = 8
    lines = {}
    for how in ("native", "torch"):
        port = str(_free_port())
        outs = [str(tmp_path / ("%s_rank%d.txt" % (how, r))) for r in range(W)]
        procs = []
        for r in range(W):
            env = dict(os.environ, WORLD_SIZE=str(W), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                       PV_BENCH_BACKEND="gloo", PV_BENCH_WORKER_SHAPE=shape)
            env.pop("PV_BENCH_FORCE_DIST", None)
            env.pop("PV_BENCH_GATHER", None)
            if how == "torch":
                env["PV_BENCH_GATHER"] = "torch"
            procs.append(subprocess.Popen([sys.executable, worker, "gloo-comm", outs[r]], env=env))
        for p in procs:
            assert p.wait(timeout=600) == 0
        for r in range(1, W):
            assert open(outs[r]).read().strip() == "", "only rank 0 prints"
        l0 = [l for l in open(outs[0]).read().splitlines() if l.strip()]
        assert len(l0) == 1, l0
        lines[how] = json.loads(l0[0])
    inflight, steps = (1, 3) if shape == "config4" else (2, 4)
    for how, j in lines.items():
        assert j["n_gpus"] == W and j["steps"] == steps and j["scaling"] == "weak"
        assert j["config"]["runs_in_flight_per_gpu"] == inflight
        assert j["timed_runs"] == 5 * steps * inflight * W and j["verified_runs"] == j["timed_runs"], how
        assert [r["rank"] for r in j["ranks"]] == list(range(W)) and [r["local_rank"] for r in j["ranks"]] == list(range(W))
        for r in j["ranks"]:
            assert len(r["block_s"]) == 5 and all(b > 0 for b in r["block_s"])
            assert ("ncclAllGather" in r["gather"]) == (how == "native") and ("PV_BENCH_GATHER=torch" in r["gather"]) == (how == "torch")
        assert j["value"] == pytest.approx(W * inflight * 4097 * 4097 * 435 * steps / (j["ms_per_step"] * steps * 1e-3), rel=1e-9)
    assert lines["native"]["timed_runs"] == lines["torch"]["timed_runs"]
    assert lines["native"]["roofline"]["algorithmic_bytes_per_launch"] == lines["torch"]["roofline"]["algorithmic_bytes_per_launch"]


@pytest.mark.parametrize("comm_mode", ["fail", "fail-rank1"])
def test_bench_orchestration_gloo_world2(comm_mode, tmp_path):
    """bench.py's own N = 2 orchestration on CPU (PV_BENCH_BACKEND=gloo, stand-in solver, tests/_bench_worker.py): the
    first multi-rank contact of the process group, the communicator fail-over (all ranks or none), the barriers, the
    max-over-ranks timing and the rank-0-only JSON line happens here, not on the driver's 8-GPU node."""
    import json
    import subprocess
    port = str(_free_port())
    worker = os.path.join(ROOT, "tests", "_bench_worker.py")
    outs = [str(tmp_path / ("rank%d.txt" % r)) for r in range(2)]
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   PV_BENCH_BACKEND="gloo")
        env.pop("PV_BENCH_FORCE_DIST", None)
        env.pop("PV_BENCH_GATHER", None)
        procs.append(subprocess.Popen([sys.executable, worker, comm_mode, outs[r]], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    lines0 = [l for l in open(outs[0]).read().splitlines() if l.strip()]
    assert open(outs[1]).read().strip() == "", "only rank 0 prints"
    assert len(lines0) == 1, lines0
    j = json.loads(lines0[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 2 and j["scaling"] == "weak"
    assert j["metric"] == "grid_cell_updates_per_s" and j["higher_is_better"] is True
    # steps x in flight x ranks, in every one of the 5 timed blocks (bench.py --repeats); every block's records verified
    assert j["repeats"] == 5 and j["timed_runs"] == 5 * 3 * 2 * 2 and j["verified_runs"] == j["timed_runs"]
    sp = j["spread"]
    assert sp["repeats"] == 5 and len(sp["ms_per_step_all"]) == 5
    assert sp["ms_per_step_min"] <= sp["ms_per_step_median"] <= sp["ms_per_step_max"]
    assert j["ms_per_step"] == pytest.approx(sp["ms_per_step_median"], rel=1e-9)
    # one record per rank: its own block times, warm-up and communicator set-up seconds, how it gathered
    assert [r["rank"] for r in j["ranks"]] == [0, 1]
    for r in j["ranks"]:
        assert len(r["block_s"]) == 5 and all(b > 0 for b in r["block_s"])
        assert r["warmup_s"] >= 0 and r["comm_init_and_first_gather_s"] >= 0
        assert "torch.distributed.all_gather_into_tensor" in r["gather"]
        # stream placement is the library's business (round 5): bench.py only says so
        assert "claimOwnQueue" in r["stream_placement"]["in"]
    assert j["roofline"]["analysis"]["cells_per_run"] == 4096 * 4096
    assert "torch.distributed.all_gather_into_tensor" in j["config"]["gather"]
    assert j["config"]["backend"] == "gloo"
    assert j["value"] == pytest.approx(2 * 2 * 4097 * 4097 * 435 * 3 / (j["ms_per_step"] * 3e-3), rel=1e-9)
    assert j["cpu_baseline"] is not None and j["cpu_baseline"]["cores"] == 1 and j["cpu_baseline"]["value"] > 0
    assert j["roofline"]["frac"] > 0
