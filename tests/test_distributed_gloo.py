"""CPU, world_size 2 over gloo: the N>1 path of the benchmark -- contiguous sharding of the
codeword batch and the host-side counter sum (the only exchange; no data-path collective)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from ldpc_toolbox_amd import sharding, simulation as sim
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    begin, end = sharding.shard_range(total, rank, world)
    # every rank "decodes" its own frames: here a deterministic stand-in for the decoder output
    k = 8
    rng = np.random.default_rng(123)
    msgs = rng.integers(0, 2, (total, k), dtype=np.uint8)
    bits = msgs.copy()
    bits[::5, 0] ^= 1
    its = np.where(np.arange(total) % 7 == 0, -1, 4).astype(np.int32)
    st = sim.fold_statistics(0.0, k, msgs[begin:end], bits[begin:end], its[begin:end], 50, 1.0)
    total_counters = sharding.reduce_counters(sharding.counters_from_statistics(st))
    # max-over-ranks timing, as bench.py does
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.concatenate([total_counters, [t.item()], [begin, end]]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_counters_sum_to_single_process(tmp_path):
    from ldpc_toolbox_amd import sharding, simulation as sim
    world, total, k = 2, 101, 8
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(123)
    msgs = rng.integers(0, 2, (total, k), dtype=np.uint8)
    bits = msgs.copy()
    bits[::5, 0] ^= 1
    its = np.where(np.arange(total) % 7 == 0, -1, 4).astype(np.int32)
    want = sharding.counters_from_statistics(sim.fold_statistics(0.0, k, msgs, bits, its, 50, 1.0))
    got = [np.load(tmp_path / f"r{r}.npy") for r in range(world)]
    for r in range(world):
        assert np.array_equal(got[r][:6].astype(np.int64), want)
        assert got[r][6] == world                         # MAX over ranks of (1 + rank)
    assert got[0][7] == 0 and got[0][8] == got[1][7] and got[1][8] == total


class _FakeSim:
    """stands in for the GPU simulator: counters are a pure function of the frame indices"""
    k = 10

    def run(self, ebn0_db, seed, first_frame, frames, max_iterations, bch_max_errors=0):
        idx = np.arange(first_frame, first_frame + frames)
        bad = (idx % 7 == 0) if ebn0_db < 2.0 else (idx % 501 == 0)
        errs = np.where(bad, 1 + idx % 5, 0)                  # bit errors of the bad frames: 1..5
        its = np.where(bad, max_iterations, 5)
        c = [frames, int(errs.sum()), int(bad.sum()), 0, int(its.sum()), int(its[~bad].sum())]
        if bch_max_errors > 0:                                # ber.rs:328-337
            worse = errs > bch_max_errors
            c += [int(errs[worse].sum()), int(worse.sum()), int(its[~worse].sum())]
        return np.array(c, dtype=np.int64)


def _sweep_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from ldpc_toolbox_amd import ber
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = ber.sweep(_FakeSim(), [1.0, 3.0], max_iterations=20, max_frame_errors=50, max_frames=3000,
                    frames_per_batch=100, seed=0, rank=rank, world=world)
    np.save(os.path.join(out_dir, f"s{rank}.npy"),
            np.array([[r.num_frames, r.ldpc.bit_errors, r.ldpc.frame_errors, r.total_iterations] for r in res]))
    res = ber.sweep(_FakeSim(), [1.0], max_iterations=20, max_frame_errors=50, max_frames=3000,
                    frames_per_batch=100, seed=0, rank=rank, world=world, bch_max_errors=3)
    np.save(os.path.join(out_dir, f"b{rank}.npy"),
            np.array([[r.num_frames, r.ldpc.frame_errors, r.bch.bit_errors, r.bch.frame_errors, r.bch.correct_iterations]
                      for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def test_ber_sweep_two_ranks_equals_one(tmp_path):
    """the sweep driver's N>1 path: frame indices sharded per batch, counters all-reduced, the same
    stop decision on every rank -- totals equal a single-process run over the same frames"""
    from ldpc_toolbox_amd import ber
    world = 2
    mp.spawn(_sweep_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    single = ber.sweep(_FakeSim(), [1.0, 3.0], max_iterations=20, max_frame_errors=50, max_frames=3000,
                       frames_per_batch=200, seed=0)          # one rank doing both shares per batch
    want = np.array([[r.num_frames, r.ldpc.bit_errors, r.ldpc.frame_errors, r.total_iterations] for r in single])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"s{r}.npy"), want)
    assert want[0][0] < 3000 and want[0][2] >= 50          # first point stopped on frame errors
    assert want[1][0] == 3000                                # second ran to max_frames
    # outer-BCH accounting: the BCH frame errors (frames with more than 3 bit errors) stop the run
    # (ber.rs:514-520), so it runs longer than the LDPC-only sweep, identically on every rank
    single = ber.sweep(_FakeSim(), [1.0], max_iterations=20, max_frame_errors=50, max_frames=3000,
                       frames_per_batch=200, seed=0, bch_max_errors=3)
    wantb = np.array([[r.num_frames, r.ldpc.frame_errors, r.bch.bit_errors, r.bch.frame_errors, r.bch.correct_iterations]
                      for r in single])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"b{r}.npy"), wantb)
    assert wantb[0][3] >= 50 and wantb[0][1] > wantb[0][3] and wantb[0][0] > want[0][0]


# ---- bench.py --gpus N: the benchmark starts its own ranks ------------------------------------

def _run_bench(argv, env=None, timeout=600):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(v, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, capture_output=True, text=True,
                          timeout=timeout, env=e, cwd=root)


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` with no torchrun environment must itself start 2 ranks (one child
    torch.distributed.run), reduce over them and print ONE line with n_gpus = 2.  --stub: gloo on CPU and
    a stand-in for the decoder (20 ms per step, every frame 'fails'), so the launcher, the barrier-bracketed
    timing, the max-over-ranks reduction and the counter sum are what is exercised."""
    import json
    r = _run_bench(["--gpus", "2", "--stub", "--steps", "3", "--warmup", "1", "--batch", "64"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["data"].startswith("stub")                       # never mistaken for a measurement
    assert out["ber"]["num_frames"] == 2 * 64                   # counters summed over both ranks
    assert out["ms_per_step"] >= 20.0                           # the stand-in sleeps 20 ms per step
    # whole-job aggregate: both ranks' codewords over the (max-over-ranks) time
    assert abs(out["value"] - 2 * 64 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]


def test_bench_refuses_a_world_size_mismatch():
    """under torchrun-style variables --gpus must equal WORLD_SIZE: a silent 1-rank run is the failure
    mode this guards against"""
    r = _run_bench(["--gpus", "2", "--stub"], env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                                                   "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_bench_child_environment_and_cpu_list_helpers(monkeypatch):
    """bench.py's counter-pass children must not inherit the parent's rendezvous / launcher / profiler variables (a child
    with RANK / MASTER_PORT would join the parent's process group), live passes are skipped under a profiler, and the
    NUMA cpulist parser reads the sysfs format"""
    import importlib
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    for k, v in {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29500",
                 "LDPC_BENCH_FORCE_LAUNCH": "1", "LDPC_BENCH_CHILD": "1", "TORCHELASTIC_RUN_ID": "x", "LD_PRELOAD": "/x/librocprofiler-sdk-tool.so",
                 "ROCP_TOOL_LIBRARIES": "/x/lib.so", "KEEP_ME": "1"}.items():
        monkeypatch.setenv(k, v)
    env = bench.child_env()
    assert env.get("KEEP_ME") == "1" and env["TMPDIR"] == "/tmp"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LDPC_BENCH_FORCE_LAUNCH", "LDPC_BENCH_CHILD",
              "TORCHELASTIC_RUN_ID", "LD_PRELOAD", "ROCP_TOOL_LIBRARIES"):
        assert k not in env, k
    assert bench.under_profiler()
    monkeypatch.delenv("ROCP_TOOL_LIBRARIES")
    monkeypatch.delenv("LD_PRELOAD")
    assert not bench.under_profiler()
    assert bench._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and bench._cpulist("") == []
    info = bench.host_cpu_info()
    assert info["hardware_threads"] >= 1 and info["cpu_model"]
