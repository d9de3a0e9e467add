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
    # every rank's own figures travel with the line (round 6): a slow rank is visible instead of hiding in the MAX
    rows = out["launch"]["per_rank"]
    assert [r["rank"] for r in rows] == [0, 1] and all(r["ms_per_step"] >= 20.0 and r["codewords_per_s"] > 0 for r in rows)
    assert out["launch"]["slowest_rank"] in (0, 1)
    assert out["launch"]["min_rank_rate_times_ranks"] <= out["launch"]["sum_of_rank_rates"] * (1 + 1e-9)
    # the job's rate cannot exceed what its slowest rank sustains, times the ranks
    assert out["value"] <= out["launch"]["min_rank_rate_times_ranks"] * (1 + 1e-6)


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


# ---- several codes as one N-rank job (ldpc_toolbox_amd/sweep_scheduler.py; BASELINE.json configs[4]) ---------------------------

class _FakeCodeSim:
    """stands in for the GPU simulator of ONE code: counters are a pure function of (code, Eb/N0, seed, frame index), so
    any split of a point's frames over calls and ranks sums to the same counters.  The frame error rate falls by a factor
    of 10 every 0.1 dB from a code-dependent waterfall: the grid's points run from "every frame fails" to "runs to the cap"."""
    k, n, n_tx = 100, 200, 200

    def __init__(self, code, calls=None):
        self.code = code
        self.shift = {"codeA": 1.0, "codeB": 1.7, "codeC": 2.4}[code]
        self.rate = {"codeA": 0.4, "codeB": 0.5, "codeC": 0.6}[code]
        self.calls = calls if calls is not None else []

    def get(self, key):
        assert key == "preferred_batch"
        return 64

    def run(self, ebn0_db, seed, first_frame, frames, max_iterations, bch_max_errors=0):
        self.calls.append((self.code, round(ebn0_db, 2), first_frame, frames))
        fer = min(1.0, 10.0 ** (-(ebn0_db - self.shift) * 10.0))
        idx = np.arange(first_frame, first_frame + frames, dtype=np.uint64)
        h = (idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        h ^= h >> np.uint64(29)
        h = (h * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        u = (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        bad = u < fer
        errs = np.where(bad, 1 + (idx % np.uint64(5)).astype(np.int64), 0)
        its = np.where(bad, max_iterations, 5)
        return np.array([frames, int(errs.sum()), int(bad.sum()), 0, int(its.sum()), int(its[~bad].sum())], dtype=np.int64)


_JOB = dict(max_iterations=20, max_frame_errors=40, max_frames=64 * 256, seed=5)
_CODES = ["codeA", "codeB", "codeC"]


def _job_tables(job):
    from ldpc_toolbox_amd import sweep_scheduler as ss
    scans = job.prescan(frames=256)
    grids = [ss.waterfall_grid(d["cross"]) for d in scans]
    pts = job.run(grids)
    return [d["cross"] for d in scans], [[p.code_index, round(p.ebn0_db, 2)] + [int(x) for x in p.total] for p in pts], pts


def _multi_worker(rank, world, port, out_dir, queue):
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from ldpc_toolbox_amd import sweep_scheduler as ss
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    job = ss.SweepJob(_CODES, lambda c: _FakeCodeSim(c, calls), rank=rank, world=world, queue=queue, **_JOB)
    cross, table, pts = _job_tables(job)
    json.dump({"cross": cross, "table": table, "calls": calls, "timeline": job.timeline,
               "alone": [p.index for p in pts if p.owner == rank and not p.deferred],
               "deferred": [p.index for p in pts if p.deferred]},
              open(os.path.join(out_dir, f"m{queue}{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,queue", [(2, "store"), (2, "static"), (3, "store")])
def test_multi_code_sweep_tables_do_not_depend_on_the_rank_count(tmp_path, world, queue):
    """sweep_scheduler.SweepJob with N gloo ranks against one rank: the pre-scan crossings and EVERY counter of every
    (code, Eb/N0) row are equal; cheap points were run by single ranks (whole, nobody else touched their frames), the
    long ones were deferred after their first call and every later batch of theirs was split over all ranks."""
    import json
    from ldpc_toolbox_amd import sweep_scheduler as ss
    one_calls = []
    one = ss.SweepJob(_CODES, lambda c: _FakeCodeSim(c, one_calls), **_JOB)
    want_cross, want_table, pts = _job_tables(one)
    assert len(want_table) == 24 and one.timeline["queue"] == "local" and one.timeline["points_shared"] == 0
    frames = [row[2] for row in want_table]
    assert min(frames) == 64 and max(frames) == _JOB["max_frames"]          # from "one group is enough" to the cap
    mp.spawn(_multi_worker, args=(world, _free_port(), str(tmp_path), queue), nprocs=world, join=True)
    got = [json.load(open(tmp_path / f"m{queue}{r}.json")) for r in range(world)]
    for r in range(world):
        assert got[r]["cross"] == want_cross
        assert got[r]["table"] == want_table, r
        assert got[r]["timeline"]["queue"] == queue
    deferred = got[0]["deferred"]
    assert deferred and all(g["deferred"] == deferred for g in got)
    alone = sorted(i for g in got for i in g["alone"])
    assert alone == sorted(set(range(24)) - set(deferred))                 # every other point: exactly one rank, whole
    if queue == "static":
        assert all(i % world == r for r in range(world) for i in got[r]["alone"])
    # phase B: every rank decoded a shard of every batch of every deferred point, and the shards tile the batch
    for i in deferred:
        code, e = _CODES[want_table[i][0]], want_table[i][1]
        per_rank = [sorted((f, n) for (c, ee, f, n) in g["calls"] if c == code and ee == e and f >= 64) for g in got]
        assert all(len(x) == len(per_rank[0]) and x for x in per_rank), (i, per_rank)
        for b in range(len(per_rank[0])):
            spans = sorted(per_rank[r][b] for r in range(world))
            for (f0, n0), (f1, n1) in zip(spans, spans[1:]):
                assert f0 + n0 == f1
    # the one-rank job and the N-rank job decoded the same number of frames at every (code, Eb/N0) -- pre-scan calls included
    for i, row in enumerate(want_table):
        code, e = _CODES[row[0]], row[1]
        n_frames = sum(n for g in got for (c, ee, f, n) in g["calls"] if c == code and ee == e)
        n_one = sum(n for (c, ee, f, n) in one_calls if c == code and ee == e)
        assert n_frames == n_one, (i, n_frames, n_one)


def test_batch_schedule_is_a_function_of_the_counters_only():
    from ldpc_toolbox_amd import sweep_scheduler as ss
    z = np.zeros(6, dtype=np.int64)
    assert ss.batch_frames(z, 4096, 100, None, 2) == 4096
    seen = np.array([4096, 0, 0, 0, 0, 0])
    assert ss.batch_frames(seen, 4096, 100, None, 2) == 64 * 4096            # no error yet: a long batch
    assert ss.batch_frames(seen, 4096, 100, 4096 + 1000, 2) == 1000          # ... clipped by the frame cap
    many = np.array([4096, 0, 50, 0, 0, 0])
    assert ss.batch_frames(many, 4096, 100, None, 2) == 8 * 4096
    assert ss.frames_needed(many, 100, 2) == 4096.0 and ss.frames_needed(np.array([10, 0, 100, 0, 0, 0]), 100, 2) == 0.0
    assert ss.counters_stop(np.array([10, 0, 100, 0, 0, 0]), 100, None, 2) and not ss.counters_stop(many, 100, None, 2)
    assert ss.counters_stop(many, 100, 4096, 2)
    assert ss.expand_codes("dvbs2:normal, nr5g:1:384") == ["dvbs2:" + r for r in ss.DVBS2_NORMAL] + ["nr5g:1:384"]
    p = ss.Point(3, 1, 1.25, 6)
    p.total[:] = [5, 4, 3, 2, 1, 0]
    p.deferred, p.owner, p.elapsed = True, 2, 0.5
    q = ss.Point.from_json(p.to_json())
    assert (q.index, q.code_index, q.ebn0_db, q.deferred, q.owner, q.elapsed, list(q.total)) == (3, 1, 1.25, True, 2, 0.5, [5, 4, 3, 2, 1, 0])


def _bch_worker(rank, world, port, out_dir):
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from ldpc_toolbox_amd import sweep_scheduler as ss
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    job = ss.SweepJob(_CODES[:2], lambda c: _FakeBchSim(c), rank=rank, world=world, max_iterations=20, max_frame_errors=30,
                      max_frames=64 * 200, seed=9, bch_max_errors=3, defer_groups=4)
    pts = job.run([[1.2, 1.3, 1.5], [1.9, 2.0, 2.2]])
    # the clock rules: a point that has its errors still runs until --min-time has passed (rank 0's clock decides for all)
    timed = ss.SweepJob(_CODES[:1], lambda c: _FakeBchSim(c), rank=rank, world=world, max_iterations=20, max_frame_errors=5,
                        max_frames=None, min_time=0.3, max_time=2.0, seed=9, defer_groups=1)
    tp = timed.run([[1.25]])
    json.dump({"table": [[p.code_index, p.ebn0_db] + [int(x) for x in p.total] for p in pts],
               "timed": [[int(x) for x in p.total] + [p.elapsed, p.deferred] for p in tp]},
              open(os.path.join(out_dir, f"bch{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


class _FakeBchSim(_FakeCodeSim):
    def run(self, ebn0_db, seed, first_frame, frames, max_iterations, bch_max_errors=0):
        self.calls.append((self.code, round(ebn0_db, 2), first_frame, frames))
        fer = min(1.0, 10.0 ** (-(ebn0_db - self.shift) * 10.0))
        idx = np.arange(first_frame, first_frame + frames, dtype=np.uint64)
        h = (idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        h ^= h >> np.uint64(29)
        h = (h * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        bad = (h >> np.uint64(11)).astype(np.float64) / float(1 << 53) < fer
        errs = np.where(bad, 1 + (idx % np.uint64(5)).astype(np.int64), 0)
        its = np.where(bad, max_iterations, 5)
        c = [frames, int(errs.sum()), int(bad.sum()), 0, int(its.sum()), int(its[~bad].sum())]
        if bch_max_errors > 0:      # outer-BCH view of the same frames (ber.rs:328-337), per frame: additive over any split
            worse = errs > bch_max_errors
            c += [int(errs[worse].sum()), int(worse.sum()), int(its[~worse].sum())]
        return np.array(c, dtype=np.int64)


def test_multi_code_sweep_with_bch_accounting_and_clock_rules(tmp_path):
    """the nine-counter form (outer-BCH accounting: the BCH frame errors stop a point, ber.rs:514-520) through both phases, and
    the time-based stop rules in the shared phase (rank 0's clock travels with the counters): every rank ends with the same
    table, equal to one rank's in the deterministic case"""
    import json
    from ldpc_toolbox_amd import sweep_scheduler as ss
    one = ss.SweepJob(_CODES[:2], lambda c: _FakeBchSim(c), max_iterations=20, max_frame_errors=30, max_frames=64 * 200, seed=9,
                      bch_max_errors=3, defer_groups=4)
    want = [[p.code_index, p.ebn0_db] + [int(x) for x in p.total] for p in one.run([[1.2, 1.3, 1.5], [1.9, 2.0, 2.2]])]
    assert all(len(r) == 2 + 9 for r in want)
    mp.spawn(_bch_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = [json.load(open(tmp_path / f"bch{r}.json")) for r in range(2)]
    assert got[0]["table"] == want and got[1]["table"] == want
    # the clock-ruled point: both ranks agree on its counters (rank 0 decided), it was shared, and it ran at least --min-time
    assert got[0]["timed"][0][:6] == got[1]["timed"][0][:6]
    assert got[0]["timed"][0][-1] is True and got[0]["timed"][0][-2] >= 0.3 and got[0]["timed"][0][2] >= 5
