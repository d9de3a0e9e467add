"""The N-rank launcher of bench.py on real hardware: `LDPC_BENCH_FORCE_LAUNCH=1 python bench.py --gpus 1` goes
parent (never touches the GPU) -> child `torch.distributed.run` -> rank 0 -> `nccl` (= RCCL) process group ->
barrier-bracketed timed region -> the two all-reduces (elapsed time, six error counters) -> one JSON line.
This is the multi-GPU path of SURVEY.md section 8(e) (reference: /root/reference/src/simulation/ber.rs:304-342,
one decoder per worker, results folded by the parent) with one rank, which is all a one-GPU box can run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """a rendezvous port nobody holds (a fixed number collides with whatever else runs on the box)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _bench(argv, env=None, timeout=900):
    e = dict(os.environ)
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(v, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                       timeout=timeout, env=e, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), r


def test_forced_launch_runs_one_rccl_rank_and_matches_the_in_process_run():
    common = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "1024", "--no-cpu-baseline", "--no-realistic",
              "--no-config3", "--no-live-traffic"]
    plain, _ = _bench(common)
    forced, r = _bench(common, env={"LDPC_BENCH_FORCE_LAUNCH": "1", "NCCL_DEBUG": "VERSION"})
    assert plain["launch"]["started_by"] == "in-process" and plain["launch"]["process_group"] is None
    assert forced["launch"]["started_by"].startswith("bench.py launch_ranks")
    assert forced["launch"]["process_group"] == "nccl" and forced["n_gpus"] == 1
    # same workload, same frames (the library's Philox stream, seed 1000, frames [rank*B, (rank+1)*B)): identical error
    # counters.  The two runs are separate processes on differently placed workspaces and a 1024-frame step is short, so
    # the wall-clock rates only have to be of the same order; the kernels' own durations (HIP events) must agree
    assert forced["ber"] == plain["ber"]
    assert forced["ber"]["num_frames"] == 1024
    assert 0.5 < forced["value"] / plain["value"] < 2.0, (forced["value"], plain["value"])
    assert abs(forced["roofline"]["iteration_us"] / plain["roofline"]["iteration_us"] - 1.0) < 0.15, \
        (forced["roofline"]["iteration_us"], plain["roofline"]["iteration_us"])
    assert "RCCL" in (r.stdout + r.stderr) or "NCCL version" in (r.stdout + r.stderr)


@pytest.mark.parametrize("workload", ["config2", "config3"])
def test_eight_real_ranks_share_one_gpu(workload):
    """Eight REAL decoder processes at once -- the shape of the 8-GPU run (reference:
    /root/reference/src/simulation/ber.rs:304-342, one decoder per worker, counters folded by the parent) rehearsed on
    the one GPU a test box has: `bench.py --gpus 8 --share-device` starts eight ranks through the same launcher, every
    rank builds its own decoder on GPU 0, pins itself to its share of the CPUs, decodes frames [r*512, (r+1)*512) of the
    one frame stream, and the six counters are summed over gloo.  The sums equal those of ONE process decoding frames
    [0, 4096).  (Config 3: the layered schedule, two lanes x 35 launches per iteration per process -- sixteen enqueuing
    threads and their progress pollers at once.)  Not a throughput measurement; the line says so."""
    common = ["--steps", "2", "--warmup", "1", "--batch", "4096", "--workload", workload, "--no-cpu-baseline", "--no-realistic",
              "--no-config3", "--no-live-traffic"]
    one, _ = _bench(["--gpus", "1"] + common)
    eight, r = _bench(["--gpus", "8", "--share-device"] + common)
    assert eight["n_gpus"] == 8 and eight["share_device"] is True and eight["launch"]["process_group"] == "gloo"
    assert "REHEARSAL" in eight["config"]["workload"] and eight["roofline"] is None
    assert eight["ber"]["num_frames"] == 4096 and eight["ber"] == one["ber"], (eight["ber"], one["ber"])
    aff = eight["launch"]["cpu_affinity_rank0"]
    assert aff and "error" not in aff and aff["ranks_on_node"] == 8 and aff["cpus"] >= 1, aff
    # every rank's own figures are in the line: eight rows, ranks 0..7, each with its own clock and kernel times
    rows = eight["launch"]["per_rank"]
    assert [r["rank"] for r in rows] == list(range(8)) and all(r["ms_per_step"] > 0 and r["codewords_per_s"] > 0 for r in rows)
    assert all(r["pinned_cpus"] >= 1 and r["numa_node"] == rows[0]["numa_node"] for r in rows)
    assert eight["launch"]["slowest_rank"] in range(8)
    assert eight["launch"]["min_rank_rate_times_ranks"] <= eight["launch"]["sum_of_rank_rates"] * (1 + 1e-9)
    assert len(one["launch"]["per_rank"]) == 1
    print(workload, "enqueue ms/step: 1 process", one["launch"]["host_enqueue_ms_per_step"], "8 processes",
          eight["launch"]["host_enqueue_ms_per_step"])


def _table(path):
    """the counter columns of a result file (everything but throughput and elapsed time)"""
    rows = []
    for ln in open(path).read().splitlines():
        cells = [c.strip() for c in ln.split("|")]
        if len(cells) == 11 and cells[0].replace(".", "").replace("-", "").isdigit():
            rows.append(cells[:9])
    return rows


def test_ber_sweep_under_a_one_rank_rccl_group_matches_the_in_process_sweep(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 1 -m ldpc_toolbox_amd.ber ...` -- the multi-GPU entry of the BER
    driver (the reference's worker pool, /root/reference/src/simulation/ber.rs:304-342, as one process per GPU whose six
    counters are summed by an RCCL all-reduce between batches; stop rule of :522-531 decided on the sums) -- run as a
    child process with one rank, against the same sweep in this process without a process group: same table."""
    args = ["--code", "ar4ja:1/2:1024", "--decoder", "Minsumf32", "--puncturing", "1,1,1,1,0", "--min-ebn0", "1.5",
            "--max-ebn0", "2.5", "--step-ebn0", "0.5", "--max-iter", "50", "--frame-errors", "50", "--max-frames", "40000",
            "--frames-per-batch", "4096", "--seed", "7"]
    from ldpc_toolbox_amd import ber
    plain = tmp_path / "plain.txt"
    ber.main(args + ["--output-file", str(plain)])
    forced = tmp_path / "forced.txt"
    e = dict(os.environ, NCCL_DEBUG="VERSION", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(v, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), "-m", "ldpc_toolbox_amd.ber"] + args +
                       ["--output-file", str(forced)], capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "process group: nccl with 1 rank(s)" in r.stdout
    assert "RCCL" in (r.stdout + r.stderr) or "NCCL version" in (r.stdout + r.stderr)
    a, b = _table(plain), _table(forced)
    assert len(a) == 3 and a == b, (a, b)
    assert "Number of GPUs: 1" in open(forced).read()


def test_ber_sweep_with_eight_ranks_sharing_one_gpu_matches_the_in_process_sweep(tmp_path):
    """BASELINE config 5's shape -- the sweep over N ranks, every batch's frame indices sharded contiguously, the counters
    summed between batches, the stop rule (/root/reference/src/simulation/ber.rs:522-531) decided on the sums -- with EIGHT
    real ranks: `python -m torch.distributed.run --nproc-per-node 8 -m ldpc_toolbox_amd.ber ... --share-device` (every rank
    its own Simulator on GPU 0, gloo).  With the per-GPU batch an eighth of the in-process run's the ranks decode exactly
    the same frames batch by batch, so the two tables agree in every counter column."""
    base = ["--code", "dvbs2:R1_2short", "--decoder", "Minsumf32", "--min-ebn0", "1.2", "--max-ebn0", "1.8", "--step-ebn0", "0.3",
            "--max-iter", "40", "--frame-errors", "60", "--max-frames", "65536", "--seed", "11"]
    from ldpc_toolbox_amd import ber
    plain = tmp_path / "plain.txt"
    ber.main(base + ["--frames-per-batch", "4096", "--output-file", str(plain)])
    shared = tmp_path / "shared.txt"
    e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(v, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), "-m", "ldpc_toolbox_amd.ber"] + base +
                       ["--frames-per-batch", "512", "--share-device", "--output-file", str(shared)],
                       capture_output=True, text=True, timeout=1200, env=e, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "process group: gloo with 8 rank(s)" in r.stdout
    a, b = _table(plain), _table(shared)
    assert len(a) == 3 and a == b, (a, b)


def _dir_tables(d):
    return {f: _table(os.path.join(d, f)) for f in sorted(os.listdir(d))}


def test_multi_code_sweep_job_with_eight_ranks_sharing_one_gpu_matches_the_one_rank_job(tmp_path):
    """BASELINE.json configs[4] as ONE N-rank job (ldpc_toolbox_amd/sweep_scheduler.py): `python -m torch.distributed.run
    --nproc-per-node 8 -m ldpc_toolbox_amd.ber --codes ... --grid waterfall --share-device` -- the per-code pre-scans and
    the cheap (code, Eb/N0) points go to ranks whole from a queue in the process group's store, the points that need many
    frames are frame-sharded over all eight ranks -- against the same job in this process on one rank: the pre-scan
    places the same grids and every counter column of every row of every code's result file is equal
    (reference: /root/reference/src/simulation/ber.rs:304-342, 522-531; SURVEY.md section 8(e): both splits)."""
    # (short frames: a group of the decoder is 16384 of them, so "needs more than 8 groups" shares the points beyond 131072
    # frames -- the cap-bound ones -- and leaves the others to single ranks)
    base = ["--codes", "dvbs2:R1_2short,dvbs2:R2_3short,nr5g:2:52", "--grid", "waterfall", "--decoder", "Minsumf32",
            "--max-iter", "40", "--frame-errors", "50", "--max-frames", "262144", "--seed", "3", "--defer-groups", "8"]
    from ldpc_toolbox_amd import ber
    one = tmp_path / "one"
    ber.main(base + ["--output-dir", str(one)])
    eight = tmp_path / "eight"
    e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(v, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), "-m", "ldpc_toolbox_amd.ber"] + base +
                       ["--share-device", "--verbose", "--output-dir", str(eight)],
                       capture_output=True, text=True, timeout=1500, env=e, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "process group: gloo with 8 rank(s)" in r.stdout
    assert "queue: store" in r.stdout, r.stdout[-2000:]
    a, b = _dir_tables(one), _dir_tables(eight)
    assert sorted(a) == sorted(b) and len(a) == 3
    for f in a:
        assert len(a[f]) == 8 and a[f] == b[f], (f, a[f], b[f])
    rows = [row for f in a for row in a[f]]
    frames = [int(row[1]) for row in rows]
    assert max(frames) == 262144 and min(frames) < 262144             # cap-bound points and cheap ones
    assert "-> shared" in r.stdout and "shared by all ranks" in r.stdout
    print(r.stdout[-1500:])
