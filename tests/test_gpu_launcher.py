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
    # same workload, same frames (seed = 1000 + rank): identical error counters; the two runs are separate processes on
    # differently placed workspaces (placement alone moves a run by several percent), so the rates only have to be of
    # the same order
    assert forced["ber"] == plain["ber"]
    assert forced["ber"]["num_frames"] == 1024
    assert 0.5 < forced["value"] / plain["value"] < 2.0, (forced["value"], plain["value"])
    assert "RCCL" in (r.stdout + r.stderr) or "NCCL version" in (r.stdout + r.stderr)


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
                        "127.0.0.1", "--master-port", "29533", "-m", "ldpc_toolbox_amd.ber"] + args +
                       ["--output-file", str(forced)], capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "process group: nccl with 1 rank(s)" in r.stdout
    assert "RCCL" in (r.stdout + r.stderr) or "NCCL version" in (r.stdout + r.stderr)
    a, b = _table(plain), _table(forced)
    assert len(a) == 3 and a == b, (a, b)
    assert "Number of GPUs: 1" in open(forced).read()
