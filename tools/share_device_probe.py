#!/usr/bin/env python3
"""Eight REAL ranks on one GPU (bench.py --gpus 8 --share-device, gloo) against one process, for both launcher workloads:
counters, per-process enqueue time (the host side of a step: a call on the caller's stream returns when its launches are
enqueued) with and without per-rank CPU pinning.  A rehearsal of the 8-GPU run's host side, NOT a throughput measurement
(profiles/r05_share_device.txt).   python3 tools/share_device_probe.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCHES = {"config2": "50 iterations x (check + variable + vn_free launch) + checkpoints = about 190 launches per step, one lane",
            "config3": "50 iterations x 35 launches x 2 lanes (two enqueuing threads) = about 3500 launches per step and process"}


def bench(argv):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=e, cwd=ROOT, timeout=900)
    if r.returncode != 0:
        print(r.stderr[-2000:])
        raise SystemExit(1)
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


for wl in ("config2", "config3"):
    common = ["--steps", "3", "--warmup", "1", "--batch", "4096", "--workload", wl, "--no-cpu-baseline", "--no-realistic", "--no-config3",
              "--no-live-traffic"]
    one = bench(["--gpus", "1"] + common)
    print(f"== {wl}: {one['config']['workload']}\n   ({LAUNCHES[wl]})")
    print(f"   1 process , 4096 frames      : {one['value']:9.0f} cw/s   enqueue {one['launch']['host_enqueue_ms_per_step']['max_over_ranks']:7.2f} ms/step   "
          f"counters {one['ber']['bit_errors']} bit errors / {one['ber']['num_frames']} frames")
    for extra, label in (([], "pinned (default)"), (["--no-affinity"], "not pinned")):
        eight = bench(["--gpus", "8", "--share-device"] + extra + common)
        h = eight["launch"]["host_enqueue_ms_per_step"]
        same = eight["ber"] == one["ber"]
        print(f"   8 processes x 512, {label:17s}: {eight['value']:9.0f} cw/s   enqueue {h['min_over_ranks']:7.2f} .. {h['max_over_ranks']:7.2f} ms/step   "
              f"counters equal to the one-process run: {same}   affinity of rank 0: {eight['launch']['cpu_affinity_rank0']}")
