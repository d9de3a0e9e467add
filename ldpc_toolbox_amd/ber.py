"""BER-vs-Eb/N0 sweep on the GPU(s): the batched counterpart of `ldpc-toolbox ber`
(/root/reference/src/cli/ber.rs:90-158 + src/simulation/ber.rs:297-368): BPSK or 8PSK (with the DVB-S2
bit interleaver) over AWGN, optional outer-BCH accounting.

Frames are generated, decoded and scored on the device (Simulator / include/ldpc_toolbox.h
part 3); per batch each rank handles a contiguous share of the frame indices and only the six
error counters are summed across ranks (no collective on the data path).  The stop rule per
Eb/N0 is the reference's (frame errors >= --frame-errors and elapsed >= --min-time, or elapsed >=
--max-time), evaluated between batches on the summed counters, plus an optional --max-frames.

  python -m ldpc_toolbox_amd.ber --code dvbs2:R1_2 --decoder Minsumf32 --min-ebn0 1.0 --max-ebn0 2.0 --step-ebn0 0.25
  python -m torch.distributed.run --nproc-per-node 8 -m ldpc_toolbox_amd.ber ...

Several codes as one job (BASELINE.json configs[4]: all DVB-S2 normal-frame rates x 8 Eb/N0 points on 8 GPUs): `--codes`,
scheduled by sweep_scheduler.py -- whole (code, Eb/N0) points to ranks from a shared queue, frame-sharding only for the
points that run long:

  python -m torch.distributed.run --nproc-per-node 8 -m ldpc_toolbox_amd.ber --codes dvbs2:normal --grid waterfall \
         --decoder Minsumf32 --max-iter 50 --max-frames 1048576 --output-dir out/
"""
import argparse
import os
import time

import numpy as np

from . import _capi, sharding
from .decoder import Simulator
from .simulation import (CodeStatistics, Statistics, _finish_code_statistics, format_duration, format_header,
                         format_progress)


def statistics_from_counters(ebn0_db, k, c, elapsed) -> Statistics:
    """ber.rs:551-581 from the six summed counters (nine with the outer-BCH accounting)"""
    st = Statistics(ebn0_db=ebn0_db)
    (st.num_frames, st.ldpc.bit_errors, st.ldpc.frame_errors, st.false_decodes, st.total_iterations,
     st.ldpc.correct_iterations) = (int(x) for x in c[:6])
    if len(c) >= 9:
        st.bch = CodeStatistics(bit_errors=int(c[6]), frame_errors=int(c[7]), correct_iterations=int(c[8]))
        _finish_code_statistics(st.bch, k, st.num_frames)
    n = st.num_frames
    st.elapsed = elapsed
    st.average_iterations = st.total_iterations / n if n else 0.0
    st.throughput_mbps = 1e-6 * k * n / elapsed if elapsed > 0 else 0.0
    st.ldpc.ber = st.ldpc.bit_errors / (k * n) if n else 0.0
    st.ldpc.fer = st.ldpc.frame_errors / n if n else 0.0
    good = n - st.ldpc.frame_errors
    st.ldpc.average_iterations_correct = st.ldpc.correct_iterations / good if good else float("nan")
    return st


def ebn0_grid(lo, hi, step):
    """cli/ber.rs:106-109: min + i*step for i < floor((max-min)/step)+1, stored as f32"""
    num = int(np.floor((hi - lo) / step)) + 1
    return [float(np.float32(lo + i * step)) for i in range(max(num, 0))]


def point_seed(seed: int, ebn0_db: float) -> int:
    """Noise seed of one Eb/N0 point: the run's seed mixed (splitmix64) with the point's Eb/N0 value.
    The generator's noise is a pure function of (seed, frame index, position), so with ONE seed for the
    whole sweep every point would see the same unit-normal realisations, only rescaled -- a fully
    correlated curve.  The reference draws fresh randomness per point (thread_rng,
    /root/reference/src/simulation/ber.rs:419).  Keyed by the value, not the index, so a point is
    reproducible on its own."""
    import struct
    z = (int(seed) ^ (struct.unpack("<I", struct.pack("<f", float(ebn0_db)))[0] * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
    z = (z + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def sweep(sim: Simulator, ebn0s_db, max_iterations=100, max_frame_errors=100, min_time=0.0, max_time=float("inf"),
          max_frames=None, frames_per_batch=None, seed=0, rank=0, world=1, device=None, report=None,
          bch_max_errors=0):
    first_batch = frames_per_batch
    if not frames_per_batch:
        # eight groups of the decoder (a group: 4096 frames, more for small graphs) per simulator call: the call's
        # straggler pool is then flushed once per eight chunks (csrc/simulator.h).  A point's FIRST call is one group:
        # where most frames fail it already meets the stop rule
        first_batch = sim.get("preferred_batch")
        frames_per_batch = 8 * first_batch
    results = []
    nc = 9 if bch_max_errors > 0 else 6
    err_field = 7 if bch_max_errors > 0 else 2          # ber.rs:514-520: the BCH frame errors stop the run
    for ebn0_db in ebn0s_db:
        total = np.zeros(nc, dtype=np.int64)
        start = time.perf_counter()
        first = 0
        pseed = point_seed(seed, ebn0_db)
        while True:
            elapsed = time.perf_counter() - start
            # identical decision on every rank: the counters are the all-reduced ones and the clock
            # test is made on rank 0's view through the same all-reduce (elapsed rides along)
            stop = (total[err_field] >= max_frame_errors and elapsed >= min_time) or elapsed >= max_time
            if max_frames is not None and total[0] >= max_frames:
                stop = True
            if world > 1 or sharding.group_is_up():
                flag = sharding.reduce_counters(np.array([int(stop) if rank == 0 else 0, 0, 0, 0, 0, 0], dtype=np.int64),
                                                device)
                stop = bool(flag[0])
            if stop:
                break
            nb = (first_batch if first == 0 else frames_per_batch) * world
            if max_frames is not None:
                nb = min(nb, max_frames - int(total[0]))
            b, e = sharding.shard_range(nb, rank, world)
            if e <= b:
                part = np.zeros(nc, dtype=np.int64)
            elif bch_max_errors > 0:
                part = sim.run(ebn0_db, pseed, first + b, e - b, max_iterations, bch_max_errors)
            else:
                part = sim.run(ebn0_db, pseed, first + b, e - b, max_iterations)
            total += sharding.reduce_counters(part, device)
            first += nb
            if report and rank == 0:
                report(statistics_from_counters(ebn0_db, sim.k, total, time.perf_counter() - start), False)
        st = statistics_from_counters(ebn0_db, sim.k, total, time.perf_counter() - start)
        results.append(st)
        if report and rank == 0:
            report(st, True)
    return results


def format_details(a, sim, world: int) -> str:
    """src/cli/ber.rs:161-211 write_details: the parameter block the reference writes to the terminal
    and at the top of its result files ("Number of worker threads" reads "Number of GPUs" here)"""
    lines = ["BER TEST PARAMETERS", "-------------------", "Simulation:",
             f" - Minimum Eb/N0: {a.min_ebn0:.2f} dB", f" - Maximum Eb/N0: {a.max_ebn0:.2f} dB",
             f" - Eb/N0 step: {a.step_ebn0:.2f} dB", f" - Number of frame errors: {a.frame_errors}"]
    if a.min_time > 0.0:
        lines.append(f" - Minimum run time per Eb/N0: {format_duration(a.min_time)}")
    if a.max_time != float("inf"):
        lines.append(f" - Maximum run time per Eb/N0: {format_duration(a.max_time)}")
    if a.max_frames is not None:
        lines.append(f" - Maximum number of frames per Eb/N0: {a.max_frames}")     # this build's addition
    lines += [f" - Number of GPUs: {world}", "Channel:", f" - Modulation: {a.modulation}", "LDPC code:",
              f" - alist: {a.alist if a.alist else a.code}"]
    if a.puncturing:
        lines.append(f" - Puncturing pattern: {a.puncturing}")
    if a.interleaving:
        lines.append(f" - Interleaving columns: {a.interleaving}")
    lines += [f" - Information bits (k): {sim.k}", f" - Codeword size (N_cw): {sim.n}", f" - Frame size (N): {sim.n_tx}",
              f" - Code rate: {sim.rate:.3f}", "LDPC decoder:", f" - Implementation: {a.decoder}",
              f" - Maximum iterations: {a.max_iter}"]
    if a.bch_max_errors > 0:
        lines += ["BCH decoder:", f" - Maximum bit errors correctable: {a.bch_max_errors}"]
    return "\n".join(lines) + "\n\n"


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--code", help='standard code, e.g. "dvbs2:R1_2", "nr5g:1:384", "ar4ja:1/2:1024"')
    ap.add_argument("--alist", help="alist file (instead of --code)")
    ap.add_argument("--decoder", default="Phif64", help="decoder implementation (cli/ber.rs:49 default Phif64)")
    ap.add_argument("--puncturing", default="")
    ap.add_argument("--modulation", default="BPSK", choices=["BPSK", "8PSK"], help="cli/ber.rs:52-53")
    ap.add_argument("--interleaving", type=int, default=0,
                    help="interleaver columns, negative = read rows backwards (cli/ber.rs:55-59)")
    ap.add_argument("--codes", help='several codes as ONE job (sweep_scheduler.py): comma-separated specs, "dvbs2:normal" = the 11 '
                                    "DVB-S2 normal-frame rates.  Under torch.distributed.run whole (code, Eb/N0) points go to ranks "
                                    "from a shared queue and only the points that need many frames are frame-sharded over all "
                                    "ranks; every counter column equals the one-rank run's")
    ap.add_argument("--grid", choices=["range", "waterfall"], default="range",
                    help="with --codes: `range` = --min/--max/--step-ebn0 for every code; `waterfall` = 8 points 0.1 dB apart "
                         "around each code's waterfall, placed by a coarse pre-scan (BASELINE.json configs[4], SURVEY.md 8(d))")
    ap.add_argument("--prescan-frames", type=int, default=4096)
    ap.add_argument("--output-dir", help="with --codes: one result file per code, in the reference's file format")
    ap.add_argument("--queue", choices=["auto", "store", "static"], default="auto",
                    help="with --codes and several ranks: the shared point queue lives in the process group's key-value store "
                         "(`static`: round-robin assignment instead)")
    ap.add_argument("--defer-groups", type=int, default=32,
                    help="with --codes and several ranks: a point that still needs more than this many groups of frames after "
                         "its first call is shared by all ranks (default 32: about 2.6 s of one GPU on a DVB-S2 normal frame)")
    ap.add_argument("--verbose", action="store_true", help="with --codes: a line per finished point")
    ap.add_argument("--min-ebn0", type=float)
    ap.add_argument("--max-ebn0", type=float)
    ap.add_argument("--step-ebn0", type=float)
    ap.add_argument("--max-iter", type=int, default=100)
    ap.add_argument("--frame-errors", type=int, default=100)
    ap.add_argument("--min-time", type=float, default=0.0, help="seconds")
    ap.add_argument("--max-time", type=float, default=float("inf"), help="seconds")
    ap.add_argument("--max-frames", type=int, default=None)
    ap.add_argument("--frames-per-batch", type=int, default=0,
                    help="per GPU; 0 = eight groups of the decoder (a group: 4096 frames, more for small graphs)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--pool-size", type=int, default=64,
                    help="distinct random messages encoded on the host and cycled over the frames (this build's "
                         "generator; the reference encodes a fresh message per frame)")
    ap.add_argument("--output-file")
    ap.add_argument("--bch-max-errors", type=int, default=0,
                    help="outer BCH code: frames with at most this many bit errors count as corrected (cli/ber.rs:83)")
    ap.add_argument("--output-file-ldpc", help="LDPC-only results when --bch-max-errors is used (cli/ber.rs:102-105)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal of the multi-GPU sweep on a one-GPU box: every rank uses GPU 0 and the counters are "
                         "summed over gloo (RCCL refuses two ranks on one device); same shards, same stop rule, same table")
    a = ap.parse_args(argv)
    if not (a.codes and a.grid == "waterfall") and (a.min_ebn0 is None or a.max_ebn0 is None or a.step_ebn0 is None):
        ap.error("--min-ebn0, --max-ebn0 and --step-ebn0 are required (except with --codes ... --grid waterfall)")
    if not a.codes and not a.code and not a.alist:
        ap.error("one of --code, --alist, --codes is required")

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = None
    # under a launcher (`python -m torch.distributed.run ...` sets RANK / WORLD_SIZE / MASTER_*) the ranks form an RCCL
    # process group -- also when there is only one of them, so that a one-GPU box runs the same code path
    distributed = world > 1 or "RANK" in os.environ
    if distributed:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if a.share_device:
            local = 0                      # all ranks on GPU 0; the counters travel as host tensors over gloo
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local)
            device = torch.device("cuda", local)
            dist.init_process_group(backend="nccl", device_id=device)
    if a.codes:
        from . import sweep_scheduler
        res = sweep_scheduler.main_multi(a, rank, local, world, device, distributed)
        if distributed:
            import torch.distributed as dist
            if rank == 0:
                print(f"process group: {dist.get_backend()} with {dist.get_world_size()} rank(s)", flush=True)
            dist.barrier()
            dist.destroy_process_group()
        return res
    alist = open(a.alist).read() if a.alist else _capi.code_alist(a.code)
    sim = Simulator(alist, a.decoder, a.puncturing, device=local, pool_size=a.pool_size, pool_seed=a.seed + 1,
                    modulation=a.modulation, interleaving=a.interleaving)
    out = open(a.output_file, "w") if (a.output_file and rank == 0) else None
    out_ldpc = open(a.output_file_ldpc, "w") if (a.output_file_ldpc and a.bch_max_errors > 0 and rank == 0) else None
    details = format_details(a, sim, world)
    if out:
        out.write(details)                                        # cli/ber.rs:134-141
        if a.bch_max_errors > 0:
            out.write("\nLDPC+BCH results\n\n")
        out.write(format_header() + "\n")
    if out_ldpc:
        out_ldpc.write(details + "\nLDPC-only results\n\n" + format_header() + "\n")  # cli/ber.rs:142-147

    def report(st, final):
        if final:
            print(format_progress(st), flush=True)
            if out:
                out.write(format_progress(st) + "\n")
                out.flush()
            if out_ldpc:
                out_ldpc.write(format_progress(st, force_ldpc=True) + "\n")
                out_ldpc.flush()

    if rank == 0:
        print(details, end="")
        print(format_header(), flush=True)
    res = sweep(sim, ebn0_grid(a.min_ebn0, a.max_ebn0, a.step_ebn0), a.max_iter, a.frame_errors, a.min_time,
                a.max_time, a.max_frames, a.frames_per_batch, a.seed, rank, world, device, report,
                bch_max_errors=a.bch_max_errors)
    if distributed:
        import torch.distributed as dist
        if rank == 0:
            print(f"process group: {dist.get_backend()} with {dist.get_world_size()} rank(s)", flush=True)
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
