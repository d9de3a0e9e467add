"""A BER sweep over SEVERAL codes as ONE N-GPU job (BASELINE.json configs[4]: every DVB-S2 normal-frame rate x 8 Eb/N0
points on 8 GPUs).  The reference runs one (code, Eb/N0) point at a time and spreads its FRAMES over worker threads
(/root/reference/src/simulation/ber.rs:304-342: one decoder per worker, results folded on the main thread, stop rule
:522-531).  On GPUs that split alone wastes most of a sweep: 70 of config 5's 88 points end within a few calls (their
batches cannot grow with the rank count without decoding frames the stop rule would never have asked for), and only the
points that run to the frame cap have frames to share.  SURVEY.md section 8(e) names both splits; this module uses each
where it pays:

  phase 0  (--grid waterfall only) the per-code pre-scans that place the grid: whole CODES to ranks, from a shared queue;
  phase A  whole POINTS to ranks, from a shared queue.  A point's first call (one group of the decoder) decides: if the
           frames it still needs at the error rate just seen are few, the rank finishes the point alone; otherwise the
           point is DEFERRED with the counters of that first call;
  phase B  the deferred (cap-bound) points, one after the other, every batch's frame indices sharded contiguously over
           all ranks, the counters summed by an all-reduce between batches, the stop rule decided on the sums.

No collective on the data path; the queue is a counter in the process group's key-value store (`add` is atomic), results
travel as small JSON strings through the same store (or, where the store cannot be reached, a static round-robin
assignment and one all_gather_object).

**The table does not depend on the number of ranks.**  A point's batches are a pure function of its own counters so far
(`batch_frames`): one group first, then 8 groups per batch, 64 where the error rate seen so far says the point needs more
than 64 groups.  Batch b covers the same frame indices whoever decodes it -- one rank alone or N ranks a shard each --
the generator's noise is a pure function of (seed, frame index, position), and the stop rule is evaluated between batches
on the point's total counters.  So every counter column of every row equals the one-rank run's (tests:
tests/test_distributed_gloo.py with two gloo ranks, tests/test_gpu_launcher.py with eight real ranks on one GPU).
The time-based stop rules (--min-time / --max-time) are honoured (phase A: the owner's clock, phase B: rank 0's) but make
a table depend on the machine, as they do in the reference.

  python -m ldpc_toolbox_amd.ber --codes dvbs2:normal --grid waterfall --decoder Minsumf32 --max-iter 50 \\
         --frame-errors 100 --max-frames 1048576 --output-dir out/
  python -m torch.distributed.run --nproc-per-node 8 -m ldpc_toolbox_amd.ber --codes dvbs2:normal --grid waterfall ...
"""
import json
import time

import numpy as np

from . import sharding

DVBS2_NORMAL = ("R1_4", "R1_3", "R2_5", "R1_2", "R3_5", "R2_3", "R3_4", "R4_5", "R5_6", "R8_9", "R9_10")
FIRST_GROUPS, STEP_GROUPS, LONG_GROUPS = 1, 8, 64
DEFER_GROUPS = 32        # a point that still needs more than this many groups after its first call is shared by all ranks
PRESCAN_SEED, PRESCAN_STEPS = 99, 40


def expand_codes(text):
    """--codes: comma-separated code specs; "dvbs2:normal" = the 11 normal-frame rates (src/codes/dvbs2.rs:22-43)"""
    out = []
    for tok in text.split(","):
        tok = tok.strip()
        if tok == "dvbs2:normal":
            out += ["dvbs2:" + r for r in DVBS2_NORMAL]
        elif tok:
            out.append(tok)
    return out


# ---- the part of a point that must not depend on who runs it ------------------------------------------------------------

def frames_needed(total, max_frame_errors, err_field):
    """frames the point still needs at the frame error rate seen so far (inf while it has seen no error)"""
    frames, errs = int(total[0]), int(total[err_field])
    if errs >= max_frame_errors:
        return 0.0
    if errs == 0:
        return float("inf")
    return (max_frame_errors - errs) * frames / errs


def batch_frames(total, group, max_frame_errors, max_frames, err_field):
    """Frames of the point's NEXT batch: a pure function of its counters so far."""
    if int(total[0]) == 0:
        nb = FIRST_GROUPS * group
    elif frames_needed(total, max_frame_errors, err_field) > LONG_GROUPS * group:
        nb = LONG_GROUPS * group
    else:
        nb = STEP_GROUPS * group
    if max_frames is not None:
        nb = min(nb, max(int(max_frames) - int(total[0]), 0))
    return nb


def counters_stop(total, max_frame_errors, max_frames, err_field):
    """the machine-independent part of ber.rs:522-531 (+ this build's --max-frames)"""
    if max_frames is not None and int(total[0]) >= max_frames:
        return True
    return int(total[err_field]) >= max_frame_errors


class Point:
    """one (code, Eb/N0) job and everything that is known about it so far"""

    def __init__(self, index, code_index, ebn0_db, nc):
        self.index, self.code_index, self.ebn0_db = index, code_index, float(ebn0_db)
        self.total = np.zeros(nc, dtype=np.int64)
        self.elapsed = 0.0          # wall time spent on the point (phase A: its owner's, phase B: rank 0's)
        self.done = False
        self.owner = -1             # rank that ran its phase-A part
        self.deferred = False

    def to_json(self):
        return json.dumps({"i": self.index, "c": self.code_index, "e": self.ebn0_db, "t": [int(x) for x in self.total],
                           "s": self.elapsed, "d": self.done, "o": self.owner, "f": self.deferred})

    @staticmethod
    def from_json(text):
        d = json.loads(text)
        p = Point(d["i"], d["c"], d["e"], len(d["t"]))
        p.total = np.array(d["t"], dtype=np.int64)
        p.elapsed, p.done, p.owner, p.deferred = d["s"], d["d"], d["o"], d["f"]
        return p


# ---- the shared queue and the result board ---------------------------------------------------------------------------------

_SERIAL = [0]


def _default_store():
    """the key-value store of the default process group (the rendezvous store every rank already holds a client of)"""
    try:
        import torch.distributed as dist
        from torch.distributed import distributed_c10d as c10d
        if dist.is_available() and dist.is_initialized():
            return c10d._get_default_store()
    except Exception:
        pass
    return None


class Board:
    """A work queue over items 0..n-1 plus a place to post one JSON string per item.  Three forms, one interface:
    `store` (N ranks: an atomic counter and keys in the process group's store -- items go to whichever rank asks next),
    `static` (N ranks without a reachable store: item i belongs to rank i % world, results by all_gather_object),
    local (one rank)."""

    def __init__(self, name, n_items, rank, world, mode="auto"):
        self.n, self.rank, self.world = n_items, rank, world
        _SERIAL[0] += 1                                  # every rank constructs its boards in the same order
        self.prefix = f"ldpc_sweep/{_SERIAL[0]}/{name}"
        self.store = None
        self.posted = {}
        if world > 1 and mode in ("auto", "store"):
            self.store = _default_store()
            if self.store is None and mode == "store":
                raise RuntimeError("no process-group store to build the shared queue on")
            if self.store is not None:
                try:      # a rank that has run out of items waits in collect() for the others' results: never time that out
                    import datetime
                    self.store.set_timeout(datetime.timedelta(hours=24))
                except Exception:
                    pass
        self.mode = "store" if self.store is not None else ("static" if world > 1 else "local")
        self._cursor = rank if self.mode == "static" else 0

    def next_item(self):
        """-> the next item nobody has taken, or None"""
        if self.mode == "store":
            i = int(self.store.add(self.prefix + "/next", 1)) - 1
        else:
            i = self._cursor
            self._cursor += self.world if self.mode == "static" else 1
        return i if i < self.n else None

    def post(self, i, text):
        self.posted[i] = text
        if self.mode == "store":
            self.store.set(f"{self.prefix}/item/{i}", text)

    def collect(self):
        """-> {item: text} of ALL items, on every rank; returns once every item has been posted"""
        if self.mode == "store":
            out = {}
            for i in range(self.n):
                out[i] = self.posted[i] if i in self.posted else self.store.get(f"{self.prefix}/item/{i}").decode()
            return out
        if self.mode == "static":
            import torch.distributed as dist
            parts = [None] * self.world
            dist.all_gather_object(parts, self.posted)
            out = {}
            for p in parts:
                out.update(p)
            return out
        return dict(self.posted)


# ---- the scheduler ------------------------------------------------------------------------------------------------------------

class SweepJob:
    """make_sim(code) -> an object with run(ebn0_db, seed, first_frame, frames, max_iterations[, bch_max_errors]) -> counters,
    get("preferred_batch"), .k and .rate (ldpc_toolbox_amd.Simulator; the CPU tests pass a stand-in)."""

    def __init__(self, codes, make_sim, rank=0, world=1, device=None, max_iterations=100, max_frame_errors=100,
                 max_frames=None, min_time=0.0, max_time=float("inf"), seed=0, bch_max_errors=0, queue="auto",
                 defer_groups=DEFER_GROUPS, point_seed=None, log=None, keep_sims=2):
        from .ber import point_seed as default_point_seed
        self.codes, self.make_sim = list(codes), make_sim
        self.rank, self.world, self.device = rank, world, device
        self.max_iterations, self.max_frame_errors, self.max_frames = max_iterations, max_frame_errors, max_frames
        self.min_time, self.max_time = min_time, max_time
        self.seed, self.bch = seed, bch_max_errors
        self.nc = 9 if bch_max_errors > 0 else 6
        self.err_field = 7 if bch_max_errors > 0 else 2        # ber.rs:514-520: the BCH frame errors stop the run
        self.queue, self.defer_groups = queue, defer_groups
        self.point_seed = point_seed or default_point_seed
        self.log = log or (lambda *_: None)
        self._sims, self._keep = {}, keep_sims
        self.timeline = {}      # phase -> seconds on this rank

    # -- simulators: built on demand, a few kept (a DVB-S2 decoder's workspace is gigabytes) --
    def sim(self, code_index):
        if code_index in self._sims:
            s = self._sims.pop(code_index)
            self._sims[code_index] = s
            return s
        while len(self._sims) >= self._keep:
            old = next(iter(self._sims))
            s = self._sims.pop(old)
            if hasattr(s, "close"):
                s.close()
        s = self.make_sim(self.codes[code_index])
        self._sims[code_index] = s
        return s

    def close(self):
        for s in self._sims.values():
            if hasattr(s, "close"):
                s.close()
        self._sims = {}

    def _run(self, sim, ebn0_db, seed, first, frames):
        if frames <= 0:
            return np.zeros(self.nc, dtype=np.int64)
        if self.bch > 0:
            return np.asarray(sim.run(ebn0_db, seed, first, frames, self.max_iterations, self.bch), dtype=np.int64)
        return np.asarray(sim.run(ebn0_db, seed, first, frames, self.max_iterations), dtype=np.int64)

    # -- phase 0: where each code's waterfall is --
    def prescan(self, frames=4096):
        """First 0.1 dB step above the BPSK Shannon limit (+0.3 dB) at which fewer than half of `frames` frames fail, per
        code ("8 Eb/N0 points per code spaced 0.1 dB around each code's waterfall (chosen by a
        coarse pre-scan)").  Deterministic per code; whole codes go to ranks.  -> [crossing per code] on every rank."""
        t0 = time.perf_counter()
        board = Board("prescan", len(self.codes), self.rank, self.world, self.queue)
        while True:
            ci = board.next_item()
            if ci is None:
                break
            s = self.sim(ci)
            r = s.rate
            shannon = 10 * np.log10((2 ** (2 * r) - 1) / (2 * r))
            cross = None
            for i in range(PRESCAN_STEPS):
                e = round(float(shannon) + 0.3 + 0.1 * i, 1)
                c = self._run(s, e, self.point_seed(PRESCAN_SEED, e), 0, frames)
                if int(c[2]) * 2 < int(c[0]):
                    cross = e
                    break
            board.post(ci, json.dumps({"cross": cross, "shannon": float(shannon)}))
        res = board.collect()
        self.timeline["prescan_s"] = time.perf_counter() - t0
        out = []
        for ci in range(len(self.codes)):
            d = json.loads(res[ci])
            if d["cross"] is None:
                raise RuntimeError(f"no waterfall found for {self.codes[ci]}")
            out.append(d)
        return out

    # -- one point, alone (phase A) --
    def run_point_alone(self, p, allow_defer):
        s = self.sim(p.code_index)
        group = int(s.get("preferred_batch"))
        pseed = self.point_seed(self.seed, p.ebn0_db)
        start = time.perf_counter()
        first = int(p.total[0])
        while True:
            elapsed = p.elapsed + (time.perf_counter() - start)
            stop = (counters_stop(p.total, self.max_frame_errors, None, self.err_field) and elapsed >= self.min_time) \
                or elapsed >= self.max_time or (self.max_frames is not None and int(p.total[0]) >= self.max_frames)
            if stop:
                p.done = True
                break
            if allow_defer and first > 0 and \
                    min(frames_needed(p.total, self.max_frame_errors, self.err_field),
                        (self.max_frames - first) if self.max_frames is not None else float("inf")) > self.defer_groups * group:
                p.deferred = True
                break
            nb = batch_frames(p.total, group, self.max_frame_errors, self.max_frames, self.err_field)
            p.total += self._run(s, p.ebn0_db, pseed, first, nb)
            first += nb
        p.elapsed += time.perf_counter() - start
        p.owner = self.rank

    # -- one point, every batch shared by all ranks (phase B) --
    def run_point_shared(self, p):
        s = self.sim(p.code_index)
        group = int(s.get("preferred_batch"))
        pseed = self.point_seed(self.seed, p.ebn0_db)
        start = time.perf_counter()
        first = int(p.total[0])
        while True:
            elapsed = p.elapsed + (time.perf_counter() - start)
            stop = (counters_stop(p.total, self.max_frame_errors, None, self.err_field) and elapsed >= self.min_time) \
                or elapsed >= self.max_time or (self.max_frames is not None and int(p.total[0]) >= self.max_frames)
            if self.world > 1 and (self.min_time > 0.0 or self.max_time != float("inf")):
                # the clock tests are rank 0's: its decision travels through the same reduction as the counters
                flag = sharding.reduce_counters(np.array([int(stop) if self.rank == 0 else 0], dtype=np.int64), self.device)
                stop = bool(flag[0])
            if stop:
                p.done = True
                break
            nb = batch_frames(p.total, group, self.max_frame_errors, self.max_frames, self.err_field)
            b, e = sharding.shard_range(nb, self.rank, self.world)
            part = self._run(s, p.ebn0_db, pseed, first + b, e - b)
            p.total += sharding.reduce_counters(part, self.device) if self.world > 1 else part
            first += nb
        p.elapsed += time.perf_counter() - start

    def run(self, grids):
        """grids[code index] = the Eb/N0 values of that code.  -> [Point] in (code, Eb/N0) order, complete on every rank."""
        points = []
        for ci, grid in enumerate(grids):
            for e in grid:
                points.append(Point(len(points), ci, e, self.nc))
        # phase A: whole points from the shared queue
        t0 = time.perf_counter()
        board = Board("points", len(points), self.rank, self.world, self.queue)
        mine = 0
        while True:
            i = board.next_item()
            if i is None:
                break
            p = points[i]
            self.run_point_alone(p, allow_defer=self.world > 1)
            board.post(i, p.to_json())
            mine += 1
            self.log(f"rank {self.rank}: {self.codes[p.code_index]} @ {p.ebn0_db:.2f} dB: {int(p.total[0])} frames, "
                     f"{int(p.total[self.err_field])} frame errors in {p.elapsed:.1f} s" + (" -> shared" if p.deferred else ""))
        t1 = time.perf_counter()
        res = board.collect()
        points = [Point.from_json(res[i]) for i in range(len(points))]
        t2 = time.perf_counter()
        # phase B: the deferred points, in order, every rank a shard of every batch
        deferred = [p for p in points if p.deferred and not p.done]
        for p in deferred:
            self.run_point_shared(p)
            self.log(f"rank {self.rank}: shared {self.codes[p.code_index]} @ {p.ebn0_db:.2f} dB: {int(p.total[0])} frames") \
                if self.rank == 0 else None
        t3 = time.perf_counter()
        self.timeline.update({"phase_a_s": t1 - t0, "phase_a_wait_s": t2 - t1, "phase_b_s": t3 - t2, "points_run_alone_here": mine,
                              "points_shared": len(deferred), "queue": board.mode})
        return points


# ---- command line (reached through `python -m ldpc_toolbox_amd.ber --codes ...`) --------------------------------------------

def waterfall_grid(cross):
    """8 points 0.1 dB apart: three below the crossing of FER = 0.5, the crossing, four above"""
    from .ber import ebn0_grid
    return ebn0_grid(cross - 0.3, cross + 0.4 + 1e-6, 0.1)


def main_multi(a, rank, local, world, device, distributed):
    """the --codes form of ldpc_toolbox_amd.ber.main (argument namespace `a` from its parser)"""
    import os

    from . import _capi
    from .ber import ebn0_grid, format_details, statistics_from_counters
    from .decoder import Simulator
    from .simulation import format_header, format_progress
    codes = expand_codes(a.codes)
    if not codes:
        raise SystemExit("--codes names no code")
    t_job = time.perf_counter()

    def make_sim(code):
        return Simulator(_capi.code_alist(code), a.decoder, a.puncturing, device=local, pool_size=a.pool_size,
                         pool_seed=a.seed + 1, modulation=a.modulation, interleaving=a.interleaving)

    job = SweepJob(codes, make_sim, rank, world, device, a.max_iter, a.frame_errors, a.max_frames, a.min_time, a.max_time,
                   a.seed, a.bch_max_errors, queue=a.queue, defer_groups=a.defer_groups,
                   # (a simulator of a DVB-S2 normal-frame code holds 5-10 GB of workspace: a rank keeps the one it works on and
                   # its predecessor -- the queue hands the points out in code order -- and only one where the ranks share a GPU)
                   keep_sims=1 if getattr(a, "share_device", False) else 2,
                   log=(lambda m: print("# " + m, flush=True)) if a.verbose else None)
    if a.grid == "waterfall":
        scans = job.prescan(a.prescan_frames)
        grids = [waterfall_grid(d["cross"]) for d in scans]
    else:
        scans = None
        grids = [ebn0_grid(a.min_ebn0, a.max_ebn0, a.step_ebn0) for _ in codes]
    points = job.run(grids)
    wall = time.perf_counter() - t_job
    timeline = dict(job.timeline, rank=rank, wall_s=wall)
    if distributed:
        import torch.distributed as dist
        all_tl = [None] * world
        dist.all_gather_object(all_tl, timeline)
    else:
        all_tl = [timeline]
    if rank == 0:
        if a.output_dir:
            os.makedirs(a.output_dir, exist_ok=True)
        frames_all = 0
        import copy
        from types import SimpleNamespace
        from .simulation import parse_puncturing_pattern
        pattern = parse_puncturing_pattern(a.puncturing) if a.puncturing else None
        for ci, code in enumerate(codes):
            # (k, n, frame size and rate from the alist's first line: no simulator is built just to print a header)
            n, m = (int(x) for x in _capi.code_alist(code).split("\n", 1)[0].split()[:2])
            n_tx = n if not pattern else n // len(pattern) * sum(pattern)
            s = SimpleNamespace(k=n - m, n=n, n_tx=n_tx, rate=(n - m) / n_tx)
            aa = copy.copy(a)
            aa.code, aa.alist = code, None
            aa.min_ebn0, aa.max_ebn0 = grids[ci][0], grids[ci][-1]
            if a.grid == "waterfall":
                aa.step_ebn0 = 0.1
            details = format_details(aa, s, world)
            rows = []
            for p in points:
                if p.code_index == ci:
                    st = statistics_from_counters(p.ebn0_db, s.k, p.total, p.elapsed)
                    rows.append(format_progress(st))
                    frames_all += st.num_frames
            head = f"# {code}: n={s.n} k={s.k} rate {s.rate:.4f}"
            if scans:
                head += f" (BPSK Shannon limit {scans[ci]['shannon']:.2f} dB, pre-scan crossing of FER = 0.5 at {scans[ci]['cross']:.1f} dB)"
            print(head)
            print(format_header())
            print("\n".join(rows), flush=True)
            if a.output_dir:
                with open(os.path.join(a.output_dir, code.replace(":", "_").replace("/", "-") + ".txt"), "w") as f:
                    f.write(details + format_header() + "\n" + "\n".join(rows) + "\n")
        shared = sum(1 for p in points if p.deferred)
        print(f"# {len(points)} points ({shared} shared by all ranks, {len(points) - shared} run by one rank each), "
              f"{frames_all} frames in {wall:.1f} s wall on {world} rank(s) (incl. graph setup and encoder construction)")
        for tl in all_tl:
            print("# rank {rank}: pre-scan {p0:.1f} s, phase A {pa:.1f} s ({n} points) + {pw:.1f} s waiting, phase B {pb:.1f} s; queue: {q}".format(
                rank=tl["rank"], p0=tl.get("prescan_s", 0.0), pa=tl["phase_a_s"], n=tl["points_run_alone_here"],
                pw=tl["phase_a_wait_s"], pb=tl["phase_b_s"], q=tl["queue"]), flush=True)
    job.close()
    return points
