#!/usr/bin/env python3
"""Headline benchmark: codewords/s of the DVB-S2 n=64800 rate-1/2 flooding min-sum f32 decoder
at 50 iterations on MI355X (BASELINE.json `metric`, configs[1]; configs[3] when --gpus 8).

A step = one pass of the hot path over one batch: 4096 synthetic AWGN frames per GPU, already
resident in HBM, decoded for exactly 50 iterations (Eb/N0 = 0 dB: far below threshold, no
frame converges -- the fixed-work operating point P1 of SURVEY.md section 8(d); the run
asserts that every frame used all iterations).  Prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 and no torchrun environment: this process starts the N ranks itself (a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...`, one rank per GPU) before it has
touched a GPU, relays rank 0's JSON line and exits with the child's code.  Under torchrun
(RANK/WORLD_SIZE set, as the driver launches it) the process is one of the ranks; --gpus must then
equal WORLD_SIZE.  The multi-GPU run is N independent workers (reference:
/root/reference/src/simulation/ber.rs:304-342, one decoder per worker thread, results folded on the
main thread): no collective on the decode path, one all-reduce of the six error counters (48 bytes)
and one of the elapsed time after the timed region.

Secondary blocks on rank 0 at N = 1 (never `value`): `realistic` (Eb/N0 = 2 dB, early termination),
`config3` (BASELINE.json configs[2]: 5G NR BG1 Zc=384, HLTanhf32, batch 8192) and `cpu_baseline`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SPEC = "dvbs2:R1_2"
IMPL = "Minsumf32"
MAX_ITER = 50
BATCH_PER_GPU = 4096
EBN0_FIXED_WORK_DB = 0.0
HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBPS = 6290.0   # same guide: 6.29 TB/s measured (float4 copy)
# vector-ALU issue peak of the chip (MI355X_MICROARCH.md: 4 SIMD-32 per CU, a wave64 instruction issues over 2 cycles):
# 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles
VALU_PEAK_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 2
# what THIS rule's instruction mix reaches when nothing else runs: its two functions back to back on the whole chip,
# 344 G evaluations/s x 172 vector instructions each / 64 lanes (tools/mb/pk_bench.hip, profiles/r03_packed_f32.txt):
# a software ceiling measured here, reported beside the hardware peak, never as `peak`
VALU_MIX_CEILING_WAVE_INSTS_PER_S = 344e9 * 172 / 64
C3_SPEC, C3_IMPL, C3_BATCH, C3_POOL, C3_EBN0_DB = "nr5g:1:384", "HLTanhf32", 8192, 64, -2.0
FRAME_POOL = 64       # distinct encoded messages behind a batch of synthetic frames (independent noise per frame)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="codewords per GPU per step (default: the workload's, 4096 for config2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=0,
                    help="execution lanes of the decoder (0 = the library's choice; tools/profile_r03.sh passes 1 so that "
                         "every traced launch is a whole-group launch, as in the bracketed region)")
    ap.add_argument("--no-realistic", action="store_true", help="skip the secondary Eb/N0 = 2 dB point")
    ap.add_argument("--no-config3", action="store_true", help="skip the secondary BASELINE configs[2] block")
    ap.add_argument("--live-traffic", action="store_true",
                    help="also re-measure config 3's traffic and instruction counters in this run (three more rocprofv3 --pmc "
                         "child passes: minutes)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="quote profiles/hbm_traffic.json for the headline kernel's roofline.traffic instead of measuring it in "
                         "this run (default: two rocprofv3 --pmc child passes of one step each, about a minute)")
    ap.add_argument("--share-device", action="store_true",
                    help="REHEARSAL of the N-rank path on a one-GPU box (never a throughput number, the line says so): the N "
                         "ranks are N real decoder processes that all use GPU 0, each decodes batch/N frames of the one frame "
                         "stream, counters and times are reduced over gloo (RCCL refuses two ranks on one device).  Exercises "
                         "the launcher, the environment, device mapping, CPU pinning and N x (enqueue threads + progress "
                         "pollers) at once")
    ap.add_argument("--workload", choices=("config2", "config3"), default="config2",
                    help="config2 (default, the metric's configuration) or config3 (5G NR BG1 Zc=384 HLTanhf32: the launch-bound "
                         "layered schedule, 35 launches per iteration and lane) for --share-device rehearsals")
    ap.add_argument("--no-affinity", action="store_true", help="do not pin a rank's threads to its GPU's NUMA node share")
    ap.add_argument("--stub", action="store_true",
                    help="launcher self-test (tests/test_distributed_gloo.py): CPU ranks over gloo and a stand-in "
                         "for the decoder; the line it prints says so and is not a measurement")
    return ap.parse_args(argv)


# ---- N-rank launch ------------------------------------------------------------------------------

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    """Parent of an N-GPU run.  Nothing here touches a GPU (no HIP call, no torch.cuda use beyond
    counting devices), and the ranks are a child process, never an exec of this one."""
    if not args.stub:
        import torch
        have = torch.cuda.device_count()      # does not initialise the GPU
        if have < (1 if args.share_device else args.gpus):
            print(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--batch", str(args.batch), "--lanes", str(args.lanes)]
    for flag, on in (("--no-cpu-baseline", args.no_cpu_baseline), ("--no-realistic", args.no_realistic),
                     ("--no-config3", args.no_config3), ("--live-traffic", args.live_traffic),
                     ("--no-live-traffic", args.no_live_traffic), ("--stub", args.stub),
                     ("--share-device", args.share_device), ("--no-affinity", args.no_affinity)):
        if on:
            cmd.append(flag)
    cmd += ["--workload", args.workload]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    env["LDPC_BENCH_CHILD"] = "1"
    return subprocess.run(cmd, env=env).returncode


# ---- the decoder stand-in of --stub ----------------------------------------------------------------

class _StubDecoder:
    """Shape of LdpcDecoder as far as a step needs it; decodes nothing (every frame 'fails')."""
    n, k, edges = 64, 32, 192

    def step(self, its):
        time.sleep(0.02)
        its.fill_(-1)


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            out += list(range(int(a), int(b or a) + 1))
    return out


def own_numa_node(device_index):
    """NUMA node of THIS rank's GPU, from the one device the rank has selected (no other GPU is queried or initialised)"""
    import torch
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        return int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
    except Exception:
        return -1


def pin_rank_to_cpus(local_rank, nodes):
    """Per-rank CPU affinity: a rank's threads (the enqueuing lane threads, the progress pollers, the staging threads) stay on
    the NUMA node its GPU hangs off, and the ranks of one node share its CPUs evenly (the launch-bound layered schedule
    enqueues 35 launches per iteration and lane: eight unpinned ranks migrate across sockets).
    nodes[r] = NUMA node of local rank r's GPU, each found by its own rank (own_numa_node) and exchanged over the process
    group.  -> description for the JSON line."""
    try:
        mine = nodes[local_rank]
        allowed = sorted(os.sched_getaffinity(0))
        cpus = allowed
        if mine >= 0:
            try:
                node_cpus = set(_cpulist(open(f"/sys/devices/system/node/node{mine}/cpulist").read()))
                cpus = [c for c in allowed if c in node_cpus] or allowed
            except OSError:
                pass
        peers = [r for r in range(len(nodes)) if nodes[r] == mine]
        share = max(len(cpus) // len(peers), 1)
        i = peers.index(local_rank)
        chunk = cpus[i * share:(i + 1) * share] or cpus
        os.sched_setaffinity(0, chunk)
        return {"numa_node": mine, "cpus": len(chunk), "first_cpu": chunk[0], "last_cpu": chunk[-1], "ranks_on_node": len(peers)}
    except Exception as e:      # pinning is an optimisation: never fail a run over it
        return {"error": str(e)}


WORKLOADS = {"config2": ("dvbs2:R1_2", "Minsumf32", 0.0, 4096), "config3": ("nr5g:1:384", "HLTanhf32", -2.0, 8192)}


def main(argv=None):
    args = parse_args(argv)
    global SPEC, IMPL, EBN0_FIXED_WORK_DB
    SPEC, IMPL, EBN0_FIXED_WORK_DB, default_batch = WORKLOADS[args.workload]
    if args.batch is None:
        args.batch = default_batch
    in_rank = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    # LDPC_BENCH_FORCE_LAUNCH=1: go through the N-rank launcher (child torch.distributed.run, RCCL init, both
    # all-reduces) even for --gpus 1 -- the rehearsal of the multi-GPU path on a one-GPU box
    force_launch = os.environ.get("LDPC_BENCH_FORCE_LAUNCH") == "1"
    if not in_rank and (args.gpus > 1 or force_launch):
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    import numpy as np
    import torch

    from ldpc_toolbox_amd import sharding, simulation as sim
    # LDPC_BENCH_FORCE_DIST=1: take the RCCL path even with one rank (exercises it on a 1-GPU box)
    distributed = world > 1 or os.environ.get("LDPC_BENCH_FORCE_DIST") == "1" or (in_rank and force_launch)
    stub = args.stub
    share = args.share_device and not stub
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    device_index = 0 if share else local_rank
    affinity = None
    if stub:
        device = torch.device("cpu")
    else:
        # the rank's own GPU first, and nothing but it: no other device is queried, so no rank initialises HIP state on
        # its neighbours' GPUs
        torch.cuda.set_device(device_index)
        device = torch.device("cuda", device_index)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub or share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)
    host_side = stub or share          # where the reductions' tensors live (gloo: host, RCCL: the rank's GPU)
    if distributed and not stub and not args.no_affinity:
        # one node (the driver's launch): every rank reports its own GPU's NUMA node, the ranks of a node split its CPUs
        mine = torch.tensor([local_rank, own_numa_node(device_index)], dtype=torch.int64, device="cpu" if host_side else device)
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        nodes = [-1] * local_world
        for t in both:
            lr, nd = int(t[0].item()), int(t[1].item())
            if 0 <= lr < local_world:
                nodes[lr] = nd
        affinity = pin_rank_to_cpus(local_rank, nodes)

    def sync():
        if not stub:
            torch.cuda.synchronize(device)

    def barrier():
        sync()
        if distributed:
            dist.barrier()
        sync()

    B = args.batch // world if share else args.batch     # --share-device: the ranks split ONE batch (one GPU's memory)
    t_setup0 = time.perf_counter()
    if stub:
        dec = _StubDecoder()
        rng = np.random.Generator(np.random.Philox(key=[1000 + rank, 0]))
        msgs = rng.integers(0, 2, size=(B, dec.k), dtype=np.uint8)
        bits = torch.zeros((B, dec.k), dtype=torch.uint8)
        its = torch.zeros(B, dtype=torch.int32)

        def step():
            dec.step(its)
    else:
        import ldpc_toolbox_amd as lt
        alist = lt.code_alist(SPEC)
        dec = lt.LdpcDecoder(alist, IMPL, device=device_index)
        if args.lanes:
            dec.set("lanes", args.lanes)
        msgs, llrs = make_frames(alist, IMPL, B, EBN0_FIXED_WORK_DB, seed=1000, device=device, device_index=device_index,
                                 first_frame=rank * B)
        bits = torch.zeros((B, dec.k), dtype=torch.uint8, device=device)
        its = torch.zeros(B, dtype=torch.int32, device=device)
        # a stream of our own: the library launches on it and records its HIP events on it.  (Handing
        # over torch's default stream, handle 0, would select the handle's private stream instead --
        # include/ldpc_toolbox.h, the _device entries.)  The frames were produced on torch's default
        # stream: synchronise before the first launch.
        stream = torch.cuda.Stream(device)
        sync()

        def step():
            dec.decode_batch_device(llrs.data_ptr(), False, B, MAX_ITER, bits.data_ptr(), dec.k,
                                    its.data_ptr(), 0, stream.cuda_stream)

    sync()
    setup_s = time.perf_counter() - t_setup0      # graph tables, this rank's frames (generated in place on the device)
    for _ in range(args.warmup):
        step()
    # The timed region: K steps of the library as a caller gets it (for the f32 flooding rules: two execution lanes,
    # half-batches on two streams whose launches overlap -- one lane's variable-node launch beside the other's
    # check-node launch).  No instrumentation inside it.
    barrier()
    t0 = time.perf_counter()
    enqueue_s = 0.0                     # host time inside the calls: a call on the caller's stream returns when its launches are enqueued
    for _ in range(args.steps):
        te = time.perf_counter()
        step()
        enqueue_s += time.perf_counter() - te
    sync()
    own_elapsed = time.perf_counter() - t0   # this rank's own K steps, before it waits for the others (per_rank below)
    barrier()
    elapsed = time.perf_counter() - t0
    lanes, group_cw = (1, min(B, 4096)) if stub else (max(dec.get("last_lanes"), 1), dec.get("last_group"))
    # The kernels' own durations: the same K steps again with the library's HIP events around every check-node /
    # variable-node launch, recorded on the launch stream.  (With those brackets in place the lanes no longer overlap
    # usefully, so the library runs a profiled call as ONE lane: a launch then covers the whole group and its time is
    # the kernel's own.  Timing the bracketed region instead of the clean one would report the slower of the two.)
    cn_launches = vn_launches = 0
    cn_ms = vn_ms = 0.0
    elapsed_prof, prof_group = 0.0, min(B, 4096)
    if not stub:
        dec.set("profiling", 1)
        dec.kernel_stats(0, reset=True)
        dec.kernel_stats(1, reset=True)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed_prof = time.perf_counter() - t1
        cn_launches, cn_ms = dec.kernel_stats(0)
        vn_launches, vn_ms = dec.kernel_stats(1)
        prof_group = dec.get("last_group")
        dec.set("profiling", 0)

    enqueue_min = enqueue_max = enqueue_s
    # every rank's own figures travel to rank 0 (a straggler shows in the line instead of hiding in the MAX)
    mine = [float(rank), own_elapsed, setup_s, enqueue_s, cn_ms / max(cn_launches, 1), vn_ms / max(vn_launches, 1),
            float(device_index), float(affinity.get("numa_node", -1)) if isinstance(affinity, dict) and "numa_node" in affinity else -1.0,
            float(affinity.get("cpus", 0)) if isinstance(affinity, dict) and "cpus" in affinity else 0.0]
    per_rank_rows = [mine]
    if distributed:
        t = torch.tensor([elapsed, setup_s, enqueue_s, -enqueue_s], dtype=torch.float64, device="cpu" if host_side else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, setup_s, enqueue_max, enqueue_min = float(t[0].item()), float(t[1].item()), float(t[2].item()), -float(t[3].item())
        row = torch.tensor(mine, dtype=torch.float64, device="cpu" if host_side else device)
        rows = [torch.zeros_like(row) for _ in range(world)]
        dist.all_gather(rows, row)
        per_rank_rows = [[float(x) for x in r.cpu().tolist()] for r in rows]
    per_rank = [{"rank": int(r[0]), "device": int(r[6]), "ms_per_step": r[1] / max(args.steps, 1) * 1e3,
                 "codewords_per_s": B * args.steps / r[1] if r[1] > 0 else None, "setup_s": r[2],
                 "host_enqueue_ms_per_step": r[3] / max(args.steps, 1) * 1e3,
                 "check_launch_us": r[4] * 1e3, "variable_phase_us": r[5] * 1e3,
                 "numa_node": int(r[7]), "pinned_cpus": int(r[8])} for r in sorted(per_rank_rows)]

    its_np = its.cpu().numpy()
    bits_np = bits.cpu().numpy()
    # P1 contract: every frame ran all iterations (none converged)
    if args.workload == "config2" or stub:
        assert (its_np == -1).all(), "fixed-work operating point violated: some frames converged"
    st = sim.fold_statistics(EBN0_FIXED_WORK_DB, dec.k, msgs, bits_np, its_np, MAX_ITER, elapsed)
    counters = sharding.reduce_counters(sharding.counters_from_statistics(st), None if host_side or not distributed else device)

    if rank != 0:
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    E, n, k = dec.edges, dec.n, dec.k
    total_cw = B * world * args.steps
    cw_per_s = total_cw / elapsed
    # ---- roofline: ONE FLOODING ITERATION (check-node launch + variable-node phase) is the unit -----------------------
    # (4E + 2N) words per codeword-iteration = 4 147 184 B.  Per-kernel attribution of those bytes
    # stopped meaning anything when work moved between the two launches (the check-node kernel rebuilds the posteriors of
    # the degree <= 2 variables, the variable-node kernel walks only the others), so the iteration's launches are timed and
    # counted together; the first of the MAX_ITER check-node launches reads no messages (E words less).
    bytes_cw_iter = (4 * E + 2 * n) * 4
    alg_bytes_iter = ((4 * E + 2 * n) * MAX_ITER - E) * 4.0 / MAX_ITER
    cn_avg_s = cn_ms / max(cn_launches, 1) * 1e-3
    vn_avg_s = vn_ms / max(vn_launches, 1) * 1e-3
    iter_s = cn_avg_s + vn_avg_s
    group = prof_group                             # codewords per launch in the bracketed region (one lane)
    iter_gbps = alg_bytes_iter * group / iter_s / 1e9 if iter_s > 0 else 0.0
    # What the kernels really move.  Row records: a check row's d messages are the record
    # {min1, min2, flip bits, argmin} (recw words), per-edge messages exist only for the variables of degree >= 3.
    recw = 0 if stub else dec.get("row_records")
    m_rows = 0 if stub else dec.get("m")
    cn_kernel = "cn_minsum_rec_kernel" if recw else "cn_minsum_lfree_kernel"
    byte_model = None
    if not stub:
        deg = np.array(alist.split("\n")[2].split(), dtype=np.int64)   # alist line 3: the column weights
        n_free = int((deg <= 2).sum())
        e_keep = int(deg[deg > 2].sum())
        n_keep = n - n_free
        if recw:
            cn_terms = {"records read + written (2 * recw * M)": 2 * recw * m_rows,
                        "posterior of a degree>=3 variable gathered per edge (E_keep)": e_keep,
                        "per-edge messages written for those edges (E_keep)": e_keep,
                        "channel LLR of each degree<=2 variable once (N_free)": n_free}
        else:
            cn_terms = {"per-edge messages read + written, posterior per edge (3E)": 3 * E,
                        "channel LLR + posterior of each degree<=2 variable (2 N_free)": 2 * n_free}
        vn_terms = {"per-edge messages read (E_keep)": e_keep, "channel LLR read + posterior written per degree>=3 variable (2 N_keep)": 2 * n_keep}
        moved = (sum(cn_terms.values()) + sum(vn_terms.values())) * 4
        # compulsory HBM bytes: the same with every posterior row fetched once instead of once per edge (a tile's
        # posteriors, 66 MB, are gathered from the 256 MB Infinity Cache)
        compulsory = moved - (e_keep - n_keep) * 4
        byte_model = {"words_of": 4, "M": m_rows, "E": E, "N": n, "N_free(degree<=2)": n_free, "N_keep": n_keep, "E_keep": e_keep,
                      "record_words": recw, "check_node_launch_words": cn_terms, "variable_node_phase_words": vn_terms,
                      "moved_bytes_per_codeword_iteration": moved,
                      "compulsory_bytes_per_codeword_iteration": compulsory,
                      "compulsory_rule": "moved - (E_keep - N_keep) words: each posterior row charged once per iteration"}

    traffic_iter, traffic_by_kernel, traffic_source = None, None, None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not stub and world == 1 and not in_rank and not args.no_live_traffic and recw:
        traffic_by_kernel, traffic_source = live_traffic(B)
        if traffic_by_kernel:
            traffic_iter = sum(traffic_by_kernel.values())
    if traffic_iter is None and not stub and os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            cn_b, vn_b = t.get(cn_kernel + "_bytes_per_launch"), t.get("vn_kernel_bytes_per_launch")
            if cn_b and vn_b:
                traffic_by_kernel = {cn_kernel: cn_b * group / 4096.0, "vn_kernel": vn_b * group / 4096.0}
                traffic_iter = sum(traffic_by_kernel.values())
                traffic_source = ("profiles/hbm_traffic.json: separate rocprofv3 --pmc passes of this command "
                                  f"({t.get('collected', '')}), not re-measured in this run (--no-live-traffic, a multi-rank run, "
                                  "or the live passes failed)")
        except (OSError, ValueError):
            traffic_iter = None

    def frac_of(gbytes_per_s, peak):
        return gbytes_per_s / peak if gbytes_per_s is not None else None

    moved_gbps = byte_model["moved_bytes_per_codeword_iteration"] * group / iter_s / 1e9 if byte_model and iter_s > 0 else None
    comp_gbps = byte_model["compulsory_bytes_per_codeword_iteration"] * group / iter_s / 1e9 if byte_model and iter_s > 0 else None
    traffic_gbps = traffic_iter / iter_s / 1e9 if traffic_iter and iter_s > 0 else None
    out = {
        "metric": "codewords/s + info-bits/s, DVB-S2 n=64800 r=1/2 min-sum 50 iters",
        "value": cw_per_s,
        "unit": "codewords/s",
        "info_bits_per_s": cw_per_s * k,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (the library's Philox4x32-10 AWGN frame generator, seed 1000, frames [rank*B, (rank+1)*B))",
        "config": {"workload": "DVB-S2 n=64800 rate-1/2 (E=226799), flooding Minsumf32, 50 iterations, "
                               f"batch={B} codewords per GPU resident in HBM, Eb/N0=0 dB (fixed work)",
                   "code": SPEC, "implementation": IMPL, "max_iterations": MAX_ITER,
                   "batch_per_gpu": B, "parallelism": f"batch-sharded x{world}, no data-path collective"},
        "roofline": {
            "bound": "hbm",
            "kernel": f"one flooding iteration = {cn_kernel} + vn_kernel" + (" (+ vn_free_rec_kernel's one launch per decode)" if recw else ""),
            # ALGORITHMIC: SURVEY 8(d)'s bytes of one iteration of one group over the iteration's measured launch time
            "achieved": iter_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": iter_gbps / HBM_PEAK_GBPS,
            "algorithmic_bytes_per_iteration": alg_bytes_iter * group,
            "iteration_us": iter_s * 1e6,
            "launch_us": {cn_kernel: cn_avg_s * 1e6, "variable-node phase (vn_kernel)": vn_avg_s * 1e6},
            "launches": {"check": cn_launches, "variable": vn_launches}, "codewords_per_launch": group,
            # COUNTERS: what crossed the fabric per iteration (FETCH_SIZE / WRITE_SIZE, gfx950-corrected)
            "traffic": traffic_iter, "traffic_by_kernel": traffic_by_kernel, "traffic_source": traffic_source,
            "traffic_over_algorithmic": (traffic_iter / (alg_bytes_iter * group) if traffic_iter else None),
            "traffic_frac_of_peak": frac_of(traffic_gbps, HBM_PEAK_GBPS),
            # BYTE MODEL: what the kernels must move at least, against what a pure HBM stream reaches on this chip
            "moved_frac_of_peak": frac_of(moved_gbps, HBM_PEAK_GBPS),
            "compulsory_frac_of_achievable": frac_of(comp_gbps, HBM_ACHIEVABLE_GBPS),
            "achievable_GBps": HBM_ACHIEVABLE_GBPS,
            "byte_model": byte_model,
            "whole_job_frac": cw_per_s / world * alg_bytes_iter * MAX_ITER / 1e9 / HBM_PEAK_GBPS,
            "formulas": {"frac": "algorithmic_bytes_per_iteration / iteration_us / peak",
                         "traffic_frac_of_peak": "traffic / iteration_us / peak",
                         "compulsory_frac_of_achievable": "byte_model.compulsory_bytes_per_codeword_iteration * codewords_per_launch / iteration_us / achievable_GBps",
                         "whole_job_frac": "value / n_gpus * max_iterations * algorithmic bytes per codeword-iteration / peak (the driver-timed job, ingest and emit included)"},
            "measured_in": ("a second region of the same K steps with the library's HIP events around every check-node launch "
                            "and every variable-node phase (" + (f"{elapsed_prof / args.steps * 1e3:.2f}" if args.steps else "0") +
                            " ms per step, one execution lane); the timed region itself carries no instrumentation and runs "
                            + str(lanes) + " lane(s) of " + str(group_cw) + " codewords"),
            "note": "frac can exceed 1: the kernels move fewer bytes than the per-edge algorithm SURVEY 8(d) prices (a row's "
                    "messages are one record, v2c is never stored) -- that is what traffic_over_algorithmic shows; it is not "
                    "skipped work (every frame runs all iterations: asserted; cpu_baseline.matches_gpu_output).  The honest "
                    "headroom is compulsory_frac_of_achievable: the bytes the design cannot avoid over the 6.29 TB/s a pure "
                    "stream reaches.  `traffic` counts Infinity-Cache hits of the gathered posterior rows as fabric traffic"},
        "ber": {"ebn0_db": EBN0_FIXED_WORK_DB, **dict(zip(sharding.COUNTER_FIELDS, (int(x) for x in counters)))},
        "regions": {"timed": {"steps": args.steps, "ms_per_step": elapsed / args.steps * 1e3, "event_brackets": False,
                              "execution_lanes": lanes, "codewords_per_launch": group_cw},
                    "bracketed": {"steps": args.steps, "ms_per_step": elapsed_prof / args.steps * 1e3, "event_brackets": True,
                                  "execution_lanes": 1, "codewords_per_launch": prof_group,
                                  "codewords_per_s": (B * args.steps / elapsed_prof if elapsed_prof > 0 else None),
                                  "note": "rank 0's clock; source of roofline.*'s launch times"}},
        "launch": {"ranks": world, "process_group": (dist.get_backend() if distributed else None),
                   "cpu_affinity_rank0": affinity,
                   "host_enqueue_ms_per_step": {"min_over_ranks": enqueue_min / args.steps * 1e3, "max_over_ranks": enqueue_max / args.steps * 1e3,
                                                "note": "host time inside the decode calls of the timed region (they return once "
                                                        "their launches are enqueued)"},
                   "started_by": ("bench.py launch_ranks -> torch.distributed.run" if os.environ.get("LDPC_BENCH_CHILD") == "1"
                                  else ("external torchrun" if in_rank else "in-process")),
                   "setup_s_max_over_ranks": setup_s,
                   # each rank's own clock over its own K steps (to its own stream synchronisation, before the closing
                   # barrier) and its own bracketed kernel times: `value` is the MAX-over-ranks job, these show who set it
                   "per_rank": per_rank,
                   "slowest_rank": max(per_rank, key=lambda r: r["ms_per_step"])["rank"],
                   "min_rank_rate_times_ranks": min((r["codewords_per_s"] or 0.0) for r in per_rank) * world,
                   "sum_of_rank_rates": sum((r["codewords_per_s"] or 0.0) for r in per_rank)},
    }
    if share or args.workload != "config2":
        # rehearsal of the N-rank path on one GPU, or another workload through the same launcher: not the metric
        if args.workload != "config2":
            out["metric"] = f"codewords/s, {SPEC} {IMPL}, {MAX_ITER} iterations (a launcher workload, not BASELINE.json's metric)"
        out["roofline"] = None
        out["config"]["workload"] = (f"{SPEC} {IMPL}, {MAX_ITER} iterations, {B} codewords per rank, Eb/N0={EBN0_FIXED_WORK_DB} dB" +
                                     (f"; --share-device REHEARSAL: {world} decoder processes share GPU 0 (gloo) -- NOT a throughput "
                                      "measurement" if share else ""))
        out["share_device"] = bool(share)
        if share:
            out["data"] += "; rehearsal: all ranks on one GPU"
    if stub:
        out["data"] = "stub (no decoder ran: launcher self-test, not a measurement)"
        out["roofline"] = None
        out["config"]["workload"] = "stub"
    elif not share and args.workload == "config2":
        # secondary, not `value`: the realistic operating point P2 (Eb/N0 = 2 dB, syndrome early
        # termination active), one untimed-warm pass over a fresh batch on rank 0
        if not args.no_realistic:
            out["realistic"] = realistic_point(dec, alist, IMPL, B, device, local_rank, stream)
            # against the fixed-work rate of this rank scaled by the iterations actually run (profiles/r04_p2_timeline.txt
            # says where the rest goes: the detection lag, two full-size iterations with frozen lanes, the re-packing)
            per_rank = out["value"] / max(world, 1)
            out["realistic"]["fraction_of_iteration_proportional_bound"] = (
                out["realistic"]["codewords_per_s"] / (per_rank * MAX_ITER / max(out["realistic"]["average_iterations"], 1e-9)))
        if world == 1 and not args.no_config3:
            out["config3"] = config3_point(device, local_rank, with_cpu=not args.no_cpu_baseline, live=args.live_traffic)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(alist, IMPL, llrs, bits_np, its_np, k)
    print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def under_profiler():
    """true when this process already runs under rocprofv3 / a tool library (nesting profilers is not attempted)"""
    return any(k.startswith(("ROCPROFILER_", "ROCP_TOOL", "ROCPROF_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def child_env():
    """environment of a counter-pass child: this one without the rendezvous / launcher / profiler variables (a child that
    inherited RANK / WORLD_SIZE / MASTER_PORT would join the parent's process group)"""
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_", "MASTER_", "TORCHELASTIC_",
            "LDPC_BENCH_", "ROCP", "LD_PRELOAD")
    env = {k: v for k, v in os.environ.items() if not k.startswith(drop)}
    env["TMPDIR"] = "/tmp"
    return env


def live_counter_pass(child_argv, counters, timeout_s=100):
    """One rocprofv3 --pmc pass of `child_argv` (a python script and its arguments) as a CHILD process -- never an
    exec of this one, which holds the GPU -- with the program itself right after `--`.
    -> {kernel name: {counter: [values per launch]}} or None (rocprofv3 missing, the pass failed)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or under_profiler():
        return None
    work = tempfile.mkdtemp(prefix="ldpc_bench_pmc_")
    try:
        cmd = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", work, "--", sys.executable] + list(child_argv)
        r = subprocess.run(cmd, cwd="/tmp", env=child_env(), capture_output=True, text=True, timeout=timeout_s)
        if r.returncode != 0:
            return None
        out = {}
        for f in glob.glob(os.path.join(work, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                out.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        return out or None
    except Exception:
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


ITERATION_KERNELS = ("cn_minsum_rec_kernel", "vn_kernel", "vn_free_rec_kernel")


def live_traffic(batch, kernels=ITERATION_KERNELS):
    """roofline.traffic measured in THIS run: two counter passes (FETCH_SIZE, then WRITE_SIZE: the TCC block has 4
    slots, they need 3 + 2) of ONE step (50 iterations, one execution lane, the same batch) of this same script.
    -> ({kernel: HBM bytes it moves per iteration of one group}, provenance) corrected as MI355X_MICROARCH.md prescribes
    for gfx950: (2 * FETCH_SIZE + WRITE_SIZE) * 1024; or (None, None) -- the caller then quotes the committed counter
    file and says so."""
    child = [os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--batch", str(batch), "--lanes", "1",
             "--no-cpu-baseline", "--no-realistic", "--no-config3", "--no-live-traffic"]
    tot = {k: 0.0 for k in kernels}
    for counter, weight in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        res = live_counter_pass(child, [counter])
        if not res:
            return None, None
        iterations = 0      # check-node launches in the pass = iterations of one group (every step the child ran, every variant)
        for k in kernels:
            vals = [v[counter] for name, v in res.items() if name.split("<")[0].split("::")[-1].split("(")[0] == k and counter in v]
            if k == kernels[0]:
                iterations = sum(len(x) for x in vals)
                if not iterations:
                    return None, None
            tot[k] += weight * 1024.0 * sum(sum(x) for x in vals) / iterations
    return (tot, "measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of this script (one lane, the same "
                 "batch) as child processes; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction), every launch of the "
                 "kernel summed and divided by the number of check-node launches (= iterations of one group)")


def live_config3_counters():
    """config 3's level kernels measured in THIS run: FETCH_SIZE, WRITE_SIZE and SQ_INSTS_VALU passes of
    tools/perf_probe.py on the same code / rule / batch / noise.  -> (HBM bytes per level launch, vector
    wavefront-instructions per level launch), launch-weighted over the kernels of the iterations after the first
    (the `false` = not-FIRST template variants), or (None, None)."""
    child = [os.path.join(ROOT, "tools", "perf_probe.py"), "--spec", C3_SPEC, "--impl", C3_IMPL, "--batch", str(C3_BATCH),
             "--iters", "6", "--groups", str(C3_BATCH), "--reps", "1", "--sigma", "1.565"]
    tot = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
        res = live_counter_pass(child, [counter])
        if not res:
            return None, None
        s_sum = s_n = 0.0
        for name, v in res.items():
            if "hl_level" in name and name.split("(")[0].rstrip(">").rstrip().endswith("false") and counter in v:
                s_sum += sum(v[counter])
                s_n += len(v[counter])
        if not s_n:
            return None, None
        tot[counter] = s_sum / s_n
    return (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0, tot["SQ_INSTS_VALU"]


def make_frames(alist, impl, batch, ebn0_db, seed, device, device_index, first_frame=0, pool=FRAME_POOL):
    """Synthetic AWGN frames from the LIBRARY's own generator (csrc/frame_gen.hip.h through
    ldpc_toolbox_sim_generate): `pool` random messages encoded once (seeded), frame f carries pool codeword (a pure
    function of seed and f), BPSK, noise = Philox4x32-10 keyed by (seed, frame, position) + polar method, LLR in f32.
    Reproducible on any box and any torch version; tests/oracle_binding.generate_llrs regenerates the same frames on the
    CPU.  Rank r of an N-GPU run takes frames [r*batch, (r+1)*batch) of the one stream.  -> (messages [B][k], llrs in HBM)"""
    import torch

    import ldpc_toolbox_amd as lt
    gen = lt.Simulator(alist, impl, "", device=device_index, pool_size=pool, pool_seed=seed)
    llrs = torch.empty((batch, gen.n_tx), dtype=torch.float32, device=device)
    idx = gen.generate_into(llrs.data_ptr(), ebn0_db, seed, first_frame, batch)     # in place: no host round trip
    msgs, _ = gen.pool_data()
    gen.close()
    return msgs[idx], llrs


def realistic_point(dec, alist, impl, B, device, device_index, stream, ebn0_db=2.0, pool=FRAME_POOL):
    import torch

    from ldpc_toolbox_amd import simulation as sim
    msgs, llrs = make_frames(alist, impl, B, ebn0_db, seed=77, device=device, device_index=device_index, pool=pool)
    bits = torch.zeros((B, dec.k), dtype=torch.uint8, device=device)
    its = torch.zeros(B, dtype=torch.int32, device=device)
    # every call here is followed by a synchronisation, so the call may pace its launches on the groups' progress words
    # (include/ldpc_toolbox.h, "throttle": what the library's simulation driver sets for its own calls)
    dec.set("throttle", 1)
    best = None
    for _ in range(3):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        dec.decode_batch_device(llrs.data_ptr(), False, B, MAX_ITER, bits.data_ptr(), dec.k, its.data_ptr(), 0,
                                stream.cuda_stream)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    dec.set("throttle", 0)
    st = sim.fold_statistics(ebn0_db, dec.k, msgs, bits.cpu().numpy(), its.cpu().numpy(), MAX_ITER, best)
    return {"ebn0_db": ebn0_db, "codewords_per_s": B / best, "average_iterations": st.average_iterations,
            "frame_errors": st.ldpc.frame_errors, "bit_errors": st.ldpc.bit_errors, "frames": st.num_frames,
            "ber": st.ldpc.ber, "note": "early termination with device-side batch compaction: converged "
                                        "codewords retire at checkpoints, live ones are packed into fewer tiles; "
                                        "best of three synchronised calls with option throttle = 1"}


def config3_point(device, device_index, with_cpu, steps=5, live=True):
    """BASELINE.json configs[2]: 5G NR base graph 1, Zc = 384 (n = 26112, k = 8448, E = 121344),
    horizontal-layered sum-product (HLTanhf32), 8192 codewords resident in HBM, 50 iterations at
    Eb/N0 = -2 dB (fixed work: asserted that no frame converges).  Roofline: the layered algorithmic
    bytes (4E + N) * 4 per codeword-iteration."""
    import torch

    import ldpc_toolbox_amd as lt
    alist = lt.code_alist(C3_SPEC)
    dec = lt.LdpcDecoder(alist, C3_IMPL, device=device_index)
    B = C3_BATCH
    msgs, llrs = make_frames(alist, C3_IMPL, B, C3_EBN0_DB, seed=31, device=device, device_index=device_index, pool=C3_POOL)
    bits = torch.zeros((B, dec.k), dtype=torch.uint8, device=device)
    its = torch.zeros(B, dtype=torch.int32, device=device)
    stream = torch.cuda.Stream(device)
    torch.cuda.synchronize(device)

    def run():
        dec.decode_batch_device(llrs.data_ptr(), False, B, MAX_ITER, bits.data_ptr(), dec.k, its.data_ptr(), 0,
                                stream.cuda_stream)

    run()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize(device)
    clean = time.perf_counter() - t0
    its_np, bits_np = its.cpu().numpy(), bits.cpu().numpy()
    assert (its_np == -1).all(), "config 3 fixed-work operating point violated: some frames converged"
    # one more pass with the HIP-event brackets (they slow a launch-bound sequence: not the timed one)
    dec.set("profiling", 1)
    dec.kernel_stats(2, reset=True)
    run()
    launches, ms = dec.kernel_stats(2)
    dec.set("profiling", 0)
    E, n, k = dec.edges, dec.n, dec.k
    layers = dec.get("layers")
    lanes = max(dec.get("last_lanes"), 1)
    cw_s = B * steps / clean
    bytes_cw_iter = (4 * E + n) * 4
    avg_us = ms / max(launches, 1) * 1e3
    # a level launch processes the codewords of ONE execution lane (half of the batch when the two lanes
    # overlap) and, on average, 1/layers of the edges: 4 words per edge (read + write R, read + write Qv)
    level_bytes = 4 * E * 4 * (B / lanes) / layers
    gbps = level_bytes / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
    traffic, traffic_source, valu = None, None, None
    v, v_source = None, None
    if live:
        traffic, v = live_config3_counters()
        if traffic is not None:
            traffic_source = ("measured in this run: rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU) of "
                              "tools/perf_probe.py on this configuration as child processes")
            v_source = "SQ_INSTS_VALU measured in this run (rocprofv3 child pass)"
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            if traffic is None:
                traffic = t.get("hl_level_reg_kernel_tanh_bytes_per_launch")
                if traffic is not None:
                    traffic_source = f"profiles/hbm_traffic.json ({t.get('collected', '')}); not re-measured in this run"
            if v is None:
                v = t.get("hl_level_reg_kernel_tanh_valu_wave_insts_per_launch")
                v_source = f"SQ_INSTS_VALU from profiles/hbm_traffic.json ({t.get('collected', '')}; not re-measured here)"
            salu = t.get("hl_level_reg_kernel_tanh_salu_wave_insts_per_launch")
            if v:
                # the kernel's real bound: vector-ALU issue.  Counter: SQ_INSTS_VALU per level launch of one
                # 4096-codeword lane (profiles/r03_config3_counters.txt); peak: VALU_PEAK_WAVE_INSTS_PER_S above
                per_cw_iter = v * layers / 4096.0
                achieved = cw_s * MAX_ITER * per_cw_iter
                valu = {"bound": "valu", "wave_insts_per_level_launch": v, "wave_insts_per_codeword_iteration": per_cw_iter,
                        "achieved": achieved / 1e9, "peak": VALU_PEAK_WAVE_INSTS_PER_S / 1e9, "unit": "G wavefront-instructions/s",
                        "frac": achieved / VALU_PEAK_WAVE_INSTS_PER_S,
                        "mix_ceiling": VALU_MIX_CEILING_WAVE_INSTS_PER_S / 1e9,
                        "frac_of_mix_ceiling": achieved / VALU_MIX_CEILING_WAVE_INSTS_PER_S,
                        "salu_over_valu": (salu / t["hl_level_reg_kernel_tanh_valu_wave_insts_per_launch"]
                                           if salu and t.get("hl_level_reg_kernel_tanh_valu_wave_insts_per_launch") else None),
                        "source": v_source + " x this run's throughput; peak = the chip's vector-instruction issue rate "
                                  "(1024 SIMDs, one wave64 instruction per 2 cycles at 2.4 GHz); mix_ceiling = the rate the "
                                  "rule's own instruction mix (tanhf + atanh, 172 vector instructions per edge) reaches "
                                  "alone on the chip, profiles/r03_packed_f32.txt"}
        except (OSError, ValueError):
            pass      # no committed counter file: a live value measured above is kept
    out = {
        "metric": "codewords/s, 5G NR BG1 Zc=384 horizontal-layered sum-product (tanh) f32, 50 iterations",
        "value": cw_s, "unit": "codewords/s", "info_bits_per_s": cw_s * k, "steps": steps,
        "ms_per_step": clean / steps * 1e3, "dtype": "f32",
        "config": {"workload": f"5G NR BG1 Zc=384 (n={n}, k={k}, E={E}), {C3_IMPL}, {MAX_ITER} iterations, "
                               f"batch={B} codewords resident in HBM, Eb/N0={C3_EBN0_DB} dB (fixed work)",
                   "code": C3_SPEC, "implementation": C3_IMPL, "max_iterations": MAX_ITER, "batch": B,
                   "dependency_levels": layers, "execution_lanes": lanes},
        "roofline": {"bound": "hbm", "kernel": "hl_level_reg_kernel<Tanh, float, DMAX> (one launch per dependency level)",
                     "achieved": cw_s * MAX_ITER * bytes_cw_iter / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": cw_s * MAX_ITER * bytes_cw_iter / 1e9 / HBM_PEAK_GBPS,
                     "traffic": traffic, "traffic_source": traffic_source,
                     # counter bytes of a whole iteration (all level launches) per codeword, and the rate they cross the
                     # fabric at in the timed job: traffic * levels / codewords per counted launch (4096: one lane)
                     "traffic_bytes_per_codeword_iteration": (traffic * layers / 4096.0 if traffic else None),
                     "traffic_over_algorithmic": (traffic * layers / 4096.0 / bytes_cw_iter if traffic else None),
                     "traffic_frac_of_peak": (traffic * layers / 4096.0 * cw_s * MAX_ITER / 1e9 / HBM_PEAK_GBPS if traffic else None),
                     "formulas": {"frac": "value * max_iterations * (4E + N) * 4 B / peak (the whole timed job)",
                                  "traffic_frac_of_peak": "traffic * dependency_levels / 4096 * value * max_iterations / peak"},
                     "per_launch": {"algorithmic_bytes": level_bytes, "avg_us": avg_us, "launches": launches,
                                    "achieved": gbps, "frac": gbps / HBM_PEAK_GBPS},
                     "note": "achieved / frac: the whole timed job (algorithmic bytes of all codeword-iterations over the "
                             f"wall time of the {steps} timed steps).  {lanes} execution lane(s): a launch covers "
                             f"{B // lanes} codewords and the lanes' launches overlap on the chip, so per_launch (one "
                             "launch's own duration, measured in a separate bracketed pass) understates the chip's rate.  "
                             "The kernel is bound by vector-ALU issue (glibc-exact tanhf / log1pf: about 200 vector "
                             "instructions per edge), not by HBM: see valu_roofline and profiles/r04_slice_persistent.txt"},
        "whole_job_frac": cw_s * MAX_ITER * bytes_cw_iter / 1e9 / HBM_PEAK_GBPS,
        "valu_roofline": valu,
    }
    # the opt-in approximate variant of the same rule (native exp2 / log2 / rcp; NOT bit-identical, never the default):
    # what the exact functions cost, on the same frames
    try:
        fdec = lt.LdpcDecoder(alist, C3_IMPL + "@fast", device=device_index)
        fbits = torch.zeros((B, dec.k), dtype=torch.uint8, device=device)
        fits = torch.zeros(B, dtype=torch.int32, device=device)

        def frun():
            fdec.decode_batch_device(llrs.data_ptr(), False, B, MAX_ITER, fbits.data_ptr(), dec.k, fits.data_ptr(), 0, stream.cuda_stream)

        frun()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            frun()
        torch.cuda.synchronize(device)
        fclean = time.perf_counter() - t0
        fcw = B * steps / fclean
        out["fast_variant"] = {"implementation": C3_IMPL + "@fast", "value": fcw, "unit": "codewords/s",
                               "whole_job_frac": fcw * MAX_ITER * bytes_cw_iter / 1e9 / HBM_PEAK_GBPS,
                               "all_frames_ran_all_iterations": bool((fits.cpu().numpy() == -1).all()),
                               "note": "opt-in by name, not bit-identical to the reference (profiles/r03_fast_variants.txt: same "
                                       "frame-error counts as the exact rule on waterfall batches, a handful of frames with "
                                       "another iteration count); never `value`"}
        fdec.close()
    except Exception as e:      # the exact measurement above stands on its own
        out["fast_variant"] = {"error": str(e)}
    # the same configuration with syndrome early termination at Eb/N0 = +2 dB (realistic_point: best of two calls)
    try:
        r = realistic_point(dec, alist, C3_IMPL, B, device, device_index, stream, ebn0_db=2.0, pool=C3_POOL)
        r["fraction_of_iteration_proportional_bound"] = r["codewords_per_s"] / (cw_s * MAX_ITER / max(r["average_iterations"], 1e-9))
        out["realistic"] = r
    except Exception as e:      # the fixed-work measurement above stands on its own
        out["realistic"] = {"error": str(e)}
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(alist, C3_IMPL, llrs, bits_np, its_np, k, budget_s=8.0)
    return out


def host_cpu_info():
    """/proc/cpuinfo: model name, sockets, physical cores (distinct (physical id, core id) pairs), hardware threads"""
    model, cores, sockets, threads = None, set(), set(), 0
    phys = None
    try:
        for line in open("/proc/cpuinfo"):
            key, _, val = line.partition(":")
            key, val = key.strip(), val.strip()
            if key == "processor":
                threads += 1
            elif key == "model name" and model is None:
                model = val
            elif key == "physical id":
                phys = val
                sockets.add(val)
            elif key == "core id":
                cores.add((phys, val))
    except OSError:
        pass
    return {"cpu_model": model, "sockets": len(sockets) or None, "physical_cores": len(cores) or None,
            "hardware_threads": threads or os.cpu_count()}


def cpu_baseline(alist, impl, llrs, gpu_bits, gpu_its, k, budget_s=20.0):
    """The oracle (C restatement of the reference's decoder with its data structures: AoS message
    slots, linear-search send, full syndrome check per iteration, one decoder per worker thread)
    timed on this box's host cores over a bounded sample of the same frames.  Decode calls only: the
    workers build their decoders before the clock starts (oracle_decode_batch_timed_f32).  The worker
    count is swept over {cores/2, cores, 2*cores} with one frame per worker (the reference's driver
    defaults to one worker per hardware thread, src/cli/ber.rs:85-86), then the best count decodes up
    to 4 frames per worker (as many as fit `budget_s`): that run is the reported value."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    g = ob.Graph(alist)
    cores = os.cpu_count() or 1
    sweep = sorted({max(1, cores // 2), cores, 2 * cores})
    tried, same = [], True
    best = None

    def run(threads, count):
        nonlocal same
        count = min(count, llrs.shape[0])
        sample = llrs[:count].cpu().numpy()
        obits, oits, dt = ob.decode_batch_timed(g, impl, sample, MAX_ITER, threads=threads)
        same = same and bool(np.array_equal(oits, gpu_its[:count]) and np.array_equal(obits[:, :k], gpu_bits[:count]))
        tried.append({"threads": threads, "frames": count, "seconds": dt, "codewords_per_s": count / dt})
        return count / dt, count, dt

    for threads in sweep:
        rate, count, dt = run(threads, threads)
        if best is None or rate > best[0]:
            best = (rate, threads, count, dt)
    per_worker = int(max(1, min(4, budget_s / max(best[3], 1e-3))))
    if per_worker > 1:
        rate, count, dt = run(best[1], best[1] * per_worker)
        best = (rate, best[1], count, dt)
    host = host_cpu_info()
    return {"value": best[0], "unit": "codewords/s", "cores": best[1], "kind": "port", **host,
            "cores_note": f"`cores` = worker threads used ({best[1]}); the box has {host['physical_cores']} physical cores / "
                          f"{host['hardware_threads']} hardware threads in {host['sockets']} socket(s) of {host['cpu_model']}",
            "sample": f"first {best[2]} frames of the GPU batch ({best[2] // best[1]} per worker), {MAX_ITER} iterations "
                      f"each, {best[1]} worker threads with their decoders built before the clock starts, {best[3]:.1f} s; "
                      f"worker count chosen from {sweep} on {cores} hardware threads (one frame per worker each)",
            "sweep": tried, "matches_gpu_output": same}


if __name__ == "__main__":
    main()
