"""Multi-GPU sharding of the decode path: codewords are independent, so the batch axis is
split contiguously over ranks (one process per GPU) and there is NO collective on the data
path.  The only exchange is the host-side sum of the per-rank error counters -- the fields
the reference's BER driver folds on its main thread
(/root/reference/src/simulation/ber.rs:113-138, 313-338)."""
import numpy as np

COUNTER_FIELDS = ("num_frames", "bit_errors", "frame_errors", "false_decodes", "total_iterations",
                  "correct_iterations")


# outer-BCH view of the same frames (ber.rs:130, 328-337), appended when --bch-max-errors > 0
BCH_COUNTER_FIELDS = ("bch_bit_errors", "bch_frame_errors", "bch_correct_iterations")


def shard_range(total: int, rank: int, world: int):
    """Frames [begin, end) of `total` owned by `rank`: contiguous, sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    per, extra = divmod(total, world)
    begin = rank * per + min(rank, extra)
    return begin, begin + per + (1 if rank < extra else 0)


def counters_from_statistics(st) -> np.ndarray:
    c = [st.num_frames, st.ldpc.bit_errors, st.ldpc.frame_errors, st.false_decodes,
         st.total_iterations, st.ldpc.correct_iterations]
    if getattr(st, "bch", None) is not None:
        c += [st.bch.bit_errors, st.bch.frame_errors, st.bch.correct_iterations]
    return np.array(c, dtype=np.int64)


def group_is_up() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def reduce_counters(counters: np.ndarray, device=None) -> np.ndarray:
    """Sum of the six (nine with BCH accounting) u64-style counters over all ranks (identity when no process group is up).
    Uses whatever process group is initialised: RCCL ("nccl") with a device tensor on the GPU
    box, gloo with a CPU tensor in the CPU tests.  Latency-only: 48 bytes."""
    import torch
    import torch.distributed as dist
    # (a process group of one rank still goes through the collective: that is what lets a one-GPU box rehearse the path)
    if not (dist.is_available() and dist.is_initialized()):
        return counters.copy()
    t = torch.from_numpy(counters.astype(np.int64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
