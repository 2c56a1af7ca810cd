"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed over RCCL.

Units are independent (states, search roots, MCCFR trajectories) and are sharded by
contiguous GLOBAL index; every device RNG stream is keyed by the global index, so the
results are the same for any world size.  The only exchange step on the path is
external-sampling MCCFR's: one all-reduce(sum) of the regret / average-policy delta
tables per mini-batch (leduc_poker: 2 x [936, 3] fp64 = 44 928 B, latency-bound).

The reference has no distributed runtime at all (SURVEY.md header); this module is
new design, not a translation.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist


def world():
    """(rank, world_size) of the default process group, (0, 1) when not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(total, rank, world_size):
    """Contiguous slice [first, first + count) of `total` units owned by `rank`.

    The first `total % world_size` ranks own one extra unit, so any total is covered
    exactly once and slices are ordered by rank."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError(f"bad rank {rank} / world {world_size}")
    base, extra = divmod(int(total), world_size)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def allreduce_sum_(tensor, comm=None):
    """In-place sum over ranks: through `comm` (a OneShotComm) when given, else torch.distributed (RCCL over
    xGMI for device tensors, gloo for CPU tensors)."""
    if comm is not None:
        return comm.allreduce_sum_(tensor)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def collective_kind():
    """OSG_COMM=oneshot|rccl (default rccl): which collective carries the path's one exchange step."""
    kind = os.environ.get("OSG_COMM", "rccl").lower()
    if kind not in ("oneshot", "rccl"):
        raise ValueError(f"OSG_COMM={kind!r}: expected 'oneshot' or 'rccl'")
    return kind


class OneShotComm:
    """The one-shot all-reduce of include/osg_abi.h (osg_comm_oneshot_*) for the path's latency-bound messages
    (SURVEY.md section 5): every rank's window is mapped by every peer through hipIpc, one launch per call pushes
    the local buffer to all peers, waits for theirs and sums the slots in rank order (bit-identical on all ranks).

    The 128-byte window handles travel over `exchange` — a callable bytes -> [bytes of rank 0, rank 1, ...]; by
    default torch.distributed.all_gather_object on the default group (any backend: the handles are host bytes).
    Ranks may share a device."""

    def __init__(self, ctx, max_doubles, rank=None, world_size=None, exchange=None):
        from ._abi import check, lib
        if rank is None or world_size is None:
            rank, world_size = world()
        self.ctx, self.rank, self.world_size, self.max_doubles = ctx, int(rank), int(world_size), int(max_doubles)
        self._lib, self._check = lib(), check
        # Every step that can fail on ONE rank (window allocation, hipIpc export / open) is followed by a round on the
        # host channel in which the ranks agree on the outcome: either all ranks hold a connected communicator or all
        # raise here — no rank is left waiting in a collective for a peer that gave up.
        self._h = None
        if exchange is None:
            def exchange(blob):
                if self.world_size == 1:
                    return [blob]
                out = [None] * self.world_size
                dist.all_gather_object(out, blob)
                return out
        failure, mine = None, b""
        try:
            h = C.c_void_p()
            check(self._lib.osg_comm_oneshot_create(ctx._h, self.rank, self.world_size, self.max_doubles, C.byref(h)))
            self._h = h
            buf = C.create_string_buffer(128)
            check(self._lib.osg_comm_oneshot_handle(self._h, buf))
            mine = buf.raw
        except Exception as e:  # noqa: BLE001 - reported to every rank below
            failure = e
        blobs = exchange(mine)
        if len(blobs) != self.world_size:
            self.close()
            raise ValueError("OneShotComm: exchange() must return one entry per rank, in rank order")
        if any(len(b) != 128 for b in blobs):
            self.close()
            bad = [r for r, b in enumerate(blobs) if len(b) != 128]
            raise RuntimeError(f"OneShotComm: rank(s) {bad} could not create or export their window"
                               + (f" (this rank: {failure})" if failure else ""))
        try:
            check(self._lib.osg_comm_oneshot_connect(self._h, C.create_string_buffer(b"".join(blobs), 128 * self.world_size)))
        except Exception as e:  # noqa: BLE001
            failure = e
        verdicts = exchange((b"\x00" if failure else b"\x01") * 128)
        if any(v[:1] != b"\x01" for v in verdicts):
            self.close()
            bad = [r for r, v in enumerate(verdicts) if v[:1] != b"\x01"]
            raise RuntimeError(f"OneShotComm: rank(s) {bad} could not map their peers' windows (hipIpcOpenMemHandle)"
                               + (f" (this rank: {failure})" if failure else ""))

    def _ptr(self, tensor, begin=False):
        if not (isinstance(tensor, torch.Tensor) and tensor.is_cuda and tensor.is_contiguous()
                and tensor.device == self.ctx.device):
            raise ValueError(f"OneShotComm: need a contiguous tensor on {self.ctx.device}")
        if tensor.dtype not in ((torch.float64,) if begin else (torch.float64, torch.int32)):
            raise ValueError("OneShotComm: float64 (or int32, synchronous form only) tensors")
        return C.c_void_p(tensor.data_ptr())

    def allreduce_sum_(self, tensor):
        """In place, on the context's stream (ordered with the kernels before and after it; no host wait)."""
        fn = self._lib.osg_allreduce_sum_f64 if tensor.dtype == torch.float64 else self._lib.osg_allreduce_sum_i32
        self._check(fn(self._h, self._ptr(tensor), tensor.numel()))
        return tensor

    def begin(self, tensor):
        """Asynchronous form: the collective runs on the communicator's own stream, after everything issued on the
        context's stream so far; kernels issued before end() overlap it."""
        self._check(self._lib.osg_allreduce_sum_f64_begin(self._h, self._ptr(tensor, begin=True), tensor.numel()))

    def end(self):
        """Orders the context's stream after the collective begun last (no host wait)."""
        self._check(self._lib.osg_allreduce_end(self._h))

    def check(self):
        """Waits for the collectives issued so far and raises if one of them timed out (the buffer of that call then
        holds NaN in the chunks that were not reduced).  Call it before trusting the result of the LAST collective of
        a job; a timeout is otherwise reported by the next call on the communicator."""
        self._check(self._lib.osg_comm_check(self._h))

    def close(self):
        if self._h:
            self._lib.osg_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedMccfr:
    """External-sampling MCCFR with trajectories sharded over the ranks.

    `solver` is anything with the TabularSolver mini-batch protocol:
      mccfr_sample(seed, count, first_trajectory) -> deltas left in the delta tables
      mccfr_delta_tables() -> (regret_delta, policy_delta) tensors (views, same device)
      mccfr_apply_deltas()
    Every rank ends each mini-batch with identical tables.

    overlap=True (the solver must also offer mccfr_new_delta_buffer / mccfr_sample_into /
    mccfr_apply_deltas_from): two delta buffers; the all-reduce of mini-batch k is issued asynchronously
    (torch.distributed runs it on the process group's own stream) and mini-batch k + 1 is sampled while it is
    in flight, against tables that do not hold k's deltas yet (stale by one mini-batch); k's deltas are folded
    when they have arrived, before mini-batch k + 2.  finish() folds what is still pending.  Ranks still end
    with identical tables; the schedule differs from the synchronous one (a different, still convergent,
    sequence of tables), which is why time-to-NashConv is reported for both (bench.py secondary.mccfr.quality).
    """

    def __init__(self, solver, overlap=False, comm=None):
        """comm: a OneShotComm that carries the delta all-reduce instead of torch.distributed; by default one is
        created when OSG_COMM=oneshot, the world has more than one rank and the solver lives on a device."""
        self.solver = solver
        self.rank, self.world_size = world()
        if comm is None and self.world_size > 1 and hasattr(solver, "mccfr_delta_flat") and collective_kind() == "oneshot":
            comm = OneShotComm(solver.ctx, solver.mccfr_delta_flat().numel())
        self.comm = comm
        self.trajectories_done = 0
        self.overlap = bool(overlap)
        self._flat = None
        self._bufs = None
        self._pending = None   # (buffer index, work handle or None)
        self._k = 0

    def run_minibatch(self, seed, trajectories):
        """One mini-batch of `trajectories` traversals (global count) starting at the
        running global trajectory index; returns the number this rank sampled."""
        first, count = shard_range(trajectories, self.rank, self.world_size)
        if self.overlap:
            return self._run_overlapped(seed, trajectories, first, count)
        self.solver.mccfr_sample(seed, count, first_trajectory=self.trajectories_done + first)
        if self.world_size > 1:
            if hasattr(self.solver, "mccfr_delta_flat"):
                # the device solver: both tables are one allocation -> one collective, in place
                allreduce_sum_(self.solver.mccfr_delta_flat(), self.comm)
            else:
                dreg, dpol = self.solver.mccfr_delta_tables()
                flat = torch.cat([dreg.reshape(-1), dpol.reshape(-1)])
                allreduce_sum_(flat)
                dreg.copy_(flat[:dreg.numel()].view_as(dreg))
                dpol.copy_(flat[dreg.numel():].view_as(dpol))
        self.solver.mccfr_apply_deltas()
        self.trajectories_done += int(trajectories)
        return count

    def _run_overlapped(self, seed, trajectories, first, count):
        if self._bufs is None:
            self._bufs = [self.solver.mccfr_new_delta_buffer(), self.solver.mccfr_new_delta_buffer()]
        cur = self._k & 1
        buf = self._bufs[cur]
        # the buffer's previous deltas (mini-batch k - 2) were folded during mini-batch k - 1
        self.solver.mccfr_sample_into(buf, seed, count, first_trajectory=self.trajectories_done + first)
        work = None
        if self.world_size > 1 and self.comm is not None:
            # one collective in flight per communicator: mini-batch k - 1's sum is folded first (its all-reduce ran
            # while k was being sampled), then k's begins on the communicator's stream behind the traversals
            self._fold_pending()
            self.comm.begin(buf)
            work = self.comm
        elif self.world_size > 1:
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
        self._fold_pending()                       # mini-batch k - 1: waits for ITS all-reduce only
        self._pending = (cur, work)
        self._k += 1
        self.trajectories_done += int(trajectories)
        return count

    def _fold_pending(self):
        if self._pending is None:
            return
        idx, work = self._pending
        if work is self.comm and work is not None:
            work.end()     # the context's stream waits for the collective, the host does not
        elif work is not None:
            work.wait()    # device tensors: the current stream waits, the host does not
        self.solver.mccfr_apply_deltas_from(self._bufs[idx])
        self._pending = None

    def finish(self):
        """Fold the deltas still in flight (overlap=True), then make sure no exchange step of the run failed: with the
        one-shot collective a timeout of the LAST call has no later call to report it (osg_comm_check; a failed call
        leaves NaN in the delta buffer, so tables it was folded into are NaN, never plausible-but-different)."""
        self._fold_pending()
        if self.comm is not None and self.world_size > 1:
            # A one-shot timeout poisons and flags only the rank whose wait ran out (include/osg_abi.h): the ranks agree
            # on the verdict — one MIN over the ranks on the process group the handles travelled on — so that either
            # every rank trusts its tables or every rank raises.
            err = None
            try:
                self.comm.check()
            except Exception as e:  # noqa: BLE001 - re-raised below, on every rank
                err = e
            ok = 0 if err is not None else 1
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() == self.world_size:
                flag = torch.tensor([ok], dtype=torch.int32,
                                    device="cuda" if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            if err is not None:
                raise err
            if not ok:
                from ._abi import OsgError
                raise OsgError("a peer's one-shot all-reduce timed out: its tables are poisoned, this rank's are not "
                               "trustworthy as a set (ShardedMccfr.finish agrees on the verdict across the ranks)")


def gather_root_results(local, total_roots):
    """All-gather per-root result tensors (e.g. best actions) sharded by shard_range
    into one [total_roots, ...] tensor on every rank (variable shard sizes allowed)."""
    rank, world_size = world()
    if world_size == 1:
        return local
    counts = [shard_range(total_roots, r, world_size)[1] for r in range(world_size)]
    pad = max(counts)
    buf = local.new_zeros((pad,) + tuple(local.shape[1:]))
    buf[:local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world_size)]
    dist.all_gather(out, buf)
    return torch.cat([o[:c] for o, c in zip(out, counts)])


def reduce_root_statistics(child_visits, child_reward):
    """Root-parallel MCTS on ONE shared root: every rank (and every local replica of the
    root) runs an independent search with its own random streams; the per-action statistics of the root's
    children are summed over the local replicas and then over the ranks with a single all-reduce
    ([A] visits + [A] total reward: ~1 KB for hex(9)), and the move is chosen on the sums with the
    reference's final ordering, visits first, then total reward (mcts.cc:114-125 without proven outcomes).

    child_visits: [replicas, A] integer tensor, child_reward: [replicas, A] float64 (StateBatch.mcts_search
    outputs).  Returns (visits [A] int64, reward [A] float64, best_action int)."""
    visits = child_visits.to(torch.float64).sum(0)
    reward = child_reward.to(torch.float64).sum(0)
    packed = torch.cat([visits, reward])  # one collective for both vectors
    allreduce_sum_(packed)
    A = visits.numel()
    visits, reward = packed[:A], packed[A:]
    # lexicographic arg-max (visits, reward); ties -> lowest action id
    best, best_key = -1, None
    v_list, r_list = visits.tolist(), reward.tolist()
    for a in range(A):
        if v_list[a] <= 0:
            continue
        key = (v_list[a], r_list[a])
        if best_key is None or key > best_key:
            best, best_key = a, key
    return visits.to(torch.int64), reward, best
