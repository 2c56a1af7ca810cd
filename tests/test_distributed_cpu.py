"""The N>1 path on CPU: world_size-2 gloo process groups (no GPU).

Covers the host logic of open_spiel_amd/distributed.py: shard ranges, the MCCFR
delta all-reduce protocol (every rank ends with identical tables, equal to the
single-process result) and the variable-size root gather.  The device solver is
replaced by a deterministic stand-in with the same mini-batch protocol: no compute
kernel runs here.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class FakeSolver:
    """Stand-in with TabularSolver's mini-batch protocol; trajectory g contributes
    deterministic integer-valued deltas (so sums are exact in any order)."""

    I, A = 12, 2

    def __init__(self):
        self.tables = torch.full((2, self.I, self.A), 1e-6, dtype=torch.float64)
        self.deltas = torch.zeros((2, self.I, self.A), dtype=torch.float64)

    def mccfr_sample(self, seed, count, first_trajectory=0):
        self.deltas.zero_()
        for g in range(first_trajectory, first_trajectory + count):
            row = (g * 7 + seed) % self.I
            self.deltas[0, row, g % self.A] += float(g % 5 - 2)
            self.deltas[1, (row + 3) % self.I, (g + 1) % self.A] += 1.0

    def mccfr_delta_tables(self):
        return self.deltas[0], self.deltas[1]

    def mccfr_apply_deltas(self):
        self.tables += self.deltas

    # the double-buffered protocol (ShardedMccfr(overlap=True)): caller-owned delta buffers
    def mccfr_new_delta_buffer(self):
        return torch.zeros((2, self.I, self.A), dtype=torch.float64)

    def mccfr_sample_into(self, buf, seed, count, first_trajectory=0):
        self.mccfr_sample(seed, count, first_trajectory)
        buf.copy_(self.deltas)
        self.sampled_against.append(self.tables.clone())

    def mccfr_apply_deltas_from(self, buf):
        self.tables += buf

    sampled_against = None


class FakeFlatSolver(FakeSolver):
    """Same, but exposing the one-allocation view like the device solver."""

    def mccfr_delta_flat(self):
        return self.deltas


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from open_spiel_amd import distributed as osd
        assert osd.world() == (rank, world_size)
        solver = FakeSolver() if rank == 0 else FakeFlatSolver()  # both reduction paths interoperate
        sharded = osd.ShardedMccfr(solver)
        sampled = [sharded.run_minibatch(seed=11, trajectories=t) for t in (1, 5, 64, 1001)]
        # variable-size gather: 7 roots over 2 ranks -> 4 + 3
        first, count = osd.shard_range(7, rank, world_size)
        local = torch.arange(first, first + count, dtype=torch.int32).unsqueeze(1) * 10
        gathered = osd.gather_root_results(local, 7)
        # root-parallel MCTS statistics: rank r contributes 3 local replicas of fake per-action stats
        vis = torch.tensor([[1 + rank, 0, 5], [2, 0, 5 - rank], [0, 0, 1]], dtype=torch.int32)
        rew = torch.tensor([[0.5, 0.0, -1.0], [1.0, 0.0, 2.0 * rank], [0.0, 0.0, 0.25]], dtype=torch.float64)
        root_stats = osd.reduce_root_statistics(vis, rew)
        # the overlapped schedule: all-reduce of mini-batch k in flight while k + 1 is sampled
        lap = FakeSolver()
        lap.sampled_against = []
        over = osd.ShardedMccfr(lap, overlap=True)
        for t in (1, 5, 64, 1001):
            over.run_minibatch(seed=11, trajectories=t)
        before_finish = lap.tables.clone()
        over.finish()
        torch.save({"tables": solver.tables, "root_stats": root_stats, "sampled": sampled, "gathered": gathered,
                    "done": sharded.trajectories_done, "overlap_tables": lap.tables, "overlap_before_finish": before_finish,
                    "overlap_sampled_against": lap.sampled_against},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_everything_once():
    from open_spiel_amd.distributed import shard_range
    for total in (0, 1, 7, 8, 65536, (1 << 24) + 3):
        for world_size in (1, 2, 3, 4, 8):
            spans = [shard_range(total, r, world_size) for r in range(world_size)]
            assert spans[0][0] == 0
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert spans[-1][0] + spans[-1][1] == total
            counts = [c for _, c in spans]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_mccfr_delta_allreduce_world2_equals_world1(tmp_path):
    # single-process result
    from open_spiel_amd import distributed as osd
    ref = FakeSolver()
    single = osd.ShardedMccfr(ref)
    for t in (1, 5, 64, 1001):
        single.run_minibatch(seed=11, trajectories=t)
    # two gloo ranks
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    assert torch.equal(r0["tables"], r1["tables"]), "ranks must end with identical tables"
    assert torch.equal(r0["tables"], ref.tables), "world=2 must equal world=1"
    assert [a + b for a, b in zip(r0["sampled"], r1["sampled"])] == [1, 5, 64, 1001]
    assert r0["done"] == r1["done"] == single.trajectories_done == 1071
    for r in (r0, r1):
        visits, reward, best = r["root_stats"]
        assert visits.tolist() == [3 + 4, 0, 11 + 10] and reward.tolist() == [3.0, 0.0, 0.5]
        assert best == 2                       # most visits wins; action 1 was never visited
    want = (torch.arange(7, dtype=torch.int32) * 10).unsqueeze(1)
    assert torch.equal(r0["gathered"], want) and torch.equal(r1["gathered"], want)
    # overlapped (stale-by-one) schedule: the same deltas arrive, one mini-batch late — identical tables on both
    # ranks, equal to the synchronous result once finish() has folded the last mini-batch, and mini-batch k was
    # sampled against the tables holding the deltas of mini-batches < k - 1 only
    assert torch.equal(r0["overlap_tables"], r1["overlap_tables"])
    assert torch.equal(r0["overlap_tables"], ref.tables)
    assert not torch.equal(r0["overlap_before_finish"], ref.tables)
    stale = FakeSolver()
    stale.sampled_against = []
    one = osd.ShardedMccfr(stale, overlap=True)
    for t in (1, 5, 64, 1001):
        one.run_minibatch(seed=11, trajectories=t)
    one.finish()
    assert torch.equal(stale.tables, ref.tables)
    for a, b in zip(stale.sampled_against, r0["overlap_sampled_against"]):
        assert torch.equal(a, b), "the tables a mini-batch is sampled against must not depend on the world size"
    fresh = torch.full((2, FakeSolver.I, FakeSolver.A), 1e-6, dtype=torch.float64)
    assert torch.equal(stale.sampled_against[0], fresh) and torch.equal(stale.sampled_against[1], fresh)
    assert not torch.equal(stale.sampled_against[2], fresh)


@pytest.mark.parametrize("world_size", [3, 8])
def test_mccfr_delta_allreduce_uneven_worlds_equal_world1(tmp_path, world_size):
    """Odd and full-node world sizes: mini-batches smaller than the world (ranks with an empty slice), uneven slices,
    the variable-size gather — every rank ends with the single-process tables."""
    from open_spiel_amd import distributed as osd
    ref = FakeSolver()
    single = osd.ShardedMccfr(ref)
    for t in (1, 5, 64, 1001):
        single.run_minibatch(seed=11, trajectories=t)
    port = _free_port()
    mp.spawn(_worker, args=(world_size, port, str(tmp_path)), nprocs=world_size, join=True)
    ranks = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world_size)]
    want = (torch.arange(7, dtype=torch.int32) * 10).unsqueeze(1)
    for r in ranks:
        assert torch.equal(r["tables"], ref.tables), f"world={world_size} must equal world=1"
        assert torch.equal(r["overlap_tables"], ref.tables)
        assert torch.equal(r["gathered"], want)
        assert r["done"] == 1071
    assert [sum(r["sampled"][k] for r in ranks) for k in range(4)] == [1, 5, 64, 1001]
    assert sum(1 for r in ranks if r["sampled"][0] == 0) == world_size - 1   # one trajectory: one rank samples it


def test_bench_gpus_n_starts_n_ranks_by_itself(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment re-executes itself under
    torch.distributed.run with one rank per GPU on a 127.0.0.1 port (the command the docstring shows);
    here only the command is checked (the run itself needs GPUs: tests/test_gpu_fullsize.py)."""
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.setenv("RANK", "7")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert "RANK" not in seen["env"] and "WORLD_SIZE" not in seen["env"] and seen["env"]["OSG_BENCH_SPAWNED"] == "1"
    # inside the spawned ranks (WORLD_SIZE set by the launcher) nothing is spawned again
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("OSG_BENCH_SPAWNED", "1")
    seen.clear()
    with pytest.raises(BaseException):
        bench.main()    # goes on to the devices, which this box does not have
    assert not seen


def test_cpp_host_shard_range_equals_the_python_one(tmp_path):
    """open_spiel::hip::ShardRange (csrc/host/osg_spiel.h; what Communicator::Shard and
    ExternalSamplingMCCFRSolver::RunShardedMiniBatch use) against distributed.shard_range: same slices,
    covering every total exactly once.  Host-only code: compiled and run here with g++."""
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    from open_spiel_amd.distributed import shard_range
    src = tmp_path / "shard.cpp"
    src.write_text(
        '#include <cstdio>\n#include "open_spiel_amd/csrc/host/osg_spiel.h"\n'
        "int main() {\n"
        "  for (long total : {0L, 1L, 7L, 64L, 65536L, 16777216L, 1000003L})\n"
        "    for (int world : {1, 2, 3, 4, 8})\n"
        "      for (int rank = 0; rank < world; ++rank) {\n"
        "        auto r = open_spiel::hip::ShardRange(total, rank, world);\n"
        '        std::printf("%ld %d %d %ld %ld\\n", total, world, rank, (long)r.first, (long)r.second);\n'
        "      }\n"
        "  return 0;\n}\n")
    exe = tmp_path / "shard"
    lib_dir = os.path.join(ROOT, "open_spiel_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", ROOT, str(src), "-o", str(exe),
                           "-L", lib_dir, "-losg_hip", f"-Wl,-rpath,{lib_dir}"])
    lines = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    seen = 0
    for line in lines:
        if not line:
            continue
        total, world, rank, first, count = (int(x) for x in line.split())
        assert shard_range(total, rank, world) == (first, count)
        seen += 1
    assert seen == 7 * (1 + 2 + 3 + 4 + 8)


class _FakeComm:
    """A one-shot communicator stand-in: sums through the process group, and check() fails on ONE rank only — what a
    one-shot timeout looks like (include/osg_abi.h: poison and sticky error are local to the rank whose wait ran out)."""

    def __init__(self, fail_on_rank):
        self.fail = dist.get_rank() == fail_on_rank

    def allreduce_sum_(self, t):
        dist.all_reduce(t)
        return t

    def check(self):
        if self.fail:
            from open_spiel_amd._abi import OsgError
            raise OsgError("one-shot all-reduce: timed out waiting for a peer (stand-in)")


def _agree_worker(rank, world_size, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from open_spiel_amd import distributed as osd
        from open_spiel_amd._abi import OsgError
        verdicts = []
        for fail_on in (1, -1):      # a timeout on rank 1 only, then a healthy run
            sharded = osd.ShardedMccfr(FakeFlatSolver(), comm=_FakeComm(fail_on))
            sharded.run_minibatch(seed=3, trajectories=64)
            try:
                sharded.finish()
                verdicts.append("ok")
            except OsgError as e:
                verdicts.append("own" if "stand-in" in str(e) else ("peer" if "a peer's one-shot" in str(e) else "?"))
        torch.save(verdicts, os.path.join(out_dir, f"agree{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_one_shot_timeout_verdict_is_agreed_by_all_ranks(tmp_path):
    """ShardedMccfr.finish(): a one-shot timeout is reported by osg_comm_check on the rank whose wait ran out only — the
    ranks agree on the verdict (one MIN over the process group) so that EVERY rank raises, none folds on alone."""
    port = _free_port()
    mp.spawn(_agree_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    v0 = torch.load(os.path.join(tmp_path, "agree0.pt"))
    v1 = torch.load(os.path.join(tmp_path, "agree1.pt"))
    assert v0 == ["peer", "ok"] and v1 == ["own", "ok"], (v0, v1)
