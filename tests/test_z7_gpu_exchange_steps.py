"""Every exchange step of the multi-GPU path with REAL kernels behind it:

* shared-root root-parallel MCTS: `distributed.reduce_root_statistics` over real `mcts_search` replicas;
* the C++ host route: `Communicator` + `ExternalSamplingMCCFRSolver::RunShardedMiniBatch` (synchronous and
  overlapped) in a compiled program over the C-ABI collective (RCCL resolved by dlopen);
* the overlapped (stale-by-one) ES-MCCFR schedule of `ShardedMccfr(overlap=True)` on the device solver;
* RCCL at world size 2 — both routes, and `bench.py --gpus 2` — wherever the box has two GPUs (skipped on one:
  two ranks cannot share a device under RCCL).

World size 1 runs the same code paths on one GPU: the collective is issued and is the identity."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import open_spiel_amd as osa
    return osa.Context(0)


def test_root_parallel_statistics_of_real_searches(ctx):
    """R independent searches of ONE shared root (replicas of the root in a batch: replica i draws from the streams
    of global index i), their root children's visit / reward vectors summed by reduce_root_statistics (one
    all-reduce on a multi-GPU job: the identity here), the move chosen on the sums."""
    import torch
    import open_spiel_amd as osa
    from open_spiel_amd import distributed as osd
    R, sims = 64, 256
    for game, moves in (("hex(board_size=5)", [12, 6, 7]), ("connect_four", [3, 3, 2, 4, 2])):
        one = osa.StateBatch(ctx, game, 1)
        for a in moves:
            one.apply_actions(torch.tensor([a], dtype=torch.int32, device="cuda"))
        replicas = one.gather(torch.zeros(R, dtype=torch.int64, device="cuda"))
        res = replicas.mcts_search(uct_c=2.0, max_simulations=sims, n_rollouts=1, seed=99)
        visits, reward, best = osd.reduce_root_statistics(res["child_visits"], res["child_reward"])
        A = one.num_distinct_actions
        legal = one.legal_actions_mask()[0, :A].bool().cpu()
        v, r = visits.cpu(), reward.cpu()
        # every simulation but the first of each replica visits exactly one root child
        assert int(v.sum()) == R * (sims - 1)
        assert bool((v[~legal] == 0).all()) and bool((r.abs() <= v.to(torch.float64) + 1e-9).all())
        assert torch.equal(v, res["child_visits"].sum(0).to(torch.int64).cpu())
        assert bool(legal[best]) and int(v[best]) == int(v.max())
        tied = (v == v.max()).nonzero().flatten().tolist()
        assert float(r[best]) == max(float(r[a]) for a in tied)
        # the replicas really are different searches of the same position
        assert len({tuple(row.tolist()) for row in res["child_visits"].cpu()}) > 1
        # and the pooled choice is the one a single search with R x the simulations prefers among the top moves:
        # both rank the pooled best action among their three most visited
        big = one.mcts_search(uct_c=2.0, max_simulations=R * sims // 4, n_rollouts=1, seed=5)
        top3 = torch.topk(big["child_visits"][0].cpu(), 3).indices.tolist()
        assert best in top3, (game, best, top3)


CPP_PROGRAM = r'''
#include <cmath>
#include <cstdio>
#include "open_spiel_amd/csrc/host/osg_spiel.h"
using namespace open_spiel::hip;
using algorithms::ExternalSamplingMCCFRSolver;
using algorithms::NashConv;
// largest difference relative to the largest table entry
static double MaxDiff(const algorithms::CFRInfoStateValuesTable& a, const algorithms::CFRInfoStateValuesTable& b) {
  double d = 0, scale = 1;
  for (const auto& kv : a) {
    const auto& o = b.at(kv.first);
    for (size_t k = 0; k < kv.second.cumulative_regrets.size(); ++k) {
      d = std::fmax(d, std::fabs(kv.second.cumulative_regrets[k] - o.cumulative_regrets[k]));
      d = std::fmax(d, std::fabs(kv.second.cumulative_policy[k] - o.cumulative_policy[k]));
      scale = std::fmax(scale, std::fmax(std::fabs(kv.second.cumulative_regrets[k]), std::fabs(kv.second.cumulative_policy[k])));
    }
  }
  return d / scale;
}
int main() {
  try {
    std::shared_ptr<const Game> game = LoadGame("leduc_poker");
    Communicator comm(0, 1, Communicator::NewId());
    if (comm.rank() != 0 || comm.world() != 1) return 2;
    auto shard = comm.Shard(1000);
    if (shard.first != 0 || shard.second != 1000) return 3;
    // synchronous: shard -> sample -> all-reduce -> fold == the unsharded mini-batch
    ExternalSamplingMCCFRSolver plain(*game, 7), sharded(*game, 7), lapped(*game, 7);
    plain.RunMiniBatch(4096);
    sharded.RunShardedMiniBatch(4096, comm);
    const double d_first = MaxDiff(plain.InfoStateValuesTable(), sharded.InfoStateValuesTable());
    for (int k = 1; k < 6; ++k) {
      plain.RunMiniBatch(4096);
      sharded.RunShardedMiniBatch(4096, comm);
    }
    const double d_sync = MaxDiff(plain.InfoStateValuesTable(), sharded.InfoStateValuesTable());
    // overlapped: mini-batch k + 1 sampled against tables without k's deltas; FinishSharded folds the last one.
    for (int k = 0; k < 6; ++k) lapped.RunShardedMiniBatch(4096, comm, /*overlap=*/true);
    lapped.FinishSharded(comm);
    const double d_lap_vs_sync = MaxDiff(plain.InfoStateValuesTable(), lapped.InfoStateValuesTable());
    std::printf("{\"first_minibatch_max_diff\": %.3e, \"sync_max_diff\": %.3e, \"overlap_differs_from_sync\": %s, \"trajectories\": %lld, \"nash_conv_overlap\": %.6f, \"nash_conv_sync\": %.6f}\n",
                d_first, d_sync, d_lap_vs_sync > 1e-9 ? "true" : "false", static_cast<long long>(lapped.TrajectoriesRun()),
                NashConv(*game, *lapped.AveragePolicy()), NashConv(*game, *sharded.AveragePolicy()));
    // kuhn: the overlapped schedule still converges inside the reference's bound (external_sampling_mccfr_test.cc:104-109)
    std::shared_ptr<const Game> kuhn = LoadGame("kuhn_poker");
    ExternalSamplingMCCFRSolver ks(*kuhn, 3);
    for (int k = 0; k < 300; ++k) ks.RunShardedMiniBatch(256, comm, true);
    ks.FinishSharded(comm);
    const double nc = NashConv(*kuhn, *ks.AveragePolicy());
    std::printf("{\"kuhn_nash_conv_overlap\": %.6f}\n", nc);
    return nc <= 0.05 ? 0 : 4;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 1;
  }
}
'''


def test_cpp_communicator_and_sharded_minibatch_program(tmp_path):
    """The C++ route of the exchange step, compiled and run on the device: Communicator (RCCL by dlopen) at world
    size 1 + RunShardedMiniBatch, synchronous (== the unsharded mini-batch) and overlapped (a different schedule
    that still converges)."""
    import __graft_entry__ as ge
    ge.build()
    src = tmp_path / "sharded.cpp"
    src.write_text(CPP_PROGRAM)
    exe = tmp_path / "sharded"
    lib_dir = os.path.join(ROOT, "open_spiel_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-I", ROOT, str(src), "-o", str(exe), "-L", lib_dir,
                           "-losg_hip", f"-Wl,-rpath,{lib_dir}"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout + r.stderr
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    print(recs)
    # relative to the largest table entry.  fp64 atomics leave the order of additions inside a launch free, so two runs
    # of the same mini-batch differ in the last bits, and later mini-batches (sampled from regret-matched policies of
    # those tables) amplify that: tight after one mini-batch, loose after six
    assert recs[0]["first_minibatch_max_diff"] < 1e-12 and recs[0]["sync_max_diff"] < 1e-6
    assert recs[0]["overlap_differs_from_sync"] is True and recs[0]["trajectories"] == 6 * 4096
    assert recs[0]["nash_conv_overlap"] < 4.75 and recs[0]["nash_conv_sync"] < 4.75   # below the uniform policy's 4.747
    assert recs[1]["kuhn_nash_conv_overlap"] <= 0.05


def test_overlapped_schedule_on_the_device_solver(ctx):
    """ShardedMccfr(overlap=True) on the device solver equals the schedule written out by hand with the two
    C-ABI halves on caller buffers (sample k into buffer k % 2, then fold buffer (k - 1) % 2), and the caller-buffer
    halves equal the solver's own delta tables when used synchronously."""
    import torch
    import open_spiel_amd as osa
    from open_spiel_amd import distributed as osd
    a = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
    b = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
    c = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
    d = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
    lap = osd.ShardedMccfr(a, overlap=True)
    bufs = [b.mccfr_new_delta_buffer(), b.mccfr_new_delta_buffer()]
    one = c.mccfr_new_delta_buffer()
    first = 0
    for k in range(8):
        lap.run_minibatch(21, 4096)
        b.mccfr_sample_into(bufs[k & 1], 21, 4096, first_trajectory=first)
        if k:
            b.mccfr_apply_deltas_from(bufs[(k - 1) & 1])
        c.mccfr_sample_into(one, 21, 4096, first_trajectory=first)      # synchronous use of a caller buffer
        c.mccfr_apply_deltas_from(one)
        d.run_mccfr(21, 4096, first_trajectory=first)
        first += 4096
    lap.finish()
    b.mccfr_apply_deltas_from(bufs[7 & 1])
    ta, tb, tc, td = a.tables(), b.tables(), c.tables(), d.tables()
    # (fp64 atomics leave the order of additions inside a launch free: two runs of one mini-batch differ in the last
    # bits, and eight mini-batches of regret matching amplify that to ~1e-9 relative)
    for key in ("regrets", "cum_policy"):
        np.testing.assert_allclose(ta[key], tb[key], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(tc[key], td[key], rtol=1e-6, atol=1e-6)
    assert np.abs(ta["regrets"] - td["regrets"]).max() > 1e-6, "stale-by-one must differ from the synchronous schedule"
    with pytest.raises(osa.OsgError):
        a.mccfr_sample_into(torch.zeros(5, dtype=torch.float64, device="cuda"), 1, 16)


def test_parked_search_without_its_answer_stays_parked(ctx):
    """osg_mcts_tree_advance with a NULL answer pointer for a request that was reported: the searches
    stay parked and report the request again instead of dereferencing NULL; with the answers they go on."""
    import ctypes as C
    import torch
    import open_spiel_amd as osa
    from open_spiel_amd import _abi
    lib = osa.lib()
    n = 256
    roots = osa.StateBatch(ctx, "connect_four", n)
    leaf = osa.StateBatch(ctx, "connect_four", n)
    cfg = _abi.MctsCfg()
    cfg.uct_c, cfg.max_simulations, cfg.n_rollouts, cfg.seed, cfg.layout = 2.0, 16, 1, 3, 1
    tree = C.c_void_p()
    _abi.check(lib.osg_mcts_tree_create(roots._h, C.byref(cfg), 1, C.byref(tree)))   # flag 1: priors from the caller
    try:
        req = torch.zeros(n, dtype=torch.uint8, device="cuda")
        counts = (C.c_int64 * 4)()
        _abi.check(lib.osg_mcts_tree_advance(tree, leaf._h, None, None, req.data_ptr(), 1 << 30, counts))
        assert counts[2] == n           # the first simulation evaluates the root: every search wants a value
        for _ in range(2):              # no answer brought: nothing moves, the request is repeated
            _abi.check(lib.osg_mcts_tree_advance(tree, leaf._h, None, None, req.data_ptr(), 1 << 30, counts))
            assert counts[2] == n and bool((req == 2).all())
        value = torch.zeros((n, 2), dtype=torch.float64, device="cuda")
        _abi.check(lib.osg_mcts_tree_advance(tree, leaf._h, None, value.data_ptr(), req.data_ptr(), 1 << 30, counts))
        assert counts[1] == n and bool((req == 5).all())   # second simulation: the ROOT's prior is wanted
        _abi.check(lib.osg_mcts_tree_advance(tree, leaf._h, None, value.data_ptr(), req.data_ptr(), 1 << 30, counts))
        assert counts[1] == n and bool((req == 5).all())   # value pointer given, prior missing: still parked
        ctx.synchronize()
    finally:
        lib.osg_mcts_tree_destroy(tree)
    # a destroyed context refuses new trees
    other = osa.Context(0, stream=torch.cuda.Stream())
    keep = osa.StateBatch(other, "tic_tac_toe", 4)
    other.close()
    t2 = C.c_void_p()
    assert lib.osg_mcts_tree_create(keep._h, C.byref(cfg), 0, C.byref(t2)) != 0
    assert b"destroyed" in lib.osg_last_error()


def test_context_trim_gives_the_node_pool_back(ctx):
    import torch
    import open_spiel_amd as osa
    roots = osa.StateBatch(ctx, "hex(board_size=9)", 1 << 12)
    ctx.trim()
    free0, _ = torch.cuda.mem_get_info()
    a = roots.mcts_search(uct_c=2.0, max_simulations=256, n_rollouts=1, seed=1)
    ctx.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 < free0 - (64 << 20), "a search without a node budget caches a large pool in the context"
    ctx.trim()
    free2, _ = torch.cuda.mem_get_info()
    assert free2 > free1 + (64 << 20)
    b = roots.mcts_search(uct_c=2.0, max_simulations=256, n_rollouts=1, seed=1)   # allocates again, same result
    assert torch.equal(a["child_visits"], b["child_visits"])


RCCL2_SCRIPT = r'''
import json, os, sys
sys.path.insert(0, os.environ["OSG_ROOT"])
import ctypes as C
import numpy as np, torch, torch.distributed as dist
import open_spiel_amd as osa
from open_spiel_amd import distributed as osd
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
ctx = osa.Context(rank)
out = {}
for overlap in (False, True):
    s = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
    sh = osd.ShardedMccfr(s, overlap=overlap)
    for _ in range(6):
        sh.run_minibatch(9, 1 << 14)
    sh.finish()
    t = s.tables()
    mine = torch.from_numpy(np.stack([t["regrets"], t["cum_policy"]])).cuda()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    out["overlap" if overlap else "sync"] = {"rank_diff": float((both[0] - both[1]).abs().max()),
                                             "regret_abs_sum": float(mine[0].abs().sum())}
    if rank == 0:   # the same schedule on one rank
        ref = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
        one = osd.ShardedMccfr(ref, overlap=overlap)
        one.rank, one.world_size = 0, 1
        for _ in range(6):
            one.run_minibatch(9, 1 << 14)
        one.finish()
        tr = ref.tables()
        out["overlap" if overlap else "sync"]["vs_one_rank"] = float(max(np.abs(tr["regrets"] - t["regrets"]).max(),
                                                                         np.abs(tr["cum_policy"] - t["cum_policy"]).max()))
__EXCHANGE_AB__
# the C-ABI collective (osg_comm_*) over the same two GPUs: the unique id travels through torch's store
lib = osa.lib()
uid = C.create_string_buffer(128)
if rank == 0:
    assert lib.osg_comm_unique_id(uid) == 0, lib.osg_last_error()
t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).cuda()
dist.broadcast(t, 0)
uid = C.create_string_buffer(bytes(t.cpu().tolist()), 128)
comm = C.c_void_p()
assert lib.osg_comm_create(ctx._h, rank, world, uid, C.byref(comm)) == 0, lib.osg_last_error()
x = torch.arange(5616, dtype=torch.float64, device="cuda") * (rank + 1)
assert lib.osg_allreduce_sum_f64(comm, C.c_void_p(x.data_ptr()), x.numel()) == 0
y = torch.arange(5616, dtype=torch.float64, device="cuda") * (rank + 1)
assert lib.osg_allreduce_sum_f64_begin(comm, C.c_void_p(y.data_ptr()), y.numel()) == 0
assert lib.osg_allreduce_end(comm) == 0
ctx.synchronize(); torch.cuda.synchronize()
want = torch.arange(5616, dtype=torch.float64, device="cuda") * 3
out["abi_allreduce_ok"] = bool(torch.equal(x, want) and torch.equal(y, want))
lib.osg_comm_destroy(comm)
if rank == 0:
    print(json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


from exchange_ab_snippet import EXCHANGE_AB  # noqa: E402  (tests/ is on sys.path under pytest's rootdir conftest)
RCCL2_SCRIPT = RCCL2_SCRIPT.replace("__EXCHANGE_AB__", EXCHANGE_AB)


def _two_gpus():
    import torch
    return torch.cuda.is_available() and torch.cuda.device_count() >= 2


def test_rccl_world_size_two_both_routes(tmp_path):
    """RCCL over two GPUs (skipped on a one-GPU box): the sharded ES-MCCFR mini-batch through torch.distributed,
    synchronous and overlapped — both ranks end with identical tables, equal to the one-rank schedule up to the
    fp64 summation order — and the C-ABI collective, synchronous and begin / end."""
    if not _two_gpus():
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    script = tmp_path / "rccl2.py"
    script.write_text(RCCL2_SCRIPT)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OSG_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print(rec)
    for key in ("sync", "overlap"):
        assert rec[key]["rank_diff"] == 0.0, "every rank folds identical all-reduced deltas"
        assert rec[key]["vs_one_rank"] < 1e-8 * max(1.0, rec[key]["regret_abs_sum"])
    assert rec["abi_allreduce_ok"]
    ab = rec["exchange_ab"]   # OSG_COMM=rccl and OSG_COMM=oneshot carry the same exchange step: bit-identical tables
    assert ab["backend"] == "nccl" and ab["trained"]
    assert ab["routes_identical_minibatches"] == ab["mini_batches"] == 8 and ab["tables_identical"] and ab["rank_diff"] == 0.0
    assert ab["max_err_vs_one_rank_over_scale"] <= 1e-10


def test_bench_gpus_2_over_rccl(tmp_path):
    """`python bench.py --gpus 2` as the driver starts it, over RCCL (skipped on a one-GPU box)."""
    if not _two_gpus():
        pytest.skip("needs two GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("OSG_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-pmc"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    import bench
    line, _full = bench.read_lines(r.stdout)
    assert line["n_gpus"] == 2 and line["collective_backend"] == "nccl" and line["rccl_world"] == 2
    assert line["secondary"]["mccfr"]["allreduce_us"]["rccl"] > 0 and line["secondary"]["mccfr"]["allreduce_us"]["oneshot"] > 0
    assert 0 < line["secondary"]["mcts"]["strong_scaling_efficiency"] < 1.5
