"""The one-shot all-reduce (include/osg_abi.h osg_comm_oneshot_*; SURVEY.md section 5 prefers it to a ring for the
path's <= 45 KB messages) with TWO ranks on ONE device: two processes map each other's windows through hipIpc just
as two GPUs would, so the 1-GPU box exercises the whole protocol (RCCL refuses two ranks on one device).

What is checked: sums of many message sizes (1 element ... the window's capacity, chunk boundaries included), fp64
and int32, hundreds of back-to-back calls with changing data (a stale line or a lost flag shows as a wrong sum), both
ranks bit-identical; the begin / end form; sharded ES-MCCFR over it ending with identical tables on both ranks that
equal the one-rank schedule up to fp64 summation order (external_sampling_mccfr.cc:122-186 is what is summed); a
peer that never arrives is a reported timeout, not a hang — reported by osg_comm_check and by every later call, with the
caller's buffer poisoned (NaN) instead of left as a mix of local and reduced chunks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.environ["OSG_ROOT"])
import numpy as np, torch, torch.distributed as dist
import open_spiel_amd as osa
from open_spiel_amd import distributed as osd
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)                       # every rank on the same device
ctx = osa.Context(0)
out = {}
cap = 32768
comm = osd.OneShotComm(ctx, cap)
gen = torch.Generator(device="cuda"); gen.manual_seed(1234 + rank)
bad = 0
sizes = [1, 2, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 5616, 20000, cap]
for rep in range(30):
    for n in sizes:
        x = torch.randn(n, dtype=torch.float64, device="cuda", generator=gen) * (1 + rep)
        parts = [torch.empty(n, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(parts, x.cpu())
        want = parts[0].clone()
        for p in parts[1:]:
            want += p                          # rank order, like the kernel
        comm.allreduce_sum_(x)
        ctx.synchronize()
        bad += int(not torch.equal(x.cpu(), want))
out["f64_mismatches"] = bad
bad = 0
for rep in range(10):
    for n in (1, 3, 513, 11232, cap):
        x = torch.randint(-1000, 1000, (n,), dtype=torch.int32, device="cuda", generator=gen)
        parts = [torch.empty(n, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(parts, x.cpu())
        comm.allreduce_sum_(x)
        ctx.synchronize()
        bad += int(not torch.equal(x.cpu(), sum(parts)))
out["i32_mismatches"] = bad
# back to back without host synchronisation: 500 calls on one buffer, values double every call (x -> world * x)
x = torch.full((5616,), 1.0, dtype=torch.float64, device="cuda")
for _ in range(40):
    comm.allreduce_sum_(x)
ctx.synchronize()
out["chain_ok"] = bool((x == float(world) ** 40).all())
# begin / end with work in between
y = torch.arange(5616, dtype=torch.float64, device="cuda") * (rank + 1)
comm.begin(y)
z = torch.zeros(1 << 20, device="cuda").add_(1.0).sum()
comm.end()
ctx.synchronize(); torch.cuda.synchronize()
out["begin_end_ok"] = bool(torch.equal(y, torch.arange(5616, dtype=torch.float64, device="cuda") * sum(range(1, world + 1))))
# latency of the 44 928-byte message (both ranks share the device here: an upper bound of the protocol's own cost)
flat = torch.ones(5616, dtype=torch.float64, device="cuda")
for _ in range(20):
    comm.allreduce_sum_(flat); flat.fill_(1.0)
ctx.synchronize(); dist.barrier()
t0 = time.perf_counter()
for _ in range(200):
    comm.allreduce_sum_(flat)
ctx.synchronize()
out["allreduce_us"] = (time.perf_counter() - t0) / 200 * 1e6
# argument checks
try:
    comm.allreduce_sum_(torch.zeros(cap + 2, dtype=torch.float64, device="cuda")); out["too_long_refused"] = False
except osa.OsgError:
    out["too_long_refused"] = True
comm.close()
# sharded ES-MCCFR over the one-shot collective (OSG_COMM=oneshot picks it), synchronous and overlapped
os.environ["OSG_COMM"] = "oneshot"
for overlap in (False, True):
    s = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
    sh = osd.ShardedMccfr(s, overlap=overlap)
    assert sh.comm is not None
    for _ in range(6):
        sh.run_minibatch(9, 1 << 14)
    sh.finish()
    t = s.tables()
    mine = torch.from_numpy(np.stack([t["regrets"], t["cum_policy"]]))
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    key = "overlap" if overlap else "sync"
    out[key] = {"rank_diff": float((both[0] - both[1]).abs().max()), "regret_abs_sum": float(mine[0].abs().sum())}
    if rank == 0:
        ref = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
        os.environ["OSG_COMM"] = "rccl"        # (no communicator for the one-rank reference: creating one is collective)
        one = osd.ShardedMccfr(ref, overlap=overlap)
        os.environ["OSG_COMM"] = "oneshot"
        one.rank, one.world_size = 0, 1
        for _ in range(6):
            one.run_minibatch(9, 1 << 14)
        one.finish()
        tr = ref.tables()
        out[key]["vs_one_rank"] = float(max(np.abs(tr["regrets"] - t["regrets"]).max(),
                                            np.abs(tr["cum_policy"] - t["cum_policy"]).max()))
    sh.comm.close()
dist.barrier()
__EXCHANGE_AB__
dist.barrier()
# a peer that never arrives: rank 1 does not call; rank 0 must get a timeout error, not hang
os.environ["OSG_ONESHOT_TIMEOUT_MS"] = "300"
lonely = osd.OneShotComm(ctx, 1024)
if rank == 0:
    v = torch.ones(1024, dtype=torch.float64, device="cuda")
    t0 = time.perf_counter()
    lonely.allreduce_sum_(v)
    try:
        lonely.check(); out["timeout_reported_by_check"] = False     # osg_comm_check: waits, then reports
    except osa.OsgError as e:
        out["timeout_reported_by_check"] = "timed out" in str(e)
    out["timeout_seconds"] = time.perf_counter() - t0
    out["timeout_poisoned_buffer"] = bool(torch.isnan(v).all())      # never a mix of local and reduced values
    try:
        lonely.allreduce_sum_(v); out["timeout_reported"] = False
    except osa.OsgError as e:
        out["timeout_reported"] = "timed out" in str(e)
dist.barrier()
# a rank that cannot create its window: BOTH ranks must raise at construction (nobody waits for the other in a collective)
try:
    osd.OneShotComm(ctx, 1024 if rank == 0 else -5)
    agreed = "no error"
except RuntimeError as e:
    agreed = str(e)
both = [None] * world
dist.all_gather_object(both, agreed)
out["failed_creation_agreed"] = all("rank(s) [1]" in m for m in both)
dist.barrier()
if rank == 0:
    print(json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


from exchange_ab_snippet import EXCHANGE_AB  # noqa: E402
SCRIPT = SCRIPT.replace("__EXCHANGE_AB__", EXCHANGE_AB)


def test_oneshot_allreduce_two_ranks_on_one_device(tmp_path):
    script = tmp_path / "oneshot2.py"
    script.write_text(SCRIPT)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OSG_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "OSG_COMM"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print(rec)
    assert rec["f64_mismatches"] == 0 and rec["i32_mismatches"] == 0
    assert rec["chain_ok"] and rec["begin_end_ok"] and rec["too_long_refused"]
    for key in ("sync", "overlap"):
        assert rec[key]["rank_diff"] == 0.0, "every rank folds bit-identical sums"
        assert rec[key]["vs_one_rank"] < 1e-8 * max(1.0, rec[key]["regret_abs_sum"])
    ab = rec["exchange_ab"]   # the same deltas through torch.distributed (gloo here, RCCL in test_z7_*) and the one-shot kernel
    assert ab["trained"] and ab["routes_identical_minibatches"] == ab["mini_batches"] == 8 and ab["tables_identical"]
    assert ab["rank_diff"] == 0.0 and ab["max_err_vs_one_rank_over_scale"] <= 1e-10, ab
    assert 0.2 < rec["timeout_seconds"] < 5.0 and rec["timeout_reported"] and rec["timeout_reported_by_check"]
    assert rec["timeout_poisoned_buffer"], "a failed collective must leave NaN, not a plausible mix"
    assert rec["failed_creation_agreed"], "a rank that cannot create its window must fail the construction on every rank"


def test_oneshot_world_one_and_argument_checks():
    import ctypes as C
    import torch
    import open_spiel_amd as osa
    from open_spiel_amd import distributed as osd
    ctx = osa.Context(0)
    comm = osd.OneShotComm(ctx, 5616, rank=0, world_size=1)
    x = torch.arange(5616, dtype=torch.float64, device="cuda")
    comm.allreduce_sum_(x)
    ctx.synchronize()
    assert torch.equal(x, torch.arange(5616, dtype=torch.float64, device="cuda"))
    comm.close()
    lib = osa.lib()
    h = C.c_void_p()
    assert lib.osg_comm_oneshot_create(ctx._h, 0, 17, 16, C.byref(h)) != 0
    assert lib.osg_comm_oneshot_create(ctx._h, 2, 2, 16, C.byref(h)) != 0
    assert lib.osg_comm_oneshot_create(ctx._h, 0, 2, 1 << 20, C.byref(h)) != 0
    assert lib.osg_comm_oneshot_create(ctx._h, 0, 2, 64, C.byref(h)) == 0
    buf = torch.zeros(8, dtype=torch.float64, device="cuda")
    assert lib.osg_allreduce_sum_f64(h, C.c_void_p(buf.data_ptr()), 8) != 0   # not connected yet
    assert b"connect" in lib.osg_last_error()
    lib.osg_comm_destroy(h)
