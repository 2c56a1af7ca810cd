"""The C-ABI collective (osg_comm_*, osg_allreduce_sum_*): RCCL resolved at run time, in place on the
context's stream.  One GPU per box here, so the device test is a world-size-1 communicator (RCCL's
all-reduce is then the identity); the N > 1 protocol itself — shard, all-reduce the delta tables, fold —
is covered by the world_size-2 gloo tests in tests/test_distributed_cpu.py."""
import ctypes as C

import numpy as np
import pytest


@pytest.mark.gpu
def test_world_size_one_allreduce_is_the_identity():
    import torch
    import open_spiel_amd as osa
    from open_spiel_amd._abi import OsgError
    lib = osa.lib()
    ctx = osa.Context(0)

    def check(rc):
        if rc != 0:
            raise OsgError(lib.osg_last_error().decode())

    uid = C.create_string_buffer(128)
    check(lib.osg_comm_unique_id(uid))
    comm = C.c_void_p()
    check(lib.osg_comm_create(ctx._h, 0, 1, uid, C.byref(comm)))
    try:
        assert lib.osg_comm_rank(comm) == 0 and lib.osg_comm_world(comm) == 1
        x = torch.arange(5000, dtype=torch.float64, device="cuda") * 0.25 - 7.0
        want = x.clone()
        check(lib.osg_allreduce_sum_f64(comm, C.c_void_p(x.data_ptr()), x.numel()))
        k = torch.arange(-300, 300, dtype=torch.int32, device="cuda")
        want_k = k.clone()
        check(lib.osg_allreduce_sum_i32(comm, C.c_void_p(k.data_ptr()), k.numel()))
        check(lib.osg_allreduce_sum_f64(comm, None, 0))       # empty buffers are fine
        ctx.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(x, want) and torch.equal(k, want_k)
        # the exchange step of a mini-batch, as a C++ host issues it: sample, all-reduce both delta
        # tables with one call, fold -> the same tables as the unsharded call
        a = osa.TabularSolver(ctx, "kuhn_poker", mccfr=True)
        b = osa.TabularSolver(ctx, "kuhn_poker", mccfr=True)
        a.run_mccfr(11, 4096)
        b.mccfr_sample(11, 4096, first_trajectory=0)
        flat = b.mccfr_delta_flat()
        check(lib.osg_allreduce_sum_f64(comm, C.c_void_p(flat.data_ptr()), flat.numel()))
        b.mccfr_apply_deltas()
        ta, tb = a.tables(), b.tables()
        # (fp64 atomics: the order of additions inside a launch is not fixed, hence a tolerance)
        np.testing.assert_allclose(ta["regrets"], tb["regrets"], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(ta["cum_policy"], tb["cum_policy"], rtol=1e-11, atol=1e-11)
    finally:
        check(lib.osg_comm_destroy(comm))


def test_comm_entry_points_reject_bad_arguments_without_a_gpu():
    import open_spiel_amd as osa
    lib = osa.lib()
    comm = C.c_void_p()
    uid = C.create_string_buffer(128)
    assert lib.osg_comm_create(None, 0, 1, uid, C.byref(comm)) != 0
    assert b"null" in lib.osg_last_error()
    assert lib.osg_allreduce_sum_f64(None, None, 4) != 0
    assert lib.osg_comm_rank(None) == -1 and lib.osg_comm_world(None) == -1
    assert lib.osg_comm_destroy(None) == 0


@pytest.mark.gpu
def test_torch_distributed_nccl_world_size_one_and_the_sharded_solver_path(tmp_path):
    """RCCL through the OTHER route the product uses: torch.distributed backend "nccl" (= RCCL on ROCm),
    initialised exactly as bench.py initialises it at N > 1, here with world size 1 — the process group comes
    up, open_spiel_amd.distributed's all-reduce of the leduc delta tables (2 x [936, 3] fp64 = 44 928 B, the
    ONE collective of the sharded ES-MCCFR mini-batch) runs on the device and leaves the tables of the
    unsharded call; the all-reduce and the C-ABI collective are timed on that buffer (printed with -s)."""
    import os
    import subprocess
    import sys
    script = tmp_path / "ws1.py"
    script.write_text('''
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.environ["OSG_ROOT"])
import numpy as np, torch, torch.distributed as dist
import open_spiel_amd as osa
from open_spiel_amd import distributed as osd
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
ctx = osa.Context(0)
a = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
b = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
a.run_mccfr(5, 1 << 14)
sh = osd.ShardedMccfr(b)
sh.run_minibatch(5, 1 << 14)
ta, tb = a.tables(), b.tables()
ok = bool(np.allclose(ta["regrets"], tb["regrets"], rtol=1e-11, atol=1e-11) and
          np.allclose(ta["cum_policy"], tb["cum_policy"], rtol=1e-11, atol=1e-11))
flat = b.mccfr_delta_flat()
t = torch.ones(8, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
for _ in range(20): dist.all_reduce(flat)  # world size 1: RCCL's all-reduce is the identity, but it IS issued
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): dist.all_reduce(flat)
torch.cuda.synchronize(); torch_us = (time.perf_counter() - t0) / 200 * 1e6
lib = osa.lib()
uid = C.create_string_buffer(128); assert lib.osg_comm_unique_id(uid) == 0
comm = C.c_void_p(); assert lib.osg_comm_create(ctx._h, 0, 1, uid, C.byref(comm)) == 0, lib.osg_last_error()
for _ in range(20): lib.osg_allreduce_sum_f64(comm, C.c_void_p(flat.data_ptr()), flat.numel())
ctx.synchronize(); t0 = time.perf_counter()
for _ in range(200): lib.osg_allreduce_sum_f64(comm, C.c_void_p(flat.data_ptr()), flat.numel())
ctx.synchronize(); abi_us = (time.perf_counter() - t0) / 200 * 1e6
lib.osg_comm_destroy(comm)
print(json.dumps({"ok": ok, "bytes": int(flat.numel() * 8), "torch_nccl_allreduce_us": torch_us, "osg_comm_allreduce_us": abi_us}))
dist.destroy_process_group()
''')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OSG_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    import json
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print("world-size-1 collectives on the 44 928-byte delta buffer:", rec)
    assert rec["ok"] and rec["bytes"] == 44928
    assert rec["torch_nccl_allreduce_us"] < 5000 and rec["osg_comm_allreduce_us"] < 5000
