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
