"""A script fragment shared by the world-size-2 exchange tests (tests/test_z7_* over RCCL on two GPUs,
tests/test_z10_* over gloo + hipIpc windows with both ranks on ONE device): 8 sharded ES-MCCFR mini-batches whose
per-rank deltas travel through BOTH exchange routes — torch.distributed's all-reduce (RCCL over xGMI / gloo) and the
one-shot all-reduce (osg_comm_oneshot_*) — from the same inputs.

Expects in scope: rank, world, ctx, out (dict), osa, osd, torch, dist, np.  Leaves out["exchange_ab"].
  * routes_identical_minibatches: mini-batches whose two sums were bit-identical (world 2: a + b is one rounding whichever
    route adds it, so all 8; the one-shot kernel sums in rank order on every rank, a ring need not above two ranks);
  * tables_identical: the solver folded with route A's sums equals the one folded with route B's, bit for bit;
  * max_err_vs_one_rank_over_scale: per mini-batch, the all-reduced deltas against the SAME mini-batch sampled whole by
    one rank on the same frozen tables (what a world-1 job adds: only the fp64 summation order differs — the device adds
    ~10^4 terms per cell by atomics — so this is ~1e-13; a lost, doubled or mis-sharded trajectory moves a cell by
    >= 1e-6 of its mass);
  * rank_diff: every rank ends with the same tables.
"""
EXCHANGE_AB = r"""
mb, nb = 1 << 14, 8
s_a = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
s_b = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
buf, whole = s_a.mccfr_new_delta_buffer(), s_a.mccfr_new_delta_buffer()
one = osd.OneShotComm(ctx, buf.numel())
same_routes, worst = 0, 0.0
for k in range(nb):
    first, count = osd.shard_range(mb, rank, world)
    s_a.mccfr_sample_into(buf, 9, count, first_trajectory=k * mb + first)      # this rank's shard of mini-batch k
    s_a.mccfr_sample_into(whole, 9, mb, first_trajectory=k * mb)                # the whole mini-batch, same frozen tables
    a, b = buf.clone(), buf.clone()
    osd.allreduce_sum_(a)                      # torch.distributed: RCCL over xGMI on two GPUs
    one.allreduce_sum_(b)                      # the one-shot kernel over peer-mapped windows
    ctx.synchronize(); torch.cuda.synchronize()
    same_routes += int(torch.equal(a, b))
    scale = max(1.0, float(whole.abs().max()))
    worst = max(worst, float((a - whole).abs().max()) / scale, float((b - whole).abs().max()) / scale)
    s_a.mccfr_apply_deltas_from(a)
    s_b.mccfr_apply_deltas_from(b)
one.check()
ta, tb = s_a.tables(), s_b.tables()
mine = torch.from_numpy(np.stack([ta["regrets"], ta["cum_policy"]]))
if dist.get_backend() != "gloo":
    mine = mine.cuda()
parts = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(parts, mine)
out["exchange_ab"] = {
    "mini_batches": nb, "routes_identical_minibatches": same_routes,
    "tables_identical": bool(all(np.array_equal(ta[k], tb[k]) for k in ("regrets", "cum_policy", "cur_policy"))),
    "max_err_vs_one_rank_over_scale": worst,
    "rank_diff": float(max((p.cpu() - parts[0].cpu()).abs().max() for p in parts)),
    "backend": dist.get_backend(), "trained": bool(np.abs(ta["regrets"]).sum() > 0)}
one.close()
del s_a, s_b
"""
