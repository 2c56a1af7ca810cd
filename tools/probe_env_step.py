"""osg_env_step on 2^20 connect_four environments: the kernel alone (HIP events around the launch, a table-lookup agent in
between so that the environments keep playing), by form — two environments per thread (the default) and
OSG_ENV_STEP_X1=1 (one per thread)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import open_spiel_amd as osa
from open_spiel_amd._abi import check, lib
ctx = osa.Context(0)
n = 1 << int(os.environ.get("PROBE_LOG_N", "20"))
want = None
for form in ("x2", "x1"):
    os.environ.pop("OSG_ENV_STEP_X1", None)
    os.environ["OSG_ENV_STEP_FUSED"] = "0"
    if form == "x1": os.environ["OSG_ENV_STEP_X1"] = "1"
    eb = osa.StateBatch(ctx, "connect_four", n)
    reset = torch.ones(n, dtype=torch.uint8, device="cuda"); cur = torch.empty(n, dtype=torch.int8, device="cuda")
    typ = torch.empty(n, dtype=torch.uint8, device="cuda"); rew = torch.empty((n, 2), dtype=torch.float64, device="cuda")
    msk = torch.empty((n, 1), dtype=torch.int32, device="cuda"); acts = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    lut = torch.tensor([-1] + [(m & -m).bit_length() - 1 for m in range(1, 128)], dtype=torch.int32, device="cuda")
    def step(t):
        check(lib().osg_env_step(eb._h, acts.data_ptr(), reset.data_ptr(), 1, 0, t, cur.data_ptr(), typ.data_ptr(), rew.data_ptr(), msk.data_ptr()))
    checksum = 0
    for t in range(10):
        step(t); torch.index_select(lut, 0, msk[:, 0].to(torch.int64), out=acts)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for t in range(10, 210):
        e0.record(); step(t); e1.record(); torch.index_select(lut, 0, msk[:, 0].to(torch.int64), out=acts)
        torch.cuda.synchronize(); tot += e0.elapsed_time(e1)
        if t % 50 == 0:
            checksum += int(rew.sum().item() * 7 + cur.to(torch.int64).sum().item() * 3 + typ.to(torch.int64).sum().item() + msk.to(torch.int64).sum().item() + reset.to(torch.int64).sum().item())
    us = tot / 200 * 1e3
    want = want if want is not None else checksum
    print(f"k_env_step {form:8s}: {us:6.2f} us per 2^{n.bit_length() - 1} environments = {60 * n / us / 1e6:.2f} TB/s "
          f"({60 * n / us / 1e6 / 8:.3f} of 8 TB/s, 60 B per step)  {'same outputs' if checksum == want else 'DIFFERENT OUTPUTS'}", flush=True)
