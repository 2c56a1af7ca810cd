"""The same connect_four env step through the HOST-buffer side of the C-ABI (on_host = 1: what the per-state host
mirror uses): actions come from pageable host memory, the legal mask and the status come back to it — the
PCIe-inclusive rate DESIGN.md quotes beside the device-resident `value` (it is never `value`)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, open_spiel_amd as osa, bench
from open_spiel_amd._abi import check, lib
ctx = osa.Context(0)
n = 1 << 20
src, actions = bench.synth_batch(osa, torch, ctx, n, bench.SEED, 0)
acts = actions.cpu().numpy().astype(np.int32)
acts[acts == 255] = -1
mask = np.empty((n, src.desc.mask_words), np.uint32)
cur = np.empty(n, np.int8); term = np.empty(n, np.uint8); rets = np.empty((n, 2), np.float64)
work = src.clone()
L = lib()
def one():
    check(L.osg_batch_copy(work._h, src._h))                      # fresh positions (device to device)
    ill = C.c_int64(0)
    check(L.osg_apply(work._h, acts.ctypes.data, 1, C.byref(ill)))   # host actions in
    check(L.osg_legal_mask(work._h, mask.ctypes.data, 1))            # host mask out
    check(L.osg_status_query(work._h, cur.ctypes.data, term.ctypes.data, rets.ctypes.data, 1))  # host status out
for _ in range(3): one()
torch.cuda.synchronize(); t0 = time.perf_counter()
reps = 20
for _ in range(reps): one()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
moved = acts.nbytes + mask.nbytes + cur.nbytes + term.nbytes + rets.nbytes
print(f"connect_four 2^20 states, host buffers: {dt * 1e3:.2f} ms per step of the batch = {n / dt:.3e} env-steps/s "
      f"({moved / 1e6:.1f} MB over PCIe per step: {moved / dt / 1e9:.1f} GB/s; int32 actions in, u32 mask words, "
      f"player / terminal bytes and fp64 returns out, pageable host memory)")
