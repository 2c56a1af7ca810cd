"""Per-kernel throughput table (HIP events on the context's stream): the individual State
queries, the fused step of every game, tensor packing, random stepping and rollouts.

Byte model: the bytes a kernel must move — the state planes it actually reads (hex's LegalActions needs the two
stone planes and the meta word, 28 of the 52 bytes of a 9x9 state; its status query only the meta word) plus its
outputs.  At 2^20 states the small games move 4-40 MB per launch, i.e. a launch lasts about as long as an EMPTY
launch (2.6 us): their "fraction of 8 TB/s" at that size measures the launch floor, not the kernel, so the
step / tensor kernels are also timed at 2^24 states (rows tagged n=2^24), where DRAM is the bound."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
ctx = osa.Context(0)

def timeit(fn, iters=200, warm=20):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

rows = []
def report(name, secs, units, unit, nbytes=None):
    r = {"kernel": name, "us": secs * 1e6, "rate": units / secs, "unit": unit}
    if nbytes: r["GB/s"] = nbytes / secs / 1e9; r["frac_of_8TBs"] = nbytes / secs / 8e12
    rows.append(r); print(json.dumps(r), flush=True)

N = 1 << 20
for game, depth in [("connect_four", 12), ("tic_tac_toe", 3), ("hex(board_size=9)", 30), ("kuhn_poker", 2), ("leduc_poker", 4)]:
    b = osa.StateBatch(ctx, game, N); b.random_steps(3, depth)
    d = b.desc
    sb = d.state_words * d.state_word_bytes
    # planes a query reads: hex keeps 4 NW + 1 words (black, white, two edge-connection planes, meta), or 4 NW with the
    # meta word folded into the last word of each plane (then the meta bits come with those four words)
    hex_nw = d.state_words // 4 if game.startswith("hex") else 0
    hex_folded = bool(hex_nw) and d.state_words % 4 == 0
    sb_legal = (4 * (2 * hex_nw + (2 if hex_folded else 1))) if hex_nw else sb
    sb_status = (16 if hex_folded else 4) if hex_nw else sb
    dst = osa.StateBatch(ctx, game, N)
    mask, status = b.step_buffers()
    lm = b.legal_actions_mask()
    acts = torch.where(lm.any(1), lm.to(torch.float32).argmax(1), torch.full((N,), 255, device="cuda")).to(torch.uint8)
    s = timeit(lambda: b.step(acts, dst=dst, mask=mask, status=status))
    report(f"k_step {game}", s, N, "env-steps/s", N * (2 * sb + 1 + d.compact_mask_bytes + 1))
    bits = torch.empty((N, d.mask_words), dtype=torch.int32, device="cuda")
    s = timeit(lambda: osa._abi.check(osa.lib().osg_legal_mask(b._h, bits.data_ptr(), 0)))
    report(f"k_legal_mask {game}", s, N, "states/s", N * (sb_legal + 4 * d.mask_words))
    cur = torch.empty(N, dtype=torch.int8, device="cuda"); term = torch.empty(N, dtype=torch.uint8, device="cuda")
    rets = torch.empty((N, d.num_players), dtype=torch.float64, device="cuda")
    s = timeit(lambda: osa._abi.check(osa.lib().osg_status_query(b._h, cur.data_ptr(), term.data_ptr(), rets.data_ptr(), 0)))
    report(f"k_status {game}", s, N, "states/s", N * (sb_status + 2 + 8 * d.num_players))
    n_obs = N if d.obs_size <= 128 else N // 4
    bo = b if n_obs == N else b.gather(torch.arange(n_obs))
    out = torch.empty((n_obs, d.obs_size), dtype=torch.float32, device="cuda")
    s = timeit(lambda: bo.observation_tensor(0, out=out), iters=50, warm=5)
    report(f"k_observation {game} [{n_obs},{d.obs_size}]", s, n_obs, "states/s", n_obs * (sb + 4 * d.obs_size))
    if d.info_size:
        out = torch.empty((N, d.info_size), dtype=torch.float32, device="cuda")
        s = timeit(lambda: b.information_state_tensor(0, out=out), iters=50, warm=5)
        report(f"k_observation(info) {game} [{N},{d.info_size}]", s, N, "states/s", N * (sb + 4 * d.info_size))
    c = torch.zeros(2, dtype=torch.int64, device="cuda")
    s = timeit(lambda: b.random_steps(9, 32, counters=c), iters=20, warm=3)
    report(f"k_random_steps {game} (32 steps/launch)", s, N * 32, "env-steps/s")
    roots = osa.StateBatch(ctx, game, 1 << 16); roots.random_steps(5, depth)
    s = timeit(lambda: roots.rollout(1, 16), iters=10, warm=2)
    report(f"k_rollout {game} (2^16 roots x 16)", s, (1 << 16) * 16, "playouts/s")
    del b, dst, roots, mask, status, lm, acts, bits, cur, term, rets, out, bo
    if sb <= 16 or hex_nw:  # the same step / tensor kernels where the launch is long enough to be memory-bound
        NB = 1 << 24
        b = osa.StateBatch(ctx, game, NB); b.random_steps(3, depth)
        dst = osa.StateBatch(ctx, game, NB)
        mask, status = b.step_buffers()
        if hex_nw:   # a uniformly random empty cell per state (what every caller on the path plays; "the lowest
            # empty cell" is on the north edge rows for nearly every state, so nearly every step runs the edge-connection
            # flood, and the launch is ~10 % slower — profiles/r03_hex_step.log has both)
            lm = b.legal_actions_mask()
            acts = torch.where(lm.any(1), (lm.to(torch.float32) * torch.rand(lm.shape, device="cuda")).argmax(1),
                               torch.full((NB,), 255, device="cuda")).to(torch.uint8)
        else:
            lm = b.legal_actions_mask_bits()[:, 0]
            acts = torch.where(lm != 0, (torch.log2((lm & -lm).to(torch.float32))).to(torch.int32), torch.full((NB,), 255, device="cuda", dtype=torch.int32)).to(torch.uint8)
        s = timeit(lambda: b.step(acts, dst=dst, mask=mask, status=status), iters=50, warm=5)
        report(f"k_step {game} n=2^24", s, NB, "env-steps/s", NB * (2 * sb + 1 + d.compact_mask_bytes + 1))
        del dst, mask, status, lm, acts
        n_obs = NB if d.obs_size <= 128 else NB // 4     # hex(9): [2^22, 729] = 12 GB
        bo = b if n_obs == NB else b.gather(torch.arange(n_obs, device="cuda"))
        out = torch.empty((n_obs, d.obs_size), dtype=torch.float32, device="cuda")
        s = timeit(lambda: bo.observation_tensor(0, out=out), iters=20, warm=3)
        report(f"k_observation {game} [{n_obs},{d.obs_size}] n=2^24", s, n_obs, "states/s", n_obs * (sb + 4 * d.obs_size))
        del out, bo
        if d.info_size:
            out = torch.empty((NB, d.info_size), dtype=torch.float32, device="cuda")
            s = timeit(lambda: b.information_state_tensor(0, out=out), iters=20, warm=3)
            report(f"k_observation(info) {game} [{NB},{d.info_size}] n=2^24", s, NB, "states/s", NB * (sb + 4 * d.info_size))
            del out
        del b

# ---- the wide boards (round 6): the instantiations that hold a board of more than 128 cells in registers (hex 13 x 13:
# NW = 6, 19 x 19: NW = 12; connect_four above 64 board bits: unsigned __int128 planes) are compiled at 1-2 wavefronts per
# SIMD (257-307 VGPRs); this is what they deliver, at sizes where every byte goes through HBM (tag "wide") ----
for game, nwide, depth_mod in [("hex(board_size=13)", 1 << 22, 100), ("hex(board_size=19)", 1 << 21, 200),
                               ("hex(board_size=11)", 1 << 22, 80), ("connect_four(rows=8,columns=8)", 1 << 23, 40),
                               ("connect_four(rows=9,columns=12)", 1 << 23, 60)]:
    b = osa.StateBatch(ctx, game, nwide)
    d = b.desc
    sb = d.state_words * d.state_word_bytes
    dst = mask = status = acts = None
    if d.num_distinct_actions <= 255:     # (the fused step takes one-byte actions: hex 19 x 19 has 361)
        acts, _ = b.synth(7, depth_mod)
        dst = osa.StateBatch(ctx, game, nwide)
        mask, status = b.step_buffers()
        s = timeit(lambda: b.step(acts, dst=dst, mask=mask, status=status), iters=20, warm=3)
        report(f"k_step {game} wide n=2^{nwide.bit_length() - 1} ({sb} B record)", s, nwide, "env-steps/s",
               nwide * (2 * sb + 1 + d.compact_mask_bytes + 1))
    else:
        b.random_steps(7, depth_mod // 2)
        cur = torch.empty(nwide, dtype=torch.int8, device="cuda"); term = torch.empty(nwide, dtype=torch.uint8, device="cuda")
        rets = torch.empty((nwide, d.num_players), dtype=torch.float64, device="cuda")
        s = timeit(lambda: osa._abi.check(osa.lib().osg_status_query(b._h, cur.data_ptr(), term.data_ptr(), rets.data_ptr(), 0)), iters=20, warm=3)
        # (hex keeps mover / result in the meta bits: a status query reads the planes' last words — 16 B of the folded record)
        report(f"k_status {game} wide ({sb} B record; 16 B of it read)", s, nwide, "states/s", nwide * (16 + 2 + 8 * d.num_players))
    bits = torch.empty((nwide, d.mask_words), dtype=torch.int32, device="cuda")
    s = timeit(lambda: osa._abi.check(osa.lib().osg_legal_mask(b._h, bits.data_ptr(), 0)), iters=20, warm=3)
    report(f"k_legal_mask {game} wide", s, nwide, "states/s", nwide * (sb + 4 * d.mask_words))
    n_obs = max(1, (1 << 30) // (4 * d.obs_size))          # a 1 GiB tensor
    n_obs = min(n_obs, nwide)
    bo = b.gather(torch.arange(n_obs, device="cuda"))
    out = torch.empty((n_obs, d.obs_size), dtype=torch.float32, device="cuda")
    s = timeit(lambda: bo.observation_tensor(0, out=out), iters=10, warm=2)
    report(f"k_observation {game} wide [{n_obs},{d.obs_size}]", s, n_obs, "states/s", n_obs * (sb + 4 * d.obs_size))
    c = torch.zeros(2, dtype=torch.int64, device="cuda")
    small = b.gather(torch.arange(1 << 18, device="cuda"))
    s = timeit(lambda: small.random_steps(9, 32, counters=c), iters=5, warm=1)
    report(f"k_random_steps {game} wide (2^18 states x 32 steps)", s, (1 << 18) * 32, "env-steps/s")
    del b, dst, mask, status, bits, acts, bo, out, small
