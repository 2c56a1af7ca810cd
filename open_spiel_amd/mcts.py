"""MCTS with the Evaluator outside the kernel: batched, network-guided search.

The reference couples MCTSBot to an `Evaluator` (open_spiel/algorithms/mcts.h:83-92) and, for AlphaZero,
batches single-state inference requests from many search threads behind a queue
(open_spiel/algorithms/alpha_zero_torch/vpevaluator.{h,cc}: Evaluate -> {v, -v} from the network's player-0
value, Prior -> the network's policy over the legal actions, ChanceOutcomes at chance nodes).  Here the
batching is the data layout: every search of a `StateBatch` of roots advances on the device until it needs its
evaluator, the leaves of all searches sit in ONE batch of states, one forward pass over `observation_tensor`
of that batch answers them all, and the searches resume (osg_mcts_tree_* in include/osg_abi.h).

    evaluator = VPNetEvaluator(model)            # model: (obs [B, obs], legal [B, A]) -> (policy [B, A], value [B])
    result = search(roots, evaluator, max_simulations=800, uct_c=1.4, puct=True,
                    dirichlet_alpha=0.3, dirichlet_epsilon=0.25)
    result["best_action"], result["child_visits"], ...

`RolloutEvaluator` is RandomRolloutEvaluator on the device; with it `search` reproduces
`StateBatch.mcts_search(layout=1)` draw for draw (tests/test_z5_gpu_mcts_evaluator.py).
"""
import contextlib
import ctypes as C
import time

import torch

from . import _abi
from ._abi import OsgError, check, lib


class RolloutEvaluator:
    """RandomRolloutEvaluator(n_rollouts, seed) (mcts.h:97-111): uniform prior, mean of n_rollouts random
    playouts — both on the device (n_rollouts and seed are those of the search's configuration)."""
    needs_prior = False

    def values(self, tree, leaf, want):
        out = torch.empty((leaf.n, leaf.num_players), dtype=torch.float64, device=leaf.ctx.device)
        check(lib().osg_mcts_tree_rollout_values(tree, leaf._h, out.data_ptr()))
        return out


class BatchedEvaluator:
    """Base class of evaluators that run outside the kernel.  Override `evaluate`: it is called ONCE per round
    of the search, whatever the number of roots.

    joint = True declares that evaluate() always returns BOTH answers for every row (a policy / value network: one
    forward gives both, as alpha_zero_torch/vpevaluator.cc:60-85 caches them per state) and is a pure function of
    the leaf batch built from device-side torch ops (no host synchronisation, no Python-side branching on tensor
    values).  search() then keeps the prior that arrives with a leaf's value until the leaf is expanded (one evaluator
    round per simulation instead of up to two) and replays the whole round — search kernel, observation pack, legal
    mask, forward — as ONE captured graph without reading anything back to the host between rounds."""
    needs_prior = True
    joint = False

    def evaluate(self, leaf, want_prior, want_value):
        """leaf: StateBatch of the states the searches ask about; want_prior / want_value: [n] bool device
        tensors saying which rows are asked for what (other rows hold stale states: their answers are ignored).
        Returns (prior [n, A] float64 — probability of action a at [i, a]; illegal actions are ignored —,
        value [n, P] float64 — Evaluate()'s returns per player); either may be None when nothing of that kind
        was asked for."""
        raise NotImplementedError


class VPNetEvaluator(BatchedEvaluator):
    """alpha_zero_torch's VPNetEvaluator for a batch: `model(obs, legal)` maps the current player's observation
    tensors [B, obs_size] float32 and the legal-action mask [B, A] bool to (policy [B, A] over the legal actions,
    value [B] of player 0); Evaluate = {v, -v} (vpevaluator.cc:73-77, two-player zero-sum), Prior = the policy
    (:79-85).  One forward pass answers a prior request and a value request for the same row."""

    joint = True   # every forward answers prior AND value of the same states: search() keeps the priors (flag 8)

    def __init__(self, model):
        self.model = model

    def evaluate(self, leaf, want_prior, want_value):
        if leaf.num_players != 2:
            raise OsgError("VPNetEvaluator assumes a two-player zero-sum game (vpevaluator.cc:74)")
        obs = leaf.observation_tensor(-1)
        legal = leaf.legal_actions_bool()
        with torch.no_grad():
            policy, value = self.model(obs, legal)
        policy = policy.to(torch.float64) * legal
        value = value.to(torch.float64).reshape(-1, 1)
        return policy, torch.cat([value, -value], dim=1)


def dirichlet_noise(count, alpha, generator):
    """dirichlet_noise (mcts.cc:188-203): `count` Gamma(alpha, 1) draws, normalised."""
    g = torch.distributions.Gamma(torch.full((count,), float(alpha), dtype=torch.float64), 1.0)
    if generator is not None:
        state = torch.random.get_rng_state()
        torch.random.set_rng_state(generator.get_state())
        noise = g.sample()
        generator.set_state(torch.random.get_rng_state())
        torch.random.set_rng_state(state)
    else:
        noise = g.sample()
    return noise / noise.sum()


def search(roots, evaluator, max_simulations=1024, uct_c=2.0, n_rollouts=1, solve=False, max_nodes=0, seed=0,
           index_offset=0, puct=False, dirichlet_alpha=0.0, dirichlet_epsilon=0.0, dont_return_chance_node=False,
           max_wall_clock_time=0.0, noise_generator=None, want_tree_of=None, graph=None):
    """MCTSBot::MCTSearch (mcts.cc:353-467) for every root of `roots` with `evaluator` outside the kernel.
    Arguments as MCTSBot's constructor (mcts.h:161-169).  max_wall_clock_time > 0 stops the searches when that
    many seconds have passed (checked between evaluator rounds) instead of after max_simulations.
    Returns the dictionary of StateBatch.mcts_search plus child_prior [n, A]; want_tree_of=i adds "tree": root
    i's whole tree as arrays (meta, first_child, explore_count, total_reward, prior; include/osg_abi.h).
    graph: None = evaluators that declare joint = True run one evaluator round per simulation (nothing read back between
    rounds); False = always the request / answer loop below; True = additionally capture the round once and replay it as
    a graph (measured slower than plain launches on ROCm 7: 1.8 vs 0.73 ms per round at 2^16 roots — kept as an option)."""
    ctx, n = roots.ctx, roots.n
    A, P = roots.num_distinct_actions, roots.num_players
    if (getattr(evaluator, "joint", False) and graph is not False and dirichlet_alpha == 0 and max_wall_clock_time <= 0
            and not isinstance(evaluator, RolloutEvaluator)):
        return _search_joint(roots, evaluator, max_simulations, uct_c, n_rollouts, solve, max_nodes, seed, index_offset,
                             puct, dont_return_chance_node, want_tree_of, use_graph=graph is True)
    needs_prior = bool(getattr(evaluator, "needs_prior", True)) or dirichlet_alpha > 0
    flags = (1 if needs_prior else 0) | (2 if dont_return_chance_node else 0)
    # (with max_wall_clock_time the searches still stop at max_simulations: the tree slots are sized for it)
    cfg = _abi.MctsCfg(uct_c, int(max_simulations), n_rollouts, int(solve), max_nodes, seed, index_offset, 1,
                       1 if puct else 0)
    tree = C.c_void_p()
    check(lib().osg_mcts_tree_create(roots._h, C.byref(cfg), flags, C.byref(tree)))
    try:
        leaf = type(roots)(ctx, roots.game_string, n)
        request = torch.zeros(n, dtype=torch.uint8, device=ctx.device)
        counts = (C.c_int64 * 4)()
        prior = value = None
        start = time.perf_counter()
        while True:
            # (one new simulation per search and launch: see _search_joint)
            check(lib().osg_mcts_tree_advance(tree, leaf._h, None if prior is None else prior.data_ptr(),
                                              None if value is None else value.data_ptr(), request.data_ptr(),
                                              1, counts))
            prior = value = None
            if counts[1] == 0 and counts[2] == 0 and counts[3] == 0:
                break
            if max_wall_clock_time > 0 and time.perf_counter() - start >= max_wall_clock_time:
                break
            if counts[1] == 0 and counts[2] == 0:
                continue  # only paused searches are left: nothing to ask the evaluator, advance again
            want_prior, want_value = (request & 3) == 1, request == 2
            if isinstance(evaluator, RolloutEvaluator):
                if counts[1]:  # uniform prior; only asked for because of the root noise
                    legal = leaf.legal_actions_mask()[:, :A].to(torch.float64)
                    prior = legal / legal.sum(1, keepdim=True).clamp(min=1.0)
                if counts[2]:
                    value = evaluator.values(tree, leaf, want_value)
            else:
                prior, value = evaluator.evaluate(leaf, want_prior, want_value)  # ONE call (one forward) per round
            if counts[1]:
                if prior is None:
                    raise OsgError("the evaluator returned no prior although searches asked for one")
                prior = prior.to(torch.float64).contiguous()
                if prior.shape != (n, A):
                    raise OsgError(f"evaluator prior has shape {tuple(prior.shape)}, expected {(n, A)}")
                at_root = request == 5
                if dirichlet_alpha > 0 and bool(at_root.any()):
                    # the root's prior (mcts.cc:284-292): (1 - epsilon) * prior + epsilon * Dirichlet(alpha) noise
                    legal = leaf.legal_actions_mask()[:, :A].bool().cpu()
                    pr = prior.cpu()
                    for i in torch.nonzero(at_root.cpu()).flatten().tolist():
                        idx = torch.nonzero(legal[i]).flatten()
                        noise = dirichlet_noise(idx.numel(), dirichlet_alpha, noise_generator)
                        pr[i, idx] = (1 - dirichlet_epsilon) * pr[i, idx] + dirichlet_epsilon * noise
                    prior = pr.to(ctx.device).contiguous()
            else:
                prior = None
            if counts[2]:
                if value is None:
                    raise OsgError("the evaluator returned no value although searches asked for one")
                value = value.to(torch.float64).contiguous()
                if value.shape != (n, P):
                    raise OsgError(f"evaluator value has shape {tuple(value.shape)}, expected {(n, P)}")
            else:
                value = None
        return _results(tree, ctx, n, A, want_tree_of)
    finally:
        lib().osg_mcts_tree_destroy(tree)


def _answer(t, rows, cols):
    """An evaluator's answer as the kernel reads it: [rows, cols] float64, contiguous (no copy when it already is)."""
    if t.shape != (rows, cols):
        raise OsgError(f"evaluator answer has shape {tuple(t.shape)}, expected {(rows, cols)}")
    return t.to(torch.float64).contiguous()


def _results(tree, ctx, n, A, want_tree_of):
    out = {
        "best_action": torch.empty(n, dtype=torch.int32, device=ctx.device),
        "child_visits": torch.empty((n, A), dtype=torch.int32, device=ctx.device),
        "child_reward": torch.empty((n, A), dtype=torch.float64, device=ctx.device),
        "child_outcome": torch.empty((n, A), dtype=torch.int8, device=ctx.device),
        "child_prior": torch.empty((n, A), dtype=torch.float64, device=ctx.device),
        "root_stats": torch.empty((n, 4), dtype=torch.float64, device=ctx.device),
    }
    check(lib().osg_mcts_tree_results(tree, *[out[k].data_ptr() for k in
                                              ("best_action", "child_visits", "child_reward", "child_outcome",
                                               "child_prior", "root_stats")]))
    if want_tree_of is not None:
        import numpy as np
        used = lib().osg_mcts_tree_nodes(tree, int(want_tree_of))
        if used < 0:
            raise OsgError("no such root")
        arrs = {"meta": np.zeros(used, np.uint32), "first_child": np.zeros(used, np.uint32),
                "explore_count": np.zeros(used, np.uint32), "total_reward": np.zeros(used, np.float64),
                "prior": np.zeros(used, np.float64)}
        check(lib().osg_mcts_tree_download(tree, int(want_tree_of), used, *[a.ctypes.data for a in arrs.values()]))
        out["tree"] = arrs
    ctx.synchronize()
    return out


def _search_joint(roots, evaluator, max_simulations, uct_c, n_rollouts, solve, max_nodes, seed, index_offset, puct,
                  dont_return_chance_node, want_tree_of, use_graph):
    """search() for an evaluator that answers prior and value together (BatchedEvaluator.joint): the prior that comes
    with a leaf's value is kept on the device until the leaf is expanded (osg_mcts_tree_create flag 8), so every
    simulation is ONE round — advance the searches, pack the leaves' observations and legal masks, one forward — and
    nothing is read back between rounds: after max_simulations + 1 rounds every search has finished (a search whose
    nodes a garbage collection cleared may ask for a prior again and take a few rounds more: the request counts are
    read once at the end and the loop goes on while any search is unfinished).  The round is captured once as a graph
    on a side stream and replayed."""
    ctx, n = roots.ctx, roots.n
    A, P = roots.num_distinct_actions, roots.num_players
    flags = 1 | 8 | (2 if dont_return_chance_node else 0)
    cfg = _abi.MctsCfg(uct_c, int(max_simulations), n_rollouts, int(solve), max_nodes, seed, index_offset, 1,
                       1 if puct else 0)
    tree = C.c_void_p()
    check(lib().osg_mcts_tree_create(roots._h, C.byref(cfg), flags, C.byref(tree)))
    side = torch.cuda.Stream(device=ctx.device) if use_graph else None
    bound_before = None
    try:
        leaf = type(roots)(ctx, roots.game_string, n)
        request = torch.zeros(n, dtype=torch.uint8, device=ctx.device)
        prior = torch.zeros((n, A), dtype=torch.float64, device=ctx.device)
        value = torch.zeros((n, P), dtype=torch.float64, device=ctx.device)
        everything = torch.ones(n, dtype=torch.bool, device=ctx.device)

        # At most ONE new simulation per search and launch.  A simulation that ends on a terminal node needs no
        # evaluator, so an uncapped search goes straight on to the next one — and a root with an immediate win among
        # its children runs dozens of simulations in one launch while the other 63 lanes of its wavefront wait
        # (measured, 2^16 connect_four roots: launches of 170-670 us where capped ones take 60-80 us; some roots ran
        # 27 simulations in one launch).  Capped, every launch is one simulation per search and the rounds stay in
        # step; searches paused by the cap (request 3) simply start their next simulation in the next launch.
        answers = [prior, value]   # the tensors the next advance reads (kept alive here)

        def one_round():
            check(lib().osg_mcts_tree_advance(tree, leaf._h, answers[0].data_ptr(), answers[1].data_ptr(),
                                              request.data_ptr(), 1, None))
            pr, va = evaluator.evaluate(leaf, everything, everything)
            if use_graph:   # a replayed graph reads fixed addresses
                prior.copy_(pr.reshape(n, A))
                value.copy_(va.reshape(n, P))
            else:
                answers[0], answers[1] = _answer(pr, n, A), _answer(va, n, P)

        rounds = int(max_simulations)   # + the advances of the loop below: the last answers, searches the cap held back
        graph = None
        if use_graph:
            ctx.synchronize()
            side.wait_stream(torch.cuda.current_stream(ctx.device))
            bound_before = ctx.set_stream(side)
            with torch.cuda.stream(side):
                one_round()                      # warm-up on the side stream (it is round 1 of the search)
                side.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    one_round()
                for _ in range(rounds - 1):
                    graph.replay()
        else:
            for _ in range(rounds):
                one_round()
        counts = (C.c_int64 * 4)()
        spare = 4 * rounds + 16
        # every planned round ended with an evaluation: one more advance consumes it; searches that still want an
        # answer afterwards (only after a garbage collection cleared kept priors) go on, request counts read each time
        stream_ctx = torch.cuda.stream(side) if use_graph else contextlib.nullcontext()
        with stream_ctx:
            while True:
                check(lib().osg_mcts_tree_advance(tree, leaf._h, answers[0].data_ptr(), answers[1].data_ptr(),
                                                  request.data_ptr(), 1, counts))
                if counts[1] == 0 and counts[2] == 0 and counts[3] == 0:
                    break
                spare -= 1
                if spare < 0:
                    raise OsgError("joint search did not finish")
                pr, va = evaluator.evaluate(leaf, everything, everything)
                answers[0], answers[1] = _answer(pr, n, A), _answer(va, n, P)
        if use_graph:
            with torch.cuda.stream(side):
                out = _results(tree, ctx, n, A, want_tree_of)
            side.synchronize()
            return out
        return _results(tree, ctx, n, A, want_tree_of)
    finally:
        if bound_before is not None:
            ctx.synchronize()
            ctx.set_stream(bound_before)
        lib().osg_mcts_tree_destroy(tree)
