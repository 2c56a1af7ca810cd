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
    of the search, whatever the number of roots."""
    needs_prior = True

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

    def __init__(self, model):
        self.model = model

    def evaluate(self, leaf, want_prior, want_value):
        if leaf.num_players != 2:
            raise OsgError("VPNetEvaluator assumes a two-player zero-sum game (vpevaluator.cc:74)")
        obs = leaf.observation_tensor(-1)
        legal = leaf.legal_actions_mask()[:, :leaf.num_distinct_actions].bool()
        with torch.no_grad():
            policy, value = self.model(obs, legal)
        policy = policy.to(torch.float64) * legal
        value = value.to(torch.float64).reshape(-1)
        return policy.contiguous(), torch.stack([value, -value], dim=1).contiguous()


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
           max_wall_clock_time=0.0, noise_generator=None, want_tree_of=None):
    """MCTSBot::MCTSearch (mcts.cc:353-467) for every root of `roots` with `evaluator` outside the kernel.
    Arguments as MCTSBot's constructor (mcts.h:161-169).  max_wall_clock_time > 0 stops the searches when that
    many seconds have passed (checked between evaluator rounds) instead of after max_simulations.
    Returns the dictionary of StateBatch.mcts_search plus child_prior [n, A]; want_tree_of=i adds "tree": root
    i's whole tree as arrays (meta, first_child, explore_count, total_reward, prior; include/osg_abi.h)."""
    ctx, n = roots.ctx, roots.n
    A, P = roots.num_distinct_actions, roots.num_players
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
            check(lib().osg_mcts_tree_advance(tree, leaf._h, None if prior is None else prior.data_ptr(),
                                              None if value is None else value.data_ptr(), request.data_ptr(),
                                              1 << 30, counts))
            prior = value = None
            if counts[1] == 0 and counts[2] == 0:
                break
            if max_wall_clock_time > 0 and time.perf_counter() - start >= max_wall_clock_time:
                break
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
    finally:
        lib().osg_mcts_tree_destroy(tree)
