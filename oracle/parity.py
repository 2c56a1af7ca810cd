"""Parity checks at BASELINE.json's full sizes that the tests and bench.py's checker leg share.

TEST INFRASTRUCTURE ONLY (like the rest of oracle/): imported by tests/ and by bench.py's parity legs, never by
the product package.  `impl` is oracle_py (the restatement) or reference_py (the genuine reference build).
"""
import numpy as np

EPS = np.finfo(np.float64).eps


def mccfr_minibatch(impl, game_string, keys, nact, regrets_before, d_regrets, d_cum_policy, seed, first, count, threads,
                    table_scale=None):
    """One WHOLE mini-batch of the device's external-sampling MCCFR against the CPU's frozen-table replay
    (osgo_mccfr_frozen_replay: ExternalSamplingMCCFRSolver::UpdateRegrets, external_sampling_mccfr.cc:122-186, on the
    device's counter streams) — every regret / average-policy cell of every infostate.

    d_regrets / d_cum_policy: what the device added, [I, Amax] in the order of `keys`; regrets_before: the table the
    mini-batch read.  Tolerance per cell: 1e-11 x mass, mass = the sum of |increments| of that cell (the CPU returns it),
    plus — when the deltas were formed as table-after minus table-before (`table_scale` = [I, Amax] magnitudes of those
    tables) — 8 ulps of the tables.  Why 1e-11: both sides add the same ~10^5 terms per cell in different orders (the
    device: one fp64 LDS atomic per term inside a workgroup, one global atomic per workgroup and cell; the CPU: trajectory
    order per thread, threads in order); any order of N additions is within N eps x mass = 2^20 x 1.1e-16 = 1.2e-10 x mass
    of exact in the worst case and ~sqrt(N) eps = 1e-13 x mass typically.  A single lost, doubled or mis-addressed
    increment moves a cell by ~mass / N >= 1e-6 x mass: five orders above the tolerance.

    Returns a dict for the record; raises AssertionError with the worst cell on a mismatch."""
    g = impl.Game(game_string)
    want = impl.mccfr_frozen_replay(g, list(keys), regrets_before, seed, first, count, threads=threads)
    nact = np.asarray(nact)
    I, A = d_regrets.shape
    live = np.arange(A)[None, :] < nact[:, None]
    cum_mass = np.abs(want["d_cum_policy"])
    floor = 8 * EPS * (np.asarray(table_scale) if table_scale is not None else 0.0) + 1e-12
    worst, worst_mass = 0.0, 0.0
    for name, got, exp, mass in (("regret", d_regrets, want["d_regrets"], want["mass"]),
                                 ("average-policy", d_cum_policy, want["d_cum_policy"], cum_mass)):
        err = np.where(live, np.abs(got - exp), 0.0)
        tol = 1e-11 * mass + floor
        bad = err > tol
        if bad.any():
            i, a = np.unravel_index(np.argmax(err - tol), err.shape)
            raise AssertionError(f"{game_string} ES-MCCFR mini-batch [{first}, {first + count}): {name} increment of "
                                 f"{keys[i]!r} action {a}: device {got[i, a]!r} vs CPU {exp[i, a]!r} (mass {mass[i, a]:.6g}, "
                                 f"{int(bad.sum())} cells off)")
        worst = max(worst, float((err / tol)[live].max(initial=0.0)))
        big = live & (mass >= 1.0)                 # (cells with a mass worth the name; tiny ones are judged by the floor)
        worst_mass = max(worst_mass, float((err / np.maximum(mass, 1e-300))[big].max(initial=0.0)))
    # cells past a row's action count must be untouched
    assert not np.where(~live, np.abs(d_regrets) + np.abs(d_cum_policy), 0.0).any()
    return {"trajectories": int(count), "infostate_visits": int(want["visits"].sum()),
            "infostates_visited": int((want["visits"] > 0).sum()), "cells": int(2 * live.sum()),
            "max_error_over_mass": worst_mass, "tolerance_over_mass": 1e-11, "max_error_over_tolerance": worst}


def cfr_tables(impl, game_string, kind, iterations, keys, nact, regrets, cum_policy, avg_policy, rtol=1e-9, policy_atol=1e-6,
               solver=None):
    """The device's CFR tables after `iterations` EvaluateAndUpdatePolicy calls against the CPU solver run for the same
    number (cfr.cc:263-469): cumulative regrets and cumulative policy to `rtol` relative (the kernels perform the
    reference's additions in the reference's order, built without fused multiply-add: differences are the last ulps of
    libm-free arithmetic, observed <= 1e-12), the average policy to north_star's 1e-6 absolute."""
    if solver is None:
        g = impl.Game(game_string)
        s = impl.Solver(g, kind)
        s.iterate(iterations)
    else:
        s = solver                                 # (a CPU solver the caller has already run for `iterations`)
    t = s.tables(regrets.shape[1])
    idx = {k: i for i, k in enumerate(keys)}
    assert sorted(idx) == sorted(t["keys"]), "infostate sets differ"
    worst_tab, worst_pol = 0.0, 0.0
    for j, k in enumerate(t["keys"]):
        i, n = idx[k], int(t["nact"][j])
        assert n == int(nact[i]), k
        for name, got, exp in (("regrets", regrets[i, :n], t["regrets"][j, :n]),
                               ("cum_policy", cum_policy[i, :n], t["cum_policy"][j, :n])):
            scale = np.maximum(np.abs(exp), 1.0)   # (entries near zero are sums that cancelled: absolute there)
            e = float((np.abs(got - exp) / scale).max())
            worst_tab = max(worst_tab, e)
            if e > rtol:
                raise AssertionError(f"{game_string} {kind} after {iterations} iterations: {name} of {k!r}: device {got!r} vs CPU {exp!r}")
        e = float(np.abs(avg_policy[i, :n] - t["avg_policy"][j, :n]).max())
        worst_pol = max(worst_pol, e)
        if e > policy_atol:
            raise AssertionError(f"{game_string} {kind} after {iterations} iterations: average policy of {k!r} off by {e}")
    return {"iterations": int(iterations), "infostates": len(keys), "max_table_rel_error": worst_tab,
            "max_average_policy_abs_error": worst_pol, "table_rtol": rtol, "average_policy_atol": policy_atol}


def cfr_large_tree(impl, game_string, kinds, iterations, device, threads=2, best_response_player=0, rtol=1e-12):
    """A tree too large to afford many CPU iterations (3-player leduc: 1.83 M histories, 6.7 s per reference iteration),
    checked at size: for every solver kind ("cfr", "cfr_plus") the CPU solver — CFRSolver / CFRPlusSolver,
    cfr.cc:263-469 — runs `iterations` iterations (kinds in parallel host threads: the solvers are independent objects)
    and then judges its own average policy: ExpectedReturns (expected_returns.cc) and one player's
    TabularBestResponse value (best_response.cc:194-227), a term of NashConv (tabular_exploitability.cc:77-89).

    device[kind] = {"tables": TabularSolver.tables(), "expected_returns": [P], "best_response_values": [P]} after the
    same number of iterations.  Every regret / cumulative-policy cell of every infostate to `rtol` relative, the average
    policy to 1e-6, the two evaluations to 1e-11 absolute.  Returns the record; AssertionError on a mismatch."""
    import time
    from concurrent.futures import ThreadPoolExecutor

    def run(kind):
        t0 = time.perf_counter()
        g = impl.Game(game_string)
        s = impl.Solver(g, kind)
        s.iterate(iterations)
        t_it = time.perf_counter() - t0
        ev = s.expected_returns()
        br = s.best_response_value(best_response_player) if best_response_player is not None else None
        return s, g, ev, br, t_it, time.perf_counter() - t0

    with ThreadPoolExecutor(max(1, min(threads, len(kinds)))) as pool:
        done = dict(zip(kinds, pool.map(run, kinds)))
    out = {"game": game_string, "iterations": int(iterations), "kinds": list(kinds), "table_rtol": rtol}
    worst_tab = worst_pol = worst_ev = worst_br = 0.0
    for kind in kinds:
        s, _g, ev, br, t_it, t_all = done[kind]
        d = device[kind]
        t = d["tables"]
        rec = cfr_tables(impl, game_string, kind, iterations, t["keys"], t["nact"], t["regrets"], t["cum_policy"],
                         t["avg_policy"], rtol=rtol, solver=s)
        out["infostates"] = rec["infostates"]
        worst_tab = max(worst_tab, rec["max_table_rel_error"])
        worst_pol = max(worst_pol, rec["max_average_policy_abs_error"])
        e = float(np.abs(np.asarray(d["expected_returns"]) - ev).max())
        assert e <= 1e-11, f"{game_string} {kind}: expected returns {d['expected_returns']!r} vs CPU {ev!r}"
        worst_ev = max(worst_ev, e)
        if br is not None:
            e = abs(float(d["best_response_values"][best_response_player]) - br)
            assert e <= 1e-11, (f"{game_string} {kind}: best-response value of player {best_response_player} "
                                f"{d['best_response_values'][best_response_player]!r} vs CPU {br!r}")
            worst_br = max(worst_br, e)
        out[f"cpu_seconds_{kind}"] = t_all
        out[f"cpu_seconds_per_iteration_{kind}"] = t_it / iterations
    out.update(max_table_rel_error=worst_tab, max_average_policy_abs_error=worst_pol,
               max_expected_returns_abs_error=worst_ev, max_best_response_value_abs_error=worst_br,
               best_response_player=best_response_player)
    return out
