"""Exit paths of a batch solve against the oracle's, by the rule SURVEY.md section 8(c) attaches to them: status, iteration,
backward-pass and rollout counts "must match except where the deciding margin is < 1e-9 relative (log those cases)".

The control flow of ILQR::solve is decided by comparisons of fp64 costs (ilqr.hh:66, :82: is_converged against rtol / atol;
:186: the Armijo inequality).  Two correct implementations whose costs agree to ~1e-13 relative can take different sides of
such a comparison only when it came out by less than that.  The oracle records every comparison of a solve with its margin --
the distance of the compared quantity from the value at which the comparison flips, as a cost difference over |cost|
(oracle/ilqr_oracle.c, orc_solve_decisions) -- and a problem whose counts differ from the oracle's is excused only if

  (1) the oracle's path holds a comparison with margin < BOUND, and
  (2) the solver's counts are what flipping THAT comparison gives: where the flip ends the solve (a convergence test the
      oracle failed by the margin), the solver's status, iteration, backward-pass and rollout counts must equal the
      oracle's counters at that comparison exactly; where the flip lets the solve go on (a convergence test the oracle
      passed by the margin, an Armijo test), the solver's counters must have moved on from there.

BOUND: on the |cost| scale every converging solve ends with comparisons inside the survey's 1e-9 (the differences compared
go to zero with the iteration), so 1e-9 would excuse anything.  The bound used is the level at which the two implementations
can actually differ: ten times the largest relative cost difference between solver and oracle over the problems of the same
batch whose paths agree (measured by the caller's own data), at least 1e-12 -- bench.py reports parity_max_rel_cost_err
~ 1.5e-13 at configs[1] -- and never more than the survey's 1e-9.  Every excused problem is printed with the comparison that
excuses it and the bound in force."""
import numpy as np

from oracle import oracle as orc

BOUND = 1e-12
KIND = {orc.DEC_EXPECTED: "expected-reduction convergence test (ilqr.hh:66)", orc.DEC_ARMIJO: "Armijo test (ilqr.hh:186)",
        orc.DEC_CONVERGED: "convergence test (ilqr.hh:82)"}


def _consistent(d, got):
    """is (status, iters, n_bwd, n_fwd) = got what a solve reports that agreed with the oracle up to comparison d and took the
    other side there?"""
    st, it, nb, nf = got
    on = it >= d["iter"] and nb >= d["n_bwd"] and nf >= d["n_fwd"]  # it has done everything the oracle had done by then
    if d["kind"] == orc.DEC_EXPECTED:
        if not d["result"]:  # the oracle went on; the other side stops here with status 0
            return (st, it, nb, nf) == (0, d["iter"], d["n_bwd"], d["n_fwd"])
        return on and nf > d["n_fwd"]  # the oracle stopped; the other side searches on: at least one more rollout
    if d["kind"] == orc.DEC_CONVERGED:
        if not d["result"]:  # the other side stops behind this accepted step with status 1
            return (st, it, nb, nf) == (1, d["iter"] + 1, d["n_bwd"], d["n_fwd"])
        return on and nb > d["n_bwd"] and it >= d["iter"] + 1  # ... goes on: at least one more backward pass
    # Armijo: accepted by the oracle, rejected by the other side -> at least one more rollout of this search, unless this was
    # the last trial it may take (status 3 with exactly these counters); rejected by the oracle, accepted by the other side ->
    # the iteration completes with this step
    if d["result"]:
        return on and (nf > d["n_fwd"] or (st == 3 and nf == d["n_fwd"]))
    return on and it >= d["iter"] + 1


def explain(got, r, bound=BOUND):
    """got = (status, iters, n_bwd, n_fwd) of a solver; r = OracleSolver.solve_decisions of the same problem.  None when the
    counts are the oracle's; otherwise the comparison of the oracle's path, margin < bound, whose other side gives these
    counts (the one with the smallest margin) -- or an AssertionError that lists what the oracle's path offers."""
    want = (r["status"], r["iters"], r["n_bwd"], r["n_fwd"])
    if tuple(got) == want:
        return None
    near = [d for d in r["decisions"] if d["margin"] < bound]
    hit = [d for d in near if _consistent(d, tuple(got))]
    assert hit, (f"(status, iters, n_bwd, n_fwd) = {tuple(got)}, oracle {want}; the oracle's comparisons with margin < {bound:g}: "
                 f"{near}; smallest margins of its path: {sorted(d['margin'] for d in r['decisions'])[:4]} -- no comparison within "
                 f"the bound explains these counts")
    return min(hit, key=lambda d: d["margin"])


def describe(b, got, r, d, label=""):
    want = (r["status"], r["iters"], r["n_bwd"], r["n_fwd"])
    return (f"[exit path]{' ' + label if label else ''} problem {b}: (status, iters, n_bwd, n_fwd) {tuple(got)} against the oracle's {want}: "
            f"{KIND[d['kind']]} of iteration {d['iter']}{', trial %d' % d['trial'] if d['kind'] == orc.DEC_ARMIJO else ''} "
            f"came out {d['result']} by a margin of {d['margin']:.2e} of the cost ({d['lhs']!r} against {d['rhs']!r})")


def batch_bound(out, ref, same):
    """ten times the solver's and the oracle's own cost disagreement on the problems of this batch that took the same path"""
    if "cost" not in out or "cost" not in ref or not np.any(same):
        return BOUND
    c, r = np.asarray(out["cost"])[same], np.asarray(ref["cost"])[same]
    ok = np.isfinite(c) & np.isfinite(r) & (np.abs(r) > 0)
    eps = float(np.max(np.abs(c[ok] - r[ok]) / np.abs(r[ok]))) if np.any(ok) else 0.0
    return min(1e-9, max(BOUND, 10.0 * eps))


def assert_same_exit_paths(out, ref, oracle, init, bound=None, label=""):
    """out / ref: results of solve_batch (the solver's and the oracle's) on init; oracle: the OracleSolver that made ref.
    Returns the indices of the excused problems (each printed with the comparison that excuses it)."""
    same = np.ones(len(ref["status"]), dtype=bool)
    for k in ("status", "iters", "n_bwd", "n_fwd"):
        same &= (np.asarray(out[k]) == np.asarray(ref[k]))
    if bound is None:
        bound = batch_bound(out, ref, same)
    excused = []
    for b in np.nonzero(~same)[0]:
        got = tuple(int(out[k][b]) for k in ("status", "iters", "n_bwd", "n_fwd"))
        r = oracle.solve_decisions(init[b])
        assert (r["status"], r["iters"], r["n_bwd"], r["n_fwd"]) == tuple(int(ref[k][b]) for k in ("status", "iters", "n_bwd", "n_fwd")), b
        try:
            d = explain(got, r, bound)
        except AssertionError as e:
            raise AssertionError(f"{label} problem {b}: {e}") from None
        print(describe(int(b), got, r, d, label) + f" [bound {bound:.1e}]")
        excused.append(int(b))
    return excused
