"""How many rounds the refinement needs under different rules for what a round prepares.

Input: the order of the reference's refinement loop for one pair, one line per
iteration: `iteration k stays improved_k ...` (k = entry refined, stays = 1 if its
alignment changed and it remains in the set, then every entry whose sum improved).
The checker's scalar refinement writes that file when ORC_REFINE_LOG=<path> is
set (oracle/minrms_oracle.c, orc_refine_clip); nothing of the checker is loaded here.

A round prepares the DP tables of a set P of entries from the state at its start;
the replay then follows the loop while its next entry is in P and has not been
improved since the round began (its transform, hence its table, would be another).
  current : P = the next `spec` entries the loop visits if nothing changes any more
            (repeated median), spec adapting between 4 and 12 as the engine does
  all     : P = every entry still to refine -- a lower bound: only true dependencies end rounds

  pipelined: as `current`, and while a round's tracebacks and its replay run (T), the distance tables and
            the DP (D) of the entries that FOLLOW its plan in the same order are computed ahead, from the
            state at the round's start; the next round uses them (and costs T only) if its first entry is
            among them and has not been improved meanwhile, else it starts over (D + T)

usage: python tools/refine_order_sim.py <log> [first_k]
"""
import sys


def median(S):
    s = sorted(S)
    return s[(len(s) + 1) // 2 - 1]


def plan_current(S, spec):
    S = set(S)
    P = []
    while S and len(P) < spec:
        k = median(S)
        P.append(k)
        S.discard(k)
    return set(P)


def plan_all(S, spec):
    return set(S)


def simulate(log, first_k, range_n, plan, spec, adapt, spec_min=4, spec_max=12):
    S = set(range(first_k, range_n))
    i = rounds = tables = 0
    while i < len(log):
        P = plan(S, spec)
        tables += len(P)
        rounds += 1
        stale, applied = set(), 0
        while i < len(log):
            k, stays, improved = log[i]
            assert k == median(S), "log and set out of step at iteration %d" % (i + 1)
            if k not in P or k in stale:
                break
            for q in improved:
                stale.add(q)
                if q != k:
                    S.add(q)
            (S.add if stays else S.discard)(k)
            i += 1
            applied += 1
        if adapt:
            if applied >= len(P):
                spec = min(spec_max, spec * 2)
            elif 4 * applied < len(P):
                spec = max(spec_min, spec // 2)
    return rounds, tables


def simulate_pipelined(log, first_k, range_n, spec, d_us, t_us, spec_min=4, spec_max=12):
    """(rounds, rounds that found their tables ready, microseconds of the chain)."""
    S = set(range(first_k, range_n))
    i = rounds = hits = 0
    us = 0.0
    ahead, ahead_stale = set(), set()          # tables computed ahead during the last round, and what that round improved
    while i < len(log):
        k0 = median(S)
        if k0 in ahead and k0 not in ahead_stale:
            P, base_stale, cost = set(ahead), set(ahead_stale), t_us      # the tables are there: tracebacks and replay only
            hits += 1
        else:
            P, base_stale, cost = plan_current(S, spec), set(), d_us + t_us
        # what is computed ahead while this round's tracebacks run: the entries after P in the same order
        rest = set(S) - P
        nxt = plan_current(rest, spec) if rest else set()
        rounds += 1
        us += cost
        stale, applied = set(base_stale), 0
        while i < len(log):
            k, stays, improved = log[i]
            if k not in P or k in stale:
                break
            for q in improved:
                stale.add(q)
                if q != k:
                    S.add(q)
            (S.add if stays else S.discard)(k)
            i += 1
            applied += 1
        ahead, ahead_stale = nxt, set(stale)
        if applied >= len(P):
            spec = min(spec_max, spec * 2)
        elif 4 * applied < len(P):
            spec = max(spec_min, spec // 2)
    return rounds, hits, us


if __name__ == "__main__":
    log = []
    for line in open(sys.argv[1]):
        a = list(map(int, line.split()))
        log.append((a[1], a[2], a[3:]))
    first_k = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # entry 0 (n = 1) has no alignment to refine
    range_n = 1 + max(max([k] + imp) for k, _, imp in log)
    print("%d iterations" % len(log))
    print("as the engine plans (4..12 adaptive): %d rounds, %d tables" % simulate(log, first_k, range_n, plan_current, 12, True))
    for spec in (4, 8, 12, 16, 32):
        print("fixed %2d per round: %d rounds, %d tables" % ((spec,) + simulate(log, first_k, range_n, plan_current, spec, False)))
    print("every entry still to refine: %d rounds, %d tables" % simulate(log, first_k, range_n, plan_all, 0, False))
    # measured at 150x150 (tools/kernel_times.py): distance tables + DP 71 us, tracebacks + replay 55 us per round
    d_us, t_us = 71.0, 55.0
    r0, _ = simulate(log, first_k, range_n, plan_current, 12, True)
    r1, hits, us = simulate_pipelined(log, first_k, range_n, 12, d_us, t_us)
    print("chain as the engine runs it: %d rounds x (D + T) = %.2f ms" % (r0, r0 * (d_us + t_us) * 1e-3))
    print("DP of the following entries computed ahead during the tracebacks: %d rounds, %d of them find their tables "
          "ready: %.2f ms (%.1f %% shorter)" % (r1, hits, us * 1e-3, 100.0 * (1.0 - us / (r0 * (d_us + t_us)))))
    r_all, _ = simulate(log, first_k, range_n, plan_all, 0, False)
    print("lower bound of any such rule (every round ends at an entry the round itself changed): %d rounds x (D + T) = %.2f ms"
          % (r_all, r_all * (d_us + t_us) * 1e-3))
