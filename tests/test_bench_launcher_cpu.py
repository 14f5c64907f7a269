"""`bench.py --gpus N` with no rendezvous in the environment starts its N ranks
itself.  Driven here on the CPU: gloo, the oracle standing in for the engine,
two ranks.  The JSON line must count the ranks that took part, carry the phase
split, and a rank that fails must fail the run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FACTORY = "oracle_engine:make_for_bench"


def _run(extra, env_extra=None):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), ROOT, os.environ.get("PYTHONPATH", "")]))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--engine-factory", FACTORY,
                           "--n1", "30", "--n2", "26", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"] + extra,
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)


def _line(p):
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.stdout, p.stderr)
    return json.loads(lines[0])


def test_launcher_starts_the_ranks_and_counts_them():
    one = _line(_run(["--gpus", "1"]))
    two = _run(["--gpus", "2", "--extra-n", "24"])
    assert two.returncode == 0, two.stderr
    two = _line(two)
    assert (one["n_gpus"], two["n_gpus"]) == (1, 2)
    assert one["config"]["orientations"] == two["config"]["orientations"] == 27 * 23
    assert one["config"]["refine_iterations"] == two["config"]["refine_iterations"]
    for d in (one, two):
        assert set(d["phases_ms"]) >= {"generate", "sweep", "all_reduce", "winners", "refine"}
        assert abs(sum(d["phases_ms"].values()) - d["ms_per_step"]) < 0.5 * d["ms_per_step"] + 5.0
        assert d["scaling"] == "strong" and d["unit"] == "DP-cells/s" and d["value"] > 0
    # the second pair's sweep is timed with more than one rank
    assert "extra" not in one and two["extra"]["orientations"] == 21 * 21 and two["extra"]["value"] > 0


def test_a_rank_that_fails_fails_the_run():
    p = _run(["--gpus", "2", "--extra-n", "0"], {"ORACLE_ENGINE_FAIL_RANK": "1"})
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_world_size_that_contradicts_gpus_is_refused():
    p = _run(["--gpus", "4"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_a_group_of_one_runs_the_collectives_and_changes_nothing():
    one = _line(_run(["--gpus", "1"]))
    p = _run(["--gpus", "1", "--group-of-one", "--extra-n", "24"])
    assert p.returncode == 0, p.stderr
    grp = _line(p)
    assert grp["n_gpus"] == 1 and grp["config"]["parallelism"] == "1 GPU"
    for k in ("orientations", "refine_iterations", "refine_dp_runs_executed", "cells_per_orientation"):
        assert grp["config"][k] == one["config"][k]
    assert grp["extra"]["orientations"] == 21 * 21
