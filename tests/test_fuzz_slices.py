"""Seeded slices of the two fuzzers, so that every `pytest -m gpu` run -- the driver's included -- sees them:

  tools/fuzz_session.py   the captured decode loop (deft_amd.DecodeSession: legacy sessions bit for bit, window-plan sessions within
                          the operator's tolerance) against the eager path, step for step, through random trees, cuts, branches,
                          speculative-decoding merge / reset steps, head_dim 64, node_chunk, lagged (host-runs-ahead) runs
  tools/fuzz_replay.py    random replays (templates, geometries, modes, sessions and eager calls) against fp64 attention of every
                          leaf over its own path, at every step

Each is a script; the slice runs it as the builder does (a child process, ~25 s of fuzzing) and checks its verdict line."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _run(script, *args):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), *map(str, args)], capture_output=True, text=True,
                       timeout=420, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout.strip().splitlines()[-1]


@pytest.mark.parametrize("seed,inc", [(31, "mix"), (7, "1")])
def test_session_fuzzer_slice(seed, inc):
    line = _run("fuzz_session.py", 25, seed, inc)
    assert line.startswith("session fuzz ok:"), line
    trees, steps = int(line.split()[3]), int(line.split("),")[1].split()[0])
    assert trees >= 3 and steps >= 100, line  # (a slice that fuzzed nothing proves nothing)


def test_replay_fuzzer_slice():
    line = _run("fuzz_replay.py", 25, 31)
    assert line.startswith("fuzz ok:"), line
    assert int(line.split()[2]) >= 3 and "none beyond 1e-3" in line, line
