"""The chunk-parallel exact float32 sums of the persistent k-means chain (csrc/km_exact_core.h), replayed on the CPU: the same
scalar functions the HIP kernel runs per lane fold / merge / stitch random chains (ReLU data, ties, zero prefixes, wide dynamic
range, signed values, NaN / inf, noisy predictions) and must reproduce the literal sequential sum bit for bit."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("kmcore") / "km_core_sim")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "csrc", "km_core_sim.cpp")], check=True)
    return exe


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_chunked_exact_sums_equal_the_sequential_sum(sim, seed):
    r = subprocess.run([sim, "450", str(seed)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "mismatches 0" in r.stdout


def test_typical_data_needs_no_slow_path(sim):
    # ReLU-like embeddings with exact predictions: records verify (a handful of failures per 100 chains at most)
    r = subprocess.run([sim, "90", "5", "0"], capture_output=True, text=True)
    assert r.returncode == 0
    fails = chunks = 0
    for line in r.stdout.splitlines():
        if line.startswith("kind relu") and "noise 0 " in line:
            t = line.split()
            chunks += int(t[t.index("chunks") + 1])
            fails += int(t[t.index("rec_fail") + 1]) + int(t[t.index("run_fail") + 1])
    assert chunks > 1000 and fails <= chunks // 500, (fails, chunks)
