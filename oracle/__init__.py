"""CPU oracle for the AOC-Net matching / calibration hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may.

It restates, de-chunked and de-looped, the reference functions listed in
SURVEY.md section 8(a) (files under /root/reference/AOC-Net/...):

* ``oracle.kmeans``      -- scipy.cluster.vq.kmeans2 (third-party, scipy 1.15.3) in C,
                            bit-exact (pinned by tests/test_oracle_kmeans.py against scipy).
* ``oracle.matching``    -- adaptive_embedding_for_matching.py (AEM) a1-a9, a14.
* ``oracle.calibration`` -- attention.py:7-17,155-189 and conditioning_layer.py (a10-a13).

Pinning: the reference ships no tests or golden vectors, so the
oracle is pinned against outputs of the reference itself, generated in the build
container by ``tests/golden/make_golden.py`` (which imports the reference files by
path) and committed as ``tests/golden/*.npz``.  ``conditioning_block`` is not
executable in the reference: for that function parity is UNPINNED (see
``oracle/calibration.py``).
"""
