"""Two processes on one card, both decoding on clusters of work-groups (tier C) at the same time.

The members of a cluster wait for each other, so a cluster launch wants all its work-groups resident; a launch's members
are neighbours in dispatch order, so whatever else holds CUs -- here: the other process's cluster launch -- delays whole
clusters, never half of one for long ("Residency").  What this checks: both processes get the results a
process alone gets, call after call, and no fill launch waits anywhere near the watchdog times.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


def test_two_processes_decode_on_clusters_at_once():
    import dnastore_amd as da
    from two_cluster_processes import run_pair
    from two_clusters_child import digest, reads_and_model
    options = "tier=C,cluster=2,arena_fraction=0.3"
    a, b = run_pair(options, calls=10)
    m, params, reads = reads_and_model(da)
    lone = da.ViterbiDecoder(m, params, options=options)
    strings, ll = lone.decode(reads)[:2]
    lone.close()
    want = [digest(strings, ll)]
    what = "\n".join(repr(r) for r in (a, b))
    assert a["tier"].startswith("tier C") and b["tier"].startswith("tier C")
    assert a["digests"] == want and b["digests"] == want, what
    # the calls did overlap (each process was still calling when the other started) ...
    assert a["t_first"] < b["t_last"] and b["t_first"] < a["t_last"], what
    # ... and a launch beside another one takes its turn at the card, not the watchdog's seconds (alone: about 60 ms; the
    # bounds are loose on purpose -- another tenant of the host's driver can hold any call up for a second or two)
    assert max(a["fill_ms"] + b["fill_ms"]) < 10000.0, what
    assert np.median(a["walls_s"] + b["walls_s"]) < 5.0, what
