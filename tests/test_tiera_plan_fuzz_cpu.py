"""Random machines through the tier-A plan on the CPU: the plan's tables, executed by the numpy restatement
of the kernel (test_tiera_plan_cpu._emulate), must reproduce the oracle's S and D lattice bit for bit."""
import numpy as np
import pytest

from random_machines import random_machine, random_read
from test_tiera_plan_cpu import _emulate


@pytest.mark.parametrize("seed,n_states,global_", [(1, 40, True), (2, 90, False), (3, 150, True), (4, 64, False), (5, 120, True), (6, 2300, True)])
def test_random_machine_plan_matches_oracle(oracle_mod, seed, n_states, global_):
    import dnastore_amd as da
    O = oracle_mod
    text = random_machine(seed, n_states)
    flags = dict(global_=global_, sub=.02, dup=.01, del_open=.02, del_ext=.1)
    fm = da.FlatModel(da.Machine.fromJSON(text), da.MutatorParams.fromFlags(**flags))
    assert fm.precompile().startswith("tier A")
    orc = O.ViterbiOracle(O.Machine.from_json(text), O.MutatorParams.from_cli(**flags))
    for r in range(2 if n_states < 1000 else 1):
        read = random_read(100 * seed + r, text, max_len=20 if n_states < 1000 else 10)
        _, _, olat = orc.decode(read, want_lattice=True)
        S_lat, D_lat = _emulate(fm, da.tokenize(read), local=not global_)
        assert np.array_equal(S_lat.view(np.uint64), np.ascontiguousarray(olat[:, :, 0]).view(np.uint64))
        assert np.array_equal(D_lat.view(np.uint64), np.ascontiguousarray(olat[:, :, 1]).view(np.uint64))
