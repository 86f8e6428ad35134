"""The general fill kernel (tier B, any machine size) held to the same bit-exact parity as tier A:
DNAS_TIER=B forces it for machines that would otherwise take the specialised kernel."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mach,fa,flags", [
    ("l4c4.json", "hello.dup.fa", dict(sub=0., del_open=0., global_=True)),
    ("h74l4c4.json", "hello.h74.sub.fa", dict()),
    ("s16h74l4c4.json", "hello.s16h74.del.fa", dict()),
])
def test_tier_b_full_lattice_bit_exact(oracle_mod, ref_data, mach, fa, flags, monkeypatch):
    import dnastore_amd as da
    O = oracle_mod
    monkeypatch.setenv("DNAS_TIER", "B")
    path = os.path.join(ref_data, mach)
    dec = da.ViterbiDecoder(da.Machine.fromFile(path), da.MutatorParams.fromFlags(**flags))
    assert dec.tier.startswith("tier B")
    orc = O.ViterbiOracle(O.Machine.from_file(path), O.MutatorParams.from_cli(**flags))
    read = da.read_fastseqs(os.path.join(ref_data, fa))[0][1]
    out, ll, st = dec.decode([read])
    s, oll, olat = orc.decode(read, want_lattice=True)      # [L+1][N][lanes]
    lat = dec.lattice(0, len(read))                          # [L+1][lanes][N]
    assert out[0] == s and ll[0] == oll and st[0] == 0
    a = np.ascontiguousarray(lat.transpose(0, 2, 1))
    assert np.array_equal(a.view(np.uint64), olat.view(np.uint64))
    dec.close()


def test_tier_a_failure_is_loud_when_forced(ref_data, monkeypatch):
    """DNAS_TIER=A turns a tier-A specialisation failure into an error instead of a silent fallback."""
    import dnastore_amd as da
    monkeypatch.setenv("DNAS_TIER", "A")
    monkeypatch.setenv("DNAS_TIERA_DEFS", "-DDNAS_K=this_does_not_compile")
    m = da.Machine.fromFile(os.path.join(ref_data, "l4c4.json"))
    with pytest.raises(Exception):
        da.ViterbiDecoder(m, da.MutatorParams.fromFlags(sub=.0123))
