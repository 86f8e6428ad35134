"""The row-program tuning records that ship with the library (dnastore_amd/tune/, what bench.py, smoke() and the GPU tests follow):
every record belongs to one of the fixture / bench machines, is readable, and was measured with the kernel source the library
carries NOW.  After an edit of csrc/viterbi_tiera.hip this test fails until tools/make_tune_records.sh has been run on a GPU box and
its records copied here -- a stale record is still followed at run time (dnas_model_tier says so), but it is no longer a measurement
of the kernel that ships."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TUNE = os.path.join(ROOT, "dnastore_amd", "tune")


def _machines(da, ref_data):
    import bench
    out = [(n, da.Machine.fromFile(os.path.join(ref_data, n)), 1) for n in ("l4c4.json", "mr2l4c4.json", "h74l4c4.json", "s16mr2l4c4.json", "s16h74l4c4.json")]
    out.append(("water64.1*l4c4", bench.workload(da, 3, "a")["machine"], 1))
    out.append(("configs[1]", bench.workload(da, 1, "a")["machine"], 0))
    out.append(("configs[3] as written", bench.workload(da, 3, "b")["machine"], 0))
    return out


@pytest.fixture(scope="module")
def da():
    import dnastore_amd
    return dnastore_amd


def test_every_bench_machine_has_a_record_of_this_kernel(da, ref_data):
    params = da.MutatorParams.fromFlags(global_=True)
    want = {}
    for name, m, members in _machines(da, ref_data):
        want[da.FlatModel(m, params).tune_record_name(members)] = name
    have = sorted(f for f in os.listdir(TUNE) if f.startswith("tune_"))
    assert have == sorted(want), "dnastore_amd/tune/ holds records of other machines (or planner versions) than the fixture and bench ones"
    now = da.FlatModel.kernel_source_hash()
    stale = []
    for f in have:
        text = open(os.path.join(TUNE, f)).read()
        m = re.match(r"order=([012]) slack=([0-8]) kernel=(\S+) ", text)
        assert m, "%s: not a record: %r" % (f, text[:60])
        if m.group(3) != now:
            stale.append("%s (%s): measured with kernel %s" % (f, want[f], m.group(3)))
    assert not stale, "records measured with another kernel source than this library's (%s) -- run tools/make_tune_records.sh on a GPU box:\n  %s" % (now, "\n  ".join(stale))


def test_record_name_does_not_depend_on_the_error_model(da, ref_data):
    m = da.Machine.fromFile(os.path.join(ref_data, "s16h74l4c4.json"))
    a = da.FlatModel(m, da.MutatorParams.fromFlags(global_=True)).tune_record_name(1)
    b = da.FlatModel(m, da.MutatorParams.fromFlags(sub=.05, dup=.01)).tune_record_name(1)
    assert a == b and a != da.FlatModel(m, da.MutatorParams.fromFlags()).tune_record_name(0)
