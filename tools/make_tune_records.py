"""Row-program verdicts for the fixture and bench machines (runtime.hip: tune_forwarded_rows), written to the directory
DNAS_KCACHE_DIR points at.  tools/make_tune_records.sh runs this on a GPU box and copies the records to dnastore_amd/tune/,
where the library finds them (csrc/jit.cpp: cacheNoteRead); the record names hash the kernel source, so they are made
again whenever csrc/viterbi_tiera.hip changes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DNAS_AUTOTUNE"] = "1"
os.environ.pop("DNAS_PLAN_FWD", None)
import dnastore_amd as da
G = os.path.join(ROOT, "tests", "golden", "ref_data")
machines = [(n, da.Machine.fromFile(os.path.join(G, n))) for n in ("l4c4.json", "mr2l4c4.json", "h74l4c4.json", "s16mr2l4c4.json", "s16h74l4c4.json")]
machines.append(("water64.1*l4c4", da.Machine.compose(da.Machine.fromFile(os.path.join(G, "water64.1.json")), da.Machine.fromFile(os.path.join(G, "l4c4.json")))))
for name, m in machines:
    dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True))
    print(name, m.nStates(), dec.tier[:40], flush=True)
    dec.close()
for f in sorted(os.listdir(os.environ["DNAS_KCACHE_DIR"])):
    if f.startswith("tune_"):
        print(f, open(os.path.join(os.environ["DNAS_KCACHE_DIR"], f)).read().strip())
