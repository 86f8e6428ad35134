"""Thread-per-read traceback alone on the GPU (behind a single fill launch): milliseconds per batch with and without node records.
    python tools/tb_probe.py [reads]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dnastore_amd as da
n = int(sys.argv[1]) if len(sys.argv) > 1 else 720
m = da.Machine.fromFile(os.path.join(ROOT, "tests", "golden", "ref_data", "s16h74l4c4.json"))
reads = []
for i in range(n):
    rng = random.Random(1000 + i)
    dna = list(m.encodeBytes(bytes(rng.randrange(256) for _ in range(29))))
    for j in range(len(dna)):
        if rng.random() < 0.01: dna[j] = rng.choice([b for b in "ACGT" if b != dna[j]])
    reads.append("".join(dna))
want = None
for label, env, opt in (("wave per read", {}, None), ("thread per read, node records", {}, "traceback=thread"), ("thread per read, CSR arrays", {"DNAS_NO_NODE_RECORDS": "1"}, "traceback=thread")):
    os.environ.pop("DNAS_NO_NODE_RECORDS", None)
    os.environ.update(env)
    dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True), options=opt)
    dec.decode(reads[:8])
    for it in range(2):
        out = dec.decode(reads)
    s = dec.stats()
    if want is None: want = out[0]
    print("%-34s traceback %.2f ms for %d reads (fill %.1f ms); strings equal: %s" % (label, s["traceback_ms"], n, s["fill_ms"], out[0] == want), flush=True)
    dec.close()
