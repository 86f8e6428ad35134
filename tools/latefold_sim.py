import os, sys, random
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tools")
import numpy as np
import dnastore_amd as da
from sweep_sim import Sim, load, plan_rows, NEG

def run(sim, seq, row_of, K, fold_at, max_cols, groups):
    """fold_at(g) -> sweep index (0 = at take) at which the history part W of the T1 hand-over of row group g is folded"""
    N, D = sim.N, sim.D
    a = sim.a
    mdl = a["mdl"].astype(np.int64); ctx = a["ctx"].astype(np.int64)
    rows = [np.nonzero(row_of == k)[0] for k in range(K)]
    S = np.full(N, NEG); S[0] = 0.0
    T = np.full((N, max(D, 1)), NEG)
    Sprev = None
    sweeps = []; raised = []
    for pos in range(0, min(len(seq), max_cols) + 1):
        W = None
        if pos > 0:
            x = seq[pos - 1]
            cand = S[sim.eS] + sim.eW + sim.noGap + sim.sub[sim.eB, x]
            Sn = cand.max(axis=1)
            has = mdl > 0
            # T[:,0] = max(Tshift0, S+tanDup+len0); early part uses S only
            early = np.where(has, (S + sim.tanDup + sim.len[0]) + sim.sub[ctx[:, 0], x], NEG)
            W = np.where(has, Tshift0 + sim.sub[ctx[:, 0], x], NEG) if pos > 1 else np.full(N, NEG)
            Sn = np.maximum(Sn, early)
            Tn = np.full_like(T, NEG)
            for k in range(D - 1):
                ok = (k < mdl - 1)
                Tn[:, k] = np.where(ok, T[:, k + 1] + sim.sub[ctx[:, k + 1], x], NEG)
            S, T = Sn, Tn
        Dl = np.full(N, NEG)
        n_sw = 0; nraised = 0
        Sfix = S.copy()
        X = np.maximum(Dl + sim.delExtend, S + sim.delOpen)
        force = np.ones(N, bool)
        while True:
            changed = False
            # fold groups scheduled at this sweep
            if W is not None:
                for g, ks in enumerate(groups):
                    if fold_at(g) == n_sw:
                        for k in ks:
                            r = rows[k]
                            up = W[r] > S[r]
                            if up.any():
                                nraised += int(up.sum())
                                S[r] = np.maximum(S[r], W[r]); changed = True
                                X[r] = np.maximum(Dl[r] + sim.delExtend, S[r] + sim.delOpen)
            n_sw += 1
            for k in range(K):
                r = rows[k]
                if not len(r): continue
                ge = (X[sim.eS[r]] + sim.eW[r]).max(axis=1)
                gd = (Dl[sim.nS[r]] + sim.nW[r]).max(axis=1)
                gs = (S[sim.nS[r]] + sim.nW[r]).max(axis=1)
                d = np.maximum(Dl[r], np.maximum(ge, gd))
                s = np.maximum(S[r], gs)
                s = np.maximum(s, d + sim.delEnd)
                ch = (s != S[r]) | (d != Dl[r])
                if ch.any():
                    changed = True
                    S[r] = s; Dl[r] = d
                    X[r] = np.maximum(d + sim.delExtend, s + sim.delOpen)
            pending = W is not None and any(fold_at(g) >= n_sw for g in range(len(groups)))
            if not changed and not pending:
                break
        sweeps.append(n_sw); raised.append(nraised)
        if pos > 0:
            Tshift0 = None
        # T update at end of column: T[:,k] = max(Tshift[:,k], S+tanDup+len[k]); remember shifted lane-1 for next column's W
        # next column needs Tshift0(next) = T(pos,1)+sub[ctx1][x_{pos+1}] -- computed at next column as Tn[:,0]; so W(next) = Tn[:,0]+...: handle via T arrays
        for k in range(D):
            ok = (k < mdl)
            T[:, k] = np.where(ok, np.maximum(T[:, k], S + sim.tanDup + sim.len[k]), T[:, k]) if pos > 0 else T[:, k]
        # precompute Tshift0 for next column: T(pos,1) + sub[ctx1][x_{pos+1}]
        if pos < len(seq):
            xn = seq[pos]
            Tshift0 = np.where(1 < mdl, T[:, 1] + sim.sub[ctx[:, 1], xn], NEG) if D > 1 else np.full(N, NEG)
    return np.array(sweeps), np.array(raised), S

if __name__ == "__main__":
    mp = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else ROOT + "/tests/golden/ref_data/s16h74l4c4.json"
    m, p, fm = load(mp)
    sim = Sim(fm)
    rng = random.Random(1000)
    dna = m.encodeBytes(bytes(rng.randrange(256) for _ in range(29)))
    seq = list(da.tokenize(dna))
    r2 = random.Random(5)
    for i in range(len(seq)):
        if r2.random() < 0.01: seq[i] = (seq[i] + 1 + r2.randrange(3)) & 3
    row_of, lane_of, T, K = plan_rows(fm)
    cols = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    g2 = [[2*i, 2*i+1] for i in range(K // 2)]
    g4 = [list(range(4*i, min(K, 4*i+4))) for i in range((K + 3) // 4)]
    ref = None
    for name, groups, fa in (("all at take (baseline)", [list(range(K))], lambda g: 0),
                             ("all at sweep 1", [list(range(K))], lambda g: 1),
                             ("all at sweep 2", [list(range(K))], lambda g: 2),
                             ("groups of 4 rows, group g at sweep g", g4, lambda g: g),
                             ("groups of 4 rows, group g at sweep g+1", g4, lambda g: g + 1),
                             ("groups of 2 rows, group g at sweep g", g2, lambda g: g),
                             ("groups of 2 rows, group g at sweep g+1", g2, lambda g: g + 1)):
        sw, ra, S = run(sim, seq, row_of, K, fa, cols, groups)
        if ref is None: ref = S
        print("%-45s sweeps/col mean %.2f max %d; raised per col %.1f; final S equal %s" % (name, sw[5:].mean(), sw.max(), ra[5:].mean(), np.array_equal(S, ref)), flush=True)
