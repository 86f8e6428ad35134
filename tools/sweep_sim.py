"""Offline model of the tier-A in-column fixpoint: how many sweeps does a row layout need?

Runs the column recurrence of the fill (same relaxations as the kernel, numpy, CPU only) on one
synthetic read under the kernel's schedule model -- every thread walks its rows 0..K-1, all
threads in step, the gathers of row k issued `pipe-1` rows early -- and counts the sweeps each
column takes to reach the fixpoint.  Used to compare state->row layouts without a GPU.

  python tools/sweep_sim.py [machine.json] [--pipe 2] [--cols 40] [--layout plan|transpose|...]
"""
import argparse, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dnastore_amd as da

NEG = -np.inf


def load(machine_path):
    m = da.Machine.fromFile(machine_path)
    p = da.MutatorParams.fromFlags(global_=True)
    fm = da.FlatModel(m, p)
    return m, p, fm


def padded_in(ptr, src, score, n):
    deg = np.diff(ptr)
    w = max(1, int(deg.max()) if len(deg) else 1)
    S = np.zeros((n, w), dtype=np.int64)
    W = np.full((n, w), NEG)
    for j in range(n):
        a, b = ptr[j], ptr[j + 1]
        S[j, :b - a] = src[a:b]
        W[j, :b - a] = score[a:b]
    return S, W


class Sim:
    def __init__(self, fm):
        a = fm.arrays()
        self.a = a
        self.N = a["n_states"]
        self.D = a["max_dup_len"]
        sc = a["scores"]
        self.delOpen, self.tanDup, self.noGap, self.delExtend, self.delEnd = sc[:5]
        self.sub = sc[5:21].reshape(4, 4)
        self.len = sc[21:]
        self.eS, self.eW = padded_in(a["ein_ptr"], a["ein_src"], a["ein_score"], self.N)
        self.nS, self.nW = padded_in(a["nin_ptr"], a["nin_src"], a["nin_score"], self.N)
        # emitted base per emit in-edge, padded
        self.eB = np.zeros_like(self.eS)
        for j in range(self.N):
            x, y = a["ein_ptr"][j], a["ein_ptr"][j + 1]
            self.eB[j, :y - x] = a["ein_base"][x:y]

    def run(self, seq, row_of, K, pipe=2, max_cols=None, verbose=False, lane_of=None):
        """lane_of given: edges between different waves (lane // 64) only deliver what the source held at
        the start of the sweep (pessimistic model of waves that drift apart)."""
        N, D = self.N, self.D
        a = self.a
        mdl = a["mdl"].astype(np.int64)
        ctx = a["ctx"].astype(np.int64)
        rows = [np.nonzero(row_of == k)[0] for k in range(K)]
        if lane_of is not None:
            wave = lane_of // 64
            eSame = wave[self.eS] == wave[:, None]
            nSame = wave[self.nS] == wave[:, None]
        S = np.full(N, NEG); S[0] = 0.0
        T = np.full((N, max(D, 1)), NEG)
        L = len(seq)
        sweeps = []
        for pos in range(0, L + 1):
            if max_cols is not None and pos > max_cols:
                break
            if pos > 0:
                x = seq[pos - 1]
                cand = S[self.eS] + self.eW + self.noGap + self.sub[self.eB, x]
                Sn = cand.max(axis=1)
                has = (mdl > 0)
                Sn = np.where(has, np.maximum(Sn, T[:, 0] + self.sub[ctx[:, 0], x]), Sn)
                Tn = np.full_like(T, NEG)
                for k in range(D - 1):
                    ok = (k < mdl - 1)
                    Tn[:, k] = np.where(ok, T[:, k + 1] + self.sub[ctx[:, k + 1], x], NEG)
                S, T = Sn, Tn
            Dl = np.full(N, NEG)
            Sp = np.maximum(S, Dl + self.delEnd)
            X = np.maximum(Dl + self.delExtend, Sp + self.delOpen)
            S = Sp
            # sweeps; the gathers of row k see the state of the arrays `pipe-1` row evaluations ago
            n_sw = 0
            while True:
                n_sw += 1
                changed = False
                X0, D0, S0 = X.copy(), Dl.copy(), S.copy()
                pending = []   # (k, gathered values) queue to model the read-ahead
                hist = []
                for k in range(K + pipe - 1):
                    if k < K and len(rows[k]):
                        r = rows[k]
                        if lane_of is None:
                            ge = (X[self.eS[r]] + self.eW[r]).max(axis=1)
                            gd = (Dl[self.nS[r]] + self.nW[r]).max(axis=1)
                            gs = (S[self.nS[r]] + self.nW[r]).max(axis=1)
                        else:
                            ge = (np.where(eSame[r], X[self.eS[r]], X0[self.eS[r]]) + self.eW[r]).max(axis=1)
                            gd = (np.where(nSame[r], Dl[self.nS[r]], D0[self.nS[r]]) + self.nW[r]).max(axis=1)
                            gs = (np.where(nSame[r], S[self.nS[r]], S0[self.nS[r]]) + self.nW[r]).max(axis=1)
                        pending.append((k, ge, gd, gs))
                    elif k < K:
                        pending.append((k, None, None, None))
                    kk = k - (pipe - 1)
                    if kk >= 0:
                        k2, ge, gd, gs = pending.pop(0)
                        assert k2 == kk
                        if ge is None:
                            continue
                        r = rows[kk]
                        d = np.maximum(Dl[r], np.maximum(ge, gd))
                        s = np.maximum(S[r], gs)
                        s = np.maximum(s, d + self.delEnd)
                        ch = (s != S[r]) | (d != Dl[r])
                        if ch.any():
                            changed = True
                            S[r] = s; Dl[r] = d
                            X[r] = np.maximum(d + self.delExtend, s + self.delOpen)
                if not changed:
                    break
            sweeps.append(n_sw)
            if pos > 0:
                for k in range(D):
                    ok = (k < mdl)
                    T[:, k] = np.where(ok, np.maximum(T[:, k], S + self.tanDup + self.len[k]), T[:, k])
            if verbose:
                print("col", pos, "sweeps", n_sw, flush=True)
        return np.array(sweeps), S


def plan_rows(fm):
    lds, lat, T, K = fm.plan_slots()
    return (lds // T).astype(np.int64), (lds % T).astype(np.int64), T, K


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("machine", nargs="?", default=os.path.join(ROOT, "tests", "golden", "ref_data", "s16h74l4c4.json"))
    ap.add_argument("--pipe", type=int, default=2)
    ap.add_argument("--cols", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1000)
    args = ap.parse_args()
    m, p, fm = load(args.machine)
    sim = Sim(fm)
    rng = random.Random(args.seed)
    dna = m.encodeBytes(bytes(rng.randrange(256) for _ in range(29)))
    seq = da.tokenize(dna)
    row_of, lane_of, T, K = plan_rows(fm)
    sw, _ = sim.run(seq, row_of, K, pipe=args.pipe, max_cols=args.cols)
    print("plan layout: sweeps/col mean %.1f max %d  (first cols %s)" % (sw.mean(), sw.max(), sw[:12]))


if __name__ == "__main__":
    main()
