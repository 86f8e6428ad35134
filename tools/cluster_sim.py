"""Offline model of the tier-C in-column fixpoint: how many sweeps does a column take when the members of a cluster poll
their inboxes once or several times per sweep?

Same relaxations as the kernel (push style, numpy, CPU only), all members and all threads in step: one time unit = one
row evaluation, a sweep = K units.  An offer into another member lands in the destination's inbox `lat` units after it
was made; the owner loads its inbox at the poll points of a sweep and folds what a load returned into its LDS cells at
the NEXT poll point (the kernel as it was: one poll per sweep, loaded at row 0, folded behind row K-1).

  python tools/cluster_sim.py [--config 1] [--polls 1,2,4] [--lat 12] [--cols 12]
"""
import argparse, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dnastore_amd as da
import bench

NEG = -np.inf


class ClusterSim:
    def __init__(self, fm, members=0):
        a = fm.arrays()
        self.a = a
        pl = fm.cluster_plan(members)
        self.G, self.K, self.T = pl["G"], pl["K"], pl["T"]
        N = self.N = a["n_states"]
        self.member = pl["member_of"].astype(np.int64)
        self.row = (pl["lds_index"] // self.T).astype(np.int64)
        sc = a["scores"]
        self.delOpen, self.tanDup, self.noGap, self.delExtend, self.delEnd = sc[:5]
        self.sub = sc[5:21].reshape(4, 4)
        self.len = sc[21:]
        self.D = a["max_dup_len"]
        # edge lists (src, dst, score, base, isNull)
        es, ed, ew, eb = [], [], [], []
        for j in range(N):
            x, y = a["ein_ptr"][j], a["ein_ptr"][j + 1]
            es.extend(a["ein_src"][x:y]); ed.extend([j] * (y - x)); ew.extend(a["ein_score"][x:y]); eb.extend(a["ein_base"][x:y])
        self.e_src, self.e_dst, self.e_w, self.e_b = (np.array(v) for v in (es, ed, ew, eb))
        ns, nd, nw = [], [], []
        for j in range(N):
            x, y = a["nin_ptr"][j], a["nin_ptr"][j + 1]
            ns.extend(a["nin_src"][x:y]); nd.extend([j] * (y - x)); nw.extend(a["nin_score"][x:y])
        self.n_src, self.n_dst, self.n_w = np.array(ns, dtype=np.int64), np.array(nd, dtype=np.int64), np.array(nw)
        self.e_remote = self.member[self.e_src] != self.member[self.e_dst]
        self.n_remote = self.member[self.n_src] != self.member[self.n_dst] if len(ns) else np.zeros(0, bool)
        # out-edges grouped by source row (all members in step)
        self.e_by_row = [np.nonzero(self.row[self.e_src] == k)[0] for k in range(self.K)]
        self.n_by_row = [np.nonzero(self.row[self.n_src] == k)[0] for k in range(self.K)] if len(ns) else [np.zeros(0, np.int64)] * self.K
        self.states_by_row = [np.nonzero(self.row == k)[0] for k in range(self.K)]
        self.cross = (self.e_remote.sum() + self.n_remote.sum()) / max(1, len(es) + len(ns))

    def run(self, seq, polls, lat, max_cols, fold_same_point=False, lag=1, load_rows=None):
        """polls: rows at which a sweep polls the inbox (fold what the previous poll loaded, then load).  polls = [0] with
        fold_at_end: the kernel of round 2 (load at row 0, fold behind the last row)."""
        N, K, D = self.N, self.K, self.D
        a = self.a
        mdl = a["mdl"].astype(np.int64)
        ctx = a["ctx"].astype(np.int64)
        S = np.full(N, NEG); S[0] = 0.0
        T = np.full((N, max(D, 1)), NEG)
        out = []
        for pos in range(0, min(len(seq), max_cols) + 1):
            if pos > 0:
                x = seq[pos - 1]
                Sn = np.full(N, NEG)
                np.maximum.at(Sn, self.e_dst, ((S[self.e_src] + self.e_w) + self.noGap) + self.sub[self.e_b, x])
                has = mdl > 0
                Sn = np.where(has, np.maximum(Sn, T[:, 0] + self.sub[ctx[:, 0], x]), Sn)
                Tn = np.full_like(T, NEG)
                for k in range(D - 1):
                    Tn[:, k] = np.where(k < mdl - 1, T[:, k + 1] + self.sub[ctx[:, k + 1], x], NEG)
                S, T = Sn, Tn
            Dv = np.full(N, np.inf)                # "fresh"
            DC = np.full(N, NEG); SC = np.full(N, NEG)
            XD = np.full(N, NEG); XS = np.full(N, NEG)          # inboxes (as the owner's loads see them)
            snaps = [(np.full(N, NEG), np.full(N, NEG)) for _ in range(lag)]    # what the last `lag` polls loaded, not folded yet (oldest first)
            flight = []                                          # (arrival time, dst array, d values, s values or None)
            t = 0
            n_sw = 0
            while True:
                n_sw += 1
                changed = False
                for k in range(K):
                    # offers that have landed by now
                    keep = []
                    for item in flight:
                        if item[0] <= t:
                            np.maximum.at(XD, item[1], item[2])
                            if item[3] is not None: np.maximum.at(XS, item[1], item[3])
                        else:
                            keep.append(item)
                    flight = keep
                    if load_rows is not None:
                        # one fold per sweep (at row 0, of what the last load returned) and loads at other rows
                        if k == 0:
                            snapD, snapS = snaps[-1]
                            g = snapD > DC
                            if g.any(): DC = np.where(g, snapD, DC); changed = True
                            g = snapS > SC
                            if g.any(): SC = np.where(g, snapS, SC); changed = True
                        if k in load_rows:
                            snaps = [(XD.copy(), XS.copy())]
                    elif k in polls:
                        # fold what the previous poll loaded; load again
                        if fold_same_point:
                            snaps = [(XD.copy(), XS.copy())]
                        snapD, snapS = snaps.pop(0)
                        g = snapD > DC
                        if g.any(): DC = np.where(g, snapD, DC); changed = True
                        g = snapS > SC
                        if g.any(): SC = np.where(g, snapS, SC); changed = True
                        snaps.append((XD.copy(), XS.copy()))
                    r = self.states_by_row[k]
                    if len(r):
                        d = DC[r]
                        s = np.maximum(S[r], SC[r])
                        grew = (d != Dv[r]) | (s != S[r])
                        if grew.any():
                            changed = True
                            s2 = np.maximum(s, d + self.delEnd)
                            S[r] = np.where(grew, s2, S[r]); Dv[r] = np.where(grew, d, Dv[r])
                            gmask = np.zeros(N, bool); gmask[r[grew]] = True
                            e = self.e_by_row[k]
                            e = e[gmask[self.e_src[e]]]
                            if len(e):
                                src = self.e_src[e]
                                xv = np.maximum(Dv[src] + self.delExtend, S[src] + self.delOpen) + self.e_w[e]
                                loc = ~self.e_remote[e]
                                np.maximum.at(DC, self.e_dst[e][loc], xv[loc])
                                if (~loc).any(): flight.append((t + lat, self.e_dst[e][~loc], xv[~loc], None))
                            e = self.n_by_row[k]
                            if len(e):
                                e = e[gmask[self.n_src[e]]]
                            if len(e):
                                src = self.n_src[e]
                                dv = Dv[src] + self.n_w[e]; sv = S[src] + self.n_w[e]
                                loc = ~self.n_remote[e]
                                np.maximum.at(DC, self.n_dst[e][loc], dv[loc]); np.maximum.at(SC, self.n_dst[e][loc], sv[loc])
                                if (~loc).any(): flight.append((t + lat, self.n_dst[e][~loc], dv[~loc], sv[~loc]))
                    t += 1
                if not changed and not flight and not (XD > DC).any() and not (XS > SC).any():
                    break
                if n_sw > 400: raise RuntimeError("no convergence")
            out.append(n_sw)
            if pos > 0:
                for k in range(D):
                    T[:, k] = np.where(k < mdl, np.maximum(T[:, k], S + self.tanDup + self.len[k]), T[:, k])
        return np.array(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--variant", default="a")
    ap.add_argument("--members", type=int, default=0)
    ap.add_argument("--polls", default="1,2,4")
    ap.add_argument("--lat", default="12")
    ap.add_argument("--cols", type=int, default=10)
    ap.add_argument("--lag", default="1", help="a poll folds what the poll `lag` polls before it loaded")
    ap.add_argument("--load-rows", default="", help="one fold per sweep at row 0; the loads at these rows (comma separated), one run each")
    args = ap.parse_args()
    wl = bench.workload(da, args.config, args.variant)
    m = wl["machine"]
    fm = da.FlatModel(m, da.MutatorParams.fromFlags(global_=True))
    sim = ClusterSim(fm, args.members)
    print("G %d K %d T %d cross edges %.3f" % (sim.G, sim.K, sim.T, sim.cross), flush=True)
    seq = da.tokenize(bench.make_reads(m, 0, 1, payload_bytes=wl["payload_bytes"])[0])
    for lat in (int(v) for v in args.lat.split(",")):
        for lr in (int(v) for v in args.load_rows.split(",") if v):
            sw = sim.run(seq, set(), lat, args.cols, load_rows={lr})
            print("lat %2d  fold at row 0, load at row %d: sweeps/col mean %.1f  %s" % (lat, lr, sw[1:].mean(), sw), flush=True)
        for P in (int(v) for v in args.polls.split(",") if v):
            if P == 0:
                polls, same = [0], True      # ideal: an offer is in the destination's LDS cell `lat` units later, seen from row 0 on
                sw = sim.run(seq, set(range(sim.K)), lat, args.cols, fold_same_point=True)
                print("lat %2d  poll every row, folded at once: sweeps/col mean %.1f  %s" % (lat, sw[1:].mean(), sw), flush=True)
                continue
            polls = set((i * sim.K) // P for i in range(P))
            for lag in (int(v) for v in args.lag.split(",")):
                sw = sim.run(seq, polls, lat, args.cols, lag=lag)
                print("lat %2d  %d polls per sweep, lag %d: sweeps/col mean %.1f  %s" % (lat, P, lag, sw[1:].mean(), sw), flush=True)


if __name__ == "__main__":
    main()
