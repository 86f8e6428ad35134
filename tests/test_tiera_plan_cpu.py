"""CPU check of the tier-A plan (dnastore_amd/csrc/host/plan.cpp): the tables the HIP fill kernel
receives -- placement, row shapes, out-edge entries, S stripes -- are executed by a plain numpy
restatement of the kernel's column recurrence (push along out-edges into per-state accumulators,
sweep to the fixpoint) and every S and D cell is compared, bit for bit, with the oracle's lattice.
No GPU: this pins the host side of tier A; the kernel itself is covered by the `-m gpu` tests."""
import os

import numpy as np
import pytest

NEG = -np.inf


def _decode_plan(fm, members):
    """Per-state out-edge lists [(dst, class, base, is_null)] and the has-S flag of every state, decoded from the
    plan tables alone (members = 1: tier A; >= 2: a tier-C cluster)."""
    a = fm.arrays()
    N = a["n_states"]
    pl = fm.cluster_plan(members)
    G, K, T, shapes, ent = pl["G"], pl["K"], pl["T"], pl["shapes"], pl["entries"]
    n_s, n_in, fold = pl["n_s_rows"], pl["n_inbox_rows"], pl["fold"]
    member_of, lds_idx = pl["member_of"], pl["lds_index"]
    dc_base = n_s * T * 8 + 64
    row_off = np.concatenate([[0], np.cumsum(np.maximum(shapes[:, 0], 0))])
    state_at = np.full((G, K * T), -1, dtype=np.int64)
    state_at[member_of, lds_idx] = np.arange(N)
    # proxies: places N, N+1, ... that hold no state of the machine -- a member's null edges into one state of another member end
    # there, and ONE null edge of class 0 carries their maximum on (host/plan.cpp)
    n_prox = len(pl["proxy_member"])
    state_at[pl["proxy_member"], pl["proxy_lds_index"]] = N + np.arange(n_prox)
    member_all = np.concatenate([member_of, pl["proxy_member"]])
    lds_all = np.concatenate([lds_idx, pl["proxy_lds_index"]])
    assert (state_at >= 0).sum() == N + n_prox, "two states share a place"
    out = [[] for _ in range(N + n_prox)]
    cells_seen = set()
    for j in range(N + n_prox):
        g = int(member_all[j])
        k, t = divmod(int(lds_all[j]), T)
        assert shapes[k, 0] >= 0, "state placed in a row the plan marks empty"
        kind, cls_common, g_out = shapes[k, 2], shapes[k, 3], shapes[k, 5]
        for e in range(int(shapes[k, 0])):
            en = int(ent[g, row_off[k] + e, t])
            if en == 0:
                assert not shapes[k, 4], "empty entry in a row the plan calls full"
                continue
            cls = en & 3
            assert cls_common < 0 or cls == cls_common
            if en & 4:                                            # destination in another member's inbox
                assert g_out in (1, 2)
                cell = (en >> 3) & 0xfffff
                assert cell not in cells_seen, "two edges share an inbox cell (a cell has ONE writer)"
                cells_seen.add(cell)
                dg, idx = divmod(cell, n_in * T)                  # cell (r, t) of a member sits at (r/2)*2T + 2t + (r&1)
                pair, rem = divmod(idx, 2 * T)
                r_in, t_in = 2 * pair + (rem & 1), rem // 2
                assert dg != g, "inbox entry into the own member"
                f = int(fold[dg, r_in, t_in])                     # the fold table says which LDS cells the cell feeds
                assert f != 0, "offer into an unused inbox cell"
                dst_idx = ((f & 0xffff) * 8 - dc_base) // 8
                dst = int(state_at[dg, dst_idx])
                is_null = bool(en & 0x800000)
                base = (en >> 24) & 3
                drow = dst_idx // T
                if is_null:
                    assert (f >> 16) != 0xffff and (f >> 16) == shapes[drow, 1] * T + dst_idx % T, "inbox S cell is not the destination's"
            else:
                assert g_out in (0, 2)
                dst_idx = ((en & 0x3fff8) - dc_base) // 8
                dst = int(state_at[g, dst_idx])
                drow = dst_idx // T
                is_null = en < 0xffe00000
                base = (en >> 19) & 3
                if is_null:
                    sc_idx = en >> 19
                    assert shapes[drow, 1] >= 0 and sc_idx == shapes[drow, 1] * T + dst_idx % T, "S cell of a null edge is not the destination's"
            assert dst >= 0, "entry points at an empty slot"
            assert kind == 0 or kind == (2 if is_null else 1)
            out[j].append((dst, cls, base, is_null))
    for i in range(n_prox):                                       # a proxy forwards over one null edge that adds nothing, into another member
        (dst, cls, _, is_null), = out[N + i]
        assert is_null and cls == 0 and dst < N and member_of[dst] != pl["proxy_member"][i]
        assert shapes[pl["proxy_lds_index"][i] // T, 1] >= 0, "a proxy needs an S cell"
    for j in range(N):                                            # ... so that an edge into a proxy is an edge to the proxy's destination
        for i, (dst, cls, base, is_null) in enumerate(out[j]):
            if dst >= N:
                assert is_null and member_of[j] == pl["proxy_member"][dst - N]
                out[j][i] = (out[dst][0][0], cls, base, True)
    out = out[:N]
    rows = lds_idx // T
    has_s = shapes[rows, 1] >= 0
    # every used inbox cell has a writer
    assert G == 1 or int((fold != 0).sum()) == len(cells_seen)
    return out, has_s


def _emulate(fm, seq, local, members=1):
    """S and D lanes [L+1][N] computed from the plan tables alone (+ the score scalars of the flat model)."""
    a = fm.arrays()
    N, D = a["n_states"], a["max_dup_len"]
    sc = a["scores"]
    del_open, tan_dup, no_gap, del_extend, del_end = sc[:5]
    sub = sc[5:21].reshape(4, 4)
    length = sc[21:]
    # distinct edge scores in the plan's class order: 0.0 first, then by first appearance over destinations
    scores = [0.0]
    for j in range(N):
        for v in list(a["ein_score"][a["ein_ptr"][j]:a["ein_ptr"][j + 1]]) + list(a["nin_score"][a["nin_ptr"][j]:a["nin_ptr"][j + 1]]):
            if v not in scores:
                scores.append(v)
    out, has_s = _decode_plan(fm, members)
    # the decoded edges are exactly the machine's usable edges
    want = sorted([(int(a["ein_src"][e]), j, int(a["ein_base"][e]), False) for j in range(N) for e in range(a["ein_ptr"][j], a["ein_ptr"][j + 1])] +
                  [(int(a["nin_src"][e]), j, 0, True) for j in range(N) for e in range(a["nin_ptr"][j], a["nin_ptr"][j + 1])])
    got = sorted((j, dst, 0 if is_null else base, is_null) for j in range(N) for dst, _, base, is_null in out[j])
    assert got == want
    out = [[(dst, cls, base, (0 if is_null else None)) for dst, cls, base, is_null in lst] for lst in out]
    mdl = a["mdl"].astype(int)
    ctx = a["ctx"].astype(int)
    L = len(seq)
    S_lat = np.full((L + 1, N), NEG)
    D_lat = np.full((L + 1, N), NEG)
    S = np.full(N, NEG)
    Tl = np.full((N, max(D, 1)), NEG)
    for pos in range(L + 1):
        DC = np.full(N, NEG)
        SC = np.full(N, NEG)
        if pos == 0:
            S[:] = 0.0 if local else NEG
            S[0] = 0.0
        else:
            x = int(seq[pos - 1])
            for j in range(N):                                   # phase A: offers along the emit edges
                if S[j] > NEG:
                    for dst, cls, base, sc_idx in out[j]:
                        if sc_idx is None:
                            DC[dst] = max(DC[dst], ((S[j] + scores[cls]) + no_gap) + sub[base][x])
            Sn = DC.copy()
            Tn = np.full_like(Tl, NEG)
            for j in range(N):
                if mdl[j] > 0:
                    Sn[j] = max(Sn[j], Tl[j, 0] + sub[ctx[j, 0]][x])
                    for q in range(mdl[j] - 1):
                        Tn[j, q] = Tl[j, q + 1] + sub[ctx[j, q + 1]][x]
            S, Tl = Sn, Tn
            DC[:] = NEG
        Dv = np.full(N, np.inf)                                   # "not evaluated yet"
        changed = True
        while changed:                                            # sweeps
            changed = False
            for j in range(N):
                d = DC[j]
                s = max(S[j], SC[j]) if has_s[j] else S[j]
                if d != Dv[j] or s != S[j]:
                    changed = True
                    s = max(s, d + del_end)
                    S[j], Dv[j] = s, d
                    xv = max(d + del_extend, s + del_open)
                    for dst, cls, base, sc_idx in out[j]:
                        if sc_idx is None:
                            DC[dst] = max(DC[dst], xv + scores[cls])
                        else:
                            DC[dst] = max(DC[dst], d + scores[cls])
                            SC[dst] = max(SC[dst], s + scores[cls])
        S_lat[pos], D_lat[pos] = S, Dv
        if pos > 0:
            for j in range(N):
                for q in range(mdl[j]):
                    Tl[j, q] = max(Tl[j, q], (S[j] + tan_dup) + length[q])
    if local:                                                     # viterbi.cpp:171-173
        S_lat[L, N - 1] = S_lat[L].max()
    return S_lat, D_lat


@pytest.mark.parametrize("mach,fa,flags", [
    ("l4c4.json", "hello.fa", dict()),
    ("l4c4.json", "hello.dup.fa", dict(global_=True)),
    ("mr2l4c4.json", "hello.mr2.fa", dict(global_=True)),
])
def test_plan_tables_reproduce_the_oracle_lattice(oracle_mod, ref_data, mach, fa, flags):
    import dnastore_amd as da
    O = oracle_mod
    path = os.path.join(ref_data, mach)
    fm = da.FlatModel(da.Machine.fromFile(path), da.MutatorParams.fromFlags(**flags))
    read = da.read_fastseqs(os.path.join(ref_data, fa))[0][1][:24]          # pure-Python loops: keep it short
    orc = O.ViterbiOracle(O.Machine.from_file(path), O.MutatorParams.from_cli(**flags))
    _, _, olat = orc.decode(read, want_lattice=True)                       # [L+1][N][lanes]
    for members in (1, 2, 3):      # tier A, and the same machine cut over a cluster of 2 and 3 work-groups (tier C)
        S_lat, D_lat = _emulate(fm, da.tokenize(read), local=not flags.get("global_", False), members=members)
        assert np.array_equal(S_lat.view(np.uint64), np.ascontiguousarray(olat[:, :, 0]).view(np.uint64))
        assert np.array_equal(D_lat.view(np.uint64), np.ascontiguousarray(olat[:, :, 1]).view(np.uint64))


@pytest.mark.parametrize("order,slack", [(0, 0), (2, 0), (2, 8), (2, 3)])
def test_plan_dealing_orders(oracle_mod, ref_data, monkeypatch, order, slack):
    """The other dealing orders (DNAS_PLAN_ORDER: 0 depth first, 2 longest-path levels with DNAS_PLAN_SLACK eighths of the room
    used; 1, breadth first, is the default every other test runs): same edges, same lattice, for tier A and a cluster of two."""
    import dnastore_amd as da
    O = oracle_mod
    monkeypatch.setenv("DNAS_PLAN_ORDER", str(order))
    monkeypatch.setenv("DNAS_PLAN_SLACK", str(slack))
    path = os.path.join(ref_data, "mr2l4c4.json")
    flags = dict(global_=True)
    fm = da.FlatModel(da.Machine.fromFile(path), da.MutatorParams.fromFlags(**flags))
    read = da.read_fastseqs(os.path.join(ref_data, "hello.mr2.fa"))[0][1][:20]
    orc = O.ViterbiOracle(O.Machine.from_file(path), O.MutatorParams.from_cli(**flags))
    _, _, olat = orc.decode(read, want_lattice=True)
    for members in (1, 2):
        S_lat, D_lat = _emulate(fm, da.tokenize(read), local=False, members=members)
        assert np.array_equal(S_lat.view(np.uint64), np.ascontiguousarray(olat[:, :, 0]).view(np.uint64))
        assert np.array_equal(D_lat.view(np.uint64), np.ascontiguousarray(olat[:, :, 1]).view(np.uint64))


def test_plan_row_program_invariants(ref_data):
    """Every state placed once, rows within capacity, null edges only into rows with an S stripe."""
    import dnastore_amd as da
    for mach in ("l4c4.json", "h74l4c4.json", "s16h74l4c4.json"):
        fm = da.FlatModel(da.Machine.fromFile(os.path.join(ref_data, mach)), da.MutatorParams.fromFlags(global_=True))
        shapes, ent, meta, n_s = fm.plan_tables()
        lds_idx, lat_slot, T, K = fm.plan_slots()
        N = fm.view.contents.n_states
        assert len(set(lds_idx.tolist())) == N and lds_idx.min() >= 0 and lds_idx.max() < K * T
        assert len(set(lat_slot.tolist())) == N
        rows, lanes = lds_idx // T, lds_idx % T
        # lattice slot = pair * 2T + 2 * lane + side: every row is one side of one cell pair (the plan pairs rows by their fill)
        pair_side = (lat_slot // (2 * T)) * 2 + (lat_slot & 1)
        assert np.array_equal((lat_slot % (2 * T)) // 2, lanes)
        row_of = {}
        for r, q in zip(rows.tolist(), pair_side.tolist()):
            assert row_of.setdefault(q, r) == r, "two rows share a side of a cell pair"
        assert len(set(row_of.values())) == len(row_of) and max(row_of) < K
        assert sorted(s for s in shapes[:, 1] if s >= 0) == list(range(n_s))
        valid = (meta >> 29) & 1
        assert int(valid.sum()) == N
        assert ent.shape == (max(int(np.maximum(shapes[:, 0], 0).sum()), 1), T)
        # rows the plan left empty hold no state
        for k in range(K):
            if shapes[k, 0] < 0:
                assert not (rows == k).any()
