"""Random transducers for fuzzing the tier-A plan and kernel: shapes the fixture machines do not have
(out-degree up to 5, self loops, high in-degree, several score classes, mixed emit/null out-edges, short and
missing left contexts).  Valid Machine JSON by construction: an emit edge into state v always emits v's last
context character (verifyContexts, trans.cpp:484-496), edges without output only point forward (the decoder's
toposort, trans.cpp:604-634), the last state is the end state."""
import json
import random


def random_machine(seed, n_states):
    rng = random.Random(seed)
    last = [rng.choice("ACGT") for _ in range(n_states)]
    ctx_len = [rng.choice([0, 1, 2, 4, 4]) for _ in range(n_states)]
    states = []
    for i in range(n_states):
        l = ""
        if ctx_len[i]:
            l = "".join(rng.choice("ACGT") for _ in range(ctx_len[i] - 1)) + last[i]
        states.append({"n": i, "id": "s%d" % i, "l": l, "trans": []})
    hub = rng.randrange(1, n_states - 1)                      # a state with many in-edges
    for i in range(n_states - 1):
        trans = states[i]["trans"]
        # a spine edge keeps every state on a path to the end
        if rng.random() < 0.7:
            trans.append({"in": rng.choice(["", "0", "1"]), "out": last[i + 1], "to": i + 1})
        else:
            trans.append({"in": rng.choice(["", "0", "1", "A"]), "out": "", "to": i + 1})
        for _ in range(rng.choice([0, 0, 1, 1, 2, 4])):
            kind = rng.random()
            if kind < 0.55:                                     # emit edge anywhere (back, self, forward)
                to = rng.randrange(0, n_states)
                trans.append({"in": rng.choice(["", "0", "1", "^", "B"]), "out": last[to], "to": to})
            elif kind < 0.85 and i + 1 < n_states:              # null edge, forward only
                to = rng.randrange(i + 1, n_states)
                trans.append({"in": rng.choice(["", "0", "1", "$"]), "out": "", "to": to})
            else:                                               # into the hub
                if hub > i:
                    trans.append({"in": "", "out": "", "to": hub})
                else:
                    trans.append({"in": "1", "out": last[hub], "to": hub})
        for t in trans:
            for k in ("in", "out"):
                if t[k] == "":
                    del t[k]
    return json.dumps({"state": states})


def random_read(seed, machine_json, max_len=40, noise=0.1):
    """Characters emitted along a random walk from state 0 towards the end state, lightly mutated."""
    rng = random.Random(seed)
    states = json.loads(machine_json)["state"]
    cur, out = 0, []
    for _ in range(4 * max_len):
        trans = states[cur]["trans"]
        if not trans or len(out) >= max_len:
            break
        fwd = [t for t in trans if t["to"] > cur]
        t = rng.choice(fwd if fwd and rng.random() < 0.7 else trans)
        if "out" in t:
            out.append(t["out"])
        cur = t["to"]
    read = []
    for c in out:
        r = rng.random()
        if r < noise / 2:
            continue                                            # deletion
        read.append(rng.choice("ACGT") if r < noise else c)    # substitution
    return "".join(read) or "A"
