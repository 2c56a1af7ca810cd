"""Generates tests/golden/observer_vectors.json from the GENUINE reference build (oracle/_ref/libspiel_ref.so):
every IIGObservationType (public_info x perfect_recall x PrivateInfoType) plus the default observer, for every
player, at every ply of a few seeded playouts of kuhn_poker / leduc_poker (2 and 3 players) and of the three board
games — the Observation's pieces, tensor, string and Compress() bytes (observer.cc, kuhn_poker.cc:65-165,
leduc_poker.cc:92-242).  Run where /root/reference exists:  python tests/golden/make_observer_vectors.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import reference_py as ref  # noqa: E402

GAMES = [("kuhn_poker", 3), ("kuhn_poker(players=3)", 2), ("leduc_poker", 3), ("leduc_poker(players=3)", 2),
         ("tic_tac_toe", 1), ("connect_four", 1), ("hex(board_size=4)", 1)]
TYPES = [None] + [(pub, rec, prv) for pub in (0, 1) for rec in (0, 1) for prv in (0, 1, 2)]


def pack(tensor):
    """None (no tensor), or {"n": size, "ones": [indices]} for a 0/1 tensor, or {"n": size, "values": [...]}."""
    if tensor is None:
        return None
    values = [float(x) for x in tensor]
    if all(v in (0.0, 1.0) for v in values):
        return {"n": len(values), "ones": [i for i, v in enumerate(values) if v]}
    return {"n": len(values), "values": values}


def unpack(packed):
    import numpy as np
    if packed is None:
        return None
    out = np.zeros(packed["n"], np.float32)
    if "ones" in packed:
        out[packed["ones"]] = 1.0
    else:
        out[:] = packed["values"]
    return out


def main():
    assert ref.available(), "oracle/_ref/libspiel_ref.so is missing: python -c 'import __graft_entry__ as g; g.build()'"
    out = {"types": [list(t) if t else None for t in TYPES], "games": {}}
    for game_string, playouts in GAMES:
        game = ref.Game(game_string)
        board = game.max_chance_outcomes == 0
        pieces, records = {}, []
        for seed in range(playouts):
            rng = random.Random(1000 + seed)
            state = game.new_initial_state()
            history = []
            while True:
                entry = {"history": list(history), "observers": []}
                for ti, t in enumerate(TYPES):
                    if board and t not in (None, (0, 0, 1), (1, 0, 1), (1, 1, 1), (1, 0, 2)):
                        entry["observers"].append("skipped")
                        continue
                    per_player = []
                    for player in range(game.num_players):
                        o = state.observer(player, t)
                        if o is None:
                            per_player = None
                            break
                        pieces[str(ti)] = o["pieces"]
                        rec = {"tensor": pack(o["tensor"]), "string": o["string"]}
                        if t in (None, (1, 1, 2), (1, 0, 0), (0, 1, 1)):   # Compress(): a few types are enough
                            rec["compressed"] = o["compressed"].hex()
                        per_player.append(rec)
                    entry["observers"].append(per_player)
                records.append(entry)
                if state.is_terminal():
                    break
                if state.is_chance_node():
                    acts = [a for a, _ in state.chance_outcomes()]
                else:
                    acts = state.legal_actions()
                a = rng.choice(acts)
                state.apply_action(a)
                history.append(int(a))
        out["games"][game_string] = {"pieces": pieces, "records": records}
    path = os.path.join(HERE, "observer_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes;", {g: len(r["records"]) for g, r in out["games"].items()})


if __name__ == "__main__":
    main()
