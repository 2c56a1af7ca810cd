#!/usr/bin/env python3
"""Extract golden vectors from the reference's playthrough files.

Run in the build container (needs /root/reference, which does NOT exist on the
GPU box):

    python tests/golden/make_golden.py

Source: /root/reference/open_spiel/integration_tests/playthroughs/*.txt, the
byte-exact per-state dumps the reference regenerates and diffs in
integration_tests/playthrough_test.py:74-95 (format produced by
python/algorithms/generate_playthrough.py:102-135,524).  For every state block
we keep what pins the hot path: IsTerminal, CurrentPlayer, History,
LegalActions (+ strings), Returns, ChanceOutcomes, the state / information /
observation strings and the full observation / information-state tensors,
plus the `action:` that leads to the next block (replay needs no RNG).

Output: tests/golden/playthroughs.json (compact; tensors as '0'/'1' strings
when binary, float lists otherwise).
"""
import ast
import json
import os
import re
import sys

SRC = "/root/reference/open_spiel/integration_tests/playthroughs"
FILES = [
    "tic_tac_toe.txt",
    "connect_four.txt",
    "hex(board_size=5).txt",
    "kuhn_poker_2p.txt",
    "kuhn_poker_3p.txt",
    "leduc_poker_773740114.txt",
    "leduc_poker_1540482260.txt",
    "leduc_poker_3977671846.txt",
    "leduc_poker_3p.txt",
]
HERE = os.path.dirname(os.path.abspath(__file__))

TENSOR_RE = re.compile(
    r"^(ObservationTensor|InformationStateTensor)\((\d+)\)(?:\.(\w+))?(: ?| = )(.*)$")
SYMS = {"◉": "1", "◯": "0"}


def _unescape(s):
    # inverse of generate_playthrough._escape
    return s.replace(r"\n", "\n").replace("\\\\", "\\")


def _sym_line(line):
    s = line.strip()
    return bool(s) and all(ch in "◉◯ " for ch in s)


def _flatten_symbols(lines):
    """lines: list of strings of ◉/◯ chunks -> '0'/'1' string in CHW order."""
    rows = [ln.split() for ln in lines if ln.strip()]
    if not rows:
        return ""
    nplanes = len(rows[0])
    assert all(len(r) == nplanes for r in rows), rows
    out = []
    for p in range(nplanes):
        for r in rows:
            out.append("".join(SYMS[ch] for ch in r[p]))
    return "".join(out)


def parse(path):
    with open(path, encoding="utf-8") as f:
        lines = f.read().split("\n")
    game = None
    header = {}
    states = []
    cur = None
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("game: "):
            game = ln[len("game: "):]
        elif re.match(r"^# State \d+$", ln):
            cur = {"to_string_lines": [], "tensors": {}, "info_str": {}, "obs_str": {}}
            states.append(cur)
            i += 1
            while (i < len(lines) and lines[i].startswith("#")
                   and not lines[i].startswith("# Apply action")):
                cur["to_string_lines"].append(lines[i][2:] if len(lines[i]) > 1 else "")
                i += 1
            continue
        elif cur is None:
            m = re.match(r"^(\w+)\(\) = (.*)$", ln)
            if m:
                header[m.group(1)] = m.group(2)
        else:
            m = TENSOR_RE.match(ln)
            if m:
                kind, player, piece, sep, rest = m.groups()
                key = ("obs" if kind == "ObservationTensor" else "info") + player
                if sep.strip() == "=":
                    vals = [float(x) for x in ast.literal_eval(rest)]
                    cur["tensors"].setdefault(key, []).append(vals)
                else:
                    chunk = [rest] if rest.strip() else []
                    while i + 1 < len(lines) and _sym_line(lines[i + 1]):
                        i += 1
                        chunk.append(lines[i])
                    cur["tensors"].setdefault(key, []).append(_flatten_symbols(chunk))
            elif ln.startswith("action: "):
                cur["action"] = int(ln[len("action: "):])
            else:
                m = re.match(r"^(\w+)\((\d*)\) = (.*)$", ln)
                if m:
                    name, arg, val = m.groups()
                    if name == "IsTerminal":
                        cur["is_terminal"] = val == "True"
                    elif name == "History":
                        cur["history"] = ast.literal_eval(val)
                    elif name == "CurrentPlayer":
                        cur["current_player"] = int(val)
                    elif name == "LegalActions" and arg == "":
                        cur["legal_actions"] = ast.literal_eval(val)
                    elif name == "StringLegalActions" and arg == "":
                        cur["string_legal_actions"] = ast.literal_eval(val)
                    elif name == "Returns":
                        cur["returns"] = [float(x) for x in ast.literal_eval(val)]
                    elif name == "ChanceOutcomes":
                        cur["chance_outcomes"] = [list(t) for t in ast.literal_eval(val)]
                    elif name == "InformationStateString":
                        cur["info_str"][arg] = _unescape(ast.literal_eval(val))
                    elif name == "ObservationString":
                        cur["obs_str"][arg] = _unescape(ast.literal_eval(val))
        i += 1
    out_states = []
    for s in states:
        if "is_terminal" not in s:  # the dump elides some mid-game blocks
            out_states.append({"skipped": True, "action": s["action"]})
            continue
        tensors = {}
        for key, pieces in s["tensors"].items():
            if all(isinstance(p, str) for p in pieces):
                tensors[key] = "".join(pieces)
            else:  # mixed binary / float pieces -> float list
                flat = []
                for p in pieces:
                    flat.extend([float(c) for c in p] if isinstance(p, str) else p)
                tensors[key] = flat
        d = {
            "to_string": "\n".join(s["to_string_lines"]),
            "is_terminal": s["is_terminal"],
            "history": s["history"],
            "current_player": s["current_player"],
            "legal_actions": s.get("legal_actions", []),
            "string_legal_actions": s.get("string_legal_actions", []),
            "info_str": s["info_str"],
            "obs_str": s["obs_str"],
            "tensors": tensors,
        }
        if "returns" in s:  # not printed at chance nodes
            d["returns"] = s["returns"]
        if "chance_outcomes" in s:
            d["chance_outcomes"] = s["chance_outcomes"]
        if "action" in s:
            d["action"] = s["action"]
        out_states.append(d)
    return {"game": game, "header": header, "states": out_states}


def main():
    if not os.path.isdir(SRC):
        sys.exit("reference playthroughs not found (this script runs in the build container only)")
    out = {}
    for fn in FILES:
        out[fn] = parse(os.path.join(SRC, fn))
        print(fn, out[fn]["game"], len(out[fn]["states"]), "states")
    with open(os.path.join(HERE, "playthroughs.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, separators=(",", ":"), ensure_ascii=False)
    print("wrote", os.path.join(HERE, "playthroughs.json"))


if __name__ == "__main__":
    main()
