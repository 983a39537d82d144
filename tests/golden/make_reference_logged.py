#!/usr/bin/env python
"""Writes tests/golden/reference_logged.json: the numbers the REFERENCE ITSELF printed when its authors ran it (A800), parsed
from the run logs it ships -- results{,_full_history}/Grad_Dependent_Nonlinear/{20,40,60,80}d/{RepeatedExperiment,SimpleUniform}/*.log.
These are outputs of the reference, not of this build: the parity tests compare the oracle and the HIP path with them.
Run where /root/reference exists (this container); the JSON is committed, the logs are not copied.

    python tests/golden/make_reference_logged.py [/root/reference]
"""
import json
import os
import re
import sys

SOLVERS = {"GP": "GP", "MLP": "MLP", "SCaSML": "ScaSML"}
SECTIONS = (("MEAN RELATIVE L2 ERROR", "rel_l2"), ("MEAN L1 ERROR", "l1"), ("MEAN L2 ERROR", "l2"))


def parse_repeated(path):
    """{metric: {solver: {mean, std, min, max}}} from a RepeatedExperiment.log."""
    text = open(path).read()
    out = {"source": None, "repetitions": int(re.search(r"Results over (\d+) successful repetitions", text).group(1))}
    for title, key in SECTIONS:
        start = text.index(title)
        nxt = min([text.index(t, start + 1) for t, _ in SECTIONS if t in text[start + 1:]] + [len(text)])
        block = text[start:nxt]
        out[key] = {}
        for name, ours in SOLVERS.items():
            m = re.search(re.escape(name) + r" - Mean [^\n]*:\s*\n\s*Mean:\s*([0-9.eE+-]+)\s*\n\s*Std:\s*([0-9.eE+-]+)\s*\n[^\n]*\n\s*Range:\s*\[([0-9.eE+-]+),\s*([0-9.eE+-]+)\]", block)
            if m:
                out[key][ours] = {"mean": float(m.group(1)), "std": float(m.group(2)), "min": float(m.group(3)), "max": float(m.group(4))}
    return out


def parse_simple(path):
    """SimpleUniform.log: lines 4-6 hold the relative L2 errors of GP, MLP, SCaSML of the single run."""
    vals = {}
    for line in open(path):
        m = re.match(r"\s*(GP|MLP|ScaSML|SCaSML)[^:]*rel[^:]*:\s*([0-9.eE+-]+)", line, re.I)
        if m:
            vals[SOLVERS.get(m.group(1), m.group(1))] = float(m.group(2))
    return vals


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = {"_about": "numbers printed by the reference's own runs (A800), parsed from the logs it ships; see make_reference_logged.py",
           "quadrature": {}, "full_history": {}}
    for kind, top in (("quadrature", "results"), ("full_history", "results_full_history")):
        for d in (20, 40, 60, 80):
            base = os.path.join(ref, top, "Grad_Dependent_Nonlinear", "%dd" % d)
            rep = os.path.join(base, "RepeatedExperiment", "RepeatedExperiment.log")
            sim = os.path.join(base, "SimpleUniform", "SimpleUniform.log")
            entry = {}
            if os.path.exists(rep):
                entry["repeated"] = parse_repeated(rep)
                entry["repeated"]["source"] = os.path.relpath(rep, ref)
            if os.path.exists(sim):
                entry["simple_uniform"] = {"source": os.path.relpath(sim, ref), "head": [l.rstrip("\n") for l in open(sim).readlines()[:21] if l.strip()]}
            out[kind][str(d)] = entry
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_logged.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(path)


if __name__ == "__main__":
    main()
