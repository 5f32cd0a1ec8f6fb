#!/usr/bin/env python3
"""The `ipc_service` surface the reference's trainers actually use, extracted from their source by the Python parser (no import: they need dgl): every
`ipc_service.<name>(...)` call in pytorch_extension/legion_graphsage.py, legion_gcn.py and lp_sage.py with its number of arguments and -- where the result is
unpacked -- the number of values the script expects back.  Output: tests/golden/trainer_api.json (data only; this script is the generator).

    python oracle/make_golden_trainer_api.py          # in the build container, where /root/reference exists
"""
import ast
import json
import os

REF = "/root/reference/pytorch_extension"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "trainer_api.json")


def calls_of(path):
    tree = ast.parse(open(path).read())
    found = []
    for node in ast.walk(tree):
        call, unpack = None, None
        if isinstance(node, ast.Assign) and isinstance(node.value, ast.Call):
            call = node.value
            if len(node.targets) == 1 and isinstance(node.targets[0], (ast.Tuple, ast.List)):
                unpack = len(node.targets[0].elts)
        elif isinstance(node, ast.Expr) and isinstance(node.value, ast.Call):
            call = node.value
        if call is None or not (isinstance(call.func, ast.Attribute) and isinstance(call.func.value, ast.Name) and call.func.value.id == "ipc_service"):
            continue
        found.append(dict(name=call.func.attr, args=len(call.args) + len(call.keywords), unpacked_into=unpack, line=node.lineno))
    return found


def main():
    out = {"generator": "oracle/make_golden_trainer_api.py", "scripts": {}}
    for f in ("legion_graphsage.py", "legion_gcn.py", "lp_sage.py"):
        out["scripts"][f] = calls_of(os.path.join(REF, f))
    sig = {}
    for f, calls in out["scripts"].items():
        for c in calls:
            sig.setdefault(c["name"], set()).add((c["args"], c["unpacked_into"]))
    out["surface"] = {k: sorted([list(x) for x in v], key=str) for k, v in sorted(sig.items())}
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=0)
    print(json.dumps(out["surface"]))


if __name__ == "__main__":
    main()
