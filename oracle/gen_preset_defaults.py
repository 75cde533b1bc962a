#!/usr/bin/env python3
"""Fixture F12: the command-line contract of the reference's preset scripts.

TEST INFRASTRUCTURE ONLY; runs in the build container, where /root/reference exists.  The presets are scripts that run
a simulation on import, so they are not executed: their ``parser.add_argument(...)`` calls are read from the syntax
tree and every literal keyword (default, choices, nargs, type name) is evaluated.  Output: tests/golden/F12_preset_cli.json
= {preset: {flag: {default, choices, nargs, type}}} - data (flag names and values), no source text.

    python oracle/gen_preset_defaults.py [--out tests/golden/F12_preset_cli.json]
"""
import argparse
import ast
import json
import os

REF = "/root/reference/presets"
PRESETS = {"3wrobot": "main_3wrobot.py", "3wrobotNI": "main_3wrobot_NI.py", "2tank": "main_2tank.py"}


def cli_of(path):
    tree = ast.parse(open(path).read())
    flags = {}
    for node in ast.walk(tree):
        if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument"):
            continue
        name = ast.literal_eval(node.args[0])
        rec = {"default": None, "choices": None, "nargs": None, "type": None}
        for kw in node.keywords:
            if kw.arg in ("default", "choices", "nargs"):
                rec[kw.arg] = ast.literal_eval(kw.value)
            elif kw.arg == "type":
                rec["type"] = kw.value.id
        flags[name] = rec
    return flags


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests",
                                                  "golden", "F12_preset_cli.json"))
    a = ap.parse_args()
    out = {name: cli_of(os.path.join(REF, fn)) for name, fn in PRESETS.items()}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print({k: len(v) for k, v in out.items()}, "->", a.out)
