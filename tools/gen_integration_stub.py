#!/usr/bin/env python3
"""Rewrite the `rcg_cfg` ctypes stub of INTEGRATION.md (between the BEGIN/END markers) from the binding's own
`_native.RcgCfg._fields_`, so that the documented struct cannot drift from include/rcg.h
(tests/test_abi.py::test_documented_stub_matches_the_header compiles the header and compares sizes)."""
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rcognita_amd import _native as N  # noqa: E402

BEGIN, END = "# BEGIN rcg_cfg stub (generated", "# END rcg_cfg stub"


FIXED = {C.c_int32: "c_int32", C.c_uint32: "c_uint32", C.c_int64: "c_int64", C.c_uint64: "c_uint64", C.c_double: "c_double"}


def tname(t):
    if hasattr(t, "_length_"):
        return f"C.{FIXED[t._type_]} * {t._length_}"
    return f"C.{FIXED[t]}"


def stub():
    items = [f'("{n}", {tname(t)})' for n, t in N.RcgCfg._fields_]
    lines, cur = [], "    _fields_ = ["
    for it in items:
        if len(cur) + len(it) + 2 > 116:
            lines.append(cur.rstrip())
            cur = "                "
        cur += it + ", "
    lines.append(cur.rstrip(", ") + "]")
    return (f"{BEGIN} by tools/gen_integration_stub.py from rcognita_amd/_native.py; sizeof = {C.sizeof(N.RcgCfg)})\n"
            "class rcg_cfg(C.Structure):            # field-for-field mirror of include/rcg.h\n" + "\n".join(lines) +
            f"\n{END}")


if __name__ == "__main__":
    path = os.path.join(ROOT, "INTEGRATION.md")
    txt = open(path).read()
    new = re.sub(re.escape(BEGIN) + r".*?" + re.escape(END), lambda m: stub(), txt, flags=re.S)
    if new != txt:
        open(path, "w").write(new)
        print("INTEGRATION.md updated")
    else:
        print("INTEGRATION.md up to date")
