"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/rcg.h declares, the ctypes struct matches the C struct, and the product path fails loudly
(no CPU fallback) when there is no GPU.  No compute is launched here."""
import ctypes
import os
import re
import subprocess

import pytest

from tests.conftest import ROOT


def test_header_symbols_all_exported():
    from rcognita_amd import _native as N

    hdr = open(os.path.join(ROOT, "include", "rcg.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rcg_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(N.SYMBOLS), declared ^ set(N.SYMBOLS)
    L = N.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert L.rcg_version() == N.RCG_VERSION


def test_cfg_struct_layout_matches_c(tmp_path):
    """sizeof/offsetof of rcg_cfg as gcc sees it == the ctypes mirror."""
    from rcognita_amd import _native as N

    src = tmp_path / "layout.c"
    fields = [f[0] for f in N.RcgCfg._fields_]
    body = "".join(f'printf("{f} %zu\\n", offsetof(rcg_cfg, {f}));' for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rcg.h"\nint main(){printf("size %zu\\n", sizeof(rcg_cfg));'
                   + body + 'printf("summary %zu\\n", sizeof(rcg_summary));return 0;}')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    assert int(out["size"]) == ctypes.sizeof(N.RcgCfg)
    assert int(out["summary"]) == ctypes.sizeof(N.RcgSummary)
    for f in fields:
        assert int(out[f]) == getattr(N.RcgCfg, f).offset, f


def test_create_argument_validation_and_no_cpu_fallback():
    from rcognita_amd import Engine, EngineConfig
    from rcognita_amd import _native as N

    L = N.lib()
    h = ctypes.c_void_p()
    bad = N.RcgCfg()
    bad.struct_size = 12
    assert L.rcg_create(ctypes.byref(bad), ctypes.byref(h)) == N.ERR_BAD_ARG
    assert b"struct_size" in L.rcg_last_error(None)
    assert L.rcg_create(None, ctypes.byref(h)) == N.ERR_BAD_ARG
    if L.rcg_device_count() == 0:
        with pytest.raises(N.NativeError) as ei:
            Engine(EngineConfig(sys_id=N.SYS_3WROBOT, batch=4, pars=[10, 1], ctrl_bnds=[[-300, 300], [-100, 100]]))
        assert ei.value.code == N.ERR_NO_DEVICE
        assert "no CPU fallback" in str(ei.value)


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure; nothing under rcognita_amd/ may reference it."""
    import ast

    pkg = os.path.join(ROOT, "rcognita_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            path = os.path.join(dp, fn)
            if fn.endswith(".py"):
                tree = ast.parse(open(path).read())
                for node in ast.walk(tree):
                    mods = []
                    if isinstance(node, ast.Import):
                        mods = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        mods = [node.module or ""]
                    assert not any(m.split(".")[0] == "oracle" or "rcg_oracle" in m or "c_oracle" in m for m in mods), path
                    if isinstance(node, ast.Constant) and isinstance(node.value, str) and len(node.value) < 200:
                        assert "liboracle" not in node.value and "oracle/_build" not in node.value, path
            elif fn.endswith((".hip", ".hpp", ".h", ".cpp")):
                txt = open(path, errors="replace").read()
                code = re.sub(r"//.*?$|/\*.*?\*/", "", txt, flags=re.S | re.M)  # comments may CITE the oracle
                assert "oracle" not in code, path  # no #include, dlopen or symbol of the oracle in product code


def test_documented_stub_matches_the_header(tmp_path):
    """The `rcg_cfg` ctypes stub a maintainer would copy out of INTEGRATION.md: same fields as the binding, and
    sizeof == sizeof(rcg_cfg) as gcc sees include/rcg.h (rcg_create refuses any other struct_size)."""
    from rcognita_amd import _native as N

    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"# BEGIN rcg_cfg stub.*?\n(class rcg_cfg.*?)\n# END rcg_cfg stub", md, flags=re.S)
    assert m, "INTEGRATION.md lost its generated rcg_cfg stub (tools/gen_integration_stub.py)"
    ns = {"C": ctypes}
    exec(m.group(1), ns)
    doc = ns["rcg_cfg"]
    assert [f[0] for f in doc._fields_] == [f[0] for f in N.RcgCfg._fields_]
    for name, _ in doc._fields_:
        assert getattr(doc, name).offset == getattr(N.RcgCfg, name).offset, name
        assert getattr(doc, name).size == getattr(N.RcgCfg, name).size, name
    src = tmp_path / "size.c"
    src.write_text('#include <stdio.h>\n#include "rcg.h"\nint main(){printf("%zu\\n", sizeof(rcg_cfg));return 0;}')
    exe = tmp_path / "size"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert int(subprocess.check_output([str(exe)]).decode()) == ctypes.sizeof(doc)


def test_the_binding_reads_no_environment_variable():
    """the shipped binding used to honour RCG_LIB.  Tools bind another build explicitly
    (rcognita_amd._native.use_library); nothing under rcognita_amd/ reads os.environ."""
    import glob
    import os

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rcognita_amd")
    for path in glob.glob(os.path.join(root, "*.py")):
        src = open(path).read()
        assert "os.environ" not in src and "getenv" not in src, path
