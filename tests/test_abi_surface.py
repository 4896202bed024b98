"""The C-ABI library loads (no GPU needed) and exports exactly what include/vrnet_hip.h declares."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    import __graft_entry__ as g
    g.build()
    import asy_vrnet_amd.hip as hip
    text = open(os.path.join(ROOT, "include", "vrnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(vrnet_\w+)\s*\(", text))
    assert declared == set(hip.EXPORTED), declared ^ set(hip.EXPORTED)
    for name in declared:
        assert hasattr(hip._lib, name)
    assert hip._lib.vrnet_abi_version() == hip.ABI_VERSION


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "asy-vrnet_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("no oracle", ""), fn
