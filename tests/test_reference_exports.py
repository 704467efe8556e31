"""Drop-in boundary, SURVEY.md section 8(b)(3): every public name of the reference package imports from this one.

The reference's export list is read from /root/reference where that tree exists (this container) and from the committed
copy tests/golden/reference_exports.txt everywhere; the two must agree, and every name must resolve - to an implemented
op or to the lazy stub that raises NotImplementedError on call."""
import ast
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_INIT = "/root/reference/python/sgl_kernel/__init__.py"
OUR_INIT = os.path.join(ROOT, "sgl-kernel-xpu_amd", "python", "sgl_kernel", "__init__.py")


def _reference_names(path):
    names = set()
    for node in ast.walk(ast.parse(open(path).read())):
        if isinstance(node, ast.ImportFrom) and (node.module or "").startswith("sgl_kernel"):
            names.update(a.asname or a.name for a in node.names if a.name != "*")
    return names


def _fixture_names():
    with open(os.path.join(HERE, "golden", "reference_exports.txt")) as f:
        return {ln.strip() for ln in f if ln.strip() and not ln.startswith("#")}


def _our_names():
    """(implemented, stubbed) read from the source: importing the package needs the built extension"""
    src = open(OUR_INIT).read()
    implemented = set()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.ImportFrom) and (node.module or "").startswith("sgl_kernel"):
            implemented.update(a.asname or a.name for a in node.names)
    m = re.search(r'_OUT_OF_SCOPE = frozenset\(\s*"""(.*?)"""', src, re.S)
    return implemented, set(m.group(1).split())


def test_fixture_matches_the_reference():
    if not os.path.exists(REF_INIT):
        pytest.skip("reference tree not present on this box")
    assert _reference_names(REF_INIT) == _fixture_names()


def test_every_reference_name_resolves():
    implemented, stubbed = _our_names()
    missing = sorted(_fixture_names() - implemented - stubbed)
    assert not missing, "reference names that raise ImportError here: %s" % missing
    assert not (implemented & stubbed), "names both implemented and stubbed: %s" % sorted(implemented & stubbed)
    stale = sorted(stubbed - _fixture_names())
    assert not stale, "stubs for names the reference does not export: %s" % stale


def test_stub_raises_not_implemented():
    try:
        import sgl_kernel
    except ImportError:
        pytest.skip("extension not built")
    fn = sgl_kernel.causal_conv1d
    with pytest.raises(NotImplementedError):
        fn()
    with pytest.raises(AttributeError):
        sgl_kernel.not_a_reference_name
