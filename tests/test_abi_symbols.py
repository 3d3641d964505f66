"""The C-ABI library loads and exports every symbol include/stylish_hip.h declares (no GPU, no compute calls)."""
import os
import re

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from stylish_tts_amd import _lib

    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "stylish_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(stts_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in stylish_hip.h but not exported"
    bound = set(_lib.SIGNATURES) | set(_lib.TEST_SIGNATURES)
    assert declared == bound, declared ^ bound
    # the test surface sits behind STTS_TEST_OPS in the header, and only there
    test_part = re.search(r"#ifdef STTS_TEST_OPS(.*?)#endif", hdr, flags=re.S).group(1)
    assert set(re.findall(r"\b(stts_[a-z0-9_]+)\s*\(", test_part)) == set(_lib.TEST_SIGNATURES)
    assert lib.stts_version() >= 1


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "stylish_tts_amd")
    for d, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"
