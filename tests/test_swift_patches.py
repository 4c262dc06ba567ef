"""The reference-side drop-in (swift/patches/*.patch) applies cleanly, in series order, to the files of the
christopherkarani/Wax checkout it was generated from (VectorEnginePreference, UnifiedSearchEngineCache, WaxSession,
WaxVectorSearchSession, Package.swift) and adds the CWaxHIP module + HIPVectorEngine.swift. There is no Swift
toolchain in this image, so the patched tree is not compiled; what IS verified: `git apply --check` of every patch,
that each selection site gained its `.hip` case, and that the shipped HIPVectorEngine.swift is the file the series adds.
Runs only where /root/reference exists (this container; not the GPU box)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PATCHES = os.path.join(ROOT, "swift", "patches")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "Sources")), reason="reference checkout not present")
def test_patch_series_applies_to_the_reference(tmp_path):
    series = open(os.path.join(PATCHES, "series")).read().split()
    assert len(series) == 6
    work = tmp_path / "wax"
    for rel in ("Package.swift", "Sources/WaxVectorSearch", "Sources/Wax/UnifiedSearch", "Sources/Wax/WaxSession.swift",
                "Sources/Wax/VectorSearchSession.swift"):
        src, dst = os.path.join(REF, rel), work / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        (shutil.copytree if os.path.isdir(src) else shutil.copy)(src, dst)
    for name in series:
        patch = os.path.join(PATCHES, name)
        chk = subprocess.run(["git", "apply", "--check", "--verbose", patch], cwd=work, capture_output=True, text=True)
        assert chk.returncode == 0, f"{name}: {chk.stderr}"
        subprocess.run(["git", "apply", patch], cwd=work, check=True)
    text = lambda rel: (work / rel).read_text()  # noqa: E731
    assert "case hipPreferred" in text("Sources/WaxVectorSearch/VectorSearchEngine.swift")
    cache = text("Sources/Wax/UnifiedSearch/UnifiedSearchEngineCache.swift")
    assert "case hip\n" in cache and "engineKind: .hip" in cache and "HIPVectorEngine(metric: metric, dimensions: dimensions)" in cache
    assert "try await hip.deserialize(bytes)" in cache
    sess = text("Sources/Wax/WaxSession.swift")
    assert sess.count("case .hip(let engine):") == 3 and "HIPVectorEngine.load(from: wax" in sess
    vss = text("Sources/Wax/VectorSearchSession.swift")
    assert vss.count("case .hip(let engine):") == 6 and "HIPVectorEngine.load(from: wax" in vss
    pkg = text("Package.swift")
    assert '.systemLibrary(\n            name: "CWaxHIP"' in pkg and '.target(name: "CWaxHIP", condition: .when(platforms: [.linux]))' in pkg
    assert 'link "waxhip"' in text("Sources/CWaxHIP/module.modulemap")
    # the engine the series installs is the one kept (and reviewed) in this repository
    assert text("Sources/WaxVectorSearch/HIPVectorEngine.swift") == open(os.path.join(ROOT, "swift", "HIPVectorEngine.swift")).read()
    # every switch over the concrete engines is still balanced: one #if / #endif pair per added case
    for f in (sess, vss, cache):
        assert f.count("#if canImport(CWaxHIP)") == f.count("#endif")


def test_swift_shim_matches_the_c_abi():
    """Every wax_hip_* call in the Swift shim exists in include/wax_hip.h with the same number of arguments."""
    import re
    header = open(os.path.join(ROOT, "include", "wax_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    decl = {}
    for m in re.finditer(r"\b(wax_hip_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S):
        args = m.group(2).strip()
        decl[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    swift = open(os.path.join(ROOT, "swift", "HIPVectorEngine.swift")).read()
    calls = 0
    for m in re.finditer(r"\b(wax_hip_[a-z_0-9]+)\(", swift):
        name = m.group(1)
        assert name in decl, f"{name} is not declared in wax_hip.h"
        i, depth, nargs, seen = m.end(), 1, 0, False
        while depth:
            c = swift[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
            elif c == "," and depth == 1:
                nargs += 1
            if not c.isspace() and depth >= 1 and c != ")":
                seen = True
            i += 1
        nargs = nargs + 1 if seen else 0
        assert nargs == decl[name], f"{name}: Swift passes {nargs} arguments, header declares {decl[name]}"
        calls += 1
    assert calls >= 15
