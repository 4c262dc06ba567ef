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


def _closure_body(text, open_brace):
    """text[open_brace] == '{': returns (body, index after the matching '}')."""
    depth, i = 0, open_brace
    while True:
        c = text[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return text[open_brace + 1:i], i + 1
        i += 1


def test_swift_shim_survives_strict_concurrency_lint():
    """What can be checked without a Swift toolchain about Swift 6 strict concurrency (Package.swift tools-version 6.2,
    `StrictConcurrency` on every target): `BlockingIOExecutor.run` takes a @Sendable closure
    (BlockingIOExecutor.swift:19), so nothing non-Sendable may be captured by an `io.run { ... }` closure — in
    particular not the engine's OpaquePointer (SE-0331) nor any Unsafe*Pointer local; the handle crosses as a
    `final class ...: @unchecked Sendable` wrapper (the reference's own pattern, USearchSendable.swift:6), and the actor
    has no `deinit` reading non-Sendable state."""
    import re
    swift = open(os.path.join(ROOT, "swift", "HIPVectorEngine.swift")).read()
    code = re.sub(r"//[^\n]*", "", swift)
    m = re.search(r"final class (\w+): @unchecked Sendable \{", code)
    assert m, "the handle wrapper must be a final class marked @unchecked Sendable"
    wrapper = m.group(1)
    body, _ = _closure_body(code, m.end() - 1)
    assert "let raw: OpaquePointer" in body and "deinit { wax_hip_engine_destroy(raw) }" in body
    a = re.search(r"public actor HIPVectorEngine \{", code)
    actor, _ = _closure_body(code, a.end() - 1)
    assert f"private let handle: {wrapper}" in actor
    assert "deinit" not in actor, "an actor deinit is nonisolated: it may not touch the (non-Sendable) raw pointer"
    assert "OpaquePointer" not in re.sub(r"var h: OpaquePointer\?", "", actor), \
        "the only OpaquePointer in the actor is the out-parameter local of the two inits"
    # every function that hops onto the executor: its locals ahead of the closure are Sendable values
    runs = [m.end() - 1 for m in re.finditer(r"io\.run \{", actor)]
    assert len(runs) >= 9
    pointer_type = re.compile(r"\b(OpaquePointer|Unsafe(Mutable)?(Raw)?(Buffer)?Pointer)\b")
    for at in runs:
        start = actor.rfind("func ", 0, at)
        prologue = actor[start:at]
        assert not pointer_type.search(prologue), f"pointer-typed local ahead of io.run: {prologue[-200:]}"
        for decl in re.finditer(r"\b(?:let|var) (\w+) = handle\b", prologue):
            name = decl.group(1)
            closure, _ = _closure_body(actor, at)
            # the wrapper is only ever dereferenced as <name>.raw, inside the closure
            assert re.search(rf"\b{name}\.raw\b", closure), f"{name} captured but never used"
            assert not re.search(rf"\b{name}\b(?!\.raw)", closure), f"{name} used as a raw pointer inside a @Sendable closure"
        closure, _ = _closure_body(actor, at)
        assert not re.search(r"\bhandle\b", closure), "actor-isolated property read from a @Sendable closure"
        assert not re.search(r"\b(dirty|metric)\b", closure), "actor-isolated state read from a @Sendable closure"
