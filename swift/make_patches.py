#!/usr/bin/env python3
"""Regenerates swift/patches/*.patch — the reference-side changes that drop the HIP backend into Wax — from a pristine
checkout of christopherkarani/Wax (default /root/reference). Development tool: the committed patches are what a
maintainer applies (`git apply swift/patches/000*.patch` in the Wax checkout); tests/test_swift_patches.py verifies
that the series applies cleanly to the reference, in order.

Each step edits a scratch copy and records `git diff` (2 lines of context). The edits touch exactly the selection /
construction sites SURVEY.md §8(b) lists:
  0001  Package.swift: CWaxHIP system-library target (module map over include/wax_hip.h, links libwaxhip), Linux only
  0002  Sources/CWaxHIP/ (module map + shim header) and Sources/WaxVectorSearch/HIPVectorEngine.swift (the actor)
  0003  VectorEnginePreference.hipPreferred (VectorSearchEngine.swift:4-8)
  0004  UnifiedSearchEngineCache: VectorEngineKind.hip, selection, construction, deserialize (:42-45, 94-152)
  0005  WaxSession: ConcreteVectorEngine.hip + loadVectorEngine (:7-37, 478-498)
  0006  WaxVectorSearchSession: ConcreteVectorEngine.hip + init + the five forwarding switches (:6-9, 29-54, 198-241)
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
FILES = ["Package.swift", "Sources/WaxVectorSearch/VectorSearchEngine.swift",
         "Sources/Wax/UnifiedSearch/UnifiedSearchEngineCache.swift", "Sources/Wax/WaxSession.swift",
         "Sources/Wax/VectorSearchSession.swift"]


def sub(path, old, new, count=1):
    text = open(path).read()
    assert text.count(old) == count, (path, text.count(old), old[:60])
    open(path, "w").write(text.replace(old, new))


def git(cwd, *args):
    return subprocess.run(["git", "-c", "user.name=wax-amd", "-c", "user.email=wax-amd@localhost", *args], cwd=cwd, check=True,
                          capture_output=True, text=True).stdout


def main():
    work = tempfile.mkdtemp(prefix="wax_patches_")
    for f in FILES:
        os.makedirs(os.path.dirname(os.path.join(work, f)) or work, exist_ok=True)
        shutil.copy(os.path.join(REF, f), os.path.join(work, f))
    git(work, "init", "-q")
    git(work, "add", "-A")
    git(work, "commit", "-q", "-m", "reference")
    out_dir = os.path.join(HERE, "patches")
    os.makedirs(out_dir, exist_ok=True)
    for old in os.listdir(out_dir):
        if old.endswith(".patch"):
            os.unlink(os.path.join(out_dir, old))
    steps = []

    def step(name, title):
        git(work, "add", "-A")
        diff = git(work, "diff", "--cached", "-U2", "--no-color")
        assert diff.strip(), name
        header = f"# {title}\n# Apply in the Wax checkout with `git apply` (series order).\n"
        open(os.path.join(out_dir, name), "w").write(header + diff)
        git(work, "commit", "-q", "-m", name)
        steps.append(name)

    P = lambda f: os.path.join(work, f)  # noqa: E731

    # 0001 Package.swift --------------------------------------------------------------------------------------------
    sub(P("Package.swift"), '''        .target(
            name: "WaxVectorSearch",
            dependencies: [
                "WaxCore",
                .product(name: "USearch", package: "USearch"),
            ],''', '''        // libwaxhip (MI355X / gfx950 brute-force scan + top-k): C ABI in include/wax_hip.h, bound through a module map.
        // Linux only; the library is found through the usual linker search path (or -Xlinker -L<dir of libwaxhip.so>).
        .systemLibrary(
            name: "CWaxHIP",
            path: "Sources/CWaxHIP"
        ),
        .target(
            name: "WaxVectorSearch",
            dependencies: [
                "WaxCore",
                .product(name: "USearch", package: "USearch"),
                .target(name: "CWaxHIP", condition: .when(platforms: [.linux])),
            ],''')
    step("0001-package-cwaxhip-system-library.patch", "Package.swift: CWaxHIP system-library target, linked into WaxVectorSearch on Linux")

    # 0002 new files ------------------------------------------------------------------------------------------------
    os.makedirs(P("Sources/CWaxHIP"), exist_ok=True)
    open(P("Sources/CWaxHIP/module.modulemap"), "w").write(
        'module CWaxHIP [system] {\n    header "shim.h"\n    link "waxhip"\n    export *\n}\n')
    open(P("Sources/CWaxHIP/shim.h"), "w").write(
        "/* The C ABI of libwaxhip; install include/wax_hip.h from the wax_amd repository next to this file\n"
        " * (or add its directory with -Xcc -I). */\n#include \"wax_hip.h\"\n")
    shutil.copy(os.path.join(HERE, "HIPVectorEngine.swift"), P("Sources/WaxVectorSearch/HIPVectorEngine.swift"))
    step("0002-cwaxhip-module-and-hip-vector-engine.patch", "Sources/CWaxHIP module map + Sources/WaxVectorSearch/HIPVectorEngine.swift (actor conforming to VectorSearchEngine)")

    # 0003 preference -----------------------------------------------------------------------------------------------
    sub(P("Sources/WaxVectorSearch/VectorSearchEngine.swift"), "    case cpuOnly\n}", '''    case cpuOnly
    /// Prefer the MI355X HIP backend (libwaxhip) when a gfx950 GPU is visible. `.auto` also selects it when present.
    case hipPreferred
}''')
    step("0003-vector-engine-preference-hip.patch", "VectorEnginePreference.hipPreferred")

    # 0004 UnifiedSearchEngineCache ---------------------------------------------------------------------------------
    f = P("Sources/Wax/UnifiedSearch/UnifiedSearchEngineCache.swift")
    sub(f, "        case usearch\n        case metal\n    }", "        case usearch\n        case metal\n        case hip\n    }")
    sub(f, '''        let allowMetal = preference != .cpuOnly && MetalVectorEngine.isAvailable

        if allowMetal {''', '''        let allowMetal = preference != .cpuOnly && MetalVectorEngine.isAvailable

        #if canImport(CWaxHIP)
        // MI355X: the HIP engine is tried first whenever a gfx950 GPU is visible (libwaxhip has no CPU fallback, so
        // `isAvailable` is the whole probe); any failure falls through to Metal / USearch exactly like Metal's does.
        if preference != .cpuOnly && preference != .metalPreferred && HIPVectorEngine.isAvailable {
            if let hipEngine = try await vectorEngine(
                for: wax,
                waxId: waxId,
                queryEmbeddingDimensions: queryEmbeddingDimensions,
                engineKind: .hip
            ) {
                return hipEngine
            }
        }
        #endif

        if allowMetal {''')
    sub(f, '''            if preferMetal {
                return try MetalVectorEngine(metric: metric, dimensions: dimensions)
            }''', '''            #if canImport(CWaxHIP)
            if engineKindTag == .hip {
                return try HIPVectorEngine(metric: metric, dimensions: dimensions)
            }
            #endif
            if preferMetal {
                return try MetalVectorEngine(metric: metric, dimensions: dimensions)
            }''')
    sub(f, '''            case .usearch:
                guard let usearch = engine as? USearchVectorEngine else {''', '''            case .hip:
                #if canImport(CWaxHIP)
                guard let hip = engine as? HIPVectorEngine else {
                    throw WaxError.invalidToc(reason: "hip engine type mismatch")
                }
                try await hip.deserialize(bytes)   // MV2V encoding 2, the same segment MetalVectorEngine writes
                #else
                throw WaxError.invalidToc(reason: "hip engine not built into this binary")
                #endif
            case .usearch:
                guard let usearch = engine as? USearchVectorEngine else {''')
    step("0004-unified-search-engine-cache-hip.patch", "UnifiedSearchEngineCache: VectorEngineKind.hip, selection before Metal, construction, deserialize")

    # 0005 WaxSession -----------------------------------------------------------------------------------------------
    f = P("Sources/Wax/WaxSession.swift")
    sub(f, "        case usearch(USearchVectorEngine)\n        case metal(MetalVectorEngine)\n", '''        case usearch(USearchVectorEngine)
        case metal(MetalVectorEngine)
        #if canImport(CWaxHIP)
        case hip(HIPVectorEngine)
        #endif
''')
    sub(f, '''            case .metal(let engine):
                return engine
            }''', '''            case .metal(let engine):
                return engine
            #if canImport(CWaxHIP)
            case .hip(let engine):
                return engine
            #endif
            }''')
    sub(f, '''            case .metal(let engine):
                try await engine.addBatch(frameIds: frameIds, vectors: vectors)
            }''', '''            case .metal(let engine):
                try await engine.addBatch(frameIds: frameIds, vectors: vectors)
            #if canImport(CWaxHIP)
            case .hip(let engine):
                try await engine.addBatch(frameIds: frameIds, vectors: vectors)
            #endif
            }''')
    sub(f, '''            case .metal(let engine):
                try await engine.stageForCommit(into: wax)
            }''', '''            case .metal(let engine):
                try await engine.stageForCommit(into: wax)
            #if canImport(CWaxHIP)
            case .hip(let engine):
                try await engine.stageForCommit(into: wax)
            #endif
            }''')
    sub(f, '''    ) async throws -> ConcreteVectorEngine {
        if preference != .cpuOnly, MetalVectorEngine.isAvailable {''', '''    ) async throws -> ConcreteVectorEngine {
        #if canImport(CWaxHIP)
        if preference != .cpuOnly, preference != .metalPreferred, HIPVectorEngine.isAvailable {
            do {
                let hip = try await HIPVectorEngine.load(from: wax, metric: metric, dimensions: dimensions)
                return .hip(hip)
            } catch {
                WaxDiagnostics.logSwallowed(
                    error,
                    context: "hip vector engine load",
                    fallback: "use Metal / CPU vector engine"
                )
            }
        }
        #endif
        if preference != .cpuOnly, MetalVectorEngine.isAvailable {''')
    step("0005-wax-session-hip.patch", "WaxSession: ConcreteVectorEngine.hip and loadVectorEngine")

    # 0006 WaxVectorSearchSession -----------------------------------------------------------------------------------
    f = P("Sources/Wax/VectorSearchSession.swift")
    sub(f, "        case usearch(USearchVectorEngine)\n        case metal(MetalVectorEngine)\n    }", '''        case usearch(USearchVectorEngine)
        case metal(MetalVectorEngine)
        #if canImport(CWaxHIP)
        case hip(HIPVectorEngine)
        #endif
    }''')
    sub(f, '''        let loadedEngine: ConcreteVectorEngine
        if preference != .cpuOnly, MetalVectorEngine.isAvailable {''', '''        var loadedEngine: ConcreteVectorEngine
        var hipLoaded: ConcreteVectorEngine?
        #if canImport(CWaxHIP)
        if preference != .cpuOnly, preference != .metalPreferred, HIPVectorEngine.isAvailable {
            do {
                hipLoaded = .hip(try await HIPVectorEngine.load(from: wax, metric: metric, dimensions: dimensions))
            } catch {
                WaxDiagnostics.logSwallowed(
                    error,
                    context: "hip vector engine load",
                    fallback: "use Metal / CPU vector engine"
                )
            }
        }
        #endif
        if let hipLoaded {
            loadedEngine = hipLoaded
        } else if preference != .cpuOnly, MetalVectorEngine.isAvailable {''')
    sub(f, '''        case .metal(let engine):
            self.engine = engine
        }''', '''        case .metal(let engine):
            self.engine = engine
        #if canImport(CWaxHIP)
        case .hip(let engine):
            self.engine = engine
        #endif
        }''')
    for call in ("try await engine.add(frameId: frameId, vector: vector)",
                 "try await engine.addBatch(frameIds: frameIds, vectors: vectors)",
                 "try await engine.remove(frameId: frameId)",
                 "return try await engine.search(vector: vector, topK: topK)",
                 "try await engine.stageForCommit(into: wax)"):
        sub(f, f'''        case .metal(let engine):
            {call}
        }}''', f'''        case .metal(let engine):
            {call}
        #if canImport(CWaxHIP)
        case .hip(let engine):
            {call}
        #endif
        }}''')
    step("0006-vector-search-session-hip.patch", "WaxVectorSearchSession: ConcreteVectorEngine.hip, init, forwarding switches")

    open(os.path.join(out_dir, "series"), "w").write("\n".join(steps) + "\n")
    shutil.rmtree(work)
    print("wrote", len(steps), "patches to", out_dir)


if __name__ == "__main__":
    main()
