"""Build libwaxhip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.

In-tree build: the .so lands in wax_amd/lib/ so that it travels to the GPU box
with the repo snapshot. hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libwaxhip.so")
SOURCES = ["kernels.hip", "multiscan.hip", "batch.hip", "filter.hip", "rrf.hip", "engine.hip"]
HEADERS = ["common.h", "topk.h", "kernels.h", "engine_types.inc", "store_internal.inc", "search_internal.inc", "batch_host.inc", "api_store.inc",
           "api_search.inc", "api_batch.inc", "api_shard.inc", "api_fusion_filter.inc", "codec.inc", "tuning.inc", "sharded.inc",
           os.path.join(ROOT, "include", "wax_hip.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libwaxhip for gfx950)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile (if needed) and return the library path. Safe when several ranks of one job call it at the
    same time: builds are serialised by a file lock and the .so is replaced atomically, so a process never
    dlopens a half-written file."""
    if not force and not needs_build():
        return LIB_PATH
    import fcntl
    import tempfile
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():  # another process built it while we waited
                return LIB_PATH
            fd, tmp = tempfile.mkstemp(prefix=".libwaxhip.", suffix=".so", dir=LIB_DIR)
            os.close(fd)
            # one object per translation unit, compiled side by side and only when stale (a unit depends on its own source
            # and on every header: the headers are few and shared), then one link
            obj_dir = os.path.join(LIB_DIR, "obj")
            os.makedirs(obj_dir, exist_ok=True)
            hdr_t = max(os.path.getmtime(h if os.path.isabs(h) else os.path.join(CSRC, h)) for h in HEADERS)
            flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include")]
            # the object cache is valid for ONE (compiler, flags) pair: a stamp file names it, anything else rebuilds every unit
            try:
                ver = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True, check=False).stdout
            except OSError:
                ver = ""
            import hashlib
            stamp = hashlib.sha256((ver + "\n" + " ".join(flags)).encode()).hexdigest()
            stamp_path = os.path.join(obj_dir, ".stamp")
            try:
                stale = open(stamp_path).read().strip() != stamp
            except OSError:
                stale = True
            if stale:
                for f_ in os.listdir(obj_dir):
                    if f_.endswith(".o"):
                        os.unlink(os.path.join(obj_dir, f_))
                with open(stamp_path, "w") as sf:
                    sf.write(stamp)

            def compile_one(src):
                obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
                path = os.path.join(CSRC, src)
                if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_t):
                    return obj
                cmd = [_hipcc()] + flags + ["-c", path, "-o", obj + ".tmp"]
                if verbose:
                    print(" ".join(cmd), file=sys.stderr)
                subprocess.run(cmd, check=True)
                os.replace(obj + ".tmp", obj)
                return obj

            try:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 2)) as pool:
                    objs = list(pool.map(compile_one, SOURCES))
                cmd = [_hipcc(), f"--offload-arch={ARCH}", "-fPIC", "-shared", "-o", tmp] + objs
                if verbose:
                    print(" ".join(cmd), file=sys.stderr)
                subprocess.run(cmd, check=True)
                os.chmod(tmp, 0o755)
                os.replace(tmp, LIB_PATH)
            finally:
                if os.path.exists(tmp):
                    os.unlink(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
