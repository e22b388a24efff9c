"""In-tree build of libhector_mpc_hip.so (hipcc, gfx950 only; cross-compiles without a GPU).

The kernel family is compiled as HMPC_VARIANT_GROUPS translation units side by side (csrc/hmpc_variants.hip with
-DHMPC_VARIANT_GROUP=k) next to the two host-side ones, then linked: ~25 s instead of the 60 s of one serial unit.
Staleness is decided by a content hash of the sources (kept next to the library), not by mtimes -- a snapshot copied to
another box keeps the prebuilt library valid -- and builds are serialised by a file lock so that N ranks started by
torchrun never compile into the same file at once."""
from __future__ import annotations

import concurrent.futures
import fcntl
import hashlib
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhector_mpc_hip.so")
VARIANT_GROUPS = 4  # = HMPC_VARIANT_GROUPS of csrc/hmpc_variants.h
HOST_SOURCES = ["hmpc_capi.hip", "hmpc_group.hip"]
DEPS = ["hmpc_capi.hip", "hmpc_group.hip", "hmpc_variants.hip", "hmpc_variants.h", "hmpc_kernel_args.h", "hmpc_kernel.h",
        "hmpc_math.h", "hmpc_builder.h", os.path.join("..", "..", "include", "hector_mpc.h")]
# -ffp-contract=off is part of the numerical contract (HMPC-A1): every fused multiply-add in the source is explicit
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed"]
EXTRA = os.environ.get("HMPC_EXTRA_FLAGS", "").split()  # developer A/B switches, e.g. -DHMPC_MFMA_SWEEP=0 (same results, other code)
FLAGS = CFLAGS + EXTRA


# developer switches that make the library WRONG for a user by design: -DHMPC_DEBUG_STATS writes per-solve debug counters where the
# objective value goes (scripts/dev/cont_probe.py), -DHMPC_PROFILE adds clock reads to every phase
DEV_ONLY_FLAGS = ("-DHMPC_DEBUG_STATS", "-DHMPC_PROFILE")


def _check_flags() -> None:
    """Developer-only switches never reach the library the package loads, unless the developer says so explicitly
    (HMPC_ALLOW_DEV_BUILD=1: the throw-away builds of scripts/gpu_*.sh, which restore the product library afterwards)."""
    bad = [f for f in EXTRA if f.startswith(DEV_ONLY_FLAGS)]
    if bad and os.environ.get("HMPC_ALLOW_DEV_BUILD") != "1":
        raise RuntimeError(f"HMPC_EXTRA_FLAGS contains {bad} (developer-only: wrong outputs by design): refused for the product "
                           "library; set HMPC_ALLOW_DEV_BUILD=1 for a throw-away build")


def source_hash() -> str:
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in DEPS:
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp_path() -> str:
    return LIB + ".srchash"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    try:
        with open(_stamp_path()) as f:
            return f.read().strip() != source_hash()
    except OSError:
        return True


def compile_commands(objdir: str, hipcc: str) -> list:
    """(object path, command) of every translation unit."""
    units = []
    for s in HOST_SOURCES:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
        units.append((o, [hipcc] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]))
    for g in range(VARIANT_GROUPS):
        o = os.path.join(objdir, f"hmpc_variants_{g}.o")
        units.append((o, [hipcc] + FLAGS + [f"-DHMPC_VARIANT_GROUP={g}", "-c", os.path.join(CSRC, "hmpc_variants.hip"), "-o", o]))
    return units


def build_to(out: str, extra_compile_flags: list | None = None, verbose: bool = False) -> str:
    """A developer copy of the library (profiling / timing builds) at `out`; the product library and its stamp are not touched."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory(prefix="hmpc_obj_") as objdir:
        # (extra flags go AFTER the product's, so that a -D here overrides one in HMPC_EXTRA_FLAGS)
        units = [(o, c[:-4] + list(extra_compile_flags or []) + c[-4:]) for o, c in compile_commands(objdir, hipcc)]  # (... -c src -o obj)

        def run(unit):
            if verbose:
                print(" ".join(unit[1]), flush=True)
            subprocess.check_call(unit[1])
            return unit[0]

        with concurrent.futures.ThreadPoolExecutor(max_workers=len(units)) as ex:
            objs = list(ex.map(run, units))
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", out])
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    _check_flags()  # (before the staleness shortcut: a library built earlier with developer flags must not be loaded silently either)
    if not force and not needs_build():
        return LIB
    lock_path = LIB + ".lock"
    with open(lock_path, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():  # another process built it while we waited
                return LIB
            tmp = LIB + f".tmp{os.getpid()}"
            build_to(tmp, verbose=verbose)
            os.replace(tmp, LIB)
            with open(_stamp_path(), "w") as f:
                f.write(source_hash())
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
