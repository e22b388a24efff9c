"""In-tree build of libhector_mpc_hip.so (hipcc, gfx950 only; cross-compiles without a GPU).

Staleness is decided by a content hash of the sources (kept next to the library), not by mtimes -- a snapshot copied to
another box keeps the prebuilt library valid -- and builds are serialised by a file lock so that N ranks started by
torchrun never compile into the same file at once."""
from __future__ import annotations

import fcntl
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhector_mpc_hip.so")
SOURCES = ["hmpc_capi.hip", "hmpc_group.hip"]
DEPS = ["hmpc_capi.hip", "hmpc_group.hip", "hmpc_kernel.h", "hmpc_math.h", "hmpc_builder.h", os.path.join("..", "..", "include", "hector_mpc.h")]
# -ffp-contract=off is part of the numerical contract (HMPC-A1): every fused multiply-add in the source is explicit
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value",
         "-Wno-pass-failed", "-ldl"] + os.environ.get("HMPC_EXTRA_FLAGS", "").split()  # developer switches, e.g. -DHMPC_WAVES_PER_EU_256=2


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


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    lock_path = LIB + ".lock"
    with open(lock_path, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():  # another process built it while we waited
                return LIB
            hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
            tmp = LIB + f".tmp{os.getpid()}"
            cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
            with open(_stamp_path(), "w") as f:
                f.write(source_hash())
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
