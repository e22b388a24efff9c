#!/usr/bin/env python3
"""Per-kernel hash of the gfx950 machine code of every kernel variant (hipcc cross-compiles: no GPU needed).

    python scripts/isa_hash.py [--out FILE] [--group 0..3 ...] [extra hipcc flags]
    python scripts/isa_hash.py --diff profiles/r06/isa_hash_before.txt      # exit code 1 when any kernel's code changed

What a refactor of hmpc_kernel.h that must not change the product (macro removal, a stage moved into a function) is checked
with: the kernel's instructions as llvm-objdump prints them (addresses and symbol-relative branch targets stripped), hashed
per kernel symbol.  Two builds with the same hash run the same instructions."""
import concurrent.futures
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hector_simulation_amd import build as hip_build  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_hashes(groups=None, extra=()):
    groups = list(range(hip_build.VARIANT_GROUPS)) if not groups else groups
    out = {}
    with tempfile.TemporaryDirectory(prefix="hmpc_isa_") as td:
        def one(g):
            co = os.path.join(td, f"g{g}.bundle")
            elf = os.path.join(td, f"g{g}.elf")
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + hip_build.CFLAGS + list(extra) +
                                  [f"-DHMPC_VARIANT_GROUP={g}", "--cuda-device-only", "-c",
                                   os.path.join(hip_build.CSRC, "hmpc_variants.hip"), "-o", co], stderr=subprocess.DEVNULL)
            subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={co}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={elf}"])
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", elf],
                                 capture_output=True, text=True, check=True).stdout
            res, name, h, n = {}, None, None, 0
            for line in dis.splitlines():
                m = re.match(r"^<(\S+)>:$", line.strip())
                if m:
                    if name:
                        res[name] = (h.hexdigest()[:16], n)
                    name, h, n = m.group(1), hashlib.sha256(), 0
                    continue
                if name and line.strip():
                    txt = re.sub(r"//.*$", "", line).strip()          # trailing address comments
                    txt = re.sub(r"<[^>]*>", "", txt)                   # symbol+offset annotations of branch targets
                    h.update(txt.encode() + b"\n")
                    n += 1
            if name:
                res[name] = (h.hexdigest()[:16], n)
            return res

        with concurrent.futures.ThreadPoolExecutor(max_workers=len(groups)) as ex:
            for r in ex.map(one, groups):
                out.update(r)
    short = {}
    for k, v in out.items():
        d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        d = re.sub(r"^void hmpc::", "", d).replace("(hmpc::KernelArgs)", "")
        short[d] = v
    return short


def main():
    args = sys.argv[1:]
    out_path = diff_path = None
    groups, extra = [], []
    i = 0
    while i < len(args):
        if args[i] == "--out":
            out_path = args[i + 1]; i += 2
        elif args[i] == "--diff":
            diff_path = args[i + 1]; i += 2
        elif args[i] == "--group":
            groups.append(int(args[i + 1])); i += 2
        else:
            extra.append(args[i]); i += 1
    hs = kernel_hashes(groups, extra)
    lines = [f"{h} {n:7d} {k}" for k, (h, n) in sorted(hs.items())]
    text = "\n".join(lines) + "\n"
    if out_path:
        with open(out_path, "w") as f:
            f.write(text)
    print(text, end="")
    if diff_path:
        old = {}
        for line in open(diff_path):
            p = line.split(None, 2)
            if len(p) == 3:
                old[p[2].strip()] = p[0]
        changed = [k for k, (h, _) in hs.items() if k in old and old[k] != h]
        missing = [k for k in old if k not in hs and not groups]
        new = [k for k in hs if k not in old]
        for k in changed:
            print("CHANGED", k)
        for k in missing:
            print("MISSING", k)
        for k in new:
            print("NEW", k)
        print(f"{len(hs) - len(changed) - len(new)} identical, {len(changed)} changed, {len(new)} new, {len(missing)} missing")
        sys.exit(1 if (changed or missing) else 0)


if __name__ == "__main__":
    main()
