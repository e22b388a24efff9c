"""The compile-time switches the kernel still has (<= 10, each with a test that builds its other side)
cannot rot: every one of them is compiled with a non-default value -- front end and template instantiation of the whole kernel
family for gfx950, no code generation, no GPU.  Round 6 deleted the 20-odd A/B switches whose other side had been measured worse."""
import concurrent.futures
import os
import re
import subprocess

from hector_simulation_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "hector_simulation_amd", "csrc", "hmpc_variants.hip")

# switch -> a value other than its default
SWITCHES = {"HMPC_REFINE": "2", "HMPC_CONT_ROUNDS": "2", "HMPC_READD_LIMIT": "0", "HMPC_CONT_ITER_BUDGET": "64",
            "HMPC_EARLY_HANDOVER_MARGIN": "4", "HMPC_QCAP_FAST": "68"}
DEV_BUILDS = ["-DHMPC_DEBUG_STATS", "-DHMPC_PROFILE"]


def _syntax_only(group: int, flags: list) -> subprocess.CompletedProcess:
    cmd = ["/opt/rocm/bin/hipcc"] + build.CFLAGS + flags + [f"-DHMPC_VARIANT_GROUP={group}", "--cuda-device-only", "-fsyntax-only", SRC]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600)


def test_the_switch_list_is_what_the_sources_have():
    macros = set()
    for fn in ("hmpc_kernel.h", "hmpc_variants.h"):
        txt = open(os.path.join(ROOT, "hector_simulation_amd", "csrc", fn)).read()
        macros |= set(re.findall(r"^#ifndef (HMPC_[A-Z0-9_]+)\n#define \1 ", txt, flags=re.M))
    macros -= {"HMPC_QCAP_CONT"}  # (the continuation variant's capacity: fixed by the hand-over layout, 96 = 6 tiles of 16 rows)
    assert macros == set(SWITCHES), (sorted(macros), sorted(SWITCHES))
    assert len(macros) <= 10


def test_every_switch_compiles_with_a_non_default_value():
    # two front-end passes over every variant group: all tuning switches at once; the two developer builds at once with the other
    # extreme of the round / refinement counts
    sets = [[f"-D{k}={v}" for k, v in SWITCHES.items()],
            DEV_BUILDS + ["-DHMPC_REFINE=0", "-DHMPC_CONT_ROUNDS=0"]]
    jobs = [(g, fl) for fl in sets for g in range(build.VARIANT_GROUPS)]
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(lambda j: _syntax_only(*j), jobs))
    for (g, fl), r in zip(jobs, res):
        assert r.returncode == 0, (g, fl, r.stderr[-1500:])
