"""Disassemble kernels of the SHIPPED libavs_hip.so and check instruction-level properties the protocol depends on.

The direct multi-GPU transport (avs_pcg.hip) orders "halo data before flag" without L2 write-back fences: the writer waits for
its own write-through stores to be acknowledged (`s_waitcnt vmcnt(0)`), then the workgroup barrier, then the ticket / flag.  The
round-2 review found that the wait was not in the instruction stream (a workgroup-scope release fence lowers to lgkmcnt only).
This module extracts the gfx950 code objects from the library's .hip_fatbin section (clang offload bundles, one per translation
unit), disassembles a kernel with llvm-objdump and exposes the instruction list, so that tests/test_isa_ordering.py can assert
the wait is where the protocol needs it.  Runs on the CPU (no GPU needed): llvm-objcopy / llvm-objdump from /opt/rocm.
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "adaptiveviscositysolver_amd", "libavs_hip.so")


def code_objects(lib: str = LIB, arch: str = "gfx950") -> list[bytes]:
    """The device ELF images for `arch` inside the library's fat binary."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(td, "discard.so")])
        blob = open(fat, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), blob):
        base = m.start()
        (nent,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(nent):
            off, size, idlen = struct.unpack_from("<QQQ", blob, pos)
            pos += 24
            ident = blob[pos:pos + idlen].decode()
            pos += idlen
            if arch in ident and size:
                out.append(blob[base + off: base + off + size])
    return out


_cache: dict[str, dict[str, list[str]]] = {}


def disassemble(lib: str = LIB) -> dict[str, list[str]]:
    """symbol -> list of instruction strings (mnemonic + operands, comments stripped), for every function in the library."""
    if lib in _cache:
        return _cache[lib]
    funcs: dict[str, list[str]] = {}
    for img in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", "--no-show-raw-insn", "-C", f.name], text=True)
        cur = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1)
                funcs.setdefault(cur, [])
                continue
            if cur is None or not line.startswith("\t") and not line.startswith(" "):
                continue
            ins = line.split("//")[0].strip()
            if ins:
                funcs[cur].append(ins)
    _cache[lib] = funcs
    return funcs


def kernels_matching(pattern: str, lib: str = LIB) -> dict[str, list[str]]:
    rx = re.compile(pattern)
    return {k: v for k, v in disassemble(lib).items() if rx.search(k)}


def is_remote_store(ins: str) -> bool:
    """A write-through store at system scope (sc0 sc1): what the halo push uses."""
    return bool(re.match(r"^(flat|global)_store_dword", ins)) and " sc0" in ins and " sc1" in ins


def is_agent_store(ins: str) -> bool:
    """A write-through store at agent scope (sc1) or wider."""
    return bool(re.match(r"^(flat|global)_store_dword", ins)) and " sc1" in ins


def is_wide_wt_store(ins: str) -> bool:
    """The resident loop's 16-B write-through stores: the workgroup's slice of u (sc1) and the boundary entries it pushes (sc0 sc1)."""
    return bool(re.match(r"^global_store_dwordx4", ins)) and " sc1" in ins


def waits_vmcnt0(ins: str) -> bool:
    return ins.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", ins) is not None


def check_store_wait_sync(ins: list[str], store_pred, sync_rx: str) -> tuple[bool, str]:
    """After the LAST store matching store_pred there must be an `s_waitcnt vmcnt(0)` before the first instruction matching
    sync_rx (barrier / atomic ticket / flag store) that follows it -- on every straight-line position order of the listing."""
    idx = [i for i, s in enumerate(ins) if store_pred(s)]
    if not idx:
        return False, "no matching store found"
    # every such store must be followed by a vmcnt(0) wait before the next sync instruction after it
    rx = re.compile(sync_rx)
    for i in idx:
        waited = False
        for j in range(i + 1, len(ins)):
            if waits_vmcnt0(ins[j]):
                waited = True
                break
            if rx.match(ins[j]):
                return False, f"store at #{i} `{ins[i]}` reaches `{ins[j]}` (#{j}) without s_waitcnt vmcnt(0)"
            if ins[j].startswith("s_endpgm"):
                break
        if not waited and i == idx[-1]:
            # a store at the very end of a path with no later sync is fine (fire-and-forget slot)
            pass
    return True, "ok"


def run_checks(lib: str = LIB) -> list[tuple[str, bool, str]]:
    """(kernel, ok, message) for every ordering property the protocol relies on."""
    rows = []
    for name, pat, pred, sync in CHECKS:
        ks = kernels_matching(pat, lib)
        if not ks:
            rows.append((name, False, "no kernel matches " + pat))
            continue
        for k, ins in ks.items():
            ok, msg = check_store_wait_sync(ins, pred, sync)
            rows.append((k, ok, msg))
    # halo_finalizer (inlined into every HALO instantiation of the SpMV kernels): stage2 store -> fin_ticket
    nhalo = 0
    for k, ins in disassemble(lib).items():
        if "k_spmv" in k and any(re.match(r"^(global|flat)_atomic_add", s) for s in ins):
            ok, msg = check_store_wait_sync(ins, is_agent_store, r"^(global|flat)_atomic_add\s")
            rows.append((k, ok, msg))
            nhalo += 1
    if nhalo == 0:
        rows.append(("k_spmv_*<HALO>", False, "no SpMV instantiation with a finalizer ticket found"))
    return rows


def main() -> int:
    bad = 0
    for k, ok, msg in run_checks(sys.argv[1] if len(sys.argv) > 1 else LIB):
        print(f"{'ok' if ok else 'FAIL':5s} {k[:110]}  {msg}")
        bad += not ok
    return 1 if bad else 0


# (label, kernel-name regex, store predicate, regex of the synchronising instruction the wait must precede)
# (the 32-bit atomics are the tickets; the 64-bit `_x2` adds are the paranoid mode's checksum accumulation, part of the data)
SYNC = r"^(s_barrier|(global|flat)_atomic_(add|inc)\s)"
CHECKS = [
    ("k_push", r"^avs::k_push\(", is_remote_store, SYNC),
    ("k_sr_update_push", r"^(void )?avs::k_sr_update_push", is_remote_store, SYNC),
    ("k_reduce_mb", r"^avs::k_reduce_mb\(", is_agent_store, r"^(global|flat)_atomic_add\s"),
    # CU-resident loop: u slice + pushed entries acknowledged before the barrier behind which the producer flag / the ticket is raised
    ("k_cg_resident", r"^(void )?avs::k_cg_resident<", is_wide_wt_store, SYNC),
]

if __name__ == "__main__":
    sys.exit(main())
