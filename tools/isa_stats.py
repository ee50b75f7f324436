#!/usr/bin/env python3
"""Per-kernel instruction statistics of the gfx950 code inside libfyrox_hip.so (no GPU needed): extracts the offload
bundles with llvm-objdump, disassembles them and counts the instructions that matter for PARITY -- the reference's
arithmetic is unfused IEEE f32, so a kernel that promises bit-exact results must not contain a contracted multiply-add.
What legitimately remains in such a kernel: the compiler's own IEEE expansions, which are built from fused operations
and are correctly rounded as a whole -- an f32 division is 3 v_fma_f32 + 2 v_fmac_f32 around v_div_scale / v_rcp /
v_div_fmas / v_div_fixup, an f32 square root 2 v_fma_f32 more, an integer division a few v_fmamk / v_fmac on converted
integers.  tests/test_isa_invariants.py holds the assertions; run this file for the table.

    python tools/isa_stats.py [path/to/libfyrox_hip.so]"""
import collections
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
COUNTED = ("v_fma_f32", "v_fmac_f32", "v_pk_fma_f32", "v_mad_f32", "v_mac_f32", "v_fma_mix_f32", "v_fmaak_f32", "v_fmamk_f32",
           "v_div_fixup_f32", "v_sqrt_f32", "v_pk_mul_f32", "v_pk_add_f32", "scratch")      # scratch: any scratch_load / _store
_PAT = re.compile(r"^\s*(" + "|".join(c for c in COUNTED if c != "scratch") + r")(_e32|_e64|_dpp|_sdwa)?\s")
_SCRATCH = re.compile(r"^\s*scratch_(load|store)_")


def kernel_stats(lib_path: str) -> dict:
    """{demangled kernel name: Counter of the instructions in COUNTED}"""
    tmp = tempfile.mkdtemp(prefix="fyx_isa_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, so)
        subprocess.run([OBJDUMP, "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        res = {}
        for obj in glob.glob(so + ".*gfx950"):
            text = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", obj], check=True, capture_output=True, text=True).stdout
            cur = None
            for ln in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
                if m:
                    cur = res.setdefault(m.group(1), collections.Counter())
                    continue
                if cur is not None:
                    mm = _PAT.match(ln)
                    if mm:
                        cur[mm.group(1)] += 1
                    elif _SCRATCH.match(ln):
                        cur["scratch"] += 1
        names = sorted(res)
        dem = subprocess.run(["c++filt"] + names, check=True, capture_output=True, text=True).stdout.splitlines()
        return {d.split("(")[0].replace("void ", ""): res[n] for n, d in zip(names, dem)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernel_resources(lib_path: str) -> dict:
    """{demangled kernel name: {"vgpr": .., "agpr": .., "sgpr": .., "scratch_bytes": .., "lds_bytes": ..}} from the code
    objects' metadata notes (what the runtime allocates per wave / workgroup; LDS: the static part only)"""
    tmp = tempfile.mkdtemp(prefix="fyx_isa_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, so)
        subprocess.run([OBJDUMP, "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = {}
        for obj in glob.glob(so + ".*gfx950"):
            text = subprocess.run([READELF, "--notes", obj], check=True, capture_output=True, text=True).stdout
            for block in re.split(r"\n  - \.agpr_count:", "\n" + text)[1:]:
                def field(key, block=block):
                    m = re.search(r"\." + key + r":\s+(\S+)", block)
                    return m.group(1) if m else None
                name = field("name")
                if name is None:
                    continue
                out[name] = {"agpr": int(block.split()[0]), "vgpr": int(field("vgpr_count")), "sgpr": int(field("sgpr_count")),
                             "scratch_bytes": int(field("private_segment_fixed_size")), "lds_bytes": int(field("group_segment_fixed_size"))}
        names = sorted(out)
        dem = subprocess.run(["c++filt"] + names, check=True, capture_output=True, text=True).stdout.splitlines()
        return {d.split("(")[0].replace("void ", ""): out[n] for n, d in zip(names, dem)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stats = kernel_stats(sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "fyrox_amd", "libfyrox_hip.so"))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "fyrox_amd", "libfyrox_hip.so")
    res = kernel_resources(lib)
    print(f"{'kernel':70s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scr B':>6s} " + " ".join(f"{c.replace('v_', '').replace('_f32', ''):>9s}" for c in COUNTED))
    for k in sorted(stats):
        r = res.get(k, {})
        print(f"{k[:70]:70s} {r.get('vgpr', -1):5d} {r.get('agpr', -1):5d} {r.get('sgpr', -1):5d} {r.get('scratch_bytes', -1):6d} "
              + " ".join(f"{stats[k][c]:9d}" for c in COUNTED))
