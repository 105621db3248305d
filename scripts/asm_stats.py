#!/usr/bin/env python3
"""Instruction mix per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only), beside the compiler's resource
report (-Rpass-analysis=kernel-resource-usage, stderr of the same compile).

    scripts/asm_stats.py build/asm/spec32.s [build/asm/spec32.res] [--filter substr ...]

Used by tests/test_kernel_resources.py for the disassembly assertions (no ds_bpermute in the forward tail) and by hand
when a kernel is re-tuned: MFMA count against VALU / LDS / cross-lane instruction counts is what the SQ counters measure
at run time (profiles/*_sq_pmc.md)."""
import re
import subprocess
import sys

CLASSES = [
    ("mfma", re.compile(r"^v_mfma")),
    ("cndmask", re.compile(r"^v_cndmask")),
    ("permlane_swap", re.compile(r"^v_permlane(16|32)_swap")),
    ("dpp", re.compile(r"^v_\S+_dpp|\bquad_perm|\brow_(shl|shr|ror|mirror|half_mirror|bcast)")),
    ("pk_fma", re.compile(r"^v_pk_(fma|mul|add)_f32")),
    ("fma", re.compile(r"^v_(fma|fmac|mac|mul|add|sub)_f32")),
    ("valu", re.compile(r"^v_(?!mfma)")),
    ("bpermute", re.compile(r"^ds_bpermute|^ds_permute")),
    ("ds_read", re.compile(r"^ds_read")),
    ("ds_write", re.compile(r"^ds_write")),
    ("vmem_load", re.compile(r"^(global|buffer|flat)_load")),
    ("vmem_store", re.compile(r"^(global|buffer|flat)_store")),
    ("smem", re.compile(r"^s_load|^s_buffer_load")),
    ("waitcnt", re.compile(r"^s_waitcnt")),
    ("barrier", re.compile(r"^s_barrier")),
    ("nop", re.compile(r"^s_nop")),
    ("salu", re.compile(r"^s_(?!waitcnt|barrier|nop|load|buffer_load|endpgm)")),
    ("scratch", re.compile(r"^scratch_")),
]


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout
        return dict(zip(names, out.splitlines()))
    except Exception:  # noqa: BLE001
        return {n: n for n in names}


def parse_asm(path):
    """{mangled name: {class: count}} for every .amdhsa kernel body in the listing"""
    kernels, cur, name = {}, None, None
    for raw in open(path):
        line = raw.strip()
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            name, cur = m.group(1), {}
            kernels[name] = cur
            continue
        if line.startswith(".Lfunc_end") or line.startswith("s_endpgm") and False:
            cur = None
            continue
        if cur is None or not line or line[0] in ".;" or line.endswith(":"):
            continue
        ins = line.split(";")[0].strip()
        if not ins:
            continue
        cur["total"] = cur.get("total", 0) + 1
        for cls, rx in CLASSES:
            if rx.search(ins):
                cur[cls] = cur.get(cls, 0) + 1
    return kernels


def parse_res(path):
    res, cur = {}, None
    for line in open(path):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    filt = []
    if "--filter" in sys.argv:
        filt = sys.argv[sys.argv.index("--filter") + 1:]
        args = [a for a in args if a not in filt]
    ks = parse_asm(args[0])
    res = parse_res(args[1]) if len(args) > 1 else {}
    dm = demangle(list(ks))
    cols = ["total", "mfma", "valu", "fma", "pk_fma", "cndmask", "permlane_swap", "dpp", "bpermute", "ds_read", "ds_write", "vmem_load",
            "vmem_store", "waitcnt", "barrier", "nop", "salu", "scratch"]
    for n, c in ks.items():
        if not c.get("total"):
            continue
        d = dm.get(n, n)
        if filt and not all(f in d or f in n for f in filt):
            continue
        r = res.get(n, {})
        short = re.sub(r"ttx::", "", d.split("(")[0].replace("void ", ""))
        print(short)
        print("   " + "  ".join(f"{k}={c.get(k, 0)}" for k in cols))
        if r:
            print("   " + "  ".join(f"{k}={v}" for k, v in r.items() if k in ("VGPRs", "AGPRs", "ScratchSize", "Occupancy", "SGPRs", "LDS Size")))


if __name__ == "__main__":
    main()
