"""Static instruction statistics of a kernel inside a hipRTC-built code object (the bytes that are
actually loaded), by disassembling it with the ROCm llvm-objdump.  Diagnostics only: bench.py derives
its fp64-VALU ceiling from this instead of a hand-maintained constant, tools/isa_stats.py prints it.
"""
import collections
import hashlib
import os
import re
import subprocess
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
_INS = re.compile(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_LABEL = re.compile(r"^([0-9a-f]+)\s+<(L[0-9A-Za-z_]+)>:")


def code_digest(blob):
    return hashlib.sha256(bytes(blob)).hexdigest()[:24]


def disassemble(blob, kernel):
    """[(address, mnemonic, operands)] of one kernel + {label: address}."""
    with tempfile.NamedTemporaryFile(suffix=".hsaco", delete=False) as f:
        f.write(bytes(blob))
        path = f.name
    try:
        out = subprocess.run([OBJDUMP, "-d", "--symbolize-operands", path], capture_output=True, text=True,
                             check=True).stdout
    finally:
        os.unlink(path)
    start = out.index("<%s>:" % kernel)
    nxt = re.search(r"\n[0-9a-f]+ <(?!L[0-9]+>)[A-Za-z0-9_$.]+>:", out[start + 20:])
    body = out[start:start + 20 + nxt.start()] if nxt else out[start:]
    ins, labels = [], {}
    for line in body.split("\n"):
        m = _LABEL.match(line)
        if m:
            labels[m.group(2)] = int(m.group(1), 16)
            continue
        m = _INS.match(line)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return ins, labels


def _mix(ins):
    c = collections.Counter(op for _, op, _ in ins)
    return {
        "instructions": len(ins),
        "valu": sum(v for k, v in c.items() if k.startswith("v_")),
        "valu_f64": sum(v for k, v in c.items() if k.startswith("v_") and "f64" in k),
        "salu": sum(v for k, v in c.items() if k.startswith("s_")),
        "s_mov": c["s_mov_b32"] + c["s_mov_b64"],
        "lane_moves": c["v_readlane_b32"] + c["v_writelane_b32"],
        "lds": sum(v for k, v in c.items() if k.startswith("ds_")),
        "vmem": sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_"))),
        "scratch": sum(v for k, v in c.items() if k.startswith("scratch_")),
    }


def kernel_stats(blob, kernel):
    """Instruction mix of the whole kernel and of its outermost loop (the time-step loop of the steppers:
    the backward branch with the longest span)."""
    ins, labels = disassemble(blob, kernel)
    addr = [a for a, _, _ in ins]
    best = None
    for a, op, args in ins:
        if not op.startswith("s_cbranch") and op != "s_branch":
            continue
        tgt = labels.get(args.split()[-1].strip("<>")) if args else None
        if tgt is not None and tgt < a and (best is None or a - tgt > best[1] - best[0]):
            best = (tgt, a)
    # identity of THIS kernel's machine code: the mnemonic sequence (operands carry pc-relative literals that
    # move with the layout of the code object)
    kdig = hashlib.sha256("\n".join(op for _, op, _ in ins).encode()).hexdigest()[:24]
    out = {"kernel": kernel, "digest": code_digest(blob), "kernel_digest": kdig, "whole": _mix(ins)}
    if best:
        loop = [t for t in ins if best[0] <= t[0] <= best[1]]
        out["step_loop"] = _mix(loop)
        out["step_loop"]["bytes"] = best[1] - best[0]
    out["code_bytes"] = (addr[-1] - addr[0]) if addr else 0
    return out


def kernel_resources(blob, kernel):
    """vgpr/sgpr/spill/LDS/scratch numbers of the kernel's metadata note."""
    with tempfile.NamedTemporaryFile(suffix=".hsaco", delete=False) as f:
        f.write(bytes(blob))
        path = f.name
    try:
        meta = subprocess.run([READELF, "--notes", path], capture_output=True, text=True, check=True).stdout
    finally:
        os.unlink(path)
    i = meta.index(".name:           %s\n" % kernel)
    lo = meta.rfind("  - .", 0, i)
    lo = meta.rfind("- .agpr_count", 0, i) if meta.rfind("- .agpr_count", 0, i) > 0 else lo
    hi = meta.find("- .agpr_count", i)
    blk = meta[max(0, lo):hi if hi > 0 else len(meta)]
    keys = ("vgpr_count", "agpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count",
            "private_segment_fixed_size", "group_segment_fixed_size")
    return {k: int(m.group(1)) for k in keys for m in [re.search(r"\.%s:\s+(\d+)" % k, blk)] if m}
