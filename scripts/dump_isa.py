#!/usr/bin/env python3
"""Disassemble the gfx950 code object of a built csrc/*.o and print per-kernel instruction statistics
(or one kernel's text).  CPU only (hipcc cross-compiles): the tool behind profiles/*/instruction_mix notes.

  python scripts/dump_isa.py                       # table: kernel, instructions, MFMA, VALU, SALU, LDS, VMEM, regs
  python scripts/dump_isa.py --kernel 'k_wino<3, 4, false>' --out /tmp/a1.s
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    work = tempfile.mkdtemp(prefix="tzisa_")
    try:
        shutil.copy(obj, os.path.join(work, "k.o"))
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "k.o"], cwd=work, stdout=subprocess.DEVNULL)
        co = [f for f in os.listdir(work) if "amdgcn" in f]
        text = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", co[0]], cwd=work, text=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    fns = {}
    for fn in re.split(r"\n(?=[0-9a-f]+ <)", text):
        m = re.match(r"[0-9a-f]+ <([^>]*(?:<[^>]*>)?[^>]*)>:", fn)
        if m:
            fns[m.group(1)] = [l.split("//")[0].strip() for l in fn.splitlines()[1:] if l.strip()]
    return fns


def classify(ins):
    op = ins.split()[0] if ins else ""
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "acc_mov"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--obj", default=os.path.join(ROOT, "tezip_amd", "csrc", "tz_prednet.o"))
    ap.add_argument("--kernel", help="substring of the demangled kernel name: print (or --out) its disassembly")
    ap.add_argument("--out")
    args = ap.parse_args()
    fns = disassemble(args.obj)
    if args.kernel:
        hits = [n for n in fns if args.kernel in n]
        if len(hits) != 1:
            sys.exit("kernel name matches %d functions: %s" % (len(hits), hits[:10]))
        text = "\n".join(fns[hits[0]])
        if args.out:
            open(args.out, "w").write(text + "\n")
            print(hits[0], len(fns[hits[0]]), "instructions ->", args.out)
        else:
            print(text)
        return
    print("%-60s %7s %6s %6s %6s %6s %6s %7s" % ("kernel", "instr", "mfma", "valu", "salu", "lds", "vmem", "acc_mov"))
    for name, body in sorted(fns.items()):
        c = {}
        for ins in body:
            k = classify(ins)
            c[k] = c.get(k, 0) + 1
        print("%-60s %7d %6d %6d %6d %6d %6d %7d" % (name[:60], len(body), c.get("mfma", 0), c.get("valu", 0), c.get("salu", 0),
                                                   c.get("lds", 0), c.get("vmem", 0), c.get("acc_mov", 0)))


if __name__ == "__main__":
    main()
