#!/usr/bin/env python3
"""Kernel-tuning helper.  HERE (no GPU): build variant libraries of ONE source file with extra -D flags,
    python scripts/variants.py build k_dwfc.hip a="-DMF_DWFC_UB=2" b="-DMF_DWFC_THREADS=1024 -DMF_DWFC_UB=2" ...
(source "ALL" = every .hip file, for a switch that lives in a shared header: build ALL r02epi="-DMF_EPI=0")
-> microflow_rs_amd/variants/lib_<tag>.so (git-ignored, travels with gpurun).  On the GPU box:
    python scripts/variants.py run "<command>"
runs <command> once per variant with that library swapped in for libmicroflow_amd.so, then restores the original."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "microflow_rs_amd")
sys.path.insert(0, PKG)
import build as B  # noqa: E402

VDIR = os.path.join(PKG, "variants")


def build(src, variants):
    B.build()  # objects of the default build
    os.makedirs(VDIR, exist_ok=True)
    if not os.environ.get("VARIANTS_KEEP"):  # (VARIANTS_KEEP=1: add to the variants of an earlier call, e.g. of another source file)
        for f in glob.glob(os.path.join(VDIR, "lib_*.so")):
            os.remove(f)
    objdir = os.path.join(PKG, "build")
    procs = []
    srcs = [f for f in B.SOURCES if f.endswith(".hip")] if src == "ALL" else [src]  # ALL: a flag that lives in a shared header
    for tag, flags in variants:
        for one in srcs:
            obj = os.path.join(VDIR, "%s.%s.o" % (one, tag))
            cmd = [B.hipcc()] + B.FLAGS + flags.split() + ["-x", "hip", "-c", os.path.join(B.CSRC, one), "-o", obj]
            procs.append((tag, flags, one, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = set()
    for tag, flags, one, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            print("variant %s failed (%s)" % (tag, one))
            failed.add(tag)
    for tag, flags in variants:
        if tag in failed:
            continue
        mine = {one: obj for t, _, one, obj, _ in procs if t == tag}
        objs = [mine.get(s, os.path.join(objdir, s + ".o")) for s in B.SOURCES]
        subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(VDIR, "lib_%s.so" % tag)] + objs)
        for obj in mine.values():
            os.remove(obj)
        print("built", tag, flags)


def run(command):
    good = B.LIB + ".good"
    shutil.copy(B.LIB, good)
    try:
        for lib in [good] + sorted(glob.glob(os.path.join(VDIR, "lib_*.so"))):
            tag = "default" if lib == good else os.path.basename(lib)[4:-3]
            shutil.copy(lib, B.LIB)
            os.utime(B.LIB)  # newer than the sources: no rebuild on import
            out = subprocess.run(command, shell=True, cwd=ROOT, capture_output=True, text=True)
            for line in (out.stdout.strip().splitlines() or ["(no output) " + out.stderr[-300:]]):
                print("%-12s %s" % (tag, line))
    finally:
        shutil.copy(good, B.LIB)
        os.utime(B.LIB)
        os.remove(good)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], [a.split("=", 1) for a in sys.argv[3:]])
    else:
        run(sys.argv[2])
