"""Builds libmicroflow_amd.so in-tree with hipcc for gfx950.

    python microflow_rs_amd/build.py [--force]

-ffp-contract=off is part of the numerical contract: the reference (Rust) never
fuses a multiply and an add, and hipcc does by default.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmicroflow_amd.so")
SOURCES = ["capi.cpp", "hostmath.cpp", "epi_fma.cpp", "switches.cpp", "tflite.cpp", "model.cpp", "ops.hip", "k_generic.hip", "k_depthwise.hip",
           "k_pointwise.hip", "k_fused_mm.hip", "k_stage.hip", "k_dwfc.hip", "k_tail3.hip", "k_gemm.hip", "k_rt.hip", "k_quad.hip", "k_quad_mm.hip", "k_chain.hip"]
HEADERS = ["mf_internal.hpp", "mf_switches.hpp", "kernels.hpp", "k_common.hpp", "k_dwtask.hpp", "k_tail.hpp", os.path.join("..", "..", "include", "microflow_amd.h")]
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950's register file is unified), which
# removes one v_accvgpr_read per accumulator element from every fused epilogue.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-result", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
# (-mllvm -amdgpu-atomic-optimizer-strategy=None was tried in round 4 together with a step-queue draw whose result is first
# touched at the END of the step: hipcc otherwise broadcasts a returning atomic with v_readfirstlane right behind it, i.e. the
# drawing wave waits for the round trip at the top of the step.  Same-box A/B, profiles/r04/dq_ab.txt: no gain -- the round
# trip is short enough -- and the flag cost the five-operator launch 2 %.  Not kept.)
EXTRA = os.environ.get("MF_EXTRA_HIPCC_FLAGS", "").split()  # kernel-tuning experiments (-DMF_...=n)
FLAGS += EXTRA
# Knock-out / diagnostic switches (-DMF_*_KO=n, -DMF_*_DIAG=n with n != 0) make kernels that are WRONG on purpose (profiling only).
# Such a build says so loudly here, carries the flags inside the library (mf_build_info) and is refused by _lib.py unless
# MF_ALLOW_DIAG_BUILD=1 is set, so that it cannot pass for the product.
import re  # noqa: E402
NONSHIPPING = [f for f in EXTRA if re.match(r"-DMF_\w*(KO|DIAG)\w*(=(?!0$)|$)", f)]  # (a bare -DMF_X_KO defines it as 1)


# measurement tool, not product: the requantisation-rate microbenchmark bench.py runs beside its timed region
UBENCH_SRC = os.path.join(HERE, "..", "scripts", "ubench", "epi_rate.hip")
UBENCH_LIB = os.path.join(HERE, "..", "scripts", "ubench", "libepi_rate.so")


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libmicroflow_amd.so cannot be built")
    return exe


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_ubench(force=False):
    deps = [UBENCH_SRC, os.path.join(CSRC, "k_common.hpp"), os.path.join(CSRC, "kernels.hpp")]
    if (not force and os.path.exists(UBENCH_LIB)
            and os.path.getmtime(UBENCH_LIB) >= max(os.path.getmtime(d) for d in deps)):
        return UBENCH_LIB
    subprocess.check_call([hipcc()] + [f for f in FLAGS if f != "-Wall"] + ["-Wno-unused-value", "-DMF_UBENCH_LIB", "-DITERS=1024", "-shared", "-x", "hip",
                                                    UBENCH_SRC, "-o", UBENCH_LIB])
    return UBENCH_LIB


def build(force=False, verbose=False):
    if NONSHIPPING:
        sys.stderr.write("\n*** microflow_rs_amd/build.py: NON-SHIPPING BUILD -- %s make kernels that are wrong on purpose. "
                         "Rebuild without MF_EXTRA_HIPCC_FLAGS before running tests or benchmarks. ***\n\n" % " ".join(NONSHIPPING))
    build_ubench(force)
    if not force and not stale():
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        cmd = [cc] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if src == "capi.cpp":
            cmd.insert(1, '-DMF_BUILD_EXTRA="%s"' % " ".join(EXTRA).replace('"', ""))
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError("hipcc failed on " + src)
        if verbose and out:
            sys.stdout.write(out.decode(errors="replace"))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
