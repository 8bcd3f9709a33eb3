"""The C++ host mirror of the reference's operator interface (suhmo_amd/host) driven like
Chombo's multigrid drives VCAMRNonLinearPoissonOp, checked bitwise against the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host_cpp", "test_mirror")


def build_exe():
    from suhmo_amd import capi
    capi.build()
    csrc = os.path.join(ROOT, "suhmo_amd", "csrc")
    srcs = [os.path.join(ROOT, "tests", "host_cpp", "test_mirror.cpp"),
            os.path.join(ROOT, "suhmo_amd", "host", "VCAMRNonLinearPoissonOpHIP.cpp")]
    objs = []
    for c in ("suhmo_oracle.c", "level_shim.c", "amr2.c", "amrm.c"):
        o = os.path.join(ROOT, "tests", "host_cpp", c.replace(".c", ".o"))
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-ffp-contract=off", "-fopenmp", "-c", os.path.join(ROOT, "oracle", c), "-o", o])
        objs.append(o)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off"] + srcs + objs +
                          ["-L" + csrc, "-lsuhmo_hip", "-Wl,-rpath," + csrc, "-fopenmp", "-lm", "-o", EXE])


def test_host_mirror_compiles_cpu():
    """the mirror is plain C++ over the C-ABI: it must build without a GPU"""
    build_exe()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_host_mirror_matches_oracle():
    build_exe()
    p = subprocess.run([EXE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode()
    print(out)
    assert p.returncode == 0 and "RESULT: PASS" in out, out


EXE_B2 = os.path.join(ROOT, "tests", "host_cpp", "test_b2")


def build_b2():
    from suhmo_amd import capi
    capi.build()
    csrc = os.path.join(ROOT, "suhmo_amd", "csrc")
    o = os.path.join(ROOT, "tests", "host_cpp", "suhmo_oracle.o")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-ffp-contract=off", "-c", os.path.join(ROOT, "oracle", "suhmo_oracle.c"), "-o", o])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "host_cpp", "test_b2.cpp"), o,
                           "-L" + csrc, "-lsuhmo_hip", "-Wl,-rpath," + csrc, "-lm", "-o", EXE_B2])


def test_b2_symbols_exported_cpu():
    """every per-box Fortran-ABI symbol of include/suhmo_chf.h is exported (link check, no GPU)"""
    import re
    build_b2()
    hdr = open(os.path.join(ROOT, "include", "suhmo_chf.h")).read()
    names = set(re.findall(r"^void ([a-z_0-9]+_)\(", hdr, flags=re.M))
    assert len(names) == 23, names          # 18 on the solve path + 5 of the time step (src/AmrHydroF.ChF)
    out = subprocess.check_output(["nm", "-D", os.path.join(ROOT, "suhmo_amd", "csrc", "libsuhmo_hip.so")]).decode()
    for n in names:
        assert re.search(r" T %s$" % n, out, flags=re.M), n


@pytest.mark.gpu
def test_b2_per_box_kernels_match_oracle():
    build_b2()
    p = subprocess.run([EXE_B2], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode()
    print(out)
    assert p.returncode == 0 and "RESULT: PASS" in out, out
