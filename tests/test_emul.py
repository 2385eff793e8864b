"""Host build of the kernels' arithmetic (csrc/jpegx_math.h via tests/emul/emul.cpp) against the
oracle: the fp32 fast tier + error bound + float64 exact tier must reproduce the reference's
integers bit for bit, and the observed fp32 error must sit far inside the bound.  CPU only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle
from conftest import CASES, MODES, REPO

SO = os.path.join(REPO, "tests", "emul", "_build", "libemul.so")


@pytest.fixture(scope="module")
def emul():
    src = os.path.join(REPO, "tests", "emul", "emul.cpp")
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    hdr = os.path.join(REPO, "implementing-jpeg-compression_amd", "csrc", "jpegx_math.h")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-fPIC", "-shared", "-o", SO, src])
    return ctypes.CDLL(SO)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def rq_table(mode, param):
    qt = oracle.tables()["qtable"].ravel()
    if mode == "qtable":
        return (1.0 / qt).astype(np.float32)
    if mode == "none":
        return np.ones(64, np.float32)
    if mode == "divide":
        return np.full(64, np.float32(1.0 / param))
    r = np.zeros((8, 8), np.float32)
    r[:int(param), :int(param)] = 1
    return r.ravel()


def run_forward(lib, plane, mode, param, pixel):
    a = np.ascontiguousarray(plane, np.float32)
    h, w = a.shape
    rq = rq_table(mode, param)
    out = np.empty((h // 8, w // 8, 64), np.int16)
    st = np.zeros(4)
    dc_exact = int(bool(pixel) and np.frexp(rq[0])[0] == 0.5 and (mode != "divide" or float(rq[0]) * param == 1.0))
    lib.emul_forward(_p(a, ctypes.c_float), h, w, oracle.MODE_BY_NAME[mode], ctypes.c_double(param),
                     _p(rq, ctypes.c_float), int(pixel), dc_exact, _p(out, ctypes.c_int16), None,
                     _p(st, ctypes.c_double))
    return out, st


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("suffix,mode,param", MODES)
def test_forward_two_tier_matches_reference(emul, golden, case, suffix, mode, param):
    c = golden(case)
    for pixel in (1, 0):
        got, st = run_forward(emul, c["pre"], mode, param, pixel)
        assert np.array_equal(got, c["zz_" + suffix])
        assert st[2] < 16.0          # observed |c32 - c64| / (u S) vs the 16 u S bound


@pytest.mark.parametrize("kind", ["noise", "smooth"])
def test_forward_two_tier_large(emul, kind):
    from jpegx import synth
    a = synth.generate_plane(kind, 512, 512, seed=9)
    for mode, param in (("qtable", 0.0), ("divide", 7.0), ("none", 0.0)):
        got, st = run_forward(emul, a, mode, param, 1)
        assert np.array_equal(got, oracle.forward_f32(a, mode, param))
        assert st[2] < 4.0           # typical error is ~1 u S: an order of magnitude inside the bound
    _, st = run_forward(emul, a, "qtable", 0.0, 1)
    assert st[1] / (a.size / 64) < 0.08   # exact-tier share of blocks stays small for the JPEG table


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("suffix,mode,param", MODES)
def test_inverse_two_tier_matches_reference(emul, golden, case, suffix, mode, param):
    c = golden(case)
    zz = np.ascontiguousarray(c["zz_" + suffix])
    h, w = zz.shape[0] * 8, zz.shape[1] * 8
    out = np.empty((h, w), np.int32)
    st = np.zeros(4)
    emul.emul_inverse(_p(zz, ctypes.c_int16), h, w, oracle.MODE_BY_NAME[mode], ctypes.c_double(param),
                      _p(out, ctypes.c_int32), _p(st, ctypes.c_double))
    assert np.array_equal(out, c["idct_" + suffix])
    assert st[2] < 1.0               # observed fp32 error / bound u (0.125 |DC| + 0.875 sum|AC|)
