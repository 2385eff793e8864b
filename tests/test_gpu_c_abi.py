"""GPU: libjpegx.so used from plain C through include/jpegx.h (tests/c_abi/abi_roundtrip.c) -- the drop-in
boundary as a C host would bind it, without Python or HIP headers on the caller's side."""
import os
import shutil
import subprocess

import pytest

from conftest import PKG, REPO

pytestmark = pytest.mark.gpu


def test_c_host_program_round_trip(gpu, tmp_path):
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler on this box")
    exe = str(tmp_path / "abi_roundtrip")
    subprocess.check_call([cc, "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"),
                           os.path.join(REPO, "tests", "c_abi", "abi_roundtrip.c"), "-o", exe,
                           "-L", PKG, "-ljpegx", "-lm", "-Wl,-rpath," + PKG])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.startswith("ok ")
