"""The C-ABI shared library: loads without a GPU, exports every symbol include/relate_hip.h declares, and
fails loudly (no CPU fallback) when no device is present."""
import ctypes as C
import os
import re

import pytest

import relate_amd
from relate_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = ""
    for f in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if f.endswith(".h"):
            with open(os.path.join(ROOT, "include", f)) as fh:
                text += fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(relate_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = relate_amd.load_library()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), n


def test_struct_layouts_match_header():
    # relate_opts: 12 doubles then 13 32-bit fields (+4 tail padding); relate_fit: 7 doubles then 4 ints
    assert C.sizeof(api.RelateOpts) == 12 * 8 + 13 * 4 + 4
    assert C.sizeof(api.RelateFit) == 7 * 8 + 4 * 4


def test_shard_partition_is_contiguous_and_complete():
    for P in (0, 1, 7, 190, 1999000):
        for n in (1, 2, 3, 8):
            cuts = [relate_amd.shard(P, r, n) for r in range(n)]
            assert cuts[0][0] == 0 and cuts[-1][1] == P
            for a, b in zip(cuts, cuts[1:]):
                assert a[1] == b[0]
            sizes = [e - b for b, e in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(relate_amd.RelateError) as e:
        relate_amd.Relate(0)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """No file of the product (relate_amd/, include/) may include, link or import anything under oracle/."""
    bad = []
    for base in ("relate_amd", "include"):
        for dp_, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".so", ".pyc")):
                    continue
                with open(os.path.join(dp_, f), errors="ignore") as fh:
                    t = fh.read()
                if re.search(r"relate_oracle|oracleload|refload|librelate_ref|orc_[a-z_]+\(", t):
                    bad.append(os.path.join(dp_, f))
    assert not bad, bad


def test_host_cli_builds_and_fails_loudly_without_gpu(tmp_path):
    """relate_amd/relateHMM_hip (reference CLI flags over the C ABI): usage without arguments; with data but no
    GPU it must stop with the library's error, not fall back to anything."""
    import subprocess
    import torch
    cli = os.path.join(ROOT, "relate_amd", "relateHMM_hip")
    assert os.path.exists(cli), "run __graft_entry__.build()"
    r = subprocess.run([cli], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "SYNTAX" in r.stdout
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    data = os.path.join(ROOT, "tests", "golden", "data")
    r = subprocess.run([cli, "-o", data + "/options.linkage.txt", "-file", data + "/dat_small", "-v", "0"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1
    assert "no HIP device" in r.stderr or "no CPU fallback" in r.stderr


def test_host_buffer_helpers_reject_what_they_do_not_own():
    """relate_hip_host_free takes only blocks of relate_hip_host_alloc; without a GPU the allocation fails with a
    message instead of returning pageable memory."""
    import ctypes as C
    import relate_amd
    L = relate_amd.load_library()
    assert L.relate_hip_host_free(None) == 0
    assert L.relate_hip_host_free(C.c_void_p(0x1000)) < 0
    assert b"relate_hip_host_alloc" in L.relate_hip_last_error(None)
    import torch
    if not torch.cuda.is_available():
        p = C.c_void_p(1)
        assert L.relate_hip_host_alloc(C.byref(p), 1 << 20) < 0
        assert not p.value
        assert b"hipHostMalloc" in L.relate_hip_last_error(None)
