"""The path that takes the REAL longwave k-distribution data in (tools/ingest_lw_data.sh), proven with a stand-in.

The reference checkout lacks rrtmg_lw_k_g.f90 (/root/reference/.MISSING_LARGE_BLOBS:3), so the shipped longwave table
blob carries synthetic raw tables and says so (lw/meta/synthetic = 1).  What a user with the file runs is:
oracle/build_ref.sh compiles the file into the reference library instead of the empty loaders, tools/pack_tables.py dumps
what the reference loaded and clears the flag, and the longwave cache comparisons switch on at 1e-8.

  test_ingest_chain...   writes today's raw tables out as a rrtmg_lw_k_g.f90 in the syntax of the reference's data files
                         (tools/write_lw_k_g.py), pushes it through that chain into a side directory and requires the
                         packed blob to equal the shipped one BIT FOR BIT with synthetic == 0 (and the reference's own
                         256 -> 140 reduction of it to equal the committed fixture).  Compiling the 7.6 MB file takes flang
                         about five minutes, so the build is kept in oracle/_ref_ingest/ (git-ignored) and redone only
                         when the file's hash changes; without that directory the test runs when RRTMG_TEST_INGEST=1
                         (__graft_entry__.build() sets the directory up in the build container).
  test_longwave_cache_comparisons_execute...   on a blob whose flag is clear, RRTMGLongwave() constructs WITHOUT
                         allow_synthetic_tables and the four longwave cache classes are compared at 1e-8 -- with the stand-in
                         data they must fail on values: the assertion is that every comparison RAN.
"""
import hashlib
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, load_cache_case, maxdiff

REF = os.environ.get("CLIMT_REFERENCE", "/root/reference")
FLANG = os.environ.get("FC", "/opt/rocm/lib/llvm/bin/flang")
SIDE = os.path.join(ROOT, "oracle", "_ref_ingest")
SHIPPED = os.path.join(ROOT, "climt_amd", "data", "rrtmg_lw_data.bin")

LW_CACHE_CLASSES = (("TestRRTMGLongwave", "column", {}),
                    ("TestRRTMGLongwaveWithClouds", "column", dict(cloud_optical_properties="single_cloud_type")),
                    ("TestRRTMGLongwaveWithExternalInterfaceTemperature", "column", dict(calculate_interface_temperature=False)),
                    ("TestRRTMGLongwaveMCICA", "3d", dict(mcica=True)))


def blob_with_flag(dst, synthetic):
    """The shipped blob with lw/meta/synthetic set -- byte for byte what the ingestion chain packs from a data file holding
    the same raw tables (test_ingest_chain_reproduces_the_shipped_blob proves that equality)."""
    from tools.pack_tables import Blob, read_blob
    src = read_blob(SHIPPED)
    b = Blob()
    for k in src:      # (file order)
        b.add(k, np.array([synthetic], dtype=np.int32) if k == "lw/meta/synthetic" else src[k], src[k].shape)
    b.write(dst)
    return dst


def compare_lw_caches(make_component):
    """-> (comparisons made, comparisons over 1e-8) over the reference's four longwave cache classes."""
    ran = failed = 0
    for cls, desc, kw in LW_CACHE_CLASSES:
        comp = make_component(**kw)
        state, tend, diag = load_cache_case(cls, desc)
        np.random.seed(0)
        t, dg = comp(state)
        for got, exp in ((t, tend), (dg, diag)):
            for k in exp:
                g = np.transpose(got[k].values, [got[k].dims.index(x) for x in exp[k].dims])
                ran += 1
                failed += maxdiff(g, exp[k].values) > 1e-8
    return ran, failed


def test_shipped_blob_round_trips_through_the_packer(tmp_path):
    from tools.pack_tables import read_blob
    p = blob_with_flag(str(tmp_path / "same.bin"), 1)
    assert open(p, "rb").read() == open(SHIPPED, "rb").read()
    q = read_blob(blob_with_flag(str(tmp_path / "flag0.bin"), 0))
    assert int(q["lw/meta/synthetic"][0]) == 0


@pytest.mark.skipif(not (os.path.isdir(os.path.join(REF, "climt/_lib/rrtmg_lw")) and os.path.exists(FLANG)),
                    reason="needs the reference checkout and flang (build container)")
def test_ingest_chain_reproduces_the_shipped_blob(tmp_path):
    from tools.pack_tables import read_blob
    kg = str(tmp_path / "rrtmg_lw_k_g.f90")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "write_lw_k_g.py"), kg])
    sha = hashlib.sha256(open(kg, "rb").read()).hexdigest()
    marker = os.path.join(SIDE, "lw_kdata.txt")
    cached = os.path.exists(marker) and open(marker).read().split()[-1] == sha and os.path.exists(os.path.join(SIDE, "librrtmg_lw_ref.so"))
    if not cached and os.environ.get("RRTMG_TEST_INGEST", "") in ("", "0"):
        pytest.skip("oracle/_ref_ingest is not built for today's tables (about 5 min of flang): RRTMG_TEST_INGEST=1 or __graft_entry__.build()")
    env = dict(os.environ, RRTMG_REF_OUT=SIDE, RRTMG_LW_K_G=kg)
    if cached:
        # same bytes as the file the cached object was compiled from: hand build_ref.sh that path so it keeps the object
        env["RRTMG_LW_K_G"] = open(marker).read().split()[1]
        if not os.path.exists(env["RRTMG_LW_K_G"]):
            os.makedirs(os.path.dirname(env["RRTMG_LW_K_G"]), exist_ok=True)
            shutil.copy(kg, env["RRTMG_LW_K_G"])
    subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_ref.sh"), "lw"], env=env)
    assert open(marker).read().startswith("file ") and open(marker).read().split()[-1] == sha      # the stub was NOT linked
    syms = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(SIDE, "librrtmg_lw_ref.so")]).decode()
    assert "lw_kgb01_" in syms and "lw_kgb16_" in syms
    out, fix = str(tmp_path / "ingested.bin"), str(tmp_path / "reduced.npz")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pack_tables.py"), "lw", "--out", out, "--fixture-out", fix],
                          env=dict(os.environ, RRTMG_REF_DIR=SIDE))
    got, want = read_blob(out), read_blob(SHIPPED)
    assert set(got) == set(want)
    assert int(got["lw/meta/synthetic"][0]) == 0 and int(want["lw/meta/synthetic"][0]) == 1
    for k in want:
        if k != "lw/meta/synthetic":
            assert got[k].shape == want[k].shape and np.array_equal(got[k], want[k]), k      # every double read back exactly
    # ... i.e. the packed file is the shipped one with the flag cleared, byte for byte
    assert open(out, "rb").read() == open(blob_with_flag(str(tmp_path / "flag0.bin"), 0), "rb").read()
    red, ref = np.load(fix), np.load(os.path.join(ROOT, "tests", "golden", "lw_reduced_tables.npz"))
    assert set(red.files) == set(ref.files) and all(np.array_equal(red[k], ref[k]) for k in ref.files)


@pytest.mark.skipif(not (os.path.isdir(os.path.join(REF, "climt/_lib/rrtmg_lw")) and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "librrtmg_lw_ref.so"))),
                    reason="needs the reference checkout and oracle/_ref (build container)")
def test_netcdf_layout_is_read_into_the_same_blob(tmp_path):
    """The other form AER ships the longwave data in: rrtmg_lw.nc, read by the reference's rrtmg_lw_read_nc.f90.
    tools/lw_netcdf.py takes every loader's hyperslabs (variable, target array, start, count, absorber index) from the text
    of that file; a stand-in rrtmg_lw.nc written in this layout from today's raw tables must come back complete -- every
    element of every raw table of rrlw_kg01..16 set -- and `pack_tables.py lw --from-nc` must pack the shipped blob from it,
    flag cleared."""
    from tools.lw_netcdf import parse_read_nc, read_lw_netcdf, write_lw_netcdf
    from tools.pack_tables import read_blob
    from tools.write_lw_k_g import raw_tables_of_blob
    plan = parse_read_nc()
    assert sorted(plan) == list(range(1, 17)) and sum(len(v) for v in plan.values()) == 115
    raw = raw_tables_of_blob(read_blob(SHIPPED))
    raw.pop((2, "refparam"), None)                 # declared by rrlw_kg02, read and used nowhere
    nc = str(tmp_path / "rrtmg_lw.nc")
    write_lw_netcdf(nc, raw)
    got = read_lw_netcdf(nc)
    assert set(got) == set(raw)
    assert all(not np.isnan(a).any() for a in got.values())
    assert all(np.array_equal(got[k], raw[k]) for k in raw)
    out, fix = str(tmp_path / "from_nc.bin"), str(tmp_path / "reduced.npz")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pack_tables.py"), "lw", "--from-nc", nc, "--out", out, "--fixture-out", fix])
    assert open(out, "rb").read() == open(blob_with_flag(str(tmp_path / "flag0.bin"), 0), "rb").read()


def test_longwave_cache_comparisons_execute_on_an_ingested_blob(tmp_path, monkeypatch):
    """CPU twin of tests/test_gpu_parity.py::test_longwave_cache_comparisons_execute_on_an_ingested_blob (host emulation of the
    device functions instead of the GPU): the flag alone decides whether the class refuses and whether values are compared."""
    import climt_amd
    from climt_amd.rrtmg import longwave
    from helpers import EmuContext
    monkeypatch.setattr(longwave, "make_context", lambda device: EmuContext(device))
    monkeypatch.delenv("RRTMG_HIP_ALLOW_SYNTHETIC_LW", raising=False)
    with pytest.raises(RuntimeError, match="SYNTHETIC"):
        climt_amd.RRTMGLongwave()                                  # the shipped blob: refused
    monkeypatch.setenv("RRTMG_HIP_LW_DATA", blob_with_flag(str(tmp_path / "ingested.bin"), 0))
    ran, failed = compare_lw_caches(lambda **kw: climt_amd.RRTMGLongwave(**kw))      # no allow_synthetic_tables: accepted
    assert ran >= 4 * 7, ran
    assert failed > 0      # stand-in data: physical values cannot match; with the real file this is `failed == 0`
