"""CPU tests of the Python host layer (component classes, sympl stand-in, helpers).  The library context is
replaced by the host emulator of the device functions (tests/helpers.EmuContext) -- test infrastructure only."""
import inspect
import json
import logging
import os

import numpy as np
import pytest

import climt_amd
from climt_amd import _sympl_compat as sc
from climt_amd._util import get_interface_values, mass_to_volume_mixing_ratio
from climt_amd.rrtmg import common, longwave, shortwave
from helpers import GOLDEN, LW_CACHE_CLASSES, LWCLASS_CASES, ROOT, EmuContext, check_lwclass_case, load_cache_case, maxdiff


@pytest.fixture(autouse=True)
def emulated_context(monkeypatch):
    def mk(device):
        return EmuContext(device)
    monkeypatch.setattr(shortwave, "make_context", mk)
    monkeypatch.setattr(longwave, "make_context", mk)


REF_IF = json.load(open(os.path.join(GOLDEN, "reference_interface.json")))


@pytest.mark.parametrize("cls", [climt_amd.RRTMGShortwave, climt_amd.RRTMGLongwave, climt_amd.Instellation, climt_amd.BergerSolarInsolation,
                                 climt_amd.SlabSurface])
def test_interface_identical_to_reference(cls):
    """class attributes, the three property dicts and constructor defaults equal the reference's."""
    ref = REF_IF[cls.__name__]
    for name, val in ref.items():
        if name == "__init__":
            sig = inspect.signature(cls.__init__)
            for k, v in val.items():
                assert sig.parameters[k].default == v, (k, sig.parameters[k].default, v)
            continue
        assert getattr(cls, name) == val, name
    for k, v in REF_IF["options"].items():
        assert getattr(common, k) == v


def _check_against_cache(comp, cls, desc, tol):
    state, tend, diag = load_cache_case(cls, desc)
    np.random.seed(0)       # tests/test_components.py:148
    t, d = comp(state)
    assert set(t) == set(tend) and set(d) == set(diag)
    for got, exp in ((t, tend), (d, diag)):
        for k in exp:
            assert got[k].attrs["units"].replace("degK", "K") == exp[k].attrs["units"].replace("degK", "K")
            assert set(got[k].dims) == set(exp[k].dims)
            g = np.transpose(got[k].values, [got[k].dims.index(x) for x in exp[k].dims])
            assert not np.isnan(g).any()
            if tol is not None:
                assert maxdiff(g, exp[k].values) <= tol, (k, maxdiff(g, exp[k].values))
    return t, d


def test_shortwave_reproduces_reference_cache_column():
    """The reference's own regression criterion |d| <= 1e-8 (tests/test_components.py:355-356)."""
    _check_against_cache(climt_amd.RRTMGShortwave(), "TestRRTMGShortwave", "column", 1e-8)


def test_shortwave_mcica_reproduces_reference_cache():
    # TestRRTMGShortwaveMCICA: default Mersenne twister, seed drawn after np.random.seed(0) = 209652396
    _check_against_cache(climt_amd.RRTMGShortwave(mcica=True), "TestRRTMGShortwaveMCICA", "3d", 1e-8)
    _check_against_cache(climt_amd.RRTMGShortwave(mcica=True), "TestRRTMGShortwaveMCICA", "column", 1e-8)


@pytest.mark.parametrize("cls,desc,kw", LW_CACHE_CLASSES)
def test_longwave_on_reference_states(cls, desc, kw):
    """The reference's four longwave cache classes (tests/test_components.py:435-480).  While the table blob is SYNTHETIC
    the cached numbers cannot match; the VALUES the class returns on these four states are then pinned to the reference
    Fortran on the same tables (ref_lwclass_<class>-<desc>.npz: the reference's host layer restated independently in
    tests/golden/make_golden.py) at 1e-9.  The day the real table file is packed the comparison against the caches switches
    itself on at the reference's own criterion, 1e-8."""
    comp = climt_amd.RRTMGLongwave(allow_synthetic_tables=True, **kw)
    if comp._ctx.lw_tables_synthetic():
        t, d = _check_against_cache(comp, cls, desc, None)
        assert check_lwclass_case("%s-%s" % (cls, desc), lambda **k: climt_amd.RRTMGLongwave(allow_synthetic_tables=True, **k)) <= 1e-9
    else:
        t, d = _check_against_cache(comp, cls, desc, 1e-8)
    assert np.array_equal(d["air_temperature_tendency_from_longwave"].values, t["air_temperature"].values)


@pytest.mark.parametrize("name", [n for n in LWCLASS_CASES if not n.startswith("TestRRTMG")])
def test_longwave_class_values_on_perturbed_states(name):
    """Every optional input of the class non-trivial along every axis (band-dependent emissivity, aerosol and cloud optical
    depths, clouds of both phases, trace gases, humidity), each cloud option / overlap / generator once, one state handed over
    with its axes in another order and external interface temperatures: a transposed axis, a missing unit factor or a wrong
    flag changes these numbers by W m-2, the bar is 1e-9."""
    check_lwclass_case(name, lambda **k: climt_amd.RRTMGLongwave(allow_synthetic_tables=True, **k))


def test_change_up_flux_arrays_are_not_overwritten_by_the_next_call():
    """calculate_change_up_flux=True publishes dF/dTs on the instance; an array a caller kept from call N holds call N's
    numbers after call N+1 (they come from the liveness-tracked output pool like every other result)."""
    lw = climt_amd.RRTMGLongwave(allow_synthetic_tables=True, calculate_change_up_flux=True)
    state, _, _ = load_cache_case("TestRRTMGLongwave", "column")
    lw(state)
    kept = lw.change_in_upward_flux_with_surface_temperature
    snapshot = kept.copy()
    assert np.all(snapshot > 0.0)
    ts = state["surface_temperature"]
    state["surface_temperature"] = type(ts)(ts.values + 15.0, dims=ts.dims, attrs=ts.attrs)
    lw(state)
    assert lw.change_in_upward_flux_with_surface_temperature is not kept
    assert np.array_equal(kept, snapshot)
    assert maxdiff(lw.change_in_upward_flux_with_surface_temperature, snapshot) > 0.0


def test_config1_radiative_equilibrium_loop_on_the_host_emulation():
    """BASELINE configs[0] as far as a GPU-less container goes: examples/radiative_equilibrium.py (the reference's
    examples/radiative_equilibrium_rrtmg.py:43-66 with `from climt_amd import ...`) for six steps on the host emulation of the
    device functions; step 0's shortwave diagnostics are the reference's TestRRTMGShortwave-column cache (1e-8)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("radiative_equilibrium", os.path.join(ROOT, "examples", "radiative_equilibrium.py"))
    ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
    first, end = ex.run(6, device_resident=False)
    _, _, cache_diag = load_cache_case("TestRRTMGShortwave", "column")
    for k, want in cache_diag.items():
        g = np.transpose(first[k].values, [first[k].dims.index(x) for x in want.dims])
        assert maxdiff(g, want.values) <= 1e-8, k
    t = end["air_temperature"].values
    assert np.all(np.isfinite(t)) and t.ravel()[-1] != 290.0


def test_state_axis_permutation_invariance():
    """tests/test_components.py:291-327: reversed / transposed horizontal axes give the same answer."""
    state, _, _ = load_cache_case("TestRRTMGShortwaveMCICA", "3d")
    comp = climt_amd.RRTMGShortwave()
    for st in state.values():
        if hasattr(st, "values") and "mid_levels" in st.dims:
            pass
    state["cloud_area_fraction_in_atmosphere_layer"].values[:] = 0.0
    t0, d0 = comp(state)
    tr = {}
    for k, v in state.items():
        if hasattr(v, "dims") and "lat" in v.dims and "lon" in v.dims:
            order = list(range(v.values.ndim))
            i, j = v.dims.index("lat"), v.dims.index("lon")
            order[i], order[j] = order[j], order[i]
            dims = list(v.dims); dims[i], dims[j] = dims[j], dims[i]
            tr[k] = sc.DataArray(np.transpose(v.values, order), dims=dims, attrs=v.attrs)
        else:
            tr[k] = v
    t1, d1 = comp(tr)
    for k in d0:
        a = d0[k]
        b = d1[k]
        bb = np.transpose(b.values, [b.dims.index(x) for x in a.dims])
        assert np.array_equal(a.values, bb), k


def test_mcica_log_messages(caplog):
    """messages asserted by tests/test_components.py:454-461, :507-530"""
    with caplog.at_level(logging.INFO):
        climt_amd.RRTMGShortwave(mcica=True, cloud_overlap_method="clear_only")
        assert "no clouds" in caplog.text.lower()
        caplog.clear()
        climt_amd.RRTMGShortwave(mcica=True, cloud_optical_properties="single_cloud_type")
        assert "must be 'direct_input' or 'liquid_and_ice_clouds'" in caplog.text
        caplog.clear()
        climt_amd.RRTMGShortwave(mcica=True, cloud_ice_properties="ebert_curry_one")
        assert "should not be set to 'ebert_curry_one'" in caplog.text
        caplog.clear()
        climt_amd.RRTMGShortwave(mcica=True, cloud_liquid_water_properties="radius_independent_absorption")
        assert "must be set to 'radius_dependent_absorption'" in caplog.text
        caplog.clear()
        climt_amd.RRTMGLongwave(mcica=True, cloud_overlap_method="clear_only", allow_synthetic_tables=True)
        assert "no clouds" in caplog.text.lower()


def test_longwave_fails_closed_on_synthetic_tables(monkeypatch):
    """A drop-in that silently integrates non-physical longwave forcing is worse than one that refuses: on the synthetic
    k-tables the component raises unless the caller opts in (keyword or environment)."""
    monkeypatch.delenv("RRTMG_HIP_ALLOW_SYNTHETIC_LW", raising=False)
    with pytest.raises(RuntimeError, match="SYNTHETIC"):
        climt_amd.RRTMGLongwave()
    climt_amd.RRTMGLongwave(allow_synthetic_tables=True)
    monkeypatch.setenv("RRTMG_HIP_ALLOW_SYNTHETIC_LW", "1")
    climt_amd.RRTMGLongwave()


def test_host_helpers():
    q = np.array([[0.01, 0.02]])
    assert np.allclose(mass_to_volume_mixing_ratio(q, 18.02), q * 28.964 / 18.02)
    with pytest.raises(ValueError):
        mass_to_volume_mixing_ratio(q)
    p = np.array([[900.0], [700.0], [400.0]]); pi = np.array([[1000.0], [800.0], [550.0], [250.0]])
    t = np.array([[290.0], [270.0], [240.0]]); ts = np.array([300.0])
    ti = get_interface_values(t, ts, p, pi)
    assert ti.shape == (4, 1) and ti[0, 0] == 300.0 and ti[-1, 0] == 240.0
    w = (np.log(800.0) - np.log(700.0)) / (np.log(900.0) - np.log(700.0))
    assert np.isclose(ti[1, 0], 270.0 - w * (270.0 - 290.0))


def test_unit_conversion_of_the_sympl_standin():
    if sc.HAVE_SYMPL:
        pytest.skip("real sympl present")
    assert np.isclose(sc.convert_units(101320.0, "Pa", "mbar"), 1013.2)
    assert np.isclose(sc.convert_units(0.3, "kg/m**2", "g m^-2"), 300.0)
    assert sc.convert_units(5.0, "\xb5m", "micrometer") == 5.0
    with pytest.raises(ValueError):
        sc.convert_units(1.0, "Pa", "K")


def test_staged_input_products_and_extraction_plans():
    """The components form their input products (unit conversions, water-vapour mixing ratio) in the background into kept
    buffers (InputStaging) and work the route of every input out once per state STRUCTURE: the values are those of the plain
    expressions, and a state whose units or axis order change gets its own plan."""
    from climt_amd.rrtmg.common import InputStaging
    from climt_amd._util import mass_to_volume_mixing_ratio
    rng = np.random.default_rng(3)
    q = rng.uniform(0, 0.02, (60, 500))
    st = InputStaging()
    a = st.scaled("q", q, 28.964, 18.02, pieces=4); b = st.scaled("p", q, 0.01)
    st.wait()
    assert np.array_equal(a, mass_to_volume_mixing_ratio(q, 18.02)) and np.array_equal(b, q * 0.01)
    assert st.scaled("q", q, 28.964, 18.02) is a     # the buffer is kept
    st.wait()
    if sc.HAVE_SYMPL:
        return
    state, _, _ = load_cache_case("TestRRTMGShortwave", "column")
    comp = climt_amd.RRTMGShortwave()
    t0, d0 = comp(state)
    other = dict(state)
    p = state["air_pressure"]
    other["air_pressure"] = sc.DataArray(p.values / 100.0, dims=p.dims, attrs={"units": "hPa"})
    t1, d1 = comp(other)
    assert len(comp._plans) == 2
    for k in d0:
        assert np.allclose(d0[k].values, d1[k].values, rtol=1e-12, atol=1e-12), k
    t2, d2 = comp(state)          # back to the first structure: its plan is still there
    assert len(comp._plans) == 2 and all(np.array_equal(d0[k].values, d2[k].values) for k in d0)


def test_unconverted_inputs_travel_under_raw_keys_only_for_components_that_ask():
    """the stand-in's extraction hands an input over unconverted (its unit factor applied by the library on the device)
    only to a component that names it in `_unit_factor_on_device`, and then under name + "@raw" -- state[name] is absent, so
    nothing on the host can read it in another unit than input_properties declares; every other component sees converted
    arrays under their names and no extra key."""
    if sc.HAVE_SYMPL:
        pytest.skip("real sympl present")
    from climt_amd.rrtmg.common import RAW, library_scales
    state, _, _ = load_cache_case("TestRRTMGShortwave", "column")
    comp = climt_amd.RRTMGShortwave()
    raw = comp._extract(state)
    assert "air_pressure" not in raw and raw["air_pressure" + RAW].max() > 5.0e4        # Pa, as the state holds it
    assert raw["_unit_factors"]["air_pressure"] == pytest.approx(0.01)
    scales, unit = library_scales(raw)
    assert scales["pressure_scale"] == pytest.approx(0.01) and unit["air_pressure"] is raw["air_pressure" + RAW]
    # the two pressures in different units: the pair is converted on the host, into the unit input_properties declares
    other = dict(state)
    p = state["air_pressure"]
    other["air_pressure"] = sc.DataArray(p.values / 100.0, dims=p.dims, attrs={"units": "hPa"})
    raw2 = comp._extract(other)
    scales2, unit2 = library_scales(raw2)
    assert "pressure_scale" not in scales2 and np.allclose(unit2["air_pressure"], p.values.reshape(unit2["air_pressure"].shape) / 100.0)
    assert np.allclose(unit2["air_pressure_on_interface_levels"].max(), state["air_pressure_on_interface_levels"].values.max() / 100.0)
    # a component that does not opt in
    class Plain(sc.DiagnosticComponent):
        input_properties = {"air_pressure": {"dims": ["mid_levels", "*"], "units": "mbar"}}
        diagnostic_properties = {}

        def array_call(self, st):
            return {}
    praw = Plain()._extract(state)
    assert "_unit_factors" not in praw and not any(k.endswith(RAW) for k in praw if isinstance(k, str))
    assert praw["air_pressure"].max() < 2.0e3          # converted to mbar on the host, under its own name


def test_instellation_time_arithmetic_matches_oracle():
    """days since 2000-01-01 12:00 (the only host arithmetic of the Instellation drop-in), incl. sub-second times."""
    import datetime
    from climt_amd import instellation
    from oracle import instellation_oracle as orc
    for t in (datetime.datetime(2000, 1, 1), datetime.datetime(1999, 12, 31, 23, 59, 59, 250000), datetime.datetime(2031, 7, 4, 6, 30),
              datetime.datetime(1850, 3, 1, 12)):
        assert instellation.days_from_2000(t) == orc.days_from_2000(t)


def test_berger_orbital_series_and_time_helpers_match_oracle():
    """The host part of the BergerSolarInsolation drop-in (orbital series of a year, vernal-equinox year fraction, day
    fraction) against the oracle's restatement, bit for bit (both are numpy on the same packed tables)."""
    import datetime
    from climt_amd import berger
    from oracle import berger_oracle as orc
    for year in (1950, 2000, 2017, 1850, 2300):
        assert berger.get_orbital_parameters(float(year - 1950)) == orc.orbital_parameters(float(year - 1950))
    for t in (datetime.datetime(2000, 1, 1), datetime.datetime(2000, 3, 20, 12), datetime.datetime(2016, 2, 29, 23, 59, 59), datetime.datetime(1999, 12, 31, 6)):
        assert berger.years_since_vernal_equinox(t) == orc.years_since_vernal_equinox(t)
        assert berger.fractional_day(t) == orc.fractional_day(t)


def test_committed_bench_lines_carry_the_contract_fields():
    """profiles/r01_bench_default*.json are bench.py's stdout on the GPU box: the keys the driver and the judge read."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default*.json")))
    assert files
    for fn in files:
        j = json.load(open(fn))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in j, (fn, k)
        assert j["unit"] == "columns/s" and j["higher_is_better"] is True and j["scaling"] == "weak" and j["dtype"] == "f64"
        assert j["vs_baseline"] is None and "workload" in j["config"] and "model" not in j["config"]
        r = j["roofline"]
        assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and (r["traffic"] is None or r["traffic"] > 0)
        # achieved = algorithmic bytes per launch / event-timed kernel duration
        n = j["config"]["columns_per_gpu"]
        assert abs(r["achieved"] - r["algorithmic_bytes_per_column"] * n / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
        c = j["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "columns/s" and c["sample"]
        assert abs(j["value"] - j["n_gpus"] * n / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]


def _cacheout(cls):
    z = np.load(os.path.join(GOLDEN, "climt_cacheout_%s-3d.npz" % cls))
    out = {"tend": {}, "diag": {}, "stepdiag": {}}
    for k in z.files:
        grp, name, what = k.split("/")
        out[grp].setdefault(name, {})[what] = z[k]
    return out


def check_3d_cache_from_generated_state(cls):
    """The reference's 32 x 16 x 28 cache classes whose stored input state is a missing blob: the state is the plain default
    state of its test (tests/test_components.py:250-255), rebuilt by climt_amd.get_default_state; outputs of the call and the
    diagnostics of the 10 s Adams-Bashforth step against the caches at the reference's 1e-8 (longwave: once the tables are real)."""
    import datetime as dtm
    comp = climt_amd.RRTMGShortwave() if cls == "TestRRTMGShortwave" else climt_amd.RRTMGLongwave(allow_synthetic_tables=True)
    tol = 1e-8 if cls == "TestRRTMGShortwave" or not comp._ctx.lw_tables_synthetic() else None
    exp = _cacheout(cls)
    state = climt_amd.get_default_state([comp], grid_state=climt_amd.get_grid(nx=32, ny=16, nz=28))
    t, d = comp(state)
    stepd, _ = climt_amd.AdamsBashforth(comp)(state, dtm.timedelta(seconds=10))
    n = 0
    for got, want in ((t, exp["tend"]), (d, exp["diag"]), (stepd, exp["stepdiag"])):
        assert set(want) <= set(got)
        for k, w in want.items():
            dims = tuple(x for x in str(w["dims"]).split(",") if x)
            g = np.transpose(got[k].values, [got[k].dims.index(x) for x in dims])
            assert g.shape == w["values"].shape and not np.isnan(g).any()
            assert tol is None or maxdiff(g, w["values"]) <= tol, (cls, k, maxdiff(g, w["values"]))
            n += 1
    assert n >= 12


@pytest.mark.parametrize("cls", ["TestRRTMGShortwave", "TestRRTMGLongwave"])
def test_3d_caches_from_generated_default_state(cls):
    check_3d_cache_from_generated_state(cls)


def _reorder_state(state, pattern):
    """The reference's test_reversed_state_gives_same_output (tests/test_components.py:291-308: every 3-d quantity transposed to
    (dims[2], dims[1], dims[0]), every 2-d one to (dims[1], dims[0])) and test_transposed_state_gives_same_output (:310-327:
    (dims[2], dims[0], dims[1])) -- the VERTICAL axis moves too."""
    out = {}
    for name, v in state.items():
        if not hasattr(v, "dims"):
            out[name] = v
            continue
        nd = len(v.dims)
        if nd == 3:
            order = (2, 1, 0) if pattern == "reversed" else (2, 0, 1)
        elif nd == 2:
            order = (1, 0)
        else:
            out[name] = v
            continue
        out[name] = sc.DataArray(np.ascontiguousarray(np.transpose(v.values, order)), dims=[v.dims[i] for i in order], attrs=v.attrs)
    return out


@pytest.mark.parametrize("pattern", ["reversed", "transposed"])
def test_reversed_and_transposed_states_give_the_cached_output(pattern):
    """tests/test_components.py:291-327 for the two radiation classes on the reference's 32 x 16 x 28 default state: the
    shortwave against the reference's 3-d cache (1e-8, its criterion), the longwave (synthetic tables: no cache to meet) against
    its own output on the untouched state, bit for bit -- columns are independent and the extraction only re-labels axes."""
    for cls, comp in (("TestRRTMGShortwave", climt_amd.RRTMGShortwave()), ("TestRRTMGLongwave", climt_amd.RRTMGLongwave(allow_synthetic_tables=True))):
        nx, ny = (32, 16) if cls == "TestRRTMGShortwave" else (6, 4)          # (the cache is 32 x 16; the emulation is slow)
        state = climt_amd.get_default_state([comp], grid_state=climt_amd.get_grid(nx=nx, ny=ny, nz=28))
        t0, d0 = comp(state)
        t1, d1 = comp(_reorder_state(state, pattern))
        for a, b in ((t0, t1), (d0, d1)):
            for k in a:
                bb = np.transpose(b[k].values, [b[k].dims.index(x) for x in a[k].dims])
                assert np.array_equal(a[k].values, bb), (cls, pattern, k)
        if cls == "TestRRTMGShortwave":
            exp = _cacheout(cls)
            for got, want in ((t1, exp["tend"]), (d1, exp["diag"])):
                for k, w in want.items():
                    dims = tuple(x for x in str(w["dims"]).split(",") if x)
                    g = np.transpose(got[k].values, [got[k].dims.index(x) for x in dims])
                    assert maxdiff(g, w["values"]) <= 1e-8, (pattern, k)


def test_output_pool_never_hands_out_an_array_somebody_still_holds():
    """climt_amd.rrtmg.common.OutputPool: the radiation components write into the arrays of an EARLIER call only when the
    caller has dropped every reference to them (DataArray, raw array, any view) -- the reference allocates afresh each call,
    and a recycled array must be indistinguishable from that."""
    from climt_amd.rrtmg.common import OutputPool, output_arrays
    pool = OutputPool()

    def mem(x):
        return x.__array_interface__["data"][0]
    a = pool.zeros_like_fresh("x", (3, 4))
    b = pool.zeros_like_fresh("x", (3, 4))
    assert mem(a) != mem(b) and not a.any()                # `a` is held: other memory
    mem_a = mem(a)
    a[:] = 7.0
    view = a[1]                                            # a view keeps the hand-out alive through its .base chain
    held = [a]                                             # ... and so does any other reference (the advisor's case:
    del a                                                  # an interpreter whose reference counts read differently)
    c = pool.zeros_like_fresh("x", (3, 4))
    assert mem(c) not in (mem_a, mem(b))
    del held[:]
    e = pool.zeros_like_fresh("x", (3, 4))                 # the slice `view` is still alive
    assert mem(e) not in (mem_a, mem(b), mem(c)) and (view == 7.0).all()
    flat = view.reshape(-1)[::2]                           # views of views end at the same root
    del view
    f = pool.zeros_like_fresh("x", (3, 4))
    assert mem(f) != mem_a and (flat == 7.0).all()
    del flat
    d = pool.zeros_like_fresh("x", (3, 4))
    assert mem(d) == mem_a                                 # nobody can see it any more: handed out again
    assert pool.zeros_like_fresh("x", (4, 3)).shape == (4, 3) and mem(pool.zeros_like_fresh("y", (3, 4))) != mem(d)
    assert d.flags.writeable and d.flags.c_contiguous and d.dtype == np.float64
    # shapes as initialize_numpy_arrays_with_properties derives them
    props_in = {"t": {"dims": ["mid_levels", "*"], "units": "K"}, "p": {"dims": ["interface_levels", "*"], "units": "Pa"}}
    raw = {"t": np.zeros((5, 7)), "p": np.zeros((6, 7))}
    out = output_arrays(pool, {"f": {"dims": ["interface_levels", "*"], "units": "W m^-2"}, "t": {"units": "K s^-1"}}, raw, props_in)
    assert out["f"].shape == (6, 7) and out["t"].shape == (5, 7)
