"""The slice of `sympl` the RRTMG components need.

climt's RRTMG components are `sympl.TendencyComponent`s (lw/component.py:30, sw/component.py:32).
When sympl is installed the real base class, `get_constant` and
`initialize_numpy_arrays_with_properties` are used unchanged, so the components drop into existing
climt/sympl model scripts.  When it is not (this build container), a small stand-in provides the
same call contract for state dictionaries of `DataArray`-like objects (`.values`, `.dims`,
`.attrs["units"]`): unit conversion to the component's `input_properties`, re-ordering of dims to
the listed ones with every unlisted dim collapsed into `"*"`, `array_call`, and re-expansion of the
outputs (last bullet).
"""
import numpy as np

try:  # pragma: no cover - sympl is absent in the build container
    from sympl import DataArray, DiagnosticComponent, TendencyComponent, get_constant, initialize_numpy_arrays_with_properties  # noqa: F401
    HAVE_SYMPL = True
except ImportError:
    HAVE_SYMPL = False

    class DataArray:
        """Minimal xarray.DataArray look-alike."""

        def __init__(self, values, dims=(), attrs=None):
            self.values = np.asarray(values)
            self.dims = tuple(dims)
            self.attrs = dict(attrs or {})
            if self.values.ndim != len(self.dims):
                raise ValueError("dims %s do not match array of shape %s" % (self.dims, self.values.shape))

        @property
        def shape(self):
            return self.values.shape

        def item(self):
            return self.values.item()

        def __repr__(self):
            return "DataArray(%s, dims=%s, units=%s)" % (self.values.shape, self.dims, self.attrs.get("units"))

    # values that reproduce the reference's golden caches (sympl default constants)
    _CONSTANTS = {
        ("gravitational_acceleration", "m/s^2"): 9.80665,
        ("planck_constant", "erg s"): 6.62607004e-27,
        ("boltzmann_constant", "erg K^-1"): 1.38064852e-16,
        ("speed_of_light", "cm s^-1"): 2.99792458e10,
        ("avogadro_constant", "mole^-1"): 6.022140857e23,
        ("loschmidt_constant", "cm^-3"): 2.6867774e19,
        ("universal_gas_constant", "erg mol^-1 K^-1"): 8.3144598e7,
        ("stefan_boltzmann_constant", "W cm^-2 K^-4"): 5.670367e-12,
        ("seconds_per_day", "dimensionless"): 86400.0,
        ("heat_capacity_of_dry_air_at_constant_pressure", "J/kg/K"): 1004.64,
        ("stellar_irradiance", "W/m^2"): 1367.0,
        ("heat_capacity_of_dry_air_at_constant_pressure", "J kg^-1 K^-1"): 1004.64,
        ("gas_constant_of_dry_air", "J kg^-1 K^-1"): 287.0,
        ("reference_air_pressure", "Pa"): 1.0132e5,
        ("top_of_model_pressure", "Pa"): 20.0,
    }
    _constant_overrides = {}

    def get_constant(name, units):
        if (name, units) in _constant_overrides:
            return _constant_overrides[(name, units)]
        try:
            return _CONSTANTS[(name, units)]
        except KeyError:
            raise KeyError("constant %r in units %r is not known to the sympl stand-in" % (name, units))

    def set_constant(name, value, units):
        _constant_overrides[(name, units)] = float(value)

    _UNIT_ALIASES = {
        "mbar": "hPa", "millibar": "hPa", "degK": "K", "kelvin": "K", "dimensionless": "1", "": "1", "g/g": "1", "kg/kg": "1",
        "kg kg^-1": "1", "g g^-1": "1", "mole/mole": "1", "micrometer": "um", "micron": "um", "µm": "um", "\xb5m": "um",
        "kg/m**2": "kg m^-2", "kg/m^2": "kg m^-2", "g/m^2": "g m^-2", "W/m^2": "W m^-2", "W/m**2": "W m^-2", "K/day": "K day^-1",
        "J/(degK*kg)": "J kg^-1 K^-1", "J kg^-1 degK^-1": "J kg^-1 K^-1", "J/kg/K": "J kg^-1 K^-1", "kg/m**3": "kg m^-3", "kg/m^3": "kg m^-3",
        "degK day^-1": "K day^-1", "degK/day": "K day^-1", "radian": "radians", "rad": "radians",
    }
    _TO_BASE = {"hPa": ("Pa", 100.0), "Pa": ("Pa", 1.0), "kPa": ("Pa", 1000.0), "kg m^-2": ("kg m^-2", 1.0), "g m^-2": ("kg m^-2", 1.e-3),
                "K day^-1": ("K s^-1", 1.0 / 86400.0), "K s^-1": ("K s^-1", 1.0), "degK s^-1": ("K s^-1", 1.0),
                "degrees": ("radians", np.pi / 180.0), "radians": ("radians", 1.0),
                "degrees_north": ("degrees_north", 1.0), "degrees_N": ("degrees_north", 1.0),
                "degrees_east": ("degrees_east", 1.0), "degrees_E": ("degrees_east", 1.0), "m": ("m", 1.0), "um": ("m", 1.e-6)}

    def _canon(u):
        u = (u or "").strip()
        return _UNIT_ALIASES.get(u, u)

    def convert_units(values, src, dst, scale=None):
        """`scale(values, factor)`: forms the product instead of `values * factor` when given (a component's input staging)."""
        s, d = _canon(src), _canon(dst)
        if s == d:
            return values
        if s in _TO_BASE and d in _TO_BASE and _TO_BASE[s][0] == _TO_BASE[d][0]:
            f = _TO_BASE[s][1] / _TO_BASE[d][1]
            return scale(values, f) if scale is not None else values * f
        raise ValueError("cannot convert units %r -> %r" % (src, dst))

    def initialize_numpy_arrays_with_properties(output_properties, raw_input_state, input_properties, dtype=np.float64):
        """Zero-filled output arrays shaped from the dims recorded while extracting the inputs."""
        lengths = {}
        for name, prop in input_properties.items():
            if name in raw_input_state and isinstance(raw_input_state[name], np.ndarray):
                for dim, n in zip(prop.get("dims", ()), raw_input_state[name].shape):
                    lengths[dim] = n
        out = {}
        for name, prop in output_properties.items():
            dims = prop.get("dims")
            if dims is None:
                dims = input_properties[name]["dims"]
            out[name] = np.zeros([lengths[d] for d in dims], dtype=dtype)
        return out

    class TendencyComponent:
        input_properties = {}
        tendency_properties = {}
        diagnostic_properties = {}

        def __init__(self, tendencies_in_diagnostics=False, name=None, **kwargs):
            self.name = name or self.__class__.__name__.lower()

        # -- state -> raw arrays ------------------------------------------------------------
        def _plan(self, state):
            """How each input gets from the state to array_call, worked out once per state STRUCTURE (dims, units, shapes):
            per input (factor or None, axis order or None, shape), plus the wildcard dims and the named dim lengths."""
            steps, lengths = [], {}
            wild_names, wild_shape = None, None
            for name, prop in self.input_properties.items():
                da = state[name]
                shape_in, dims = np.shape(da.values), tuple(da.dims)
                numeric = np.asarray(da.values).dtype.kind in "fiub"     # string quantities (area_type) pass through
                factor = None
                if numeric:
                    probe = convert_units(np.ones(()), da.attrs.get("units", ""), prop.get("units", da.attrs.get("units", "")))
                    factor = None if float(probe) == 1.0 else float(probe)
                want = list(prop["dims"])
                named = [d for d in want if d != "*"]
                for d in named:
                    if d not in dims:
                        raise ValueError("quantity %r lacks dimension %r" % (name, d))
                wild = [d for d in dims if d not in named]
                if "*" in want:
                    if wild_names is None:
                        wild_names = wild
                        wild_shape = [shape_in[dims.index(d)] for d in wild]
                    elif wild and wild != wild_names:
                        # same wildcard dims in another order are transposed to the first ordering
                        if sorted(wild) != sorted(wild_names):
                            raise ValueError("inconsistent wildcard dimensions for %r: %s vs %s" % (name, wild, wild_names))
                        wild = wild_names
                elif wild:
                    raise ValueError("quantity %r has unexpected dimensions %s" % (name, wild))
                order = []
                for d in want:
                    order.extend([dims.index(w) for w in (wild if d == "*" else [d])])
                shape = []
                for d in want:
                    if d == "*":
                        shape.append(int(np.prod([shape_in[dims.index(w)] for w in wild])) if wild else 1)
                    else:
                        shape.append(shape_in[dims.index(d)])
                        lengths[d] = shape[-1]
                identity = order == list(range(len(order)))
                steps.append((name, numeric, factor, None if identity else order, tuple(shape) if want else None))
            return steps, lengths, wild_names or [], wild_shape or []

        def _extract(self, state):
            raw = {"time": state.get("time")}
            staging = getattr(self, "_input_staging", None)   # see climt_amd/rrtmg/common.py: the products form in the background
            try:
                sig = tuple((tuple(state[n].dims), state[n].attrs.get("units", ""), np.shape(state[n].values)) for n in self.input_properties)
            except KeyError as e:
                raise KeyError("state is missing input quantity %r" % e.args[0])
            plans = self.__dict__.setdefault("_plans", {})
            plan = plans.get(sig)
            if plan is None:
                plan = plans[sig] = self._plan(state)
            steps, self._dim_lengths, self._wild_names, self._wild_shape = plan
            # components whose library applies unit factors on the device name the inputs it may do that for
            # (climt_amd/rrtmg: pressures, cloud water paths): those go through unconverted under the key name + "@raw", the
            # factor beside them in "_unit_factors" -- raw[name] itself is then ABSENT, so that whatever reads state[name] on the
            # host always gets the unit input_properties declares, or a KeyError, never a silently different unit.  Components
            # that do not opt in see neither key.
            on_device = getattr(self, "_unit_factor_on_device", ())
            unit_factors = {}
            if on_device:
                raw["_unit_factors"] = unit_factors
            for name, numeric, factor, order, shape in steps:
                values = np.asarray(state[name].values)
                key = name
                if numeric:
                    values = values.astype(np.float64, copy=False)
                    if factor is not None and name in on_device and values.ndim >= 2:
                        unit_factors[name] = factor
                        key = name + "@raw"
                    elif factor is not None:
                        if staging is not None and values.ndim >= 2:
                            values = staging.scaled(name, values, factor)
                            if order is not None:
                                staging.wait()   # the re-ordering below reads the product
                        else:
                            values = values * factor
                if order is not None:
                    values = np.transpose(values, order)
                raw[key] = np.ascontiguousarray(values.reshape(shape)) if shape is not None else values
            if staging is not None:
                staging.wait()
            return raw

        def _wrap(self, arrays, properties):
            out = {}
            for name, arr in arrays.items():
                prop = properties[name]
                dims = prop["dims"] if prop.get("dims") is not None else self.input_properties[name]["dims"]
                arr = np.asarray(arr)
                shape, names = [], []
                for d, n in zip(dims, arr.shape):
                    if d == "*":
                        shape.extend(self._wild_shape)
                        names.extend(self._wild_names)
                    else:
                        shape.append(n)
                        names.append(d)
                out[name] = DataArray(arr.reshape(shape), dims=names, attrs={"units": prop["units"]})
            return out

        def __call__(self, state):
            raw = self._extract(state)
            tendencies, diagnostics = self.array_call(raw)
            return self._wrap(tendencies, self.tendency_properties), self._wrap(diagnostics, self.diagnostic_properties)

    class DiagnosticComponent(TendencyComponent):
        """sympl.DiagnosticComponent stand-in: array_call(raw) -> diagnostics only."""

        def __call__(self, state):
            raw = self._extract(state)
            return self._wrap(self.array_call(raw), self.diagnostic_properties)
