"""`AdamsBashforth`: the tendency stepper the reference's model scripts and stepping tests put around the radiation
components (tests/test_components.py:123-160 wraps every TendencyComponent in AdamsBashforth and steps 10 s).

It is sympl's class (un-vendored dependency of the reference); sympl's own is used when sympl is installed.  The
stand-in keeps the contract the golden `*_stepping` caches pin: tendencies of all wrapped components are summed,
brought to "<state units> per second", the order ramps 1 -> 2 -> 3 as history accumulates (Euler first step), and the
call returns (diagnostics, new_state) where new_state carries every untouched quantity over unchanged.
O(state) numpy work per step next to the radiation kernels' O(state x 252 g-points): left on the host."""
from datetime import timedelta

import numpy as np

from . import _sympl_compat as _sc

try:  # pragma: no cover - sympl is absent in the build container
    from sympl import AdamsBashforth  # noqa: F401
except ImportError:
    _AB = {1: (1.0,), 2: (1.5, -0.5), 3: (23.0 / 12.0, -16.0 / 12.0, 5.0 / 12.0), 4: (55.0 / 24.0, -59.0 / 24.0, 37.0 / 24.0, -9.0 / 24.0)}

    def _per_second(units, state_units):
        """Factor taking a tendency in `units` to state_units s^-1 ('K day^-1' for a state in degK -> 1/86400)."""
        u = _sc._canon(units)
        for suffix, seconds in ((" s^-1", 1.0), ("/s", 1.0), (" day^-1", 86400.0), ("/day", 86400.0)):
            if u.endswith(suffix):
                base = u[:-len(suffix)].strip()
                _sc.convert_units(np.zeros(()), base, state_units)   # raises when the quantities do not match
                return 1.0 / seconds
        raise ValueError("tendency units %r are not a rate" % units)

    class AdamsBashforth:
        def __init__(self, *components, order=3, **kwargs):
            if len(components) == 1 and isinstance(components[0], (list, tuple)):
                components = tuple(components[0])
            if order not in _AB:
                raise ValueError("order must be 1..4")
            self.component_list = list(components)
            self._order = order
            self._history = []
            self._timestep = None

        @property
        def input_properties(self):
            from .initialization import aggregate_input_properties
            return aggregate_input_properties(self.component_list)

        def _tendencies(self, state):
            total, diagnostics = {}, {}
            for comp in self.component_list:
                tend, diag = comp(state)
                overlap = set(diag) & set(diagnostics)
                if overlap:
                    raise ValueError("two components compute the same diagnostics: %s" % sorted(overlap))
                diagnostics.update(diag)
                for name, da in tend.items():
                    rate = np.asarray(da.values, dtype=np.float64) * _per_second(da.attrs.get("units", ""), state[name].attrs.get("units", ""))
                    order = [da.dims.index(d) for d in state[name].dims]
                    rate = np.transpose(rate, order)
                    total[name] = total[name] + rate if name in total else rate
            return total, diagnostics

        def __call__(self, state, timestep):
            if not isinstance(timestep, timedelta):
                raise TypeError("timestep must be a datetime.timedelta")
            if self._timestep is None:
                self._timestep = timestep
            elif timestep != self._timestep:
                raise ValueError("timestep must be constant for Adams-Bashforth time stepping")
            tend, diagnostics = self._tendencies(state)
            self._history = [tend] + self._history[: self._order - 1]
            weights = _AB[len(self._history)]
            dt = timestep.total_seconds()
            new_state = {k: v for k, v in state.items() if k not in tend}
            for name in tend:
                incr = sum(w * h[name] for w, h in zip(weights, self._history))
                old = state[name]
                new_state[name] = _sc.DataArray(np.asarray(old.values) + dt * incr, dims=old.dims, attrs=dict(old.attrs))
            return diagnostics, new_state
