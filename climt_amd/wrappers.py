"""`UpdateFrequencyWrapper`: call the wrapped component only every `update_timedelta` of model time and hand back the
cached tendencies / diagnostics in between -- how model scripts (examples/gmd_aquaplanet.py:61-76) run radiation
less often than the dynamics.  It is sympl's class (un-vendored dependency of the reference; semantics as documented
in sympl: the first call always computes, later calls compute when state["time"] >= last update + update_timedelta,
attribute access falls through to the wrapped component); sympl's own is used when sympl is installed."""
from datetime import timedelta

try:  # pragma: no cover - sympl is absent in the build container
    from sympl import UpdateFrequencyWrapper  # noqa: F401
except ImportError:

    class UpdateFrequencyWrapper:
        def __init__(self, component, update_timedelta):
            if not isinstance(update_timedelta, timedelta):
                raise TypeError("update_timedelta must be a datetime.timedelta, got %r" % (update_timedelta,))
            self.component = component
            self._update_timedelta = update_timedelta
            self._cached_output = None
            self._last_update_time = None

        def __call__(self, state, timestep=None, **kwargs):
            now = state["time"]
            if self._last_update_time is None or now >= self._last_update_time + self._update_timedelta:
                if timestep is not None:
                    kwargs["timestep"] = timestep
                self._cached_output = self.component(state, **kwargs)
                self._last_update_time = now
            return self._cached_output

        def __getattr__(self, item):
            return getattr(self.component, item)
