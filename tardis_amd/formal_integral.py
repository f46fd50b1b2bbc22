"""Host-side mirror of the reference's formal integrator interface (tardis/spectrum/formal_integral/formal_integral_numba.py:
563-642, `NumbaFormalIntegrator`) on top of the HIP engine.  The ray integration runs on the device against the engine's
resident geometry, line list, Sobolev optical depths and electron densities."""
from __future__ import annotations

import numpy as np

from .engine import Engine


class FormalIntegratorHIP:
    """Drop-in for NumbaFormalIntegrator(geometry, time_explosion, plasma, n_impact_parameters)."""

    def __init__(self, geometry, time_explosion, plasma, n_impact_parameters: int = 1000, engine: Engine | None = None):
        self.geometry = geometry
        self.time_explosion = float(time_explosion)
        self.plasma = plasma
        self.n_impact_parameters = int(n_impact_parameters)
        self._engine = engine
        self._owns = engine is None

    def _ensure_engine(self, tau_sobolev, electron_densities) -> Engine:
        if self._engine is None:
            from . import state as st
            self._engine = Engine(0)
            self._engine.set_geometry(self.geometry, self.time_explosion)
            L = len(self.plasma.line_list_nu)
            # only line_list_nu, tau_sobolev and electron_density are read by the integrator
            S = len(electron_densities)
            ost = st.OpacityState(np.asarray(electron_densities), np.zeros(S), np.asarray(self.plasma.line_list_nu),
                                  np.asarray(tau_sobolev), np.ones((L, S)), np.arange(L, dtype=np.int64),
                                  np.arange(L + 1, dtype=np.int64), -np.ones(L, dtype=np.int64), np.zeros(L, dtype=np.int64),
                                  np.arange(L, dtype=np.int64))
            self._engine.set_opacity(ost)
        return self._engine

    def formal_integral(self, inner_temperature, frequencies, att_S_ul, mean_intensity_red_lu, mean_intensity_blue_lu, tau_sobolev,
                        electron_densities, n_impact_parameters):
        """Same arguments and return value as NumbaFormalIntegrator.formal_integral: (luminosity_densities, intensities_nu_p).
        tau_sobolev / electron_densities must be the ones resident in the engine when one was passed in."""
        eng = self._ensure_engine(tau_sobolev, electron_densities)
        return eng.formal_integral(inner_temperature, frequencies, att_S_ul, mean_intensity_red_lu, mean_intensity_blue_lu,
                                   n_impact_parameters, want_intensities=True)

    def close(self):
        if self._owns and self._engine is not None:
            self._engine.close()
            self._engine = None
