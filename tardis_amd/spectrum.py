"""Real-packet spectrum (the parity metric of BASELINE.json).

Restates SpectrumSolver.montecarlo_emitted_luminosity (tardis/spectrum/base.py:140-159) on plain arrays:
``np.histogram(emitted_packet_nu, weights=emitted_packet_luminosity, bins=spectrum_frequency_grid)`` with
emitted = output_energies >= 0 and luminosity = energy / time_of_simulation
(montecarlo_transport_state.py:130-160).
"""
from __future__ import annotations

import numpy as np


def emitted_luminosity_histogram(output_nus, output_energies, time_of_simulation, spectrum_frequency_grid):
    mask = output_energies >= 0
    lum = output_energies[mask] / time_of_simulation
    hist, _ = np.histogram(output_nus[mask], weights=lum, bins=spectrum_frequency_grid)
    return hist


def relative_l2(a, b) -> float:
    """||a - b||_2 / ||b||_2 (0 when both vanish)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = float(np.sqrt(np.sum(b * b)))
    num = float(np.sqrt(np.sum((a - b) ** 2)))
    return num / den if den > 0 else (0.0 if num == 0 else float("inf"))


def reabsorbed_luminosity_histogram(output_nus, output_energies, time_of_simulation, spectrum_frequency_grid):
    """SpectrumSolver.montecarlo_reabsorbed_luminosity (tardis/spectrum/base.py:140-148)."""
    mask = output_energies < 0
    lum = -(output_energies[mask] / time_of_simulation)
    hist, _ = np.histogram(output_nus[mask], weights=lum, bins=spectrum_frequency_grid)
    return hist


def calculate_filtered_luminosity(packet_nu, packet_luminosity, luminosity_nu_start=0.0, luminosity_nu_end=np.inf):
    """tardis/spectrum/luminosity.py:5-30 on plain arrays."""
    f = (packet_nu > luminosity_nu_start) & (packet_nu < luminosity_nu_end)
    return packet_luminosity[f].sum()
