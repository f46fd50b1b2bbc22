"""TEST INFRASTRUCTURE (oracle): numpy restatement of the radiation-field update.

Follows MCRadiationFieldPropertiesSolver.solve (tardis/transport/montecarlo/estimators/mc_rad_field_solver.py:37-144),
DilutePlanckianRadiationField.calculate_mean_intensity (tardis/plasma/radiation_field/planck_rad_field.py:58-73) and
intensity_black_body (tardis/util/base.py:279-302) on plain arrays with the cgs CODATA-2010 constants of
tardis/constants.py:1.  PINNED: tests/golden/radfield_*.npz hold what the reference's own solver returned in the dev container
(tools/make_golden_next_rows.py imports it unmodified through tools/ref_shim.py); tests/test_next_rows_golden.py holds this
restatement to them at 1e-13 (the reference evaluates exp through numexpr, the fixtures through numpy), tests/test_radfield.py
adds closed-form checks.
"""
import numpy as np

H, K_B, SIGMA_SB, C = 6.62606957e-27, 1.3806488e-16, 5.670373e-5, 2.99792458e10
ZETA5 = 1.0369277551433699
T_RADIATIVE_ESTIMATOR_CONSTANT = (np.pi**4 / (15 * 24 * ZETA5)) * (H / K_B)  # mc_rad_field_solver.py:26-28


def intensity_black_body(nu, temperature):
    beta_rad = 1 / (K_B * temperature)
    coefficient = 2 * H / C**2
    return coefficient * nu**3 / (np.exp(H * nu * beta_rad) - 1)


def solve(mean_intensity_total, mean_frequency, mean_intensity_blueward, time_explosion, time_of_simulation, volume,
          line_list_nu, w_epsilon=1e-10, detailed_optical_window=False):
    t_rad = T_RADIATIVE_ESTIMATOR_CONSTANT * mean_frequency / mean_intensity_total
    w = mean_intensity_total / (4 * SIGMA_SB * t_rad**4 * time_of_simulation * volume)
    norm = C * time_explosion / (4 * np.pi * time_of_simulation * volume)
    j_blues = mean_intensity_blueward * norm
    planck = w * intensity_black_body(line_list_nu[np.newaxis].T, t_rad)
    zero = j_blues == 0.0
    if detailed_optical_window:
        wav = C / line_list_nu * 1e8
        optical = np.logical_and(wav > 2500.0, wav < 10000.0)
        j_blues[~optical] = planck[~optical]
    j_blues[zero] = w_epsilon * planck[zero]
    return t_rad, w, j_blues
