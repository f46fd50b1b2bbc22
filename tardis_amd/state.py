"""Plain-Python state containers mirroring the attribute names of the reference's jitclass structs.

The HIP engine (and the CPU oracle used by the tests) only needs flat float64/int64 arrays.  These
containers carry them under the *same attribute names* the reference uses, so that objects built by an
installed TARDIS (``OpacityStateNumba``, ``NumbaHomologousRadial1DGeometry``, ``PacketCollection``,
``MonteCarloConfiguration``) and the objects defined here are interchangeable at the drop-in boundary
(duck typing; see ``tardis_amd.transport``).

Reference layouts mirrored:
  * PacketCollection            tardis/transport/montecarlo/packets/packet_collections.py:14-76
  * geometry                    tardis/model/geometry/radial1d_homologous.py:199-226
  * OpacityStateNumba           tardis/opacities/opacity_state_numba.py:14-196
  * MonteCarloConfiguration     tardis/transport/montecarlo/configuration/base.py:11-49
  * EstimatorsBulk / Line       tardis/transport/montecarlo/estimators/estimators_bulk.py:15-56,
                                tardis/transport/montecarlo/estimators/estimators_line.py:15-60
  * VPacketCollection (result)  tardis/transport/montecarlo/packets/packet_collections.py:103-307
"""
from __future__ import annotations

import numpy as np

# CODATA 2010 cgs values (tardis/constants.py:1 -> astropy.constants.astropyconst13;
# tardis/transport/montecarlo/configuration/constants.py:1-8)
C_SPEED_OF_LIGHT = 2.99792458e10
SIGMA_THOMSON = 6.652458734e-25
H_PLANCK = 6.62606957e-27
K_BOLTZMANN = 1.3806488e-16
SIGMA_SB = 5.670373e-5
CLOSE_LINE_THRESHOLD = 1e-14
MISS_DISTANCE = 1e99

# enums (tardis/transport/montecarlo/packets/radiative_packet.py:12-43,
#        tardis/transport/montecarlo/interaction_events.py:220-223)
INTERACTION_NONE, INTERACTION_BOUNDARY, INTERACTION_LINE, INTERACTION_ESCATTERING = -1, 1, 2, 4
STATUS_IN_PROCESS, STATUS_EMITTED, STATUS_REABSORBED = 0, 1, 2
LINE_SCATTER, LINE_DOWNBRANCH, LINE_MACROATOM = 0, 1, 2
LINE_INTERACTION_TYPES = {"scatter": LINE_SCATTER, "downbranch": LINE_DOWNBRANCH, "macroatom": LINE_MACROATOM}


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class PacketCollection:
    """Input packet arrays + in-place output arrays (initial fill -99, as in the reference)."""

    def __init__(self, initial_radii, initial_nus, initial_mus, initial_energies, packet_seeds,
                 radiation_field_luminosity):
        self.initial_radii = _f64(initial_radii)
        self.initial_nus = _f64(initial_nus)
        self.initial_mus = _f64(initial_mus)
        self.initial_energies = _f64(initial_energies)
        self.packet_seeds = _i64(packet_seeds)
        self.radiation_field_luminosity = float(radiation_field_luminosity)
        self.time_of_simulation = 1 / self.radiation_field_luminosity
        self.output_nus = np.ones_like(self.initial_radii) * -99.0
        self.output_energies = np.ones_like(self.initial_radii) * -99.0

    @property
    def number_of_packets(self) -> int:
        return len(self.initial_radii)

    def shard(self, rank: int, world_size: int) -> "PacketCollection":
        """Contiguous packet-index shard [rank*P/W, (rank+1)*P/W).

        The output arrays of the shard are *views* into this collection's outputs, so every rank
        writes a disjoint slice and no collective is needed for per-packet results.
        """
        n = self.number_of_packets
        lo, hi = (rank * n) // world_size, ((rank + 1) * n) // world_size
        sub = PacketCollection.__new__(PacketCollection)
        for name in ("initial_radii", "initial_nus", "initial_mus", "initial_energies", "packet_seeds",
                     "output_nus", "output_energies"):
            setattr(sub, name, getattr(self, name)[lo:hi])
        sub.radiation_field_luminosity = self.radiation_field_luminosity
        sub.time_of_simulation = self.time_of_simulation
        return sub


class HomologousRadial1DGeometry:
    def __init__(self, r_inner, r_outer, v_inner, v_outer, time_explosion):
        self.r_inner = _f64(r_inner)
        self.r_outer = _f64(r_outer)
        self.v_inner = _f64(v_inner)
        self.v_outer = _f64(v_outer)
        self.time_explosion = float(time_explosion)
        self.velocity_gradient = 1.0 / self.time_explosion
        self.volume = (4 / 3) * np.pi * (self.r_outer**3 - self.r_inner**3)

    def get_velocity(self, r, shell_id):
        return r / self.time_explosion


class OpacityState:
    """Classic-mode opacity state; continuum fields exist (empty) for attribute compatibility."""

    def __init__(self, electron_density, t_electrons, line_list_nu, tau_sobolev, transition_probabilities,
                 line2macro_level_upper, macro_block_edge_index, transition_type, destination_level_id,
                 transition_line_id):
        self.electron_density = _f64(electron_density)
        self.t_electrons = _f64(t_electrons)
        self.line_list_nu = _f64(line_list_nu)
        self.tau_sobolev = np.asarray(tau_sobolev, dtype=np.float64)
        self.transition_probabilities = np.asarray(transition_probabilities, dtype=np.float64)
        self.line2macro_level_upper = _i64(line2macro_level_upper)
        self.macro_block_edge_index = _i64(macro_block_edge_index)
        self.transition_type = _i64(transition_type)
        self.destination_level_id = _i64(destination_level_id)
        self.transition_line_id = _i64(transition_line_id)
        # continuum placeholders (opacity_state_numba.py:28-44)
        self.bf_threshold_list_nu = np.zeros(0)
        self.p_fb_deactivation = np.zeros((0, 0))
        self.photo_ion_nu_threshold_mins = np.zeros(0)
        self.photo_ion_nu_threshold_maxs = np.zeros(0)
        self.photo_ion_block_references = np.zeros(0, dtype=np.int64)
        self.chi_bf = np.zeros((0, 0))
        self.x_sect = np.zeros(0)
        self.phot_nus = np.zeros(0)
        self.ff_opacity_factor = np.zeros(0)
        self.emissivities = np.zeros((0, 0))
        self.photo_ion_activation_idx = np.zeros(0, dtype=np.int64)
        self.k_packet_idx = -1

    def __getitem__(self, i: slice) -> "OpacityState":
        """Shell slice (opacity_state_numba.py:157-196)."""
        if not isinstance(i, slice):
            raise TypeError("OpacityState supports only shell slices")
        return OpacityState(
            self.electron_density[i], self.t_electrons[i], self.line_list_nu, self.tau_sobolev[:, i],
            self.transition_probabilities[:, i], self.line2macro_level_upper, self.macro_block_edge_index,
            self.transition_type, self.destination_level_id, self.transition_line_id)


class MonteCarloConfiguration:
    """Same field names and defaults as configuration/base.py:28-49."""

    def __init__(self):
        self.ENABLE_FULL_RELATIVITY = False
        self.TEMPORARY_V_PACKET_BINS = 0
        self.NUMBER_OF_VPACKETS = 0
        self.MONTECARLO_SEED = 0
        self.LINE_INTERACTION_TYPE = 0
        self.PACKET_SEEDS = np.empty(1, dtype=np.int64)
        self.DISABLE_ELECTRON_SCATTERING = False
        self.DISABLE_LINE_SCATTERING = False
        self.SURVIVAL_PROBABILITY = 0.0
        self.VPACKET_TAU_RUSSIAN = 10.0
        self.INITIAL_TRACKING_ARRAY_LENGTH = 0
        self.LEGACY_MODE_ENABLED = False
        self.VPACKET_SPAWN_START_FREQUENCY = 0
        self.VPACKET_SPAWN_END_FREQUENCY = 1e200
        self.ENABLE_VPACKET_TRACKING = False


class EstimatorsBulk:
    def __init__(self, mean_intensity_total, mean_frequency):
        self.mean_intensity_total = mean_intensity_total
        self.mean_frequency = mean_frequency

    def increment(self, other: "EstimatorsBulk") -> None:
        self.mean_intensity_total += other.mean_intensity_total
        self.mean_frequency += other.mean_frequency


class EstimatorsLine:
    def __init__(self, mean_intensity_blueward, energy_deposition_line_rate):
        self.mean_intensity_blueward = mean_intensity_blueward
        self.energy_deposition_line_rate = energy_deposition_line_rate

    def increment(self, other: "EstimatorsLine") -> None:
        self.mean_intensity_blueward += other.mean_intensity_blueward
        self.energy_deposition_line_rate += other.energy_deposition_line_rate


class VPacketCollection:
    """Consolidated virtual-packet log (only filled when ENABLE_VPACKET_TRACKING)."""

    def __init__(self, source_rpacket_index, spectrum_frequency_grid, v_packet_spawn_start_frequency,
                 v_packet_spawn_end_frequency, number_of_vpackets, length):
        n = max(int(length), 0)
        self.source_rpacket_index = source_rpacket_index
        self.spectrum_frequency_grid = spectrum_frequency_grid
        self.v_packet_spawn_start_frequency = v_packet_spawn_start_frequency
        self.v_packet_spawn_end_frequency = v_packet_spawn_end_frequency
        self.number_of_vpackets = number_of_vpackets
        self.nus = np.empty(n)
        self.energies = np.empty(n)
        self.initial_mus = np.empty(n)
        self.initial_rs = np.empty(n)
        # placeholders the reference fills with -99 (virtual_packet.py:375-386)
        self.last_interaction_in_nu = np.full(n, -99.0)
        self.last_interaction_in_r = np.full(n, -99.0)
        self.last_interaction_type = np.full(n, -99, dtype=np.int64)
        self.last_interaction_in_id = np.full(n, -99, dtype=np.int64)
        self.last_interaction_out_id = np.full(n, -99, dtype=np.int64)
        self.last_interaction_shell_id = np.full(n, -99, dtype=np.int64)
        self.idx = n
        self.length = n


class LastInteractionTrackers:
    """SoA replacement for the reference's list of P ``TrackerLastInteraction`` jitclass objects
    (packets/trackers/tracker_last_interaction.py:8-254).  Field names match the per-object
    attributes; ``to_dataframe`` reproduces ``trackers_last_interaction_to_df``
    (packets/trackers/tracker_last_interaction_util.py:33-134).
    """

    F64_FIELDS = ("radius", "nu", "mu", "energy", "before_nu", "before_mu", "before_energy", "after_nu",
                  "after_mu", "after_energy")
    I64_FIELDS = ("shell_id", "interaction_type", "interaction_line_absorb_id", "interaction_line_emit_id",
                  "interactions_count")

    def __init__(self, n_packets: int):
        for f in self.F64_FIELDS:
            setattr(self, f, np.full(n_packets, np.nan))
        for f in self.I64_FIELDS:
            setattr(self, f, np.full(n_packets, -1, dtype=np.int64))
        self.interactions_count[:] = 0

    def __len__(self):
        return len(self.radius)
