"""Host-side logic of the drop-in boundary that needs no GPU: marshalling of the reference's own placeholder shapes and the
tracker hand-back."""
import types

import numpy as np
import pytest

from tardis_amd import _abi, state as st, synthetic, transport


def test_scatter_mode_placeholders_of_the_reference_are_accepted():
    """OpacityState.to_numba passes np.zeros((1, 1)) and size-1 macro tables for line_interaction_type 'scatter'
    (tardis/opacities/opacity_state.py:199-209), whatever the number of shells."""
    prob = synthetic.make_problem(seed=2, n_packets=10, n_shells=5, n_lines=100, line_interaction_type="scatter")
    op = prob.opacity_state
    ref_like = types.SimpleNamespace(
        electron_density=op.electron_density, line_list_nu=op.line_list_nu, tau_sobolev=op.tau_sobolev,
        transition_probabilities=np.zeros((1, 1)), line2macro_level_upper=np.zeros(1, np.int64),
        macro_block_edge_index=np.zeros(1, np.int64), transition_type=np.zeros(1, np.int64),
        destination_level_id=np.zeros(1, np.int64), transition_line_id=np.zeros(1, np.int64))
    m = _abi.marshal_opacity(ref_like)
    assert m.struct.n_transitions == 1 and m.struct.n_shells == 5 and m.struct.n_macro_block_edges == 1
    # a genuinely mis-shaped table is still rejected
    ref_like.transition_probabilities = np.zeros((3, 2))
    with pytest.raises(ValueError):
        _abi.marshal_opacity(ref_like)


class _Tracker:  # attribute names of TrackerLastInteraction (packets/trackers/tracker_last_interaction.py:8-254)
    def __init__(self):
        for f in st.LastInteractionTrackers.F64_FIELDS:
            setattr(self, f, -1.0)
        for f in st.LastInteractionTrackers.I64_FIELDS:
            setattr(self, f, -1)


def test_fill_trackers_writes_a_list_of_per_packet_objects():
    n = 7
    soa = st.LastInteractionTrackers(n)
    rng = np.random.default_rng(0)
    for f in soa.F64_FIELDS:
        getattr(soa, f)[:] = rng.random(n)
    for f in soa.I64_FIELDS:
        getattr(soa, f)[:] = rng.integers(0, 100, n)
    lst = [_Tracker() for _ in range(n)]
    transport._fill_trackers(lst, soa)
    for i, t in enumerate(lst):
        for f in soa.F64_FIELDS + soa.I64_FIELDS:
            assert getattr(t, f) == getattr(soa, f)[i], f
        assert isinstance(t.shell_id, int) and isinstance(t.radius, float)
    with pytest.raises(ValueError):
        transport._fill_trackers(lst[:3], soa)


def test_full_rpacket_trackers_are_rejected_loudly():
    class TrackerFull(_Tracker):
        pass

    soa = st.LastInteractionTrackers(2)
    with pytest.raises(NotImplementedError):
        transport._fill_trackers([TrackerFull(), TrackerFull()], soa)
    arr = _Tracker()
    arr.radius = np.zeros(4)  # array-valued fields: one row per event
    with pytest.raises(NotImplementedError):
        transport._fill_trackers([arr, _Tracker()], soa)
