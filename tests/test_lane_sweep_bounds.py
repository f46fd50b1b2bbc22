"""CPU checks of the two algebraic shortcuts the lane-sweep wave kernel relies on (tardis_amd/csrc/propagate_wave.hpp,
tardis_amd/csrc/mc_device.hpp), in IEEE double arithmetic with the reference's operation order
(trace_packet, modes/homologous_rad_packet_transport.py:100-156; update_line_estimators, estimators/radfield_estimator_calcs.py):

* whenever the four cheap bounds of a line hold, the reference's own tests for that line come out negative (no boundary /
  electron-scattering / line stop, no "nu difference" error) -- including inputs placed within a few ulps of every threshold;
* the line-estimator term energy * (1 - (d_line + mu r) / (t c)) equals energy * nu_line / nu to rounding.
"""
import numpy as np

C_LIGHT = 29979245800.0
CLOSE_LINE_THRESHOLD = 1e-14


def reference_line_outcome(nu_line, tau_line, tau_prev, nu, comov_nu, chi, tau_event, d_boundary, t_exp):
    """0: the trace goes on; 1/2/3: boundary / electron / line stop; 4: MonteCarloException (one non-last line)."""
    tau_incl = tau_prev + tau_line
    d_cont = (tau_event - tau_prev) / chi
    nu_diff = comov_nu - nu_line
    q = nu_diff / nu
    close = np.abs(q) < CLOSE_LINE_THRESHOLD
    err = ~close & ~(nu_diff >= 0)
    d_far = q * C_LIGHT * t_exp
    d_trace = np.where(close, 0.0, d_far)
    tau_combined = tau_incl + chi * d_trace
    dmin = d_trace.copy()
    dmin = np.where(d_boundary < dmin, d_boundary, dmin)
    dmin = np.where(d_cont < dmin, d_cont, dmin)
    stop_b = ~err & (d_trace != 0) & (dmin == d_boundary)
    stop_e = ~err & (d_trace != 0) & ~stop_b & (dmin == d_cont)
    stop_l = ~err & ~stop_b & ~stop_e & (tau_combined > tau_event)
    return np.where(stop_b, 1, np.where(stop_e, 2, np.where(stop_l, 3, np.where(err, 4, 0))))


def cheap_bounds_hold(nu_line, tau_line, tau_prev, nu, comov_nu, chi, tau_event, d_boundary, t_exp):
    tc = t_exp * C_LIGHT
    rcp_tc = 1.0 / tc
    kp = ((chi * tc) / nu) * (1.0 + 2.0 ** -40)
    xb = ((d_boundary * nu) * rcp_tc) * (1.0 - 2.0 ** -40)
    X = comov_nu - nu_line
    x = kp * X
    D = tau_event - tau_prev
    tau_n = tau_prev + tau_line
    total = tau_n + x
    return (X >= 0.0) & (X < xb) & (x < D) & (total <= tau_event)


def lean_bounds_hold(nu_line, tau_line, tau_prev, nu, comov_nu, chi, tau_event, d_boundary, t_exp, first_entry=True):
    """The no-stop proof of the interleaved-table instantiations (propagate_wave_kernel<..., NT != 0>, round 6): X < X_b and
    RN(RN(tau_prev + tau_line) + x) < tau_event -- and X >= 0 only for the first entry of a run (the list is sorted: X grows along it).
    Needs tau_line >= 0 (the host builds the table only then)."""
    tc = t_exp * C_LIGHT
    rcp_tc = 1.0 / tc
    kp = ((chi * tc) / nu) * (1.0 + 2.0 ** -40)
    xb = ((d_boundary * nu) * rcp_tc) * (1.0 - 2.0 ** -40)
    X = comov_nu - nu_line
    x = kp * X
    total = (tau_prev + tau_line) + x
    ok = (X < xb) & (total < tau_event)
    return ok & (X >= 0.0) if first_entry else ok


def _near(rng, n):
    """factors 1 + delta with |delta| log-uniform between 1e-17 and 1e-6, both signs, and exact 1"""
    d = 10.0 ** rng.uniform(-17, -6, n) * rng.choice([-1.0, 1.0], n)
    d[rng.random(n) < 0.05] = 0.0
    return 1.0 + d


def test_cheap_bounds_imply_the_reference_goes_on():
    rng = np.random.default_rng(2024)
    n = 400_000
    total_ok = total_lean = 0
    for kind in range(5):
        t_exp = 10.0 ** rng.uniform(5.5, 7.0, n)
        nu = 10.0 ** rng.uniform(14.0, 16.5, n)
        dop = 1.0 - rng.uniform(-0.05, 0.05, n)
        comov = nu * dop
        chi = 10.0 ** rng.uniform(-17, -12, n)
        d_boundary = 10.0 ** rng.uniform(12.0, 15.5, n)
        tau_event = -np.log(rng.random(n))
        tau_prev = tau_event * rng.uniform(0.0, 1.0, n) * (rng.random(n) < 0.7)
        tau_line = 10.0 ** rng.uniform(-8, 1, n) * (rng.random(n) < 0.9)
        tc = t_exp * C_LIGHT
        if kind == 0:    # generic lines redward of the packet
            X = d_boundary * nu / tc * rng.uniform(0.0, 1.5, n)
        elif kind == 1:  # resonance within ulps of the shell boundary
            X = d_boundary * nu / tc * _near(rng, n)
        elif kind == 2:  # resonance within ulps of the electron-scattering distance
            X = (tau_event - tau_prev) / chi * nu / tc * _near(rng, n)
        elif kind == 3:  # tau_combined within ulps of tau_event
            room = tau_event - (tau_prev + tau_line)
            X = np.where(room > 0, room / chi * nu / tc * _near(rng, n), d_boundary * nu / tc * 0.3)
        else:            # lines (almost) at the packet's comoving frequency, on either side
            X = comov * 10.0 ** rng.uniform(-18, -12, n) * rng.choice([-1.0, 1.0], n)
        nu_line = comov - X
        args = (nu_line, tau_line, tau_prev, nu, comov, chi, tau_event, d_boundary, t_exp)
        ok = cheap_bounds_hold(*args)
        outcome = reference_line_outcome(*args)
        assert not np.any(ok & (outcome != 0)), (kind, int(np.sum(ok & (outcome != 0))))
        total_ok += int(ok.sum())
        # the lean form (tau_line >= 0 here): as the first entry of a run, and as a later one -- whose X >= 0 follows from the sorted list
        lean = lean_bounds_hold(*args)
        assert not np.any(lean & (outcome != 0)), ("lean", kind, int(np.sum(lean & (outcome != 0))))
        later = lean_bounds_hold(*args, first_entry=False) & (X >= 0.0)
        assert not np.any(later & (outcome != 0)), ("lean, later entry", kind)
        total_lean += int(lean.sum())
        # the frequency slot of the last line of the list holds -inf in the interleaved table: the lean proof fails there whatever else holds
        assert not np.any(lean_bounds_hold(np.full(n, -np.inf), *args[1:]))
    assert total_ok > 200_000    # (the bounds are not vacuous: most generic lines pass them)
    assert total_lean >= total_ok - 2_000  # (nor is the lean form weaker in practice: it only gives up the ties of sum == tau_event)


def test_line_estimator_term_is_energy_times_nu_line_over_nu():
    rng = np.random.default_rng(7)
    n = 1_000_000
    t_exp = 10.0 ** rng.uniform(5.5, 7.0, n)
    tc = t_exp * C_LIGHT
    nu = 10.0 ** rng.uniform(14.0, 16.5, n)
    energy = rng.uniform(0.5, 1.5, n) * 1e-7
    r = 10.0 ** rng.uniform(14.0, 15.5, n)
    mu = rng.uniform(-1.0, 1.0, n)
    r = np.minimum(r, 0.3 * tc)                 # v < 0.3 c
    dop = 1.0 - mu * (r / t_exp) / C_LIGHT       # partial relativity (frame_transformations.py:12-47)
    comov = nu * dop
    nu_line = comov * (1.0 - 10.0 ** rng.uniform(-16, -1.5, n) * (rng.random(n) < 0.98))
    q = (comov - nu_line) / nu
    d = np.where(np.abs(q) < CLOSE_LINE_THRESHOLD, 0.0, q * C_LIGHT * t_exp)
    e_ref = energy * (1.0 - (d + mu * r) / tc)   # update_line_estimators
    jb_ref = e_ref / nu
    inv_nu = 1.0 / nu                            # the kernel's record: c_e = energy / nu, c_jb = c_e / nu
    c_e = energy * inv_nu
    c_jb = c_e * inv_nu
    assert np.max(np.abs(c_e * nu_line / e_ref - 1.0)) < 5e-14
    assert np.max(np.abs(c_jb * nu_line / jb_ref - 1.0)) < 5e-14


def test_straight_line_chunk_stops_where_the_loop_stops():
    """The twelve-line instantiation evaluates a chunk without a branch per line (propagate_wave.hpp, lane sweep, `WPE == 3`): serial sums of
    the chunk first, the four bounds of every line into one bit each, count-trailing-zeros.  Same decision and same carried optical depth as the
    loop that tests line after line -- on chunks with stops at every position, at the end of the line list, and for traces outside mid_range."""
    rng = np.random.default_rng(7)
    CH = 12
    for trial in range(20_000):
        comov = 10.0 ** rng.uniform(14.5, 15.5)
        nl = comov * (1.0 - np.cumsum(10.0 ** rng.uniform(-9, -3, CH)))  # descending line frequencies just red of the packet
        if trial % 7 == 0:
            nl[rng.integers(CH)] = comov * (1.0 + 1e-6)                   # a line blue of the packet: X < 0
        tl = 10.0 ** rng.uniform(-6, 1, CH)
        s_tau0 = float(rng.choice([0.0, 10.0 ** rng.uniform(-3, 1)]))
        tau_event = s_tau0 + 10.0 ** rng.uniform(-3, 2)
        kp = 10.0 ** rng.uniform(-18, -12)
        xb = (comov - nl[rng.integers(CH)]) * float(rng.choice([0.5, 1.0, 1.0 + 1e-12, 50.0]))
        n_fast = int(rng.choice([CH + 5, CH, rng.integers(0, CH), -3]))
        s_fast = bool(rng.random() > 0.05)
        # the loop
        alive, adv, s_tau = s_fast, 0, s_tau0
        for k in range(CH):
            if alive:
                X = comov - nl[k]; x = kp * X; D = tau_event - s_tau; tau_n = s_tau + tl[k]; total = tau_n + x
                if k < n_fast and X >= 0.0 and X < xb and x < D and total <= tau_event:
                    s_tau = tau_n; adv += 1
                else:
                    alive = False
        # straight-line
        t = [s_tau0]
        for k in range(CH):
            t.append(t[k] + tl[k])
        fail = 0
        for k in range(CH - 1, -1, -1):
            X = comov - nl[k]; x = kp * X; D = tau_event - t[k]; total = t[k + 1] + x
            ok = X >= 0.0 and X < xb and x < D and total <= tau_event
            fail = fail + fail + (0 if ok else 1)
        fail |= 1 << (min(max(n_fast, 0), CH) if s_fast else 0)
        adv2 = (fail & -fail).bit_length() - 1
        assert adv2 == adv and t[adv2] == s_tau and (adv2 == CH) == alive, trial
