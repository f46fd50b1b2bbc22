/* tardis_mc.h -- C ABI of the MI355X-native Monte Carlo packet-propagation engine (libtardis_mc_hip.so).
 *
 * This is the drop-in boundary for ONE path of tardis-sn/tardis: the Monte Carlo main loop
 *     montecarlo_transport_with_vpackets(packet_collection, geometry_state_numba, time_explosion,
 *         opacity_state_numba, montecarlo_configuration, spectrum_frequency_grid, trackers,
 *         number_of_vpackets, show_progress_bars, packet_propagation_function)
 *       -> (v_packets_energy_hist, vpacket_tracker, estimators_bulk, estimators_line)
 *     tardis/transport/montecarlo/modes/montecarlo_transport.py:238-373
 * as called from MCTransportSolverClassic.run_classic
 *     tardis/transport/montecarlo/modes/classic/solver.py:223-234.
 * The reference has no FFI for this path (it is a Numba @njit function), so these entry points are what a
 * ctypes binding inside run_classic would bind; see INTEGRATION.md for the stub.
 *
 * Conventions
 *   - plain pointers and sizes only; all reals are float64, all integers int64 (the reference's dtypes);
 *   - every array is caller-owned, C-contiguous, and is neither retained nor freed by the library after
 *     the call that takes it returns (inputs are copied to HBM; outputs are written in place);
 *   - 2-D arrays use the REFERENCE's layout: tau_sobolev[L,S], transition_probabilities[T,S],
 *     j_blue[L,S], Edotlu[L,S], row-major (line index slow, shell index fast).  The engine keeps its own
 *     shell-major copies in HBM;
 *   - functions return 0 on success or a negative TARDIS_MC_ERR_* code; tardis_mc_last_error() gives text;
 *   - a context is bound to one HIP device and one stream; entry points are not re-entrant per context.
 */
#ifndef TARDIS_MC_H
#define TARDIS_MC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TARDIS_MC_ABI_VERSION 2  /* 2 (round 6): + tardis_mc_comm_check, tardis_mc_stream_results, tardis_mc_streamed_packets, tardis_mc_last_compactions, microbench 15 */

enum {
    TARDIS_MC_OK = 0,
    TARDIS_MC_ERR_INVALID_ARGUMENT = -1,
    TARDIS_MC_ERR_HIP = -2,          /* HIP runtime failure (no device, OOM, launch failure) */
    TARDIS_MC_ERR_MONTECARLO = -3,   /* reference: MonteCarloException("nu difference is less than 0.0"),
                                        transport/geometry/calculate_distances.py:105-106 */
    TARDIS_MC_ERR_MACRO_ATOM = -4,   /* reference: MacroAtomError, transport/montecarlo/macro_atom.py:94-99 */
    TARDIS_MC_ERR_UNSUPPORTED = -5,  /* transition type outside classic mode (continuum processes) */
    TARDIS_MC_ERR_COMM = -6,         /* RCCL failure */
    TARDIS_MC_ERR_STATE = -7         /* call order violated (e.g. propagate before set_opacity) */
};

/* LineInteractionType, transport/montecarlo/interaction_events.py:220-223 */
enum { TARDIS_MC_LINE_SCATTER = 0, TARDIS_MC_LINE_DOWNBRANCH = 1, TARDIS_MC_LINE_MACROATOM = 2 };

/* MonteCarloConfiguration fields the classic path reads, transport/montecarlo/configuration/base.py:11-49,
 * plus SIGMA_THOMSON (configuration/constants.py:3; 1e-200 when electron scattering is disabled,
 * modes/classic/solver.py:291-300). */
typedef struct TardisMcConfig {
    int32_t enable_full_relativity;        /* ENABLE_FULL_RELATIVITY */
    int32_t line_interaction_type;         /* LINE_INTERACTION_TYPE */
    int32_t disable_line_scattering;       /* DISABLE_LINE_SCATTERING */
    int32_t enable_vpacket_tracking;       /* ENABLE_VPACKET_TRACKING */
    int64_t number_of_vpackets;            /* NUMBER_OF_VPACKETS */
    double survival_probability;           /* SURVIVAL_PROBABILITY (default 0.0) */
    double vpacket_tau_russian;            /* VPACKET_TAU_RUSSIAN (default 10.0) */
    double vpacket_spawn_start_frequency;  /* VPACKET_SPAWN_START_FREQUENCY */
    double vpacket_spawn_end_frequency;    /* VPACKET_SPAWN_END_FREQUENCY */
    double sigma_thomson;                  /* SIGMA_THOMSON [cm^2] */
    int64_t n_spectrum_grid;               /* len(spectrum_frequency_grid) = bins + 1 */
    const double *spectrum_frequency_grid; /* uniform ascending edges [Hz] */
} TardisMcConfig;

/* PacketCollection inputs, transport/montecarlo/packets/packet_collections.py:14-76 */
typedef struct TardisMcPackets {
    int64_t n_packets;
    const double *initial_radii;
    const double *initial_nus;
    const double *initial_mus;
    const double *initial_energies;
    const int64_t *packet_seeds;           /* MT19937 init_genrand seed per packet, in [0, 2^32-2] */
} TardisMcPackets;

/* NumbaHomologousRadial1DGeometry, model/geometry/radial1d_homologous.py:199-226 */
typedef struct TardisMcGeometry {
    int64_t n_shells;
    const double *r_inner;
    const double *r_outer;
    double time_explosion;
} TardisMcGeometry;

/* OpacityStateNumba fields used by classic mode, opacities/opacity_state_numba.py:14-196 */
typedef struct TardisMcOpacity {
    int64_t n_lines;                       /* L */
    int64_t n_shells;                      /* S */
    int64_t n_transitions;                 /* T (1 for "scatter" dummies) */
    int64_t n_macro_block_edges;           /* len(macro_block_edge_index) = levels + 1 */
    const double *electron_density;        /* [S] */
    const double *line_list_nu;            /* [L] descending */
    const double *tau_sobolev;             /* [L,S] */
    const double *transition_probabilities;/* [T,S] */
    const int64_t *line2macro_level_upper; /* [L] */
    const int64_t *macro_block_edge_index; /* [levels+1] */
    const int64_t *transition_type;        /* [T] */
    const int64_t *destination_level_id;   /* [T] */
    const int64_t *transition_line_id;     /* [T] */
} TardisMcOpacity;

/* Work counters accumulated by the kernels (bytes = 48 V + 56 E + 8 M + 16 Vv + 56 P) */
enum {
    TARDIS_MC_CNT_LINE_VISITS = 0,         /* V: iterations of the trace_packet line loop */
    TARDIS_MC_CNT_EVENTS = 1,              /* E: trace_packet calls */
    TARDIS_MC_CNT_MACRO_TRANSITIONS = 2,   /* M: transition probabilities examined */
    TARDIS_MC_CNT_VPACKET_LINE_VISITS = 3, /* Vv */
    TARDIS_MC_CNT_VPACKETS = 4,            /* v-packets traced */
    TARDIS_MC_CNT_RNG_DRAWS = 5,           /* MT19937 doubles consumed */
    TARDIS_MC_CNT_PACKETS = 6,             /* P */
    TARDIS_MC_CNT_RESERVED = 7,
    TARDIS_MC_N_COUNTERS = 8
};

/* Outputs.  Any pointer may be NULL to skip that output (sizes in comments). */
typedef struct TardisMcResult {
    double *output_nus;                    /* [P]  packet_collection.output_nus */
    double *output_energies;               /* [P]  +e EMITTED / -e REABSORBED (montecarlo_transport.py:70-90) */
    double *j_estimator;                   /* [S]  EstimatorsBulk.mean_intensity_total */
    double *nu_bar_estimator;              /* [S]  EstimatorsBulk.mean_frequency */
    double *j_blue_estimator;              /* [L,S] EstimatorsLine.mean_intensity_blueward */
    double *edotlu_estimator;              /* [L,S] EstimatorsLine.energy_deposition_line_rate */
    double *v_packets_energy_hist;         /* [n_spectrum_grid] */
    /* TrackerLastInteraction as SoA, packets/trackers/tracker_last_interaction.py:8-254 */
    double *li_radius, *li_nu, *li_energy;                 /* [P] each */
    double *li_before_nu, *li_before_mu, *li_before_energy;
    double *li_after_nu, *li_after_mu, *li_after_energy;
    int64_t *li_shell_id, *li_interaction_type;
    int64_t *li_line_absorb_id, *li_line_emit_id, *li_interactions_count;
    /* consolidated v-packet log (only when enable_vpacket_tracking), packet order */
    int64_t vpacket_log_capacity;          /* in: entries available in the four arrays below */
    int64_t vpacket_log_count;             /* out: entries produced (may exceed capacity -> truncated) */
    double *vpacket_nus, *vpacket_energies, *vpacket_initial_mus, *vpacket_initial_rs;
    int64_t counters[TARDIS_MC_N_COUNTERS];
    int64_t first_error_packet;            /* out: lowest packet index that failed, or -1 */
    int32_t error_code;                    /* out: 0 or TARDIS_MC_ERR_MONTECARLO / _MACRO_ATOM / _UNSUPPORTED */
    int32_t reserved;
} TardisMcResult;

typedef struct TardisMcContext TardisMcContext;

/* ---- library / device ---------------------------------------------------------------------------- */
int tardis_mc_abi_version(void);
int tardis_mc_device_count(void);
/* Create a context on HIP device `device_id` (one process per GPU: pass LOCAL_RANK). */
int tardis_mc_create(int device_id, TardisMcContext **out_ctx);
void tardis_mc_destroy(TardisMcContext *ctx);
const char *tardis_mc_last_error(const TardisMcContext *ctx);   /* ctx may be NULL: last create() error */

/* Tunables.  name: "track_last_interaction" (0/1, default 1), "vpacket_log_capacity" (entries),
 * "variant" (kernel variant: -1 automatic, 0 lane-per-packet, 1 group-per-packet, 2 wave-owner with group sweeps, 3 wave-owner
 * with lane sweeps, 4 wave-owner with the volley queue: v-packets traced by a kernel of their own between its launches --
 * never the automatic choice; falls back to 2/3 without v-packets and to 1 with a survival probability > 0),
 * "vq_min_items" (variant 4: switch the queue off for the rest of a call once a launch requests fewer v-packets; -1 automatic,
 * 0 never), "vq_min_active", "vq_oversubscribe", "vq_tracer_waves_per_simd" (variant 4 launch shape),
 * "lane_sweep_min_active" / "lane_sweep_max_steps" (when variant 3 leaves its sweep phase; lane_sweep_min_active < 0: automatic --
 * 8, or 12 where most macro-atom blocks are entered through hot sectors),
 * "walk_min_active" (macroatom walks are carried over to the next pass once this few lanes still walk; -1 never, < -1 automatic:
 * 8 / 12 likewise),
 * "log_tail_split" (1, the default: a call whose line-visit log fits one epoch is split where the drain of its longest-lived packets
 * begins -- "log_tail_packets" (8) packets' worth of traces per lane before the estimated end -- so that the estimator passes over
 * the bulk run beside the drain; 0 off), "log_chunk_records" (records per chunk of the log's pool, <= 4096 by default),
 * "est_pipeline" (how an epoch's line-visit log becomes j_blue / Edotlu: 1, the default: the records are partitioned by shell and then
 * by (shell, 2048-line tile) in two LDS-staged passes and added up in order, csrc/estimator_partition.hpp; 0: an index of the records is
 * counting-sorted and the records are fetched through it, csrc/estimator_log.hpp -- also what runs for line lists of more than 2e6 lines
 * or more than 1024 shells), "est_accumulate" (est_pipeline 0 only: 1 block sums, 0 one LDS add per line visit),
 * "log_capacity" (line-visit records per epoch and buffer set of the wave-owner kernel; a call that logs more runs as several
 * launches over one packet supply), "log_sets" (1: the estimator passes of an epoch run before the next
 * epoch instead of beside it), "chunk_packets" (packets per launch of the group kernel), "waves_per_simd", "group_size",
 * "blocks_per_cu", "estimator_copies" (1..8 private j_blue/Edotlu copies),
 * "vpacket_screening" (v-packets whose Russian roulette is decided from prefix sums of tau instead of a line-by-line trace,
 * csrc/tau_prefix.hpp: -1 automatic -- on where a shell crossing passes many lines --, 0 off, 1 on),
 * "walk_sector_packing" (1, the default: blocks of the compact walk tables do not straddle 64-byte sectors; takes effect at the
 * next tardis_mc_set_opacity),
 * "walk_hot" (hot sectors of the macro-atom walk, csrc/walk_tables.hpp: one 64-byte record per (shell, block) with the block's
 * six widest probability intervals decides most jumps out of skewed blocks in one request; -1, the default: for the blocks
 * whose six intervals cover at least "walk_hot_min_mass" (per mille, default 800; blocks of more than 32 transitions:
 * "walk_hot_min_mass_long", default 400) of the block on average over the shells; 0 none; 1 every block; all three take effect
 * at the next tardis_mc_set_opacity), "drain_split" (1: the drain of a call runs as a launch of its own beside the estimator passes of
 * what was logged before it; measured, off by default),
 * "vp_carry_min_active" (pooled v-packet volleys of the wave-owner kernel: a volley phase ends once every v-packet of the round has been handed
 * to a lane and at most this many lanes still trace; those keep their v-packets for the next pass; default 16, 0: a phase runs to its end),
 * "bucket_lines_permille" (resolution of the frequency-bucket index of the line list, lines per bucket x 1000; default 750; takes effect at the
 * next tardis_mc_set_opacity), "vpk_wide_registers" (1, the default: v-packet calls on grids whose per-shell LDS arrays allow at most eight
 * waves per CU run the instantiation compiled for two waves per SIMD -- 239 VGPRs, no spills; 0 never; 2 always),
 * "vpk_wave_min_packets" (v-packet calls on fine grids take the wave-owner kernel from this many packets on, the group kernel below; default
 * 100000), "ls_waves_per_simd" (which instantiation of the lane-sweep kernel: 4 = 128 VGPRs, sixteen waves per CU, eight lines per step; 3 = 166 VGPRs,
 * twelve waves per CU, twelve lines per step -- faster where a call is mostly the drain of its longest packets; 0, the default: the engine times both on
 * the first calls of a (packet count, tables) key and keeps the faster; per-packet results are bit-identical either way), "pass_cus" (CUs per XCD set aside for the line-estimator passes through CU-masked streams; default 0 = off: measured, never pays),
 * "debug_flags" (profiling experiments / cross-checks only: 1 skips the j_blue/Edotlu updates, 2 the J/nu_bar updates, 128
 * walks the macro atom by a per-lane search in the fp64 running sums, 8192 by the cooperative group scan; tests: 16384 counts
 * the jumps out of blocks longer than one window of the compact walk tables into counters[7], 32768 the jumps decided by the
 * fp64 running sums because 16-bit entries tie, 65536 / 131072 the jumps a hot sector decided / handed on to the block's own
 * tables, 67108864 the v-packets decided by the screening into counters[7] >> 40;
 * diagnostics of a call's drain: 2097152 / 4194304 / 8388608 sum, per wave and from the pass in which its packet supply ran out,
 * the 10-ns ticks to its end / its passes / its live lanes over those passes into counters[7]; 16777216 restores the fixed
 * cut-offs of the sweep and walk phases of rounds 1-2; 33554432 switches the v-packet screening off; 134217728 / 268435456 sum the lanes that
 * traced / the steps of the pooled volleys' worker loop into counters[7]; 2048 starts a new packet's roulette predictor at zero as in rounds
 * 1-4.  The flags that read a profiling counter or switch an ablation -- 1, 2, 4, 16, 32, 16384 ... 131072, 524288, 2097152 ... 16777216,
 * 134217728, 268435456 -- make the engine launch the cross-check instantiation of the kernel, which alone carries them). */
int tardis_mc_set_option(TardisMcContext *ctx, const char *name, long long value);

/* ---- staged API: inputs resident in HBM, kernels timed separately -------------------------------- */
int tardis_mc_set_geometry(TardisMcContext *ctx, const TardisMcGeometry *geometry);
/* Uploads and re-lays the opacity tables shell-major; once per MC iteration (the plasma changes them). */
int tardis_mc_set_opacity(TardisMcContext *ctx, const TardisMcOpacity *opacity);
int tardis_mc_set_config(TardisMcContext *ctx, const TardisMcConfig *config);
int tardis_mc_set_packets(TardisMcContext *ctx, const TardisMcPackets *packets);
/* Zero J, nu_bar, j_blue, Edotlu, v-hist and the counters (start of an iteration). */
int tardis_mc_reset_estimators(TardisMcContext *ctx);
/* Launch the propagation kernels for the resident packets on the context stream.
 * Estimators ACCUMULATE across calls until tardis_mc_reset_estimators (packet chunks of one iteration).
 * Blocking behaviour: the lane and group kernels (variants 0, 1) are queued and the call returns at once.  The wave-owner
 * kernel (variants 2-4, the automatic choice for sorted line lists) runs a call as a sequence of launches ("epochs") over one
 * packet supply; after every launch the HOST waits for one word (did any wave suspend on a full line-visit log region?) to
 * decide whether another launch follows, so the call returns when the LAST propagation launch has been queued and all earlier
 * ones have finished -- for a one-epoch call that is after its only launch.  The estimator passes of the last epoch, the
 * tracker unpacking and the result copies are still asynchronous: call tardis_mc_synchronize before reading results.  One host
 * thread driving several contexts therefore serialises their propagations; use one thread (or process) per context.
 * Returns TARDIS_MC_ERR_STATE if the launch bound of a call is exhausted with waves still suspended (results incomplete). */
int tardis_mc_propagate(TardisMcContext *ctx);
int tardis_mc_synchronize(TardisMcContext *ctx);
/* Device time of the kernels launched by the last tardis_mc_propagate (HIP events on the ctx stream). */
int tardis_mc_last_propagate_ms(TardisMcContext *ctx, double *out_ms);
/* The same, split per kernel: total time of the MT19937 seeding launches and of the propagation launches of the last
 * tardis_mc_propagate, and how many propagation launches there were (packet chunks). */
int tardis_mc_last_kernel_times(TardisMcContext *ctx, double *out_seed_ms, double *out_propagate_ms, int *out_launches);
/* summed duration of the line-estimator passes (record binning + accumulation) of the last propagate call; 0 when the
 * kernel variant in use updates the estimators with atomics */
int tardis_mc_last_estimator_ms(TardisMcContext *ctx, double *out_ms);
/* raw work counters of the last kernel launches (TARDIS_MC_CNT_*; after tardis_mc_formal_integral counters[0] is the number
 * of resonances crossed by all rays) */
int tardis_mc_last_counters(TardisMcContext *ctx, int64_t out_counters[TARDIS_MC_N_COUNTERS]);
/* Which propagation kernel the last tardis_mc_propagate ran (the "variant" option, or the automatic choice): 0 lane-per-packet,
 * 1 group-per-packet, 2 wave-owner with group sweeps, 3 wave-owner with lane sweeps, 4 wave-owner with the volley queue;
 * -1 before the first call. */
int tardis_mc_last_variant(TardisMcContext *ctx);
/* How often the last tardis_mc_propagate packed the live lanes of its drain into fewer waves (option "drain_compact" = T: once the packet supply has run out a wave
 * suspends when T or fewer of its lanes still hold a packet; the live lanes of all waves are packed into full waves and the rest of the call runs as a launch of
 * fewer waves, beside the line-estimator passes of the launch before; 0 = off.  Per-packet results do not depend on it). */
int tardis_mc_last_compactions(TardisMcContext *ctx);
/* Progress of the propagate call that is running (or of the last one): packets handed to the propagation kernel so far and the call's
 * packet count -- what the reference's packet progress bar shows (update_packets_pbar, modes/montecarlo_transport.py:94-120,
 * progress_bars.py).  Safe to call from ANOTHER host thread while tardis_mc_propagate blocks (it reads one device word on a stream of
 * its own); exact for the wave-owner kernel (one packet supply per call), 0 until the call is complete for the chunked / lane kernels. */
int tardis_mc_progress(TardisMcContext *ctx, int64_t *out_packets_started, int64_t *out_packets_total);
/* Per-packet results of the resident packets + estimators (re-laid to [L,S]) to caller memory. */
int tardis_mc_get_results(TardisMcContext *ctx, TardisMcResult *result);

/* ---- producer next to the path: black-body packet source on the device --------------------------------
 * BlackBodySimpleSource.create_packets (transport/montecarlo/packet_source/base.py:195-253, black_body.py:140-222) with
 * NumPy's Generator(PCG64) streams reproduced by jump-ahead: fills the resident packet inputs (radii, nus, mus,
 * energies, seeds) of packets [first, first+count) of a global n_total-packet draw, as if set_packets had been called
 * with that slice of the host-sampled collection.  pcg_state = {state_hi, state_lo, inc_hi, inc_lo} of the PCG64
 * bit generator (numpy: default_rng(seed).bit_generator.state, or tardis_mc_pcg64_seed below); max_seed_val is
 * BasePacketSource.MAX_SEED_VAL (base.py:25, 2^32-1); l_array is cumsum(arange(1, l_samples)**-4) (black_body.py:174).
 * packet_seeds are bit-exact; mus bit-exact; nus within 1 ulp of a host run (the reference's log is numexpr's). */
int tardis_mc_pcg64_seed(uint64_t seed, uint64_t out_state[4]);  /* SeedSequence(seed) -> PCG64 state; host only */
int tardis_mc_create_blackbody_packets(TardisMcContext *ctx, int64_t n_total, int64_t first, int64_t count, double radius,
                                       double temperature, const uint64_t pcg_state[4], uint32_t max_seed_val,
                                       const double *l_array, int64_t n_l);
/* download the resident packet inputs (any pointer may be NULL) */
int tardis_mc_get_packets(TardisMcContext *ctx, double *initial_radii, double *initial_nus, double *initial_mus,
                          double *initial_energies, int64_t *packet_seeds);

/* ---- consumer next to the path: real-packet spectrum + filtered luminosities on the device -------------
 * From the per-packet outputs resident after tardis_mc_propagate: histograms of emitted / reabsorbed packet luminosity
 * (+/- output_energy / time_of_simulation) over the spectrum_frequency_grid passed to tardis_mc_set_config, with
 * numpy.histogram's edge rules (tardis/spectrum/base.py:140-159), and the luminosity sums over
 * luminosity_nu_start < nu < luminosity_nu_end (tardis/spectrum/luminosity.py:5-30).  Histograms have n_grid-1 bins. */
int tardis_mc_packet_spectrum(TardisMcContext *ctx, double time_of_simulation, double luminosity_nu_start,
                              double luminosity_nu_end, double *emitted_luminosity_hist, double *reabsorbed_luminosity_hist,
                              double *out_emitted_luminosity, double *out_reabsorbed_luminosity);

/* ---- consumer next to the path: radiation-field update from the resident estimators -------------------
 * MCRadiationFieldPropertiesSolver.solve (transport/montecarlo/estimators/mc_rad_field_solver.py:37-144):
 *   t_radiative[s] = C_T nu_bar[s] / J[s];  dilution_factor[s] = J[s] / (4 sigma_sb t_rad^4 time_of_simulation volume[s]);
 *   j_blues[l][s] = j_blue_estimator[l][s] c t_exp / (4 pi time_of_simulation volume[s]), cells with a zero estimator get
 *   w_epsilon * W[s] B_nu(nu_l, t_rad[s]), and with detailed_optical_window lines outside 2500-10000 A get W B_nu.
 * Runs after tardis_mc_propagate (and, multi-GPU, after tardis_mc_allreduce_estimators).  volume: [n_shells] cm^3.
 * Outputs (host, any may be NULL): t_radiative[n_shells], dilution_factor[n_shells], j_blues[n_lines*n_shells] line-major. */
int tardis_mc_radiation_field(TardisMcContext *ctx, double time_of_simulation, const double *volume, double w_epsilon,
                              int detailed_optical_window, double *t_radiative, double *dilution_factor, double *j_blues);

/* ---- consumer next to the path: the formal integral of the spectrum ----------------------------------------
 * NumbaFormalIntegrator.formal_integral / numba_formal_integral (tardis/spectrum/formal_integral/formal_integral_numba.py:
 * 375-560, 563-642; CUDA twin formal_integral_cuda.py:272-489) on the resident geometry, line list, tau_sobolev and electron
 * densities (set_geometry / set_opacity).  att_S_ul, Jred_lu, Jblue_lu: host, [n_shells * n_lines] in the shell-major flat
 * order the reference passes them (make_source_function, source_function.py:71-75).  Outputs (host): luminosity_densities
 * [n_frequencies]; intensities_nu_p [n_frequencies * n_impact_parameters] or NULL.  tardis_mc_last_propagate_ms then
 * reports the device time of the integration kernels. */
int tardis_mc_formal_integral(TardisMcContext *ctx, double inner_temperature, const double *frequencies, int64_t n_frequencies,
                              const double *att_S_ul, const double *Jred_lu, const double *Jblue_lu, int64_t n_impact_parameters,
                              double *luminosity_densities, double *intensities_nu_p);

/* ---- result streaming (optional).  Registers the caller's per-packet result arrays (output_nus / output_energies and the fourteen li_* arrays of *dst; any may
 * be NULL; the other fields are ignored) as the destination of the NEXT tardis_mc_propagate: a call of the wave-owner kernel that runs as several launches copies the
 * results of the packets handed out so far to these arrays at every launch boundary, beside the next launch, and tardis_mc_get_results -- given the SAME pointers --
 * only copies what is left and the packets that were still in flight when their range was copied.  The arrays must stay valid until tardis_mc_get_results returns.
 * dst = NULL disarms.  Results are identical with and without (tests/test_boundary_gpu.py). */
int tardis_mc_stream_results(TardisMcContext *ctx, const TardisMcResult *dst);
/* What the last tardis_mc_propagate streamed: packets [0, *out_streamed) were copied while the call ran (0: the call did not stream -- one launch, another kernel,
 * fewer packets than the option `stream_min_packets`), *out_resent of them are sent again by tardis_mc_get_results. */
int tardis_mc_streamed_packets(TardisMcContext *ctx, int64_t *out_streamed, int64_t *out_resent);

/* ---- one-shot API: the reference boundary in one call ---------------------------------------------- */
int tardis_mc_run(TardisMcContext *ctx, const TardisMcPackets *packets, const TardisMcGeometry *geometry,
                  const TardisMcOpacity *opacity, const TardisMcConfig *config, TardisMcResult *result);

/* ---- multi-GPU: packets shard by index, one all-reduce of the estimators per iteration (RCCL) ----- */
#define TARDIS_MC_UNIQUE_ID_BYTES 128
int tardis_mc_comm_get_unique_id(uint8_t out_id[TARDIS_MC_UNIQUE_ID_BYTES]);
int tardis_mc_comm_init(TardisMcContext *ctx, int rank, int world_size,
                        const uint8_t id[TARDIS_MC_UNIQUE_ID_BYTES]);
/* In-place sum over ranks of J, nu_bar, j_blue, Edotlu, v-hist (device buffers), on the ctx stream. */
int tardis_mc_allreduce_estimators(TardisMcContext *ctx);
/* Self-check of the communicator (call on every rank after tardis_mc_comm_init): a one-element all-reduce of (rank + 1) must
 * come back as N (N + 1) / 2.  *out_ranks = N on success, 0 otherwise (TARDIS_MC_ERR_COMM / _STATE). */
int tardis_mc_comm_check(TardisMcContext *ctx, int *out_ranks);

/* ---- diagnostics: element-wise device arithmetic, used by the numerics parity tests ------------------
 * op: 0 x+y, 1 x*y, 2 x/y, 3 sqrt(x), 4 log(x) [engine's portable log], 5 exp(x), 6 x*y+x (un-fused),
 *     7 MT19937 doubles of seed (uint32)x[0] (n outputs), 8 floor(x), 9 x/y through the engine's exact 3-fma division. */
int tardis_mc_debug_eval(TardisMcContext *ctx, int op, const double *x, const double *y, double *out, int64_t n);
/* Memory-system micro-benchmarks used to size the kernels (design input): which = 0 random fp64 atomics (agent
 * scope), 1 same at workgroup scope in a per-XCD slice, 2/3 the same with 16 consecutive doubles per 16 lanes,
 * 4 random 8-byte loads, 5 16-lane-coalesced loads; 15 a wide coalesced copy of the table's first half onto its second (iters
 * passes, n_doubles x 8 bytes of traffic each: the box's streaming rate).  blocks x 256 threads x iters operations; time in ms. */
int tardis_mc_debug_microbench(TardisMcContext *ctx, int which, int64_t n_doubles, int iters, int blocks, double *out_ms);

#ifdef __cplusplus
}
#endif
#endif /* TARDIS_MC_H */
