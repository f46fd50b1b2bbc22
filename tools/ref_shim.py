"""DEV-CONTAINER-ONLY tool: import the reference's Monte Carlo hot path in pure-Python mode.

The reference (tardis-sn/tardis, mounted read-only at /root/reference) is Python + Numba.
Numba/astropy are not installed here, but every ``@njit`` / ``@jitclass`` body on the hot path is
plain Python (the reference itself runs them with NUMBA_DISABLE_JIT=1 in its dispatch tests,
tardis/conftest.py:193-199).  This module registers ~60 lines of stub modules so that
``tardis.transport.montecarlo.modes.montecarlo_transport.montecarlo_transport_with_vpackets`` and friends
import and run UNMODIFIED.  It is used only by ``tools/make_golden.py`` to generate the committed
fixtures under ``tests/golden/``.  Nothing here is imported by the product, the tests, or bench.py, and
nothing here travels to the GPU box in a usable form (/root/reference does not exist there).

Recipe documented in SURVEY.md §8(c) / Appendix A.
"""
import os
import sys
import types
import typing

REF_ROOT = os.environ.get("TARDIS_REFERENCE_ROOT", "/root/reference")
REF = os.path.join(REF_ROOT, "tardis")

# CODATA 2010 values (tardis/constants.py:1 = astropy.constants.astropyconst13), cgs
CGS = dict(
    c=2.99792458e10,
    sigma_T=6.652458734e-25,
    h=6.62606957e-27,
    k_B=1.3806488e-16,
    m_e=9.10938291e-28,
    e=4.803204506e-10,
    m_p=1.672621777e-24,
    alpha=7.2973525698e-3,
    sigma_sb=5.670373e-5,
)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Register the stub modules. Must run before any ``import tardis...``."""
    if "tardis" in sys.modules:
        return
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF} (dev container only)")
    if not hasattr(typing, "Self"):
        import typing_extensions

        typing.Self = typing_extensions.Self

    # ---- numba
    def _njit(*a, **k):
        if len(a) == 1 and callable(a[0]):  # @njit, @njit(**opts) applied directly: njit(f, **opts)
            return a[0]
        return lambda f: f

    class _T:
        def __getitem__(self, i):
            return self

        def __call__(self, *a, **k):
            return self

    _t = _T()

    def _jitclass(x=None, *a, **k):
        if isinstance(x, type):
            return x
        return lambda c: c

    class _objmode:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def __call__(self, *a, **k):
            return self

    nb = _mod(
        "numba",
        njit=_njit,
        jit=_njit,
        prange=range,
        objmode=_objmode(),
        float64=_t,
        int64=_t,
        boolean=_t,
        set_num_threads=lambda n: None,
        cuda=types.SimpleNamespace(is_available=lambda: False, jit=_njit),
    )
    _mod("numba.experimental", jitclass=_jitclass)
    nb.experimental = sys.modules["numba.experimental"]
    _mod("numba.typed", List=list)
    nb.typed = sys.modules["numba.typed"]
    _mod("numba.np")
    _mod("numba.np.ufunc")
    _mod("numba.np.ufunc.parallel", get_num_threads=lambda: 1, get_thread_id=lambda: 0)
    # ---- llvmlite
    _mod("llvmlite", binding=types.SimpleNamespace(set_option=lambda *a: None))
    _mod("llvmlite.binding", set_option=lambda *a: None)
    # ---- radioactivedecay (atomic masses at import time of tardis.opacities.opacities)
    _mod(
        "radioactivedecay",
        Nuclide=lambda s: types.SimpleNamespace(
            atomic_mass={"Si-28": 27.9769265350, "Fe-56": 55.9349363}[s]
        ),
    )

    # ---- astropy: `units as u` + a Quantity-lite.  Everything on the paths imported here is already cgs, so a Quantity is its
    # value plus a unit NAME; the only conversion performed is the one the reference performs, Hz -> Angstrom through
    # u.spectral() (mc_rad_field_solver.py:136).  ndarray * Quantity defers to the Quantity (as astropy does).
    import numpy as _np

    class Q:
        __array_priority__ = 1.0e4
        __array_ufunc__ = None

        def __init__(self, v, unit=""):
            self.value = v
            self.unit = unit

        def _v(self, o):
            return o.value if isinstance(o, Q) else o

        def to(self, target=None, equivalencies=None, *a, **k):
            tname = target.unit if isinstance(target, Q) else str(target or "")
            if self.unit == "Hz" and tname == "AA":  # u.spectral(): lambda = c / nu
                return Q(CGS["c"] / _np.asarray(self.value) * 1e8, "AA")
            return Q(self.value, tname or self.unit)

        cgs = property(lambda self: self)
        esu = property(lambda self: self)
        si = property(lambda self: self)

        def copy(self):
            return Q(_np.array(self.value, copy=True), self.unit)

        def __len__(self):
            return len(self.value)

        def __getitem__(self, i):
            return Q(self.value[i], self.unit)

        def __mul__(self, o):
            return Q(self.value * self._v(o), self.unit or getattr(o, "unit", ""))

        __rmul__ = __mul__

        def __truediv__(self, o):
            return Q(self.value / self._v(o), self.unit)

        def __rtruediv__(self, o):
            return Q(o / self.value, self.unit)

        def __pow__(self, p):
            return Q(self.value**p, self.unit)

        def __neg__(self):
            return Q(-self.value, self.unit)

        def __add__(self, o):
            return Q(self.value + self._v(o), self.unit)

        __radd__ = __add__

        def __sub__(self, o):
            return Q(self.value - self._v(o), self.unit)

        def __gt__(self, o):
            return self.value > self._v(o)

        def __lt__(self, o):
            return self.value < self._v(o)

        def __ge__(self, o):
            return self.value >= self._v(o)

        def __le__(self, o):
            return self.value <= self._v(o)

        def __array__(self, dtype=None, copy=None):
            return _np.asarray(self.value, dtype=dtype)

    class _U:
        Quantity = Q

        def __getattr__(self, n):
            return Q(1.0, n)

        def spectral(self):
            return "spectral"

    _mod("astropy", units=_U())
    sys.modules["astropy.units"] = _U()

    # ---- fake `tardis` root so tardis/__init__.py (importlib.metadata + astropy) is not executed
    t = _mod("tardis")
    t.__path__ = [REF]
    const = _mod("tardis.constants", **{k: Q(v) for k, v in CGS.items()})
    t.constants = const
    for pkg in ["tardis.model", "tardis.model.geometry", "tardis.opacities", "tardis.io", "tardis.plasma"]:
        m = _mod(pkg)
        m.__path__ = [REF + "/" + "/".join(pkg.split(".")[1:])]
    import importlib

    importlib.import_module("tardis.transport")
    _m = _mod("tardis.transport.montecarlo.estimators")
    _m.__path__ = [REF + "/transport/montecarlo/estimators"]
    _noop = lambda *a, **k: None
    _mod(
        "tardis.transport.montecarlo.progress_bars",
        update_packets_pbar=_noop,
        reset_packet_pbar=_noop,
        refresh_packet_pbar=_noop,
        update_iterations_pbar=_noop,
        initialize_iterations_pbar=_noop,
        iterations_pbar=None,
        packet_pbar=None,
    )


def load():
    """Return a namespace with the reference's hot-path callables/classes."""
    install()
    from tardis.model.geometry.radial1d_homologous import NumbaHomologousRadial1DGeometry
    from tardis.opacities.opacity_state_numba import OpacityStateNumba
    from tardis.transport.montecarlo.configuration.base import MonteCarloConfiguration
    from tardis.transport.montecarlo.modes.classic.packet_propagation import packet_propagation
    from tardis.transport.montecarlo.modes.montecarlo_transport import (
        montecarlo_transport_with_vpackets,
    )
    from tardis.transport.montecarlo.packets.packet_collections import PacketCollection
    from tardis.transport.montecarlo.packets.trackers.tracker_last_interaction import (
        TrackerLastInteraction,
    )

    return types.SimpleNamespace(**locals())


def _function_from_reference(path, name, namespace):
    """Compile ONE function of a reference module from the reference's own source (the module as a whole needs pandas /
    radioactivedecay data / package metadata at import) and return it; nothing is copied into this repository."""
    import ast

    src = open(path).read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
    exec(code, namespace)
    return namespace[name]


def load_next_rows():
    """The reference's radiation-field solver and black-body packet source (the rows next to the hot path)."""
    install()
    import numpy as np
    import numpy as _np

    try:  # (pandas probes numexpr's version at import: let it load before the stand-in exists)
        import pandas  # noqa: F401
    except Exception:
        pass
    Q = sys.modules["astropy.units"].Quantity

    # ---- numexpr: evaluate() of an expression over the caller's variables, with numpy's functions (numexpr's own log / exp
    # kernels differ from numpy's by at most an ulp or two: the fixtures made through this stand-in are compared at 1e-13)
    def _ne_evaluate(expr, local_dict=None, global_dict=None, **kw):
        f = sys._getframe(1)
        ns = {"log": _np.log, "exp": _np.exp, "sqrt": _np.sqrt}
        for d in (f.f_globals, f.f_locals if local_dict is None else local_dict):
            for k, v in d.items():
                ns[k] = _np.asarray(v.value) if isinstance(v, Q) else v
        return eval(expr, {"__builtins__": {}}, ns)

    if "numexpr" not in sys.modules:
        _mod("numexpr", evaluate=_ne_evaluate)

    if "tardis.util.base" not in sys.modules:
        # tardis/util/base.py:279-302 intensity_black_body, with the module-level constants it reads (util/base.py:30-40)
        ns = {"ne": sys.modules["numexpr"], "np": np, "k_B_cgs": CGS["k_B"], "h_cgs": CGS["h"], "c_cgs": CGS["c"]}
        fn = _function_from_reference(os.path.join(REF, "util", "base.py"), "intensity_black_body", ns)
        _mod("tardis.util")
        _mod("tardis.util.base", intensity_black_body=fn)
        _mod("tardis.io.hdf_writer_mixin", HDFWriterMixin=type("HDFWriterMixin", (), {}))
        # (the package __init__ pulls every packet source in, the gamma-ray ones with their pandas / data-file imports)
        m = _mod("tardis.transport.montecarlo.packet_source")
        m.__path__ = [REF + "/transport/montecarlo/packet_source"]
    from tardis.transport.montecarlo.estimators.estimators_bulk import EstimatorsBulk
    from tardis.transport.montecarlo.estimators.estimators_line import EstimatorsLine
    from tardis.transport.montecarlo.estimators.mc_rad_field_solver import MCRadiationFieldPropertiesSolver
    from tardis.transport.montecarlo.packet_source.black_body import BlackBodySimpleSource

    Quantity = sys.modules["astropy.units"].Quantity
    return types.SimpleNamespace(**locals())
