"""ctypes binding for the GENUINE reference build (oracle/_ref/libspiel_ref.so).

TEST INFRASTRUCTURE ONLY, like oracle_py: imported by tests/ and by bench.py's
cpu_baseline leg, never by the product package.

oracle/_ref/libspiel_ref.so is the reference's own .cc files for the hot path,
compiled unmodified from /root/reference by oracle/Makefile.ref (against the
private abseil / nlohmann stand-ins in oracle/ref_shim) and driven through the
same extern "C" entry points as the restatement (spiel_oracle_capi.cpp built
with -DOSGO_GENUINE_REFERENCE).  This module is oracle_py's code bound to that
library, so `reference_py.Game("kuhn_poker")` and `oracle_py.Game("kuhn_poker")`
answer the same calls from the two implementations.

The library is built in the development container (where /root/reference
exists) and travels to the GPU box as a prebuilt file; `available()` says
whether it can be used, `build()` (re)builds it when the sources are present.
"""
import importlib.util
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libspiel_ref.so")
REFERENCE_ROOT = os.environ.get("OSG_REFERENCE_ROOT", "/root/reference")


def sources_present():
    return os.path.exists(os.path.join(REFERENCE_ROOT, "open_spiel", "spiel.cc"))


def build(force=False):
    """make -f Makefile.ref (no-op without the reference sources)."""
    if not sources_present():
        return LIB_PATH if os.path.exists(LIB_PATH) else None
    if force:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-f", "Makefile.ref", "clean"])
    subprocess.check_call(["make", "-s", "-C", _HERE, "-f", "Makefile.ref", "-j8",
                           "REF=" + REFERENCE_ROOT])
    return LIB_PATH


def available():
    return os.path.exists(LIB_PATH)


def _bind():
    spec = importlib.util.spec_from_file_location("_oracle_py_bound_to_reference",
                                                  os.path.join(_HERE, "oracle_py.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    mod._LIB_PATH = LIB_PATH
    mod.build = build
    return mod


_impl = _bind()
# Same names as oracle_py.
for _name in dir(_impl):
    if not _name.startswith("_"):
        globals().setdefault(_name, getattr(_impl, _name))
lib = _impl.lib
