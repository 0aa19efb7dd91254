"""stcsp-solver_amd -- MI355X-native stream-CSP propagation + search engine.

Host-side Python mirror of the C-ABI in include/stcsp_engine.h / include/stcsp_host.h (ctypes,
no torch types in any signature).  The package name contains a hyphen (it mirrors the upstream
repository name), so import it with::

    import importlib; stcsp = importlib.import_module("stcsp-solver_amd")

The HIP engine library is REQUIRED for `Engine`: there is no CPU fallback anywhere in this
package; a missing / unloadable libstcsp_hip.so raises immediately.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
REPO = PKG_DIR.parent

from . import instances  # noqa: E402  (pure python, no native deps)


# ------------------------------------------------------------------ ctypes mirrors
class Node(C.Structure):
    _fields_ = [("token", C.c_int32), ("num", C.c_int32), ("var", C.c_int32), ("arr", C.c_int32),
                ("left", C.c_int32), ("right", C.c_int32)]


class Problem(C.Structure):
    _fields_ = [("n_vars", C.c_int32), ("prefix_k", C.c_int32),
                ("var_lb", C.POINTER(C.c_int32)), ("var_ub", C.POINTER(C.c_int32)),
                ("var_names", C.POINTER(C.c_char_p)),
                ("n_arrays", C.c_int32), ("array_off", C.POINTER(C.c_int32)), ("array_data", C.POINTER(C.c_int32)),
                ("n_nodes", C.c_int32), ("nodes", C.POINTER(Node)),
                ("n_constraints", C.c_int32), ("constraint_root", C.POINTER(C.c_int32))]


class Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("batch_nodes", C.c_int32),
                ("max_search_nodes", C.c_int64), ("time_limit_s", C.c_double), ("flags", C.c_int32),
                ("reserved", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("search_nodes", C.c_int64), ("gac_calls", C.c_int64), ("fails", C.c_int64),
                ("dominance", C.c_int64), ("leaves", C.c_int64), ("revisions", C.c_int64),
                ("evaluations", C.c_int64), ("levels", C.c_int64),
                ("seconds_search", C.c_double), ("seconds_export", C.c_double),
                ("seconds_expand_kernel", C.c_double), ("expand_launches", C.c_int64),
                ("wave_revisions", C.c_int64), ("sweeps", C.c_int64), ("skipped_revisions", C.c_int64),
                ("translation_stops", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Result(C.Structure):
    _fields_ = [("n_states", C.c_int64), ("sig_len", C.c_int32), ("n_sig_vars", C.c_int32),
                ("n_until", C.c_int32), ("n_until_cons", C.c_int32),
                ("state_cid", C.POINTER(C.c_int32)), ("state_sig", C.POINTER(C.c_int32)),
                ("state_fail", C.POINTER(C.c_uint8)),
                ("n_edges", C.c_int64), ("edge_src", C.POINTER(C.c_int64)), ("edge_dst", C.POINTER(C.c_int64)),
                ("edge_values", C.POINTER(C.c_int32)),
                ("n_vars", C.c_int32), ("n_constraint_sets", C.c_int32),
                ("var_is_signature", C.POINTER(C.c_uint8)),
                ("root_final", C.c_int32), ("truncated", C.c_int32),
                ("counters", Counters)]


class PostOptions(C.Structure):
    _fields_ = [("adversarial_var", C.c_int32), ("adversarial2_op", C.c_int32), ("adversarial2_ava", C.c_int32),
                ("reserved", C.c_int32)]


class PostResult(C.Structure):
    _fields_ = [("n_states", C.c_int64), ("n_edges", C.c_int64),
                ("state_valid", C.POINTER(C.c_uint8)), ("state_final", C.POINTER(C.c_uint8)),
                ("edge_alive", C.POINTER(C.c_uint8)),
                ("adver1", C.c_int32), ("adver2", C.c_int32), ("rounds", C.c_int32 * 3), ("seconds", C.c_double)]


F_KEEP_RAW_EDGES = 1
F_NO_EXPORT = 2
F_PROFILE = 4
F_STEPPED = 8
GID_SHIFT = 40

ENGINE_SYMBOLS = [
    "stcsp_engine_create", "stcsp_engine_solve", "stcsp_engine_export", "stcsp_engine_destroy",
    "stcsp_engine_last_error", "stcsp_engine_begin", "stcsp_engine_expand_local",
    "stcsp_engine_candidate_bytes", "stcsp_engine_outbox", "stcsp_engine_commit", "stcsp_engine_finish",
    "stcsp_engine_counters", "stcsp_engine_sets_blob", "stcsp_engine_sets_import", "stcsp_engine_postprocess",
    "stcsp_engine_propagate", "stcsp_engine_set_expand_budget", "stcsp_engine_node_bytes", "stcsp_engine_donate",
    "stcsp_engine_adopt",
]
# include/stcsp_sharded.h: the superstep loop + in-process transport (libstcsp_hip.so), the RCCL transport (libstcsp_rccl.so)
SHARDED_SYMBOLS_HIP = ["stcsp_engine_solve_sharded", "stcsp_local_group_create", "stcsp_local_group_transport", "stcsp_local_group_destroy"]
SHARDED_SYMBOLS_RCCL = ["stcsp_rccl_unique_id", "stcsp_transport_rccl_create", "stcsp_transport_rccl_destroy"]
HOST_SYMBOLS = [
    "stcsp_model_load_file", "stcsp_model_load_text", "stcsp_model_problem", "stcsp_model_free",
    "stcsp_host_last_error", "stcsp_model_constraint_string",
    "stcsp_automaton_build", "stcsp_automaton_free", "stcsp_automaton_traverse",
    "stcsp_automaton_adversarial", "stcsp_automaton_adversarial2", "stcsp_automaton_renumber",
    "stcsp_automaton_import_flags", "stcsp_automaton_flags", "stcsp_automaton_order_by_label", "stcsp_automaton_write_binary", "stcsp_automaton_read_binary",
    "stcsp_automaton_write_dot", "stcsp_automaton_canonical", "stcsp_automaton_num_states",
    "stcsp_automaton_num_live_states", "stcsp_automaton_num_live_edges",
    "stcsp_merge_shards", "stcsp_merged_result", "stcsp_merged_free", "stcsp_host_free",
]


class StcspError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"stcsp error {code}: {msg}")
        self.code = code


def engine_source_sha() -> str:
    """sha256 (first 16 hex digits) over the sources libstcsp_hip.so is built from: ties a committed PMC figure to the
    engine it was measured on (bench.py quotes profiles/*_traffic.json only when this matches)."""
    h = hashlib.sha256()
    for f in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("dev_*.hpp")) + [CSRC / "device_types.hpp", CSRC / "cset.cpp", CSRC / "cset.hpp", CSRC / "sharded_native.hpp"]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


# ------------------------------------------------------------------ library loading / building
def build(verbose: bool = False) -> None:
    """Compile every native library in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    cmd = ["make", "-C", str(CSRC), "all"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout, r.stderr)
    if r.returncode != 0:
        raise RuntimeError("native build failed")
    global _host, _hip
    _host = _hip = None


_host = None
_hip = None


def host_lib() -> C.CDLL:
    global _host
    if _host is None:
        path = CSRC / "libstcsp_host.so"
        if not path.exists():
            raise RuntimeError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(str(path))
        lib.stcsp_model_load_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        lib.stcsp_model_load_text.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        lib.stcsp_model_problem.argtypes = [C.c_void_p]
        lib.stcsp_model_problem.restype = C.POINTER(Problem)
        lib.stcsp_model_free.argtypes = [C.c_void_p]
        lib.stcsp_host_last_error.restype = C.c_char_p
        lib.stcsp_model_constraint_string.argtypes = [C.c_void_p, C.c_int]
        lib.stcsp_model_constraint_string.restype = C.c_void_p
        lib.stcsp_automaton_build.argtypes = [C.POINTER(Problem), C.POINTER(Result), C.POINTER(C.c_void_p)]
        lib.stcsp_automaton_free.argtypes = [C.c_void_p]
        lib.stcsp_automaton_traverse.argtypes = [C.c_void_p]
        lib.stcsp_automaton_adversarial.argtypes = [C.c_void_p, C.c_int]
        lib.stcsp_automaton_adversarial2.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.stcsp_automaton_renumber.argtypes = [C.c_void_p]
        lib.stcsp_automaton_order_by_label.argtypes = [C.c_void_p]
        lib.stcsp_automaton_import_flags.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.stcsp_automaton_flags.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.stcsp_automaton_write_dot.argtypes = [C.c_void_p, C.c_char_p]
        lib.stcsp_automaton_write_binary.argtypes = [C.c_void_p, C.c_char_p]
        lib.stcsp_automaton_read_binary.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.stcsp_automaton_canonical.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        lib.stcsp_automaton_canonical.restype = C.c_void_p
        for f in ("stcsp_automaton_num_states", "stcsp_automaton_num_live_states", "stcsp_automaton_num_live_edges"):
            getattr(lib, f).argtypes = [C.c_void_p]
            getattr(lib, f).restype = C.c_int64
        lib.stcsp_merge_shards.argtypes = [C.POINTER(C.POINTER(Result)), C.c_int, C.POINTER(C.c_void_p)]
        lib.stcsp_merged_result.argtypes = [C.c_void_p]
        lib.stcsp_merged_result.restype = C.POINTER(Result)
        lib.stcsp_merged_free.argtypes = [C.c_void_p]
        lib.stcsp_host_free.argtypes = [C.c_void_p]
        _host = lib
    return _host


def bind_engine_api(lib: C.CDLL, prefix: str = "stcsp_engine") -> None:
    """Attach argtypes to an engine-shaped API (the HIP engine, or -- in tests only -- an
    oracle exposing the same shape under another prefix)."""
    g = lambda n: getattr(lib, f"{prefix}_{n}")  # noqa: E731
    g("create").argtypes = [C.POINTER(Problem), C.POINTER(Options), C.POINTER(C.c_void_p)]
    g("solve").argtypes = [C.c_void_p, C.POINTER(Result)]
    g("destroy").argtypes = [C.c_void_p]
    g("destroy").restype = None
    for n in ("export",):
        if hasattr(lib, f"{prefix}_{n}"):
            g(n).argtypes = [C.c_void_p, C.POINTER(Result)]
    if hasattr(lib, f"{prefix}_last_error"):
        g("last_error").argtypes = [C.c_void_p]
        g("last_error").restype = C.c_char_p
    if hasattr(lib, f"{prefix}_begin"):
        g("begin").argtypes = [C.c_void_p]
        g("expand_local").argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        g("candidate_bytes").argtypes = [C.c_void_p]
        g("outbox").argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        g("commit").argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        g("finish").argtypes = [C.c_void_p]
    if hasattr(lib, f"{prefix}_donate"):
        g("set_expand_budget").argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        g("node_bytes").argtypes = [C.c_void_p]
        g("donate").argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        g("adopt").argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    if hasattr(lib, f"{prefix}_counters"):
        g("counters").argtypes = [C.c_void_p, C.POINTER(Counters)]
        g("sets_blob").argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int64)]
        g("sets_import").argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int64]
    if hasattr(lib, f"{prefix}_postprocess"):
        g("postprocess").argtypes = [C.c_void_p, C.POINTER(PostOptions), C.POINTER(PostResult)]
    if hasattr(lib, f"{prefix}_propagate"):
        g("propagate").argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.POINTER(C.c_uint32), C.c_int64, C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int64)]


def hip_lib() -> C.CDLL:
    """The HIP engine. Fails loudly when the extension is missing or cannot be loaded."""
    global _hip
    if _hip is None:
        path = Path(os.environ.get("STCSP_HIP_LIB", CSRC / "libstcsp_hip.so"))  # override: tuning builds
        if not path.exists():
            raise RuntimeError(f"HIP engine library {path} is missing -- build it (make -C {CSRC}); "
                               "there is no CPU fallback")
        lib = C.CDLL(str(path))
        bind_engine_api(lib)
        _hip = lib
    return _hip


# ------------------------------------------------------------------ model (front end)
class Model:
    """A built stCSP model = what solverSolve() receives (reference src/solver.h:22-49)."""

    def __init__(self, text: str | None = None, path: str | None = None, prefix_k: int = 2):
        lib = host_lib()
        h = C.c_void_p()
        if path is not None:
            rc = lib.stcsp_model_load_file(os.fsencode(path), prefix_k, C.byref(h))
        else:
            rc = lib.stcsp_model_load_text(text.encode(), prefix_k, C.byref(h))
        if rc != 0:
            raise StcspError(rc, lib.stcsp_host_last_error().decode())
        self._h = h
        self.problem = lib.stcsp_model_problem(h)

    @classmethod
    def from_name(cls, name: str, prefix_k: int = 2) -> "Model":
        return cls(text=instances.by_name(name), prefix_k=prefix_k)

    @property
    def n_vars(self):
        return self.problem.contents.n_vars

    @property
    def n_constraints(self):
        return self.problem.contents.n_constraints

    @property
    def var_names(self):
        p = self.problem.contents
        return [p.var_names[i].decode() for i in range(p.n_vars)]

    def var_bounds(self):
        p = self.problem.contents
        return [(p.var_lb[i], p.var_ub[i]) for i in range(p.n_vars)]

    def constraint_string(self, i: int) -> str:
        lib = host_lib()
        p = lib.stcsp_model_constraint_string(self._h, i)
        s = C.string_at(p).decode()
        lib.stcsp_host_free(p)
        return s

    def close(self):
        if self._h:
            host_lib().stcsp_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ automaton (post-processing)
class Automaton:
    def __init__(self, model: Model, result: Result):
        lib = host_lib()
        h = C.c_void_p()
        rc = lib.stcsp_automaton_build(model.problem, C.byref(result), C.byref(h))
        if rc != 0:
            raise StcspError(rc, "automaton build failed")
        self._h = h
        self._model = model
        self._n_edges = result.n_edges

    def traverse(self):
        host_lib().stcsp_automaton_traverse(self._h)
        return self

    def adversarial(self, var_index: int = 5) -> int:
        return host_lib().stcsp_automaton_adversarial(self._h, var_index)

    def adversarial2(self, opponent: int = 5, avatar: int = 6) -> int:
        return host_lib().stcsp_automaton_adversarial2(self._h, opponent, avatar)

    def renumber(self):
        host_lib().stcsp_automaton_renumber(self._h)
        return self

    def order_by_label(self):
        """Scheduling-independent output order (reproducible solutions.dot / binary files)."""
        host_lib().stcsp_automaton_order_by_label(self._h)
        return self

    def import_flags(self, post: "PostResult"):
        """Adopt the flags of Engine.postprocess() (device passes) instead of traverse()/adversarial*()."""
        rc = host_lib().stcsp_automaton_import_flags(self._h, post.state_valid, post.state_final, post.edge_alive)
        if rc != 0:
            raise StcspError(rc, "import_flags failed")
        return self

    def flags(self):
        """(valid, final, alive) as bytes objects."""
        lib = host_lib()
        ns = lib.stcsp_automaton_num_states(self._h)
        ne = self._n_edges
        v, f, a = C.create_string_buffer(max(ns, 1)), C.create_string_buffer(max(ns, 1)), C.create_string_buffer(max(ne, 1))
        lib.stcsp_automaton_flags(self._h, v, f, a)
        return v.raw[:ns], f.raw[:ns], a.raw[:ne]

    def write_dot(self, path: str):
        rc = host_lib().stcsp_automaton_write_dot(self._h, os.fsencode(path))
        if rc != 0:
            raise StcspError(rc, f"cannot write {path}")

    def write_binary(self, path: str):
        rc = host_lib().stcsp_automaton_write_binary(self._h, os.fsencode(path))
        if rc != 0:
            raise StcspError(rc, f"cannot write {path}")

    @classmethod
    def read_binary(cls, path: str) -> "Automaton":
        h = C.c_void_p()
        rc = host_lib().stcsp_automaton_read_binary(os.fsencode(path), C.byref(h))
        if rc != 0:
            raise StcspError(rc, f"cannot read {path}")
        a = cls.__new__(cls)
        a._h, a._model, a._n_edges = h, None, 0
        return a

    def canonical(self) -> str:
        lib = host_lib()
        n = C.c_size_t()
        p = lib.stcsp_automaton_canonical(self._h, C.byref(n))
        s = C.string_at(p, n.value).decode()
        lib.stcsp_host_free(p)
        return s

    def canonical_sha256(self) -> str:
        return hashlib.sha256(self.canonical().encode()).hexdigest()

    @property
    def n_states(self):
        return host_lib().stcsp_automaton_num_states(self._h)

    @property
    def n_live_states(self):
        return host_lib().stcsp_automaton_num_live_states(self._h)

    @property
    def n_live_edges(self):
        return host_lib().stcsp_automaton_num_live_edges(self._h)

    def close(self):
        if self._h:
            host_lib().stcsp_automaton_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ engine
class EngineBase:
    """Common driver for an engine-shaped C-ABI (create / solve / destroy + stepping calls)."""

    _prefix = "stcsp_engine"

    def __init__(self, lib: C.CDLL, model: Model, **opts):
        self._lib = lib
        self._model = model
        o = Options()
        o.device = opts.get("device", 0)
        o.rank = opts.get("rank", 0)
        o.world = opts.get("world", 1)
        o.batch_nodes = opts.get("batch_nodes", 0)
        o.max_search_nodes = opts.get("max_search_nodes", 0)
        o.time_limit_s = opts.get("time_limit_s", 0.0)
        o.flags = opts.get("flags", 0)
        self.options = o
        h = C.c_void_p()
        rc = self._f("create")(model.problem, C.byref(o), C.byref(h))
        if rc != 0:
            raise StcspError(rc, self._last_error(None))
        self._h = h
        self.result = Result()

    def _f(self, name):
        return getattr(self._lib, f"{self._prefix}_{name}")

    def _last_error(self, h):
        if hasattr(self._lib, f"{self._prefix}_last_error"):
            m = self._f("last_error")(h)
            return m.decode() if m else ""
        return ""

    def _check(self, rc):
        if rc != 0:
            raise StcspError(rc, self._last_error(self._h))

    def solve(self) -> Result:
        self._check(self._f("solve")(self._h, C.byref(self.result)))
        return self.result

    def export(self) -> Result:
        self._check(self._f("export")(self._h, C.byref(self.result)))
        return self.result

    # sharded stepping interface
    def begin(self):
        self._check(self._f("begin")(self._h))

    def expand_local(self) -> int:
        left = C.c_int64()
        self._check(self._f("expand_local")(self._h, C.byref(left)))
        return left.value

    def candidate_bytes(self) -> int:
        return self._f("candidate_bytes")(self._h)

    def outbox(self, peer: int):
        p = C.c_void_p()
        n = C.c_int64()
        self._check(self._f("outbox")(self._h, peer, C.byref(p), C.byref(n)))
        return p.value or 0, n.value

    def commit(self, ptr: int, count: int):
        self._check(self._f("commit")(self._h, C.c_void_p(ptr), count))

    def finish(self):
        self._check(self._f("finish")(self._h))

    def set_expand_budget(self, max_rounds: int, min_open: int):
        self._check(self._f("set_expand_budget")(self._h, max_rounds, min_open))

    def node_bytes(self) -> int:
        return self._f("node_bytes")(self._h)

    def donate(self, want: int):
        """Remove up to `want` of the oldest open nodes; returns (pointer, count) of their transfer records."""
        ptr, n = C.c_void_p(), C.c_int64()
        self._check(self._f("donate")(self._h, want, C.byref(ptr), C.byref(n)))
        return (ptr.value or 0), n.value

    def adopt(self, ptr: int, count: int):
        self._check(self._f("adopt")(self._h, C.c_void_p(ptr), count))

    def counters(self) -> Counters:
        c = Counters()
        self._check(self._f("counters")(self._h, C.byref(c)))
        return c

    def sets_blob(self):
        """This shard's constraint-set registry as a list of int32 words."""
        p = C.POINTER(C.c_int32)()
        n = C.c_int64()
        self._check(self._f("sets_blob")(self._h, C.byref(p), C.byref(n)))
        return [p[i] for i in range(n.value)]

    def sets_count(self) -> int:
        """Number of constraint sets this shard knows (word 0 of the registry blob)."""
        p = C.POINTER(C.c_int32)()
        n = C.c_int64()
        self._check(self._f("sets_blob")(self._h, C.byref(p), C.byref(n)))
        return p[0] if n.value else 0

    def sets_import(self, words):
        arr = (C.c_int32 * len(words))(*words)
        self._check(self._f("sets_import")(self._h, arr, len(words)))

    def postprocess(self, adversarial: int = -1, adversarial2: tuple | None = None) -> PostResult:
        """graphTraverse [+ adversarialTraverse(var)] [+ adversarialTraverse2(op, ava)] on the device."""
        po = PostOptions(adversarial, adversarial2[0] if adversarial2 else -1, adversarial2[1] if adversarial2 else -1, 0)
        out = PostResult()
        self._check(self._f("postprocess")(self._h, C.byref(po), C.byref(out)))
        return out

    def automaton(self, result: Result | None = None) -> Automaton:
        return Automaton(self._model, result if result is not None else self.result)

    def close(self):
        if getattr(self, "_h", None):
            self._f("destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine(EngineBase):
    """The MI355X engine (HIP, gfx950).  Replaces solverSolve's search (reference
    src/solveralgorithm.cpp:966-971)."""

    def __init__(self, model: Model, **opts):
        super().__init__(hip_lib(), model, **opts)

    def propagate(self, blocks, set_index: int = 0, expire: int = 0):
        """Kernel-granularity check: run the device code of one search node on each row of `blocks`
        (numpy uint32 [count, N*K]). Returns (propagated blocks, outcome per block, skipped revisions)."""
        import numpy as np
        b = np.ascontiguousarray(blocks, dtype=np.uint32).copy()
        count = b.shape[0]
        outcome = np.zeros(count, dtype=np.int32)
        skipped = C.c_int64(0)
        self._check(self._f("propagate")(self._h, set_index, expire, b.ctypes.data_as(C.POINTER(C.c_uint32)), count,
                                         outcome.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(skipped)))
        return b, outcome, skipped.value


# ------------------------------------------------------------------ native sharded solve (include/stcsp_sharded.h)
class ShardedOptions(C.Structure):
    _fields_ = [("budget_rounds", C.c_int64), ("share_per_rank", C.c_int64), ("max_supersteps", C.c_int64)]


class ShardedStats(C.Structure):
    _fields_ = [("supersteps", C.c_int64), ("nodes_donated", C.c_int64), ("nodes_adopted", C.c_int64),
                ("candidates_sent", C.c_int64), ("candidates_received", C.c_int64), ("seconds_collectives", C.c_double),
                ("seconds_expand", C.c_double), ("seconds_pack", C.c_double), ("seconds_commit", C.c_double)]


def _bind_sharded(lib):
    if getattr(lib, "_stcsp_sharded_bound", False):
        return
    lib.stcsp_engine_solve_sharded.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ShardedOptions), C.POINTER(ShardedStats)]
    lib.stcsp_local_group_create.argtypes = [C.c_int32, C.POINTER(C.c_void_p)]
    lib.stcsp_local_group_transport.argtypes = [C.c_void_p, C.c_int32]
    lib.stcsp_local_group_transport.restype = C.c_void_p
    lib.stcsp_local_group_destroy.argtypes = [C.c_void_p]
    lib.stcsp_local_group_destroy.restype = None
    lib._stcsp_sharded_bound = True


def solve_sharded_native(engine: "Engine", transport: int, budget_rounds: int = 8, share_per_rank: int = 64) -> dict:
    """stcsp_engine_solve_sharded: the whole superstep loop of this rank inside the engine library (no Python, no torch
    between two bursts of k_expand); `transport` = pointer to a stcsp_transport (LocalGroup.transport(r), RcclTransport.ptr)."""
    lib = hip_lib()
    _bind_sharded(lib)
    o = ShardedOptions(budget_rounds, share_per_rank, 0)
    st = ShardedStats()
    engine._check(lib.stcsp_engine_solve_sharded(engine._h, C.c_void_p(transport), C.byref(o), C.byref(st)))
    return {n: getattr(st, n) for n, _ in st._fields_}


class LocalGroup:
    """stcsp_local_group: `world` transports for `world` engines of ONE process, one host thread each (same or different
    GPUs; records move with hipMemcpy[Peer]Async)."""

    def __init__(self, world: int):
        lib = hip_lib()
        _bind_sharded(lib)
        self._lib, self.world = lib, world
        h = C.c_void_p()
        rc = lib.stcsp_local_group_create(world, C.byref(h))
        if rc != 0:
            raise StcspError(rc, "local group")
        self._h = h

    def transport(self, rank: int) -> int:
        return self._lib.stcsp_local_group_transport(self._h, rank)

    def solve(self, engines, **knobs):
        """Run stcsp_engine_solve_sharded on every engine, one thread per rank; returns the per-rank stats (raises the
        first rank's error after ALL threads have ended)."""
        import threading
        out, errs = [None] * self.world, [None] * self.world

        def run(r):
            try:
                out[r] = solve_sharded_native(engines[r], self.transport(r), **knobs)
            except Exception as ex:  # noqa: BLE001
                errs[r] = ex

        ts = [threading.Thread(target=run, args=(r,)) for r in range(self.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        self.errors = errs
        for ex in errs:
            if ex is not None:
                raise ex
        return out

    def close(self):
        if self._h:
            self._lib.stcsp_local_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_rccl = None


def rccl_lib() -> C.CDLL:
    """libstcsp_rccl.so: the stcsp_transport over RCCL (one process per GPU)."""
    global _rccl
    if _rccl is None:
        path = CSRC / "libstcsp_rccl.so"
        if not path.exists():
            raise RuntimeError(f"{path} is missing -- build it (make -C {CSRC})")
        lib = C.CDLL(str(path))
        lib.stcsp_rccl_unique_id.argtypes = [C.c_void_p]
        lib.stcsp_transport_rccl_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        lib.stcsp_transport_rccl_destroy.argtypes = [C.c_void_p]
        lib.stcsp_transport_rccl_destroy.restype = None
        _rccl = lib
    return _rccl


RCCL_ID_BYTES = 128


def rccl_unique_id() -> bytes:
    buf = C.create_string_buffer(RCCL_ID_BYTES)
    rc = rccl_lib().stcsp_rccl_unique_id(buf)
    if rc != 0:
        raise StcspError(rc, "ncclGetUniqueId failed")
    return buf.raw


class RcclTransport:
    def __init__(self, unique_id: bytes, rank: int, world: int, device: int):
        lib = rccl_lib()
        h = C.c_void_p()
        rc = lib.stcsp_transport_rccl_create(C.create_string_buffer(unique_id, RCCL_ID_BYTES), rank, world, device, C.byref(h))
        if rc != 0:
            raise StcspError(rc, "ncclCommInitRank failed")
        self._lib, self.ptr = lib, h.value

    def close(self):
        if self.ptr:
            self._lib.stcsp_transport_rccl_destroy(C.c_void_p(self.ptr))
            self.ptr = None


def merge_shards(results):
    """stcsp_merge_shards over a list of Result structs; returns (handle, Result)."""
    lib = host_lib()
    arr = (C.POINTER(Result) * len(results))(*[C.pointer(r) for r in results])
    h = C.c_void_p()
    rc = lib.stcsp_merge_shards(arr, len(results), C.byref(h))
    if rc != 0:
        raise StcspError(rc, "merge failed")
    return h, lib.stcsp_merged_result(h).contents


def solve_to_canonical(model: Model, engine: EngineBase, adversarial: bool = False):
    """solverSolve's tail (reference src/solveralgorithm.cpp:972-997) on top of an engine."""
    res = engine.solve()
    a = engine.automaton(res)
    a.traverse()
    adv = a.adversarial(5) if adversarial else None
    a.renumber()
    return a, adv
