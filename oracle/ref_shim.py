"""ORACLE (test infrastructure) — import the reference (/root/reference) in THIS container.

Used only by tests/golden/make_golden.py to generate golden vectors; nothing on
the GPU box imports this (the reference does not travel).  Recipe: SURVEY App. C.
"""
import importlib
import sys
import types

REF = "/root/reference"


def install(cqt_cls):
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name in ("torchaudio", "torchaudio.functional", "torchaudio.transforms", "plotly", "plotly.express"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    m = types.ModuleType("cqt_nsgt_pytorch")
    m.CQT_nsgt = cqt_cls
    sys.modules["cqt_nsgt_pytorch"] = m


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(o):
    import re
    if isinstance(o, dict):
        return AttrDict({k: to_attr(v) for k, v in o.items()})
    if isinstance(o, list):
        return [to_attr(v) for v in o]
    if isinstance(o, str) and re.fullmatch(r"[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+", o):
        return float(o)
    return o


def load_args(network="cqtdiff+", exp="maestro22k_8s", tester="blind_bwe_formal_3000_opt_2", overrides=None):
    import yaml
    def rd(p):
        with open(f"{REF}/conf/{p}.yaml") as f:
            return yaml.safe_load(f)
    args = to_attr(dict(network=rd(f"network/{network}"), exp=rd(f"exp/{exp}"),
                        diff_params=rd("diff_params/edm"), tester=rd(f"tester/{tester}")))
    for k, v in (overrides or {}).items():
        node = args
        ks = k.split(".")
        for kk in ks[:-1]:
            node = node[kk]
        node[ks[-1]] = v
    return args
